"""SDF volume evaluation (ndjir_amd/extract.py) vs the oracle's step-by-step restatement of
python/extract_by_mc.py:46-74 on the same parameters."""
import numpy as np
import pytest
import torch

from oracle import graph as G
from tests.parity_utils import small_conf

pytestmark = pytest.mark.gpu


def _params(conf, gpu, perturb=True):
    from ndjir_amd import network, parameter as P
    P.clear_parameters()
    P.set_device(gpu)
    network.seed(313)
    with torch.no_grad():
        network.geometric_network(torch.zeros(4, 3, device=gpu), conf)
        if perturb:     # make the grid matter: the N(0, 1e-3) initial features barely move the SDF
            for k, p in P.get_parameters().items():
                if k.endswith("feature/F"):
                    p.mul_(100.0)
    return {k: v.detach().cpu().clone() for k, v in P.get_parameters().items()}


@pytest.mark.parametrize("variant,grid", [("default", 16), ("no_voxel", 16)])
def test_volume_matches_oracle(gpu, variant, grid):
    from ndjir_amd.extract import compute_pts_vol, compute_vol, slab_range
    conf = small_conf(grid_size=grid, variant=variant)
    params = _params(conf, gpu)
    mins, maxs, Gv = [-1.0, -0.9, -0.8], [0.7, 0.8, 1.0], 21          # anisotropic box: catches axis mix-ups
    pts_o, vol_o = G.compute_pts_vol(mins, maxs, Gv, params, conf, batch_size=1000)
    vol64 = G.compute_pts_vol(mins, maxs, Gv, {k: v.double() for k, v in params.items()}, conf)[1]
    pts, vol = compute_pts_vol(mins, maxs, Gv, conf, chunk=4000)
    np.testing.assert_array_equal(pts, pts_o)
    assert vol.shape == (Gv, Gv, Gv)
    err, ref = np.abs(vol - vol64).max(), np.abs(vol_o - vol64).max()
    assert err <= max(4 * ref, 2e-6), (err, ref)
    # vol[i, j, k] belongs to (x_i, y_j, z_k): the sphere-initialised SDF grows with |x|
    assert vol[0, Gv // 2, Gv // 2] > vol[Gv // 2, Gv // 2, Gv // 2]
    # slabs of a 3-way shard tile the volume exactly
    parts = [compute_vol(mins, maxs, Gv, conf, chunk=3000, rank=r, world=3).cpu().numpy() for r in range(3)]
    assert [p.shape[0] for p in parts] == [7, 7, 7] and slab_range(22, 0, 3) == (0, 8) and slab_range(22, 2, 3) == (15, 22)
    np.testing.assert_array_equal(np.concatenate(parts, axis=0), vol)


def test_volume_is_batch_invariant(gpu):
    """Rows are independent: the chunk size must not change a single bit."""
    from ndjir_amd.extract import compute_vol
    conf = small_conf(grid_size=16)
    _params(conf, gpu)
    a = compute_vol([-1] * 3, [1] * 3, 40, conf, chunk=1 << 20)
    b = compute_vol([-1] * 3, [1] * 3, 40, conf, chunk=12345)
    assert torch.equal(a, b)
