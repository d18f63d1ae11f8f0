"""CPU sanity of the torch restatement of the reference's Python graph (oracle/graph.py):
fp32 vs fp64 agreement of loss and every parameter gradient, layout facts the MLP weights
depend on, and sampler invariants.  (No reference numbers exist for this part: parity unpinned.)"""
import numpy as np
import pytest
import torch

from ndjir_amd.renderer import make_rand
from ndjir_amd.synthetic import make_rays
from oracle import graph as G
from tests.parity_utils import random_oracle_params, run_oracle_step, small_conf


def _inputs(conf, B, R):
    camloc, raydir, color = make_rays(B, R)
    return dict(camloc=camloc, raydir=raydir, color_gt=color, rand=make_rand(B, R, conf, "cpu"),
                cos_anneal=torch.tensor([0.6]))


def test_positional_encoding_layout():
    """[x, cos(band-fastest), sin(...)] (network.py:105-115)."""
    x = torch.tensor([[0.1, 0.2, 0.3]])
    pe = G.positional_encoding(x, 6)
    assert pe.shape == (1, 39)
    np.testing.assert_allclose(pe[0, :3], x[0])
    np.testing.assert_allclose(pe[0, 3:9], torch.cos(0.1 * 2.0 ** torch.arange(6.0)), rtol=1e-6)
    np.testing.assert_allclose(pe[0, 9:15], torch.cos(0.2 * 2.0 ** torch.arange(6.0)), rtol=1e-6)
    np.testing.assert_allclose(pe[0, 21:27], torch.sin(0.1 * 2.0 ** torch.arange(6.0)), rtol=1e-6)


@pytest.mark.parametrize("variant", ["no_voxel", "default"])
def test_fp32_matches_fp64(variant):
    torch.set_num_threads(8)
    conf = small_conf(grid_size=8, n_rays=4, variant=variant)
    p = random_oracle_params(conf)
    inp = _inputs(conf, 1, 4)
    r32 = run_oracle_step(conf, p, inp, torch.float32)
    samples = [r32["out"][k] for k in ("x_fg", "t_fg", "x_bg", "t_bg", "mask")]
    r64 = run_oracle_step(conf, p, inp, torch.float64, samples=samples)
    assert abs(float(r32["loss"]) - float(r64["loss"])) < 1e-5 * abs(float(r64["loss"]))
    for k, g in r32["grads"].items():
        g64 = r64["grads"][k]
        assert g is not None, k
        err = float((g.double() - g64).norm() / max(float(g64.norm()), 1e-30))
        assert err < 2e-3, (k, err)


def test_sampler_invariants():
    conf = small_conf(grid_size=8, n_rays=8, variant="no_voxel")
    p = random_oracle_params(conf)
    inp = _inputs(conf, 2, 8)
    rec = {}
    x_fg, t_fg, x_bg, t_bg, mask = G.sample_points(inp["camloc"], inp["raydir"], inp["rand"]["stratified_sample"],
                                                   inp["rand"]["background_sample"], p, conf, rec)
    N = conf.renderer.n_samples0 + conf.renderer.n_samples1 * conf.renderer.n_upsamples
    assert x_fg.shape == (2, 8, N, 3) and t_fg.shape == (2, 8, N + 1, 1)
    assert x_bg.shape == (2, 8, 32, 4) and t_bg.shape == (2, 8, 33, 1) and mask.shape == (2, 8, 1, 1)
    assert bool((t_fg[:, :, 1:] >= t_fg[:, :, :-1]).all()), "sorted along the ray"
    assert bool((t_bg[:, :, 1:] >= t_bg[:, :, :-1]).all())
    np.testing.assert_allclose((x_bg[..., :3] ** 2).sum(-1).sqrt(), 1.0, atol=1e-5)   # inverted sphere
    for u, idx in enumerate(rec["idx"]):
        n_in = 64 + 16 * u
        assert idx.dtype == torch.int64 and int(idx.min()) >= 0 and int(idx.max()) <= n_in - 2
        assert bool((idx[..., 1:] >= idx[..., :-1]).all()), "u is increasing, so idx is non-decreasing"


def test_empty_rays_miss_box():
    """rays that miss the box: mask 0, all foreground samples collapse onto the camera."""
    conf = small_conf(grid_size=8, n_rays=2, variant="no_voxel")
    p = random_oracle_params(conf)
    camloc = torch.tensor([[0.0, 0.0, 3.0]])
    raydir = torch.tensor([[[0.0, 0.0, 1.0], [0.0, 0.0, -1.0]]])   # away from / towards the box
    rand = make_rand(1, 2, conf, "cpu")
    x_fg, t_fg, x_bg, t_bg, mask = G.sample_points(camloc, raydir, rand["stratified_sample"],
                                                   rand["background_sample"], p, conf)
    assert mask.flatten().tolist() == [0.0, 1.0]
    assert torch.isfinite(x_fg).all() and torch.isfinite(x_bg).all()
    np.testing.assert_allclose(x_fg[0, 0], camloc.expand(x_fg.shape[2], 3))


@pytest.mark.parametrize("N,M", [(64, 16), (112, 16), (128, 32), (224, 32), (240, 16)])
def test_importance_round_c_vs_torch(N, M):
    """The C round (fixed Kogge-Stone / butterfly orders of include/ndjir_math.h, up to 256 slots) against the stock-op
    restatement of python/sampler.py:194-240: same bins except where a CDF value sits within round-off of u."""
    rng = np.random.RandomState(N + M)
    R = 64
    tn = (1.0 + rng.rand(1, R, 1, 1)).astype(np.float32)
    tf = (tn + 1.0 + rng.rand(1, R, 1, 1)).astype(np.float32)
    frac = ((np.arange(N)[None, :] + rng.rand(R, N)) / N).reshape(1, R, N, 1)
    t = (tn + (tf - tn) * frac).astype(np.float32)
    sdf = (np.abs(t - (tn + tf) / 2) - 0.35 + 0.01 * rng.randn(1, R, N, 1)).astype(np.float32)
    a = [torch.from_numpy(x) for x in (t, sdf, tn, tf)]
    t_c, i_c = G._importance_round_c(*a, 64.0, M)
    t_t, i_t = G.importance_round_torch(*a, 64.0, M)
    assert i_c.shape == i_t.shape == (1, R, M) and t_c.shape == (1, R, N + M, 1)
    mism = float((i_c != i_t).float().mean())
    assert mism <= 5e-3, mism
    assert bool((t_c[:, :, 1:] >= t_c[:, :, :-1]).all())
    same = (i_c == i_t).all(dim=2)
    np.testing.assert_allclose(t_c[same].numpy(), t_t[same].numpy(), atol=2e-5)
