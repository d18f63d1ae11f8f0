"""Sampler parity.  The up-sampling round (python/sampler.py:194-240) is bit-exact between the HIP
kernel and the C oracle on identical (t, sdf): integer bin indices AND merged distances, for every
round size of the default config, ragged ray counts, rays that miss the box, flat and steep SDFs."""
import numpy as np
import pytest
import torch

from oracle import kernels as K

pytestmark = pytest.mark.gpu


def _case(R, N, seed, kind):
    rng = np.random.RandomState(seed)
    tn = (1.0 + rng.rand(R)).astype(np.float32)
    tf = (tn + 1.0 + rng.rand(R)).astype(np.float32)
    if kind == "miss":
        tn[:] = 0.0
        tf[:] = 0.0
    frac = (np.arange(N)[None, :] + rng.rand(R, N)) / N
    t = (tn[:, None] + (tf - tn)[:, None] * frac).astype(np.float32)
    if kind == "sphere":
        sdf = np.abs(t - (tn + tf)[:, None] / 2) - 0.35 + 0.01 * rng.randn(R, N)
    elif kind == "flat":
        sdf = np.full((R, N), 0.3) + 1e-4 * rng.randn(R, N)
    elif kind == "steep":
        sdf = 5.0 * (((tn + tf)[:, None] / 2) - t)
    else:
        sdf = 0.5 + 0.0 * t
    return t, sdf.astype(np.float32), tn, tf


@pytest.mark.parametrize("N,u", [(64, 0), (80, 1), (96, 2), (112, 3), (5, 0)])
@pytest.mark.parametrize("kind", ["sphere", "flat", "steep", "miss"])
@pytest.mark.parametrize("R", [1, 7, 512])
def test_importance_round_bit_exact(gpu, N, u, kind, R):
    from ndjir_amd import lib
    M = 16
    gain = 64.0 * 2 ** u
    t, sdf, tn, tf = _case(R, N, 100 + N + R, kind)
    t_ref = np.zeros((R, N + M), np.float32)
    i_ref = np.zeros((R, M), np.int32)
    K.call("sampler_importance_round", R, N, M, gain, t, sdf, tn, tf, t_ref, i_ref)
    d = lambda a: torch.from_numpy(a).to(gpu)
    t_out = torch.empty((R, N + M), device=gpu)
    idx = torch.empty((R, M), device=gpu, dtype=torch.int32)
    lib.call("sampler_importance_round", R, N, M, gain, d(t), d(sdf), d(tn), d(tf), t_out, idx)
    assert np.array_equal(idx.cpu().numpy(), i_ref), "sample indices must be bit-exact"
    assert np.array_equal(t_out.cpu().numpy().view(np.uint32), t_ref.view(np.uint32)), "merged distances bit-exact"
    assert (np.diff(t_ref, axis=1) >= 0).all()
    assert (i_ref >= 0).all() and (i_ref <= N - 1).all()


def test_sampler_end_to_end_indices(gpu):
    """Whole SamplePoints: with each side evaluating its own SDF network the inputs of a round differ by
    fp32 round-off, so indices can only differ where a CDF value sits within round-off of u."""
    from tests.parity_utils import run_oracle_step, run_product_step, small_conf
    conf = small_conf(grid_size=32, n_rays=64)
    rec_p, rec_o = {}, {}
    prod = run_product_step(conf, B=1, R=64, device=gpu, backward=False, record=rec_p)
    run_oracle_step(conf, prod["params_cpu"], prod["inputs_cpu"], backward=False, record=rec_o)
    tot = mis = 0
    for a, b in zip(rec_p["idx"], rec_o["idx"]):
        tot += b.numel()
        mis += int((a.cpu() != b).sum())
    assert mis <= 2e-3 * tot, (mis, tot)
