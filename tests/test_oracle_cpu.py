"""CPU tests of the oracle itself (no GPU): golden vectors from the reference's numpy test
oracles, and cross-validation of the two independent restatements of the grid ops
(C from csrc/*.cu vs torch from python/grid_feature/*_composite.py) at the reference tests'
shapes, seeds and tolerances (python/grid_feature/test/test_voxel_feature.py:25-150)."""
import os

import numpy as np
import pytest
import torch

from oracle import composite as C
from oracle import kernels as K

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_oracle_exports_every_symbol():
    lib = K.lib()
    assert all(hasattr(lib, s) for s in K.symbols())


@pytest.mark.parametrize("k", [0, 1, 2])
def test_ray_aabb_golden(k):
    g = np.load(os.path.join(GOLD, "ray_aabb_intersection.npz"))
    c, r, s = g[f"c{k}_camloc"], g[f"c{k}_raydir"], float(g[f"c{k}_size"])
    B, R, _ = r.shape
    tn, tf, nh = (np.zeros((B, R, 1), np.float32) for _ in range(3))
    K.call("ray_aabb_intersection", B * R, tn, tf, nh, c, r, B, R, [-s] * 3, [s] * 3)
    # tolerances of the reference test (test_ray_aabb_intersection.py:145-147)
    np.testing.assert_allclose(tn.ravel(), g[f"c{k}_t_near"].ravel(), atol=1e-6)
    np.testing.assert_allclose(tf.ravel(), g[f"c{k}_t_far"].ravel(), atol=1e-6)
    np.testing.assert_array_equal(nh.ravel(), g[f"c{k}_n_hits"].ravel())


@pytest.mark.parametrize("k", [0, 1, 2])
def test_ray_sphere_golden(k):
    g = np.load(os.path.join(GOLD, "ray_sphere_intersection.npz"))
    c, r, s = g[f"c{k}_camloc"], g[f"c{k}_raydir"], float(g[f"c{k}_radius"])
    B, R, _ = r.shape
    tn, tf, nh = (np.zeros((B, R, 1), np.float32) for _ in range(3))
    K.call("ray_sphere_intersection", B * R, tn, tf, nh, c, r, B, R, s)
    np.testing.assert_allclose(tn.ravel(), g[f"c{k}_t_near"].ravel(), atol=1e-6)
    np.testing.assert_allclose(tf.ravel(), g[f"c{k}_t_far"].ravel(), atol=1e-6)
    np.testing.assert_array_equal(nh.ravel(), g[f"c{k}_n_hits"].ravel())


@pytest.mark.parametrize("eps", [0.0, 1e-12])
def test_sample_directions_golden(eps):
    g = np.load(os.path.join(GOLD, "sample_directions.npz"))
    for k in range(int(g["n_cases"])):
        n, ct, cp = g[f"c{k}_normal"], g[f"c{k}_cdf_the"], g[f"c{k}_cdf_phi"]
        B, R, _ = n.shape
        nt, nph = ct.shape[-1], cp.shape[-1]
        M = nt * nph
        out = np.zeros((B, R, M, 3), np.float32)
        if f"c{k}_alpha" in g:
            K.call("sample_importance_directions", B * R * M, out, n, ct, cp, g[f"c{k}_alpha"], B * R, M, nt, nph, eps)
        else:
            K.call("sample_uniform_directions", B * R * M, out, n, ct, cp, B * R, M, nt, nph, eps)
        # test_sampler.py:111
        np.testing.assert_allclose(out, g[f"c{k}_light_dirs"].reshape(out.shape), atol=1e-5)


def _comp(o, hash_cfg):
    kind = "cosine" if o.family.startswith("cosine") else "linear"

    def f(q, ft):
        if o.topo == "voxel":
            return C.lanczos_query_on_voxel(q, ft) if o.lanczos else C.query_on_voxel(q, ft, kind=kind)
        if o.topo == "triplane":
            return C.lanczos_query_on_triplane(q, ft) if o.lanczos else C.query_on_triplane(q, ft, kind=kind)
        if o.topo == "triline":
            return C.lanczos_query_on_triline(q, ft) if o.lanczos else C.query_on_triline(q, ft, kind=kind)
        G0, gf, T0, L, D = hash_cfg
        return C.query_on_voxel_hash(q, ft, G0, gf, T0, L, D, kind="lanczos" if o.lanczos else "linear")
    return f


def feature_shape(o, G, D, hash_cfg):
    if o.topo == "voxel":
        return (G, G, G, D)
    if o.topo == "triplane":
        return (3, G, G, D)
    if o.topo == "triline":
        return (3, G, D)
    return (K.hash_num_params(*hash_cfg),)


CASES = []
for fam in K.FAMILIES:
    if "hash" in fam:
        CASES += [(fam, P, None, h) for P in (2, 16) for h in ((2, 1.5, 2 ** 10, 1, 2), (4, 1.5, 2 ** 10, 4, 2))]
    else:
        CASES += [(fam, P, G, None) for P in (2, 16) for G in (2, 8)]


@pytest.mark.parametrize("family,P,G,hash_cfg", CASES)
def test_c_kernels_match_composite(family, P, G, hash_cfg):
    rng = np.random.RandomState(412)
    o = K.GridOracle(family, hash=hash_cfg)
    fs = feature_shape(o, G, 4, hash_cfg)
    q = (rng.rand(P, 3) * 2 - 1).astype(np.float32)
    f = (rng.randn(*fs) * 0.01).astype(np.float32)
    comp = _comp(o, hash_cfg)
    qt = torch.tensor(q, requires_grad=True)
    ft = torch.tensor(f, requires_grad=True)
    out_t = comp(qt, ft)
    out = o.query(q, f)
    lz = o.lanczos
    np.testing.assert_allclose(out, out_t.detach().numpy(), atol=1e-6)           # test_voxel_feature.py:64
    og = rng.randn(*out.shape).astype(np.float32)
    ogt = torch.tensor(og, requires_grad=True)
    gq_t, = torch.autograd.grad(out_t, qt, ogt, create_graph=True)
    gf_t, = torch.autograd.grad(out_t, ft, ogt, retain_graph=True)
    np.testing.assert_allclose(o.grad_feature(og, q, fs), gf_t.numpy(), atol=2e-5 if lz else 1e-6)   # :77
    np.testing.assert_allclose(o.grad_query(og, q, f), gq_t.detach().numpy(), atol=5e-6 if lz else 1e-6)  # :129
    ggq = rng.randn(P, 3).astype(np.float32)
    ggo_t, ggf_t = torch.autograd.grad(gq_t, [ogt, ft], torch.tensor(ggq))
    # reference 2nd-order tolerances: 1e-3 (linear), 5e-3 / rtol 1e-1 (lanczos)
    np.testing.assert_allclose(o.grad_query_grad_grad_output(ggq, q, f), ggo_t.numpy(), atol=1e-5)
    np.testing.assert_allclose(o.grad_query_grad_feature(ggq, og, q, fs), ggf_t.numpy(), atol=1e-3 if lz else 1e-5)


@pytest.mark.parametrize("P,G", [(2, 2), (16, 8)])
def test_voxel_second_order_extras_match_composite(P, G):
    """grad_query_grad_query, grad_feature_grad_grad_output, grad_feature_grad_query of the linear
    dense voxel grid (untested in the reference, test_voxel_feature.py:152-170 commented out)."""
    rng = np.random.RandomState(412)
    D = 4
    q = (rng.rand(P, 3) * 2 - 1).astype(np.float32)
    f = (rng.randn(G, G, G, D) * 0.01).astype(np.float32)
    og = rng.randn(P, D).astype(np.float32)
    ggq = rng.randn(P, 3).astype(np.float32)
    qt = torch.tensor(q, dtype=torch.float64, requires_grad=True)
    ft = torch.tensor(f, dtype=torch.float64, requires_grad=True)
    ogt = torch.tensor(og, dtype=torch.float64, requires_grad=True)
    out = C.query_on_voxel(qt, ft)
    gq, = torch.autograd.grad(out, qt, ogt, create_graph=True)
    gqgq, = torch.autograd.grad(gq, qt, torch.tensor(ggq, dtype=torch.float64), retain_graph=True)
    mine = np.zeros((P, 3), np.float32)
    K.call("voxel_grad_query_grad_query", P * D, mine, ggq, og, q, f, [G] * 3, D, [-1] * 3, [1] * 3)
    np.testing.assert_allclose(mine, gqgq.numpy(), atol=1e-4, rtol=1e-4)
    gf, = torch.autograd.grad(out, ft, ogt, create_graph=True)
    ggf = rng.randn(G, G, G, D).astype(np.float32)
    a, b = torch.autograd.grad(gf, [ogt, qt], torch.tensor(ggf, dtype=torch.float64))
    m1 = np.zeros((P, D), np.float32)
    K.call("voxel_grad_feature_grad_grad_output", P * D, m1, ggf, q, [G] * 3, D, [-1] * 3, [1] * 3, 0)
    np.testing.assert_allclose(m1, a.numpy(), atol=1e-5)
    m2 = np.zeros((P, 3), np.float32)
    K.call("voxel_grad_feature_grad_query", P * D, m2, ggf, og, q, [G] * 3, D, [-1] * 3, [1] * 3)
    np.testing.assert_allclose(m2, b.numpy(), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("topo", ["voxel", "triplane", "triline", "voxel_hash"])
@pytest.mark.parametrize("sym", [False, True])
def test_tv_matches_composite(topo, sym):
    """total_variation_loss tests: fwd 1e-6, bwd 1e-4 (test_total_variation_loss.py:65, 75)."""
    rng = np.random.RandomState(412)
    P, G, D = 16, 8, 4
    hc = (4, 1.5, 2 ** 10, 4, 2)
    q = (rng.rand(P, 3) * 2 - 1).astype(np.float32)
    if topo == "voxel":
        fs, sa, comp = (G, G, G, D), [[G] * 3, D], lambda qt, ft: C.tv_loss_on_voxel(qt, ft, sym_backward=sym)
        n, C_out = P * D, D
    elif topo == "triplane":
        fs, sa, comp = (3, G, G, D), [G, D], lambda qt, ft: C.tv_loss_on_triplane(qt, ft, sym_backward=sym)
        n, C_out = P * D * 3, D * 3
    elif topo == "triline":
        fs, sa, comp = (3, G, D), [G, D], lambda qt, ft: C.tv_loss_on_triline(qt, ft, sym_backward=sym)
        n, C_out = P * D * 3, D * 3
    else:
        fs, sa = (K.hash_num_params(*hc),), list(hc)
        comp = lambda qt, ft: C.tv_loss_on_voxel_hash(qt, ft, *hc, sym_backward=sym)
        n, C_out = hc[3] * P, hc[3] * hc[4]
    f = (rng.randn(*fs) * 0.01).astype(np.float32)
    name = "tv_loss_on_" + topo
    native_shape = (C_out, P) if topo == "voxel_hash" else (P, C_out)
    out = np.zeros(native_shape, np.float32)
    K.call(name, n, out, q, f, *sa, [-1] * 3, [1] * 3)
    ft = torch.tensor(f, requires_grad=True)
    out_t = comp(torch.tensor(q), ft)
    mine = out.T if topo == "voxel_hash" else out
    np.testing.assert_allclose(mine, out_t.detach().numpy(), atol=1e-6)
    og = rng.randn(P, C_out).astype(np.float32)
    gf_t, = torch.autograd.grad(out_t, ft, torch.tensor(og))
    gf = np.zeros(fs, np.float32)
    og_n = np.ascontiguousarray(og.T) if topo == "voxel_hash" else og
    K.call(name + "_backward", n, gf, og_n, q, f, *sa, [-1] * 3, [1] * 3, int(sym))
    np.testing.assert_allclose(gf, gf_t.numpy(), atol=1e-4)


def test_out_of_box_queries_agree():
    """clamping / extrapolation outside [min, max] (voxel_feature_cuda.cu:57-64)."""
    rng = np.random.RandomState(7)
    P, G, D = 64, 8, 4
    q = (rng.rand(P, 3) * 3 - 1.5).astype(np.float32)
    f = (rng.randn(G, G, G, D) * 0.01).astype(np.float32)
    o = K.GridOracle("voxel")
    out_t = C.query_on_voxel(torch.tensor(q), torch.tensor(f))
    np.testing.assert_allclose(o.query(q, f), out_t.numpy(), atol=1e-6)


def test_hash_table_layout_quirks():
    """force_align is `s + s % 8`, table sizes are computed in float (common_voxel_hash.cuh:24-43)."""
    assert K.lib().hash_force_align(10, 8) == 12 and K.lib().hash_force_align(16, 8) == 16
    assert K.lib().hash_force_align(13, 8) == 18
    for (G0, gf, T0, L, D) in [(16, 1.5, 2 ** 15, 16, 2), (4, 1.5, 2 ** 10, 4, 2), (2, 1.5, 2 ** 10, 1, 2)]:
        assert K.hash_num_params(G0, gf, T0, L, D) == C.compute_num_params(G0, gf, T0, D, L)
        for l in range(L):
            assert K.hash_grid_size(G0, gf, l) == C.compute_grid_size(G0, gf, l)


def _expf_inputs():
    rng = np.random.RandomState(3)
    x = np.concatenate([np.linspace(-87.0, 88.0, 400001), rng.uniform(-87.0, 88.0, 200000), rng.uniform(-1.0, 1.0, 100000),
                        np.asarray([0.0, -0.0, 1e-30, -1e-30, 88.0, -87.0, 100.0, -100.0, 0.6931472, -0.6931472])])
    return x.astype(np.float32)


def _ulps(y, ref64):
    return np.abs(y.astype(np.float64) - ref64) / np.spacing(ref64.astype(np.float32)).astype(np.float64)


def test_shared_expf_against_float64():
    """include/ndjir_math.h (shared by the sampler kernel and the oracle, so a defect in it would be invisible to the bit-exact
    sampler tests): ndjir_expf within 1.5 ulp of float64 exp over the clamped range, ndjir_sigmoidf within 3 ulp."""
    x = _expf_inputs()
    y = np.empty_like(x)
    K.call("math_expf", x.size, y, x, 0)
    xc = np.clip(x.astype(np.float64), -87.0, 88.0)           # the definition clamps its argument
    assert float(_ulps(y, np.exp(xc)).max()) <= 1.5
    s = np.empty_like(x)
    K.call("math_expf", x.size, s, x, 1)
    assert float(_ulps(s, 1.0 / (1.0 + np.exp(np.clip(-x.astype(np.float64), -87.0, 88.0)))).max()) <= 3.0
