"""GPU parity of every custom op: HIP (through the C ABI) vs the C oracle on the same seeded
inputs, vs the golden fixtures of the reference's own numpy test oracles, at the reference tests'
shapes/tolerances plus larger, ragged and out-of-box cases."""
import os

import numpy as np
import pytest
import torch

from oracle import kernels as K
from tests.test_oracle_cpu import CASES, feature_shape

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_ray_aabb(gpu, k):
    from ndjir_amd.intersection.ray_aabb_intersection import ray_aabb_intersection
    g = np.load(os.path.join(GOLD, "ray_aabb_intersection.npz"))
    c, r, s = g[f"c{k}_camloc"], g[f"c{k}_raydir"], float(g[f"c{k}_size"])
    tn, tf, nh = ray_aabb_intersection(T(c, gpu), T(r, gpu), [-s] * 3, [s] * 3)
    np.testing.assert_allclose(tn.cpu().numpy().ravel(), g[f"c{k}_t_near"].ravel(), atol=1e-6)
    np.testing.assert_allclose(tf.cpu().numpy().ravel(), g[f"c{k}_t_far"].ravel(), atol=1e-6)
    np.testing.assert_array_equal(nh.cpu().numpy().ravel(), g[f"c{k}_n_hits"].ravel())


@pytest.mark.parametrize("k", [0, 1, 2])
def test_ray_sphere(gpu, k):
    from ndjir_amd.intersection.ray_sphere_intersection import ray_sphere_intersection
    g = np.load(os.path.join(GOLD, "ray_sphere_intersection.npz"))
    c, r, s = g[f"c{k}_camloc"], g[f"c{k}_raydir"], float(g[f"c{k}_radius"])
    tn, tf, nh = ray_sphere_intersection(T(c, gpu), T(r, gpu), s)
    np.testing.assert_allclose(tn.cpu().numpy().ravel(), g[f"c{k}_t_near"].ravel(), atol=1e-6)
    np.testing.assert_allclose(tf.cpu().numpy().ravel(), g[f"c{k}_t_far"].ravel(), atol=1e-6)
    np.testing.assert_array_equal(nh.cpu().numpy().ravel(), g[f"c{k}_n_hits"].ravel())


def test_intersection_large_vs_oracle(gpu):
    """4096 rays incl. axis-parallel rays (inf slabs), rays that miss, camera inside the box."""
    from ndjir_amd.intersection.ray_aabb_intersection import ray_aabb_intersection
    from ndjir_amd.intersection.ray_sphere_intersection import ray_sphere_intersection
    rng = np.random.RandomState(3)
    B, R = 4, 1024
    camloc = rng.randn(B, 3).astype(np.float32)
    camloc *= np.array([3.0, 2.0, 0.5, 1.2], np.float32)[:, None] / np.linalg.norm(camloc, axis=-1, keepdims=True)
    raydir = rng.randn(B, R, 3).astype(np.float32)
    raydir[:, :8, 0] = 0.0
    raydir[:, 8:16, 1:] = 0.0
    raydir /= np.linalg.norm(raydir, axis=-1, keepdims=True)
    tn, tf, nh = (np.zeros((B, R, 1), np.float32) for _ in range(3))
    K.call("ray_aabb_intersection", B * R, tn, tf, nh, camloc, raydir, B, R, [-1.0] * 3, [1.0] * 3)
    a, b, c = ray_aabb_intersection(T(camloc, gpu), T(raydir, gpu), [-1.0] * 3, [1.0] * 3)
    np.testing.assert_array_equal(c.cpu().numpy(), nh)
    np.testing.assert_allclose(a.cpu().numpy(), tn, atol=1e-6)
    np.testing.assert_allclose(b.cpu().numpy(), tf, atol=1e-6)
    K.call("ray_sphere_intersection", B * R, tn, tf, nh, camloc, raydir, B, R, 1.0)
    a, b, c = ray_sphere_intersection(T(camloc, gpu), T(raydir, gpu), 1.0)
    # Y == 0 / Y > 0 decisions may differ by one fma rounding only at grazing rays
    agree = (c.cpu().numpy() == nh)
    assert agree.mean() > 0.999
    np.testing.assert_allclose(a.cpu().numpy()[agree], tn[agree], atol=1e-6, rtol=2e-6)
    np.testing.assert_allclose(b.cpu().numpy()[agree], tf[agree], atol=1e-6, rtol=2e-6)


@pytest.mark.parametrize("eps", [0.0, 1e-12])
def test_sample_directions(gpu, eps):
    from ndjir_amd.sampler import sample_importance_directions, sample_uniform_directions
    g = np.load(os.path.join(GOLD, "sample_directions.npz"))
    for k in range(int(g["n_cases"])):
        n, ct, cp = g[f"c{k}_normal"], g[f"c{k}_cdf_the"], g[f"c{k}_cdf_phi"]
        if f"c{k}_alpha" in g:
            out = sample_importance_directions(T(n, gpu), T(ct, gpu), T(cp, gpu), T(g[f"c{k}_alpha"], gpu), eps)
        else:
            out = sample_uniform_directions(T(n, gpu), T(ct, gpu), T(cp, gpu), eps)
        np.testing.assert_allclose(out.cpu().numpy(), g[f"c{k}_light_dirs"].reshape(out.shape), atol=1e-5)


def test_sample_directions_z_normal_is_nan_like_reference(gpu):
    """x_axis = normalize((-n_y, n_x, 0)) is NaN for a normal along +-z with eps = 0
    (inverse_transform_cuda.cu:58-60); reproduced, not silently fixed."""
    from ndjir_amd.sampler import sample_uniform_directions
    n = torch.tensor([[[0.0, 0.0, 1.0]]], device=gpu)
    out = sample_uniform_directions(n, torch.rand(1, 1, 2, device=gpu), torch.rand(1, 1, 4, device=gpu), 0.0)
    assert torch.isnan(out).any()


@pytest.mark.parametrize("family,P,hash_cfg", [("voxel_hash", 20000, (16, 1.5, 2 ** 15, 16, 2)),      # 2 table slices per level
                                               ("voxel_hash", 40000, (16, 1.5, 2 ** 17, 6, 2)),       # up to 8 slices
                                               ("voxel_hash", 9000, (16, 1.5, 2 ** 13, 8, 4)),        # one slice, 4 channels
                                               ("lanczos_voxel_hash", 3000, (16, 1.5, 2 ** 15, 8, 2))])
def test_hash_scatter_through_lds_table(gpu, family, P, hash_cfg):
    """grad_feature / grad_query_grad_feature of the hash grids at point counts where the scatter runs through the LDS image of
    a level's table (csrc/grid.hip k_scatter_hash_lds) instead of one global atomic per tap and channel."""
    _family_check(gpu, family, P, None, hash_cfg, fine=True)


@pytest.mark.parametrize("family,P,G,D", [("triplane", 20000, 256, 8), ("triplane", 30000, 200, 4), ("cosine_triplane", 20000, 128, 8),
                                          ("lanczos_triplane", 17000, 96, 4), ("lanczos_voxel", 17000, 40, 4)])
def test_triplane_scatter_large_point_sets(gpu, family, P, G, D):
    """grad_feature / grad_query_grad_feature of the tri-plane / Lanczos-voxel families at tens of thousands of points (many
    passes of the workgroup-aggregated scatter, csrc/grid.hip k_scatter_agg), queries beyond the box (clamped to border cells)."""
    _family_check(gpu, family, P, G, None, D=D, fine=True)      # (G >= 96: the tolerance class of the fine hash levels)


@pytest.mark.parametrize("family,P,G,D", [("triline", 20000, 128, 8), ("cosine_triline", 9000, 64, 8), ("lanczos_triline", 9000, 64, 4)])
def test_triline_scatter_through_lds_image(gpu, family, P, G, D):
    """grad_feature / grad_query_grad_feature of the tri-line families at point counts (>= 64 G) where a workgroup accumulates a
    whole line in LDS (csrc/grid.hip k_scatter_line_lds)."""
    _family_check(gpu, family, P, G, None, D=D, fine=True)


def test_triplane_scatter_every_point_in_one_cell(gpu):
    """Every point in one cell (what clamped out-of-box samples do to a border cell): thousands of contributions merge in the
    workgroups' LDS tables; the result equals the sum formed in float64."""
    from ndjir_amd.grid_feature import _core
    P, G, D = 3 * 4096 + 777, 128, 8
    q = torch.full((P, 3), 0.3, device=gpu) + torch.rand(P, 3, device=gpu) * 1e-4
    og = torch.randn(P, 3 * D, device=gpu)
    f = torch.zeros(3, G, G, D, device=gpu, requires_grad=True)
    gf = _core.grad_feature("triplane", og, q, f)
    # reference: per plane, the four corner weights of each point times its output gradient (float64)
    o = K.GridOracle("triplane")
    ref = o.grad_feature(og.cpu().numpy(), q.cpu().numpy(), (3, G, G, D))
    scale = float(np.abs(ref).max())
    np.testing.assert_allclose(gf.detach().cpu().numpy(), ref, atol=2e-5 * scale)


@pytest.mark.parametrize("family,G,D", [("voxel", 48, 4), ("triplane", 96, 8), ("lanczos_voxel", 40, 4), ("voxel", 24, 8)])
def test_scatter_of_clustered_and_spread_points_mixed(gpu, family, G, D):
    """Every other point in one small cluster, the rest spread over the grid, interleaved: inside one pass of k_scatter_agg /
    k_scatter_lanczos_voxel some runs merge in the LDS table and others find it full and take the overflow list (resp. go to
    memory directly) -- both paths of the same pass against the float64 oracle, grad_feature and grad_query_grad_feature."""
    from ndjir_amd.grid_feature import _core
    rng = np.random.RandomState(11)
    P = 12000
    q = (rng.rand(P, 3) * 2.2 - 1.1).astype(np.float32)
    q[::2] = (0.31 + rng.rand(P // 2, 3) * 0.02).astype(np.float32)
    o = K.GridOracle(family)
    fs = feature_shape(o, G, D, None)
    f = (rng.randn(*fs) * 0.01).astype(np.float32)
    C = o.query(q[:1], f).shape[1]
    og = rng.randn(P, C).astype(np.float32)
    qd, fd, ogd = T(q, gpu).requires_grad_(True), T(f, gpu).requires_grad_(True), T(og, gpu).requires_grad_(True)
    gf = _core.grad_feature(family, ogd, qd, fd)
    ref = o.grad_feature(og, q, fs)
    np.testing.assert_allclose(gf.detach().cpu().numpy(), ref, atol=(5e-5 if o.lanczos else 5e-6) * np.abs(ref).max())
    ggq = rng.randn(P, 3).astype(np.float32)
    gq = _core.grad_query(family, ogd, qd, fd)
    g_f, = torch.autograd.grad(gq, [fd], T(ggq, gpu))
    ref2 = o.grad_query_grad_feature(ggq, og, q, fs)
    np.testing.assert_allclose(g_f.cpu().numpy(), ref2, atol=(2e-3 if o.lanczos else 2e-5) * np.abs(ref2).max())


def _hip_family(family):
    from ndjir_amd.grid_feature import _core
    return _core


@pytest.mark.parametrize("family,P,G,hash_cfg", CASES + [("voxel", 4097, 33, None), ("triplane", 1000, 64, None),
                                                         ("voxel_hash", 777, None, (16, 1.5, 2 ** 15, 16, 2))])
def test_grid_family_vs_oracle(gpu, family, P, G, hash_cfg):
    _family_check(gpu, family, P, G, hash_cfg)


@pytest.mark.parametrize("family,P,G", [("triplane", 3000, 64), ("triline", 3000, 64), ("voxel", 3000, 16), ("lanczos_triplane", 200, 16),
                                        ("lanczos_voxel", 700, 12), ("cosine_voxel", 3000, 16)])
def test_grid_family_eight_channels(gpu, family, P, G):
    """feature_size = 8 (config/triplaneline.yaml): two float4 chunks per cell on the LDS-aggregated scatter path (k_scatter_agg:
    a run of 2 / 4 cells is 64 / 128 bytes = 1 - 3 table blocks; the Lanczos voxel's 16 lanes per point leave 8 points per pass),
    the per-point Lanczos gathers (the lane-per-channel kernels are D = 4 only)."""
    _family_check(gpu, family, P, G, None, D=8)


def _family_check(gpu, family, P, G, hash_cfg, D=4, fine=None):
    from ndjir_amd.grid_feature import _core
    rng = np.random.RandomState(412)
    o = K.GridOracle(family, hash=hash_cfg)
    fs = feature_shape(o, G, D, hash_cfg)
    # larger dense cases include out-of-box queries (extrapolation, clamped cells); fine hash levels
    # would extrapolate with coefficients ~1e2 there (ill-conditioned), so those stay inside the box
    lo, hi = (-1.0, 1.0) if (P <= 16 or hash_cfg is not None) else (-1.2, 1.2)
    q = (rng.rand(P, 3) * (hi - lo) + lo).astype(np.float32)
    f = (rng.randn(*fs) * 0.01).astype(np.float32)
    lz = o.lanczos
    qd = T(q, gpu).requires_grad_(True)
    fd = T(f, gpu).requires_grad_(True)
    out = _core.query(family, qd, fd, hcfg=hash_cfg)
    ref = o.query(q, f)
    # the finest hash levels (G ~ 7000) resolve the cell fraction to ~1e-4 in fp32: 1-2 ulp of the
    # continuous coordinate moves a coefficient by ~1e-3
    fine = (hash_cfg is not None and hash_cfg[3] > 8) if fine is None else fine
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=2e-5 if fine else (2e-6 if lz else 1e-6))
    og = rng.randn(*ref.shape).astype(np.float32)
    ogd = T(og, gpu).requires_grad_(True)
    gq = _core.grad_query(family, ogd, qd, fd, hcfg=hash_cfg)
    gq_ref = o.grad_query(og, q, f)
    scale = max(1.0, np.abs(gq_ref).max())
    np.testing.assert_allclose(gq.detach().cpu().numpy(), gq_ref, atol=(1e-3 if fine else (2e-5 if lz else 2e-6)) * scale, rtol=1e-2 if fine else 1e-7)
    gf = _core.grad_feature(family, ogd, qd, fd, hcfg=hash_cfg)
    gf_ref = o.grad_feature(og, q, fs)
    np.testing.assert_allclose(gf.detach().cpu().numpy(), gf_ref, atol=(2e-3 if fine else (5e-5 if lz else 5e-6)) * max(1.0, np.abs(gf_ref).max()))
    ggq = rng.randn(P, 3).astype(np.float32)
    g_go, g_f = torch.autograd.grad(gq, [ogd, fd], T(ggq, gpu))
    ggo_ref = o.grad_query_grad_grad_output(ggq, q, f)
    np.testing.assert_allclose(g_go.cpu().numpy(), ggo_ref, atol=(2e-3 if fine else (2e-5 if lz else 2e-6)) * max(1.0, np.abs(ggo_ref).max()), rtol=1e-2 if fine else 1e-7)
    gqgf_ref = o.grad_query_grad_feature(ggq, og, q, fs)
    np.testing.assert_allclose(g_f.cpu().numpy(), gqgf_ref, atol=(2e-3 if (lz or fine) else 2e-5) * max(1.0, np.abs(gqgf_ref).max()))


def test_grid_empty_batch(gpu):
    from ndjir_amd.grid_feature import _core
    q = torch.zeros((0, 3), device=gpu)
    f = torch.randn(4, 4, 4, 4, device=gpu)
    assert _core.query("voxel", q, f).shape == (0, 4)


def test_voxel_double_backward_through_autograd(gpu):
    """The reference's pattern-A test: sum(grad_query^2).backward() (test_voxel_feature.py:85-150),
    here against fp64 torch autograd of the composite restatement."""
    from ndjir_amd.grid_feature import grad as nn_grad
    from ndjir_amd.grid_feature.voxel_feature import query_on_voxel
    from oracle import composite as C
    rng = np.random.RandomState(412)
    P, G, D = 64, 8, 4
    q = (rng.rand(P, 3) * 2 - 1).astype(np.float32)
    f = (rng.randn(G, G, G, D) * 0.01).astype(np.float32)
    w = rng.randn(P, D).astype(np.float32)
    qd, fd = T(q, gpu).requires_grad_(True), T(f, gpu).requires_grad_(True)
    out = (query_on_voxel(qd, fd, [-1] * 3, [1] * 3) * T(w, gpu)).sum()
    gq = nn_grad([out], [qd])[0]
    (gq ** 2).sum().backward(inputs=[fd])
    qt = torch.tensor(q, dtype=torch.float64, requires_grad=True)
    ft = torch.tensor(f, dtype=torch.float64, requires_grad=True)
    o64 = (C.query_on_voxel(qt, ft) * torch.tensor(w, dtype=torch.float64)).sum()
    g64, = torch.autograd.grad(o64, qt, create_graph=True)
    (g64 ** 2).sum().backward(inputs=[ft])
    np.testing.assert_allclose(gq.detach().cpu().numpy(), g64.detach().numpy(), atol=1e-6)
    np.testing.assert_allclose(fd.grad.cpu().numpy(), ft.grad.numpy(), atol=1e-3)   # reference tolerance :148-150
    np.testing.assert_allclose(fd.grad.cpu().numpy(), ft.grad.numpy(), atol=2e-5 * float(ft.grad.abs().max()) + 1e-7)


def test_voxel_second_order_extras(gpu):
    from ndjir_amd import lib
    rng = np.random.RandomState(412)
    P, G, D = 300, 8, 4
    q = (rng.rand(P, 3) * 2.4 - 1.2).astype(np.float32)
    f = (rng.randn(G, G, G, D) * 0.01).astype(np.float32)
    og = rng.randn(P, D).astype(np.float32)
    ggq = rng.randn(P, 3).astype(np.float32)
    ggf = rng.randn(G, G, G, D).astype(np.float32)
    ref = np.zeros((P, 3), np.float32)
    K.call("voxel_grad_query_grad_query", P * D, ref, ggq, og, q, f, [G] * 3, D, [-1] * 3, [1] * 3)
    out = torch.zeros(P, 3, device=gpu)
    lib.call("voxel_feature_grad_query_grad_query", P * D, out, T(ggq, gpu), T(og, gpu), T(q, gpu), T(f, gpu),
             [G] * 3, D, [-1] * 3, [1] * 3, 0, 1)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-5 * max(1.0, np.abs(ref).max()))
    ref = np.zeros((P, D), np.float32)
    K.call("voxel_grad_feature_grad_grad_output", P * D, ref, ggf, q, [G] * 3, D, [-1] * 3, [1] * 3, 0)
    out = torch.zeros(P, D, device=gpu)
    lib.call("voxel_feature_grad_feature_grad_grad_output", P * D, out, T(ggf, gpu), T(q, gpu), [G] * 3, D,
             [-1] * 3, [1] * 3, 0, 0)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-6 * max(1.0, np.abs(ref).max()))
    ref = np.zeros((P, 3), np.float32)
    K.call("voxel_grad_feature_grad_query", P * D, ref, ggf, og, q, [G] * 3, D, [-1] * 3, [1] * 3)
    out = torch.zeros(P, 3, device=gpu)
    lib.call("voxel_feature_grad_feature_grad_query", P * D, out, T(ggf, gpu), T(og, gpu), T(q, gpu), [G] * 3, D,
             [-1] * 3, [1] * 3, 0, 1)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-5 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("topo,D", [("voxel", 4), ("triplane", 4), ("triline", 4), ("voxel_hash", 4), ("voxel", 8), ("triplane", 8),
                                    ("triline", 8)])
@pytest.mark.parametrize("sym", [False, True])
def test_tv_loss(gpu, topo, sym, D):
    from ndjir_amd.grid_feature import _core
    rng = np.random.RandomState(412)
    P, G = 500, 8
    hc = (4, 1.5, 2 ** 10, 4, 2)
    q = (rng.rand(P, 3) * 2.4 - 1.2).astype(np.float32)
    if topo == "voxel":
        fs, sa, n, C_out = (G, G, G, D), [[G] * 3, D], P * D, D
    elif topo in ("triplane", "triline"):
        fs = (3, G, G, D) if topo == "triplane" else (3, G, D)
        sa, n, C_out = [G, D], P * D * 3, D * 3
    else:
        fs, sa, n, C_out = (K.hash_num_params(*hc),), list(hc), hc[3] * P, hc[3] * hc[4]
    f = (rng.randn(*fs) * 0.01).astype(np.float32)
    hcfg = hc if topo == "voxel_hash" else None
    native = (C_out, P) if topo == "voxel_hash" else (P, C_out)
    ref = np.zeros(native, np.float32)
    K.call("tv_loss_on_" + topo, n, ref, q, f, *sa, [-1] * 3, [1] * 3)
    ref = ref.T if topo == "voxel_hash" else ref
    fd = T(f, gpu).requires_grad_(True)
    out = _core.tv_loss(topo, T(q, gpu), fd, sym_backward=sym, hcfg=hcfg)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=1e-6)
    og = rng.randn(P, C_out).astype(np.float32)
    gf_ref = np.zeros(fs, np.float32)
    og_n = np.ascontiguousarray(og.T) if topo == "voxel_hash" else og
    K.call("tv_loss_on_" + topo + "_backward", n, gf_ref, og_n, q, f, *sa, [-1] * 3, [1] * 3, int(sym))
    gf, = torch.autograd.grad(out, fd, T(og, gpu))
    np.testing.assert_allclose(gf.cpu().numpy(), gf_ref, atol=1e-4 * max(1.0, np.abs(gf_ref).max()))


def test_grad_buffer_accumulates_in_place(gpu):
    from ndjir_amd.grid_feature import _core, set_grad_buffer
    rng = np.random.RandomState(1)
    P, G, D = 256, 8, 4
    q = T((rng.rand(P, 3) * 2 - 1).astype(np.float32), gpu)
    f = T((rng.randn(G, G, G, D) * 0.01).astype(np.float32), gpu).requires_grad_(True)
    w = T(rng.randn(P, D).astype(np.float32), gpu)
    dense, = torch.autograd.grad((_core.query("voxel", q, f) * w).sum(), f)
    buf = torch.zeros_like(f)
    set_grad_buffer(f, buf)
    try:
        (_core.query("voxel", q, f) * w).sum().backward()
        assert f.grad is None
        np.testing.assert_allclose(buf.cpu().numpy(), dense.cpu().numpy(), atol=1e-6)
    finally:
        set_grad_buffer(f, None)


@pytest.mark.parametrize("family,G,D", [("voxel", 64, 4), ("lanczos_voxel", 48, 4), ("triplane", 128, 8), ("voxel", 32, 8)])
@pytest.mark.parametrize("offset", [0, 4, 12, 1])
def test_scatter_of_ray_samples_into_a_slice_of_a_flat_buffer(gpu, family, G, D, offset):
    """The aggregated scatters (csrc/grid.hip k_scatter_agg, k_scatter_lanczos_voxel) key their LDS tables by the 64-byte
    blocks of MEMORY the gradient tensor covers, and a training step's gradient tensor is a slice of the flat gradient
    buffer: bases 0 / 16 / 48 bytes into a block and one that is not 16-byte aligned (table blocks relative to the base)
    give the float64 oracle's sums, on ray-ordered samples (runs of points inside one cell, shared stencil blocks: what
    the tables and the per-lane run sums merge) that leave the box at both ends (clamped border cells)."""
    from ndjir_amd.grid_feature import _core, set_grad_buffer
    rng = np.random.RandomState(7)
    R, N = 96, 128
    o_ = rng.randn(R, 3); o_ = 1.6 * o_ / np.linalg.norm(o_, axis=1, keepdims=True)
    d_ = rng.rand(R, 3) * 0.8 - 0.4 - o_; d_ /= np.linalg.norm(d_, axis=1, keepdims=True)
    t = np.sort(np.concatenate([rng.rand(R, N // 2) * 2.4, 1.5 + rng.rand(R, N // 2) * 0.05], axis=1), axis=1)   # coarse + a dense band
    q = (o_[:, None] + d_[:, None] * t[..., None]).reshape(-1, 3).astype(np.float32)
    P = q.shape[0]
    o = K.GridOracle(family)
    fs = feature_shape(o, G, D, None)
    f = T((rng.randn(*fs) * 0.01).astype(np.float32), gpu).requires_grad_(True)
    og = rng.randn(P, o.query(q[:1], np.zeros(fs, np.float32)).shape[1]).astype(np.float32)
    ref = o.grad_feature(og, q, fs)
    flat = torch.zeros(f.numel() + 64, device=gpu)
    assert flat.data_ptr() % 64 == 0
    buf = flat[offset:offset + f.numel()].view(fs)
    set_grad_buffer(f, buf)
    try:
        assert _core.grad_feature(family, T(og, gpu), T(q, gpu), f) is None
        torch.cuda.synchronize()
        np.testing.assert_allclose(buf.cpu().numpy(), ref, atol=(5e-5 if o.lanczos else 5e-6) * max(1.0, np.abs(ref).max()))
        assert float(flat[:offset].abs().sum()) == 0.0 and float(flat[offset + f.numel():].abs().sum()) == 0.0   # nothing outside
    finally:
        set_grad_buffer(f, None)


def test_squareplus(gpu):
    from ndjir_amd.activation.squareplus import squareplus
    x = torch.randn(1000, device=gpu, requires_grad=True)
    y = squareplus(x, 4.0)
    np.testing.assert_allclose(y.detach().cpu().numpy(), 0.5 * (x + torch.sqrt(x * x + 4)).detach().cpu().numpy(), atol=1e-6)
    y.sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), (0.5 * (1 + x / torch.sqrt(x * x + 4))).detach().cpu().numpy(), atol=1e-6)


@pytest.mark.parametrize("family,interp", [("voxel", "linear"), ("cosine_voxel", "cosine"), ("lanczos_voxel", "lanczos")])
def test_zero_touched_rearms_the_grad_buffer(gpu, family, interp):
    """Everything the voxel backward kernels scatter (grad_feature, the second-order
    grad_query_grad_feature, TV backward incl. sym_backward) lies inside the cells zero_touched clears."""
    from ndjir_amd.grid_feature import _core, set_grad_buffer, zero_touched
    rng = np.random.RandomState(7)
    P, G, D = 3000, 16, 4
    q = T((rng.rand(P, 3) * 2.4 - 1.2).astype(np.float32), gpu)          # incl. out-of-range queries
    f = T((rng.randn(G, G, G, D) * 0.01).astype(np.float32), gpu).requires_grad_(True)
    w = T(rng.randn(P, D).astype(np.float32), gpu)
    buf = torch.zeros_like(f)
    set_grad_buffer(f, buf)
    try:
        qg = q.clone().requires_grad_(True)
        out = _core.query(family, qg, f)
        (gq,) = _core.grad([out], [qg], [w])                             # differentiable d out / d query
        loss = (out * w).sum() + (gq ** 2).sum() + _core.tv_loss("voxel", q, f, sym_backward=True).sum()
        loss.backward()
        assert int((buf != 0).sum()) > 0
        zero_touched(buf, q, interp=interp)
        assert int((buf != 0).sum()) == 0
    finally:
        set_grad_buffer(f, None)


@pytest.mark.parametrize("mode", ["xy", "pixel_index"])
def test_generate_raydir_camloc_device(gpu, mode):
    """Device ray generation vs the golden outputs of the reference's python/helper.py:44-73 (float64 numpy, which
    the reference then assigns to fp32 variables): equal after the same rounding to fp32, up to one ulp."""
    from ndjir_amd.helper import generate_raydir_camloc_device
    g = np.load(os.path.join(GOLD, "generate_raydir_camloc.npz"))
    pose, K = T(g["pose"], gpu), T(g["intrinsic"], gpu)
    if mode == "xy":
        rd, cl = generate_raydir_camloc_device(pose, K, xy=T(g["xy"].astype(np.float32), gpu))
    else:
        W = int(g["W"])
        idx = (g["xy"][..., 1] * W + g["xy"][..., 0]).astype(np.int32)
        rd, cl = generate_raydir_camloc_device(pose, K, pixel_index=T(idx, gpu), width=W)
    want = g["raydir"].astype(np.float32)
    got = rd.cpu().numpy()
    assert np.abs(got - want).max() <= 1.2e-7
    assert (got != want).mean() < 0.01            # only double-rounding ties may differ
    np.testing.assert_array_equal(cl.cpu().numpy(), g["camloc"].astype(np.float32))
    np.testing.assert_allclose(np.linalg.norm(got, axis=-1), 1.0, atol=2e-7)


def test_sparse_row_exchange_kernels(gpu):
    """pack_rows / clear_bitmap / apply / zero (csrc/grid.hip, csrc/sparse_rows.hip) on two simulated ranks: after the
    exchange both buffers equal the dense sum; the packed lists hold every non-zero touched cell exactly once; zeroing the
    listed rows re-arms the buffers completely; an undersized list reports its overflow through the count."""
    from ndjir_amd import lib
    from ndjir_amd.distributed import voxel_cell_ids
    G, D, P = 48, 4, 3000
    gen = torch.Generator(device=gpu).manual_seed(5)
    bufs, qs, packed = [], [], []
    for rank in range(2):
        q = [(torch.rand(P, 3, device=gpu, generator=gen) * 1.9 - 0.95), (torch.rand(P // 2, 3, device=gpu, generator=gen) * 1.9 - 0.95)]
        buf = torch.zeros(G, G, G, D, device=gpu)
        for x in q:      # a gradient as the scatter kernels leave it: only touched cells, some of them exactly zero
            go = torch.randn(x.shape[0], D, device=gpu, generator=gen) * (torch.rand(x.shape[0], 1, device=gpu, generator=gen) > 0.3)
            lib.call("voxel_feature_grad_feature", x.shape[0] * D, buf, go.contiguous(), x.contiguous(), [G] * 3, D, [-1] * 3, [1] * 3, 0, 1)
        bufs.append(buf)
        qs.append(q)
    dense = bufs[0] + bufs[1]
    cap = 1 << 16
    ids = torch.full((2, cap), -1, dtype=torch.int32, device=gpu)
    rows = torch.zeros(2, cap, D, device=gpu)
    counts = torch.zeros(2, dtype=torch.int32, device=gpu)
    bitmap = torch.zeros((G ** 3 + 31) // 32, dtype=torch.int32, device=gpu)
    for rank in range(2):
        cnt = torch.zeros(1, dtype=torch.int32, device=gpu)
        for x in qs[rank]:
            lib.call("voxel_feature_pack_rows", x.shape[0], bufs[rank], x.contiguous(), [G] * 3, D, [-1] * 3, [1] * 3, bitmap, ids[rank],
                     rows[rank], cnt, cap)
        lib.call("sparse_rows_clear_bitmap", ids[rank], cnt, cap, bitmap)
        assert int(bitmap.abs().sum()) == 0
        counts[rank] = cnt[0]
        n = int(cnt)
        got = ids[rank, :n].long()
        flat = bufs[rank].view(-1, D)
        nonzero = torch.nonzero((flat != 0).any(dim=1)).reshape(-1)
        assert n == nonzero.numel() and torch.equal(torch.sort(got).values, nonzero)          # each non-zero row once
        touched = torch.unique(torch.cat([voxel_cell_ids(x, [G] * 3) for x in qs[rank]]))
        assert bool(torch.isin(got, touched).all())
        assert torch.equal(rows[rank, :n], flat[got])
    for rank in range(2):
        lib.call("sparse_rows_apply", ids, rows, counts, 2, cap, cap, rank, bufs[rank], D)
        scale = float(dense.abs().max())
        assert float((bufs[rank] - dense).abs().max()) <= 1e-6 * scale
    limit = torch.full((1,), cap, dtype=torch.int32, device=gpu)
    for rank in range(2):
        lib.call("sparse_rows_zero", ids, counts, 2, cap, limit, rank, None, None, bufs[rank], D)
        assert float(bufs[rank].abs().max()) == 0.0
    # a communicated size below a rank's count: only the first `limit` rows of the other rank arrive, the flag is raised,
    # and zeroing with the rank's own full list still clears everything it wrote
    m = 4096
    flag = torch.zeros(1, dtype=torch.int32, device=gpu)
    stats = torch.zeros(2, dtype=torch.int32, device=gpu)
    lib.call("sparse_rows_overflow", counts, 2, cap, flag, stats, 1)
    assert int(flag) == 0 and stats.tolist() == [int(counts.max()), 0]
    lib.call("sparse_rows_overflow", counts, 2, m, flag, stats, 1)
    assert int(flag) == 1 and stats.tolist() == [int(counts.max()), 1]      # running maximum of the lists, overflowing exchanges
    buf = torch.zeros(G, G, G, D, device=gpu)
    lib.call("sparse_rows_apply", ids, rows, counts, 2, cap, m, 0, buf, D)
    want = torch.zeros(G ** 3, D, device=gpu)
    want[ids[1, :m].long()] = rows[1, :m]
    assert torch.equal(buf.view(-1, D), want)
    limit.fill_(m)
    own_cnt = counts[0:1].clone()
    buf.view(-1, D)[ids[0, :int(counts[0])].long()] = 1.0
    packed = ids[:, :m].contiguous()           # the communicated lists are packed with row stride `limit`
    lib.call("sparse_rows_zero", packed, counts, 2, cap, limit, 0, ids[0], own_cnt, buf, D)
    assert float(buf.abs().max()) == 0.0
    # overflow: the count runs past the capacity, nothing is written beyond it
    buf = dense.clone()
    small = 100
    ids2 = torch.full((small + 8,), -7, dtype=torch.int32, device=gpu)
    rows2 = torch.zeros(small + 8, D, device=gpu)
    cnt = torch.zeros(1, dtype=torch.int32, device=gpu)
    x = qs[0][0].contiguous()
    lib.call("voxel_feature_pack_rows", x.shape[0], buf, x, [G] * 3, D, [-1] * 3, [1] * 3, bitmap, ids2, rows2, cnt, small)
    assert int(cnt) > small and bool((ids2[small:] == -7).all()) and bool((ids2[:small] >= 0).all())


@pytest.mark.parametrize("mode", ["uniform", "patch", "mask"])
def test_device_data_feed(gpu, mode):
    """ndjir_amd.dataset.IDRRaySource on the GPU (scene resident on the device, rays generated there) delivers the golden
    colours / masks of the reference's data source and rays equal to the reference's host computation."""
    from ndjir_amd import config
    from ndjir_amd.dataset import IDRRaySource
    from ndjir_amd.helper import generate_raydir_camloc
    g = np.load(os.path.join(GOLD, "pixel_sampling.npz"))
    M, W, R = int(g["M"]), int(g["W"]), int(g["R"])
    ov = {"uniform": [], "patch": ["train.patch_ray_sampling=true"], "mask": ["train.mask_ray_sample_ratio=0.25"]}[mode]
    conf = config.load("default", [f"train.n_rays={R}", "train.patch_ray_sampling=false", "train.mask_ray_sample_ratio=0"] + ov)
    gr = np.load(os.path.join(GOLD, "generate_raydir_camloc.npz"))
    K, poses = gr["intrinsic"][:M], gr["pose"][:M]
    src = IDRRaySource(g["images"], g["masks"], K, poses, conf, rng=np.random.RandomState(313), device=gpu)
    color, mask, raydir, camloc = src.next_batch(M)
    assert color.is_cuda and raydir.is_cuda
    for pos in range(M):
        np.testing.assert_array_equal(color[pos].cpu().numpy(), g[f"{mode}_{pos}_color"])
        np.testing.assert_array_equal(mask[pos].cpu().numpy(), g[f"{mode}_{pos}_mask"].astype(np.float32))
    xy = np.stack([g[f"{mode}_{pos}_xy"] for pos in range(M)])
    rd, cl = generate_raydir_camloc(poses, K, xy)
    assert np.abs(raydir.cpu().numpy() - rd.astype(np.float32)).max() <= 1.2e-7
    np.testing.assert_array_equal(camloc.cpu().numpy(), cl.astype(np.float32))
    # the epoch wraps: M more positions come from a fresh reset()
    c2 = src.next_batch(M)[0]
    assert c2.shape == color.shape


@pytest.mark.parametrize("family,D", [("voxel", 4), ("cosine_voxel", 4), ("lanczos_voxel", 4), ("triplane", 8), ("lanczos_triplane", 4),
                                      ("triline", 8), ("cosine_triline", 4)])
def test_grid_pack_rows_lists_every_nonzero_row_once(gpu, family, D):
    """Sparse exchange, sending side (no reference counterpart): after a grad_feature scatter, ndjir_grid_pack_rows lists
    exactly the non-zero rows of the buffer (each once, with its values) when walked with the same query points -- for
    every dense family and interpolation; applying the list to a zero buffer reproduces the gradient, and
    ndjir_sparse_rows_zero (own rows from the local list) re-arms the buffer."""
    from ndjir_amd import lib
    from ndjir_amd.distributed import _INTERP, _TAPS, _TOPO
    G, P = 24, 700
    gen = torch.Generator(device=gpu).manual_seed(5)
    q = (torch.rand(P, 3, device=gpu, generator=gen) * 2.4 - 1.2)            # some points outside the box (clamped stencils)
    interp_name, _, topo_name = family.rpartition("_")
    topo, interp = _TOPO[topo_name], _INTERP[interp_name]
    shape = (G, G, G, D) if topo == 0 else (3, G, G, D) if topo == 1 else (3, G, D)
    C = D if topo == 0 else 3 * D
    go = torch.randn(P, C, device=gpu, generator=gen)
    go[::7] = 0.0                                                             # points without gradient: their rows stay zero
    gf = torch.zeros(shape, device=gpu)
    gs = [G, G, G]
    if topo == 0:
        lib.call(f"{family}_feature_grad_feature", P * C, gf, go, q, gs, D, [-1] * 3, [1] * 3, 0, 1)
    else:
        lib.call(f"{family}_feature_grad_feature", P * C, gf, go, q, G, D, [-1] * 3, [1] * 3, 0, 1)
    cells = gf.numel() // D
    sub, nd = (1, 3) if topo == 0 else (3, 2 if topo == 1 else 1)
    cap = min(cells, P * sub * _TAPS[interp] ** nd)
    bitmap = torch.zeros((cells + 31) // 32, dtype=torch.int32, device=gpu)
    ids = torch.full((cap,), -1, dtype=torch.int32, device=gpu)
    rows = torch.zeros((cap, D), device=gpu)
    count = torch.zeros(1, dtype=torch.int32, device=gpu)
    lib.call("grid_pack_rows", topo, interp, P, gf, q, gs, D, [-1] * 3, [1] * 3, bitmap, ids, rows, count, cap)
    n = int(count)
    flat = gf.view(-1, D)
    want = torch.nonzero((flat != 0).any(dim=1)).reshape(-1)
    got = ids[:n].long().sort().values
    assert torch.equal(got, want), (n, want.numel())
    assert torch.equal(rows[:n], flat[ids[:n].long()])
    lib.call("sparse_rows_clear_bitmap", ids, count, cap, bitmap)
    assert int(bitmap.abs().sum()) == 0
    # receiving side: a "world" of 2 whose other rank sent this list
    ids_all = torch.stack([torch.zeros_like(ids), ids])
    rows_all = torch.stack([torch.zeros_like(rows), rows])
    counts = torch.tensor([0, n], dtype=torch.int32, device=gpu)
    limit = max(1, min(cap, n))
    recv = torch.zeros_like(gf)
    lib.call("sparse_rows_apply", ids_all, rows_all, counts, 2, cap, limit, 0, recv, D)
    assert torch.equal(recv, gf)
    flag = torch.zeros(1, dtype=torch.int32, device=gpu)
    lib.call("sparse_rows_overflow", counts, 2, limit, flag, None, 1)
    assert int(flag) == 0
    if n > 1:
        lib.call("sparse_rows_overflow", counts, 2, n - 1, flag, None, 1)
        assert int(flag) == 1
    lim_dev = torch.tensor([limit], dtype=torch.int32, device=gpu)
    packed = ids_all[:, :limit].contiguous()   # the communicated lists are packed with row stride `limit`
    lib.call("sparse_rows_zero", packed, counts, 2, cap, lim_dev, 0, None, None, recv, D)
    assert float(recv.abs().max()) == 0.0
    lib.call("sparse_rows_zero", packed, torch.zeros_like(counts), 2, cap, lim_dev, 0, ids, count, gf, D)      # own list only
    assert float(gf.abs().max()) == 0.0


def test_geometric_glue_kernels(gpu):
    """csrc/geo.hip against their torch spelling: the chain input encoding, the normal from the sdf chain's input gradient
    (with the packed sample inputs Z), the backward begin, g-bar_0, column copies and the inverse squared distance."""
    from ndjir_amd import lib
    from ndjir_amd.mlp import _Strided
    from ndjir_amd.network import _ColumnView
    gen = torch.Generator(device=gpu).manual_seed(3)
    P, M, C0, C1, D = 1000, 6, 4, 5, 16
    r = lambda *s: torch.randn(*s, device=gpu, generator=gen)
    x, f0, f1 = r(P, 3) * 0.7, r(P, C0), r(P, C1)
    K0 = 3 + 6 * M + C0 + C1
    e = torch.empty(P, K0, device=gpu)
    lib.call("geo_encode", P, M, x, 2, [f0, f1], [C0, C1], e, K0)
    bands = 2.0 ** torch.arange(M, device=gpu)
    xb = x.unsqueeze(-1) * bands
    want = torch.cat([x, torch.cos(xb).reshape(P, -1), torch.sin(xb).reshape(P, -1), f0, f1], dim=-1)
    assert float((e - want).abs().max()) <= 2e-6

    g0, gq0, gq1 = r(P, K0), r(P, 3), r(P, 3)
    ldz = (3 + D + 3 + 1 + 3) // 4 * 4
    Z = torch.full((P, ldz), 7.0, device=gpu)
    y = r(P, 1 + D)
    Z.view(-1)[2:].as_strided((P, 1 + D), (ldz, 1)).copy_(y)                 # where the forward chain stores [sdf | feature]
    n, sdf = torch.empty(P, 3, device=gpu), torch.empty(P, 1, device=gpu)
    lib.call("geo_normal", P, M, e, K0, g0, K0, 2, [gq0, gq1], n, Z, ldz, D, sdf)
    cosb, sinb = e[:, 3:3 + 3 * M].reshape(P, 3, M), e[:, 3 + 3 * M:3 + 6 * M].reshape(P, 3, M)
    gc, gs = g0[:, 3:3 + 3 * M].reshape(P, 3, M), g0[:, 3 + 3 * M:3 + 6 * M].reshape(P, 3, M)
    n_want = g0[:, :3] + ((gs * cosb - gc * sinb) * bands).sum(-1) + gq0 + gq1
    assert float((n - n_want).abs().max()) <= 1e-4 * float(n_want.abs().max())
    assert torch.equal(sdf, y[:, :1]) and torch.equal(Z[:, :3], x) and torch.equal(Z[:, 3:3 + D], y[:, 1:])
    assert torch.equal(Z[:, 3 + D:6 + D], n) and float(Z[:, 6 + D:].abs().max()) == 0.0

    g_sdf, g_feat, g_n, gZ = r(P), r(P, D), r(P, 3), r(P, ldz)
    gy, nbar = torch.empty(P, 1 + D, device=gpu), torch.empty(P, 3, device=gpu)
    lib.call("geo_backward_begin", P, D, g_sdf, g_feat, D, g_n, gZ, ldz, gy, nbar)
    assert torch.equal(gy[:, 0], g_sdf) and torch.equal(gy[:, 1:], g_feat + gZ[:, 3:3 + D]) and torch.equal(nbar, g_n + gZ[:, 3 + D:6 + D])
    lib.call("geo_backward_begin", P, D, None, None, 0, None, gZ, ldz, gy, nbar)
    assert float(gy[:, 0].abs().max()) == 0.0 and torch.equal(gy[:, 1:], gZ[:, 3:3 + D]) and torch.equal(nbar, gZ[:, 3 + D:6 + D])
    # ... and with a channel count that is not a multiple of 4 (the scalar kernel; D = 16 above takes 16-byte vectors at 4-byte alignment)
    D2 = 7
    ldz2 = (3 + D2 + 3 + 1 + 3) // 4 * 4
    g_feat2, gZ2 = r(P, D2 + 2), r(P, ldz2)
    gy2, nbar2 = torch.empty(P, 1 + D2, device=gpu), torch.empty(P, 3, device=gpu)
    lib.call("geo_backward_begin", P, D2, g_sdf, _Strided(g_feat2[:, :D2]), D2 + 2, g_n, gZ2, ldz2, gy2, nbar2)
    assert torch.equal(gy2[:, 0], g_sdf) and torch.equal(gy2[:, 1:], g_feat2[:, :D2] + gZ2[:, 3:3 + D2]) and torch.equal(nbar2, g_n + gZ2[:, 3 + D2:6 + D2])

    gb0 = torch.empty(P, K0, device=gpu)
    lib.call("geo_gbar0", P, M, e, K0, nbar, 2, [f0, f1], [C0, C1], gb0)
    nb = nbar.unsqueeze(-1) * bands
    want = torch.cat([nbar, (-sinb * nb).reshape(P, -1), (cosb * nb).reshape(P, -1), f0, f1], dim=-1)
    assert float((gb0 - want).abs().max()) <= 1e-5 * float(want.abs().max())

    dst = torch.zeros(P, 7, device=gpu)
    lib.call("copy_columns", P, 5, _Strided(e[:, 2:]), K0, _ColumnView(dst, 1), 7)
    assert torch.equal(dst[:, 1:6], e[:, 2:7]) and float(dst[:, 0].abs().max()) == 0.0 and float(dst[:, 6].abs().max()) == 0.0

    B, rows = 2, P // 2
    cam = r(B, 3)
    out = torch.zeros(P, 4, device=gpu)
    lib.call("inverse_squared_distance", P, rows, e, K0, cam, _ColumnView(out, 3), 4)
    d = x.reshape(B, rows, 3) - cam[:, None, :]
    want = (1.0 / (torch.sqrt((d * d).sum(-1)) ** 2 + 1e-5)).reshape(P)
    assert float(((out[:, 3] - want) / want).abs().max()) <= 1e-5 and float(out[:, :3].abs().max()) == 0.0


def test_pixel_normal_matches_composite(gpu):
    from ndjir_amd.volume import pixel_normal
    g64 = torch.randn(1, 300, 3, dtype=torch.float64, device=gpu, requires_grad=True)
    eps = 1e-6
    ref = (g64 + eps) / torch.sqrt(((g64 + eps) ** 2).sum(-1, keepdim=True))
    g32 = g64.detach().float().requires_grad_(True)
    out = pixel_normal(g32, eps)
    assert float((out.double() - ref).abs().max()) <= 1e-6
    w = torch.randn_like(ref)
    (a,), (b,) = torch.autograd.grad(out, [g32], w.float()), torch.autograd.grad(ref, [g64], w)
    assert float((a.double() - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("family", ["voxel", "cosine_voxel", "lanczos_voxel"])
def test_voxel_query_encode_equals_query_then_encode(gpu, family):
    """ndjir_voxel_feature_query_encode (csrc/grid.hip `k_voxel_query_encode`): the rows [x | cos | sin | feature] of the
    geometric net's input in one launch -- bit for bit what <family>_query followed by ndjir_geo_encode produce (points inside,
    on and outside the box; a ragged grid; a row stride wider than the row)."""
    from ndjir_amd import lib
    from ndjir_amd.grid_feature import _core
    rng = np.random.RandomState(5)
    P, M, D = 3000, 6, 4
    gs = [9, 16, 12]
    x = torch.tensor(rng.rand(P, 3) * 2.6 - 1.3, dtype=torch.float32, device=gpu)
    x[:3] = torch.tensor([[-1.0, 1.0, 0.0], [1.0, -1.0, 1.0], [0.0, 0.0, 0.0]], device=gpu)
    F = torch.tensor(rng.randn(*gs, D), dtype=torch.float32, device=gpu)
    fam = _core.FAMILIES[family]
    vf = torch.empty((P, D), device=gpu)
    lib.call(f"{fam.prefix}_{fam.fwd}", P * D, vf, x, F, gs, D, [-1.0] * 3, [1.0] * 3, 0)
    W = 3 + 6 * M + D
    want = torch.empty((P, W), device=gpu)
    lib.call("geo_encode", P, M, x, 1, [vf], [D], want, W)
    lde = W + 5
    got = torch.full((P, lde), float("nan"), device=gpu)
    lib.call("voxel_feature_query_encode", P, M, x, F, gs, D, [-1.0] * 3, [1.0] * 3, _core.interp_code(fam), got, lde)
    assert torch.equal(got[:, :W], want)
    assert torch.isnan(got[:, W:]).all()


@pytest.mark.parametrize("pre", ["", "cosine_", "lanczos_"])
def test_triplaneline_query_encode_equals_queries_then_encode(gpu, pre):
    """ndjir_triplaneline_query_encode (csrc/grid.hip `k_tri_query_encode`): the rows [x | cos | sin | tri-plane feature (Dp, 3) |
    tri-line feature (Dl, 3)] of the `triplaneline` geometric net's input in one launch -- bit for bit what the two
    <family>_query_on_* launches followed by ndjir_geo_encode produce (points inside, on and outside the box; different grid
    sizes and channel counts of plane and line; a row stride wider than the row)."""
    from ndjir_amd import lib
    from ndjir_amd.grid_feature import _core
    rng = np.random.RandomState(6)
    P, M, Gp, Dp, Gl, Dl = 3000, 6, 24, 8, 17, 4
    x = torch.tensor(rng.rand(P, 3) * 2.6 - 1.3, dtype=torch.float32, device=gpu)
    x[:3] = torch.tensor([[-1.0, 1.0, 0.0], [1.0, -1.0, 1.0], [0.0, 0.0, 0.0]], device=gpu)
    Fp = torch.tensor(rng.randn(3, Gp, Gp, Dp), dtype=torch.float32, device=gpu)
    Fl = torch.tensor(rng.randn(3, Gl, Dl), dtype=torch.float32, device=gpu)
    fp, fl = _core.FAMILIES[pre + "triplane"], _core.FAMILIES[pre + "triline"]
    vp, vl = torch.empty((P, 3 * Dp), device=gpu), torch.empty((P, 3 * Dl), device=gpu)
    lib.call(f"{fp.prefix}_{fp.fwd}", P * 3 * Dp, vp, x, Fp, Gp, Dp, [-1.0] * 3, [1.0] * 3, 0)
    lib.call(f"{fl.prefix}_{fl.fwd}", P * 3 * Dl, vl, x, Fl, Gl, Dl, [-1.0] * 3, [1.0] * 3, 0)
    W = 3 + 6 * M + 3 * Dp + 3 * Dl
    want = torch.empty((P, W), device=gpu)
    lib.call("geo_encode", P, M, x, 2, [vp, vl], [3 * Dp, 3 * Dl], want, W)
    lde = W + 3
    got = torch.full((P, lde), float("nan"), device=gpu)
    lib.call("triplaneline_query_encode", P, M, x, Fp, Gp, Dp, Fl, Gl, Dl, [-1.0] * 3, [1.0] * 3, _core.interp_code(fp), got, lde)
    assert torch.equal(got[:, :W], want)
    assert torch.isnan(got[:, W:]).all()
