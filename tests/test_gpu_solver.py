"""GPU parity of the optimizer step (ndjir_amd/csrc/solver.hip through the C ABI) against oracle/solver.py:
bit-exact in fp32 (the kernel is compiled without FMA contraction and in the oracle's expression order)."""
import numpy as np
import pytest
import torch

from ndjir_amd import lib
from oracle import solver as OS

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _state(dev, lr):
    st = torch.zeros(4, dtype=torch.float32, device=dev)
    st[0] = float(np.float32(lr))
    return st


@pytest.mark.parametrize("n", [4, 1024, 4096 + 4, 1000003, 7])
@pytest.mark.parametrize("decay", [0.0, 1e-3])
def test_dense_adam_bit_exact(gpu, n, decay):
    rng = np.random.RandomState(412 + n)
    w0 = (rng.randn(n) * 1e-3).astype(np.float32)
    o = OS.Adam(alpha=5e-4)
    wo = w0.copy()
    o.set_parameters({"F": wo})
    w, m, v = T(w0, gpu), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
    g = torch.zeros(n, device=gpu)
    st = _state(gpu, 5e-4)
    for step in range(5):
        gl = (rng.randn(n) * 10.0 ** rng.randint(-6, 1)).astype(np.float32)
        gl[rng.rand(n) < 0.7] = 0.0            # most grid cells receive no loss gradient
        o.zero_grad()
        o.weight_decay(decay)
        o.grads["F"] += gl
        o.update()
        g += T(gl, gpu)                        # the buffer was re-armed by the previous update
        lib.call("solver_adam_begin", st, 0.9, 0.999, None, None)
        lib.call("solver_adam", n, w, g, m, v, 0.0, 0.9, 0.999, 1e-8, decay, 1, st)
        assert float(g.abs().max()) == 0.0
        np.testing.assert_array_equal(w.cpu().numpy(), wo)
        np.testing.assert_array_equal(m.cpu().numpy(), o.m["F"])
        np.testing.assert_array_equal(v.cpu().numpy(), o.v["F"])
    sti = st.view(torch.int32).cpu()
    assert int(sti[1]) == 5 and int(sti[3]) == 0
    assert float(st[2]) == np.float32(o.alpha_t())


def test_dense_adam_by_value_and_keep_grad(gpu):
    """state = null: alpha_t by value, gradient buffer left alone when zero_grad = 0."""
    n = 4096
    rng = np.random.RandomState(1)
    w0, g0 = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    o = OS.Adam(alpha=1e-3)
    wo = w0.copy()
    o.set_parameters({"w": wo})
    o.grads["w"] += g0
    o.update()
    w, g, m, v = T(w0, gpu), T(g0, gpu), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
    lib.call("solver_adam", n, w, g, m, v, float(np.float32(o.alpha_t())), 0.9, 0.999, 1e-8, 0.0, 0, None)
    np.testing.assert_array_equal(w.cpu().numpy(), wo)
    np.testing.assert_array_equal(g.cpu().numpy(), g0)


def test_multi_tensor_adam_bit_exact(gpu):
    rng = np.random.RandomState(7)
    sizes = [1, 3, 256, 257, 1024, 1025, 39 * 256, 256 * 256, 2, 128] * 3 + [5, 301 * 128]   # 32 tensors: two launches
    w0 = [rng.randn(s).astype(np.float32) for s in sizes]
    o = OS.Adam(alpha=5e-4)
    wo = {f"p{i}": a.copy() for i, a in enumerate(w0)}
    o.set_parameters(wo)
    w = [T(a, gpu) for a in w0]
    m = [torch.zeros_like(a) for a in w]
    v = [torch.zeros_like(a) for a in w]
    st = _state(gpu, 5e-4)
    for step in range(3):
        gl = [rng.randn(s).astype(np.float32) for s in sizes]
        o.zero_grad()
        o.weight_decay(1e-3)
        for i, a in enumerate(gl):
            if i != 4:                               # tensor 4: no gradient (null pointer) = decay only
                o.grads[f"p{i}"] += a
        o.update()
        g = [T(a, gpu) if i != 4 else None for i, a in enumerate(gl)]
        lib.call("solver_adam_begin", st, 0.9, 0.999, None, None)
        lib.call("solver_adam_multi", len(w), w, g, m, v, sizes, 0.0, 0.9, 0.999, 1e-8, 1e-3, st)
        for i in range(len(w)):
            np.testing.assert_array_equal(w[i].cpu().numpy(), wo[f"p{i}"], err_msg=f"tensor {i} step {step}")
            np.testing.assert_array_equal(v[i].cpu().numpy(), o.v[f"p{i}"])


@pytest.mark.parametrize("fa,fb,skip", [(1, 1, True), (1, 0, False), (0, 1, False), (0, 0, False)])
def test_guard_skips_only_when_both_flags_are_raised(gpu, fa, fb, skip):
    n = 2048
    w0 = np.arange(n, dtype=np.float32)
    w, g = T(w0, gpu), torch.ones(n, device=gpu)
    m, v = torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
    st = _state(gpu, 1e-2)
    A = torch.tensor([fa], dtype=torch.int32, device=gpu)
    B = torch.tensor([fb], dtype=torch.int32, device=gpu)
    lib.call("solver_adam_begin", st, 0.9, 0.999, A, B)
    lib.call("solver_adam", n, w, g, m, v, 0.0, 0.9, 0.999, 1e-8, 0.0, 1, st)
    lib.call("solver_adam_multi", 1, [w[:5].clone()], [g[:5].clone()], [m[:5].clone()], [v[:5].clone()], [5], 0.0, 0.9,
             0.999, 1e-8, 0.0, st)
    sti = st.view(torch.int32).cpu()
    assert int(sti[3]) == int(skip) and int(sti[1]) == (0 if skip else 1)     # a vetoed step does not advance t
    assert float(g.abs().max()) == 0.0                                        # the buffer is re-armed either way
    changed = bool((w.cpu().numpy() != w0).any())
    assert changed == (not skip)
    assert bool(m.abs().max() > 0) == (not skip)
    # one flag only: that flag decides
    st2 = _state(gpu, 1e-2)
    lib.call("solver_adam_begin", st2, 0.9, 0.999, A, None)
    assert int(st2.view(torch.int32)[3]) == fa


def test_check_inf_or_nan(gpu):
    n = (1 << 20) + 3
    g = torch.randn(n, device=gpu)[: n - 3]          # 16-byte aligned view of n - 3 floats
    flag = torch.zeros(1, dtype=torch.int32, device=gpu)
    lib.call("solver_check_inf_or_nan", g.numel(), g, flag)
    assert int(flag) == 0
    for pos, val in [(0, float("nan")), (g.numel() - 1, float("inf")), (12345, -float("inf"))]:
        h = g.clone()
        h[pos] = val
        flag.zero_()
        lib.call("solver_check_inf_or_nan", h.numel(), h, flag)
        assert int(flag) == 1, (pos, val)
    tail = torch.randn(1027, device=gpu)
    flag.zero_()
    lib.call("solver_check_inf_or_nan", 1027, tail, flag)
    assert int(flag) == 0
    tail[1026] = float("nan")
    lib.call("solver_check_inf_or_nan", 1027, tail, flag)
    assert int(flag) == 1
    ts = [torch.randn(s, device=gpu) for s in [1, 7, 1024, 1025, 70000] * 6]
    flag.zero_()
    lib.call("solver_check_inf_or_nan_multi", len(ts), ts, [t.numel() for t in ts], flag)
    assert int(flag) == 0
    ts[28][1024] = float("inf")
    lib.call("solver_check_inf_or_nan_multi", len(ts), ts, [t.numel() for t in ts], flag)
    assert int(flag) == 1


def test_check_touched_cells(gpu):
    G, D = 32, 4
    buf = torch.zeros(G, G, G, D, device=gpu)
    q = torch.rand(500, 3, device=gpu) * 1.6 - 0.8
    flag = torch.zeros(1, dtype=torch.int32, device=gpu)
    args = (q.shape[0], buf, q.contiguous(), [G] * 3, D, [-1, -1, -1], [1, 1, 1], flag)
    lib.call("voxel_feature_check_touched", *args)
    assert int(flag) == 0
    # a cell that query 17 touches: its lower corner
    idx = ((q[17] + 1) / 2 * (G - 1)).floor().long()
    buf[idx[0], idx[1], idx[2], 2] = float("nan")
    lib.call("voxel_feature_check_touched", *args)
    assert int(flag) == 1


def test_sum_squares(gpu):
    x = torch.randn((1 << 21) + 1, device=gpu)
    out = torch.zeros(1, dtype=torch.float64, device=gpu)
    lib.call("solver_sum_squares", x.numel(), x, out)
    ref = float((x.double() ** 2).sum())
    assert float(out) == pytest.approx(ref, rel=1e-7)


def test_solvers_against_oracle(gpu):
    """The reference's step order (zero_grad, weight_decay, clip, backward, guard, update) through the product's
    `Solvers` on a small parameter set: MLP tensors + a voxel grid with an accumulate-in-place gradient buffer."""
    from ndjir_amd import config, parameter as P
    from ndjir_amd.grid_feature import set_grad_buffer
    from ndjir_amd.solver import Solvers
    conf = config.load("default", ["train.batch_size=1", "train.epoch=40", "train.warmup_term_ratio=0.1"])
    rng = np.random.RandomState(3)
    shapes = {"geometric-network/affine-00/affine/W": (43, 256), "geometric-network/affine-00/affine/b": (256,),
              "geometric-network/gain": (1,), "geometric-network/voxel_feature/F": (8, 8, 8, 4)}
    init = {k: (rng.randn(*s) * 0.1).astype(np.float32) for k, s in shapes.items()}
    P.clear_parameters()
    P.set_device(gpu)
    for k, a in init.items():
        P.get_parameter_or_create(k, a.shape, a, True)
    P.get_parameter_or_create("photogrammetric-light-network/gain", (1,), np.asarray([1.0]), False)
    F = P.get_parameters()["geometric-network/voxel_feature/F"]
    buf = torch.zeros_like(F)
    set_grad_buffer(F, buf)
    try:
        s = Solvers(conf)
        s.set_parameters()
        assert set(s.solver_feat.names) == {"geometric-network/voxel_feature/F"} and len(s.solver_weight.names) == 3
        o = OS.Solvers(dict(conf.train))
        ow = {k: a.copy() for k, a in init.items()}
        o.set_parameters(ow)
        for i in range(6):
            s.update_learning_rate(i + 3)
            o.update_learning_rate(i + 3)
            lg = {k: (rng.randn(*sh) * (rng.rand(*sh) < 0.3)).astype(np.float32) for k, sh in shapes.items()}
            if i == 4:                                   # both solvers see a non-finite gradient: the step is skipped
                lg["geometric-network/gain"][0] = np.nan
                lg["geometric-network/voxel_feature/F"][1, 2, 3, 0] = np.inf
            if i == 2:                                   # one solver only: the reference still updates (`and`)
                pass
            stepped = o.step(lg)
            s.zero_grad()
            s.weight_decay()
            s.clip_grad_by_norm()
            buf += T(lg["geometric-network/voxel_feature/F"], gpu)          # "backward" accumulates into the buffer
            s.set_gradients({k: T(a, gpu) for k, a in lg.items() if not k.endswith("feature/F")})
            s.guarded_update()
            assert s.solver_feat.skipped() == (not stepped) and s.solver_weight.skipped() == (not stepped)
            for k, p in P.get_parameters(grad_only=True).items():
                np.testing.assert_array_equal(p.detach().cpu().numpy(), ow[k], err_msg=f"{k} step {i}")
            assert float(buf.abs().max()) == 0.0
        assert s.solver_feat.step_count() == o.solver_feat.t == 5
        # schedules wrote their parameters
        assert float(P.get_parameters()["cos_anneal_ratio"]) == pytest.approx(OS.cos_anneal_ratio(dict(conf.train), 8))
    finally:
        set_grad_buffer(F, None)
        P.clear_parameters()


@pytest.mark.parametrize("skip", [False, True])
def test_adam_touched_bitmap_equals_dense(gpu, skip):
    """ndjir_solver_adam_touched + ndjir_voxel_feature_mark_touched: reading g only in the marked cells gives the dense
    kernel's result bit for bit (the gradient is zero elsewhere), g and the bitmap come back zero; a vetoed step leaves w, m, v."""
    G, D, P = 32, 4, 2000
    gen = torch.Generator(device=gpu).manual_seed(9)
    q = (torch.rand(P, 3, device=gpu, generator=gen) * 2.2 - 1.1).contiguous()          # some points outside the box
    go = torch.randn(P, D, device=gpu, generator=gen)
    g = torch.zeros(G, G, G, D, device=gpu)
    lib.call("voxel_feature_grad_feature", P * D, g, go, q, [G] * 3, D, [-1] * 3, [1] * 3, 0, 1)
    w0 = torch.randn(G, G, G, D, device=gpu, generator=gen) * 1e-3
    m0, v0 = torch.rand_like(w0) * 1e-3, torch.rand_like(w0) * 1e-6
    n = w0.numel()
    flag = torch.tensor([1 if skip else 0], dtype=torch.int32, device=gpu)
    res = []
    for mode in ("dense", "touched"):
        w, gg, m, v = w0.clone(), g.clone(), m0.clone(), v0.clone()
        st = _state(gpu, 5e-4)
        lib.call("solver_adam_begin", st, 0.9, 0.999, flag, None)
        if mode == "dense":
            lib.call("solver_adam", n, w, gg, m, v, 0.0, 0.9, 0.999, 1e-8, 1e-3, 1, st)
        else:
            bm = torch.zeros(n // 128, dtype=torch.int32, device=gpu)
            lib.call("voxel_feature_mark_touched", P, q, [G] * 3, D, [-1] * 3, [1] * 3, bm)
            marked = int(sum(bin(x & 0xffffffff).count("1") for x in bm.cpu().tolist()))
            assert marked >= int((g != 0).any(-1).sum()) > 0
            lib.call("solver_adam_touched", n, w, gg, m, v, 0.0, 0.9, 0.999, 1e-8, 1e-3, bm, st)
            assert int(bm.abs().sum()) == 0
        assert float(gg.abs().max()) == 0.0
        res.append((w, m, v))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][0], w0) == skip
