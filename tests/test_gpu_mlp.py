"""Fused MLP chain kernel (ndjir_amd/csrc/mlp.hip) vs a plain torch reference of the same op
(fp64 on the CPU): forward, input gradient, weight and bias gradients; IDR-style skip connection;
ragged point counts (tile = 64 points); narrow (N <= 32) and 257-wide output layers."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def ref_mlp(x, Ws, bs, beta=100.0, skip_layer=-1, skip_scale=1.0):
    h = x
    for j, (W, b) in enumerate(zip(Ws, bs)):
        z = h @ W + b
        if j == len(Ws) - 1:
            return z
        h = TF.softplus(z, beta=beta)
        if j == skip_layer:
            h = torch.cat([h, x], dim=-1) * skip_scale


def make(dims, seed, skip_layer=-1):
    rng = np.random.RandomState(seed)
    Ws, bs = [], []
    kin = dims[0]
    for j in range(len(dims) - 1):
        Ws.append(torch.tensor(rng.randn(kin, dims[j + 1]) * np.sqrt(2.0 / kin), dtype=torch.float32))
        bs.append(torch.tensor(rng.randn(dims[j + 1]) * 0.1, dtype=torch.float32))
        kin = dims[j + 1] + (dims[0] if j == skip_layer else 0)
    return Ws, bs


CASES = [
    ((259, 256, 256, 256, 3), 64, -1),       # base colour net
    ((39, 128, 128, 128, 1), 1000, -1),      # environment light net, ragged P
    ((301, 128, 128, 128, 1), 130, -1),      # soft visibility net (K0 = 301)
    ((262, 128, 128, 128, 6), 257, -1),      # specular reflectance net
    ((52, 256, 256, 256, 257), 96, -1),      # background geometric net (257-wide output)
    ((43, 256, 256, 256, 213, 256, 256, 256, 257), 200, 3),   # geometric net with skip
    ((43, 256, 256, 256, 213, 256, 256, 256, 1), 77, 3),      # sdf-only variant (sampler)
    ((5, 64, 2), 3, -1),
    # output blocks that do not fill a round of 8 waves (row-split remainder units, shared output slots)
    ((43, 128, 128, 129), 200, -1),
    ((43, 128, 161), 130, -1),
    ((32, 128, 97), 64, -1),
    ((43, 288, 288, 289), 100, -1),
]


@pytest.mark.parametrize("dims,P,skip", CASES)
def test_fused_mlp_matches_reference(gpu, dims, P, skip):
    from ndjir_amd.mlp import fused_mlp
    scale = 1.0 / np.sqrt(2.0) if skip >= 0 else 1.0
    Ws, bs = make(dims, 7, skip)
    rng = np.random.RandomState(11)
    x = torch.tensor(rng.randn(P, dims[0]) * 0.5, dtype=torch.float32)
    gy = torch.tensor(rng.randn(P, dims[-1]), dtype=torch.float32)

    xd = x.to(gpu).requires_grad_(True)
    Wd = [w.to(gpu).requires_grad_(True) for w in Ws]
    bd = [b.to(gpu).requires_grad_(True) for b in bs]
    y = fused_mlp(xd, Wd, bd, 100.0, skip, scale)
    grads = torch.autograd.grad(y, [xd] + Wd + bd, gy.to(gpu))

    x64 = x.double().requires_grad_(True)
    W64 = [w.double().requires_grad_(True) for w in Ws]
    b64 = [b.double().requires_grad_(True) for b in bs]
    y64 = ref_mlp(x64, W64, b64, 100.0, skip, scale)
    g64 = torch.autograd.grad(y64, [x64] + W64 + b64, gy.double())

    def rel(a, b):
        return float((a.detach().cpu().double() - b).norm() / max(float(b.norm()), 1e-30))

    assert rel(y, y64) < 2e-6, rel(y, y64)
    names = ["x"] + [f"W{j}" for j in range(len(Ws))] + [f"b{j}" for j in range(len(bs))]
    for n, a, b in zip(names, grads, g64):
        assert rel(a, b) < 2e-5, (n, rel(a, b))


def test_fused_mlp_no_input_grad(gpu):
    """chains whose input needs no gradient stop one GEMM early (environment light net)."""
    from ndjir_amd.mlp import fused_mlp
    dims = (39, 128, 128, 128, 1)
    Ws, bs = make(dims, 3)
    x = torch.randn(500, 39)
    Wd = [w.to(gpu).requires_grad_(True) for w in Ws]
    bd = [b.to(gpu).requires_grad_(True) for b in bs]
    y = fused_mlp(x.to(gpu), Wd, bd)
    g = torch.autograd.grad(y.sum(), Wd + bd)
    W64 = [w.double().requires_grad_(True) for w in Ws]
    b64 = [b.double().requires_grad_(True) for b in bs]
    g64 = torch.autograd.grad(ref_mlp(x.double(), W64, b64).sum(), W64 + b64)
    for a, b in zip(g, g64):
        assert float((a.cpu().double() - b).norm() / b.norm()) < 2e-5


def test_fused_mlp_throughput(gpu):
    """Not a parity test: prints achieved fp32 TFLOP/s of the chain kernel at the bench shape."""
    import time
    from ndjir_amd.mlp import chain_forward
    dims = (43, 256, 256, 256, 213, 256, 256, 256, 257)
    Ws, bs = make(dims, 1, 3)
    Wd = [w.to(gpu) for w in Ws]
    bd = [b.to(gpu) for b in bs]
    P = 65536
    x = torch.randn(P, 43, device=gpu)
    flops = 2 * P * (43 * 256 + 256 * 256 * 2 + 256 * 213 + 256 * 256 * 3 + 256 * 257)
    for keep in (False, True):
        chain_forward(x, Wd, bd, 100.0, 3, 0.7071, keep_hidden=keep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            chain_forward(x, Wd, bd, 100.0, 3, 0.7071, keep_hidden=keep)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"\nchain fwd geometric P={P} keep_hidden={keep}: {dt * 1e6:.0f} us, {flops / dt / 1e12:.1f} TFLOP/s")


@pytest.mark.parametrize("P,K,N", [(1000, 301, 128), (65, 39, 1), (4096, 256, 257), (33, 5, 7), (20000, 213, 256), (65536, 128, 6),
                                   (3000, 600, 2), (100, 128, 8), (100, 128, 9), (5000, 259, 256), (5000, 290, 256),
                                   (5000, 43, 256), (777, 52, 40), (3000, 300, 300), (2000, 128, 192), (1000, 64, 64)])
def test_wgrad_kernel(gpu, P, K, N):
    """split-P weight-gradient GEMM vs fp64; strided operand views (delta of the skip layer)."""
    from ndjir_amd.mlp import wgrad
    rng = np.random.RandomState(5)
    A = torch.tensor(rng.randn(P, K + 3), dtype=torch.float32)
    B = torch.tensor(rng.randn(P, N + 5), dtype=torch.float32)
    out = wgrad(A.to(gpu)[:, :K], B.to(gpu)[:, 2:2 + N])
    ref = A[:, :K].double().t() @ B[:, 2:2 + N].double()
    err = float((out.cpu().double() - ref).norm() / ref.norm())
    assert err < 2e-6, err


def test_wgrad_throughput(gpu):
    import time
    from ndjir_amd.mlp import wgrad
    P, K, N = 65536, 256, 256
    A = torch.randn(P, K, device=gpu)
    B = torch.randn(P, N, device=gpu)
    for fn, name in ((wgrad, "ndjir wgrad"), (lambda a, b: a.t().mm(b), "library A^T B")):
        fn(A, B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn(A, B)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"\n{name}: {dt * 1e6:.0f} us, {2 * P * K * N / dt / 1e12:.1f} TFLOP/s")


@pytest.mark.parametrize("P,N,ld", [(65536, 257, 257), (1000, 3, 3), (77, 1, 1), (4097, 256, 300), (5, 600, 600), (0, 4, 4), (70000, 2048, 2048)])
def test_colsum_kernel(gpu, P, N, ld):
    """bias-gradient column sums vs fp64, strided rows, accumulate flag."""
    from ndjir_amd.mlp import colsum
    rng = np.random.RandomState(P + N)
    X = torch.tensor(rng.randn(max(P, 1), ld), dtype=torch.float32, device=gpu)[:P, :N]
    ref = X.double().sum(0)
    out = colsum(X)
    scale = max(float(ref.abs().max()), 1.0)
    assert float((out.double() - ref).abs().max()) / scale < 2e-5
    base = torch.ones(N, device=gpu)
    colsum(X, out=base, accum=True)
    assert float((base.double() - 1.0 - ref).abs().max()) / scale < 2e-5


@pytest.mark.parametrize("grid", [True, False])
def test_geometric_main_double_backward(gpu, grid):
    """sdf, feature, n = d sdf/dx and the gradients of a loss that depends on all three (second-order
    terms through n) vs fp64 torch autograd of the composite restatement."""
    from ndjir_amd.geometric import geometric_main
    from oracle import composite as C
    from oracle import graph as G
    rng = np.random.RandomState(3)
    P, Gs, D0, M = 200, 8, 4, 6
    Dh = 128
    K0 = 39 + (D0 if grid else 0)
    dims = [K0, Dh, Dh, Dh, Dh - K0, Dh, Dh, Dh, Dh + 1]
    Ws, bs = make(tuple(dims), 5, skip_layer=3)
    Ws = [w * 1.5 for w in Ws]
    x = (rng.rand(P, 3) * 1.6 - 0.8).astype(np.float32)
    F = (rng.randn(Gs, Gs, Gs, D0) * 0.3).astype(np.float32)
    w_sdf = rng.randn(P, 1).astype(np.float32)
    w_feat = rng.randn(P, Dh).astype(np.float32)
    w_n = rng.randn(P, 3).astype(np.float32)
    c = 1.0 / np.sqrt(2.0)

    xd = torch.tensor(x, device=gpu)
    Fd = torch.tensor(F, device=gpu).requires_grad_(True) if grid else None
    Wd = [w.to(gpu).requires_grad_(True) for w in Ws]
    bd = [b.to(gpu).requires_grad_(True) for b in bs]
    sdf, feat, n, Z = geometric_main(xd, Fd, Wd, bd, M, 3, c)
    # the packed sample inputs: [x | feature | n | zeros]; a gradient arriving through Z counts for feature and n
    assert Z.shape == (P, (3 + Dh + 3 + 1 + 3) // 4 * 4) and feat.data_ptr() == Z.data_ptr() + 12
    assert torch.equal(Z[:, :3], xd) and torch.equal(Z[:, 3 + Dh:6 + Dh], n) and float(Z[:, 6 + Dh:].abs().max()) == 0.0
    w_Z = rng.randn(P, Z.shape[1]).astype(np.float32)
    loss = (sdf * torch.tensor(w_sdf, device=gpu)).sum() + (feat * torch.tensor(w_feat, device=gpu)).sum() \
        + ((n * torch.tensor(w_n, device=gpu)).sum(-1) ** 2).sum() + (Z * torch.tensor(w_Z, device=gpu)).sum()
    params = Wd + bd + ([Fd] if grid else [])
    grads = torch.autograd.grad(loss, params)

    x64 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    F64 = torch.tensor(F, dtype=torch.float64, requires_grad=True)
    W64 = [w.double().requires_grad_(True) for w in Ws]
    b64 = [b.double().requires_grad_(True) for b in bs]
    e = G.positional_encoding(x64, M)
    if grid:
        e = torch.cat([e, C.query_on_voxel(x64, F64)], dim=-1)
    y = ref_mlp(e, W64, b64, 100.0, 3, c)
    sdf64, feat64 = y[:, :1], y[:, 1:]
    n64, = torch.autograd.grad(sdf64.sum(), x64, create_graph=True)
    loss64 = (sdf64 * torch.tensor(w_sdf).double()).sum() + (feat64 * torch.tensor(w_feat).double()).sum() \
        + ((n64 * torch.tensor(w_n).double()).sum(-1) ** 2).sum() \
        + (feat64 * torch.tensor(w_Z[:, 3:3 + Dh]).double()).sum() + (n64 * torch.tensor(w_Z[:, 3 + Dh:6 + Dh]).double()).sum()
    g64 = torch.autograd.grad(loss64, W64 + b64 + ([F64] if grid else []))

    def rel(a, b):
        return float((a.detach().cpu().double() - b).norm() / max(float(b.norm()), 1e-30))

    assert rel(sdf, sdf64) < 1e-5 and rel(feat, feat64) < 1e-5
    assert rel(n, n64) < 2e-5, rel(n, n64)
    names = [f"W{j}" for j in range(8)] + [f"b{j}" for j in range(8)] + (["F"] if grid else [])
    for nm, a, b in zip(names, grads, g64):
        assert rel(a, b) < 1e-3, (nm, rel(a, b))   # fp32 vs fp64 through beta=100 second-order terms


def _rel(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).norm() / max(float(b.detach().cpu().double().norm()), 1e-30))


@pytest.mark.parametrize("P,div,K0,Dh,No", [(4096, 128, 39, 128, 1), (640, 64, 39, 128, 1), (300, 100, 7, 64, 3)])
def test_fused_mlp_row_bias(gpu, P, div, K0, Dh, No):
    """per-row-group first-layer term: y = MLP(x; z_0 += row_term[row // div]) vs fp64 autograd, all gradients
    (incl. the group-wise column sum that is the row term's gradient)."""
    from ndjir_amd.mlp import fused_mlp
    rng = np.random.RandomState(P + div)
    dims = [K0, Dh, Dh, No]
    Ws = [torch.tensor(rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i]), dtype=torch.float64, device=gpu) for i in range(3)]
    bs = [None] + [torch.tensor(rng.randn(dims[i + 1]) * 0.1, dtype=torch.float64, device=gpu) for i in (1, 2)]
    x = torch.tensor(rng.randn(P, K0), dtype=torch.float64, device=gpu)
    rt = torch.tensor(rng.randn(P // div, Dh) * 0.3, dtype=torch.float64, device=gpu)

    def ref(x, rt, Ws, bs):
        h = x @ Ws[0] + rt.repeat_interleave(div, dim=0)
        h = TF.softplus(h, beta=100)
        h = TF.softplus(h @ Ws[1] + bs[1], beta=100)
        return h @ Ws[2] + bs[2]

    a64 = [x.clone().requires_grad_(True), rt.clone().requires_grad_(True)] + [w.clone().requires_grad_(True) for w in Ws] \
        + [b.clone().requires_grad_(True) for b in bs[1:]]
    y64 = ref(a64[0], a64[1], a64[2:5], [None] + a64[5:7])
    a32 = [t.detach().float().requires_grad_(True) for t in a64]
    y32 = fused_mlp(a32[0], a32[2:5], [None] + a32[5:7], 100.0, row_bias=a32[1], row_bias_div=div)
    assert _rel(y32, y64.detach().cpu()) < 2e-6
    g = torch.tensor(rng.randn(*y64.shape), dtype=torch.float64, device=gpu)
    g64 = torch.autograd.grad(y64, a64, g)
    g32 = torch.autograd.grad(y32, a32, g.float())
    for i, (a, b) in enumerate(zip(g32, g64)):
        assert _rel(a, b.cpu()) < 2e-5, i


def test_multi_mlp_on_packed_input_with_row_term(gpu):
    """mlp.multi_mlp on a packed input (P, 24): two nets read its first 18 columns, a third reads 19 and takes a per-group
    row term; outputs, the shared input gradient (accumulated in the kernels; undefined beyond the widest net) and every
    parameter gradient against fp64 autograd of the separate nets."""
    from ndjir_amd.mlp import multi_mlp
    rng = np.random.RandomState(11)
    P, ld, div, Dh = 768, 24, 64, 64

    def net(K, No, seed):
        r = np.random.RandomState(seed)
        dims = [K, Dh, Dh, No]
        return ([torch.tensor(r.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i]), dtype=torch.float64, device=gpu) for i in range(3)],
                [torch.tensor(r.randn(dims[i + 1]) * 0.1, dtype=torch.float64, device=gpu) for i in range(3)])

    nets64 = [net(18, 3, 1), net(18, 2, 2), net(19, 1, 3)]
    x64 = torch.tensor(rng.randn(P, ld), dtype=torch.float64, device=gpu, requires_grad=True)
    rt64 = torch.tensor(rng.randn(P // div, Dh) * 0.3, dtype=torch.float64, device=gpu, requires_grad=True)

    def ref(x, W, b, K, rt=None):
        h = x[:, :K] @ W[0] + (b[0] if rt is None else rt.repeat_interleave(div, dim=0))
        h = TF.softplus(h, beta=100)
        h = TF.softplus(h @ W[1] + b[1], beta=100)
        return h @ W[2] + b[2]

    leaves64 = [x64, rt64]
    for W, b in nets64:
        for t in W + b:
            t.requires_grad_(True)
        leaves64 += W + b
    y64 = [ref(x64, *nets64[0], 18), ref(x64, *nets64[1], 18), ref(x64, *nets64[2], 19, rt64)]
    gs = [torch.tensor(rng.randn(*y.shape), dtype=torch.float64, device=gpu) for y in y64]
    g64 = torch.autograd.grad(y64, leaves64, gs, allow_unused=True)

    x32 = x64.detach().float().requires_grad_(True)
    rt32 = rt64.detach().float().requires_grad_(True)
    nets32 = [([w.detach().float().requires_grad_(True) for w in W], [t.detach().float().requires_grad_(True) for t in b]) for W, b in nets64]
    call = [(nets32[0][0], nets32[0][1]), (nets32[1][0], nets32[1][1]), (nets32[2][0], [None] + nets32[2][1][1:])]
    y32 = multi_mlp(x32, call, widths=[18, 18, 19], row_terms=[None, None, (rt32, div)])
    for a, b in zip(y32, y64):
        assert _rel(a, b.detach().cpu()) < 2e-6
    leaves32 = [x32, rt32]
    for W, b in nets32:
        leaves32 += W + b
    g32 = torch.autograd.grad(list(y32), leaves32, [g.float() for g in gs], allow_unused=True)
    assert _rel(g32[0][:, :19], g64[0][:, :19].cpu()) < 2e-5 and float(g32[0][:, 19:].abs().max()) == 0.0
    for i, (a, b) in enumerate(zip(g32[1:], g64[1:]), 1):
        if b is None or a is None:
            assert i == len(leaves32) - 3 and a is None          # the third net's first bias lives in the row term: unused here
            continue
        assert _rel(a, b.cpu()) < 2e-5, i


def test_multi_mlp_widest_net_without_gradient(gpu):
    """When the widest net of a MultiMLP receives no gradient, the first backward chain that runs assigns only its own
    input columns: the columns up to the widest net's width must come out as zeros, not as uninitialised memory."""
    from ndjir_amd.mlp import multi_mlp
    rng = np.random.RandomState(4)
    P, ld, Dh = 256, 24, 64

    def net(K, No, seed):
        r = np.random.RandomState(seed)
        dims = [K, Dh, No]
        return ([torch.tensor(r.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i]), dtype=torch.float32, device=gpu).requires_grad_(True)
                 for i in range(2)],
                [torch.tensor(r.randn(dims[i + 1]) * 0.1, dtype=torch.float32, device=gpu).requires_grad_(True) for i in range(2)])

    nets = [net(12, 3, 1), net(20, 2, 2)]
    x = torch.tensor(rng.randn(P, ld), dtype=torch.float32, device=gpu, requires_grad=True)
    torch.empty(1 << 22, device=gpu).fill_(float("nan"))          # poison the allocator's free blocks
    y = multi_mlp(x, [(nets[0][0], nets[0][1]), (nets[1][0], nets[1][1])], widths=[12, 20])
    gx, = torch.autograd.grad(y[0], [x], torch.ones_like(y[0]))      # only the NARROW net gets a gradient
    ref = x.detach().double().requires_grad_(True)
    h = TF.softplus(ref[:, :12] @ nets[0][0][0].double() + nets[0][1][0].double(), beta=100) @ nets[0][0][1].double() + nets[0][1][1].double()
    gref, = torch.autograd.grad(h, [ref], torch.ones_like(h))
    assert torch.isfinite(gx).all()
    assert _rel(gx[:, :12], gref[:, :12].cpu()) < 2e-5 and float(gx[:, 12:].abs().max()) == 0.0


@pytest.mark.parametrize("P,K,N", [(512, 262, 128), (100, 43, 256), (7, 3, 5), (4096, 256, 257)])
def test_linear_is_twice_differentiable(gpu, P, K, N):
    """mlp.linear (PF.affine, python/network.py:88-93) on the chain / wgrad / colsum kernels: value, first derivatives and
    the derivative of a gradient-dependent scalar (the nn.grad pattern of python/renderer.py:52) against fp64 autograd."""
    from ndjir_amd.mlp import linear
    from tests.parity_utils import rel_err
    g = torch.Generator(device="cpu").manual_seed(P + K)
    x64 = torch.randn(P, K, generator=g, dtype=torch.float64).requires_grad_(True)
    W64 = (torch.randn(K, N, generator=g, dtype=torch.float64) / K ** 0.5).requires_grad_(True)
    b64 = torch.randn(N, generator=g, dtype=torch.float64).requires_grad_(True)
    c64 = torch.randn(P, N, generator=g, dtype=torch.float64)

    def run(x, W, b, c, lin):
        y = torch.tanh(lin(x, W, b))
        (gx,) = torch.autograd.grad((y * c).sum(), x, create_graph=True)      # first-order graph, kept differentiable
        loss = (gx * gx).sum() + (y * y).sum()
        return y, gx, torch.autograd.grad(loss, [x, W, b])
    y64, gx64, gg64 = run(x64, W64, b64, c64, lambda x, W, b: x @ W + b)
    x = x64.detach().float().to(gpu).requires_grad_(True)
    W = W64.detach().float().to(gpu).requires_grad_(True)
    b = b64.detach().float().to(gpu).requires_grad_(True)
    y, gx, gg = run(x, W, b, c64.float().to(gpu), linear)
    assert rel_err(y, y64) < 2e-6 and rel_err(gx, gx64) < 5e-6
    for a, r, name in zip(gg, gg64, "xWb"):
        assert rel_err(a, r) < 2e-5, (name, rel_err(a, r))


# ---- arithmetic of the three engines ------------------------------------------------------------------------------------
def _geo_net_error(gpu, math_mode, P=4096, seed=3):
    """Relative error (vs fp64) of the 8-layer geometric net's forward value and input gradient under `math_mode`."""
    from ndjir_amd import mlp
    dims = (43, 256, 256, 256, 213, 256, 256, 256, 257)
    Ws, bs = make(dims, seed, 3)
    rng = np.random.RandomState(seed)
    x = torch.tensor(rng.randn(P, 43) * 0.5, dtype=torch.float32)
    gy = torch.tensor(rng.randn(P, 257), dtype=torch.float32)
    x64 = x.double().requires_grad_(True)
    W64 = [w.double().requires_grad_(True) for w in Ws]
    y64 = ref_mlp(x64, W64, [b.double() for b in bs], 100.0, 3, 0.7071067811865476)
    g64 = torch.autograd.grad(y64, [x64, W64[0], W64[4]], gy.double())
    old = mlp.get_math()
    mlp.set_math(math_mode)
    try:
        xd = x.to(gpu).requires_grad_(True)
        Wd = [w.to(gpu).requires_grad_(True) for w in Ws]
        y = mlp.fused_mlp(xd, Wd, [b.to(gpu) for b in bs], 100.0, 3, 0.7071067811865476)
        g = torch.autograd.grad(y, [xd, Wd[0], Wd[4]], gy.to(gpu))
    finally:
        mlp.set_math(old)
    rel = lambda a, b: float((a.detach().cpu().double() - b).norm() / b.norm())
    return rel(y, y64), [rel(a, b) for a, b in zip(g, g64)]


def test_f16x3_is_at_least_as_accurate_as_an_fp32_fma_chain(gpu):
    """The default engine (two-way f16 split, three MFMA partial products) against the fp32-input MFMA engine, which is
    bitwise an fp32 FMA chain (MI355X_MICROARCH.md): its error vs fp64 must not exceed the fp32 chain's -- the claim
    behind calling the arithmetic fp32-equivalent.  The exact bf16x6 engine is printed beside them."""
    from ndjir_amd import mlp
    e32, g32 = _geo_net_error(gpu, mlp.MATH_FP32)
    e6, g6 = _geo_net_error(gpu, mlp.MATH_BF16X6)
    e3, g3 = _geo_net_error(gpu, mlp.MATH_F16X3)
    print(f"\nforward error vs fp64: fp32 chain {e32:.2e}  bf16x6 {e6:.2e}  f16x3 {e3:.2e}")
    print(f"gradient errors (x, W0, W4): fp32 {g32}  bf16x6 {g6}  f16x3 {g3}")
    assert e3 <= 1.05 * e32, (e3, e32)
    for a, b in zip(g3, g32):
        assert a <= 1.25 * b, (g3, g32)
    assert e3 < 2e-6


@pytest.mark.parametrize("P,K,N", [(1, 27, 256), (37, 128, 262), (512, 262, 128), (512, 30, 1), (4096, 129, 33)])
@pytest.mark.parametrize("transpose", [False, True])
def test_small_affine_kernel(gpu, P, K, N, transpose):
    """`linear` on few rows (csrc/affine.hip: the per-ray terms of python/network.py:438, 528, 619 and their input gradients):
    ragged tiles in every dimension, odd K (a k-step of 2 with one half past the end), a weight that is a block of ROWS of a wider
    parameter's column range (row stride > width), bias, both orientations -- against fp64, and the chain kernel's result for the
    same call beside it."""
    from ndjir_amd import lib, mlp
    rng = np.random.RandomState(P + K + N)
    x = torch.tensor(rng.randn(P, K), dtype=torch.float32, device=gpu)
    Wfull = torch.tensor(rng.randn(K + 5, N + 7) if not transpose else rng.randn(N + 5, K + 7), dtype=torch.float32, device=gpu) / np.sqrt(K)
    W = Wfull[2:2 + K, 3:3 + N] if not transpose else Wfull[2:2 + N, 3:3 + K]           # column stride 1, row stride > width
    b = torch.tensor(rng.randn(N), dtype=torch.float32, device=gpu)
    ref = x.double() @ (W.double().t() if transpose else W.double()) + b.double()
    y = torch.full((P, N), float("nan"), device=gpu)
    lib.call("mlp_small_affine", P, x, K, K, mlp._Strided(W), W.stride(0), N, int(transpose), b, y, N)
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) * np.sqrt(K)
    assert float((y.double() - ref).norm() / ref.norm()) < 1e-6
    # the path `linear` takes for this many rows, with and without the switch
    assert P <= mlp.SMALL_AFFINE_ROWS
    y2 = mlp._mm(x, W, transpose, b)
    assert torch.equal(y2, y)


def test_pipelined_kernel_one_accumulator_accuracy(gpu):
    """csrc/mlp3p.hip keeps ONE fp32 accumulator per block (hi hi' + hi lo' + lo hi' summed together, lo unscaled) where
    mlp3.hip / mlp3w.hip keep two: the same 8-layer net (forward with stored activations, backward) on 128-point tiles with the
    pipeline switched on, against fp64 and beside the strict-fp32 engine and the two-accumulator kernels -- its error may not
    exceed the fp32 FMA chain's by more than a quarter."""
    from ndjir_amd import mlp
    old_tile, old_pipe = mlp.get_tile_rows(), mlp.get_chain_pipeline()
    try:
        mlp.set_tile_rows(128)
        e32, g32 = _geo_net_error(gpu, mlp.MATH_FP32)
        e3, g3 = _geo_net_error(gpu, mlp.MATH_F16X3)
        mlp.set_chain_pipeline(7)
        e3p, g3p = _geo_net_error(gpu, mlp.MATH_F16X3)
    finally:
        mlp.set_tile_rows(old_tile)
        mlp.set_chain_pipeline(old_pipe)
    print(f"\nforward error vs fp64: fp32 chain {e32:.2e}  f16x3 two accumulators {e3:.2e}  one accumulator (pipelined) {e3p:.2e}")
    print(f"gradient errors (x, W0, W4): fp32 {g32}  two accumulators {g3}  one accumulator {g3p}")
    assert e3p <= 1.25 * e32 and e3p < 2e-6, (e3p, e32)
    for a, b in zip(g3p, g32):
        assert a <= 1.25 * b, (g3p, g32)


@pytest.mark.parametrize("spread", [1e-6, 1e6])
def test_f16x3_scaling_groups(gpu, spread):
    """Rows of very different magnitude in one tile (the activations of a tile share one power-of-two scale) and weight
    columns of very different magnitude: every output row / column keeps fp32-class relative accuracy as long as it is
    within 2^-28 of its group's maximum; magnitudes far outside fp16's own range (1e-12 .. 1e9) are handled by the scales."""
    from ndjir_amd.mlp import fused_mlp
    rng = np.random.RandomState(11)
    P, K, H, N = 256, 64, 128, 96
    row_scale = np.exp(rng.uniform(0, np.log(1e5), size=(P, 1))) * spread          # rows spread over 5 decades
    col_scale = np.exp(rng.uniform(0, np.log(1e3), size=(1, N)))
    x = torch.tensor(rng.randn(P, K) * row_scale, dtype=torch.float32)
    W0 = torch.tensor(rng.randn(K, H) / np.sqrt(K) / spread, dtype=torch.float32)      # pre-activations stay O(1..1e5)
    b0 = torch.tensor(rng.randn(H) * 0.1, dtype=torch.float32)
    W1 = torch.tensor(rng.randn(H, N) / np.sqrt(H) * col_scale, dtype=torch.float32)
    b1 = torch.zeros(N)
    y = fused_mlp(x.to(gpu), [W0.to(gpu), W1.to(gpu)], [b0.to(gpu), b1.to(gpu)], 100.0).cpu().double()
    ref = ref_mlp(x.double(), [W0.double(), W1.double()], [b0.double(), b1.double()], 100.0)
    # per-row and per-column relative errors, not just the norm of the whole matrix
    row_err = (y - ref).norm(dim=1) / ref.norm(dim=1)
    col_err = (y - ref).norm(dim=0) / ref.norm(dim=0)
    assert float(row_err.max()) < 3e-6, float(row_err.max())
    assert float(col_err.max()) < 3e-6, float(col_err.max())


def test_f16x3_nonfinite_rows_stay_confined(gpu):
    """An Inf / NaN in one row of a tile must not leak into the other rows of the tile through the shared scale."""
    from ndjir_amd.mlp import fused_mlp
    rng = np.random.RandomState(2)
    P, K, H, N = 128, 40, 128, 33
    x = torch.tensor(rng.randn(P, K), dtype=torch.float32)
    Ws = [torch.tensor(rng.randn(K, H) / np.sqrt(K), dtype=torch.float32), torch.tensor(rng.randn(H, N) / np.sqrt(H), dtype=torch.float32)]
    bs = [torch.zeros(H), torch.zeros(N)]
    ref = ref_mlp(x.double(), [w.double() for w in Ws], [b.double() for b in bs], 100.0)
    xb = x.clone()
    xb[5, 3] = float("nan")
    xb[70, 0] = float("inf")
    y = fused_mlp(xb.to(gpu), [w.to(gpu) for w in Ws], [b.to(gpu) for b in bs], 100.0).cpu()
    bad = torch.zeros(P, dtype=torch.bool)
    bad[5] = bad[70] = True
    assert not torch.isfinite(y[5]).any() and not torch.isfinite(y[70]).all()
    assert torch.isfinite(y[~bad]).all()
    assert float((y[~bad].double() - ref[~bad]).norm() / ref[~bad].norm()) < 2e-6


@pytest.mark.parametrize("scale_a,scale_b", [(1.0, 1.0), (1e-7, 1e4), (3e5, 1e-9)])
def test_wgrad_operand_scales(gpu, scale_a, scale_b):
    """Weight-gradient kernel with and without recorded maxima (the kernel's own pre-pass), operands far from O(1)."""
    from ndjir_amd.mlp import wgrad
    rng = np.random.RandomState(7)
    P, K, N = 5000, 200, 130
    A = torch.tensor(rng.randn(P, K) * scale_a, dtype=torch.float32)
    B = torch.tensor(rng.randn(P, N) * scale_b, dtype=torch.float32)
    ref = A.double().t() @ B.double()
    out = wgrad(A.to(gpu), B.to(gpu))
    assert float((out.cpu().double() - ref).norm() / ref.norm()) < 2e-6
    am = (A.abs().max() * 3.0).reshape(1).to(gpu)           # any upper bound will do
    bm = B.abs().max().reshape(1).to(gpu)
    out2 = wgrad(A.to(gpu), B.to(gpu), amax_a=am, amax_b=bm)
    assert float((out2.cpu().double() - ref).norm() / ref.norm()) < 2e-6


@pytest.mark.parametrize("blocked", [False, True])
def test_wgrad_one_outlier_row_elementwise(gpu, blocked):
    """The weight gradient's operands are scaled per TENSOR (csrc/wgrad.hip: one power of two from the recorded maximum; lo is
    kept unscaled, so x s is held to 2^-22 relative or 2^-25 absolute = 2^-39 of the tensor's largest element).  One outlier
    row at 10^6 x the typical delta pushes every other row 20 bits down its operand's range: the worst case of that design.
    Checked ELEMENT-WISE, and in particular on the rows of dW the outlier does not reach (its A entries are zero there), whose
    sums are made of typical rows only: |d| <= 3e-5 (|ref| + rms of those rows; measured 9.1e-6) -- norm-wise they would vanish beside the
    outlier's 10^6-fold contribution.  (At 10^9 the same entries would degrade to ~1e-3: the documented limit of a per-tensor
    scale; the step's deltas span ~3 decades, tests/test_gpu_trained_parity.py measures them after training.)"""
    from ndjir_amd.mlp import PB, wgrad_group
    rng = np.random.RandomState(23)
    P, K, N = 32768, 256, 256
    A = rng.randn(P, K).astype(np.float32)
    B = rng.randn(P, N).astype(np.float32)
    out_row = 12345
    A[out_row, :K // 2] = 0.0                         # rows k < K / 2 of dW never see the outlier
    B[out_row] *= 1e6
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    At, Bt = torch.tensor(A, device=gpu), torch.tensor(B, device=gpu)
    am, bm = At.abs().max().reshape(1), Bt.abs().max().reshape(1)

    def blk(t):          # (p, f) -> ((p >> 5) * ld + f) * 32 + (p & 31)
        Pn, W = t.shape
        return PB(t.view(Pn // 32, 32, W).permute(0, 2, 1).contiguous().view(Pn, W))
    out = torch.zeros(K, N, device=gpu)
    wgrad_group([(out, False, [((blk(At) if blocked else At), (blk(Bt) if blocked else Bt), am, bm)])])
    got = out.cpu().double().numpy()
    clean, hit = slice(0, K // 2), slice(K // 2, K)
    rms = np.sqrt((ref[clean] ** 2).mean())
    e_clean = np.abs(got[clean] - ref[clean]) / (np.abs(ref[clean]) + rms)
    e_hit = np.abs(got[hit] - ref[hit]) / (np.abs(ref[hit]) + 1e-3 * np.abs(ref[hit]).max())
    print(f"\noutlier row x 1e6 ({'blocked' if blocked else 'row-major'} operands): element-wise error, rows without the outlier "
          f"{e_clean.max():.2e}, rows with it {e_hit.max():.2e}")
    assert e_clean.max() <= 3e-5, e_clean.max()
    assert e_hit.max() <= 1e-5, e_hit.max()


@pytest.mark.parametrize("P,dims,in_place", [(256, (43, 128, 128, 257), False), (32768, (43, 256, 256, 257), True), (640, (12, 64, 9), False)])
def test_packed_output_runs_without_column_zero(gpu, P, dims, in_place):
    """fused_mlp(pack=): the result is Zp = [pack | y_1 .. | spare]; the output layer runs without its column 0 forward and
    backward (python/renderer.py:186-193's perturbed pass needs the features only).  Value and every gradient against the fp64
    composite -- dL/dW[:, 0] and dL/db[0] are exactly zero --, with and without accumulate-in-place gradient buffers, on the 32-,
    64- and 128-point kernels (the packed rows start 3 floats into a row: unaligned 16-byte stores)."""
    from ndjir_amd import mlp
    rng = np.random.RandomState(P % 97)
    Ws = [torch.tensor(rng.randn(a, b) / np.sqrt(a), dtype=torch.float32, device=gpu, requires_grad=True) for a, b in zip(dims[:-1], dims[1:])]
    bs = [torch.tensor(rng.randn(b) * 0.1, dtype=torch.float32, device=gpu, requires_grad=True) for b in dims[1:]]
    x = torch.tensor(rng.randn(P, dims[0]), dtype=torch.float32, device=gpu, requires_grad=True)
    pk = torch.tensor(rng.randn(P, 3), dtype=torch.float32, device=gpu)
    No = dims[-1]
    bufs = []
    try:
        if in_place:
            for t in Ws + bs:
                bufs.append(torch.zeros_like(t))
                mlp.set_grad_buffer(t, bufs[-1])
        Zp = mlp.fused_mlp(x, Ws, bs, pack=pk)
        assert Zp.shape[0] == P and Zp.shape[1] % 4 == 0 and Zp.shape[1] >= 3 + No - 1
        g = torch.tensor(rng.randn(*Zp.shape), dtype=torch.float32, device=gpu)
        with torch.no_grad() if in_place else torch.enable_grad():
            grads = torch.autograd.grad(Zp, [x] + Ws + bs, g, allow_unused=True)
        got = [grads[0]] + ([b.clone() for b in bufs] if in_place else list(grads[1:]))
    finally:
        mlp.clear_grad_buffers()
    h = x.detach().double().requires_grad_(True)
    W64 = [w.detach().double().requires_grad_(True) for w in Ws]
    b64 = [b.detach().double().requires_grad_(True) for b in bs]
    a = h
    for j, (w, b) in enumerate(zip(W64, b64)):
        a = a @ w + b
        if j < len(W64) - 1:
            a = TF.softplus(a, beta=100)
    ref = torch.cat([pk.double(), a[:, 1:]], dim=1)
    assert torch.equal(Zp[:, :3], pk)
    assert float((Zp[:, 3:3 + No - 1].detach().double() - ref[:, 3:]).abs().max()) < 2e-5 * float(ref.abs().max())
    gref = torch.autograd.grad(ref, [h] + W64 + b64, g[:, :3 + No - 1].double())
    for i, (a_, b_) in enumerate(zip(got, gref)):
        assert a_ is not None, i
        assert float((a_.double() - b_).norm()) <= 3e-5 * float(b_.norm()), i
    assert float(got[len(Ws)][:, 0].abs().max()) == 0.0 and float(got[-1][0].abs()) == 0.0      # column 0 of the output layer


def test_rows_except_value_gradient_and_cache(gpu):
    """mlp.rows_except (the per-sample rows of a first-layer weight whose per-ray rows sit in the middle): value, gradient
    through autograd, gradient into an accumulate-in-place buffer, and the refresh of the cached copy after an in-place update."""
    from ndjir_amd import mlp
    W = torch.randn(12, 8, device=gpu, requires_grad=True)
    c = torch.randn(9, 8, device=gpu)
    y = mlp.rows_except(W, 3, 6)
    assert torch.equal(y, torch.cat([W[:3], W[6:]]).detach())
    g, = torch.autograd.grad((y * c).sum(), W)
    want = torch.zeros_like(W)
    want[:3], want[6:] = c[:3], c[3:]
    assert torch.equal(g, want)
    buf = torch.zeros(12, 8, device=gpu)
    try:
        mlp.set_grad_buffer(W, buf)
        y = mlp.rows_except(W, 3, 6)
        res = torch.autograd.grad((y * c).sum() + (W * 2).sum(), W)      # the stock path still returns its own part
        assert torch.equal(res[0], torch.full_like(W, 2.0)) and torch.equal(buf, want)
    finally:
        mlp.set_grad_buffer(W, None)
    with torch.no_grad():
        W.mul_(3.0)
    assert torch.equal(mlp.rows_except(W, 3, 6), torch.cat([W[:3], W[6:]]).detach())


def _group_case(rng, gpu, P, K, N, n_src, strided_out, accum, scale=1.0):
    """one output of a grouped weight-gradient launch: operands (strided views), the destination, its fp64 reference"""
    srcs, ref = [], torch.zeros(K, N, dtype=torch.float64)
    for i in range(n_src):
        Pi = P if i == 0 else max(P // 2 + 7, 1)
        A = torch.tensor(rng.randn(Pi, K + 3) * scale, dtype=torch.float32)
        B = torch.tensor(rng.randn(Pi, N + 5) * (3.0 + i), dtype=torch.float32)
        ref += A[:, :K].double().t() @ B[:, 2:2 + N].double()
        Ad, Bd = A.to(gpu)[:, :K], B.to(gpu)[:, 2:2 + N]
        srcs.append((Ad, Bd, Ad.abs().max().reshape(1), (Bd.abs().max() * 2.0).reshape(1)))
    base = torch.tensor(rng.randn(K, N + (1 if strided_out else 0)), dtype=torch.float32)
    full = base.to(gpu).clone() if accum else torch.full_like(base, float("nan")).to(gpu)
    out = full[:, 1:] if strided_out else full
    if accum:
        ref = ref + (base[:, 1:] if strided_out else base).double()
    return (out, accum, srcs), ref


@pytest.mark.parametrize("items", [64, 1024])
def test_wgrad_group_kernel(gpu, items, monkeypatch):
    """Many weight gradients in one launch (csrc/wgrad.hip k_wgrad_group + k_wgrad_group_reduce) vs fp64: main tiles, ragged
    strips in K and N, streaming outputs <= 8 wide, two operand pairs of different length per output, strided and unaligned
    destinations, accumulate and overwrite, more outputs than one argument block holds (24)."""
    from ndjir_amd import mlp
    monkeypatch.setattr(mlp, "WGRAD_GROUP_ITEMS", items)
    rng = np.random.RandomState(11)
    shapes = [(4096, 256, 256, 1), (20000, 259, 256, 2), (5000, 256, 257, 1), (3000, 43, 256, 1), (65536, 128, 6, 1), (1000, 301, 128, 2),
              (33, 5, 7, 1), (3000, 600, 2, 1), (100, 128, 9, 1), (5000, 290, 256, 1), (777, 52, 40, 1), (3000, 300, 300, 1), (65, 39, 1, 1),
              (2000, 128, 192, 2), (1000, 64, 64, 1), (70000, 262, 128, 1), (9000, 256, 213, 2), (512, 27, 256, 1)]
    jobs, refs = [], []
    for rep in range(2):            # 36 outputs, 46 operand pairs: three argument blocks
        for i, (P, K, N, n_src) in enumerate(shapes):
            job, ref = _group_case(rng, gpu, P, K, N, n_src, strided_out=(i % 3 == 1), accum=(i % 2 == 0), scale=10.0 ** (i % 5 - 2))
            jobs.append(job)
            refs.append(ref)
    mlp.wgrad_group(jobs)
    for (out, _, _), ref in zip(jobs, refs):
        err = float((out.cpu().double() - ref).norm() / ref.norm())
        assert err < 2e-6, (tuple(out.shape), err)


def test_wgrad_group_deferred_and_fallback(gpu):
    """`deferred_wgrads`: accumulating jobs are queued and launched at the end of the block, overwriting jobs at once; operand
    pairs without recorded maxima take the per-layer kernel; empty operand lists leave / zero the destination."""
    from ndjir_amd import mlp
    rng = np.random.RandomState(12)
    j_acc, r_acc = _group_case(rng, gpu, 3000, 128, 128, 1, False, True)
    j_new, r_new = _group_case(rng, gpu, 3000, 100, 36, 1, True, False)
    before = j_acc[0].clone()
    with mlp.deferred_wgrads():
        mlp.wgrad_group([j_acc])
        torch.cuda.synchronize()
        assert torch.equal(j_acc[0], before)                 # queued, not run
        mlp.wgrad_group([j_new])                             # overwriting: runs now
        assert float((j_new[0].cpu().double() - r_new).norm() / r_new.norm()) < 2e-6
    assert float((j_acc[0].cpu().double() - r_acc).norm() / r_acc.norm()) < 2e-6
    # no recorded maxima -> per-layer kernel (its own pre-pass), contiguous and strided destinations, accumulate and overwrite
    for strided, accum in ((False, True), (True, True), (True, False), (False, False)):
        (out, acc, srcs), ref = _group_case(rng, gpu, 2000, 130, 70, 2, strided, accum)
        mlp.wgrad_group([(out, acc, [(a, b, None, None) for a, b, _, _ in srcs])])
        assert float((out.cpu().double() - ref).norm() / ref.norm()) < 2e-6
    z = torch.ones(4, 5, device=gpu)
    mlp.wgrad_group([(z, True, [])])
    assert torch.equal(z, torch.ones_like(z))
    mlp.wgrad_group([(z, False, [])])
    assert torch.equal(z, torch.zeros_like(z))


def test_wgrad_group_same_and_overlapping_destinations(gpu):
    """Jobs of one grouped call that write the same memory: the same destination twice (a net that runs twice per step) is one
    reduction over both operand pairs; a parameter and its columns 1.. (the packed first-order pass) are reduced by separate
    launches -- the result is the sum either way, never a lost update."""
    from ndjir_amd import mlp
    rng = np.random.RandomState(13)
    K, N, P = 256, 257, 6000
    W0 = torch.tensor(rng.randn(K, N), dtype=torch.float32)
    Wg = W0.to(gpu).clone()
    ops, ref = [], W0.double().clone()
    for i, cols in enumerate((slice(0, N), slice(1, N), slice(0, N), slice(1, N))):
        A = torch.tensor(rng.randn(P, K), dtype=torch.float32)
        B = torch.tensor(rng.randn(P, cols.stop - cols.start), dtype=torch.float32)
        ref[:, cols] += A.double().t() @ B.double()
        Ad, Bd = A.to(gpu), B.to(gpu)
        ops.append((Wg[:, cols], True, [(Ad, Bd, Ad.abs().max().reshape(1), Bd.abs().max().reshape(1))]))
    with mlp.deferred_wgrads():
        for job in ops:
            mlp.wgrad_group([job])
    assert float((Wg.cpu().double() - ref).norm() / ref.norm()) < 2e-6


def _to_blocked(x):
    """(P, K) row-major -> the same P * K floats point-blocked: element (p, f) at ((p >> 5) * K + f) * 32 + (p & 31)."""
    P, K = x.shape
    return x.reshape(P // 32, 32, K).permute(0, 2, 1).contiguous().reshape(P, K)


@pytest.mark.parametrize("recorded", [True, False])
@pytest.mark.parametrize("wide", [True, False])
def test_wgrad_group_point_blocked_operands(gpu, wide, recorded, monkeypatch):
    """Grouped weight gradients with point-blocked operands (`mlp.PB`: what the 128-point-tile chains write) in every layout
    combination, vs fp64: full and ragged 128 x 128 tiles and 128 x 256 items (`k_wgrad_group_wide`; NDJIR_WGRAD_NO_WIDE is read
    once per process, so the second parametrisation shrinks the item count instead and covers the other split lengths), a feature
    range of a wider blocked buffer, narrow outputs with a blocked A, an odd number of 32-point chunks per item, two operand
    pairs per output.  recorded = False: no operand comes with a recorded maximum -- the call's pre-pass (`k_wgg_absmax`) finds them,
    over row-major and point-blocked operands and feature ranges of wider buffers alike."""
    from ndjir_amd import mlp
    monkeypatch.setattr(mlp, "WGRAD_GROUP_ITEMS", 0 if wide else 96)
    rng = np.random.RandomState(21)
    shapes = [(4096, 256, 256, 3), (4096 + 32 * 3, 256, 213, 3), (2048, 128, 128, 3), (4096, 256, 256, 1), (4096, 259, 256, 2), (8192, 262, 128, 2),
              (4096, 256, 257, 1), (2048, 256, 3, 1), (2048, 128, 1, 1), (4096, 213, 300, 3), (32, 256, 256, 3), (96, 128, 256, 3)]
    jobs, refs = [], []
    for i, (P, K, N, lay) in enumerate(shapes):
        srcs, ref = [], torch.zeros(K, N, dtype=torch.float64)
        for j in range(2 if i % 4 == 0 else 1):
            A = torch.tensor(rng.randn(P, K + 8) * 10.0 ** (i % 3 - 1), dtype=torch.float32)      # (feature range [4, 4 + K) of a wider buffer)
            B = torch.tensor(rng.randn(P, N) * (2.0 + j), dtype=torch.float32)
            ref += A[:, 4:4 + K].double().t() @ B.double()
            Ad = mlp.PB(_to_blocked(A.to(gpu)))[:, 4:4 + K] if lay & 1 else A.to(gpu)[:, 4:4 + K]
            Bd = mlp.PB(_to_blocked(B.to(gpu))) if lay & 2 else B.to(gpu)
            srcs.append((Ad, Bd, A.abs().max().reshape(1).to(gpu) if recorded else None, (B.abs().max() * 1.5).reshape(1).to(gpu) if recorded else None))
        accum = i % 2 == 0
        base = torch.tensor(rng.randn(K, N), dtype=torch.float32)
        out = base.to(gpu).clone() if accum else torch.full((K, N), float("nan"), device=gpu)
        jobs.append((out, accum, srcs))
        refs.append(ref + base.double() if accum else ref)
    mlp.wgrad_group(jobs)
    for (out, _, _), ref, sh in zip(jobs, refs, shapes):
        err = float((out.cpu().double() - ref).norm() / ref.norm())
        assert err < 2e-6, (sh, err)
