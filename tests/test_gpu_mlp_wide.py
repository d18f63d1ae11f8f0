"""The 128-point-tile chain kernels (ndjir_amd/csrc/mlp3w.hip: operands swapped, epilogue in the accumulator registers;
csrc/mlp3p.hip: the same tile with its two halves software-pipelined, one accumulator per block -- TRAINING passes of nets wider
than 128 columns with point-blocked side tensors) against the 64-point-tile kernel (mlp3.hip) and against fp64: a point's FORWARD
result of a pass WITHOUT side tensors must be bit-identical whichever kernel evaluates it (the sampler merges SDF values of
different launches bit for bit); training passes, the backward / tangent chains agree to round-off.  `ndjir_mlp_set_tile_rows(128)` sends launches of any size that the wide kernel supports (P % 128 == 0, hidden
layers of 2..8 column blocks, planes within the LDS) to it; by default only launches of >= 32768 points go there."""
import numpy as np
import pytest
import torch

from tests.test_gpu_mlp import make, ref_mlp

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 7], ids=["chainw", "chainp"])
def tile(request):
    """Setter of the forced tile height; every test that takes it runs twice -- with the training passes of nets wider than 128
    columns on mlp3w.hip's kernel (the default) and on the software-pipelined kernel (mlp3p.hip; `set_chain_pipeline(7)`)."""
    from ndjir_amd import mlp
    old, old_pipe = mlp.get_tile_rows(), mlp.get_chain_pipeline()
    mlp.set_chain_pipeline(request.param)
    yield mlp.set_tile_rows
    mlp.set_tile_rows(old)
    mlp.set_chain_pipeline(old_pipe)


CASES = [
    ((259, 256, 256, 256, 3), 256, -1),       # base colour net: K0 = 259 -> 272 columns of planes, narrow output
    ((39, 128, 128, 128, 1), 384, -1),        # environment light net: two waves per column block
    ((262, 128, 128, 128, 6), 128, -1),       # specular reflectance net
    ((52, 256, 256, 256, 257), 256, -1),      # background geometric net: 257-wide output = 9 column blocks, two rounds
    ((43, 256, 256, 256, 213, 256, 256, 256, 257), 256, 3),   # geometric net: skip concatenation, a 7-block layer
    ((43, 256, 256, 256, 213, 256, 256, 256, 1), 128, 3),     # sdf-only variant (sampler)
    ((43, 128, 128, 129), 256, -1),           # 5-block output layer behind 4-block hidden layers (rounds of 4)
    ((43, 96, 96, 40), 128, -1),              # 3 column blocks: one idle wave pair; 2-block output
    ((32, 160, 160, 97), 128, -1),            # 5 column blocks: three idle waves
]


@pytest.mark.parametrize("dims,P,skip", CASES)
def test_forward_is_bitwise_the_64_point_kernel(gpu, tile, dims, P, skip):
    from ndjir_amd.mlp import chain_forward
    scale = 1.0 / np.sqrt(2.0) if skip >= 0 else 1.0
    Ws, bs = make(dims, 5, skip)
    rng = np.random.RandomState(17)
    x = torch.tensor(rng.randn(P, dims[0]) * 0.5, dtype=torch.float32).to(gpu)
    Wd = [w.to(gpu) for w in Ws]
    bd = [b.to(gpu) for b in bs]
    res = {}
    for rows in (64, 32, 128):
        tile(rows)
        y, hidden, am = chain_forward(x, Wd, bd, 100.0, skip, scale, keep_hidden=True)
        y_ng, _, _ = chain_forward(x, Wd, bd, 100.0, skip, scale, keep_hidden=False)
        torch.cuda.synchronize()
        res[rows] = (y.clone(), [h.clone() for h in hidden], am.clone(), y_ng.clone())
    y64 = ref_mlp(x.cpu().double(), [w.double() for w in Ws], [b.double() for b in bs], 100.0, skip, scale)
    assert float((res[128][0].cpu().double() - y64).norm() / y64.norm()) < 2e-6
    for other in (64, 32):
        # (row-major side tensors: mlp3w.hip's kernel at 128 rows, bit-identical by construction; so is the pass without side tensors)
        assert torch.equal(res[128][0], res[other][0]), f"output differs from the {other}-point tiles"
        assert torch.equal(res[128][3], res[other][3]) and torch.equal(res[128][3], res[128][0]), "pass without side tensors differs"
        for j, (h_w, h_o) in enumerate(zip(res[128][1], res[other][1])):
            assert torch.equal(h_w, h_o), f"stored activation {j + 1} differs from the {other}-point tiles"
        assert torch.equal(res[128][2], res[other][2]), "recorded maxima differ"
    # the training pass with POINT-BLOCKED side tensors: nets wider than 128 columns run on the pipelined kernel (mlp3p.hip, one
    # accumulator per block) -- equal to round-off; narrower nets stay on mlp3w.hip -- equal bit for bit
    tile(128)
    yb, hb, amb = chain_forward(x, Wd, bd, 100.0, skip, scale, keep_hidden=True, blocked=True)
    torch.cuda.synchronize()
    from ndjir_amd import mlp
    wide = max(dims[1:-1]) > 128 and mlp.get_chain_pipeline() != 0

    def unblock(t):          # element (p, f) of a point-blocked (P, W) buffer lives at ((p >> 5) * W + f) * 32 + (p & 31)
        Pn, Wn = t.shape
        return t.reshape(Pn // 32, Wn, 32).permute(0, 2, 1).reshape(Pn, Wn)

    def rel(a_, b_):
        return float((a_.double() - b_.double()).norm() / max(float(b_.double().norm()), 1e-30))
    if wide:
        assert rel(yb, res[64][0]) < 2e-6, rel(yb, res[64][0])
        assert float((yb.cpu().double() - y64).norm() / y64.norm()) < 2e-6
        for j, (h_b, h_o) in enumerate(zip(hb, res[64][1])):
            assert rel(unblock(h_b), h_o) < 2e-6, (j, rel(unblock(h_b), h_o))
        assert torch.allclose(amb, res[64][2], rtol=1e-5, atol=0)        # (recorded maxima: positive floats)
    else:
        assert torch.equal(yb, res[64][0])
        for j, (h_b, h_o) in enumerate(zip(hb, res[64][1])):
            assert torch.equal(unblock(h_b), h_o), j


@pytest.mark.parametrize("dims,P,skip", CASES)
def test_gradients_match_fp64_and_the_64_point_kernel(gpu, tile, dims, P, skip):
    from ndjir_amd.mlp import fused_mlp
    scale = 1.0 / np.sqrt(2.0) if skip >= 0 else 1.0
    Ws, bs = make(dims, 7, skip)
    rng = np.random.RandomState(11)
    x = torch.tensor(rng.randn(P, dims[0]) * 0.5, dtype=torch.float32)
    gy = torch.tensor(rng.randn(P, dims[-1]), dtype=torch.float32)

    def run(rows):
        tile(rows)
        xd = x.to(gpu).requires_grad_(True)
        Wd = [w.to(gpu).requires_grad_(True) for w in Ws]
        bd = [b.to(gpu).requires_grad_(True) for b in bs]
        y = fused_mlp(xd, Wd, bd, 100.0, skip, scale)
        return y, torch.autograd.grad(y, [xd] + Wd + bd, gy.to(gpu))

    y_w, g_w = run(128)
    y_o, g_o = run(64)
    x64 = x.double().requires_grad_(True)
    W64 = [w.double().requires_grad_(True) for w in Ws]
    b64 = [b.double().requires_grad_(True) for b in bs]
    y64 = ref_mlp(x64, W64, b64, 100.0, skip, scale)
    g64 = torch.autograd.grad(y64, [x64] + W64 + b64, gy.double())

    def rel(a, b):
        return float((a.detach().cpu().double() - b.detach().cpu().double()).norm() / max(float(b.norm()), 1e-30))

    from ndjir_amd import mlp
    if max(dims[1:-1]) > 128 and mlp.get_chain_pipeline() != 0:      # (training pass of a net wider than 128 columns on the pipelined kernel: equal to round-off)
        assert rel(y_w, y_o) < 2e-6, rel(y_w, y_o)
    else:
        assert torch.equal(y_w, y_o)
    names = ["x"] + [f"W{j}" for j in range(len(Ws))] + [f"b{j}" for j in range(len(bs))]
    for n, a, o, r in zip(names, g_w, g_o, g64):
        assert rel(a, r) < 2e-5, (n, rel(a, r))
        assert rel(a, o) < 5e-6, (n, rel(a, o))


def test_row_term_and_accumulating_output(gpu, tile):
    """The per-row-group term of the first layer (soft-visibility / photogrammetric nets) and a backward chain that
    accumulates into an existing dL/dx (MultiMLP), on the wide kernel."""
    from ndjir_amd.mlp import fused_mlp
    P, div, K0, Dh, No = 512, 128, 39, 128, 1
    Ws, bs = make((K0, Dh, Dh, Dh, No), 3)
    rng = np.random.RandomState(5)
    x = torch.tensor(rng.randn(P, K0) * 0.5, dtype=torch.float32)
    rb = torch.tensor(rng.randn(P // div, Dh) * 0.3, dtype=torch.float32)
    gy = torch.tensor(rng.randn(P, No), dtype=torch.float32)
    outs = {}
    for rows in (128, 64):
        tile(rows)
        xd = x.to(gpu).requires_grad_(True)
        rbd = rb.to(gpu).requires_grad_(True)
        Wd = [w.to(gpu).requires_grad_(True) for w in Ws]
        bd = [b.to(gpu).requires_grad_(True) for b in bs]
        y = fused_mlp(xd, Wd, bd, 100.0, row_bias=rbd, row_bias_div=div)
        outs[rows] = (y, torch.autograd.grad(y, [xd, rbd] + Wd + bd, gy.to(gpu)))
    assert torch.equal(outs[128][0], outs[64][0])
    for a, o in zip(outs[128][1], outs[64][1]):
        assert float((a - o).norm() / max(float(o.norm()), 1e-30)) < 5e-6
    # fp64
    x64, rb64 = x.double().requires_grad_(True), rb.double().requires_grad_(True)
    W64 = [w.double().requires_grad_(True) for w in Ws]
    h = torch.nn.functional.softplus(x64 @ W64[0] + bs[0].double() + rb64.repeat_interleave(div, dim=0), beta=100.0)
    y64 = ref_mlp(h, W64[1:], [b.double() for b in bs[1:]], 100.0)
    g64 = torch.autograd.grad(y64, [x64, rb64] + W64, gy.double())
    assert float((outs[128][0].cpu().double() - y64).norm() / y64.norm()) < 2e-6
    for a, r in zip(outs[128][1][:2 + len(Ws)], g64):
        assert float((a.cpu().double() - r).norm() / r.norm()) < 2e-5


@pytest.mark.parametrize("grid", [True, False])
def test_geometric_double_backward_on_the_wide_kernel(gpu, tile, grid):
    """sdf chain, tangent chain and the augmented backward chain of the geometric network's main pass (nn.grad) on
    128-point tiles: same values as on 64-point tiles to round-off."""
    from ndjir_amd.geometric import geometric_main
    rng = np.random.RandomState(2)
    P, M = 256, 6
    K0 = 3 + 6 * M + (4 if grid else 0)
    dims = (K0, 256, 256, 256, 256 - K0, 256, 256, 256, 257)
    Ws, bs = make(dims, 4, 3)
    x = torch.tensor(rng.rand(P, 3) * 1.6 - 0.8, dtype=torch.float32)
    F = torch.tensor(rng.randn(16, 16, 16, 4) * 0.05, dtype=torch.float32)
    cot = [torch.tensor(rng.randn(P, c), dtype=torch.float32) for c in (1, 256, 3)]
    res = {}
    for rows in (128, 64):
        tile(rows)
        Wd = [w.to(gpu).requires_grad_(True) for w in Ws]
        bd = [b.to(gpu).requires_grad_(True) for b in bs]
        Fd = F.to(gpu).requires_grad_(True)
        sdf, feat, n, Z = geometric_main(x.to(gpu), Fd if grid else None, Wd, bd, M, 3, 1.0 / np.sqrt(2.0))
        loss = (sdf * cot[0].to(gpu)).sum() + (feat * cot[1].to(gpu)).sum() + (n * cot[2].to(gpu)).sum()
        g = torch.autograd.grad(loss, Wd + bd + ([Fd] if grid else []))
        res[rows] = (sdf, feat, n, g)
    from ndjir_amd import mlp
    if mlp.get_chain_pipeline() != 0:      # (the main pass at 128 rows is a training pass of a 256-wide net: the pipelined kernel -- equal to round-off)
        for i in (0, 1):
            assert float((res[128][i] - res[64][i]).norm()) <= 2e-6 * float(res[64][i].norm()), i
    else:
        assert torch.equal(res[128][0], res[64][0]) and torch.equal(res[128][1], res[64][1])
    assert float((res[128][2] - res[64][2]).abs().max()) <= 1e-5 * float(res[64][2].abs().max())
    for a, o in zip(res[128][3], res[64][3]):
        assert float((a - o).norm() / max(float(o.norm()), 1e-30)) < 2e-5


# ---- the grid-stride tile loop of the wide kernel (mlp3w.hip: `for (tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)`, the
# xpar / cur ping-pong of the row maxima, bias-gradient sums carried across a workgroup's tiles): a workgroup takes a second
# tile only when a launch has more than 2048 tiles, or more than CHAIN_MAX_GRID_BG = 512 when it produces bias gradients --
# which is what the bench's 65 536 / 131 072-point backward launches do.  fp64 references run on the device (torch stock ops).
def _rel64(a, b):
    return float((a.detach().double() - b.detach()).norm() / max(float(b.detach().norm()), 1e-30))


@pytest.mark.parametrize("dims,skip", [((259, 256, 256, 256, 3), -1), ((39, 128, 128, 128, 1), -1),
                                       ((43, 256, 256, 256, 213, 256, 256, 256, 257), 3)])
def test_tile_loop_forward_2100_tiles(gpu, tile, dims, skip):
    """Forward chain at P = 128 x 2100: every workgroup loops over >= 2 tiles; output and every stored activation vs fp64,
    and bitwise against the 64-point kernel (one tile per workgroup step there too, but a different kernel)."""
    from ndjir_amd.mlp import chain_forward
    P = 128 * 2100
    scale = 1.0 / np.sqrt(2.0) if skip >= 0 else 1.0
    Ws, bs = make(dims, 21, skip)
    g = torch.Generator(device="cpu").manual_seed(5)
    x = (torch.randn(P, dims[0], generator=g) * 0.5).to(gpu)
    Wd, bd = [w.to(gpu) for w in Ws], [b.to(gpu) for b in bs]
    tile(0)                                   # the default dispatch: P >= 32768, P % 128 == 0 -> mlp3w.hip
    y, hidden, am = chain_forward(x, Wd, bd, 100.0, skip, scale, keep_hidden=True)
    h = x.double()
    for j, (W, b) in enumerate(zip(Wd, bd)):
        z = h @ W.double() + b.double()
        if j == len(Wd) - 1:
            assert _rel64(y, z) < 2e-6
            break
        h = torch.nn.functional.softplus(z, beta=100.0)
        if j == skip:
            h = torch.cat([h, x.double()], dim=-1) * scale
        assert _rel64(hidden[j], h) < 2e-6, j
        assert abs(float(am[j + 1]) - float(hidden[j].abs().max())) <= 1e-6 * float(hidden[j].abs().max())
    tile(64)
    y64, hidden64, _ = chain_forward(x, Wd, bd, 100.0, skip, scale, keep_hidden=True)
    assert torch.equal(y, y64) and all(torch.equal(a, b) for a, b in zip(hidden, hidden64))
    # the same tile loop with point-blocked side tensors: nets wider than 128 columns on the pipelined kernel (mlp3p.hip)
    tile(0)
    yb, hb, _ = chain_forward(x, Wd, bd, 100.0, skip, scale, keep_hidden=True, blocked=True)
    assert _rel64(yb, y64.double()) < 2e-6
    for j, (a, b) in enumerate(zip(hb, hidden64)):
        Pn, Wn = a.shape
        assert _rel64(a.reshape(Pn // 32, Wn, 32).permute(0, 2, 1).reshape(Pn, Wn), b.double()) < 2e-6, j


@pytest.mark.parametrize("dims,skip", [((259, 256, 256, 256, 3), -1), ((39, 128, 128, 128, 1), -1), ((262, 128, 128, 128, 6), -1)])
def test_tile_loop_backward_with_bias_gradients_1100_tiles(gpu, tile, dims, skip):
    """Backward chain WITH bias gradients at P = 128 x 1100: the launch is clamped to 512 workgroups, each loops over 2-3
    tiles and carries its bias-gradient sums across them; dL/dx, every dL/dW, every dL/db vs fp64 autograd."""
    from ndjir_amd.mlp import fused_mlp
    P = 128 * 1100
    Ws, bs = make(dims, 23, skip)
    g = torch.Generator(device="cpu").manual_seed(7)
    x = (torch.randn(P, dims[0], generator=g) * 0.5).to(gpu)
    gy = torch.randn(P, dims[-1], generator=g).to(gpu)
    tile(0)
    xd = x.clone().requires_grad_(True)
    Wd = [w.to(gpu).requires_grad_(True) for w in Ws]
    bd = [b.to(gpu).requires_grad_(True) for b in bs]
    y = fused_mlp(xd, Wd, bd, 100.0)
    grads = torch.autograd.grad(y, [xd] + Wd + bd, gy)
    x64 = x.double().requires_grad_(True)
    W64 = [w.to(gpu).double().requires_grad_(True) for w in Ws]
    b64 = [b.to(gpu).double().requires_grad_(True) for b in bs]
    y64 = ref_mlp(x64, W64, b64, 100.0)
    g64 = torch.autograd.grad(y64, [x64] + W64 + b64, gy.double())
    assert _rel64(y, y64) < 2e-6
    names = ["x"] + [f"W{j}" for j in range(len(Ws))] + [f"b{j}" for j in range(len(bs))]
    for n, a, r in zip(names, grads, g64):
        assert _rel64(a, r) < 2e-5, (n, _rel64(a, r))


def test_tile_loop_tangent_chain_1100_tiles(gpu, tile):
    """The geometric network's main pass at P = 128 x 1100 (no grid): sdf chain, TANGENT chain (mode 2, with its column sum
    for the last layer -> clamped to 512 workgroups, 2-3 tiles each) and the augmented backward chain on the wide kernel;
    sdf, feature, n and every parameter gradient of a loss through all three vs fp64 autograd (nn.grad's double backward,
    python/renderer.py:52)."""
    from ndjir_amd.geometric import geometric_main
    from oracle import graph as G
    P, M = 128 * 1100, 6
    K0 = 3 + 6 * M
    dims = (K0, 256, 256, 256, 256 - K0, 256, 256, 256, 257)
    Ws, bs = make(dims, 29, 3)
    c = 1.0 / np.sqrt(2.0)
    g = torch.Generator(device="cpu").manual_seed(9)
    x = (torch.rand(P, 3, generator=g) * 1.6 - 0.8).to(gpu)
    cot = [torch.randn(P, k, generator=g).to(gpu) for k in (1, 256, 3)]
    tile(0)
    Wd = [w.to(gpu).requires_grad_(True) for w in Ws]
    bd = [b.to(gpu).requires_grad_(True) for b in bs]
    sdf, feat, n, Z = geometric_main(x, None, Wd, bd, M, 3, c)
    loss = (sdf * cot[0]).sum() + (feat * cot[1]).sum() + ((n * cot[2]).sum(-1) ** 2).sum()
    grads = torch.autograd.grad(loss, Wd + bd)
    x64 = x.double().requires_grad_(True)
    W64 = [w.to(gpu).double().requires_grad_(True) for w in Ws]
    b64 = [b.to(gpu).double().requires_grad_(True) for b in bs]
    y = ref_mlp(G.positional_encoding(x64, M), W64, b64, 100.0, 3, c)
    sdf64, feat64 = y[:, :1], y[:, 1:]
    n64, = torch.autograd.grad(sdf64.sum(), x64, create_graph=True)
    loss64 = (sdf64 * cot[0].double()).sum() + (feat64 * cot[1].double()).sum() + ((n64 * cot[2].double()).sum(-1) ** 2).sum()
    g64 = torch.autograd.grad(loss64, W64 + b64)
    assert _rel64(sdf, sdf64) < 1e-5 and _rel64(feat, feat64) < 1e-5 and _rel64(n, n64) < 2e-5
    names = [f"W{j}" for j in range(8)] + [f"b{j}" for j in range(8)]
    for nm, a, r in zip(names, grads, g64):
        assert _rel64(a, r) < 1e-3, (nm, _rel64(a, r))        # (fp32 vs fp64 through the beta = 100 second-order terms)


def _material_like_nets(gpu, seed, ld):
    """Five nets on one packed input (P, ld), shaped like the per-sample material nets of the default configuration: three
    128-wide ones on 22 columns, a 256-wide one on 22 and a 256-wide one on 23 columns with a per-group row term."""
    def net(K, Dh, L, No, s):
        r = np.random.RandomState(seed + s)
        dims = [K] + [Dh] * L + [No]
        return ([torch.tensor(r.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i]), dtype=torch.float32, device=gpu).requires_grad_(True)
                 for i in range(len(dims) - 1)],
                [torch.tensor(r.randn(dims[i + 1]) * 0.1, dtype=torch.float32, device=gpu).requires_grad_(True) for i in range(len(dims) - 1)])
    return [net(22, 128, 3, 2, 1), net(22, 256, 4, 3, 2), net(22, 128, 3, 2, 3), net(22, 128, 3, 6, 4), net(23, 256, 4, 3, 5)]


def _run_multi(gpu, P, ld, div, grouped):
    from ndjir_amd import mlp
    nets = _material_like_nets(gpu, 40, ld)
    rng = np.random.RandomState(9)
    x = torch.tensor(rng.randn(P, ld), dtype=torch.float32, device=gpu, requires_grad=True)
    rt = torch.tensor(rng.randn(P // div, 256) * 0.3, dtype=torch.float32, device=gpu, requires_grad=True)
    call = [(W, b) for W, b in nets[:4]] + [(nets[4][0], [None] + nets[4][1][1:])]
    old = mlp._NO_CHAIN_GROUP
    mlp._NO_CHAIN_GROUP = not grouped
    mlp.PROFILE = []
    try:
        ys = mlp.multi_mlp(x, call, widths=[22, 22, 22, 22, 23], row_terms=[None, None, None, None, (rt, div)])
        gs = [torch.tensor(np.random.RandomState(70 + i).randn(*y.shape), dtype=torch.float32, device=gpu) for i, y in enumerate(ys)]
        leaves = [x, rt] + [t for W, b in nets for t in W + b if t is not nets[4][1][0]]
        grads = torch.autograd.grad(ys, leaves, gs)
        torch.cuda.synchronize()
        launches = [(e[0], e[7], e[4]) for e in mlp.PROFILE if e[0].startswith("chain")]
    finally:
        mlp._NO_CHAIN_GROUP = old
        mlp.PROFILE = None
    return [y.detach().cpu() for y in ys], [g.cpu() for g in grads], launches


def test_chain_group_is_bitwise_the_separate_launches(gpu, tile):
    """mlp.chain_group / ndjir_mlp_chain_group_begin..end: the five per-sample nets of one MultiMLP as two launches forward
    (the three 128-wide nets; the two 256-wide ones) and two backward, a workgroup looping over its tile's nets -- outputs,
    the shared input gradient (assigned by the first net of the first launch, accumulated by all the others), the row-term
    gradient and every weight gradient equal the five separate launches bit for bit (bias gradients, summed by LDS atomics
    in either form, to round-off), at 3 tiles per workgroup."""
    tile(128)
    P, ld, div = 128 * 1600, 24, 128
    y1, g1, l1 = _run_multi(gpu, P, ld, div, grouped=False)
    y2, g2, l2 = _run_multi(gpu, P, ld, div, grouped=True)
    assert len([l for l in l1 if l[0] == "chain_fwd"]) == 5 and len([l for l in l1 if l[0] == "chain_bwd"]) == 5
    assert len([l for l in l2 if l[0] == "chain_fwd"]) == 2 and len([l for l in l2 if l[0] == "chain_bwd"]) == 2, l2
    assert all(l[1] == 1 for l in l2), l2            # (the library issued ONE kernel launch per group)
    for a, b in zip(y1, y2):
        assert torch.equal(a, b)
    for a, b in zip(g1, g2):
        if a.dim() == 1:         # bias gradients: LDS float atomics, whose order varies between any two launches
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
        else:
            assert torch.equal(a, b)


def test_chain_group_bracket_falls_back_and_reports(gpu, tile):
    """The C bracket itself: calls the wide kernel cannot take together (different numbers of points; a 64-point-tile launch)
    are launched one by one in call order, `launches` says how many kernels ran; a second _begin, an _end without a group
    and a 17th recorded call are argument errors."""
    import ctypes
    from ndjir_amd import lib, mlp
    so = lib.load()
    so.ndjir_mlp_chain_group_end.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = ctypes.c_int(-1)
    assert so.ndjir_mlp_chain_group_end(stream, ctypes.byref(n)) != 0            # no group open
    dims = (40, 128, 128, 3)
    Ws, bs = make(dims, 3, -1)
    Ws, bs = [w.to(gpu) for w in Ws], [b.to(gpu) for b in bs]
    xa = torch.randn(128 * 300, 40, device=gpu)
    xb = torch.randn(128 * 301, 40, device=gpu)
    ref = [mlp.chain_forward(x, Ws, bs)[0] for x in (xa, xa, xb)]
    assert so.ndjir_mlp_chain_group_begin() == 0
    assert so.ndjir_mlp_chain_group_begin() != 0                                    # already open
    got = [mlp.chain_forward(x, Ws, bs)[0] for x in (xa, xa, xb)]                 # recorded
    assert so.ndjir_mlp_chain_group_end(stream, ctypes.byref(n)) == 0
    assert n.value == 2                  # (xa, xa) share a launch; xb has another number of points
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    tile(64)                             # forced 64-point tiles: nothing is grouped
    assert so.ndjir_mlp_chain_group_begin() == 0
    got = [mlp.chain_forward(x, Ws, bs)[0] for x in (xa, xa)]
    assert so.ndjir_mlp_chain_group_end(stream, ctypes.byref(n)) == 0 and n.value == 2
    for a, b in zip(ref[:2], got):
        assert torch.equal(a, b)
    tile(128)
    assert so.ndjir_mlp_chain_group_begin() == 0
    with pytest.raises(lib.NdjirHipError):
        for _ in range(17):
            mlp.chain_forward(xa, Ws, bs)
    assert so.ndjir_mlp_chain_group_end(stream, ctypes.byref(n)) != 0               # the error is reported again at the end
    torch.cuda.synchronize()
