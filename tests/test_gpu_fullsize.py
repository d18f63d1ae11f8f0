"""Full-size checks at BASELINE.json's configuration (config/default.yaml: 512 rays x 128 samples, 512^3 x 4 voxel
grid), where the CPU oracle would take minutes: size-independent properties of the domain instead of a second
implementation -- additivity over ray shards (the multi-GPU invariant), sortedness of the samples, partition of
unity of the trilinear scatter, and the closed form of Adam's first step on the 2 GiB grid."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def step(gpu):
    import bench
    from ndjir_amd import config as cfg
    s = bench.Step(cfg.load("default"), 512, gpu, 0, 1)
    yield s
    from ndjir_amd import parameter as P
    from ndjir_amd.grid_feature import set_grad_buffer
    for p in s.grid_params:
        set_grad_buffer(p, None)
    P.clear_parameters()
    torch.cuda.empty_cache()


def test_samples_are_sorted_and_inside(step):
    from ndjir_amd.sampler import sample_points
    conf = step.conf
    with torch.no_grad():
        x_fg, t_fg, x_bg, t_bg, mask = sample_points(step.camloc, step.raydir, step.rand["stratified_sample"],
                                                         step.rand["background_sample"], conf)
    N = conf.renderer.n_samples0 + conf.renderer.n_samples1 * conf.renderer.n_upsamples
    assert x_fg.shape == (1, 512, N, 3) and t_fg.shape == (1, 512, N + 1, 1)
    hit = mask.reshape(-1) > 0
    assert 0.5 < float(hit.float().mean()) < 1.0            # the synthetic rays: most hit the box, some miss
    t = t_fg[0, :, :, 0][hit]
    assert bool((t[:, 1:] >= t[:, :-1]).all())               # sorted along the ray, t_far appended last
    r = conf.renderer.bounding_sphere_radius
    assert float(x_fg[0][hit].abs().max()) <= r * (1 + 1e-5)   # foreground samples stay inside the box
    assert bool((t_bg[0, :, 1:, 0] >= t_bg[0, :, :-1, 0]).all())
    np.testing.assert_allclose(x_bg[..., :3].norm(dim=-1).cpu().numpy(), 1.0, atol=1e-5)   # inverted-sphere parametrisation


def test_loss_and_gradients_add_over_ray_shards(step):
    """total_loss with the GLOBAL normalisers is a sum over ray shards: two 256-ray halves reproduce the 512-ray
    step (loss, every MLP gradient, the grid gradient) -- the invariant the multi-GPU exchange relies on."""
    from ndjir_amd.loss import total_loss
    from ndjir_amd.sampler import SamplePoints
    s = step
    loss_full = float(s.forward_backward())
    g_full = [g.clone() if g is not None else None for g in s.grads]
    from ndjir_amd import mlp
    mlp.clear_grad_buffers()        # the shard passes below take their gradients from autograd (the step's bucket is filled in place)
    grid_full = {k: v.clone() for k, v in s.grid_bufs.items()}
    with torch.no_grad():
        _, _, mask = SamplePoints(s.conf).t_near_far(s.camloc, s.raydir)
    msum = mask.sum().reshape(())
    for buf in s.grid_bufs.values():
        buf.zero_()
    loss_sum, g_sum = 0.0, [None] * len(g_full)
    for h in range(2):
        sl = slice(256 * h, 256 * (h + 1))
        rand = {k: v[:, sl].contiguous() for k, v in s.rand.items()}
        out = total_loss(s.camloc, s.raydir[:, sl].contiguous(), s.color_gt[:, sl].contiguous(), None, s.car, s.conf,
                         rand, ray_shards=2, mask_sum_global=msum)
        grads = torch.autograd.grad(out["loss"], s.mlp_params + s.grid_params, allow_unused=True)
        loss_sum += float(out["loss"].detach())
        for i, g in enumerate(grads):
            if g is not None:
                g_sum[i] = g.clone() if g_sum[i] is None else g_sum[i] + g
    assert loss_sum == pytest.approx(loss_full, rel=2e-6)
    for name, a, b in zip(s.mlp_names, g_sum, g_full):
        if b is None:
            assert a is None, name
            continue
        scale = max(float(b.abs().max()), 1e-12)
        assert float((a - b).abs().max()) <= 2e-4 * scale, name
    for k, v in s.grid_bufs.items():
        scale = float(grid_full[k].abs().max())
        assert scale > 0 and float((v - grid_full[k]).abs().max()) <= 2e-4 * scale, k
        assert int((grid_full[k] != 0).any(dim=-1).sum()) < 3 * 512 * 128 * 8      # sparse: only touched cells


def test_trilinear_scatter_is_a_partition_of_unity(gpu):
    """grad_feature on the 512^3 x 4 grid: the 8 trilinear weights of a point sum to one, so the scattered
    gradient sums to the sum of the incoming gradients, channel by channel (checksum of checksums); and a
    constant grid interpolates to the constant."""
    import ndjir_amd.grid_feature.voxel_feature as VF
    from ndjir_amd import lib
    G, D, P = 512, 4, 65536
    gen = torch.Generator(device=gpu).manual_seed(412)
    q = (torch.rand(P, 3, device=gpu, generator=gen) * 2 - 1) * 0.999
    go = torch.randn(P, D, device=gpu, generator=gen)
    gf = torch.zeros(G, G, G, D, device=gpu)
    lib.call("voxel_feature_grad_feature", P * D, gf, go, q, [G] * 3, D, [-1] * 3, [1] * 3, 0, 1)
    want = go.double().sum(0)
    got = gf.view(-1, D).double().sum(0)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=1e-3 * float(go.abs().sum(0).max()) * 1e-3)
    assert int((gf != 0).any(dim=-1).sum()) <= 8 * P
    gf.fill_(0.75)
    out = torch.empty(P, D, device=gpu)
    lib.call("voxel_feature_query_on_voxel", P * D, out, q, gf, [G] * 3, D, [-1] * 3, [1] * 3, 0)
    np.testing.assert_allclose(out.cpu().numpy(), 0.75, rtol=2e-6)
    assert VF is not None


def test_adam_first_step_on_the_full_grid(gpu):
    """2 GiB parameter, first step from m = v = 0 without decay: w moves by -alpha_1 (1-b1) g / (sqrt((1-b2) g^2) + eps)
    where g != 0 and not at all elsewhere; the gradient buffer comes back zero."""
    from ndjir_amd import lib
    n = 512 ** 3 * 4
    w = torch.full((n,), 0.25, device=gpu)
    g = torch.zeros(n, device=gpu)
    idx = torch.arange(0, n, 1009, device=gpu)
    g[idx] = ((idx % 7).float() - 3.0) * 1e-3               # values in {-3..3} * 1e-3, some exactly zero
    gref = g[idx].clone()
    m, v = torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
    st = torch.zeros(4, device=gpu)
    st[0] = 1e-2
    lib.call("solver_adam_begin", st, 0.9, 0.999, None, None)
    lib.call("solver_adam", n, w, g, m, v, 0.0, 0.9, 0.999, 1e-8, 0.0, 1, st)
    assert float(g.abs().max()) == 0.0
    a1 = float(st[2])
    f32 = lambda x: float(np.float32(x))                     # the kernel receives float arguments
    assert a1 == pytest.approx(f32(1e-2) * math.sqrt(1 - f32(0.999)) / (1 - f32(0.9)), rel=2e-7)
    moved = w != 0.25
    assert int(moved.sum()) == int((gref != 0).sum())
    b1, b2 = np.float32(0.9), np.float32(0.999)
    gr = gref.cpu().numpy()
    mm = (np.float32(1) - b1) * gr
    vv = (np.float32(1) - b2) * gr * gr
    want = np.float32(0.25) - np.float32(a1) * mm / (np.sqrt(vv) + np.float32(1e-8))
    np.testing.assert_array_equal(w[idx].cpu().numpy(), want)
    assert float(m.abs().sum()) == pytest.approx(float(np.abs(mm).astype(np.float64).sum()), rel=1e-5)


# ---- BASELINE.json config 3: config/triplaneline.yaml, 1024 rays x 128 samples, 3 x 2048^2 x 8 tri-plane + 3 x 2048 x 8 tri-line
@pytest.fixture(scope="module")
def step_tpl(gpu):
    import bench
    from ndjir_amd import config as cfg
    s = bench.Step(cfg.load("triplaneline"), 1024, gpu, 0, 1)
    yield s
    from ndjir_amd import parameter as P
    from ndjir_amd.grid_feature import set_grad_buffer
    for p in s.grid_params:
        set_grad_buffer(p, None)
    P.clear_parameters()
    torch.cuda.empty_cache()


def test_cfg3_loss_and_gradients_add_over_ray_shards(step_tpl):
    """cfg3 at full size: two 512-ray halves with the global normalisers reproduce the 1024-ray step -- loss, every MLP
    gradient, and the 403 MB tri-plane / tri-line gradient buffers."""
    from ndjir_amd.loss import total_loss
    from ndjir_amd.sampler import SamplePoints
    s = step_tpl
    v = s.conf.geometric_network.voxel
    assert v.type == "triplaneline" and v.grid_size == 2048 and v.feature_size == 8
    shapes = sorted(tuple(p.shape) for p in s.grid_params)
    assert shapes == [(3, 2048, 8), (3, 2048, 2048, 8)]
    loss_full = float(s.forward_backward())
    assert math.isfinite(loss_full)
    g_full = [g.clone() if g is not None else None for g in s.grads]
    from ndjir_amd import mlp
    mlp.clear_grad_buffers()        # the shard passes below take their gradients from autograd (the step's bucket is filled in place)
    grid_full = {k: b.clone() for k, b in s.grid_bufs.items()}
    with torch.no_grad():
        _, _, mask = SamplePoints(s.conf).t_near_far(s.camloc, s.raydir)
    msum = mask.sum().reshape(())
    for buf in s.grid_bufs.values():
        buf.zero_()
    loss_sum, g_sum = 0.0, [None] * len(g_full)
    for h in range(2):
        sl = slice(512 * h, 512 * (h + 1))
        rand = {k: t[:, sl].contiguous() for k, t in s.rand.items()}
        out = total_loss(s.camloc, s.raydir[:, sl].contiguous(), s.color_gt[:, sl].contiguous(), None, s.car, s.conf,
                         rand, ray_shards=2, mask_sum_global=msum)
        grads = torch.autograd.grad(out["loss"], s.mlp_params + s.grid_params, allow_unused=True)
        loss_sum += float(out["loss"].detach())
        for i, g in enumerate(grads):
            if g is not None:
                g_sum[i] = g.clone() if g_sum[i] is None else g_sum[i] + g
    assert loss_sum == pytest.approx(loss_full, rel=2e-6)
    for name, a, b in zip(s.mlp_names, g_sum, g_full):
        if b is None:
            assert a is None, name
            continue
        assert float((a - b).abs().max()) <= 2e-4 * max(float(b.abs().max()), 1e-12), name
    for k, b in s.grid_bufs.items():
        scale = float(grid_full[k].abs().max())
        assert scale > 0 and float((b - grid_full[k]).abs().max()) <= 2e-4 * scale, k


def test_cfg3_scatter_is_a_partition_of_unity(gpu):
    """grad_feature on the 3 x 2048^2 x 8 tri-plane and the 3 x 2048 x 8 tri-line: the bilinear / linear weights of a
    point sum to one per plane / line, so each plane's scattered gradient sums to the incoming gradient of its output
    channels (checksum of checksums); a constant grid interpolates to the constant.  Output channel = d*3 + plane."""
    from ndjir_amd import lib
    G, D, P = 2048, 8, 65536
    gen = torch.Generator(device=gpu).manual_seed(412)
    q = (torch.rand(P, 3, device=gpu, generator=gen) * 2 - 1) * 0.999
    go = torch.randn(P, D * 3, device=gpu, generator=gen)
    for fam, shape in (("triplane", (3, G, G, D)), ("triline", (3, G, D))):
        gf = torch.zeros(shape, device=gpu)
        lib.call(f"{fam}_feature_grad_feature", P * D * 3, gf, go, q, G, D, [-1] * 3, [1] * 3, 0, 1)
        want = go.double().reshape(P, D, 3).sum(0).t()             # (plane, d)
        got = gf.reshape(3, -1, D).double().sum(1)
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=1e-6 * float(go.abs().sum(0).max()))
        gf.fill_(0.75)
        out = torch.empty(P, D * 3, device=gpu)
        lib.call(f"{fam}_feature_query_on_{fam}", P * D * 3, out, q, gf, G, D, [-1] * 3, [1] * 3, 0)
        np.testing.assert_allclose(out.cpu().numpy(), 0.75, rtol=2e-6)


# ---- BASELINE.json configs 1 and 5 at full ray counts: 512 rays x 64 samples (no up-sampling) and a 4000-ray tile x 256 samples
@pytest.mark.parametrize("name,ov,R,N", [("cfg1", ["renderer.n_upsamples=0"], 512, 64),
                                         ("cfg5", ["renderer.n_samples0=128", "renderer.n_samples1=32"], 4000, 256)])
def test_cfg1_cfg5_forward_at_full_size(gpu, name, ov, R, N):
    """Sampler + pb_render forward on the 512^3 x 4 grid at the configs' ray / sample counts: shapes, sortedness, pixel
    range, and tile additivity (the first half of the rays rendered alone gives the same pixels)."""
    from ndjir_amd import config as cfg, network, parameter as P
    from ndjir_amd.renderer import make_rand, pb_render
    from ndjir_amd.sampler import sample_points
    from ndjir_amd.synthetic import make_rays
    conf = cfg.load("default", ov)
    P.clear_parameters()
    P.set_device(gpu)
    network.seed(313)
    camloc, raydir, _ = make_rays(1, R, seed=412, device=gpu)
    rand = make_rand(1, R, conf, gpu)
    one = torch.ones(1, device=gpu)

    def render(sl):
        rd = raydir[:, sl].contiguous()
        rn = {k: v[:, sl].contiguous() for k, v in rand.items()}
        with torch.no_grad():
            x_fg, t_fg, x_bg, t_bg, mask = sample_points(camloc, rd, rn["stratified_sample"], rn["background_sample"], conf)
            res = pb_render(x_fg, t_fg, x_bg, t_bg, camloc, rd, mask, one, conf, rn, render_only=True)
        return x_fg, t_fg, mask, res["color_pixel"]
    x_fg, t_fg, mask, col = render(slice(0, R))
    assert x_fg.shape == (1, R, N, 3) and t_fg.shape == (1, R, N + 1, 1) and col.shape == (1, R, 3)
    hit = mask.reshape(-1) > 0
    t = t_fg[0, :, :, 0][hit]
    assert bool((t[:, 1:] >= t[:, :-1]).all())
    assert bool(torch.isfinite(col).all()) and float(col.min()) >= -1e-6
    _, _, _, col_half = render(slice(0, R // 2))
    assert float((col_half - col[:, :R // 2]).abs().max()) <= 1e-6
    P.clear_parameters()
    torch.cuda.empty_cache()
