"""GPU parity of the whole hot path: sampler -> pb_render -> total_loss -> backward through the
HIP product vs the CPU oracle, on identical rays, random tensors and parameters.
Tolerances from BASELINE.json north_star: pixel RGB and loss within 1e-4 relative (fp32)."""
import numpy as np
import pytest
import torch

from tests.parity_utils import rel_err, run_oracle_step, run_product_step, small_conf

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-4
PIXEL_TOL = 1e-4
GRAD_RTOL = 2e-3   # fp32 round-off through double backward; fp32-vs-fp64 oracle itself shows ~1e-4


@pytest.fixture
def tile_rows(request):
    """Points per workgroup tile of the chain kernels for the test (0 = chosen per launch).  128 sends every launch the wide
    kernel supports to csrc/mlp3w.hip -- the kernel bench.py times; at the parity tests' 4 096 points the default would
    pick csrc/mlp3.hip's 64 / 32-point tiles."""
    from ndjir_amd import mlp
    old = mlp.get_tile_rows()
    mlp.set_tile_rows(request.param)
    yield request.param
    mlp.set_tile_rows(old)


@pytest.fixture
def chain_pipeline(request):
    """Mode mask of the software-pipelined chain kernel (csrc/mlp3p.hip) for the test; 0 = mlp3w.hip's kernel (the default)."""
    from ndjir_amd import mlp
    old = mlp.get_chain_pipeline()
    mlp.set_chain_pipeline(request.param)
    yield request.param
    mlp.set_chain_pipeline(old)


@pytest.mark.parametrize("variant,G,tile_rows,chain_pipeline",
                         [("default", 32, 0, 0), ("no_voxel", 8, 0, 0), ("triplaneline", 64, 0, 0), ("custom", 32, 0, 0),
                          ("ste", 32, 0, 0), ("default", 32, 128, 0), ("no_voxel", 8, 128, 0), ("triplaneline", 64, 128, 0),
                          ("default", 32, 128, 7), ("no_voxel", 8, 128, 7), ("custom", 32, 128, 7)],
                         indirect=["tile_rows", "chain_pipeline"])
def test_step_parity_given_samples(gpu, variant, G, tile_rows, chain_pipeline):
    """Renderer + loss + backward parity with the oracle fed the product's sample points.  `ste` = config/ste.yaml
    (`voxel.use_ste`: the grid lookups stay out of n = d(sdf)/dx, python/grid_feature/voxel_feature.py:383-399).
    tile_rows = 128: the same step with every chain launch (forward, backward, tangent; 4 096 sample points, 8 192 light
    directions, 1 024 background samples -- all multiples of 128) on the 128-point-tile kernel; chain_pipeline = 7: the training
    passes of the nets wider than 128 columns (geometric main pass incl. its tangent and augmented backward chains, base colour,
    photogrammetric, background) on the software-pipelined kernel (csrc/mlp3p.hip: one accumulator per block)."""
    conf = small_conf(grid_size=G, n_rays=16, variant=variant)
    prod = run_product_step(conf, B=2, R=16, device=gpu)
    s = prod["samples"]
    ref = run_oracle_step(conf, prod["params_cpu"], prod["inputs_cpu"],
                          samples=(s["x_fg"], s["t_fg"], s["x_bg"], s["t_bg"], s["mask"]))
    l0, l1 = float(prod["loss"]), float(ref["loss"])
    assert abs(l0 - l1) <= LOSS_RTOL * abs(l1), (l0, l1)
    for k, v in ref["terms"].items():
        assert abs(float(prod["terms"][k]) - float(v)) <= 2e-4 * max(abs(float(v)), 1e-3), k
    dc = (prod["color_pixel"].cpu() - ref["color_pixel"]).abs().max()
    assert float(dc) <= PIXEL_TOL, float(dc)
    # gradients: against the fp32 oracle within GRAD_RTOL, or -- where fp32 itself is ill-conditioned (custom.yaml's 1e-4
    # roughness prior: the fp32 oracle is 3e-3 off its own fp64 evaluation) -- within 3x the fp32 oracle's distance to fp64
    ref64 = None
    for k, g in ref["grads"].items():
        gp = prod["grads"][k]
        assert (g is None) == (gp is None), k
        if g is None:
            continue
        e = rel_err(gp, g)
        if e >= GRAD_RTOL:
            if ref64 is None:
                ref64 = run_oracle_step(conf, prod["params_cpu"], prod["inputs_cpu"], dtype=torch.float64,
                                        samples=(s["x_fg"], s["t_fg"], s["x_bg"], s["t_bg"], s["mask"]))
            e64, o64 = rel_err(gp, ref64["grads"][k]), rel_err(g, ref64["grads"][k])
            assert e64 <= 3 * o64, (k, e, e64, o64)


def test_ste_fused_path_matches_layer_by_layer(gpu):
    """config/ste.yaml through the fused geometric operator (J_e without the grid part) and through the layer-by-layer
    autograd path (the grid operator's nn.grad backward returning None for the query): same loss and gradients."""
    from ndjir_amd import network
    conf = small_conf(grid_size=32, n_rays=8, variant="ste")
    assert network.uses_fused_geometric(conf)
    fused = run_product_step(conf, B=1, R=8, device=gpu)
    network.USE_FUSED = False
    try:
        plain = run_product_step(conf, B=1, R=8, device=gpu)
    finally:
        network.USE_FUSED = True
    assert abs(float(fused["loss"]) - float(plain["loss"])) <= 2e-5 * abs(float(plain["loss"]))
    for k, g in plain["grads"].items():
        gp = fused["grads"][k]
        assert (g is None) == (gp is None), k
        if g is not None:
            assert rel_err(gp, g) < GRAD_RTOL, (k, rel_err(gp, g))


@pytest.mark.parametrize("switch", ["NDJIR_NO_FUSED_LIGHTS", "NDJIR_NO_BACKGROUND_HEAD"])
def test_fused_ray_operators_equal_their_stock_spelling(gpu, switch, monkeypatch):
    """volume.direct_light (light nets' output activations + both light integrals + pixel composition) and
    volume.background_head (+ the lighting net's per-ray row term) against the same step with that stage spelled the
    reference's way -- slices of the 2 M light tensors / the (B,R,N,287) background concatenation, stock activations."""
    conf = small_conf(grid_size=16, n_rays=8)
    fused = run_product_step(conf, B=1, R=8, device=gpu)
    monkeypatch.setenv(switch, "1")
    plain = run_product_step(conf, B=1, R=8, device=gpu)
    assert abs(float(fused["loss"]) - float(plain["loss"])) <= 2e-6 * abs(float(plain["loss"]))
    assert float((fused["color_pixel"] - plain["color_pixel"]).abs().max()) <= 2e-6
    for k, g in plain["grads"].items():
        gp = fused["grads"][k]
        assert (g is None) == (gp is None), k
        if g is not None:
            assert rel_err(gp, g) < 2e-4, (k, rel_err(gp, g))


@pytest.mark.parametrize("fused_tail", [True, False])
def test_prior_normaliser_without_eikonal_term(gpu, fused_tail, monkeypatch):
    """python/loss.py:36, 72, 118: with train.eikonal_weight = 0 the priors are divided by sum(mask) n_samples0, not by
    sum(mask) times the sample count -- in the fused loss tail (csrc/loss.hip, N_prior) and in the stock-op branch."""
    from ndjir_amd import loss as L
    monkeypatch.setattr(L, "_NO_FUSED_TAIL", not fused_tail)
    conf = small_conf(grid_size=16, n_rays=8, overrides=["train.eikonal_weight=0", "train.base_color_prior_weight=1.0",
                                                         "train.roughness_prior_weight=0.01"])
    prod = run_product_step(conf, B=1, R=8, device=gpu)
    s = prod["samples"]
    ref = run_oracle_step(conf, prod["params_cpu"], prod["inputs_cpu"],
                          samples=(s["x_fg"], s["t_fg"], s["x_bg"], s["t_bg"], s["mask"]))
    l0, l1 = float(prod["loss"]), float(ref["loss"])
    assert abs(l0 - l1) <= LOSS_RTOL * abs(l1), (l0, l1)
    assert float(prod["terms"]["loss_eikonal"]) == 0.0
    for k, v in ref["terms"].items():
        assert abs(float(prod["terms"][k]) - float(v)) <= 2e-4 * max(abs(float(v)), 1e-3), k
    for k, g in ref["grads"].items():
        gp = prod["grads"][k]
        assert (g is None) == (gp is None), k
        if g is not None:
            assert rel_err(gp, g) < 3 * GRAD_RTOL, (k, rel_err(gp, g))


# BASELINE.json config 1 (no up-sampling rounds: 64 samples / ray) and config 5's sampler setting
# (renderer.n_samples0=128, n_samples1=32: rounds at 128/160/192/224 -> 256 samples, render_image.py's shape)
CFG1 = ["renderer.n_upsamples=0"]
CFG5 = ["renderer.n_samples0=128", "renderer.n_samples1=32"]


@pytest.mark.parametrize("name,ov,N", [("cfg1", CFG1, 64), ("cfg5", CFG5, 256)])
def test_step_parity_other_sample_counts(gpu, name, ov, N):
    """Whole step (sampler -> pb_render -> total_loss -> backward) at 64 and at 256 samples per ray: bit-exact
    sample indices on the product's per-round inputs, then loss / terms / pixels / gradients vs the oracle."""
    from oracle import graph as G
    conf = small_conf(grid_size=32, n_rays=12, overrides=ov)
    rec = {}
    prod = run_product_step(conf, B=1, R=12, device=gpu, record=rec)
    s = prod["samples"]
    assert s["x_fg"].shape == (1, 12, N, 3) and s["t_fg"].shape == (1, 12, N + 1, 1)
    tnear, tfar, _ = G.t_near_far(prod["inputs_cpu"]["camloc"], prod["inputs_cpu"]["raydir"], conf)
    assert len(rec.get("idx", [])) == conf.renderer.n_upsamples
    for u in range(conf.renderer.n_upsamples):
        t_in, sdf = rec["t_in"][u].cpu(), rec["sdf"][u].cpu()
        t_out, idx = G.importance_round(t_in, sdf, tnear.reshape(1, 12, 1, 1), tfar.reshape(1, 12, 1, 1),
                                        conf.renderer.sampling_sigmoid_gain * 2 ** u, conf.renderer.n_samples1)
        assert torch.equal(rec["idx"][u].cpu(), idx), f"round {u}: sample indices differ"
        assert torch.equal(rec["t_out"][u].cpu(), t_out), f"round {u}: merged distances differ"
    ref = run_oracle_step(conf, prod["params_cpu"], prod["inputs_cpu"],
                          samples=(s["x_fg"], s["t_fg"], s["x_bg"], s["t_bg"], s["mask"]))
    l0, l1 = float(prod["loss"]), float(ref["loss"])
    assert abs(l0 - l1) <= LOSS_RTOL * abs(l1), (l0, l1)
    for k, v in ref["terms"].items():
        assert abs(float(prod["terms"][k]) - float(v)) <= 2e-4 * max(abs(float(v)), 1e-3), k
    assert float((prod["color_pixel"].cpu() - ref["color_pixel"]).abs().max()) <= PIXEL_TOL
    for k, g in ref["grads"].items():
        gp = prod["grads"][k]
        assert (g is None) == (gp is None), k
        if g is not None:
            assert rel_err(gp, g) < GRAD_RTOL, (k, rel_err(gp, g))


@pytest.mark.parametrize("ov", [["specular_brdf.model=ue4"], ["specular_brdf.model=ue4", "specular_brdf.sampling=uniform"],
                                ["specular_brdf.sampling=uniform"], ["specular_brdf.use_split_sum=True"],
                                ["specular_brdf.model=ue4", "specular_brdf.use_split_sum=True"],
                                ["specular_reflectance_network.fixme=true", "train.specular_reflectance_prior_weight=0"], ["implicit_illumination_network.use_me=false"],
                                ["roughness_network.use_normal=false", "photogrammetric_light_network.use_inverse_distance=false"]],
                         ids=["ue4-importance", "ue4-uniform", "filament-uniform", "filament-split-sum", "ue4-split-sum", "fixed-specular",
                              "no-implicit-light", "mixed-prefix-widths"])
def test_step_parity_brdf_variants(gpu, ov):
    """The non-default BRDF branches (python/specular_brdf.py:121-191 ue4; uniform sampling :104-110; split sum
    python/renderer.py:152-154) run on the templated kernel pair csrc/render.hip k_specular_light_g (round 5; the default filament +
    importance integral keeps its own kernels); a fixed specular reflectance / no implicit light take the renderer's net-by-net branch instead of the
    fused material head (the fixed reflectance has no std output, so its prior term is switched off: python/loss.py:154 divides
    by it); a roughness net without the normal and a photogrammetric net without the inverse distance change the prefix
    widths the nets read of the packed sample inputs (262 / 259 / 259 / 262 / 262 columns): whole-step parity with the oracle,
    gradients included.  (base_color_network.use_normal=true is not a case: the reference's perturbed pass hands that net
    normal=None, python/renderer.py:193, and fails in its concatenation exactly like the product.)"""
    conf = small_conf(grid_size=16, n_rays=8, overrides=ov)
    prod = run_product_step(conf, B=1, R=8, device=gpu)
    s = prod["samples"]
    ref = run_oracle_step(conf, prod["params_cpu"], prod["inputs_cpu"],
                          samples=(s["x_fg"], s["t_fg"], s["x_bg"], s["t_bg"], s["mask"]))
    l0, l1 = float(prod["loss"]), float(ref["loss"])
    assert abs(l0 - l1) <= LOSS_RTOL * abs(l1), (l0, l1)
    assert float((prod["color_pixel"].cpu() - ref["color_pixel"]).abs().max()) <= PIXEL_TOL
    ref64 = None
    for k, g in ref["grads"].items():
        gp = prod["grads"][k]
        assert (g is None) == (gp is None), k
        if g is None:
            continue
        e = rel_err(gp, g)
        if e >= GRAD_RTOL:
            if ref64 is None:
                ref64 = run_oracle_step(conf, prod["params_cpu"], prod["inputs_cpu"], dtype=torch.float64,
                                        samples=(s["x_fg"], s["t_fg"], s["x_bg"], s["t_bg"], s["mask"]))
            e64, o64 = rel_err(gp, ref64["grads"][k]), rel_err(g, ref64["grads"][k])
            assert e64 <= 3 * o64, (k, e, e64, o64)


def test_sampler_parity(gpu):
    """Sample indices: the oracle's importance round applied to the product's own per-round
    (t, sdf) gives the SAME integer indices and bit-identical merged distances."""
    from oracle import graph as G
    conf = small_conf(grid_size=32, n_rays=64)
    rec = {}
    prod = run_product_step(conf, B=1, R=64, device=gpu, backward=False, record=rec)
    tnear, tfar, _ = G.t_near_far(prod["inputs_cpu"]["camloc"], prod["inputs_cpu"]["raydir"], conf)
    for u in range(conf.renderer.n_upsamples):
        t_in, sdf = rec["t_in"][u].cpu(), rec["sdf"][u].cpu()
        B, R, N, _ = t_in.shape
        t_out, idx = G.importance_round(t_in, sdf, tnear.reshape(B, R, 1, 1), tfar.reshape(B, R, 1, 1),
                                        conf.renderer.sampling_sigmoid_gain * 2 ** u, conf.renderer.n_samples1)
        assert torch.equal(rec["idx"][u].cpu(), idx), f"round {u}: sample indices differ"
        assert torch.equal(rec["t_out"][u].cpu(), t_out), f"round {u}: merged distances differ"


@pytest.mark.parametrize("tile_rows", [0, 128], indirect=True)
def test_end_to_end_including_sampler(gpu, tile_rows):
    """Full path with each side running its own sampler (tile_rows = 128: the sampler's SDF evaluations -- 2 048 and 512
    points per round -- and the renderer's nets on the 128-point-tile kernel)."""
    conf = small_conf(grid_size=32, n_rays=32)
    prod = run_product_step(conf, B=1, R=32, device=gpu, backward=False)
    ref = run_oracle_step(conf, prod["params_cpu"], prod["inputs_cpu"], backward=False)
    x0, x1 = prod["samples"]["x_fg"].detach().cpu(), ref["out"]["x_fg"].detach()
    # each side evaluates its own SDF network inside the sampler: fp32 round-off may move a sample
    # across a bin edge; such samples must stay rare and small
    d = (x0 - x1).abs().amax(-1)
    assert float((d > 1e-4).float().mean()) < 0.01 and float(d.max()) < 5e-2
    assert abs(float(prod["loss"]) - float(ref["loss"])) <= 5e-4 * abs(float(ref["loss"]))
    assert float((prod["color_pixel"].cpu() - ref["color_pixel"]).abs().max()) <= 5e-4


def test_render_image_tiles(gpu):
    """renderer.render_image: tiled forward, output (1,3,H,W) in [0,1]; tiles are independent."""
    from ndjir_amd import network, parameter as P
    from ndjir_amd.renderer import render_image
    conf = small_conf(grid_size=16, n_rays=16, overrides=["valid.n_rays=48", "valid.n_down_samples=0"])
    P.clear_parameters()
    P.set_device(gpu)
    network.seed(313)
    pose = np.eye(4, dtype=np.float64)[None]
    pose[0, :3, 3] = [0.0, 0.0, -2.5]
    K = np.array([[[20.0, 0, 8], [0, 20.0, 6], [0, 0, 1]]])
    img = render_image(pose, K, (16, 12), conf, device=gpu)
    assert img.shape == (1, 3, 12, 16) and np.isfinite(img).all() and img.min() >= 0 and img.max() <= 1
    # tiles shard round-robin over ranks: the partial images of a 3-way shard add up to the frame
    parts = [render_image(pose, K, (16, 12), conf, device=gpu, rank=r, world=3, reduce=False) for r in range(3)]
    np.testing.assert_array_equal(parts[0] + parts[1] + parts[2], img)


def test_checkpoint_h5_round_trip_renders_the_same_frame(gpu, tmp_path):
    """python/train.py:100-101 -> python/render_image.py:43: parameters saved as nnabla `.h5` and loaded into an empty
    registry render the identical frame (names, order, need_grad flags and every byte of every array survive)."""
    from ndjir_amd import network, parameter as P
    from ndjir_amd.renderer import render_image
    conf = small_conf(grid_size=16, n_rays=16, overrides=["valid.n_rays=48", "valid.n_down_samples=0"])
    P.clear_parameters()
    P.set_device(gpu)
    network.seed(313)
    pose, K = _camera()
    img = render_image(pose, K, (16, 12), conf, device=gpu)
    before = {k: (v.detach().cpu().clone(), v.requires_grad) for k, v in P.get_parameters().items()}
    path = str(tmp_path / "model_00001.h5")
    P.save_parameters(path)
    P.clear_parameters()
    P.load_parameters(path, device=gpu)
    after = P.get_parameters()
    assert list(after) == list(before)
    for k, (v, need) in before.items():
        assert after[k].is_cuda and torch.equal(after[k].cpu(), v) and after[k].requires_grad == need, k
    np.testing.assert_array_equal(render_image(pose, K, (16, 12), conf, device=gpu), img)


def _camera():
    pose = np.eye(4, dtype=np.float64)[None]
    pose[0, :3, 3] = [0.1, -0.05, -2.5]
    K = np.array([[[22.0, 0.3, 8], [0, 21.0, 6], [0, 0, 1]]])
    return pose, K


@pytest.mark.parametrize("name,ov,res,n_rays", [
    ("default", [], (16, 12), 52),          # 192 px, tile P = 52 - mod(192, 52) = 16 -> 12 full tiles
    ("padded", [], (15, 9), 100),           # 135 px, P = 100 - 35 = 65 -> 3 tiles, the last one holds 5 pixels
    ("cfg5", CFG5, (8, 6), 48),             # 256 samples per ray, one 48-ray tile
    ("cfg1", CFG1, (8, 6), 48),             # 64 samples per ray
])
def test_render_image_parity(gpu, name, ov, res, n_rays):
    """renderer.render_image (tiled forward, render_only branch, rays generated on the device) against the oracle's
    restatement of python/renderer.py:212-272 (host rays in float64, per-tile sample_points -> pb_render)."""
    from oracle import graph as G
    from ndjir_amd import network, parameter as P
    from ndjir_amd.renderer import make_rand, pb_render, render_image
    from ndjir_amd.sampler import sample_points
    from ndjir_amd.synthetic import make_rays
    conf = small_conf(grid_size=16, n_rays=16, overrides=[f"valid.n_rays={n_rays}", "valid.n_down_samples=0"] + ov)
    P.clear_parameters()
    P.set_device(gpu)
    network.seed(313)
    # parameters are created on first use: one throw-away tile creates them all
    camloc, raydir, _ = make_rays(1, 4, seed=1, device=gpu)
    rand4 = make_rand(1, 4, conf, gpu)
    x_fg, t_fg, x_bg, t_bg, mask = sample_points(camloc, raydir, rand4["stratified_sample"], rand4["background_sample"], conf)
    full = pb_render(x_fg.requires_grad_(True), t_fg, x_bg, t_bg, camloc, raydir, mask, torch.ones(1, device=gpu), conf, rand4)
    # the render_only branch (no perturbed base colour, renderer.py: only color_pixel is forwarded) gives the same pixels
    ro = pb_render(x_fg, t_fg, x_bg, t_bg, camloc, raydir, mask, torch.ones(1, device=gpu), conf, rand4, render_only=True)
    assert float((ro["color_pixel"] - full["color_pixel"]).abs().max()) <= 1e-6
    params_cpu = {k: v.detach().cpu().clone() for k, v in P.get_parameters().items()}

    pose, K = _camera()
    img = render_image(pose, K, res, conf, device=gpu)
    W, H = res
    m = (W * H) % n_rays
    tile = n_rays - m
    rand_cpu = {k: v.cpu() for k, v in make_rand(1, tile, conf, gpu).items()}
    ref = G.render_image(pose, K, res, rand_cpu, params_cpu, conf)
    assert img.shape == ref.shape == (1, 3, H, W)
    d = np.abs(img - ref)
    # each side runs its own sampler: a sample that sits on a CDF bin edge may move (see test_sampler_end_to_end_indices);
    # such pixels must stay rare and small, all others agree to the pixel tolerance
    assert float((d > PIXEL_TOL).mean()) <= 0.01, float((d > PIXEL_TOL).mean())
    assert float(d.max()) <= 5e-4, float(d.max())


def test_bench_step_graph_replay_matches_eager(gpu):
    """bench.py times the step replayed from one captured HIP graph: the replayed step must produce the
    eager step's loss and gradients (incl. the sparse re-zeroing of the grid gradient buffer between steps)."""
    import bench
    from ndjir_amd import config as cfg
    conf = cfg.load("default", ["geometric_network.voxel.grid_size=64"])
    step = bench.Step(conf, 64, gpu, 0, 1)
    for _ in range(2):
        loss_e = step.forward_backward()
    eager = [g.clone() if g is not None else None for g in step.grads]
    grid_e = {k: v.clone() for k, v in step.grid_bufs.items()}
    graph, loss_g = bench.capture_step(step)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert abs(float(loss_g) - float(loss_e)) <= 1e-6 * abs(float(loss_e))
    for a, b in zip(step.grads, eager):
        if b is None:
            continue
        assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-12)
    for k, v in step.grid_bufs.items():
        assert float((v - grid_e[k]).abs().max()) <= 1e-5 * max(float(grid_e[k].abs().max()), 1e-12), k
    # replay -> eager steps -> replay (a replay that followed eager steps once produced another loss -- memset
    # nodes of the library, since replaced by kernels): state shared between eager launches and the captured graph
    # (library scratch, the operand-maximum arena, packed weights, the gradient bucket) must not leak either way
    for _ in range(2):
        loss_e2 = step.forward_backward()
    assert abs(float(loss_e2) - float(loss_e)) <= 1e-6 * abs(float(loss_e))
    graph.replay()
    torch.cuda.synchronize()
    assert abs(float(loss_g) - float(loss_e)) <= 1e-6 * abs(float(loss_e))
    for a, b in zip(step.grads, eager):
        if b is not None:
            assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-12)


@pytest.mark.parametrize("case", ["all_miss", "single_ray", "ragged_rays"])
def test_step_edge_cases(gpu, case):
    """Whole step (sampler included) on degenerate ray batches vs the oracle: every ray missing the box (all masks zero:
    the mask-normalised terms divide by 1e-5 only), a single ray, and a ray count that fills no tile (R = 7)."""
    from ndjir_amd import network, parameter as P
    from ndjir_amd.loss import total_loss
    from ndjir_amd.renderer import make_rand
    from oracle import graph as G
    conf = small_conf(grid_size=16)
    R = {"all_miss": 8, "single_ray": 1, "ragged_rays": 7}[case]
    rng = np.random.RandomState(7)
    camloc = np.array([[0.3, -2.4, 0.5]], np.float32)
    tgt = rng.rand(1, R, 3).astype(np.float32) * 1.6 - 0.8
    d = tgt - camloc[:, None]
    if case == "all_miss":
        d = -d                                              # looking away from the box
    raydir = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    color_gt = rng.rand(1, R, 3).astype(np.float32)
    P.clear_parameters()
    P.set_device(gpu)
    network.seed(313)
    rand = make_rand(1, R, conf, gpu)
    car = torch.tensor([0.6], device=gpu)
    T = lambda a: torch.from_numpy(a).to(gpu)
    out = total_loss(T(camloc), T(raydir), T(color_gt), None, car, conf, rand)
    params = P.get_parameters()
    names = [k for k, v in params.items() if v.requires_grad]
    grads = torch.autograd.grad(out["loss"], [params[k] for k in names], allow_unused=True)
    mask = out["samples"]["mask"]
    assert float(mask.sum()) == (0.0 if case == "all_miss" else float(R))
    pc = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in params.items()}
    ref = G.total_loss(torch.from_numpy(camloc), torch.from_numpy(raydir), torch.from_numpy(color_gt), None, car.cpu(),
                       {k: v.cpu() for k, v in rand.items()}, pc, conf)
    rgrads = torch.autograd.grad(ref["loss"], [pc[k] for k in names], allow_unused=True)
    assert np.isfinite(float(out["loss"].detach()))
    assert abs(float(out["loss"]) - float(ref["loss"])) <= 2e-4 * abs(float(ref["loss"]))
    assert float((out["render"]["color_pixel"].cpu() - ref["render"]["color_pixel"]).abs().max()) <= 2e-4
    for k, a, b in zip(names, grads, rgrads):
        if b is None or float(b.abs().max()) == 0.0:
            assert a is None or float(a.abs().max()) <= 1e-6, k
            continue
        assert a is not None and bool(torch.isfinite(a).all()), k
        assert rel_err(a, b) <= 5e-3, (k, rel_err(a, b))


def test_training_iterations_on_the_device_data_feed(gpu):
    """The pieces of the reference's training loop (python/train.py:124-148) end to end on the GPU: IDRRaySource batches
    -> Step.set_rays -> forward + backward -> weight decay, guard and the two Adam solvers.  A uniformly coloured scene is
    fitted: the RGB loss must fall and nothing may turn non-finite."""
    from ndjir_amd import config as cfg, parameter as P
    from ndjir_amd.dataset import IDRRaySource
    from ndjir_amd.grid_feature import set_grad_buffer
    from ndjir_amd.step import Step
    R = 64
    conf = cfg.load("default", ["geometric_network.voxel.grid_size=32", f"train.n_rays={R}", "train.batch_size=1"])
    M, H, W = 4, 24, 32
    rng = np.random.RandomState(3)
    images = np.broadcast_to(np.array([0.8, 0.3, 0.1], np.float32), (M, H, W, 3)).copy()
    masks = np.ones((M, H, W, 1))
    poses = np.zeros((M, 4, 4))
    Ks = np.zeros((M, 3, 3))
    for m in range(M):
        c = rng.randn(3)
        c = 2.5 * c / np.linalg.norm(c)
        fwd = -c / np.linalg.norm(c)
        right = np.cross(fwd, rng.randn(3))
        right /= np.linalg.norm(right)
        poses[m, :3, 0], poses[m, :3, 1], poses[m, :3, 2], poses[m, :3, 3] = right, np.cross(fwd, right), fwd, c
        poses[m, 3, 3] = 1
        Ks[m] = [[1.5 * W, 0, W / 2], [0, 1.5 * W, H / 2], [0, 0, 1]]
    src = IDRRaySource(images, masks, Ks, poses, conf, rng=np.random.RandomState(313), device=gpu)
    step = Step(conf, R, gpu, 0, 1)
    try:
        step.enable_training()
        gen = torch.Generator(device=gpu).manual_seed(0)
        first = last = None
        for it in range(40):
            color, mask, raydir, camloc = src.next_batch(1)
            step.set_rays(camloc, raydir, color)
            step.redraw_rand(gen)
            loss = float(step.train_step())
            assert np.isfinite(loss), it
            first = loss if first is None else first
            last = loss
        assert step.solvers.solver_feat.step_count() == 40 and not step.solvers.solver_feat.skipped()
        print(f"loss {first:.4f} -> {last:.4f}")
        assert last < 0.7 * first, (first, last)
        for k, p in P.get_parameters().items():
            assert bool(torch.isfinite(p).all()), k
    finally:
        for p in step.grid_params:
            set_grad_buffer(p, None)
        P.clear_parameters()


def test_training_iterations_from_a_scene_directory(gpu, tmp_path):
    """f2 end to end: a synthetic scene on disk in the IDR / DTU layout (image/, mask/, cameras.npz with world_mat_i and a
    scale_mat_i of non-unit scale and translation) -> `IDRRaySource.from_path` (camera decode without OpenCV) -> device ray
    feed -> three training iterations.  The rays of the device feed equal the reference's numpy ray generation on the
    decoded cameras (python/helper.py:44-73), colours / masks are the image's pixels."""
    from ndjir_amd import config as cfg, parameter as P
    from ndjir_amd.dataset import IDRRaySource
    from ndjir_amd.grid_feature import set_grad_buffer
    from ndjir_amd.helper import generate_raydir_camloc
    from ndjir_amd.step import Step
    from tests.test_dataset_cpu import write_idr_scene
    R = 64
    conf = cfg.load("default", ["geometric_network.voxel.grid_size=32", f"train.n_rays={R}", "train.batch_size=1"])
    path = str(tmp_path / "scan")
    images, masks, cams, _ = write_idr_scene(path, M=3, H=24, W=32, seed=4)
    src = IDRRaySource.from_path(path, conf, rng=np.random.RandomState(313), device=gpu)
    assert src.size == 3 and abs(float(src.scale) - 2.5) < 1e-6
    img, idx = src.pixel_indices(0)
    color, mask, raydir, camloc = src.next_batch(1)
    xy = np.stack([idx % 32, idx // 32], axis=-1)[None]
    rd, cl = generate_raydir_camloc(src.poses[img:img + 1].cpu().numpy(), src.intrinsics[img:img + 1].cpu().numpy(), xy)
    assert float(np.abs(raydir.cpu().numpy() - rd).max()) <= 1e-6 and float(np.abs(camloc.cpu().numpy() - cl).max()) <= 1e-6
    np.testing.assert_allclose(camloc.cpu().numpy()[0], cams[img][2], atol=1e-4)           # the camera the scene was written with
    np.testing.assert_allclose(color.cpu().numpy()[0], images[img].reshape(-1, 3)[idx] / 255.0, atol=1e-6)
    assert np.array_equal(mask.cpu().numpy()[0, :, 0], (masks[img].reshape(-1)[idx] > 127.5) * 1.0)
    step = Step(conf, R, gpu, 0, 1)
    try:
        step.enable_training()
        gen = torch.Generator(device=gpu).manual_seed(0)
        for it in range(3):
            color, mask, raydir, camloc = src.next_batch(1)
            step.set_rays(camloc, raydir, color)
            step.redraw_rand(gen)
            assert np.isfinite(float(step.train_step())), it
        assert step.solvers.solver_feat.step_count() == 3 and not step.solvers.solver_feat.skipped()
    finally:
        for p in step.grid_params:
            set_grad_buffer(p, None)
        P.clear_parameters()


@pytest.mark.parametrize("gtype,G,D", [("cosine_voxel", 16, 4), ("lanczos_triplaneline", 32, 4), ("cosine_triplaneline", 32, 8),
                                       ("triplane", 32, 8)])
def test_step_parity_other_grid_types(gpu, gtype, G, D):
    """The remaining values of geometric_network.voxel.type (python/network.py:120-151) through the fused geometric pass:
    loss, pixels and every gradient against the oracle."""
    conf = small_conf(grid_size=G, n_rays=8, overrides=[f"geometric_network.voxel.type={gtype}",
                                                        f"geometric_network.voxel.feature_size={D}"])
    prod = run_product_step(conf, B=1, R=8, device=gpu)
    s = prod["samples"]
    ref = run_oracle_step(conf, prod["params_cpu"], prod["inputs_cpu"], samples=(s["x_fg"], s["t_fg"], s["x_bg"], s["t_bg"], s["mask"]))
    assert abs(float(prod["loss"]) - float(ref["loss"])) <= LOSS_RTOL * abs(float(ref["loss"]))
    assert float((prod["color_pixel"].cpu() - ref["color_pixel"]).abs().max()) <= PIXEL_TOL
    ref64 = None
    for k, g in ref["grads"].items():
        gp = prod["grads"][k]
        assert (g is None) == (gp is None), k
        if g is None:
            continue
        e = rel_err(gp, g)
        if e >= GRAD_RTOL:
            if ref64 is None:
                ref64 = run_oracle_step(conf, prod["params_cpu"], prod["inputs_cpu"], dtype=torch.float64,
                                        samples=(s["x_fg"], s["t_fg"], s["x_bg"], s["t_bg"], s["mask"]))
            e64, o64 = rel_err(gp, ref64["grads"][k]), rel_err(g, ref64["grads"][k])
            assert e64 <= 3 * o64, (k, e, e64, o64)
