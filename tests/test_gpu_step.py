"""The training step object (ndjir_amd/step.py = python/train.py:75-78, 124-148) on the GPU: sparse re-arming of the grid
gradient buffer across redraws of the random tensors, the cos-anneal-ratio registry parameter, the object-mask loss
path and the NaN-loss veto of the optimizer step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _step(gpu, overrides=(), R=32, G=32, config="default"):
    from ndjir_amd import config as cfg
    from ndjir_amd.step import Step
    conf = cfg.load(config, [f"geometric_network.voxel.grid_size={G}"] + list(overrides))
    return Step(conf, R, gpu, 0, 1)


@pytest.mark.parametrize("config,G", [("default", 32), ("custom", 32), ("triplaneline", 64)])      # (the tri-plane buffer is re-armed by a memset: sparse zeroing measured no faster, round 6)
def test_rearm_after_redraw_clears_the_previous_steps_cells(gpu, config, G):
    """forward_backward, redraw_rand, forward_backward: the second step's grid gradient equals a run that zeroes the
    whole buffer (the perturbed points of step 1 were formed with the noise of step 1, not the redrawn one).  Dense voxel
    (linear; Lanczos: 4 x 4 x 4 taps); the tri-plane + tri-line buffers of config/triplaneline.yaml are re-armed whole."""
    step = _step(gpu, G=G, config=config)
    gen = torch.Generator(device=gpu)
    gen.manual_seed(5)
    step.forward_backward()
    step.redraw_rand(gen)
    step.forward_backward()
    sparse = {k: v.clone() for k, v in step.grid_bufs.items()}
    # dense reference: same inputs (the rand tensors now hold the redrawn values), buffer fully zeroed first
    for v in step.grid_bufs.values():
        v.zero_()
    step.compute(rearm=False)
    for k, v in step.grid_bufs.items():
        assert float(v.abs().max()) > 0
        assert float((v - sparse[k]).abs().max()) <= 1e-6 * float(v.abs().max()), k
    # and a third step after another redraw leaves no stale rows either
    step.redraw_rand(gen)
    step.forward_backward()
    third = {k: v.clone() for k, v in step.grid_bufs.items()}
    for v in step.grid_bufs.values():
        v.zero_()
    step.compute(rearm=False)
    for k, v in step.grid_bufs.items():
        assert float((v - third[k]).abs().max()) <= 1e-6 * float(v.abs().max()), k


def test_step_reads_the_cos_anneal_ratio_parameter(gpu):
    """python/train.py:75-78: the loss graph reads the registry parameter "cos_anneal_ratio", which
    Solvers.update_cos_anneal_ratio rewrites in place (python/solver.py:100-108)."""
    from ndjir_amd import parameter as P
    from ndjir_amd.solver import Solvers
    step = _step(gpu)
    car = P.get_parameters()["cos_anneal_ratio"]
    assert car.data_ptr() == step.car.data_ptr() and not car.requires_grad
    l1 = float(step.forward_backward())
    s = Solvers(step.conf)
    s.update_cos_anneal_ratio(100)          # 0.5 cos(pi 100 / 225) + 0.5
    want = 0.5 * np.cos(np.pi * 100 / (1500 * 0.15)) + 0.5
    assert float(step.car) == pytest.approx(want, rel=1e-6)
    l2 = float(step.forward_backward())
    assert l1 != l2
    # same value through the explicit-argument API
    from ndjir_amd.loss import total_loss
    out = total_loss(step.camloc, step.raydir, step.color_gt, None, torch.tensor([want], device=gpu, dtype=torch.float32),
                     step.conf, step.rand)
    assert float(out["loss"]) == pytest.approx(l2, rel=1e-6)


def test_object_mask_path(gpu):
    """train.mask_weight > 0 (python/loss.py:59-66, 107-115): Step feeds the rays' object mask; parity with the oracle."""
    from oracle import graph as G
    step = _step(gpu, ["train.mask_weight=0.5"], R=16)
    rng = np.random.RandomState(3)
    om = torch.from_numpy((rng.rand(1, 16, 1) > 0.4).astype(np.float32)).to(gpu)
    with pytest.raises(ValueError):
        step.set_rays(step.camloc.clone(), step.raydir.clone(), step.color_gt.clone())
    step.set_rays(step.camloc.clone(), step.raydir.clone(), step.color_gt.clone(), om)
    loss = float(step.forward_backward())
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in step.P.get_parameters().items()}
    rand = {k: v.cpu() for k, v in step.rand.items()}
    ref = G.total_loss(step.camloc.cpu(), step.raydir.cpu(), step.color_gt.cpu(), om.cpu(), step.car.detach().cpu(), rand,
                       params, step.conf)
    assert float(ref["loss_mask"]) > 0
    assert loss == pytest.approx(float(ref["loss"]), rel=2e-4)


@pytest.mark.parametrize("bad", [False, True])
def test_nan_loss_vetoes_the_update(gpu, bad):
    """python/train.py:144-146: a NaN loss skips the update even when at most one solver's gradients are non-finite."""
    from ndjir_amd import lib
    fa = torch.zeros(1, dtype=torch.int32, device=gpu)
    fb = torch.zeros(1, dtype=torch.int32, device=gpu)
    loss = torch.tensor([float("nan") if bad else 1.5], device=gpu)
    lib.call("solver_veto_if_nan", 1, loss, fa, fb)
    assert int(fa) == int(fb) == (1 if bad else 0)
    lib.call("solver_veto_if_nan", 1, torch.tensor([float("inf")], device=gpu), fa, fb)      # inf is not NaN (np.isnan)
    assert int(fa) == int(fb) == (1 if bad else 0)
    # end to end through Solvers.guarded_update
    step = _step(gpu, R=16, G=16)
    step.enable_training()
    step.train_step()
    w = step.mlp_params[0].detach().clone()
    t0 = step.solvers.solver_weight.step_count()
    step.train_compute.__func__     # (exists)
    s = step.solvers
    s.zero_grad(); s.weight_decay(); s.clip_grad_by_norm()
    step.compute(rearm=False)
    if bad:
        step.loss.fill_(float("nan"))
    step.optimizer_step()
    moved = float((step.mlp_params[0].detach() - w).abs().max()) > 0
    assert moved == (not bad)
    assert step.solvers.solver_weight.step_count() == t0 + (0 if bad else 1)
    assert step.solvers.solver_weight.skipped() == bad


def test_tracked_packed_weights(gpu):
    """Training mode of the packed-weight store (ndjir_amd/mlp.py `track_weights`): persistent packed copies, one
    re-pack launch after the optimizer's update (ndjir_mlp_pack_table), a weight changed by someone else re-packed at its
    next use, and a column slice of a wider matrix packed in place (the sdf column of the geometric net's last layer)."""
    from ndjir_amd import mlp
    rng = np.random.RandomState(2)
    Ws = [torch.tensor(rng.randn(*s) * 0.1, dtype=torch.float32, device=gpu) for s in ((39, 128), (128, 128), (128, 257))]
    bs = [torch.zeros(s, device=gpu) for s in (128, 128, 257)]
    x = torch.tensor(rng.randn(256, 39), dtype=torch.float32, device=gpu)
    try:
        mlp.track_weights(False)
        want0 = mlp.fused_mlp(x, Ws, bs).clone()
        mlp.track_weights(True)
        assert torch.equal(mlp.fused_mlp(x, Ws, bs), want0)                 # first use packs each weight on its own
        n_entries = len(mlp._TRACK)
        assert n_entries == 3
        # (a) an in-place change that autograd sees: re-packed at the next use
        Ws[1].mul_(1.5)
        got = mlp.fused_mlp(x, Ws, bs).clone()
        mlp.track_weights(False)
        assert torch.equal(got, mlp.fused_mlp(x, Ws, bs))
        # (b) a change behind autograd's back (what the optimizer kernels and a replayed graph do) + the one-launch re-pack
        mlp.track_weights(True)
        mlp.fused_mlp(x, Ws, bs)
        for W in Ws:
            W.data.mul_(0.5)                                              # no version bump
        stale = mlp.fused_mlp(x, Ws, bs).clone()
        mlp.repack_tracked()
        fresh = mlp.fused_mlp(x, Ws, bs).clone()
        mlp.track_weights(False)
        mlp._PACK_CACHE.clear()                # (the per-version cache cannot see a change that moved no version either)
        want = mlp.fused_mlp(x, Ws, bs)
        assert torch.equal(fresh, want) and not torch.equal(stale, want)
        # (c) a column slice packs like its contiguous copy, both orientations
        mlp.track_weights(True)
        col = Ws[2][:, 0:1]
        assert not col.is_contiguous()
        for tr in (False, True):
            a = mlp._packed(col, tr).clone()
            mlp.repack_tracked()
            b = mlp._packed(col, tr).clone()
            mlp.track_weights(False)
            c = mlp._packed(col.contiguous(), tr)
            assert torch.equal(a, c) and torch.equal(b, c)
            mlp.track_weights(True)
    finally:
        mlp.track_weights(False)


def test_training_graph_replay_then_eager(gpu):
    """A captured training iteration (optimizer + the one-launch re-pack inside the graph) replayed, then the same step
    issued eagerly, then replayed again: the eager step must run on the weights the replays left behind -- the packed
    copies are rewritten by the graph itself, not keyed on a version counter that replays do not move."""
    from ndjir_amd import mlp
    step = _step(gpu, R=16, G=16)
    try:
        step.enable_training()
        for _ in range(2):
            step.train_step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step.train_compute()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        live = [(k, e, e[2]()) for k, e in mlp._ROWS_CACHE.items() if e[2]() is not None]
        assert live                               # (the row-block copies of first-layer weights are refreshed by the graph too)
        for (_, _, a, b), e, W in live:
            assert torch.equal(e[1], torch.cat([W.detach()[:a], W.detach()[b:]]))
        loss_eager = float(step.compute(rearm=True))                       # eager forward on the replays' weights
        # reference: the same forward with every packed copy rebuilt from the current weights (the tracked buffers stay
        # alive meanwhile: the graph holds their addresses)
        with mlp.tracking_suspended():
            mlp._PACK_CACHE.clear()
            loss_fresh = float(step.compute(rearm=True))
        assert loss_eager == loss_fresh
        g.replay()
        torch.cuda.synchronize()
        assert np.isfinite(float(step.loss))
    finally:
        mlp.track_weights(False)


@pytest.mark.parametrize("variant,B", [("default", 1), ("no_voxel", 1), ("default", 2)])
def test_in_place_gradient_bucket_equals_autograd(gpu, variant, B):
    """Step's flat MLP gradient bucket, filled in place by the operators (ndjir_amd.mlp.set_grad_buffer: weights, biases, the
    row blocks of the split first-layer weights), against the same step with every gradient returned through autograd."""
    from ndjir_amd import config as cfg, mlp
    from ndjir_amd.step import Step
    conf = cfg.load(variant, ["geometric_network.voxel.grid_size=16"])
    step = Step(conf, 16, gpu, 0, 1, B=B)          # (B = 2: two images per step, python/train.py:38-51 -- bench.py's `b4` leg runs B = 4)
    assert step.raydir.shape[:2] == (B, 16) and step.rand["noise"].shape[:2] == (B, 16)
    # the views are registered only while `compute` runs (mlp.grad_buffers): outside a step the registry is empty, so another
    # backward pass over the same parameters gets ordinary autograd gradients and cannot add into the bucket
    assert step.in_place and len(step._grad_pairs) == len(step.mlp_params) and len(mlp._GRAD_BUF) == 0
    for _ in range(2):                       # twice: the bucket is re-zeroed, not accumulated across steps
        step.forward_backward()
    got = step.flat_grad.clone()
    assert len(mlp._GRAD_BUF) == 0
    pairs, step._grad_pairs = step._grad_pairs, []      # the same step with every gradient returned through autograd
    step.forward_backward()
    want = step.flat_grad.clone()
    step._grad_pairs = pairs
    assert float(want.abs().max()) > 0
    for name, v in zip(step.mlp_names, step.grad_views):
        off = v.storage_offset()
        a, b = got[off:off + v.numel()], want[off:off + v.numel()]
        err = float((a - b).norm() / max(float(b.norm()), 1e-30))
        assert err < 2e-5, (name, err)
    # a backward pass of somebody else over the step's parameters, after the step: real gradients, bucket untouched
    before = step.flat_grad.clone()
    W = step.mlp_params[[i for i, p in enumerate(step.mlp_params) if p.dim() == 2][0]]
    x = torch.randn(64, W.shape[0], device=gpu)
    g, = torch.autograd.grad(mlp.fused_mlp(x, [W], [None]).square().sum(), [W])
    assert g is not None and float(g.abs().max()) > 0 and torch.equal(step.flat_grad, before)


def test_training_iteration_without_a_feature_grid(gpu):
    """`no_voxel` (BASELINE config 4) has no grid parameters: the feature solver's group is empty, its guard finds nothing
    (python/solver.py:67-69 over an empty set) and the weight solver still updates -- `bench.py --config no_voxel` crashed
    in its training leg until round 5."""
    from ndjir_amd import config as cfg
    from ndjir_amd.step import Step
    conf = cfg.load("no_voxel", [])
    step = Step(conf, 16, gpu, 0, 1)
    step.enable_training()
    assert not step.solvers.solver_feat.params
    w = step.mlp_params[0].detach().clone()
    for _ in range(2):
        step.train_step()
    assert bool(torch.isfinite(step.loss).all())
    assert float((step.mlp_params[0].detach() - w).abs().max()) > 0
    t0 = step.solvers.solver_weight.step_count()
    s = step.solvers
    s.zero_grad(); s.weight_decay(); s.clip_grad_by_norm()
    step.compute(rearm=False)
    step.loss.fill_(float("nan"))            # a NaN loss still vetoes the update
    step.optimizer_step()
    assert step.solvers.solver_weight.step_count() == t0
