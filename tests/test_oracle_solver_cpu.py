"""CPU checks of the optimizer-step oracle (oracle/solver.py) and of the product's host-side schedule logic
(ndjir_amd/solver.py).  The reference holds no solver test (parity unpinned); the Adam recurrences and the bias
correction are pinned independently against torch.optim.Adam in the regime where the two rules coincide (eps = 0:
nnabla keeps eps outside the bias correction, torch inside)."""
import os

import numpy as np
import pytest
import torch

from ndjir_amd import config
from ndjir_amd.solver import Solvers
from oracle import solver as OS


def test_adam_matches_torch_when_eps_is_zero():
    rng = np.random.RandomState(412)
    w0 = rng.randn(257).astype(np.float64)
    grads = [rng.randn(257) for _ in range(12)]
    o = OS.Adam(alpha=2.0 ** -8, eps=0.0)
    w = w0.copy()
    o.set_parameters({"w": w})
    tw = torch.tensor(w0.copy(), requires_grad=True)
    opt = torch.optim.Adam([tw], lr=2.0 ** -8, betas=(float(np.float32(0.9)), float(np.float32(0.999))), eps=0.0)
    for g in grads:
        o.zero_grad()
        o.grads["w"] += g
        o.update()
        tw.grad = torch.tensor(g)
        opt.step()
    np.testing.assert_allclose(w, tw.detach().numpy(), rtol=1e-12, atol=1e-14)


def test_adam_first_step_is_sign_step():
    """t = 1: m = (1-b1) g, v = (1-b2) g^2, alpha_t = alpha sqrt(1-b2)/(1-b1)  =>  dw = -alpha g / (|g| + eps')."""
    o = OS.Adam(alpha=1e-2)
    w = np.zeros(4, np.float64)
    o.set_parameters({"w": w})
    o.grads["w"] += np.array([1.0, -2.0, 0.5, 1e-3])
    o.update()
    np.testing.assert_allclose(w, -1e-2 * np.sign([1.0, -2.0, 0.5, 1e-3]), rtol=1e-3)


def test_step_order_decay_is_clipped_alone():
    """python/train.py:136-148: zero_grad, weight_decay, clip_grad_by_norm, backward accumulates, update."""
    tr = dict(batch_size=1, n_rays=512, base_learning_rate_weight=1e-3, base_learning_rate_feat=1e-3, weight_decay=0.5,
              clip_grad_norm=0.1, epoch=10, warmup_term_ratio=0.0, learning_rate_end_ratio=0.01)
    s = OS.Solvers(tr)
    w = np.full(4, 2.0)
    s.set_parameters({"net/affine/W": w})
    s.update_learning_rate(0)
    loss_grad = np.array([100.0, 0.0, 0.0, 0.0])
    assert s.step({"net/affine/W": loss_grad})
    g = s.solver_weight.grads["net/affine/W"]
    # decay gradient 0.5 * 2 = 1 per entry, norm 2 -> scaled to norm 0.1; the loss gradient is NOT clipped
    np.testing.assert_allclose(g, np.array([100.05, 0.05, 0.05, 0.05]), rtol=1e-12)


def test_guard_uses_and():
    """python/solver.py:67-69: the update is skipped only when BOTH solvers see an inf / nan."""
    tr = dict(batch_size=1, n_rays=512, base_learning_rate_weight=1e-3, base_learning_rate_feat=1e-3, weight_decay=0.0,
              clip_grad_norm=0, epoch=10, warmup_term_ratio=0.0, learning_rate_end_ratio=0.01)
    s = OS.Solvers(tr)
    s.set_parameters({"a/W": np.ones(3), "g/voxel_feature/F": np.ones(3)})
    s.update_learning_rate(0)
    assert s.step({"a/W": np.array([np.nan, 0, 0]), "g/voxel_feature/F": np.zeros(3)})           # one bad: still updates
    assert not s.step({"a/W": np.array([np.inf, 0, 0]), "g/voxel_feature/F": np.array([np.nan, 0, 0])})
    assert s.solver_feat.t == 1 and s.solver_weight.t == 1


@pytest.mark.parametrize("B,R", [(1, 512), (4, 512)])
def test_schedules_product_vs_oracle(B, R):
    conf = config.load("default", [f"train.batch_size={B}", f"train.n_rays={R}"])
    s = Solvers(conf)
    tr = dict(conf.train)
    o = OS.Solvers(tr)
    assert s.learning_rate_weight == o.learning_rate_weight == 0.0005 * B * R / 512
    warm = int(conf.train.epoch * conf.train.warmup_term_ratio)
    assert warm == 22
    for i in [0, 1, warm - 1, warm, warm + 1, 700, conf.train.epoch - 1, conf.train.epoch]:
        a = s.compute_learning_rate(i, s.learning_rate_feat)
        b = OS.compute_learning_rate(tr, i, o.learning_rate_feat)
        assert a == pytest.approx(b, rel=1e-14, abs=0)
    lr = s.learning_rate_weight
    assert s.compute_learning_rate(0, lr) == 0.0
    assert s.compute_learning_rate(warm, lr) == pytest.approx(lr, rel=2e-3)      # top of the cosine
    assert s.compute_learning_rate(conf.train.epoch, lr) == pytest.approx(0.01 * lr, rel=1e-6)
    assert OS.cos_anneal_ratio(tr, 0) == 1.0 and OS.cos_anneal_ratio(tr, 224) == pytest.approx(0.0, abs=1e-4)
    assert OS.cos_anneal_ratio(tr, 225) == 1.0          # sic: jumps back to 1 once the anneal term is over (x >= 1)
    assert OS.light_visibility_gain(tr, 0) == 1.0 and OS.light_visibility_gain(tr, 1500) == 1.0


def test_learning_rate_schedule_against_reference_golden():
    """tests/golden/solver_schedule.npz: outputs of the reference's own Solvers.compute_learning_rate
    (python/solver.py:82-98, extracted by tests/golden/make_golden.py) -- pins the oracle and the product's schedule."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "solver_schedule.npz"))
    for k in range(int(g["n_cases"])):
        tr = dict(epoch=int(g[f"c{k}_epoch"]), warmup_term_ratio=float(g[f"c{k}_warmup_term_ratio"]),
                  learning_rate_end_ratio=float(g[f"c{k}_learning_rate_end_ratio"]))
        conf = config.load("default", [f"train.{a}={b}" for a, b in tr.items()])
        s = Solvers(conf)
        lr0 = float(g[f"c{k}_lr0"])
        for i, want in zip(g[f"c{k}_i"], g[f"c{k}_lr"]):
            assert OS.compute_learning_rate(tr, int(i), lr0) == pytest.approx(want, rel=1e-13, abs=1e-300)
            assert s.compute_learning_rate(int(i), lr0) == pytest.approx(want, rel=1e-13, abs=1e-300)


def test_host_ray_generation_against_reference_golden():
    """tests/golden/generate_raydir_camloc.npz: outputs of the reference's python/helper.py:44-81."""
    from ndjir_amd.helper import generate_all_pixels, generate_raydir_camloc
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "generate_raydir_camloc.npz"))
    raydir, camloc = generate_raydir_camloc(g["pose"], g["intrinsic"], g["xy"])
    np.testing.assert_allclose(raydir, g["raydir"], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(camloc, g["camloc"])
    np.testing.assert_array_equal(generate_all_pixels(int(g["W"]), int(g["H"])), g["all_pixels"])


@pytest.mark.parametrize("ext", [".h5", ".npz"])
def test_parameter_file_round_trip(tmp_path, ext):
    """save_parameters / load_parameters (python/train.py:101, python/render_image.py:43): names, values, need_grad."""
    from ndjir_amd import parameter as P
    P.clear_parameters()
    P.set_device("cpu")
    rng = np.random.RandomState(0)
    with P.parameter_scope("geometric-network"):
        with P.parameter_scope("affine-00"), P.parameter_scope("affine"):
            w = P.get_parameter_or_create("W", (5, 7), rng.randn(5, 7))
        g = P.get_parameter_or_create("gain", (1,), np.asarray([0.3]), True)
    fixed = P.get_parameter_or_create("cos_anneal_ratio", (1,), np.asarray([0.25]), False)
    path = str(tmp_path / ("model_00010" + ext))
    P.save_parameters(path)
    want = {k: v.detach().clone() for k, v in P.get_parameters().items()}
    need = {k: v.requires_grad for k, v in P.get_parameters().items()}
    with torch.no_grad():
        w.zero_()
    P.load_parameters(path)                       # in place: the same tensor objects come back filled
    assert P.get_parameters()["geometric-network/affine-00/affine/W"] is w and torch.equal(w, want["geometric-network/affine-00/affine/W"])
    P.clear_parameters()
    P.load_parameters(path, device="cpu")
    got = P.get_parameters()
    assert list(got) == list(want)
    for k in want:
        assert torch.equal(got[k], want[k]) and got[k].requires_grad == need[k], k
    assert not got["cos_anneal_ratio"].requires_grad
    with pytest.raises(ValueError):
        P.save_parameters(str(tmp_path / "model.protobuf"))
    P.clear_parameters()
    P.set_device(None)


@pytest.mark.parametrize("mode", ["uniform", "patch", "mask"])
def test_pixel_sampling_against_reference_golden(mode):
    """tests/golden/pixel_sampling.npz: what the reference's own IDRDataSource._get_data returns (python/dataset.py:33-108)
    for three positions in each sampling mode; ndjir_amd.dataset.IDRRaySource must pick the same pixels (same numpy RNG
    call sequence) and deliver the same colours and mask values."""
    from ndjir_amd.dataset import IDRRaySource
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pixel_sampling.npz"))
    M, W, R = int(g["M"]), int(g["W"]), int(g["R"])
    ov = {"uniform": [], "patch": ["train.patch_ray_sampling=true"], "mask": ["train.mask_ray_sample_ratio=0.25"]}[mode]
    conf = config.load("default", [f"train.n_rays={R}", "train.patch_ray_sampling=false", "train.mask_ray_sample_ratio=0"] + ov)
    rng = np.random.RandomState(7)
    K = rng.rand(M, 3, 3) + np.eye(3) * 3
    poses = np.tile(np.eye(4), (M, 1, 1))
    src = IDRRaySource(g["images"], g["masks"], K, poses, conf, rng=np.random.RandomState(313), device="cpu")
    for pos in range(M):
        img, idx = src.pixel_indices(pos)
        xy = g[f"{mode}_{pos}_xy"]
        assert img == pos
        np.testing.assert_array_equal(idx, xy[:, 1] * W + xy[:, 0])
    # the batch interface delivers the reference's colours / masks (fresh source: same RNG stream from the start)
    src = IDRRaySource(g["images"], g["masks"], K, poses, conf, rng=np.random.RandomState(313), device="cpu")
    color, mask, raydir, camloc = src.next_batch(M)
    for pos in range(M):
        np.testing.assert_array_equal(color[pos].numpy(), g[f"{mode}_{pos}_color"])
        np.testing.assert_array_equal(mask[pos].numpy(), g[f"{mode}_{pos}_mask"].astype(np.float32))
    assert raydir.shape == (M, R, 3) and camloc.shape == (M, 3)
    np.testing.assert_allclose(raydir.norm(dim=-1).numpy(), 1.0, atol=1e-6)
