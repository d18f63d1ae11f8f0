"""The N > 1 step on the GPU: two ranks (sharing GPU 0 over gloo) run the ray-sharded step with the HIP sparse grid
exchange; the summed loss, the all-reduced MLP gradient bucket and the exchanged grid gradient must equal the
single-process step over the union of the rays -- for two consecutive steps (the second one exercises the re-arming of
the rows received from the other rank)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode,variant", [("eager", "default"), ("graph", "default"), ("eager", "custom"), ("eager", "triplaneline"),
                                          ("graph", "triplaneline"), ("eager", "no_voxel"), ("graph", "no_voxel")])
def test_two_ranks_equal_one_process(gpu, mode, variant):
    """variant: default (linear voxel, D = 4), custom (Lanczos voxel: 4 x 4 x 4 taps), triplaneline (tri-plane + tri-line,
    D = 8): every grid gradient goes through the sparse row exchange -- no all-reduce larger than the MLP bucket;
    no_voxel (BASELINE.json config 4: config/no_voxel.yaml, ray-sharded): no grid at all, the exchange is the bucket alone."""
    import bench
    from ndjir_amd import config as cfg, parameter as P
    from ndjir_amd.grid_feature import set_grad_buffer
    R, G, steps = 64, 64, 2
    with tempfile.TemporaryDirectory() as out:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", NDJIR_TEST_VARIANT=variant)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29533", os.path.join(ROOT, "tests", "multi_rank_worker.py"), out, str(R), str(G), str(steps)] + \
            (["graph"] if mode == "graph" else [])
        res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
        assert res.returncode == 0, res.stderr[-3000:]
        ranks = [torch.load(os.path.join(out, f"rank{r}.pt")) for r in range(2)]
    want = {"geometric-network/triplane_feature/F": "SparseRows", "geometric-network/triline_feature/F": "SparseRows"} \
        if variant == "triplaneline" else {} if variant == "no_voxel" else {"geometric-network/voxel_feature/F": "SparseRows"}
    assert ranks[0]["handle"] == want                                                      # the HIP path ran
    n_mlp = int(ranks[0]["flat"].numel())
    # (the bucket carries 4 trailing floats: the next step's mask counts ride with it)
    assert max(ranks[0]["reduced"]) <= n_mlp + 4, ("a dense grid all-reduce was issued", max(ranks[0]["reduced"]), n_mlp)
    # collectives per step once the rays are in place before the exchange: the bucket's all-reduce alone (+ two all-gathers
    # per sparsely exchanged grid on the second communicator) -- the mask counts need none of their own after the first step
    # (two scalar-sized all-reduces in total: the parameter-creating pass of Step.__init__ and the first step)
    assert ranks[0]["reduced"].count(4) <= 2, ranks[0]["reduced"]
    for k, lim in ranks[0]["limits"].items():
        assert int(ranks[0]["counts"][k].max()) <= lim, k                                  # nothing was cut off the wire
    conf = cfg.load(variant, [f"geometric_network.voxel.grid_size={G}"])
    step = bench.Step(conf, 2 * R, gpu, 0, 1)
    try:
        for _ in range(steps):
            loss = float(step.forward_backward())
        assert ranks[0]["losses"][-1] + ranks[1]["losses"][-1] == pytest.approx(loss, rel=5e-6)
        flat = torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for p, g in zip(step.mlp_params, step.grads)]).cpu()
        for r in range(2):      # both ranks hold the same sums
            d = (ranks[r]["flat"] - flat).abs().max()
            assert float(d) <= 2e-4 * float(flat.abs().max()), r
            for k, v in step.grid_bufs.items():
                ref = v.cpu()
                assert float((ranks[r]["grid"][k] - ref).abs().max()) <= 2e-4 * float(ref.abs().max()), (r, k)
                assert int(((ranks[r]["grid"][k] != 0).any(-1) != (ref != 0).any(-1)).sum()) <= 0.001 * int((ref != 0).any(-1).sum()) + 2
    finally:
        for p in step.grid_params:
            set_grad_buffer(p, None)
        P.clear_parameters()


@pytest.mark.timeout(1800)          # a cold box pages torch and librccl in once per process: minutes, not seconds
def test_bench_graph_replay_around_rccl_calls(gpu):
    """bench.py's N > 1 execution scheme against the real collective library: a 1-rank RCCL group forces the distributed
    code path (mask all-reduce, gradient bucket, sparse grid exchange) with the compute part replayed from a HIP graph.
    The JSON line must be the last line on stdout (RCCL's banner goes through C stdio) and report graph execution."""
    import json
    env = dict(os.environ, NDJIR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rays", "64", "--override", "geometric_network.voxel.grid_size=64",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--train-steps", "2"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads(res.stdout.strip().splitlines()[-1])
    assert d["execution"].startswith("one captured HIP graph"), (d["execution"], d.get("graph_capture_error"))
    assert d["n_gpus"] == 1 and np.isfinite(d["loss"]) and d["value"] > 0
    assert np.isfinite(d["train_step"]["loss_after"])
    # the same step without the process group gives the same loss
    res2 = subprocess.run(cmd, env={k: v for k, v in env.items() if k != "NDJIR_BENCH_FORCE_DIST"}, cwd=ROOT, capture_output=True,
                          text=True, timeout=800)
    assert res2.returncode == 0, res2.stderr[-3000:]
    d2 = json.loads(res2.stdout.strip().splitlines()[-1])
    assert d["loss"] == pytest.approx(d2["loss"], rel=1e-6)


def _launch(world, mode, out, port, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "multi_rank_worker2.py"), mode, out]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("world", [4, 8])
def test_sparse_exchange_overflow_and_limit_growth(gpu, world):
    """ndjir_amd/distributed.py `exchange_grid_rows_hip` at world 4 and 8 (3 / 7 remote lists per rank): equal to a dense
    all-reduce; a wire size cut below the lists raises the device flag on every rank, counts the exchange in the device
    statistics and delivers an incomplete sum WITHOUT touching the wire size; at the next look (every CHECK_EVERY exchanges,
    from the running maximum the device kept) the new wire size is decided -- it takes effect with the exchange AFTER the
    look (the counts travel in the id lists' headers since round 5: no collective of their own, so the look's own wire was
    already sized) -- and the sums are whole again."""
    with tempfile.TemporaryDirectory() as out:
        _launch(world, "exchange", out, 29551 + world)
        recs = [torch.load(os.path.join(out, f"rank{r}.pt")) for r in range(world)]
    for r, rec in enumerate(recs):
        assert rec["world"] == world and rec["first_ok"] and rec["first_overflow"] == 0 and rec["zero_ok"], (r, rec)
        assert rec["first_limit"] >= max(x["n_own"] for x in recs), (r, rec)          # every rank sized the wire for the largest list
        assert rec["cut_overflow_flag"] == 1 and rec["cut_incomplete"] and rec["cut_limit_unchanged"], (r, rec)
        assert rec["cut_stats"][0] >= max(x["n_own"] for x in recs) and rec["cut_stats"][1] >= 1, (r, rec)
        assert rec["cut3_limit"] == 4096 and rec["look_limit"] == 4096, (r, rec)   # no look between the checks; the look's own wire
        assert rec["grown_limit"] >= max(x["n_own"] for x in recs) and rec["grown_ok"], (r, rec)
        assert rec["overflowed_exchanges"] == 3 and rec["after_ok"], (r, rec)          # exchanges 2, 3 and 4 (the look) overflowed, 5 did not
    assert len({rec["grown_limit"] for rec in recs}) == 1                              # ... and they agree on the new wire size


@pytest.mark.timeout(1200)
def test_sparse_exchange_query_points_outgrow_the_list_capacity(gpu):
    """the query count grows between exchanges beyond the capacity the first exchange sized the lists
    for (world 2).  `k_pack_rows` then drops cells -- it must not leave their bitmap bits set (they would never be listed
    again), every such exchange must raise the overflow flag (the optimizer step is vetoed), the re-arm must clear the whole
    buffer (the dropped cells hold gradient no list names), and after the look at which every rank re-creates its state with
    the announced capacity the sums equal the dense all-reduce again."""
    world = 2
    with tempfile.TemporaryDirectory() as out:
        _launch(world, "grow", out, 29581)
        recs = [torch.load(os.path.join(out, f"rank{r}.pt")) for r in range(world)]
    for r, rec in enumerate(recs):
        assert rec["ok1"] and rec["zero1"], (r, rec)
        assert rec["n2"] > rec["cap1"], (r, rec)                                     # the second point set really does not fit
        assert rec["drop_flags"] == [1, 1] and rec["drop_zero"] == [True, True], (r, rec)
        assert all(c > rec["cap1"] for c in rec["drop_counts"]), (r, rec)            # the SAME count twice: no cell went missing
        assert rec["bitmap_clean"] and rec["look_flag"] == 1, (r, rec)
        assert rec["cap2"] >= rec["n2"] and rec["generation_moved"], (r, rec)
        assert rec["ok5"] and rec["flag5"] == 0 and rec["zero5"] and rec["ok6"], (r, rec)


@pytest.mark.timeout(1200)
def test_overflowing_exchange_vetoes_the_optimizer_step(gpu):
    """Step.optimizer_step under an overflowing sparse exchange (4 ranks): the update is skipped on the device (solver step
    counter and weights unchanged, python/train.py:141-146's skip), counted once per step, and reported."""
    world = 4
    with tempfile.TemporaryDirectory() as out:
        _launch(world, "veto", out, 29571)
        recs = [torch.load(os.path.join(out, f"rank{r}.pt")) for r in range(world)]
    for r, rec in enumerate(recs):
        assert np.isfinite(rec["l0"]) and np.isfinite(rec["l1"]), (r, rec)
        assert rec["steps_after"] == rec["steps_before"] and rec["weights_unchanged"], (r, rec)
        assert rec["report"]["vetoed_optimizer_steps"] == 1, (r, rec)
        b = rec["report"]["buffers"]["geometric-network/voxel_feature/F"]
        assert b["overflowed_exchanges"] == 1 and b["largest_list"] > 4096 and b["wire_rows"] == 4096, (r, rec)


@pytest.mark.timeout(900)
def test_render_image_on_two_ranks_equals_one_process(gpu):
    """renderer.render_image(rank, world) (BASELINE.json config 5's tiling, python/renderer.py:212-272): two processes render
    alternate tiles and sum their partial frames over torch.distributed -- the frame of the single-process render, bit for
    bit (a pixel is written by exactly one rank); each partial frame holds its rank's tiles only."""
    from ndjir_amd import network, parameter as P
    from ndjir_amd.renderer import render_image
    from tests.parity_utils import small_conf
    with tempfile.TemporaryDirectory() as out:
        _launch(2, "render", out, 29581)
        parts = [np.load(os.path.join(out, f"rank{r}.npz")) for r in range(2)]
        parts = [dict(img=p["img"], part=p["part"]) for p in parts]
    conf = small_conf(grid_size=16, n_rays=16, overrides=["valid.n_rays=48", "valid.n_down_samples=0"])
    P.clear_parameters()
    P.set_device(gpu)
    network.seed(313)
    pose = np.eye(4, dtype=np.float64)[None]
    pose[0, :3, 3] = [0.0, 0.0, -2.5]
    K = np.array([[[20.0, 0, 8], [0, 20.0, 6], [0, 0, 1]]])
    img = render_image(pose, K, (16, 12), conf, device=gpu)
    np.testing.assert_array_equal(parts[0]["img"], img)
    np.testing.assert_array_equal(parts[1]["img"], img)
    np.testing.assert_array_equal(parts[0]["part"] + parts[1]["part"], img)
    flat = [p["part"].reshape(3, -1).sum(0) for p in parts]
    assert not np.any((flat[0] != 0) & (flat[1] != 0))                    # disjoint tiles
    P.clear_parameters()


def _bench_json(args, env_extra, timeout=800):
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True,
                         timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


@pytest.mark.timeout(900)
def test_bench_starts_its_own_ranks(gpu):
    """`python bench.py --gpus 2` with no launcher around it (the way a user -- and a driver without torchrun -- calls it):
    bench.py starts the two ranks itself, relays the one JSON line and the exit code.  Both ranks share GPU 0 over gloo
    (NDJIR_BENCH_SAME_DEVICE: the only two-rank set-up a one-GPU box offers)."""
    out = _bench_json(["--gpus", "2", "--rays", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs",
                       "--train-steps", "0", "--override", "geometric_network.voxel.grid_size=64"], {"NDJIR_BENCH_SAME_DEVICE": "1"})
    assert out["n_gpus"] == 2 and out["ranks"]["world_size"] == 2 and len(out["ranks"]["devices"]) == 2
    assert out["scaling"] == "weak" and out["config"]["rays_per_gpu"] == 64
    assert "exchange" in out and out["value"] > 0


@pytest.mark.timeout(900)
def test_bench_strong_scaling_config4_two_ranks(gpu):
    """BASELINE.json config 4 (config/no_voxel.yaml, rays split over the ranks) through the same self-launch."""
    out = _bench_json(["--gpus", "2", "--scaling", "strong", "--config", "no_voxel", "--total-rays", "128", "--steps", "2", "--warmup", "1",
                       "--no-cpu-baseline", "--no-extra-legs", "--train-steps", "0"], {"NDJIR_BENCH_SAME_DEVICE": "1"})
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["rays_per_gpu"] == 64
    assert out["ranks"]["world_size"] == 2 and "exchange" in out


@pytest.mark.timeout(1800)
def test_bench_two_ranks_on_two_devices_over_rccl(gpu):
    """The day a multi-GPU node is there (task 8): `bench.py --gpus 2` with one rank per DEVICE over RCCL --
    not the shared-GPU gloo harness every other two-rank test has to use.  Skipped on a one-GPU box (the pool's gpurun boxes);
    `torch.cuda.device_count()` does not initialise the GPU, so the skip costs nothing."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL with N > 1 cannot run on this box")
    out = _bench_json(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs", "--train-steps", "2"], {},
                      timeout=1500)
    assert out["n_gpus"] == 2 and out["ranks"]["world_size"] == 2 and out["ranks"]["backend"] == "nccl"
    pci = [d["pci"] for d in out["ranks"]["devices"]]
    assert len(set(pci)) == 2 and None not in pci, pci                     # two ranks on two distinct devices
    assert "exchange" in out and "warning" not in out["exchange"], out.get("exchange")
    assert out["execution"].startswith("one captured HIP graph"), (out["execution"], out.get("graph_capture_error"))
    assert np.isfinite(out["loss"]) and out["value"] > 0 and np.isfinite(out["train_step"]["loss_after"])
