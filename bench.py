"""bench.py -- rays/s (forward + backward of total_loss) at 512 rays x 128 samples per GPU.

  python bench.py --gpus 1 --steps K --warmup W            (single GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = sample_points -> pb_render -> total_loss -> backward to every parameter gradient
(MLP weights + the 512^3 x 4 feature grid), config/default.yaml, B=1, R=512, N=128 (+32 bg
samples, 128 lights), synthetic rays, reference-initialised weights.  Rays shard across ranks
(weak scaling: 512 rays per GPU); one gradient exchange per step over RCCL.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.

`roofline` is for the hand-written KERNEL (one symbol, as rocprofv3 prints it; the library reports which kernel a launch runs:
ndjir_mlp_chain_kernel) with the largest time per step: achieved = sum of the algorithmic FLOPs (2*in*out per affine per point,
SURVEY 8d) / bytes of ITS launches / sum of their durations, measured live with HIP events recorded on the launching stream
around every launch.  In the default f16x3 arithmetic one algorithmic FLOP costs three f16 MFMA FLOPs, so the matrix peak it is
priced against is the dense 16-bit MFMA peak / 3 (bf16x6: / 6).  `kernels_by_symbol` lists every kernel of the engine that way,
`kernels` the same launches by class (forward / backward / tangent chains, weight gradients), each with both yardsticks.
`b4` = the reference's real training-step shape, 4 images x 512 rays (config/default.yaml:126-127); `redraw` = replays with
new pixels (device data feed) and new random tensors before every replay, as a training loop has them.
`--scaling strong --total-rays 4096 --config no_voxel` is BASELINE.json's config 4 (rays split over the ranks).
Execution: the compute part of the step (ndjir_amd/step.py `Step.compute`: no collective inside) is captured once
into a HIP graph and the timed region replays it (`--exec graph`, default; no host-side launch work in the timed
region; N > 1: the scalar mask all-reduce and the gradient exchange are issued eagerly around every replay).  HIP events
cannot be recorded inside a captured graph, so `roofline` / `kernels` come from the same K steps issued eagerly right
after (`eager` reports their wall time).  `--exec eager` times ordinary stream launches.  If the capture fails on any
rank, or replaying is not faster than eager launches in a short untimed trial, the run falls back to eager.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# dense-GEMM FLOPs (2*in*out per affine per point), default config
MFLOP_PER_RAY_FWD_BWD = 2168.9
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
# bf16x6 engine: every algorithmic fp32 FLOP is executed as 6 bf16 MFMA FLOPs (three-way operand split);
# f16x3 engine: as 3 f16 MFMA FLOPs (scaled two-way split) -- f16 and bf16 MFMA run at the same rate
PEAK_BF16X6_EFFECTIVE_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
PEAK_F16X3_EFFECTIVE_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PROFILE_ROUND = "r06"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)      # 0.3 s of timed replays (10 until round 6: VERDICT round 5, weak #9)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=512, help="rays per GPU per step (B=1); weak scaling")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's contract): --rays per GPU whatever N; strong: --total-rays split over the "
                         "N ranks (BASELINE.json config 4: `--scaling strong --config no_voxel`, 4096 rays / N per GPU)")
    ap.add_argument("--total-rays", type=int, default=4096, help="rays per step over all GPUs with --scaling strong")
    ap.add_argument("--config", default="default")
    ap.add_argument("--override", action="append", default=[])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exec", dest="exec_mode", choices=["graph", "eager"], default=None,
                    help="how the timed steps are issued: 'graph' (default) = the compute part of the step is captured "
                         "once into a HIP graph and the timed region replays it (N > 1: the collectives are issued "
                         "eagerly around every replay); 'eager' = ordinary stream launches.  Falls back to "
                         "eager if the capture fails on any rank or replaying is not faster.")
    ap.add_argument("--cpu-rays", type=int, default=32)
    ap.add_argument("--no-extra-legs", dest="extra_legs", action="store_false",
                    help="skip the `redraw` (new pixels + new random tensors before every replay) and `b4` (4 images x R rays, the "
                         "reference's training-step shape) side measurements")
    ap.add_argument("--train-steps", type=int, default=10,
                    help="after the fwd+bwd measurement: time this many full training iterations (fwd+bwd + the two Adam "
                         "solvers with weight decay, python/train.py:136-148) and report them as `train_step`; 0 = skip")
    return ap.parse_args()


from ndjir_amd.step import Step  # noqa: E402,F401  (the step itself lives in the package; bench.py only times it)


def cpu_baseline(conf, step, n_rays):
    """The CPU oracle (restatement of the reference graph) timed on this box's host cores on a
    bounded sample: `n_rays` of rank 0's rays, same parameters, same random tensors."""
    from tests.parity_utils import run_oracle_step
    cores = min(32, os.cpu_count() or 1)     # small per-ray GEMMs do not scale past a few dozen threads
    torch.set_num_threads(cores)
    params_cpu = {k: v.detach().cpu() for k, v in step.P.get_parameters().items()}
    sl = slice(0, n_rays)
    inputs = dict(camloc=step.camloc.cpu(), raydir=step.raydir[:, sl].cpu().contiguous(),
                  color_gt=step.color_gt[:, sl].cpu().contiguous(),
                  rand={k: v[:, sl].cpu().contiguous() for k, v in step.rand.items()},
                  cos_anneal=step.car.cpu())
    warm = dict(inputs, raydir=inputs["raydir"][:, :2].contiguous(), color_gt=inputs["color_gt"][:, :2].contiguous(),
                rand={k: v[:, :2].contiguous() for k, v in inputs["rand"].items()})
    run_oracle_step(conf, params_cpu, warm)              # warm-up on 2 rays (page-in, thread pools)
    t0 = time.perf_counter()
    reps = 0
    while True:
        run_oracle_step(conf, params_cpu, inputs)
        reps += 1
        el = time.perf_counter() - t0
        if el > 15.0 or reps >= 5:
            break
    out = dict(value=n_rays * reps / el, unit="rays/s", cores=torch.get_num_threads(), kind="port",
               sample=f"{reps} x fwd+bwd of {n_rays} rays (B=1) of the same workload, oracle/graph.py torch-CPU fp32, "
                      f"{cores} of {os.cpu_count()} host cores, {el:.1f} s")
    # BASELINE.json config 1 (SURVEY 8d "CPU baseline": the reference's own CPU-runnable shape, 512 rays x 64 samples =
    # renderer.n_upsamples 0), same rays and parameters, bounded to a few seconds
    try:
        import copy
        conf1 = copy.deepcopy(conf)
        conf1.renderer.n_upsamples = 0
        n1 = conf1.renderer.n_samples0
        in1 = dict(inputs, rand=dict(inputs["rand"], noise=inputs["rand"]["noise"][:, :, :n1].contiguous()))
        t0 = time.perf_counter()
        reps1 = 0
        while True:
            run_oracle_step(conf1, params_cpu, in1)
            reps1 += 1
            el1 = time.perf_counter() - t0
            if el1 > 6.0 or reps1 >= 4:
                break
        out["cfg1"] = dict(value=n_rays * reps1 / el1, unit="rays/s",
                           sample=f"{reps1} x fwd+bwd of {n_rays} rays x {n1} samples (n_upsamples=0), {el1:.1f} s")
    except Exception as e:
        out["cfg1"] = dict(value=None, sample=f"failed: {type(e).__name__}: {e}")
    return out


def parity_block(conf, step, n_rays):
    """Product vs CPU oracle at the HEADLINE configuration (the bench's own parameters -- 512^3 x 4 grid for default.yaml --
    and the first `n_rays` of rank 0's rays): loss, pixel colours, every parameter gradient, and the sampler's indices.
    BASELINE.json north_star: "pixel RGB and loss within 1e-4 relative; sample indices bit-exact" (indices: each side
    evaluates its own SDF network, so an index can differ where a CDF value sits within fp32 round-off of the uniform
    draw -- the count is reported; the round kernel itself is bit-exact given equal inputs, tests/test_gpu_sampler.py).
    The product pass runs with 128-point tiles forced (the kernels the timed region runs; at n_rays x 128 points the
    dispatch would otherwise pick the small-launch kernels)."""
    from ndjir_amd import mlp
    from ndjir_amd.loss import total_loss
    from tests.parity_utils import rel_err, run_oracle_step
    sl = slice(0, n_rays)
    rand = {k: v[:, sl].contiguous() for k, v in step.rand.items()}
    raydir, color = step.raydir[:, sl].contiguous(), step.color_gt[:, sl].contiguous()
    for buf in step.grid_bufs.values():
        buf.zero_()
    old_tile = mlp.get_tile_rows()
    mlp.set_tile_rows(128)
    try:
        rec_p = {}
        out = total_loss(step.camloc, raydir, color, None, step.car, conf, rand, record=rec_p)
        # (the grid parameters too: their gradients arrive in the step's accumulate-in-place buffers, zeroed above)
        grads = torch.autograd.grad(out["loss"], step.mlp_params + step.grid_params, allow_unused=True)[:len(step.mlp_params)]
        torch.cuda.synchronize()
    finally:
        mlp.set_tile_rows(old_tile)
    params = step.P.get_parameters()
    params_cpu = {k: v.detach().cpu() for k, v in params.items()}
    inputs = dict(camloc=step.camloc.cpu(), raydir=raydir.cpu(), color_gt=color.cpu(),
                  rand={k: v.cpu() for k, v in rand.items()}, cos_anneal=step.car.cpu())
    rec_o = {}
    ref = run_oracle_step(conf, params_cpu, inputs, record=rec_o)
    l0, l1 = float(out["loss"]), float(ref["loss"])
    worst, worst_name = 0.0, None
    for name, g in zip(step.mlp_names, grads):
        go = ref["grads"].get(name)
        if g is None or go is None:
            continue
        e = rel_err(g, go)
        if e > worst:
            worst, worst_name = e, name
    grid = {}
    for name, buf in step.grid_bufs.items():
        go = ref["grads"].get(name)
        if go is not None:
            grid[name] = rel_err(buf, go)
        buf.zero_()
    tot = mis = 0
    for a, b in zip(rec_p.get("idx", []), rec_o.get("idx", [])):
        tot += b.numel()
        mis += int((a.cpu() != b).sum())
    return {"n_rays": n_rays, "loss_rel": abs(l0 - l1) / max(abs(l1), 1e-30),
            "pixel_max_abs": float((out["render"]["color_pixel"].detach().cpu() - ref["color_pixel"]).abs().max()),
            "grad_rel_max": worst, "grad_rel_max_param": worst_name, "grid_grad_rel": grid,
            "idx_mismatch": mis, "idx_total": tot,
            "tolerances": "tests: loss 1e-4 relative, pixels 1e-4 absolute, gradients 2e-3 relative (norm-wise)",
            "oracle": "oracle/graph.py, torch-CPU fp32: pinned to the reference for a1 / a2 / a12 / ray generation / schedules / hash "
                      "layout / initialiser only (no executable reference for the grid, MLP, renderer arithmetic in this image)"}


def fp32_engine_leg(step, steps=10):
    """The same step with every dense layer on the strict-fp32 engine (v_mfma_f32_32x32x2_f32, exact fp32 products,
    csrc/mlp.hip): what the headline's emulated-fp32 arithmetic (f16 x 3) is to be read beside.  Captured into ITS OWN HIP
    graph (own packed weights, held by the graph, released with it) and replayed, like the headline: a kernel-bound number,
    not the host's launch rate (round 5 timed an eager loop here)."""
    from ndjir_amd import mlp
    old = mlp.get_math()
    mlp.set_math(mlp.MATH_FP32)
    graph = None
    try:
        for _ in range(2):
            loss = step.forward_backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step.forward_backward()
        torch.cuda.synchronize()
        eager_ms = 1e3 * (time.perf_counter() - t0) / 3
        mode, err = "eager stream launches", None
        try:
            graph, loss = capture_step(step)
            mode = "one captured HIP graph per step, replayed"
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
            torch.cuda.synchronize()
        fn = graph.replay if graph is not None else step.forward_backward
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
            loss = loss if graph is not None else r
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out = {"ms_per_step": 1e3 * el / steps, "rays_per_s": step.B * step.R * steps / el, "steps": steps, "loss": float(loss),
               "execution": mode, "eager_ms_per_step": eager_ms,
               "engine": "NDJIR_MLP_MATH=fp32: fp32-input MFMA (157 TFLOP/s peak), per-layer weight gradients"}
        if err is not None:
            out["graph_capture_error"] = err
        return out
    finally:
        del graph
        mlp.set_math(old)


def _all_ranks_ok(step, ok):
    """Logical AND of a per-rank flag (every rank must take the same branch: the step holds collectives)."""
    if not step.multi:
        return ok
    t = torch.tensor([1.0 if ok else 0.0], device=step.device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
    return float(t.item()) > 0.5


def capture_step(step):
    """Capture the compute part of one step (sampling, forward, loss, backward to every parameter gradient,
    gradient packing -- all custom launches are stream-ordered and allocation-stable, no collective inside)
    into a HIP graph.  Returns (graph, loss tensor of the captured step) or raises; on N > 1 every rank
    raises if any rank failed, after completing the same sequence of collectives."""
    multi = step.multi
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ref = step.forward_backward()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    step.pre_exchange()
    # N > 1: other threads of the process (the collective library's watchdog) keep issuing runtime calls
    kw = dict(capture_error_mode="thread_local") if multi else {}
    err, loss = None, None
    from ndjir_amd import mlp
    try:
        # the graph keeps every library buffer it was captured with (packed weights, workspaces) alive for as long as it
        # lives itself: a later engine / tile-height switch or a cache clean-up cannot pull them from under a replay
        with mlp.hold_buffers() as held:
            with torch.cuda.graph(g, **kw):
                loss = step.compute()
        g.ndjir_held = held
        g.ndjir_engine = mlp.engine_state()
        g.ndjir_exchange_generation = step.exchange_generation()
    except Exception as e:
        err = f"{type(e).__name__}: {e}"
    if not _all_ranks_ok(step, err is None):
        torch.cuda.synchronize()
        step.compute()                 # finish the step that pre_exchange started, identically on all ranks
        step.exchange()
        raise RuntimeError(err or "graph capture failed on another rank")
    g.replay()
    step.exchange()
    torch.cuda.synchronize()
    same = abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    if not _all_ranks_ok(step, same):
        raise RuntimeError(f"graph replay loss {float(loss)} != eager loss {float(ref)} (on some rank)")
    return g, loss


def replay_step(step, graph):
    if os.environ.get("NDJIR_BENCH_TRACE"):
        def T(fn):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0)
        a, b, c = T(step.pre_exchange), T(graph.replay), T(step.exchange)
        print(f"[rank {step.rank}] pre {a:.1f} ms, graph {b:.1f} ms, exchange {c:.1f} ms", file=sys.stderr, flush=True)
        return
    step.pre_exchange()
    if step.multi and getattr(graph, "ndjir_exchange_generation", 0) != step.exchange_generation():
        # the sparse exchange re-created its lists (grown capacity -- cannot happen while the number of query points is fixed,
        # as it is here): the captured re-arm names the old ones.  A benchmark must not die in its timed loop: the step is
        # issued as ordinary launches from now on and the line says so (`graph_capture_error`)
        graph.ndjir_stale = True
        step.compute()
    else:
        graph.replay()
    if step.multi and torch.distributed.get_backend() != "nccl":
        torch.cuda.current_stream().synchronize()   # gloo stages through the host: it must see the replayed step's results
    step.exchange()


def redraw_leg(step, graph, loss_t, steps, barrier):
    """The timed replays again, but as a training loop sees them (python/train.py:124-133): before every replay the next
    batch of pixels comes from the device data feed (ndjir_amd/dataset.py: new rays, colours -- other grid cells than the
    step before) and every random tensor is redrawn on the device.  The headline number replays ONE set of rays, whose
    ~1.6 M touched grid cells can sit in the 256 MiB Infinity Cache from replay to replay; this one cannot."""
    import copy
    import numpy as np
    from ndjir_amd.dataset import IDRRaySource
    from ndjir_amd.synthetic import make_scene
    conf = copy.deepcopy(step.conf)
    conf.train.n_rays = step.R
    images, masks, Ks, poses = make_scene(8, 256, 256, seed=5)
    src = IDRRaySource(images, masks, Ks, poses, conf, rng=np.random.RandomState(313), device=step.device)
    gen = torch.Generator(device=step.device).manual_seed(0)

    def one():
        color, _mask, raydir, camloc = src.next_batch(step.B)
        step.set_rays(camloc, raydir, color)
        step.redraw_rand(gen)
        if graph is not None:
            graph.replay()
        else:
            step.forward_backward()
    for _ in range(2):
        one()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    barrier()
    el = time.perf_counter() - t0
    loss = float(loss_t if graph is not None else step.loss)      # (the captured step writes the tensor it was captured with)
    return {"ms_per_step": 1e3 * el / steps, "rays_per_s": step.B * step.R * steps / el, "steps": steps, "loss": loss,
            "scope": "fwd+bwd replays with the next batch from the device data feed (8 synthetic views, 256 x 256) and freshly "
                     "drawn random tensors before every replay; includes the feed's launches and one small host->device copy"}


def b4_leg(conf, R, device, steps, barrier, use_graph, B=4):
    """The reference's real training-step shape: train.batch_size = 4 images x train.n_rays = 512 rays
    (config/default.yaml:126-127, python/train.py:38-51), forward + backward.  Beside the headline metric (B = 1), not it.
    (B = 1: a second step of another size, for `--scaling strong`'s projection.)"""
    step = Step(conf, R, device, 0, 1, B=B)
    for _ in range(2):
        step.forward_backward()
    graph, mode = None, "eager stream launches"
    if use_graph:
        try:
            graph, _ = capture_step(step)
            mode = "one captured HIP graph per step"
        except Exception:
            torch.cuda.synchronize()
    fn = (lambda: replay_step(step, graph)) if graph is not None else step.forward_backward
    fn()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier()
    el = time.perf_counter() - t0
    return {"ms_per_step": 1e3 * el / steps, "rays_per_s": B * R * steps / el, "steps": steps, "images": B, "rays_per_image": R,
            "execution": mode, "loss": float(step.loss)}


def train_leg(step, steps, barrier, use_graph):
    """Full training iterations (fwd+bwd + optimizer), timed like the main region.  Reported next to the headline
    metric, never as it: the metric is fwd+bwd (BASELINE.json)."""
    from ndjir_amd import mlp
    step.enable_training()
    for _ in range(2):
        step.train_step()
    torch.cuda.synchronize()
    mode, err, graph = "eager stream launches", None, None
    if use_graph and not step.multi:
        try:
            mlp._PACK_CACHE.clear()          # the weights change every step: their re-packing must be part of the graph
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss = step.train_compute()
            g.replay()
            torch.cuda.synchronize()
            graph, mode = g, "one captured HIP graph per training iteration"
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
            torch.cuda.synchronize()
            mlp._PACK_CACHE.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        if graph is not None:
            graph.replay()
        else:
            loss = step.train_step()
    barrier()
    el = time.perf_counter() - t0
    # a replayed optimizer graph rewrites the weights without bumping tensor._version (increment_version ran at capture
    # only), and the packed-weight cache keys on (data_ptr, _version): drop it, or an eager forward after the replays
    # would run on weights packed before the last update
    mlp._PACK_CACHE.clear()
    # the optimizer alone (HIP events on the launching stream, ordinary launches)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    step.solvers.zero_grad()
    step.solvers.weight_decay()
    step.set_solver_gradients()
    sparse = not step.multi and step.conf.geometric_network.voxel.type == "voxel"
    reps = 5
    e0.record()
    for _ in range(reps):
        step.solvers.solver_weight._check()
        step.solvers.solver_weight.update()
    e1.record()
    for _ in range(reps):
        step.solvers.solver_feat.update()
    e2.record()
    torch.cuda.synchronize()
    n_grid = sum(p.numel() for p in step.grid_params)
    grid_ms = e1.elapsed_time(e2) / reps
    out = {"ms_per_step": 1e3 * el / steps, "steps": steps, "execution": mode,
           "loss_after": float(loss),
           "optimizer": {"mlp_ms": e0.elapsed_time(e1) / reps, "grid_ms": grid_ms,
                         "grid_params": n_grid,
                         "grid_bytes_per_launch": (24 if sparse else 32) * n_grid,
                         "grid_GBps": (24 if sparse else 32) * n_grid / (grid_ms * 1e-3) / 1e9 if grid_ms > 0 else None,
                         "note": ("ndjir::k_adam<4, true, true> (+ the two mark_touched launches): per float 3 reads (w, m, v) + "
                                  "3 writes = 24 B, g read / cleared only in the < 1 % touched cells; HBM peak 8000 GB/s") if sparse else
                                 ("ndjir::k_adam: per float 4 reads (w, g, m, v) + 4 writes (w, m, v, g = 0) = 32 B; "
                                  "HBM peak 8000 GB/s")}}
    if err is not None:
        out["graph_capture_error"] = err
    return out


def committed_pmc_traffic(kernels):
    """HBM bytes per launch of the kernel class made of the symbols `kernels` (launch-weighted mean) from the committed
    rocprofv3 PMC passes of this same command (profiles/<round>_pmc_hbm_bench.txt: FETCH_SIZE and WRITE_SIZE in KB per launch,
    collected in separate runs as MI355X_MICROARCH.md prescribes; FETCH doubled per its gfx950 note -- calibrated here on
    ndjir::k_adam, whose 6.44 GB of reads per launch show as 3.17 GB).  Not a live measurement: None if the file is absent."""
    if isinstance(kernels, str):
        kernels = [kernels]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"{PROFILE_ROUND}_pmc_hbm_bench.txt")
    try:
        vals, section = {}, None
        lines = open(path).read().splitlines()
        for i, line in enumerate(lines):
            if line.startswith("## "):
                section = line[3:].strip()
                continue
            for k in kernels:
                if section and line.startswith(k + "  launches=") and i + 1 < len(lines):
                    n = int(line.split("launches=")[1].split()[0])
                    vals.setdefault(k, {})[section] = (n, float(lines[i + 1].split()[-1]))
        tot, cnt = 0.0, 0
        for k, v in vals.items():
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                n = v["FETCH_SIZE"][0]
                tot += n * (2.0 * v["FETCH_SIZE"][1] + v["WRITE_SIZE"][1]) * 1024.0
                cnt += n
        return tot / cnt if cnt else None
    except Exception:
        return None


def committed_kernel_only_us(symbols):
    """Kernel-only time per launch of a symbol -- or, for several symbols (a library call), the sum of their time per step --
    from the committed rocprofv3 --kernel-trace --stats summary of this command (profiles/<round>_bench_kernel_summary.txt:
    columns kernel, calls/step, ms/step, %, avg us).  None if the file or the symbol is absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"{PROFILE_ROUND}_bench_kernel_summary.txt")
    try:
        rows = {}
        for line in open(path).read().splitlines()[2:]:
            parts = line.rsplit(None, 4)
            if len(parts) == 5:
                rows[parts[0].strip()] = (float(parts[1]), float(parts[2]), float(parts[4]))
        if len(symbols) == 1:
            return rows[symbols[0]][2] if symbols[0] in rows else None
        tot = sum(rows[k][1] for k in symbols if k in rows)
        return 1e3 * tot if tot > 0 else None
    except Exception:
        return None


def kernel_report(profile, by_symbol=False):
    """Aggregate the live HIP-event records of the MLP engine (ndjir_amd.mlp.PROFILE) by class, or by kernel symbol."""
    agg = {}
    for kind, flops, e0, e1, _shape, nbytes, sym, n_kernels in profile:
        a = agg.setdefault(sym if by_symbol else kind, [0, 0.0, 0.0, 0.0])
        a[0] += n_kernels            # (a grouped weight-gradient call = one kernel launch per argument block)
        a[1] += flops
        a[2] += e0.elapsed_time(e1) * 1e-3
        a[3] += nbytes
    rep = {}
    for kind, (n, flops, sec, nbytes) in agg.items():
        rep[kind] = dict(launches=n, avg_us=1e6 * sec / max(n, 1), tflops=flops / max(sec, 1e-12) / 1e12,
                         gflop_per_launch=flops / max(n, 1) / 1e9, mbytes_per_launch=nbytes / max(n, 1) / 1e6,
                         tbps=nbytes / max(sec, 1e-12) / 1e12)
    return rep


def kernel_table(kr, steps, peak):
    """Per kernel class of the MLP engine: launches and time per step, algorithmic TFLOP/s and its fraction of both
    yardsticks -- the peak of the executed instruction mix (dense 16-bit MFMA / partial products per algorithmic FLOP) and
    the fp32-input MFMA peak (what an fp32 GEMM engine could reach at best on this chip)."""
    out = {}
    for k, v in kr.items():
        out[k] = dict(launches_per_step=round(v["launches"] / max(steps, 1), 2), avg_us=round(v["avg_us"], 2),
                      ms_per_step=round(v["launches"] * v["avg_us"] / max(steps, 1) / 1e3, 4),
                      gflop_per_launch=round(v["gflop_per_launch"], 3), tflops=round(v["tflops"], 2),
                      frac_of_executed_mix_peak=round(v["tflops"] / peak, 4),
                      frac_of_fp32_mfma_peak=round(v["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4),
                      mbytes_per_launch=round(v["mbytes_per_launch"], 2), hbm_tbps=round(v["tbps"], 3),
                      frac_of_hbm_peak=round(v["tbps"] * 1e3 / PEAK_HBM_GBPS, 4))
    return out


def kernel_description(sym, math):
    from ndjir_amd import mlp
    if "k_wgrad" in sym:
        return ("weight gradients A^T delta of every layer of the step in one grouped call (csrc/wgrad.hip: k_wgrad_group_wide -- "
                "128 x 256 items, one workgroup per CU -- then k_wgrad_group -- 128 x 128 tiles, strips, narrow outputs -- over a "
                "work table in device memory, P split over workgroups) + the split reduction k_wgrad_group_reduce (3 launches); "
                "the event interval spans the call's 7-8 launches, rocprofv3's kernel-only sum of the same is ~12 % less")
    what = {"<0": "forward", "<1": "backward", "<2": "tangent"}.get(sym[sym.find("<"):sym.find("<") + 2], "")
    if "k_chainp" in sym:
        return (f"fused MLP {what} chain of a training pass on 128-point tiles, the tile's two 64-point halves software-pipelined "
                "(csrc/mlp3p.hip: the k-loop MFMAs of one half carry the epilogue instructions of the other, one MFMA then one "
                "slice of an epilogue item; template argument: mode): fp32 operands scaled by powers of two and split into 2 f16 "
                "planes (lo unscaled), 3 v_mfma_f32_32x32x16_f16 partial products per fp32 product in ONE fp32 accumulator"
                + ("; k_chainp_nets: up to three nets of one MultiMLP in one launch" if "k_chainp_nets" in sym else ""))
    if "k_chainw" in sym:
        return (f"fused MLP {what} chain on 128-point tiles, epilogue in the accumulator registers (csrc/mlp3w.hip; template "
                "arguments: mode, row blocks per wave -- 4: nets up to 256 wide, 2: up to 128 wide --, waves per workgroup): fp32 "
                "operands scaled by powers of two and split into 2 f16 planes, 3 v_mfma_f32_32x32x16_f16 partial products per fp32 "
                "product in two fp32 accumulators"
                + ("; k_chainw_nets: up to three nets of one MultiMLP in one launch, a workgroup takes its tile through them "
                   "in turn (the per-sample material nets)" if "k_chainw_nets" in sym else ""))
    if "k_chain3" in sym:
        return f"fused MLP {what} chain, 64 / 32-point tiles (csrc/mlp3.hip; small launches: the sampler's rounds, per-ray nets)"
    if "k_chain6" in sym:
        return f"fused MLP {what} chain; 3 bf16 planes, 6 partial products per fp32 product (csrc/mlp6.hip)"
    return f"fused MLP {what} chain, fp32 MFMA 32x32x2 (csrc/mlp.hip)"


def kernel_detail(profile, steps):
    """Per (kind, shape) table of the engine's launches -> stderr (NDJIR_BENCH_DETAIL=1)."""
    agg = {}
    for kind, flops, e0, e1, shape, nbytes, _sym, _n in profile:
        a = agg.setdefault((kind, shape), [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += flops
        a[2] += e0.elapsed_time(e1) * 1e-3
        a[3] += nbytes
    print(f"{'kind':10s} {'shape':48s} {'n/step':>6s} {'avg us':>8s} {'ms/step':>8s} {'TFLOP/s':>8s} {'MB':>8s} {'TB/s':>6s}", file=sys.stderr)
    for (kind, shape), (n, flops, sec, nbytes) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        print(f"{kind:10s} {shape:48s} {n / steps:6.1f} {1e6 * sec / n:8.1f} {1e3 * sec / steps:8.3f} "
              f"{flops / max(sec, 1e-12) / 1e12:8.1f} {nbytes / n / 1e6:8.1f} {nbytes / max(sec, 1e-12) / 1e12:6.2f}", file=sys.stderr)


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, as the driver's torchrun command would --
    a child process (`python -m torch.distributed.run ... bench.py <same arguments>`), issued BEFORE this process touches
    the GPU; its output is relayed, its exit code is ours.  (Never an exec: a process that initialised the GPU must not
    replace itself.)"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("NDJIR_BENCH_SAME_DEVICE"):
        local_rank = 0      # debugging aid: several ranks on ONE GPU over gloo, to exercise the N > 1 code path
    force_dist = world == 1 and bool(os.environ.get("NDJIR_BENCH_FORCE_DIST"))
    if world > 1 or force_dist:
        import torch.distributed as dist
        if force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29555")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        elif os.environ.get("NDJIR_BENCH_SAME_DEVICE"):
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from ndjir_amd import config as cfg
    from ndjir_amd import lib, mlp
    lib.load()  # fail loudly if the HIP extension is missing
    conf = cfg.load(a.config, a.override)
    R = a.rays
    if a.scaling == "strong":
        assert a.total_rays % world == 0, f"--total-rays {a.total_rays} does not split over {world} ranks"
        R = a.total_rays // world
    step = Step(conf, R, device, rank, world)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # default: graph replay of the compute part; on N > 1 the collectives are issued eagerly around every replay
    # (validated against RCCL with a 1-rank group, NDJIR_BENCH_FORCE_DIST=1: 12.9 ms/step vs 15.9 ms eager -- the
    # exchange's host synchronisation serialises the eager step's launch time with the GPU's).  Safety nets below:
    # capture failure, loss mismatch or a replay that is not faster than eager launches all fall back to eager.
    exec_mode = a.exec_mode or os.environ.get("NDJIR_BENCH_EXEC") or "graph"
    if step.multi and os.environ.get("NDJIR_BENCH_SAME_DEVICE") and not (a.exec_mode or os.environ.get("NDJIR_BENCH_EXEC")):
        exec_mode = "eager"          # several processes replaying graphs on ONE GPU time-slice badly (debug harness only)
    for _ in range(a.warmup):
        step.forward_backward()
    eager = None
    graph_error = None
    profile_steps = a.steps

    def timed(fn, n=2):
        barrier()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        barrier()
        return time.perf_counter() - t
    # eager reference time taken BEFORE anything is captured (N > 1: what the replay has to beat)
    t_eager_pre = timed(step.forward_backward) if (exec_mode == "graph" and step.multi) else None
    if exec_mode == "graph":
        try:
            graph, loss = capture_step(step)          # untimed, like the warm-up
        except Exception as e:                        # capture is an optimisation: never lose the measurement to it
            graph_error = f"{type(e).__name__}: {e}"
            exec_mode = "eager"
            torch.cuda.synchronize()
        dbg = (lambda tag: print("DEBUG", tag, float(loss), file=sys.stderr)) if os.environ.get("NDJIR_BENCH_DEBUG") else (lambda tag: None)
        if exec_mode == "graph":
            ref_loss = float(loss)       # the captured step's loss right after capture (= the eager step's, checked by capture_step)
            dbg("after capture")
            # ... and never a pessimisation: two untimed steps each way, keep replaying only if it is not slower
            t_graph = timed(lambda: replay_step(step, graph))
            dbg("after 2 replays")
            t_eager = t_eager_pre if t_eager_pre is not None else timed(step.forward_backward)
            if not _all_ranks_ok(step, t_graph <= 1.05 * t_eager):
                graph_error = f"replay not faster than eager launches ({1e3 * t_graph / 2:.1f} vs {1e3 * t_eager / 2:.1f} ms/step)"
                exec_mode = "eager"
    if exec_mode == "graph":
        dbg("after eager trial")
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            replay_step(step, graph)
        barrier()
        el = time.perf_counter() - t0
        dbg("after timed replays")
        # the inputs did not change: the timed replays must still produce the eager step's loss -- a replay that computed
        # something else is not a measurement of the step (seen once: a 64-ray tri-plane debug configuration after eager
        # steps had run between replays); fall back to timing ordinary launches
        if not _all_ranks_ok(step, abs(float(loss) - ref_loss) <= 1e-5 * abs(ref_loss)):
            graph_error = f"timed replays produced loss {float(loss)} instead of {ref_loss}: timed as eager launches instead"
            barrier()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                loss = step.forward_backward()
            barrier()
            el = time.perf_counter() - t0
            exec_mode = "eager (graph replay rejected)"
        # HIP events cannot be recorded inside a captured graph: the same K steps are issued once more as
        # ordinary stream launches, with events around every engine launch, for `roofline` / `kernels`
        mlp.PROFILE = [] if rank == 0 else None
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step.forward_backward()
        barrier()
        eager = {"ms_per_step": 1e3 * (time.perf_counter() - t1) / a.steps,
                 "note": "same K steps as ordinary stream launches (with the HIP-event instrumentation)"}
        profile, mlp.PROFILE = mlp.PROFILE, None
    elif world > 1:
        # N > 1: the event instrumentation costs rank 0 host time every step, which the other ranks then wait
        # for.  `roofline` / `kernels` come from two extra untimed steps (all ranks run them: the step contains
        # collectives); the timed region is clean.
        mlp.PROFILE = [] if rank == 0 else None
        for _ in range(2):
            step.forward_backward()
        profile, mlp.PROFILE = mlp.PROFILE, None
        profile_steps = 2
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step.forward_backward()
        barrier()
        el = time.perf_counter() - t0
    else:
        barrier()
        mlp.PROFILE = []
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step.forward_backward()
        barrier()
        el = time.perf_counter() - t0
        profile, mlp.PROFILE = mlp.PROFILE, None
    ranks_info = None
    if world > 1:
        # what the collective library saw: one line per rank (device, its own time over the timed region), so that a scaling
        # record can be checked for "N ranks on N devices" and for the spread between them
        import torch.distributed as dist
        mine = dict(rank=rank, local_rank=local_rank, device=torch.cuda.current_device(), name=torch.cuda.get_device_name(),
                    pci=getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "pci_bus_id", None),
                    ms_per_step=1e3 * el / a.steps)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks_info = dict(world_size=dist.get_world_size(), backend=dist.get_backend(), devices=gathered,
                          ms_per_step_min=min(g["ms_per_step"] for g in gathered),
                          ms_per_step_max=max(g["ms_per_step"] for g in gathered))
        t = torch.tensor([el], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())

    if rank == 0:
        r = conf.renderer
        N = r.n_samples0 + r.n_samples1 * r.n_upsamples
        rays_per_s = world * R * a.steps / el
        ms = 1e3 * el / a.steps
        kr = kernel_report(profile)
        if os.environ.get("NDJIR_BENCH_DETAIL"):
            kernel_detail(profile, profile_steps)
        step_tflops = MFLOP_PER_RAY_FWD_BWD * 1e6 * R / (ms * 1e-3) / 1e12      # per GPU
        math = mlp.get_math()
        x6, x3 = math == mlp.MATH_BF16X6, math == mlp.MATH_F16X3
        peak = PEAK_F16X3_EFFECTIVE_TFLOPS if x3 else PEAK_BF16X6_EFFECTIVE_TFLOPS if x6 else PEAK_FP32_MFMA_TFLOPS
        # the roofline block is for the KERNEL (one symbol, as rocprofv3 prints it) with the largest time per step, with its
        # own launches' FLOPs and bytes -- not a class average over several instantiations
        table = kernel_table(kr, profile_steps, peak)
        ks = kernel_report(profile, by_symbol=True)
        sym_table = kernel_table(ks, profile_steps, peak)
        # `roofline`: the dominant SINGLE kernel symbol (what rocprofv3's --stats table puts first); a library CALL that
        # spans several kernels (the grouped weight gradients: 7-8 launches of three symbols) is reported beside it as
        # `roofline_call`, never instead of it
        call_syms = {rec[6] for rec in profile if rec[7] > 1 or "k_wgrad" in rec[6]}
        single = {k: v for k, v in sym_table.items() if k not in call_syms}
        ksym = max(single, key=lambda k: single[k]["ms_per_step"]) if single else "none"
        csym = max((k for k in sym_table if k in call_syms), key=lambda k: sym_table[k]["ms_per_step"], default=None)
        peak_note = ("dense 16-bit MFMA peak 2500 TFLOP/s (MI355X_MICROARCH.md) / 3 partial products per algorithmic FLOP" if x3 else
                     "dense bf16 MFMA peak 2500 TFLOP/s (MI355X_MICROARCH.md) / 6 partial products per algorithmic FLOP" if x6 else
                     "fp32-input MFMA peak (MI355X_MICROARCH.md)")
        dtype = ("f32 (f16x2-split MFMA products, f32 accumulate; error vs fp64 below an fp32 FMA chain's)" if x3 else
                 "f32 (bf16x3-split MFMA products, f32 accumulate)" if x6 else "f32")
        ridge = peak * 1e12 / (PEAK_HBM_GBPS * 1e9)      # FLOP per byte at which the two roofs meet

        def roof_block(sym, pmc_syms, kernel_only_syms):
            """Roofline of one symbol (or call): both yardsticks from the live HIP events; `bound` by the launch's arithmetic
            intensity against the ridge (not by whichever fraction happens to be larger); the same fraction again from the
            kernel-only average duration of the committed rocprofv3 summary; HBM traffic from the committed PMC passes."""
            dom = ks.get(sym, dict(tflops=0.0, avg_us=0.0, launches=0, gflop_per_launch=0.0, tbps=0.0, mbytes_per_launch=0.0))
            kind = next((rec[0] for rec in profile if rec[6] == sym), "none")
            mfma = {"achieved": dom["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": dom["tflops"] / peak,
                    "algorithmic_gflop_per_launch": dom["gflop_per_launch"]}
            hbm = {"achieved": dom.get("tbps", 0.0) * 1e3, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                   "frac": dom.get("tbps", 0.0) * 1e3 / PEAK_HBM_GBPS,
                   "algorithmic_mbytes_per_launch": dom.get("mbytes_per_launch", 0.0),
                   "note": "algorithmic bytes = every (points, width) fp32 tensor of the launch counted once (ndjir_amd/mlp.py "
                           "_launch_bytes): chain input, output, per hidden layer the stored activation it reads and the delta / "
                           "activation it writes; packed weights (L2-resident) not counted"}
            nbytes = dom.get("mbytes_per_launch", 0.0) * 1e6
            ai = dom["gflop_per_launch"] * 1e9 / nbytes if nbytes > 0 else float("inf")
            hbm_bound = ai < ridge
            top = ({"bound": "hbm", "achieved": hbm["achieved"], "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": hbm["frac"]}
                   if hbm_bound else
                   {"bound": "mfma", "achieved": mfma["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": mfma["frac"]})
            ko = committed_kernel_only_us(kernel_only_syms)
            ko_block = None
            if ko:
                sec = ko * 1e-6
                # (a call: `ko` is the kernel time of ALL its launches per step -> price the call's bytes / FLOPs per step)
                per = dom["launches"] / max(profile_steps, 1) if len(kernel_only_syms) > 1 else 1.0
                ko_block = {"avg_launch_us": ko,
                            "frac": (per * nbytes / sec / 1e9 / PEAK_HBM_GBPS) if hbm_bound
                                    else (per * dom["gflop_per_launch"] / sec / 1e3 / peak),
                            "source": f"profiles/{PROFILE_ROUND}_bench_kernel_summary.txt (rocprofv3 --kernel-trace --stats of this "
                                      "command, kernel-only durations: no launch gaps of the host-bound eager pass)"}
            traffic = sum(t for t in (committed_pmc_traffic(x) for x in pmc_syms) if t) or None
            return {**top,
                    "traffic": traffic,
                    "traffic_note": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) of " + " + ".join(pmc_syms) + " from the "
                                    f"committed rocprofv3 --pmc passes of this command, profiles/{PROFILE_ROUND}_pmc_hbm_bench.txt, "
                                    "beside `hbm.algorithmic_mbytes_per_launch` (null: no committed pass lists the symbol)",
                    "arithmetic_intensity_flop_per_byte": ai, "ridge_flop_per_byte": ridge,
                    "bound_rule": "hbm if algorithmic FLOP / algorithmic byte of the launch < ridge (executed-mix matrix peak / HBM peak), else mfma",
                    "kernel_only": ko_block,
                    "mfma": mfma, "hbm": hbm,
                    "kernel_symbol": sym, "kernel": kernel_description(sym, math), "peak_note": peak_note,
                    # the same algorithmic fp32 FLOP/s against the fp32-input MFMA peak (what an fp32 GEMM engine could
                    # reach at best on this chip) and against the bf16x6 engine's effective peak (round 1's yardstick)
                    "frac_of_fp32_mfma_peak": dom["tflops"] / PEAK_FP32_MFMA_TFLOPS,
                    "frac_of_bf16x6_peak": dom["tflops"] / PEAK_BF16X6_EFFECTIVE_TFLOPS,
                    "launches_per_step": dom["launches"] / max(profile_steps, 1), "avg_launch_us": dom["avg_us"],
                    "algorithmic_gflop_per_launch": dom["gflop_per_launch"],
                    "kernel_class": kind,
                    "ms_per_step": sym_table.get(sym, {}).get("ms_per_step")}
        method = (("HIP events on the launching stream around every launch of the same K steps issued "
                   "eagerly right after the timed graph replays (events cannot be recorded inside a "
                   "captured graph)") if exec_mode == "graph" else
                  ("HIP events on the launching stream around every launch of two untimed steps before "
                   "the timed region (N > 1)") if world > 1 else
                  "HIP events on the launching stream around every launch in the timed region")
        roofline = roof_block(ksym, [ksym], [ksym])
        roofline["note"] = ("the single kernel symbol with the largest time per step (`kernels_by_symbol` lists every kernel of the "
                            "engine with both yardsticks, `kernels` the same launches by class; `roofline_call` = the largest "
                            "multi-kernel library call); an event interval spans the launch gap of the host-bound eager pass as well "
                            "as the kernel, a few % more than rocprofv3's kernel-only average (`kernel_only`)")
        roofline["method"] = method
        roofline_call = None
        if csym is not None:
            wg = ["ndjir::k_wgrad_group_wide", "ndjir::k_wgrad_group", "ndjir::k_wgrad_group_reduce"]
            roofline_call = roof_block(csym, wg if "k_wgrad_group" in csym else [csym], wg if "k_wgrad_group" in csym else [csym])
            roofline_call["note"] = ("the library call with the largest time per step: one event interval over its 7-8 kernel launches "
                                     "(`kernel_only.avg_launch_us` = the sum of its symbols' per-step kernel time)")
            roofline_call["method"] = method
        out = {
            "metric": "rays/sec (fwd+bwd) at 512 rays x 128 samples",
            "value": rays_per_s, "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"config/{a.config}.yaml, B=1, R={R} rays/GPU x N={N} fg samples (+{r.n_bg_samples} bg, "
                                   f"{r.n_thetas * 2 * r.n_thetas} lights), voxel {conf.geometric_network.voxel.type} "
                                   f"{conf.geometric_network.voxel.grid_size}^3x{conf.geometric_network.voxel.feature_size}, "
                                   f"total_loss fwd+bwd to all parameter gradients",
                       "rays_per_gpu": R, "samples_per_ray": N, "parallelism": f"ray-sharded x{world}",
                       **({"total_rays": a.total_rays} if a.scaling == "strong" else {})},
            "roofline": roofline,
            "roofline_call": roofline_call,
            "kernels": table,
            "kernels_by_symbol": sym_table,
            "step_roofline": {"achieved": step_tflops, "peak": peak, "unit": "TFLOP/s",
                              "frac": step_tflops / peak,
                              "scope": "whole step: 2168.9 MFLOP/ray fwd+bwd (SURVEY 8d) / step time",
                              # the reference's formulation re-evaluates all current samples with the full 257-wide output
                              # layer in each of the 4 up-sampling rounds (323.1 MFLOP/ray); this build evaluates only the 16
                              # new samples of a round and only the sdf column (100.7): same values, fewer executed FLOPs
                              "executed_mflop_per_ray": 2168.9 - 323.1 + 128 * 0.786944,
                              "executed_tflops": step_tflops * (2168.9 - 323.1 + 128 * 0.786944) / 2168.9},
            "loss": float(loss),
        }
        if ranks_info is not None:
            out["ranks"] = ranks_info
        if step.multi:
            # an exchange that overflowed its wire size delivered an incomplete grid gradient: the measurement must say so
            rep = step.exchange_report()
            out["exchange"] = rep
            if any(b["overflowed_exchanges"] > 0 for b in rep["buffers"].values()):
                out["exchange"]["warning"] = "sparse grid exchange overflowed during the run: gradients of those steps were incomplete"
        out["execution"] = (("one captured HIP graph per step (torch.cuda.CUDAGraph), K replays timed"
                             + ("; the RCCL gradient exchange is issued eagerly between replays" if world > 1 else ""))
                            if exec_mode == "graph" else "eager stream launches")
        if eager is not None:
            out["eager"] = eager
        if exec_mode == "graph" and getattr(graph, "ndjir_stale", False):
            graph_error = ("the sparse grid exchange re-created its state during the timed region (list capacity grown): the "
                           "captured step was stale, later steps were issued as ordinary launches")
        if graph_error is not None:
            out["graph_capture_error"] = graph_error
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(conf, step, a.cpu_rays)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "rays/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
            try:
                out["parity"] = parity_block(conf, step, a.cpu_rays)
            except Exception as e:
                out["parity"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not force_dist and a.extra_legs:
        try:
            out["redraw"] = redraw_leg(step, graph if exec_mode == "graph" else None, loss if exec_mode == "graph" else None,
                                       a.steps, barrier)
        except Exception as e:      # a side measurement must never take the headline number down with it
            out["redraw"] = {"error": f"{type(e).__name__}: {e}"}
    if a.train_steps > 0:
        tl = train_leg(step, a.train_steps, barrier, use_graph=(a.exec_mode or os.environ.get("NDJIR_BENCH_EXEC") or "graph") == "graph")
        if world > 1:
            t = torch.tensor([tl["ms_per_step"]], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            tl["ms_per_step"] = float(t.item())
        if rank == 0:
            tl["rays_per_s"] = world * R / (tl["ms_per_step"] * 1e-3)
            tl["scope"] = ("fwd+bwd + weight decay + finite-gradient guard + Adam update of every parameter "
                           "(python/train.py:136-148); reported beside the headline metric, not as it")
            out["train_step"] = tl
    if world == 1 and not force_dist and a.extra_legs and rank == 0:
        # (LAST user of `step`: switching the engine re-packs the weights, and the graphs captured above hold the addresses of
        #  the packed copies they were captured with -- round 5's first placement, before the `redraw` replays, gave them NaNs)
        try:
            out["fp32_engine"] = fp32_engine_leg(step)
        except Exception as e:
            out["fp32_engine"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not force_dist and a.extra_legs:
        use_graph = (a.exec_mode or os.environ.get("NDJIR_BENCH_EXEC") or "graph") == "graph"
        try:
            del step
            graph = None
            if a.scaling == "strong":
                # BASELINE.json config 4 on ONE GPU: what 8 ranks would each do (total_rays / 8 rays), and the strong-scaling
                # ratio that follows if the exchange costs 0.3 ms per step -- a projection from single-GPU times, not a measurement
                small = b4_leg(conf, a.total_rays // 8, device, a.steps, barrier, use_graph, B=1)
                out["projected_strong_scaling_8"] = {
                    "ms_per_step_all_rays_one_gpu": ms, "ms_per_step_one_eighth": small["ms_per_step"], "assumed_exchange_ms": 0.3,
                    "value": ms / (small["ms_per_step"] + 0.3),
                    "note": f"t({a.total_rays} rays) / (t({a.total_rays // 8} rays) + 0.3 ms), both on this one GPU"}
            else:
                out["b4"] = b4_leg(conf, R, device, a.steps, barrier, use_graph)
        except Exception as e:
            out["b4" if a.scaling != "strong" else "projected_strong_scaling_8"] = {"error": f"{type(e).__name__}: {e}"}
    if world > 1 or force_dist:
        torch.distributed.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which is block-buffered on a pipe and would otherwise land
        # after this line at exit: flush it first so that the JSON line is the last thing on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
