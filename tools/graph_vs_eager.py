"""Debug: loss terms of the eager step vs the same step replayed from a captured HIP graph.
usage: python tools/graph_vs_eager.py <config> <rays> <grid>"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ndjir_amd import config as cfg, loss as L

variant, R, G = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
conf = cfg.load(variant, [f"geometric_network.voxel.grid_size={G}"])
dev = torch.device("cuda", 0)
stash = {}
_tl = L.total_loss


def tl(*a, **k):
    out = _tl(*a, **k)
    d = {k2: v.detach() for k2, v in out.items() if torch.is_tensor(v) and v.dim() == 0}
    s = out["samples"]
    d["x_fg_sum"] = s["x_fg"].detach().double().sum()
    d["t_fg_sum"] = s["t_fg"].detach().double().sum()
    d["x_bg_sum"] = s["x_bg"].detach().double().sum()
    d["color_sum"] = out["render"]["color_pixel"].detach().double().sum()
    d["sdf_sum"] = out["render"]["sdf_x_fg"].detach().double().sum()
    d["grad_sum"] = out["render"]["grad_x_fg"].detach().double().sum()
    d["alpha_sum"] = out["render"]["alpha_fg"].detach().double().sum()
    stash["out"] = d
    return out


import ndjir_amd.loss
ndjir_amd.loss.total_loss = tl
step = bench.Step(conf, R, dev, 0, 1)


def terms():
    return {k: float(v) for k, v in stash["out"].items()}


for i in range(2):
    step.forward_backward()
    torch.cuda.synchronize()
    e = terms()
    print("eager", i, {k: round(v, 7) for k, v in e.items()})
graph, loss_t = bench.capture_step(step)
for i in range(2):
    graph.replay()
    torch.cuda.synchronize()
    g = terms()
    print("graph", i, {k: round(v, 7) for k, v in g.items()})
for k in e:
    if abs(e[k] - g[k]) > 1e-6 * max(abs(e[k]), 1e-3):
        print("DIFF", k, e[k], g[k])
# the sequence bench.py runs: replays, eager steps, replays again
print("sequence:")
for tag, fn in (("replay", lambda: bench.replay_step(step, graph)), ("eager", step.forward_backward), ("replay", lambda: bench.replay_step(step, graph)),
                ("eager", step.forward_backward), ("replay", lambda: bench.replay_step(step, graph))):
    for i in range(2):
        fn()
        torch.cuda.synchronize()
        print(" ", tag, i, "captured loss tensor", float(loss_t), "terms", {k: round(v, 6) for k, v in terms().items() if k in ("loss", "loss_tv", "sdf_sum", "x_fg_sum")})
print("back-to-back replays (no synchronisation in between):")
for n in (2, 3):
    for _ in range(n):
        graph.replay()
    torch.cuda.synchronize()
    t = terms()
    print(" ", n, "replays:", {k: round(v, 6) for k, v in t.items()})
    for k in e:
        if abs(e[k] - t[k]) > 1e-6 * max(abs(e[k]), 1e-3):
            print("   DIFF", k, e[k], t[k])
