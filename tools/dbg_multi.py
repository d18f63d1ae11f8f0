"""Debug: graph capture/replay of the compute part with a process group alive (N ranks share GPU 0 over gloo)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import bench
from ndjir_amd import config as cfg
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
step = bench.Step(cfg.load("default", []), 512, dev, rank, world)
def T(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for i in range(2):
    step.forward_backward()
mode = os.environ.get("CAPMODE", "thread_local")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step.forward_backward()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
step.pre_exchange()
with torch.cuda.graph(g, capture_error_mode=mode):
    step.compute()
g.replay(); step.exchange(); torch.cuda.synchronize()
for i in range(3):
    a = T(step.pre_exchange); b = T(g.replay); c = T(step.exchange)
    if rank == 0: print(f"[{mode}] replay step {i}: pre {a:.1f} ms, graph {b:.1f} ms, exchange {c:.1f} ms", flush=True)
for i in range(2):
    a = T(step.pre_exchange); b = T(step.compute); c = T(step.exchange)
    if rank == 0: print(f"[{mode}] eager  step {i}: pre {a:.1f} ms, compute {b:.1f} ms, exchange {c:.1f} ms", flush=True)
dist.destroy_process_group()
