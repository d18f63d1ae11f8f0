"""Can one training step (forward + backward to all parameter gradients) be captured into a HIP graph
and replayed?  Prints eager vs replay time per step and checks that the gradients agree."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ndjir_amd import config as cfg  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
step = bench.Step(cfg.load("default", []), 512, dev, 0, 1)
for _ in range(3):
    loss = step.forward_backward()
torch.cuda.synchronize()
ref = [g.clone() if g is not None else None for g in step.grads]
K = 10
t0 = time.perf_counter()
for _ in range(K):
    step.forward_backward()
torch.cuda.synchronize()
print(f"eager  {1e3 * (time.perf_counter() - t0) / K:.2f} ms/step  loss {float(loss):.10f}")

g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step.forward_backward()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    static_loss = step.forward_backward()
    static_grads = step.grads
torch.cuda.synchronize()
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    g.replay()
torch.cuda.synchronize()
print(f"replay {1e3 * (time.perf_counter() - t0) / K:.2f} ms/step  loss {float(static_loss):.10f}")
worst = 0.0
for a, b in zip(static_grads, ref):
    if a is None:
        continue
    worst = max(worst, float((a - b).abs().max() / (b.abs().max() + 1e-20)))
print("max relative gradient difference eager vs replay:", worst)
