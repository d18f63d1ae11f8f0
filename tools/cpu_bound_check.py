"""Is the step CPU(launch)-bound?  Compares the host time to enqueue K steps with the time until the
GPU has finished them."""
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
    step.forward_backward()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    step.forward_backward()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / K:.2f} ms/step, finished {1e3 * (t2 - t0) / K:.2f} ms/step")
