"""Sparse re-zeroing of the 2 GiB grid gradient buffer vs a dense zero fill: same gradients, time per step."""
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
sparse = {k: v.clone() for k, v in step.grid_bufs.items()}
step.touched = None                      # force the dense path for one step
step.forward_backward()
for k, v in step.grid_bufs.items():
    d = (v - sparse[k]).abs().max().item()
    print(k, "max |sparse - dense| =", d, " max |grad| =", v.abs().max().item(), " nonzeros", int((v != 0).sum()), int((sparse[k] != 0).sum()))


def timeit(dense, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if dense:
            step.touched = None
        step.forward_backward()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print("dense  ms/step", timeit(True))
print("sparse ms/step", timeit(False))
