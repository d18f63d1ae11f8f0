"""torch.profiler view of one bench step: device time per ATen op and input shape (which parts of
the torch glue around the HIP kernels cost what).  usage: python tools/torch_profile.py [rows]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ndjir_amd import config as cfg  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    conf = cfg.load("default", [])
    step = bench.Step(conf, 512, dev, 0, 1)
    for _ in range(2):
        step.forward_backward()
    torch.cuda.synchronize()
    n = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(n):
            step.forward_backward()
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=True)
    evs = sorted(ka, key=lambda e: -e.self_device_time_total)
    tot = sum(e.self_device_time_total for e in evs)
    print(f"self device time total {tot / n / 1e3:.2f} ms/step")
    print(f"{'op':40s} {'calls/step':>10s} {'ms/step':>8s}  shapes")
    if os.environ.get("ATEN_ONLY"):
        evs = [e for e in evs if e.key.startswith("aten::")]
        print(f"aten ops only: {sum(e.self_device_time_total for e in evs) / n / 1e3:.2f} ms/step")
    for e in evs[:rows]:
        print(f"{e.key[:40]:40s} {e.count / n:10.1f} {e.self_device_time_total / n / 1e3:8.3f}  {str(e.input_shapes)[:110]}")


if __name__ == "__main__":
    main()
