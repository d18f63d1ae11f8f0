"""Chain kernel time vs number of tiles (one 64-point tile per workgroup, one workgroup per CU at a time):
separates per-tile compute from dispatch / tail effects.  usage: python tools/chain_scaling.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndjir_amd.mlp import chain_forward  # noqa: E402
from kernel_bench import make  # noqa: E402

dims = (43, 256, 256, 256, 213, 256, 256, 256, 257)
Ws, bs = make(dims, 1, 3)
for keep in (False, True):
    for P in (8192, 16384, 32768, 65536, 131072, 262144):
        x = torch.randn(P, 43, device="cuda")
        for _ in range(3):
            chain_forward(x, Ws, bs, 100.0, 3, 0.7071, keep_hidden=keep)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            chain_forward(x, Ws, bs, 100.0, 3, 0.7071, keep_hidden=keep)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        fl = 2 * P * sum(a * b for a, b in zip((43, 256, 256, 256, 256, 256, 256, 256), dims[1:]))
        print(f"keep_hidden={keep} P={P:7d} tiles={P // 64:5d}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  ({us / max(P // 64 / 256, 1):.1f} us per round of 256 tiles)")
