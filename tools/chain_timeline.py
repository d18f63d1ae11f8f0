"""Phase timeline of the fused MLP chain kernel (workgroup 0, first tile): shader-clock stamps per
layer and wave -> where the cycles of a layer go (k-loop / staging / epilogue / barrier wait).
usage: python tools/chain_timeline.py [fwd|bwd]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndjir_amd import lib  # noqa: E402
from ndjir_amd.mlp import chain_forward, fused_mlp  # noqa: E402
from kernel_bench import make  # noqa: E402


def show(buf, L, title):
    t = buf.cpu().numpy().reshape(10, 5, 8)[:L].astype(np.int64)
    t0 = t[0, 0].min()
    print(title)
    print("layer |  start  | k-loop (min/max over waves) | stage | epilogue | barrier wait (min/max) | layer total")
    for li in range(L):
        s, k, p1, p2, b = (t[li, i] for i in range(5))
        if k.max() == 0:      # narrow layer: no unit stamps
            print(f"  {li}   | {s.min() - t0:7d} | (narrow path)  total {b.max() - s.min():6d}")
            continue
        act = k > 0
        print(f"  {li}   | {s.min() - t0:7d} | {(k - s)[act].min():6d} {(k - s)[act].max():6d} | "
              f"{(p1 - k)[act].mean():6.0f} | {(p2 - p1)[act].mean():6.0f} | {(b - p2).min():6d} {(b - p2).max():6d} | "
              f"{b.max() - s.min():6d}")
    print(f"total {t[L - 1, 4].max() - t0} cycles")
    if os.environ.get("TIMELINE_RAW"):
        li = int(os.environ["TIMELINE_RAW"])
        print(f"layer {li} per wave (relative to layer start): k-loop done, staged, epilogue done, barrier passed")
        for w in range(8):
            print(f"  wave {w}: " + " ".join(f"{t[li, i, w] - t[li, 0].min():7d}" for i in range(5)))


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    P = 65536
    dims = (43, 256, 256, 256, 213, 256, 256, 256, 257)
    Ws, bs = make(dims, 1, 3)
    x = torch.randn(P, 43, device="cuda")
    buf = torch.zeros(10 * 5 * 8, dtype=torch.int64, device="cuda")
    so = lib.load()
    so.ndjir_mlp_debug_timeline.argtypes = [ctypes.c_void_p]
    if mode in ("fwd", "fwd_nostore"):
        keep = mode == "fwd"
        chain_forward(x, Ws, bs, 100.0, 3, 0.7071, keep_hidden=keep)
        torch.cuda.synchronize()
        so.ndjir_mlp_debug_timeline(buf.data_ptr())
        chain_forward(x, Ws, bs, 100.0, 3, 0.7071, keep_hidden=keep)
        torch.cuda.synchronize()
        so.ndjir_mlp_debug_timeline(None)
        show(buf, 8, "geometric net forward chain (43-256-256-256-213|+43-256-256-256-257), tile of 64 points")
    else:
        dims2 = (259, 256, 256, 256, 3) if mode == "bwd" else (39, 128, 128, 128, 1)
        W2, b2 = make(dims2, 2)
        xg = torch.randn(P, dims2[0], device="cuda").requires_grad_(mode == "bwd")
        Wg = [w.clone().requires_grad_(True) for w in W2]
        bg = [b.clone().requires_grad_(True) for b in b2]
        y = fused_mlp(xg, Wg, bg)
        gy = torch.randn_like(y)
        torch.cuda.synchronize()
        so.ndjir_mlp_debug_timeline(buf.data_ptr())
        torch.autograd.grad(y, ([xg] if mode == "bwd" else []) + Wg + bg, gy)
        torch.cuda.synchronize()
        so.ndjir_mlp_debug_timeline(None)
        show(buf, 4 if mode == "bwd" else 3, f"backward chain of net {dims2}")


if __name__ == "__main__":
    main()
