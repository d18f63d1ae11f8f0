"""Phase timeline of the grouped weight-gradient's pipelined chunk loop (variant built with -DWGG_TIMELINE):
  tools/build_variant.sh wg_tl wgrad.hip -DWGG_TIMELINE;  NDJIR_HIP_LIB=.../wg_tl.so python tools/wgrad_timeline.py
Stamps of wave 0 of workgroups 0 and 1200, per chunk: 0 top, 1..8 after each (3 MFMAs + side item), 9 after the barrier
(s_memtime), 11 = s_memrealtime (100 MHz) after the barrier, to calibrate the first counter."""
import os, sys
import ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndjir_amd import mlp, lib
L, P, K, N = 8, 65536, 256, 256
BLK = int(os.environ.get("WGT_BLOCKED", "3"))
jobs = []
for _ in range(L):
    A = torch.randn(P, K, device="cuda"); B = torch.randn(P, N, device="cuda") * 1e-3
    am = torch.tensor([A.abs().max()], device="cuda"); bm = torch.tensor([B.abs().max()], device="cuda")
    jobs.append((torch.zeros(K, N, device="cuda"), True, [(mlp.pb(A, BLK & 1), mlp.pb(B, BLK & 2), am, bm)]))
for _ in range(3):
    mlp.wgrad_group(jobs)
torch.cuda.synchronize()
h = lib.load()
buf = np.zeros((2, 64, 12), dtype=np.int64)
rc = h.ndjir_debug_wgrad_stamps(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
for w in range(2):
    st = buf[w]
    n = int((st[:, 0] > 0).sum())
    tick_ns = (st[n - 1, 11] - st[0, 11]) * 10.0 / max(st[n - 1, 9] - st[0, 9], 1)
    print(f"workgroup {'0' if w == 0 else '1200'}: {n} chunks; one s_memtime tick = {tick_ns:.3f} ns; "
          f"item = {(st[n - 1, 11] - st[0, 11]) * 0.01:.1f} us")
    d = np.diff(st[:n, :10], axis=1)
    names = ["frags+side0", "side1", "side2", "side3", "side4", "side5", "side6", "side7", "barrier"]
    print("  mean ticks per phase: " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, d[2:n - 1].mean(axis=0))))
    print("  chunk period (mean): %.0f ticks = %.2f us" % (np.diff(st[:n, 0]).mean(), np.diff(st[:n, 0]).mean() * tick_ns * 1e-3))
    for c in (0, 1, 2, 3, n // 2, n // 2 + 1):
        print(f"  chunk {c}: " + " ".join(f"{v}" for v in d[c]))
