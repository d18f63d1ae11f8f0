"""Average device time of the 256 x 256 x 65536 weight-gradient launch (HIP events over 20 launches): python tools/wgrad_time.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndjir_amd.mlp import wgrad
P = 65536
for K, N in ((256, 256), (128, 128)):
    A = torch.randn(P, K, device="cuda"); B = torch.randn(P, N, device="cuda") * 1e-3
    am = torch.tensor([A.abs().max()], device="cuda"); bm = torch.tensor([B.abs().max()], device="cuda")
    out = torch.zeros(K, N, device="cuda")
    for _ in range(3):
        wgrad(A, B, out=out, accum=True, amax_a=am, amax_b=bm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        wgrad(A, B, out=out, accum=True, amax_a=am, amax_b=bm)
    e1.record(); torch.cuda.synchronize()
    print(f"{K}x{N}x{P}: {e0.elapsed_time(e1) * 50:.1f} us per launch (incl. the reduction)")
