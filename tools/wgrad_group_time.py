"""Device time of grouped weight-gradient launches at the step's dominant shapes (HIP events over 10 launches):
python tools/wgrad_group_time.py [layers]   (NDJIR_WGRAD_ITEMS; WGT_BLOCKED=1|2|3: A / B / both operands point-blocked -- the
values are then read in another order, the timing is what counts; WGT_NARROW=1: adds 256 x 3 outputs)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndjir_amd import mlp
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = 65536
SHAPES = ((256, 256), (128, 128), (262, 128), (256, 257)) if not os.environ.get("WGT_ONE") else ((256, 256),)
if os.environ.get("WGT_NARROW"):
    SHAPES = SHAPES + ((256, 3), (128, 1))
BLK = int(os.environ.get("WGT_BLOCKED", "0"))
for K, N in SHAPES:
    jobs = []
    for _ in range(L):
        A = torch.randn(P, K, device="cuda"); B = torch.randn(P, N, device="cuda") * 1e-3
        am = torch.tensor([A.abs().max()], device="cuda"); bm = torch.tensor([B.abs().max()], device="cuda")
        jobs.append((torch.zeros(K, N, device="cuda"), True, [(mlp.pb(A, BLK & 1), mlp.pb(B, bool(BLK & 2) and N > 8), am, bm)]))
    for _ in range(2):
        mlp.wgrad_group(jobs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        mlp.wgrad_group(jobs)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100 / L
    print(f"{L} x {K}x{N}x{P}: {us:.1f} us per layer = {4e-6 * P * (K + N) / us:.2f} TB/s of operands, {2e-6 * P * K * N / us:.0f} TFLOP/s")
