"""Device time of grouped weight-gradient launches at the step's dominant shapes (HIP events over 10 launches):
python tools/wgrad_group_time.py [layers]   (NDJIR_WGRAD_BIG=0: the two-workgroups-per-CU tiles only; NDJIR_WGRAD_ITEMS)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndjir_amd import mlp
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = 65536
SHAPES = ((256, 256), (128, 128), (262, 128), (256, 257)) if not os.environ.get("WGT_ONE") else ((256, 256),)
for K, N in SHAPES:
    jobs = []
    for _ in range(L):
        A = torch.randn(P, K, device="cuda"); B = torch.randn(P, N, device="cuda") * 1e-3
        am = torch.tensor([A.abs().max()], device="cuda"); bm = torch.tensor([B.abs().max()], device="cuda")
        jobs.append((torch.zeros(K, N, device="cuda"), True, [(A, B, am, bm)]))
    for _ in range(2):
        mlp.wgrad_group(jobs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        mlp.wgrad_group(jobs)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100 / L
    print(f"{L} x {K}x{N}x{P}: {us:.1f} us per layer = {4e-6 * P * (K + N) / us:.2f} TB/s of operands, {2e-6 * P * K * N / us:.0f} TFLOP/s")
