"""Micro-benchmark of the MLP engine kernels at the bench shapes (used under rocprofv3)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndjir_amd.mlp import chain_forward, fused_mlp, wgrad  # noqa: E402


def make(dims, seed, skip_layer=-1, dev="cuda"):
    rng = np.random.RandomState(seed)
    Ws, bs = [], []
    kin = dims[0]
    for j in range(len(dims) - 1):
        Ws.append(torch.tensor(rng.randn(kin, dims[j + 1]) * np.sqrt(2.0 / kin), dtype=torch.float32, device=dev))
        bs.append(torch.tensor(rng.randn(dims[j + 1]) * 0.1, dtype=torch.float32, device=dev))
        kin = dims[j + 1] + (dims[0] if j == skip_layer else 0)
    return Ws, bs


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    P = 65536
    dims = (43, 256, 256, 256, 213, 256, 256, 256, 257)
    Ws, bs = make(dims, 1, 3)
    x = torch.randn(P, 43, device="cuda")
    flops = 2 * P * (43 * 256 + 256 * 256 * 2 + 256 * 213 + 256 * 256 * 3 + 256 * 257)
    dt = timeit(lambda: chain_forward(x, Ws, bs, 100.0, 3, 0.7071, keep_hidden=True), n)
    print(f"chain fwd geometric (8 layers, skip) P={P}: {dt * 1e6:.0f} us  {flops / dt / 1e12:.1f} TFLOP/s")
    dims2 = (259, 256, 256, 256, 3)
    W2, b2 = make(dims2, 2)
    x2 = torch.randn(P, 259, device="cuda")
    f2 = 2 * P * (259 * 256 + 256 * 256 * 2 + 256 * 3)
    dt = timeit(lambda: chain_forward(x2, W2, b2, 100.0, -1, 1.0, keep_hidden=True), n)
    print(f"chain fwd base-colour (259-256-256-256-3) P={P}: {dt * 1e6:.0f} us  {f2 / dt / 1e12:.1f} TFLOP/s")
    xg = x2.clone().requires_grad_(True)
    Wg = [w.clone().requires_grad_(True) for w in W2]
    bg = [b.clone().requires_grad_(True) for b in b2]
    gy = torch.randn(P, 3, device="cuda")

    def fb():
        y = fused_mlp(xg, Wg, bg)
        torch.autograd.grad(y, [xg] + Wg + bg, gy)
    dt = timeit(fb, n)
    print(f"fused_mlp fwd+bwd base-colour: {dt * 1e6:.0f} us  {3 * f2 / dt / 1e12:.1f} TFLOP/s (3x fwd flops)")
    A = torch.randn(P, 256, device="cuda")
    B = torch.randn(P, 256, device="cuda")
    dt = timeit(lambda: wgrad(A, B), n)
    print(f"wgrad 256x256 P={P}: {dt * 1e6:.0f} us  {2 * P * 256 * 256 / dt / 1e12:.1f} TFLOP/s")
    A = torch.randn(P, 128, device="cuda")
    B = torch.randn(P, 128, device="cuda")
    dt = timeit(lambda: wgrad(A, B), n)
    print(f"wgrad 128x128 P={P}: {dt * 1e6:.0f} us  {2 * P * 128 * 128 / dt / 1e12:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
