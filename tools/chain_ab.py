"""A/B of chain launches at the bench shapes by HIP events (mlp.PROFILE): forward + backward of one net, wgrad excluded.
usage: python tools/chain_ab.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndjir_amd import mlp  # noqa: E402
from kernel_bench import make  # noqa: E402


def run(dims, P, reps):
    W, b = make(dims, 2)
    x = torch.randn(P, dims[0], device="cuda").requires_grad_(True)
    Wg = [w.clone().requires_grad_(True) for w in W]
    bg = [t.clone().requires_grad_(True) for t in b]
    res = {}
    for it in range(reps + 2):
        mlp.PROFILE = []
        y = mlp.fused_mlp(x, Wg, bg)
        torch.autograd.grad(y, [x] + Wg + bg, torch.ones_like(y))
        torch.cuda.synchronize()
        if it >= 2:
            for e in mlp.PROFILE:
                res.setdefault((e[0], e[6]), []).append(e[2].elapsed_time(e[3]) * 1e3)
        mlp.PROFILE = None
    for (k, sym), v in res.items():
        if k.startswith("chain"):
            v.sort()
            print(f"  {dims} {k:10s} {sym:36s} median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f}")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    P = 65536
    run((262, 256, 256, 256, 256, 3), P, reps)
    run((262, 128, 128, 128, 3), P, reps)
    run((39, 128, 128, 128, 128, 1), 2 * P, reps)


if __name__ == "__main__":
    main()
