"""Forward + backward of the step's net shapes, one at a time, for per-dispatch kernel durations under
`rocprofv3 --kernel-trace` (host overhead does not enter).  usage: python tools/chain_shapes.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndjir_amd.mlp import fused_mlp  # noqa: E402
from kernel_bench import make  # noqa: E402

SHAPES = [("geometric", (43, 256, 256, 256, 213, 256, 256, 256, 257), 65536, 3),
          ("base_colour", (259, 256, 256, 256, 3), 65536, -1),
          ("photogrammetric", (290, 256, 256, 256, 1), 65536, -1),
          ("roughness", (262, 128, 128, 128, 2), 65536, -1),
          ("env_light", (39, 128, 128, 128, 1), 131072, -1),
          ("soft_vis", (39, 128, 128, 128, 1), 131072, -1),
          ("background", (52, 256, 256, 256, 257), 16384, -1)]
for name, dims, P, skip in SHAPES:
    Ws, bs = make(dims, 1, skip)
    Ws = [w.requires_grad_(True) for w in Ws]
    bs = [b.requires_grad_(True) for b in bs]
    x = torch.randn(P, dims[0], device="cuda", requires_grad=True)
    for _ in range(3):
        y = fused_mlp(x, Ws, bs, 100.0, skip, 0.7071 if skip >= 0 else 1.0)
        torch.autograd.grad(y, [x] + Ws + bs, torch.randn_like(y))
        torch.cuda.synchronize()
print("done")
