"""Times the SDF volume evaluation of mesh extraction (ndjir_amd/extract.py; python/extract_by_mc.py:46-74) at the
reference's size: 512^3 lattice points through the default.yaml geometric network (512^3 x 4 voxel grid)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ndjir_amd import config, network, parameter as P
from ndjir_amd.extract import compute_vol

G = int(sys.argv[1]) if len(sys.argv) > 1 else 512
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dev = torch.device("cuda:0")
conf = config.load("default")
P.clear_parameters(); P.set_device(dev); network.seed(313)
with torch.no_grad():
    network.geometric_network(torch.zeros(4, 3, device=dev), conf)
compute_vol([-1] * 3, [1] * 3, 64, conf, chunk=chunk)        # warm-up (weight packing, allocator)
torch.cuda.synchronize()
t0 = time.perf_counter()
vol = compute_vol([-1] * 3, [1] * 3, G, conf, chunk=chunk)
torch.cuda.synchronize()
el = time.perf_counter() - t0
n = G ** 3
flop = 786944.0          # sdf-only forward, 2 in out per layer, last layer 1 column
print(json.dumps({"lattice": f"{G}^3", "points": n, "seconds": el, "points_per_s": n / el,
                  "algorithmic_tflops": n * flop / el / 1e12, "chunk": chunk,
                  "inside_fraction": float((vol < 0).float().mean())}))
