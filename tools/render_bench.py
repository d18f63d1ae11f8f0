"""Forward-only rendering throughput (BASELINE.json cfg5: `render_image.py`, full 1200x1600 frame in tiles of
valid.n_rays = 4000 rays), at 128 samples per ray (default.yaml) and at 256 (renderer.n_samples0=128, n_samples1=32).
usage: python tools/render_bench.py [W H]        (default 1600 1200)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ndjir_amd import config, network, parameter as P
from ndjir_amd.renderer import render_image

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1600, 1200)
dev = torch.device("cuda:0")
pose = np.eye(4)[None].copy(); pose[0, :3, 3] = [0.0, 0.0, -2.5]
f = 1.2 * W
K = np.array([[[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]])
for name, ov in (("N=128 (default.yaml)", []), ("N=256 (renderer.n_samples0=128 n_samples1=32)",
                                                ["renderer.n_samples0=128", "renderer.n_samples1=32"])):
    conf = config.load("default", ["valid.n_rays=4000", "valid.n_down_samples=0"] + ov)
    r = conf.renderer
    N = r.n_samples0 + r.n_samples1 * r.n_upsamples
    P.clear_parameters(); P.set_device(dev); network.seed(313)
    render_image(pose, K, (80, 50), conf, device=dev)           # creates the parameters, warms up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = render_image(pose, K, (W, H), conf, device=dev)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"config": name, "samples_per_ray": N, "frame": f"{W}x{H}", "rays": W * H, "tile_rays": 4000,
                      "seconds": el, "rays_per_s": W * H / el, "mean": float(img.mean()),
                      "full_1600x1200_s": 1600 * 1200 / (W * H / el)}), flush=True)
