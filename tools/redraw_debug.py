"""Loss of the fwd+bwd step on successive batches of the synthetic data feed (what bench.py's `redraw` leg replays), eagerly,
with a NaN / inf check per step: python tools/redraw_debug.py [steps]"""
import os, sys, copy
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndjir_amd import config as cfg
from ndjir_amd.step import Step
from ndjir_amd.dataset import IDRRaySource
from ndjir_amd.synthetic import make_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
conf = cfg.load("default", [])
dev = torch.device("cuda:0")
step = Step(conf, 512, dev, 0, 1)
c2 = copy.deepcopy(conf); c2.train.n_rays = 512
images, masks, Ks, poses = make_scene(8, 256, 256, seed=5)
src = IDRRaySource(images, masks, Ks, poses, c2, rng=np.random.RandomState(313), device=dev)
gen = torch.Generator(device=dev).manual_seed(0)
for it in range(n):
    color, _m, raydir, camloc = src.next_batch(1)
    step.set_rays(camloc, raydir, color)
    step.redraw_rand(gen)
    step.forward_backward()
    loss = float(step.loss)
    bad = [nm for nm, g in zip(step.mlp_names, step.grad_views) if not torch.isfinite(g).all()]
    print(it, "loss", loss, "non-finite MLP grads:", bad[:4], "inputs finite:", bool(torch.isfinite(raydir).all() and torch.isfinite(camloc).all() and torch.isfinite(color).all()), flush=True)
