"""How many distinct voxel cells does one bench step touch (sizes the sparse gradient exchange)?"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ndjir_amd import config as cfg
from ndjir_amd.distributed import voxel_cell_ids
dev = torch.device("cuda:0")
conf = cfg.load("default")
step = bench.Step(conf, 512, dev, 0, 1)
step.forward_backward()
v = conf.geometric_network.voxel
x = step.x_fg
xp = x + step.rand["noise"] * (math.sqrt(3) * 2 * conf.renderer.bounding_sphere_radius / v.grid_size)
a = voxel_cell_ids(x, [v.grid_size] * 3); b = voxel_cell_ids(xp, [v.grid_size] * 3)
print("points", x.numel() // 3, "corner refs", a.numel() + b.numel(), "unique main", torch.unique(a).numel(), "unique ptb", torch.unique(b).numel(),
      "unique both", torch.unique(torch.cat([a, b])).numel())
buf = next(iter(step.grid_bufs.values()))
print("non-zero rows in the gradient buffer", int((buf.view(-1, 4) != 0).any(dim=1).sum()))
