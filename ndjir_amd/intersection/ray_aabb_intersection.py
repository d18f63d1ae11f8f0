"""Ray / axis-aligned-box intersection.

Reference: python/intersection/ray_aabb_intersection.py:26-116 (RayAABBIntersection,
`ray_aabb_intersection(camloc, raydir, min, max)`) and
csrc/intersection/ray_aabb_intersection_cuda.cu:70-162.  No gradient (backward_impl is `pass`).
"""
import torch

from .. import functions as F
from .. import lib


def ray_aabb_intersection(camloc, raydir, min=(-1, -1, -1), max=(1, 1, 1)):
    """camloc (B, 3), raydir (B, R, 3) -> t_near, t_far, n_hits, each (B, R, 1).
    n_hits is a float tensor (0, 1 or 2), as in the reference."""
    assert camloc.dim() == 2 and camloc.shape[-1] == 3
    assert raydir.dim() == 3 and raydir.shape[-1] == 3 and raydir.shape[0] == camloc.shape[0]
    B, R, _ = raydir.shape
    c = camloc.detach().contiguous()
    d = raydir.detach().contiguous()
    t_near = torch.empty((B, R, 1), device=d.device, dtype=torch.float32)
    t_far = torch.empty_like(t_near)
    n_hits = torch.empty_like(t_near)
    lib.call("ray_aabb_intersection", B * R, t_near, t_far, n_hits, c, d, B, R, min, max)
    return t_near, t_far, n_hits


F.ray_aabb_intersection = ray_aabb_intersection
