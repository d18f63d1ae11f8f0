"""Synthetic step inputs (no DTU data exists on the build or GPU boxes).

Construction follows SURVEY.md section 8(d) and the reference tests' style
(python/intersection/test/test_ray_aabb_intersection.py:118-134): cameras on a sphere of radius
2.5, ray targets uniform in [-1.2, 1.2]^3, so most rays cross the unit box and a minority miss.
"""
import numpy as np
import torch


def make_rays(B, R, seed=412, cam_radius=2.5, target_half=1.2, device="cpu", ray_offset=0, total_rays=None):
    """Returns camloc (B,3), raydir (B,R,3), color_gt (B,R,3) as float32 tensors.
    `ray_offset` / `total_rays` select a contiguous slice of a larger per-image ray set so that N
    ranks together see exactly the rays a single rank would see with R = total_rays."""
    rng = np.random.RandomState(seed)
    camloc = rng.randn(B, 3)
    camloc /= np.linalg.norm(camloc, ord=2, axis=-1, keepdims=True)
    camloc *= cam_radius
    Rt = total_rays if total_rays is not None else R
    target = rng.rand(B, Rt, 3) * target_half * 2 - target_half
    color = rng.rand(B, Rt, 3)
    target = target[:, ray_offset:ray_offset + R]
    color = color[:, ray_offset:ray_offset + R]
    raydir = target - camloc.reshape(B, 1, 3)
    raydir /= np.linalg.norm(raydir, ord=2, axis=-1, keepdims=True)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(device)
    return f(camloc), f(raydir), f(color)
