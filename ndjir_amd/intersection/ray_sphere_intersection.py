"""Ray / origin-centred-sphere intersection.

Reference: python/intersection/ray_sphere_intersection.py:26-112 and
csrc/intersection/ray_sphere_intersection_cuda.cu:26-96.  No gradient.
"""
import numpy as np
import torch

from .. import functions as F
from .. import lib


def ray_sphere_intersection(camloc, raydir, radius=1.0):
    """camloc (B, 3), raydir (B, R, 3) -> t_near (clamped >= 0), t_far, n_hits, each (B, R, 1)."""
    assert camloc.dim() == 2 and camloc.shape[-1] == 3
    assert raydir.dim() == 3 and raydir.shape[-1] == 3 and raydir.shape[0] == camloc.shape[0]
    B, R, _ = raydir.shape
    c = camloc.detach().contiguous()
    d = raydir.detach().contiguous()
    t_near = torch.empty((B, R, 1), device=d.device, dtype=torch.float32)
    t_far = torch.empty_like(t_near)
    n_hits = torch.empty_like(t_near)
    lib.call("ray_sphere_intersection", B * R, t_near, t_far, n_hits, c, d, B, R, float(radius))
    return t_near, t_far, n_hits


def sample_inside_sphere(B, R, radius=1, rng=None):
    """Uniform points inside a sphere, (B, R, 3); test-input helper of the reference module
    (ray_sphere_intersection.py:115-131)."""
    rng = np.random.RandomState(412) if rng is None else rng
    phi = rng.rand(B, R) * 2 * np.pi
    cos_t = rng.rand(B, R) * 2 - 1
    r = radius * rng.rand(B, R) ** (1.0 / 3.0)
    sin_t = np.sqrt(1 - cos_t ** 2)
    return np.stack([r * sin_t * np.cos(phi), r * sin_t * np.sin(phi), r * cos_t], axis=-1)


F.ray_sphere_intersection = ray_sphere_intersection
