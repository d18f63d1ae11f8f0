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


def make_scene(M, H, W, seed=3, cam_radius=2.5, focal=1.5):
    """A synthetic multi-view scene for the device data feed (ndjir_amd/dataset.py `IDRRaySource`): M cameras on a sphere of
    radius `cam_radius` looking at the origin, pinhole intrinsics with focal length `focal` x W, random colours, full masks.
    Returns images (M,H,W,3) float32, masks (M,H,W,1), intrinsics (M,3,3), poses (M,4,4) camera-to-world."""
    rng = np.random.RandomState(seed)
    images = rng.rand(M, H, W, 3).astype(np.float32)
    masks = np.ones((M, H, W, 1))
    poses = np.zeros((M, 4, 4))
    Ks = np.zeros((M, 3, 3))
    for m in range(M):
        c = rng.randn(3)
        c = cam_radius * c / np.linalg.norm(c)
        fwd = -c / np.linalg.norm(c)
        right = np.cross(fwd, rng.randn(3))
        right /= np.linalg.norm(right)
        poses[m, :3, 0], poses[m, :3, 1], poses[m, :3, 2], poses[m, :3, 3] = right, np.cross(fwd, right), fwd, c
        poses[m, 3, 3] = 1
        Ks[m] = [[focal * W, 0, W / 2], [0, focal * W, H / 2], [0, 0, 1]]
    return images, masks, Ks, poses
