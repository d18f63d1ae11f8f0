"""Ray generation helpers (reference: python/helper.py:44-81).

`generate_raydir_camloc` / `generate_all_pixels` are the reference's host-side numpy functions (float64);
`generate_raydir_camloc_device` is the same computation as one HIP kernel on cameras and pixels that already live
on the GPU -- no per-iteration host work or host->device copies (python/train.py:131-133).
"""
import numpy as np


def generate_raydir_camloc(pose, intrinsic, xy):
    """pose (B,4,4) camera-to-world, intrinsic (B,3,3), xy (B,R,2) pixel coordinates ->
    raydir (B,R,3) unit vectors, camloc (B,3).  x_w = R_c2w K^-1 (x, y, 1)^T, normalised."""
    B, R, _ = xy.shape
    R_c2w = pose[:, np.newaxis, :3, :3]
    camloc = pose[:, np.newaxis, :3, 3:4]
    K_inv = np.linalg.inv(intrinsic[:, np.newaxis, :, :])
    xyz_pixel = np.concatenate([xy, np.ones([B, R, 1])], axis=-1)[:, :, :, np.newaxis]
    xyz_world = np.matmul(R_c2w, np.matmul(K_inv, xyz_pixel)).reshape((B, R, 3))
    raydir = xyz_world / np.sqrt(np.sum(xyz_world ** 2, axis=-1, keepdims=True))
    return raydir, camloc.reshape((B, 3))


def generate_all_pixels(W, H):
    """All pixel coordinates (H*W, 2) as (x, y), row-major over y then x."""
    xx, yy = np.meshgrid(np.arange(0, W), np.arange(0, H))
    return np.asarray([xx.flatten(), yy.flatten()]).T


def generate_raydir_camloc_device(pose, intrinsic, xy=None, pixel_index=None, width=None):
    """pose (B,4,4) and intrinsic (B,3,3): float64 GPU tensors.  Pixels: xy (B,R,2) float32 GPU tensor, or
    pixel_index (B,R) int32 GPU tensor of flat indices into a `width`-wide image (python/dataset.py:96-101).
    Returns raydir (B,R,3), camloc (B,3) float32 GPU tensors."""
    import torch
    from . import lib
    assert pose.dtype == torch.float64 and intrinsic.dtype == torch.float64, "cameras are kept in float64 like the reference's"
    assert (xy is None) != (pixel_index is None)
    B = pose.shape[0]
    R = xy.shape[1] if xy is not None else pixel_index.shape[1]
    raydir = torch.empty((B, R, 3), device=pose.device, dtype=torch.float32)
    camloc = torch.empty((B, 3), device=pose.device, dtype=torch.float32)
    lib.call("generate_raydir_camloc", B, R, pose.contiguous(), intrinsic.contiguous(),
             pixel_index.contiguous() if pixel_index is not None else None,
             xy.contiguous() if xy is not None else None, int(width or 0), raydir, camloc)
    return raydir, camloc
