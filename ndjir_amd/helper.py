"""Ray generation helpers (reference: python/helper.py:44-81; pure numpy, float64)."""
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
