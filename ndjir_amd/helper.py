"""Ray generation helpers (reference: python/helper.py:44-81).

`generate_raydir_camloc` / `generate_all_pixels` are the reference's host-side numpy functions (float64);
`generate_raydir_camloc_device` is the same computation as one HIP kernel on cameras and pixels that already live
on the GPU -- no per-iteration host work or host->device copies (python/train.py:131-133).
"""
import numpy as np


def rq_decomposition_3x3(M):
    """M (3, 3) = K R with K upper triangular and R a proper rotation, normalised the way OpenCV's `RQDecomp3x3` leaves it
    (what `cv2.decomposeProjectionMatrix` calls): K[0, 0] > 0 and K[1, 1] > 0, obtained by rotating by 180 degrees about one
    axis where needed (two columns of K and two rows of R change sign together, det R stays +1) -- so K[2, 2] carries the
    sign of det M.  With these conditions the factorisation is unique: any RQ algorithm gives OpenCV's result to rounding."""
    M = np.asarray(M, np.float64)
    # RQ from the QR decomposition of the row-reversed transpose:  (J M)^T = Q' R'  =>  M = (J R'^T J) (J Q'^T)
    J = np.eye(3)[::-1]
    q, r = np.linalg.qr((J @ M).T)
    K = J @ r.T @ J
    R = J @ q.T
    if np.linalg.det(R) < 0:                 # make R a rotation (flip one axis; the diagonal signs are fixed below)
        K[:, 2] = -K[:, 2]
        R[2, :] = -R[2, :]
    if K[0, 0] < 0 and K[1, 1] < 0:          # 180 degrees about z
        d = np.diag([-1.0, -1.0, 1.0])
    elif K[0, 0] < 0:                        # 180 degrees about y
        d = np.diag([-1.0, 1.0, -1.0])
    elif K[1, 1] < 0:                        # 180 degrees about x
        d = np.diag([1.0, -1.0, -1.0])
    else:
        d = np.eye(3)
    return K @ d, d @ R


def load_K_Rt_from_P(P):
    """python/helper.py:27-41 without OpenCV: `cv2.decomposeProjectionMatrix(P)` = (K, R, c) with P[:, :3] = K R
    (RQ decomposition, OpenCV's sign conventions: `rq_decomposition_3x3`) and c the homogeneous camera centre (P c = 0);
    then intrinsic = K / K[2, 2] in a 4 x 4 identity, pose = [R^T | c / c_w] (camera-to-world | camera location).
    OpenCV returns arrays of P's dtype (the reference passes float32, python/dataset.py:123-133): K, R and c are rounded to
    it before the reference's own arithmetic is applied, the intrinsic is float64 and the pose float32 as there.
    PARITY UNPINNED: cv2 is not in this image and the reference holds no camera fixtures; tested by construction
    (tests/test_dataset_cpu.py: random K, R, c -> P -> decode)."""
    P = np.asarray(P)
    dt = P.dtype if P.dtype in (np.float32, np.float64) else np.float64
    P64 = P.astype(np.float64)
    K, R = rq_decomposition_3x3(P64[:, :3])
    # camera centre: the right null vector of P, c = (-M^-1 p_4, 1) up to scale (OpenCV takes it from an SVD; the reference
    # divides by c_w, so the scale does not matter)
    c = np.concatenate([-np.linalg.solve(P64[:, :3], P64[:, 3]), [1.0]])
    K, R, c = K.astype(dt), R.astype(dt), c.astype(dt)
    K = K / K[2, 2]
    intrinsic = np.eye(4)
    intrinsic[:3, :3] = K
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.transpose()
    pose[:3, 3] = c[:3] / c[3]
    return intrinsic, pose


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
