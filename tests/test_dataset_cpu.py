"""f2: the `cameras.npz` decode of python/helper.py:27-41 / python/dataset.py:110-140 without OpenCV.
PARITY UNPINNED (cv2 is absent from the image, the reference holds no camera fixtures): tested by construction --
random K (upper triangular, positive diagonal), R in SO(3), c -> P = K [R | -R c] -> decode recovers them."""
import os

import numpy as np
import pytest


def _random_camera(rng, scale=1.0):
    K = np.array([[rng.uniform(500, 3000), rng.uniform(-5, 5), rng.uniform(200, 900)],
                  [0.0, rng.uniform(500, 3000), rng.uniform(200, 700)],
                  [0.0, 0.0, 1.0]]) * scale
    q, _ = np.linalg.qr(rng.randn(3, 3))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    c = rng.randn(3) * 3.0
    P = K @ np.concatenate([q, (-q @ c)[:, None]], axis=1)
    return K, q, c, P


@pytest.mark.parametrize("seed", range(8))
def test_load_K_Rt_from_P_recovers_the_camera(seed):
    from ndjir_amd.helper import load_K_Rt_from_P, rq_decomposition_3x3
    rng = np.random.RandomState(seed)
    # an overall scale (and sign) of P does not change the camera: K is normalised by K[2, 2]
    K, R, c, P = _random_camera(rng, scale=rng.uniform(0.1, 10.0))
    intrinsic, pose = load_K_Rt_from_P(P)
    assert intrinsic.shape == (4, 4) and intrinsic.dtype == np.float64 and pose.shape == (4, 4) and pose.dtype == np.float32
    np.testing.assert_allclose(intrinsic[:3, :3], K / K[2, 2], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(pose[:3, :3], R.T, atol=1e-6)          # camera-to-world
    np.testing.assert_allclose(pose[:3, 3], c, rtol=1e-6, atol=1e-6)
    assert np.array_equal(intrinsic[3], [0, 0, 0, 1]) and np.array_equal(pose[3], [0, 0, 0, 1])
    # OpenCV's normalisation: positive K[0, 0], K[1, 1], R a proper rotation, M = K R
    Kq, Rq = rq_decomposition_3x3(P[:, :3])
    assert Kq[0, 0] > 0 and Kq[1, 1] > 0 and abs(Kq[1, 0]) + abs(Kq[2, 0]) + abs(Kq[2, 1]) < 1e-9 * abs(Kq).max()
    assert abs(np.linalg.det(Rq) - 1.0) < 1e-9
    np.testing.assert_allclose(Kq @ Rq, P[:, :3], rtol=1e-9, atol=1e-9 * abs(P).max())
    # -P is the same camera (homogeneous): K[2, 2] < 0 after the 180-degree normalisation, divided out by the reference's K / K[2, 2]
    i2, p2 = load_K_Rt_from_P(-P)
    Kn, Rn = rq_decomposition_3x3(-P[:, :3])
    assert Kn[2, 2] < 0 and abs(np.linalg.det(Rn) - 1.0) < 1e-9
    np.testing.assert_allclose(p2[:3, 3], c, rtol=1e-6, atol=1e-6)
    # float32 input (what the reference passes): the same camera to float32 accuracy
    i32, p32 = load_K_Rt_from_P(P.astype(np.float32))
    np.testing.assert_allclose(i32[:3, :3], K / K[2, 2], rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(p32[:3, :3], R.T, atol=1e-4)
    np.testing.assert_allclose(p32[:3, 3], c, rtol=1e-3, atol=1e-3)


def write_idr_scene(path, M=3, H=24, W=32, seed=0, png=True, scale=2.5, trans=(0.1, -0.2, 0.3)):
    """A synthetic scene in the IDR / DTU directory layout; returns the cameras it was made from (in NORMALISED coordinates:
    the scene sits in the unit sphere after x_norm = (x_world - trans) / scale, which is what world_mat @ scale_mat encodes)."""
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(path, "image"))
    os.makedirs(os.path.join(path, "mask"))
    S = np.eye(4)
    S[:3, :3] *= scale
    S[:3, 3] = trans
    cams, mats = [], {}
    images = (rng.rand(M, H, W, 3) * 255).astype(np.uint8)
    masks = (rng.rand(M, H, W) > 0.4).astype(np.uint8) * 255
    for i in range(M):
        K = np.array([[1.5 * W, 0.0, W / 2], [0.0, 1.5 * W, H / 2], [0.0, 0.0, 1.0]])
        q, _ = np.linalg.qr(rng.randn(3, 3))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        c_norm = rng.randn(3)
        c_norm = 2.5 * c_norm / np.linalg.norm(c_norm)
        c_world = scale * c_norm + np.asarray(trans)
        Wm = np.eye(4)
        Wm[:3, :4] = K @ np.concatenate([q, (-q @ c_world)[:, None]], axis=1)
        mats[f"world_mat_{i}"], mats[f"scale_mat_{i}"] = Wm, S
        cams.append((K, q, c_norm))
        if png:
            from PIL import Image
            Image.fromarray(images[i]).save(os.path.join(path, "image", f"{i:06d}.png"))
            Image.fromarray(masks[i]).save(os.path.join(path, "mask", f"{i:03d}.png"))
        else:
            np.save(os.path.join(path, "image", f"{i:06d}.npy"), images[i])
            np.save(os.path.join(path, "mask", f"{i:03d}.npy"), masks[i])
    np.savez(os.path.join(path, "cameras.npz"), **mats)
    return images, masks, cams, S


@pytest.mark.parametrize("png", [True, False])
def test_load_idr_scene(tmp_path, png):
    """world_mat_i @ scale_mat_i with a non-unit scale and a translation: the decoded cameras are those of the NORMALISED scene
    (python/dataset.py:121-137), images / masks as the reference scales and thresholds them."""
    from ndjir_amd.dataset import load_idr_scene
    path = str(tmp_path / "scan")
    images, masks, cams, S = write_idr_scene(path, png=png)
    d = load_idr_scene(path)
    assert d["images"].shape == (3, 24, 32, 3) and d["images"].dtype == np.float32
    np.testing.assert_allclose(d["images"], images / 255.0, atol=1e-7)
    assert d["masks"].shape == (3, 24, 32, 1) and np.array_equal(d["masks"][..., 0], (masks > 127.5) * 1.0)
    assert d["intrinsics"].shape == (3, 3, 3) and d["poses"].shape == (3, 4, 4)
    for i, (K, R, c) in enumerate(cams):
        np.testing.assert_allclose(d["intrinsics"][i], K, rtol=1e-4, atol=1e-3)     # (P is formed in float32, as in the reference)
        np.testing.assert_allclose(d["poses"][i][:3, :3], R.T, atol=1e-5)
        np.testing.assert_allclose(d["poses"][i][:3, 3], c, atol=1e-4)
    assert abs(float(d["scale"]) - 2.5) < 1e-6
    np.testing.assert_allclose(d["trans"], [0.1, -0.2, 0.3], atol=1e-6)
    with pytest.raises(FileNotFoundError):
        load_idr_scene(str(tmp_path / "nothing"))
