"""Training data feed with the pixel payloads resident on the GPU (second half).

Reference: `IDRDataSource` (python/dataset.py:26-189): all images / masks / cameras of a scene in host memory; per
iteration `_get_data` picks `n_rays` pixels of one image -- uniformly (indices drawn per epoch in `reset`, :180-189), as a
random 2^a x 2^b patch (`_generate_patch_rays`, :58-84) or with a fixed share inside the object mask
(`_generate_mask_rays`, :86-108) -- and `train.py:124-133` copies colours, mask, and the host-computed rays to the device.
Here the scene lives on the device once; an iteration draws the pixel INDICES on the host with the reference's own
sequence of numpy RNG calls (so the same pixels are read), uploads those few KB, gathers colours and mask on the device
and generates the rays there (`ndjir_generate_raydir_camloc`).  `load_idr_scene` reads a scene directory in the IDR / DTU
layout (`_load_data`, :110-140: image/*, mask/*, cameras.npz with world_mat_i / scale_mat_i) -- the camera decomposition
without OpenCV (`helper.load_K_Rt_from_P`), the images through PIL.
"""
import glob
import os

import numpy as np
import torch


def _read_image(path, gray=False):
    if path.endswith(".npy"):
        return np.load(path)
    try:
        from PIL import Image
    except ImportError as e:      # (the reference reads through imageio / nnabla's image_utils)
        raise ImportError(f"reading {path} needs PIL; store the scene's images as .npy arrays otherwise") from e
    with Image.open(path) as im:
        return np.asarray(im.convert("F")) if gray else np.asarray(im.convert("RGB"))


def load_idr_scene(path):
    """python/dataset.py:110-140 (`IDRDataSource._load_data`): every image / mask / camera of a scene directory

        <path>/image/*      RGB images (sorted by name; .png / .jpg through PIL, or .npy (H, W, 3) uint8)
        <path>/mask/*       object masks, thresholded at 127.5 of their grey value
        <path>/cameras.npz  world_mat_<i>, scale_mat_<i> (4 x 4) per image

    -> dict(images (M,H,W,3) float32 in [0,1], masks (M,H,W,1) float64 in {0,1}, intrinsics (M,3,3), poses (M,4,4) float32
    camera-to-world, scale, trans).  P_i = (world_mat_i @ scale_mat_i)[:3, :4] in float32 as in the reference, decomposed by
    `helper.load_K_Rt_from_P`; `scale` / `trans` are those of the LAST scale matrix (the reference's loop variable)."""
    from .helper import load_K_Rt_from_P
    image_files = sorted(glob.glob(os.path.join(path, "image", "*")))
    mask_files = sorted(glob.glob(os.path.join(path, "mask", "*")))
    if not image_files:
        raise FileNotFoundError(f"no images under {os.path.join(path, 'image')}")
    images = np.asarray([_read_image(f) for f in image_files]) / 255.0
    masks = np.asarray([np.asarray(_read_image(f, gray=True), np.float64).reshape(images.shape[1], images.shape[2], -1)[:, :, :1] > 127.5
                        for f in mask_files]) * 1.0
    cameras = np.load(os.path.join(path, "cameras.npz"))
    intrinsics, poses = [], []
    S = np.eye(4, dtype=np.float32)
    for idx in range(len(images)):
        W = cameras["world_mat_%d" % idx].astype(np.float32)
        S = cameras["scale_mat_%d" % idx].astype(np.float32)
        P = (W @ S)[:3, :4]
        intrinsic, pose = load_K_Rt_from_P(P)
        intrinsics.append(intrinsic[:3, :3])
        poses.append(pose)
    return dict(images=images.astype(np.float32), masks=masks, intrinsics=np.asarray(intrinsics), poses=np.asarray(poses),
                scale=S[0, 0], trans=S[:3, 3])


class IDRRaySource:
    def __init__(self, images, masks, intrinsics, poses, conf, shuffle=False, rng=None, device=None):
        """images (M,H,W,3) float in [0,1]; masks (M,H,W,1) in {0,1}; intrinsics (M,3,3); poses (M,4,4) camera-to-world."""
        from . import parameter as P
        self.conf = conf
        self.device = torch.device(device) if device is not None else P.get_device()
        M, H, W, _ = images.shape
        self._size, self._H, self._W, self._pixels = M, H, W, H * W
        self._n_rays = conf.train.n_rays
        self._shuffle = shuffle
        self.rng = rng if rng is not None else np.random.RandomState(313)          # dataset.py:171-172
        self._masks_host = np.asarray(masks, np.float64).reshape(M, H * W)            # mask-ratio mode thresholds on the host
        self.images = torch.from_numpy(np.ascontiguousarray(images, np.float32).reshape(M, H * W, 3)).to(self.device)
        self.masks = torch.from_numpy(np.ascontiguousarray(masks, np.float32).reshape(M, H * W, 1)).to(self.device)
        self.intrinsics = torch.from_numpy(np.ascontiguousarray(intrinsics, np.float64)).to(self.device)
        self.poses = torch.from_numpy(np.ascontiguousarray(poses, np.float64)).to(self.device)
        self._position = 0
        self.reset()

    @classmethod
    def from_path(cls, path, conf, shuffle=False, rng=None, device=None):
        """The scene directory `path` (IDR / DTU layout, `load_idr_scene`) on the device; `scale` / `trans` as attributes
        (python/dataset.py:135-136: read by the mesh extraction to map back to the scan's coordinates)."""
        d = load_idr_scene(path)
        src = cls(d["images"], d["masks"], d["intrinsics"], d["poses"], conf, shuffle=shuffle, rng=rng, device=device)
        src.scale, src.trans = d["scale"], d["trans"]
        return src

    @property
    def size(self):
        return self._size

    def reset(self):
        """dataset.py:180-189: image order and the epoch's uniform pixel indices (drawn whether or not they are used)."""
        self._img_indices = self.rng.permutation(self._size) if self._shuffle else np.arange(self._size)
        self._pixel_idx = self.rng.randint(0, self._pixels, (self._size, self.conf.train.n_rays))
        self._position = 0

    # -- which pixels (host, the reference's RNG call sequence) --------------------------------------------------------
    def pixel_indices(self, position):
        """(image index, flat pixel indices y * W + x of the position's rays) -- dataset.py:33-108."""
        img = int(self._img_indices[position])
        tr = self.conf.train
        W, H = self._W, self._H
        if tr.patch_ray_sampling:
            n = int(np.log2(tr.n_rays))
            if self.rng.randint(0, 2):                       # height gets the smaller exponent
                nH = n // 2
                nW = n - nH
            else:
                nW = n // 2
                nH = n - nW
            pH, pW = 2 ** nH, 2 ** nW
            H0 = self.rng.randint(0, H - pH)
            W0 = self.rng.randint(0, W - pW)
            # the reference enumerates the patch column by column: (x, y) with y fastest
            xs = np.repeat(np.arange(W0, W0 + pW), pH)
            ys = np.tile(np.arange(H0, H0 + pH), pW)
            return img, ys * W + xs
        if tr.mask_ray_sample_ratio > 0:
            n_in = int(tr.mask_ray_sample_ratio * tr.n_rays)
            m = self._masks_host[img]
            inside = np.where(m >= 0.5)[0]
            inside = inside[self.rng.randint(0, len(inside), n_in)]
            outside = np.where(m < 0.5)[0]
            outside = outside[self.rng.randint(0, len(outside), tr.n_rays - n_in)]
            return img, np.concatenate([inside, outside], axis=0)
        return img, self._pixel_idx[img]

    # -- one batch on the device ------------------------------------------------------------------------------------------
    def next_batch(self, batch_size):
        """color_gt (B,R,3), obj_mask (B,R,1), raydir (B,R,3), camloc (B,3) for the next `batch_size` positions; the epoch
        wraps with `reset()` like nnabla's data iterator."""
        from .helper import generate_raydir_camloc_device
        imgs, idxs = [], []
        for _ in range(batch_size):
            if self._position >= self._size:
                self.reset()
            img, idx = self.pixel_indices(self._position)
            self._position += 1
            imgs.append(img)
            idxs.append(np.asarray(idx, np.int64))
        idx = torch.from_numpy(np.stack(idxs)).to(self.device)                       # the iteration's only host->device copy
        sel = torch.as_tensor(imgs, device=self.device, dtype=torch.int64)
        color = torch.gather(self.images.index_select(0, sel), 1, idx.unsqueeze(-1).expand(-1, -1, 3))
        mask = torch.gather(self.masks.index_select(0, sel), 1, idx.unsqueeze(-1))
        if self.device.type == "cuda":
            raydir, camloc = generate_raydir_camloc_device(self.poses.index_select(0, sel), self.intrinsics.index_select(0, sel),
                                                           pixel_index=idx.to(torch.int32), width=self._W)
        else:       # host-side use (tests of the sampling logic): the reference's numpy function
            from .helper import generate_raydir_camloc
            xy = np.stack([np.stack([i % self._W, i // self._W], axis=-1) for i in idxs])
            rd, cl = generate_raydir_camloc(self.poses[sel].numpy(), self.intrinsics[sel].numpy(), xy)
            raydir, camloc = torch.from_numpy(rd.astype(np.float32)), torch.from_numpy(cl.astype(np.float32))
        return color, mask, raydir, camloc
