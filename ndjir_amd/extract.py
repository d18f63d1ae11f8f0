"""SDF volume evaluation for mesh extraction.

Reference: `compute_pts_vol` (python/extract_by_mc.py:46-74): G^3 points of a regular lattice through
`geometric_network(...)[0]` in batches of `extraction.batch_size` (50 000), each batch built on the host,
copied to the device, evaluated layer by layer, and its 257-column output copied back.  Here the lattice
points are formed on the device chunk by chunk, the geometric net runs as the fused sdf-only MFMA chain
(grid gather + 8 layers in one launch sequence, only column 0 of the last layer is computed) and the volume
stays on the device until the caller asks for it.  Marching cubes itself is out of scope (skimage on the CPU
in the reference, `extract_by_mc.py:31-43`).

The lattice shards over x-slabs: `slab_range(G, rank, world)`; each rank evaluates its slabs independently
(no collective on the data path), `gather_volume` concatenates them.
"""
import numpy as np
import torch

from .network import geometric_network


def lattice_axes(mins, maxs, grid_size):
    """The reference's coordinates: float64 `np.linspace`, rounded to fp32 (extract_by_mc.py:48-50)."""
    return [np.linspace(mins[a], maxs[a], grid_size).astype(np.float32) for a in range(3)]


def slab_range(grid_size, rank=0, world=1):
    """x-index range [i0, i1) of `rank`: contiguous slabs, sizes differing by at most one."""
    base, rem = divmod(grid_size, world)
    i0 = rank * base + min(rank, rem)
    return i0, i0 + base + (1 if rank < rem else 0)


def compute_vol(mins, maxs, grid_size, conf, device=None, chunk=1 << 20, rank=0, world=1):
    """vol[i - i0, j, k] = sdf(x_i, y_j, z_k) for this rank's x-slabs, a float32 GPU tensor of shape
    (i1 - i0, G, G) -- the layout `compute_pts_vol` returns after its reshape / transpose (:71)."""
    from . import parameter as P
    device = device or P.get_device()
    G = int(grid_size)
    xs, ys, zs = (torch.from_numpy(a).to(device) for a in lattice_axes(mins, maxs, G))
    i0, i1 = slab_range(G, rank, world)
    n = (i1 - i0) * G * G
    vol = torch.empty(n, device=device, dtype=torch.float32)
    with torch.no_grad():
        for p0 in range(0, n, chunk):
            p1 = min(n, p0 + chunk)
            idx = torch.arange(p0, p1, device=device, dtype=torch.int64) + i0 * G * G
            i = torch.div(idx, G * G, rounding_mode="floor")
            jk = idx - i * (G * G)
            j = torch.div(jk, G, rounding_mode="floor")
            pts = torch.stack([xs[i], ys[j], zs[jk - j * G]], dim=1)
            sdf = geometric_network(pts, conf, first_order_only=True, sdf_only=True)[0]
            vol[p0:p1] = sdf.reshape(-1)
    return vol.view(i1 - i0, G, G)


def gather_volume(local, grid_size, world=1):
    """All ranks' slabs -> the full (G, G, G) volume on every rank (slab sizes may differ by one)."""
    if world == 1:
        return local
    import torch.distributed as dist
    G = int(grid_size)
    parts = [torch.empty((slab_range(G, r, world)[1] - slab_range(G, r, world)[0], G, G), device=local.device,
                         dtype=local.dtype) for r in range(world)]
    dist.all_gather(parts, local.contiguous())
    return torch.cat(parts, dim=0)


def compute_pts_vol(mins, maxs, grid_size, conf, **kw):
    """Reference signature (extract_by_mc.py:46): returns (pts (G^3, 3), vol (G, G, G)) as numpy arrays, pts in the
    reference's meshgrid order."""
    x, y, z = lattice_axes(mins, maxs, grid_size)
    X, Y, Z = np.meshgrid(x, y, z)
    pts = np.stack((X.reshape(-1), Y.reshape(-1), Z.reshape(-1)), axis=1)
    vol = compute_vol(mins, maxs, grid_size, conf, **kw)
    return pts, vol.cpu().numpy()
