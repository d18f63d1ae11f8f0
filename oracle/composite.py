"""Torch restatement of the reference's composite grid ops (python/grid_feature/*_composite.py).

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Each function restates one composite
graph of the reference -- the reference's own test oracle for its CUDA ops -- with stock
torch ops, so torch autograd gives derivatives of any order in fp32 or fp64.  Used to
(a) cross-validate the C restatement of the CUDA kernels (oracle/csrc/ndjir_oracle.c) and
(b) provide the differentiable grid query of the CPU reference graph (oracle/graph.py).

Differences from the CUDA kernels that the reference tests never exercise (queries are
sampled inside [min, max)): the composites do not clamp the lower cell; `clamp=True`
(default) applies the kernels' clamping (voxel_feature_cuda.cu:57-60) so that both
restatements also agree for out-of-box queries.
"""
import math

import torch


def _cells(query, G, min_, max_, clamp=True):
    """voxel_feature_composite.py:24-30 (+ kernel clamping). Returns pointf, point0, point1."""
    scale = (G - 1) / (max_ - min_)
    pointf = (query - min_) * scale
    point0 = torch.floor(pointf).detach()
    if clamp:
        point0 = point0.clamp(0, G - 1)
        point1 = (point0 + 1).clamp(max=G - 1)
    else:
        point1 = point0 + 1
    return pointf, point0, point1


def _coeffs(pointf, point0, point1, kind):
    if kind == "linear":      # voxel_feature_composite.py:28-29
        c0 = point1 - pointf
        # composite: coeff1 = pointf - point0; where the upper cell is clamped the kernels'
        # `1 - c0` (voxel_feature_cuda.cu:64) is the only meaningful definition
        c1 = torch.where(point1 == point0 + 1, pointf - point0, 1 - c0)
        return c0, c1
    if kind == "cosine":      # cosine_voxel_feature_composite.py:29-30
        c0 = 0.5 * torch.cos(math.pi * (pointf - point0)) + 0.5
        return c0, 1 - c0
    raise ValueError(kind)


def query_on_voxel(query, feature, min_=-1.0, max_=1.0, kind="linear", clamp=True):
    """query (P,3), feature (G,G,G,D) -> (P,D). voxel_feature_composite.py:18-64."""
    G = feature.shape[0]
    D = feature.shape[-1]
    pf, p0, p1 = _cells(query, G, min_, max_, clamp)
    c0, c1 = _coeffs(pf, p0, p1, kind)
    i0, i1 = p0.long(), p1.long()
    flat = feature.reshape(-1, D)

    def at(ix, iy, iz):
        return (ix * G + iy) * G + iz

    x0, y0, z0 = i0[:, 0], i0[:, 1], i0[:, 2]
    x1, y1, z1 = i1[:, 0], i1[:, 1], i1[:, 2]
    idx = torch.stack([at(x0, y0, z0), at(x0, y0, z1), at(x0, y1, z0), at(x0, y1, z1),
                       at(x1, y0, z0), at(x1, y0, z1), at(x1, y1, z0), at(x1, y1, z1)], dim=1)
    f = flat[idx]  # (P, 8, D): one gather -> one scatter in backward
    p0_, q0, r0 = c0[:, 0:1], c0[:, 1:2], c0[:, 2:3]
    p1_, q1, r1 = c1[:, 0:1], c1[:, 1:2], c1[:, 2:3]
    return ((p0_ * q0 * r0) * f[:, 0] + (p0_ * q0 * r1) * f[:, 1] + (p0_ * q1 * r0) * f[:, 2]
            + (p0_ * q1 * r1) * f[:, 3] + (p1_ * q0 * r0) * f[:, 4] + (p1_ * q0 * r1) * f[:, 5]
            + (p1_ * q1 * r0) * f[:, 6] + (p1_ * q1 * r1) * f[:, 7])


def query_on_triplane(query, feature, min_=-1.0, max_=1.0, kind="linear", clamp=True):
    """query (P,3), feature (3,G,G,D) -> (P, D*3), channel = d*3 + plane.
    triplane_feature_composite.py:18-78 (planes xy, yz, zx)."""
    P = query.shape[0]
    G = feature.shape[1]
    D = feature.shape[-1]
    pf, p0, p1 = _cells(query, G, min_, max_, clamp)
    c0, c1 = _coeffs(pf, p0, p1, kind)
    i0, i1 = p0.long(), p1.long()
    outs = []
    for pl, (u, v) in enumerate([(0, 1), (1, 2), (2, 0)]):
        fp = feature[pl]
        f00 = fp[i0[:, u], i0[:, v]]
        f01 = fp[i0[:, u], i1[:, v]]
        f10 = fp[i1[:, u], i0[:, v]]
        f11 = fp[i1[:, u], i1[:, v]]
        u0, u1, v0, v1 = c0[:, u:u + 1], c1[:, u:u + 1], c0[:, v:v + 1], c1[:, v:v + 1]
        outs.append((u0 * v0 * f00 + u0 * v1 * f01 + u1 * v0 * f10 + u1 * v1 * f11).reshape(P, D, 1))
    return torch.cat(outs, dim=-1).reshape(P, D * 3)


def query_on_triline(query, feature, min_=-1.0, max_=1.0, kind="linear", clamp=True):
    """query (P,3), feature (3,G,D) -> (P, D*3), channel = d*3 + line.
    triline_feature_composite.py."""
    P = query.shape[0]
    G = feature.shape[1]
    D = feature.shape[-1]
    pf, p0, p1 = _cells(query, G, min_, max_, clamp)
    c0, c1 = _coeffs(pf, p0, p1, kind)
    i0, i1 = p0.long(), p1.long()
    outs = []
    for ln in range(3):
        fl = feature[ln]
        f = c0[:, ln:ln + 1] * fl[i0[:, ln]] + c1[:, ln:ln + 1] * fl[i1[:, ln]]
        outs.append(f.reshape(P, D, 1))
    return torch.cat(outs, dim=-1).reshape(P, D * 3)


# ---- Lanczos (a = 2): lanczos_voxel_feature_composite.py:18-60 -------------------------------
def _sinc(x):
    # F.sinc(x) = sin(x)/x with sinc(0) = 1 (un-normalised)
    safe = torch.where(x == 0, torch.ones_like(x), x)
    return torch.where(x == 0, torch.ones_like(x), torch.sin(safe) / safe)


def lanczos(x, a):
    z = math.pi * x
    return _sinc(z) * _sinc(z / a)


def _lanczos_taps(pointf_axis, G, w):
    x0 = torch.floor(pointf_axis).detach()
    taps = []
    for i in range(-w + 1, w + 1):
        xi = (x0 + i).clamp(0, G - 1)
        taps.append((xi.long(), lanczos(pointf_axis - xi, w).reshape(-1, 1)))
    return taps


def lanczos_query_on_voxel(query, feature, min_=-1.0, max_=1.0, window_size=2):
    G = feature.shape[0]
    scale = (G - 1) / (max_ - min_)
    pf = (query - min_) * scale
    tx, ty, tz = (_lanczos_taps(pf[:, k], G, window_size) for k in range(3))
    f = 0
    for xi, cx in tx:
        for yj, cy in ty:
            for zk, cz in tz:
                f = f + feature[xi, yj, zk] * (cx * cy * cz)
    return f


def lanczos_query_on_triplane(query, feature, min_=-1.0, max_=1.0, window_size=2):
    P = query.shape[0]
    G = feature.shape[1]
    D = feature.shape[-1]
    scale = (G - 1) / (max_ - min_)
    pf = (query - min_) * scale
    t = [_lanczos_taps(pf[:, k], G, window_size) for k in range(3)]
    outs = []
    for pl, (u, v) in enumerate([(0, 1), (1, 2), (2, 0)]):
        f = 0
        for ui, cu in t[u]:
            for vj, cv in t[v]:
                f = f + feature[pl][ui, vj] * (cu * cv)
        outs.append(f.reshape(P, D, 1))
    return torch.cat(outs, dim=-1).reshape(P, D * 3)


def lanczos_query_on_triline(query, feature, min_=-1.0, max_=1.0, window_size=2):
    P = query.shape[0]
    G = feature.shape[1]
    D = feature.shape[-1]
    scale = (G - 1) / (max_ - min_)
    pf = (query - min_) * scale
    outs = []
    for ln in range(3):
        f = 0
        for ui, cu in _lanczos_taps(pf[:, ln], G, window_size):
            f = f + feature[ln][ui] * cu
        outs.append(f.reshape(P, D, 1))
    return torch.cat(outs, dim=-1).reshape(P, D * 3)


# ---- multi-resolution hash grid: voxel_hash_feature_composite.py:93-175 -----------------------
def force_align(size, mod=8):            # voxel_hash_feature.py:26-28 (sic: not a round-up)
    return size + size % mod


def compute_grid_size(G0, growth_factor, level):   # voxel_hash_feature.py:30-33
    import numpy as np
    return int(np.floor(G0 * np.float64(np.float32(growth_factor)) ** level))


def compute_table_size(G, T0):           # voxel_hash_feature.py:35-38
    import numpy as np
    Gf = np.float32(G)
    return int(min(int(min(Gf * Gf * Gf, np.float32(T0))), int(T0)))


def compute_params_boundary(G0, growth_factor, T0, D, level):   # voxel_hash_feature.py:52-62: padding included
    n = 0
    for l in range(level):
        n += force_align(compute_table_size(compute_grid_size(G0, growth_factor, l), T0) * D)
    T = compute_table_size(compute_grid_size(G0, growth_factor, level), T0)
    return n, n + force_align(T * D)


def compute_num_params(G0, growth_factor, T0, D, L):
    n = 0
    for l in range(L):
        n += force_align(compute_table_size(compute_grid_size(G0, growth_factor, l), T0) * D)
    return n


def _hash(x, y, z, T):
    m = 0xFFFFFFFF
    r = ((x * 1) & m) ^ ((y * 2654435761) & m) ^ ((z * 805459861) & m)
    return r % T


def query_on_voxel_hash(query, feature, G0, growth_factor, T0, L, D, min_=-1.0, max_=1.0,
                        kind="linear"):
    """query (P,3), feature (n_params,) -> (P, D*L), channel = d*L + l."""
    P = query.shape[0]
    feats = []
    for l in range(L):
        G = compute_grid_size(G0, growth_factor, l)
        T = compute_table_size(G, T0)
        n0, n1 = compute_params_boundary(G0, growth_factor, T0, D, l)
        fl = feature[n0:n0 + T * D].reshape(T, D)
        if kind == "linear":
            pf, p0, p1 = _cells(query, G, min_, max_, True)
            c0, c1 = _coeffs(pf, p0, p1, "linear")
            i0, i1 = p0.long(), p1.long()
            f = 0
            for a, (ix, cx) in enumerate([(i0[:, 0], c0[:, 0:1]), (i1[:, 0], c1[:, 0:1])]):
                for b, (iy, cy) in enumerate([(i0[:, 1], c0[:, 1:2]), (i1[:, 1], c1[:, 1:2])]):
                    for c, (iz, cz) in enumerate([(i0[:, 2], c0[:, 2:3]), (i1[:, 2], c1[:, 2:3])]):
                        f = f + (cx * cy * cz) * fl[_hash(ix, iy, iz, T)]
        else:  # lanczos_voxel_hash_feature_composite.py
            scale = (G - 1) / (max_ - min_)
            pf = (query - min_) * scale
            tx, ty, tz = (_lanczos_taps(pf[:, k], G, 2) for k in range(3))
            f = 0
            for xi, cx in tx:
                for yj, cy in ty:
                    for zk, cz in tz:
                        f = f + fl[_hash(xi, yj, zk, T)] * (cx * cy * cz)
        feats.append(f)
    feats = torch.stack(feats, dim=1)           # (P, L, D)
    return feats.transpose(1, 2).reshape(P, D * L)


# ---- sampled TV loss: total_variation_loss*_composite.py --------------------------------------
class _SqrtTV(torch.autograd.Function):
    """sqrt whose backward is g * 0.5 * rsqrt(s + 1e-12): with s = sum(delta^2) this gives the
    kernels' d/d(delta) = g * delta * rsqrt(s + 1e-12) (total_variation_loss_cuda.cu:158-163)."""

    @staticmethod
    def forward(ctx, s):
        ctx.save_for_backward(s)
        return torch.sqrt(s)

    @staticmethod
    def backward(ctx, g):
        (s,) = ctx.saved_tensors
        return g * 0.5 / torch.sqrt(s + 1e-12)


def _tv_finish(deltas):
    s = 0
    for d in deltas:
        s = s + d * d
    return _SqrtTV.apply(s)


def tv_loss_on_voxel(query, feature, min_=-1.0, max_=1.0, sym_backward=False):
    """total_variation_loss_composite.py:18-50 -> (P, D)."""
    G = feature.shape[0]
    pf, p0, p1 = _cells(query, G, min_, max_, True)
    i0, i1 = p0.long(), p1.long()
    f000 = feature[i0[:, 0], i0[:, 1], i0[:, 2]]
    if not sym_backward:
        f000 = f000.detach()
    f001 = feature[i0[:, 0], i0[:, 1], i1[:, 2]]
    f010 = feature[i0[:, 0], i1[:, 1], i0[:, 2]]
    f100 = feature[i1[:, 0], i0[:, 1], i0[:, 2]]
    return _tv_finish([f100 - f000, f010 - f000, f001 - f000])


def tv_loss_on_triplane(query, feature, min_=-1.0, max_=1.0, sym_backward=False):
    P = query.shape[0]
    G = feature.shape[1]
    D = feature.shape[-1]
    pf, p0, p1 = _cells(query, G, min_, max_, True)
    i0, i1 = p0.long(), p1.long()
    outs = []
    for pl, (u, v) in enumerate([(0, 1), (1, 2), (2, 0)]):
        fp = feature[pl]
        f00 = fp[i0[:, u], i0[:, v]]
        if not sym_backward:
            f00 = f00.detach()
        outs.append(_tv_finish([fp[i1[:, u], i0[:, v]] - f00, fp[i0[:, u], i1[:, v]] - f00]).reshape(P, D, 1))
    return torch.cat(outs, dim=-1).reshape(P, D * 3)


def tv_loss_on_triline(query, feature, min_=-1.0, max_=1.0, sym_backward=False):
    P = query.shape[0]
    G = feature.shape[1]
    D = feature.shape[-1]
    pf, p0, p1 = _cells(query, G, min_, max_, True)
    i0, i1 = p0.long(), p1.long()
    outs = []
    for ln in range(3):
        f0 = feature[ln][i0[:, ln]]
        if not sym_backward:
            f0 = f0.detach()
        outs.append(_tv_finish([feature[ln][i1[:, ln]] - f0]).reshape(P, D, 1))
    return torch.cat(outs, dim=-1).reshape(P, D * 3)


def tv_loss_on_voxel_hash(query, feature, G0, growth_factor, T0, L, D, min_=-1.0, max_=1.0,
                          sym_backward=False):
    P = query.shape[0]
    feats = []
    for l in range(L):
        G = compute_grid_size(G0, growth_factor, l)
        T = compute_table_size(G, T0)
        n0, n1 = compute_params_boundary(G0, growth_factor, T0, D, l)
        fl = feature[n0:n0 + T * D].reshape(T, D)
        pf, p0, p1 = _cells(query, G, min_, max_, True)
        i0, i1 = p0.long(), p1.long()
        f000 = fl[_hash(i0[:, 0], i0[:, 1], i0[:, 2], T)]
        if not sym_backward:
            f000 = f000.detach()
        f001 = fl[_hash(i0[:, 0], i0[:, 1], i1[:, 2], T)]
        f010 = fl[_hash(i0[:, 0], i1[:, 1], i0[:, 2], T)]
        f100 = fl[_hash(i1[:, 0], i0[:, 1], i0[:, 2], T)]
        feats.append(_tv_finish([f100 - f000, f010 - f000, f001 - f000]))
    feats = torch.stack(feats, dim=1)
    return feats.transpose(1, 2).reshape(P, D * L)
