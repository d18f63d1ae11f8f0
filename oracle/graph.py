"""Torch (CPU) restatement of the reference's Python graph for the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  PARITY UNPINNED: the reference holds no
test, fixture or golden value for the sampler, the networks, the BRDFs, the renderer or the
loss, and nnabla cannot be imported here; every function follows the cited reference lines
with nnabla op semantics taken from nnabla's public documentation:
  PF.affine: x @ W + b with W (in, out);  F.softplus(x, beta) = log(1 + exp(beta x)) / beta;
  F.cumprod(exclusive=True) = [1, a0, a0 a1, ...];  F.searchsorted(right=False) = first i with
  seq[i] >= v;  F.gather(batch_dims=2) = per-ray gather;  F.sort ascending;
  nn.grad = reverse-mode graph that is itself differentiable.

Works in fp32 or fp64 (dtype follows `params`).  Parameters are a flat dict keyed by the
reference's nnabla parameter-scope names ("geometric-network/affine-00/affine/W", ...).
Random tensors are explicit inputs (`rand` dict) -- never regenerated.
"""
import math

import torch
import torch.nn.functional as TF

from . import composite as C


# ----------------------------------------------------------------------------------------------
# network.py
# ----------------------------------------------------------------------------------------------
def affine(params, scope, h):
    """network.py:88-93 (PF.affine, base_axis = last)."""
    return h @ params[scope + "/affine/W"] + params[scope + "/affine/b"]


def softplus100(x):
    return TF.softplus(x, beta=100)


def positional_encoding(x, M=6, include_input=True):
    """network.py:96-117: [x, cos(x_i 2^k), sin(x_i 2^k)], band index fastest."""
    bands = (2.0 ** torch.arange(0, M, dtype=x.dtype, device=x.device))
    b = bands.reshape((1,) * x.dim() + (M,)) * x.unsqueeze(-1)
    b = b.reshape(x.shape[:-1] + (-1,))
    g = [x, torch.cos(b), torch.sin(b)] if include_input else [torch.cos(b), torch.sin(b)]
    return torch.cat(g, dim=-1)


def query_on_grid(x, params, conf, scope="geometric-network"):
    """network.py:120-151.  x (..., 3) -> (..., C) or None.
    `voxel.use_ste` (config/ste.yaml:22): the registered `nn.grad` backward of every grid op returns (None, None)
    (grid_feature/voxel_feature.py:383-399 and the same lines of the other families), i.e. d(sdf)/dx does not flow
    through the grid lookup; `.backward()` (backward_impl, :108-116) still reaches the grid parameter.  The query
    point is no parameter, so detaching it for the lookup is exactly that."""
    v = conf.geometric_network.voxel
    typ = v.type
    if typ == "none":
        return None
    q = x.reshape(-1, 3)
    if v.use_ste:
        q = q.detach()

    def one(kind_topo):
        if kind_topo.startswith("lanczos_"):
            topo = kind_topo[len("lanczos_"):]
            fn = {"voxel": C.lanczos_query_on_voxel, "triplane": C.lanczos_query_on_triplane,
                  "triline": C.lanczos_query_on_triline}[topo]
            return fn(q, params[f"{scope}/{topo}_feature/F"])
        kind = "cosine" if kind_topo.startswith("cosine_") else "linear"
        topo = kind_topo[len("cosine_"):] if kind == "cosine" else kind_topo
        fn = {"voxel": C.query_on_voxel, "triplane": C.query_on_triplane,
              "triline": C.query_on_triline}[topo]
        return fn(q, params[f"{scope}/{topo}_feature/F"], kind=kind)

    if typ.endswith("triplaneline"):
        pre = typ[:-len("triplaneline")]
        out = torch.cat([one(pre + "triplane"), one(pre + "triline")], dim=-1)
    else:
        out = one(typ)
    return out.reshape(x.shape[:-1] + (-1,))


def geometric_network(x, params, conf):
    """network.py:154-232 (geometric_init branch).  Returns sdf, feature, gain."""
    g = conf.geometric_network
    assert g.geometric_init and g.act == "softplus"
    L, M = g.layers, g.pe_bands
    skip_layers = list(g.skip_layers)
    scope = "geometric-network"
    pe_x = positional_encoding(x, M) if M > 0 else x
    vfeat = query_on_grid(x, params, conf)
    inputs = torch.cat([pe_x, vfeat], dim=-1) if vfeat is not None else pe_x
    h = inputs
    for l in range(L):
        if l == L - 1:
            h = affine(params, f"{scope}/affine-last", h)
        else:
            h = softplus100(affine(params, f"{scope}/affine-{l:02d}", h))
            if l != 0 and l not in skip_layers and (l + 1) in skip_layers:
                h = torch.cat([h, inputs], dim=-1)
                if g.use_inv_square:
                    h = h / math.sqrt(2)
    sdf, feature = h[..., 0:1], h[..., 1:]
    gain = torch.exp(params[f"{scope}/gain"] * 10).clamp(1e-6, 5e4)
    return sdf, feature, gain


def _mlp(params, scope, h, names, act=softplus100):
    for n in names[:-1]:
        h = act(affine(params, f"{scope}/{n}", h))
    return affine(params, f"{scope}/{names[-1]}", h)


def _names(L, shift=0):
    """affine-00.. ; roughness/specular nets use l-1 for hidden layers (network.py:450-454)."""
    return [f"affine-{l - shift:02d}" for l in range(L - 1)] + [f"affine-{L - 1:02d}"]


def _last_act(name, beta):
    return {"softplus": lambda v: TF.softplus(v, beta=beta), "relu": torch.relu,
            "sigmoid": torch.sigmoid}[name]


def base_color_network(x, feature, normal, params, conf):
    """network.py:235-263."""
    c = conf.base_color_network
    inputs = [x] + ([feature] if c.use_geometric_feature else []) + ([normal] if c.use_normal else [])
    h = torch.cat(inputs, dim=-1)
    return torch.sigmoid(_mlp(params, "base-color-network", h, _names(c.layers)))


def environment_light_network(light_dirs, params, conf):
    """network.py:266-297."""
    c = conf.environment_light_network
    h = positional_encoding(light_dirs, c.pe_bands) if c.pe_bands > 0 else light_dirs
    h = _mlp(params, "environment-light-network", h, _names(c.layers))
    out = _last_act(c.act_last, c.inverse_black_degree)(h)
    if c.upper_bound > 0:
        out = out.clamp(0.0, c.upper_bound)
    return out


def implicit_illumination_network(x, feature, normal, params, conf):
    """network.py:300-336."""
    c = conf.implicit_illumination_network
    if not c.use_me:
        return torch.zeros(x.shape[:-1] + (1,), dtype=x.dtype)
    inputs = [x] + ([feature] if c.use_geometric_feature else []) + ([normal] if c.use_normal else [])
    h = _mlp(params, "implicit-illumination-network", torch.cat(inputs, dim=-1), _names(c.layers))
    return _last_act(c.act_last, c.inverse_black_degree)(h)


def soft_visibility_light_network(x, light_dirs, feature, normal, params, conf):
    """network.py:339-377."""
    c = conf.soft_visibility_light_network
    pe = positional_encoding(light_dirs, c.pe_bands) if c.pe_bands > 0 else light_dirs
    inputs = [x, pe] + ([feature] if c.use_geometric_feature else []) + ([normal] if c.use_normal else [])
    h = _mlp(params, "soft-visibility-light-network", torch.cat(inputs, dim=-1), _names(c.layers))
    return _last_act(c.act_last, c.inverse_black_degree)(h)


def photogrammetric_light_network(x, camloc, view, feature, normal, params, conf):
    """network.py:380-424."""
    c = conf.photogrammetric_light_network
    B, R, N, _ = x.shape
    view = view.expand(B, R, N, 3)
    pe_view = positional_encoding(view, c.pe_bands) if c.pe_bands > 0 else view
    inputs = [x, pe_view, feature, normal]
    if c.use_inverse_distance:
        d = x - camloc.reshape(B, 1, 1, 3)
        dist2 = torch.sqrt((d * d).sum(-1, keepdim=True)) ** 2
        inputs.append(1.0 / (dist2 + 1e-5))
    h = _mlp(params, "photogrammetric-light-network", torch.cat(inputs, dim=-1), _names(c.layers))
    gain = params["photogrammetric-light-network/gain"].reshape((1,) * h.dim())
    return torch.sigmoid(gain * h)


def roughness_network(x, feature, normal, params, conf):
    """network.py:427-464."""
    c = conf.roughness_network
    inputs = [x] + ([feature] if c.use_geometric_feature else []) + ([normal] if c.use_normal else [])
    h = _mlp(params, "roughness-network", torch.cat(inputs, dim=-1), _names(c.layers, shift=1))
    h0, h1 = h[..., 0:1], h[..., 1:2]
    std = TF.softplus(h1)
    r = torch.sigmoid(h0)
    if conf.specular_brdf.model == "filament" and conf.specular_brdf.remap:
        r = r ** 2
    return r.clamp(c.lower_bound, 1.0), std


def specular_reflectance_network(x, feature, normal, params, conf):
    """network.py:467-509."""
    c = conf.specular_reflectance_network
    Do = c.channels
    if c.fixme:
        return torch.full(x.shape[:-1] + (Do,), 0.04, dtype=x.dtype), None
    inputs = [x] + ([feature] if c.use_geometric_feature else []) + ([normal] if c.use_normal else [])
    h = _mlp(params, "specular-reflectance-network", torch.cat(inputs, dim=-1), _names(c.layers, shift=1))
    h0, h1 = h[..., :-Do], h[..., Do:]
    std = TF.softplus(h1)
    s = torch.sigmoid(h0)
    if conf.specular_brdf.model == "filament" and conf.specular_brdf.remap:
        s = 0.16 * (s ** 2)
    else:
        s = c.upper_bound_scale * s
    return s, std


def background_network(x, view, delta, params, conf):
    """network.py:512-561."""
    c = conf.background_network
    B, R, N, _ = x.shape
    pe_x = positional_encoding(x, c.pe_bands0) if c.pe_bands0 > 0 else x
    h = _mlp(params, "background-network/geometric-network", pe_x, _names(c.layers0))
    density, feature = softplus100(h[..., 0:1]), h[..., 1:]
    alpha = 1 - torch.exp(-density * delta)
    view = view.expand(B, R, N, 3)
    if c.pe_bands1 > 0:
        h = torch.cat([x, feature, view, positional_encoding(view, c.pe_bands1)], dim=-1)
    else:
        h = torch.cat([x, feature, view], dim=-1)
    color = torch.sigmoid(_mlp(params, "background-network/lighting-network", h, _names(c.layers1)))
    return alpha, color


# ----------------------------------------------------------------------------------------------
# sampler.py (forward only, no gradient: backward_impl is `pass`, sampler.py:301-302)
# ----------------------------------------------------------------------------------------------
def _np_call(name, *args):
    from . import kernels as K
    K.call(name, *args)


def t_near_far(camloc, raydir, conf):
    """sampler.py:71-138 -> t_near (B,R,1), t_far (B,R,1), mask (B,R,1)."""
    import numpy as np
    B, R, _ = raydir.shape
    method = conf.renderer.t_near_far_method
    radius = conf.renderer.bounding_sphere_radius
    dt = raydir.dtype
    if method in ("intersect_with_aabb", "intersect_with_r_sphere"):
        c = np.ascontiguousarray(camloc.detach().numpy().astype(np.float32))
        d = np.ascontiguousarray(raydir.detach().numpy().astype(np.float32))
        tn = np.zeros((B, R, 1), np.float32)
        tf = np.zeros_like(tn)
        nh = np.zeros_like(tn)
        if method == "intersect_with_aabb":
            _np_call("ray_aabb_intersection", B * R, tn, tf, nh, c, d, B, R, [-radius] * 3, [radius] * 3)
        else:
            _np_call("ray_sphere_intersection", B * R, tn, tf, nh, c, d, B, R, radius)
        return (torch.from_numpy(tn).to(dt), torch.from_numpy(tf).to(dt),
                torch.from_numpy((nh > 1.0).astype(np.float32)).to(dt))
    if method == "intersect_with_midpoint":
        b = 2.0 * (camloc.reshape(B, 1, 3) * raydir).sum(-1, keepdim=True)
        mid = -b / 2.0
        return (mid - radius).clamp(min=0), mid + radius, torch.ones(B, R, 1, dtype=dt)
    if method == "intersect_with_camloc_dists":
        return _camloc_dists(camloc, R, radius)
    raise ValueError(method)


def _camloc_dists(camloc, R, radius):
    B = camloc.shape[0]
    d = torch.sqrt((camloc * camloc).sum(-1, keepdim=True))
    tn = (d - radius).reshape(B, 1, 1).expand(B, R, 1)
    tf = (d + radius).reshape(B, 1, 1).expand(B, R, 1)
    return tn, tf, torch.ones(B, R, 1, dtype=camloc.dtype)


def sample_stratified_dists(t_near, t_far, stratified_sample, conf):
    """sampler.py:140-165."""
    B, R, _ = t_far.shape
    N = conf.renderer.n_samples0
    tn, tf = t_near.reshape(B, R, 1, 1), t_far.reshape(B, R, 1, 1)
    step = (tf - tn) / N
    i = torch.arange(0, N, dtype=tn.dtype).reshape(1, 1, N, 1)
    return tn + step * (i + stratified_sample)


def importance_round(t, sdf, t_near, t_far, gain, M):
    """One up-sampling round of sampler.py:194-240 given the SDF at the current samples.
    t, sdf: (B,R,N,1).  Returns new sorted t (B,R,N+M,1) and the integer idx (B,R,M).
    fp32: the C restatement with the scan orders / exp of include/ndjir_math.h (the definition the
    HIP kernel is bit-compared against); other dtypes: the torch restatement below."""
    if t.dtype == torch.float32:
        return _importance_round_c(t, sdf, t_near, t_far, gain, M)
    return importance_round_torch(t, sdf, t_near, t_far, gain, M)


def _importance_round_c(t, sdf, t_near, t_far, gain, M):
    import numpy as np
    B, R, N, _ = t.shape
    f = lambda a, shp: np.ascontiguousarray(a.detach().expand(shp).numpy().astype(np.float32).reshape(-1))
    t_out = np.zeros((B * R, N + M), np.float32)
    idx = np.zeros((B * R, M), np.int32)
    _np_call("sampler_importance_round", B * R, N, M, float(gain), f(t, (B, R, N, 1)), f(sdf, (B, R, N, 1)),
             f(t_near, (B, R, 1, 1)), f(t_far, (B, R, 1, 1)), t_out, idx)
    return (torch.from_numpy(t_out).reshape(B, R, N + M, 1), torch.from_numpy(idx.astype(np.int64)).reshape(B, R, M))


def importance_round_torch(t, sdf, t_near, t_far, gain, M):
    """Stock-op restatement of sampler.py:194-240 (association of the scans left to torch)."""
    B, R, N, _ = t.shape
    ts_end = t[:, :, N - 1:N, :]
    sdf0, sdf1 = sdf[:, :, :-1, :], sdf[:, :, 1:, :]
    t0, t1 = t[:, :, :-1, :], t[:, :, 1:, :]
    sdfm = (sdf0 + sdf1) * 0.5
    cos_val1 = (sdf1 - sdf0) / (t1 - t0 + 1e-5)
    cos_val0 = torch.cat([torch.ones(B, R, 1, 1, dtype=t.dtype), cos_val1[:, :, :-1, :]], dim=2)
    cos_val = torch.minimum(cos_val0, cos_val1).clamp(-1e3, 0.0)
    dist = t1 - t0
    sdf0 = sdfm - cos_val * dist * 0.5
    sdf1 = sdfm + cos_val * dist * 0.5
    cdf0 = torch.sigmoid(sdf0 * gain)
    cdf1 = torch.sigmoid(sdf1 * gain)
    alpha = ((cdf0 - cdf1 + 1e-5) / (cdf0 + 1e-5)).clamp(0.0, 1.0)
    one_m = 1 - alpha
    excl = torch.cat([torch.ones_like(one_m[:, :, :1]), torch.cumprod(one_m, dim=2)[:, :, :-1]], dim=2)
    weights = (alpha * excl).reshape(B, R, N - 1)
    weights = weights / weights.sum(dim=2, keepdim=True)
    cumsum_w = torch.cumsum(weights, dim=2)
    u = (torch.arange(0, M, dtype=torch.float32) / (M - 1 + 1 / M)).to(t.dtype)
    u = u.reshape(1, 1, M).expand(B, R, M).contiguous()
    idx = torch.searchsorted(cumsum_w.contiguous(), u, right=False)
    cumsum_w0 = torch.cat([torch.zeros(B, R, 1, dtype=t.dtype), cumsum_w], dim=2)
    # nnabla's gather raises on an out-of-range index; idx == N-1 needs cumsum[-1] < u_max = 0.9959
    gi = idx.clamp(max=N - 2)
    denorm = torch.gather(weights, 2, gi)
    lower = torch.gather(cumsum_w0, 2, idx)
    ratio = ((u - lower) / denorm).reshape(B, R, M, 1)
    steps = torch.cat([t[:, :, 1:, :] - t[:, :, :-1, :], t_far - ts_end], dim=2)
    steps_idx = torch.gather(steps, 2, idx.unsqueeze(-1))
    ts_idx = torch.gather(t, 2, idx.unsqueeze(-1))
    t_new = ts_idx + steps_idx * ratio
    t_new = torch.maximum(torch.minimum(t_new, t_far), t_near)
    t_all, _ = torch.sort(torch.cat([t, t_new], dim=2), dim=2)
    return t_all, idx


def sample_importance_dists(camloc, raydir, t_near, t_far, t, params, conf, record=None):
    """sampler.py:167-242."""
    B, R, N, _ = t.shape
    M, U = conf.renderer.n_samples1, conf.renderer.n_upsamples
    c = camloc.reshape(B, 1, 1, 3)
    d = raydir.reshape(B, R, 1, 3)
    tn, tf = t_near.reshape(B, R, 1, 1), t_far.reshape(B, R, 1, 1)
    for u in range(U):
        x = c + t * d
        sdf, _, _ = geometric_network(x, params, conf)
        gain = conf.renderer.sampling_sigmoid_gain * 2 ** u
        if record is not None:
            record.setdefault("t_in", []).append(t.clone())
            record.setdefault("sdf", []).append(sdf.clone())
        t, idx = importance_round(t, sdf, tn, tf, gain, M)
        if record is not None:
            record.setdefault("idx", []).append(idx.clone())
            record.setdefault("t_out", []).append(t.clone())
    return t


def sample_points(camloc, raydir, stratified_sample, background_sample, params, conf, record=None):
    """sampler.py:256-299 (SamplePoints._forward_impl).
    Returns x_fg (B,R,N,3), t_fg (B,R,N+1,1), x_bg (B,R,Nb,4), t_bg (B,R,Nb+1,1), mask (B,R,1,1)."""
    with torch.no_grad():
        B, R, _ = raydir.shape
        t_near, t_far, mask = t_near_far(camloc, raydir, conf)
        t = sample_stratified_dists(t_near, t_far, stratified_sample, conf)
        t = sample_importance_dists(camloc, raydir, t_near, t_far, t, params, conf, record)
        c = camloc.reshape(B, 1, 1, 3)
        d = raydir.reshape(B, R, 1, 3)
        x_fg = c + t * d
        t_fg = torch.cat([t, t_far.reshape(B, R, 1, 1)], dim=2)
        Nb = conf.renderer.n_bg_samples
        if conf.background_modeling:
            tn_bg, _, _ = _camloc_dists(camloc, R, conf.renderer.bounding_sphere_radius)
            t_base = t_far * mask + tn_bg * (1 - mask)
            tb = t_base.reshape(B, R, 1, 1) / background_sample
            tb, _ = torch.sort(tb, dim=2)
            xb = c + tb[:, :, :-1, :] * d
            dists = torch.sqrt((xb * xb).sum(-1, keepdim=True)) + 1e-6
            x_bg = torch.cat([xb / dists, 1.0 / dists], dim=-1)
            t_bg = tb
        else:
            x_bg = torch.ones(B, R, Nb, 4, dtype=t.dtype)
            t_bg = torch.ones(B, R, Nb + 1, 1, dtype=t.dtype)
        return x_fg, t_fg, x_bg, t_bg, mask.reshape(B, R, 1, 1)


def sample_directions(normal, cdf_the, cdf_phi, alpha=None, eps=0.0):
    """sampler.py:317-408 -> C oracle of inverse_transform_cuda.cu.  No gradient."""
    import numpy as np
    B, R, _ = normal.shape
    nt, nph = cdf_the.shape[-1], cdf_phi.shape[-1]
    M = nt * nph
    f = lambda a: np.ascontiguousarray(a.detach().numpy().astype(np.float32))
    out = np.zeros((B, R, M, 3), np.float32)
    if alpha is None:
        _np_call("sample_uniform_directions", B * R * M, out, f(normal), f(cdf_the), f(cdf_phi),
                 B * R, M, nt, nph, eps)
    else:
        _np_call("sample_importance_directions", B * R * M, out, f(normal), f(cdf_the), f(cdf_phi),
                 f(alpha), B * R, M, nt, nph, eps)
    return torch.from_numpy(out).to(normal.dtype)


# ----------------------------------------------------------------------------------------------
# specular_brdf.py
# ----------------------------------------------------------------------------------------------
def dot(u, v, with_mask=False, eps=1e-8):
    """specular_brdf.py:23-37."""
    uv = (u * v).sum(-1, keepdim=True)
    mask = (uv > eps).to(uv.dtype).detach()
    uv = uv.clamp(min=eps)
    return (uv, mask) if with_mask else uv


def specular_brdf_model(normal, view_dir, light_dir, roughness, specular_color, conf):
    """specular_brdf.py:40-118 (filament) and :121-191 (ue4)."""
    B, R, _ = normal.shape
    M = light_dir.shape[2]
    normal = normal.reshape(B, R, 1, 3).expand(B, R, M, 3)
    view_dir = view_dir.reshape(B, R, 1, 3).expand(B, R, M, 3)
    roughness = roughness.reshape(B, R, 1, 1).expand(B, R, M, 1)
    specular_color = specular_color.reshape(B, R, 1, -1).expand(B, R, M, specular_color.shape[-1])
    half_dir = light_dir + view_dir
    half_dir = half_dir / torch.sqrt((half_dir * half_dir).sum(-1, keepdim=True))
    eps_dot = conf.renderer.eps_dot
    nol, m_nol = dot(normal, light_dir, True, eps_dot)
    nov, m_nov = dot(normal, view_dir, True, eps_dot)
    noh, m_noh = dot(normal, half_dir, True, eps_dot)
    voh = dot(view_dir, half_dir, False, eps_dot)
    eps = 1e-6
    model, sampling = conf.specular_brdf.model, conf.specular_brdf.sampling
    if model == "filament":
        a2 = roughness ** 2
        V1 = lambda nou: 1 / (nou + (a2 + (1 - a2) * nou ** 2) ** 0.5 + eps)
        V = V1(nol) * V1(nov)
        Fs = specular_color + (1 - specular_color) * (1 - voh) ** 5
        if sampling == "importance":
            s = V * Fs * (4 * voh / noh)
        else:
            D = a2 / (math.pi * (noh ** 2 * (a2 - 1) + 1) ** 2 + eps)
            s = math.pi * D * V * Fs
    else:
        a = roughness ** 2
        a2 = a ** 2
        k = (roughness + 1) ** 2 / 8
        G1 = lambda nou: nou / (nou * (1 - k) + k + eps)
        G = G1(nol) * G1(nov)
        Fs = specular_color + (1 - specular_color) * 2 ** ((-5.55473 * voh - 6.98316) * voh)
        if sampling == "importance":
            s = G * Fs * (voh / (noh * nov))
        else:
            D = a2 / (math.pi * (noh ** 2 * (a2 - 1) + 1) ** 2 + eps)
            s = math.pi * D * G * Fs / (4 * nov * nol)
    return s * (m_nol * m_nov * m_noh), nol


# ----------------------------------------------------------------------------------------------
# renderer.py
# ----------------------------------------------------------------------------------------------
def pb_render(x_fg, t_fg, x_bg, t_bg, camloc, raydir, mask, cos_anneal_ratio, rand, params, conf):
    """renderer.py:32-209.  `rand`: diffuse_cdf_the/phi, specular_cdf_the/phi (B,R,n), noise (B,R,N,3).
    x_fg must require grad."""
    B, R, N, _ = x_fg.shape
    raydir = raydir.reshape(B, R, 1, 3)
    view_dir = -raydir
    eps_normal = conf.renderer.eps_normal

    sdf, feature, gain = geometric_network(x_fg, params, conf)
    grad_x, = torch.autograd.grad(sdf, x_fg, torch.ones_like(sdf), create_graph=True)

    car = cos_anneal_ratio.reshape((1,) * x_fg.dim())
    true_cos = (raydir * grad_x).sum(-1, keepdim=True)
    iter_cos = -(torch.relu(-true_cos * 0.5 + 0.5) * (1.0 - car) + torch.relu(-true_cos) * car)
    delta_t = t_fg[:, :, 1:, :] - t_fg[:, :, :-1, :]
    sdf1 = sdf + iter_cos * delta_t * 0.5
    sdf0 = sdf - iter_cos * delta_t * 0.5
    g = gain.reshape((1,) * sdf.dim())
    cdf0 = torch.sigmoid(g * sdf0)
    cdf1 = torch.sigmoid(g * sdf1)
    alpha_fg = ((cdf0 - cdf1 + 1e-5) / (cdf0 + 1e-5)).clamp(0.0, 1.0)

    if conf.background_modeling:
        delta_bg = (t_bg[:, :, 1:, :] - t_bg[:, :, :-1, :]).detach()
        alpha_bg, color_bg = background_network(x_bg, view_dir, delta_bg, params, conf)
    else:
        alpha_bg = torch.ones(B, R, 1, 1, dtype=x_fg.dtype)
        color_bg = torch.full((B, R, 1, 3), conf.background_color, dtype=x_fg.dtype)

    alpha = torch.cat([alpha_fg * mask, alpha_bg], dim=2)
    one_m = 1 - alpha
    trans = torch.cat([torch.ones_like(one_m[:, :, :1]), torch.cumprod(one_m, dim=2)[:, :, :-1]], dim=2)
    weights = alpha * trans
    trans_fg, weights_fg, weights_bg = trans[:, :, :N], weights[:, :, :N], weights[:, :, N:]

    VR = lambda v, w=weights_fg: (w * v).sum(dim=2)

    gpix = VR(grad_x) + eps_normal
    normal_pixel = gpix / torch.sqrt((gpix * gpix).sum(-1, keepdim=True))

    M = rand["diffuse_cdf_the"].shape[-1] * rand["diffuse_cdf_phi"].shape[-1]
    D = feature.shape[-1]
    x_pix = VR(x_fg).reshape(B, R, 1, 3).expand(B, R, M, 3)
    f_pix = VR(feature).reshape(B, R, 1, D).expand(B, R, M, D)
    n_bc = normal_pixel[:, :, None, :].expand(B, R, M, 3)

    uni_dir = sample_directions(normal_pixel, rand["diffuse_cdf_the"], rand["diffuse_cdf_phi"])
    env = environment_light_network(uni_dir, params, conf)
    soft_vis = soft_visibility_light_network(x_pix, uni_dir, f_pix, n_bc, params, conf)

    implicit = implicit_illumination_network(x_fg, feature, grad_x, params, conf)
    implicit_pix = VR(implicit)

    cos = dot(n_bc, uni_dir)
    env_pix = (soft_vis * env * cos).mean(dim=2)
    diffuse_light_pix = env_pix + implicit_pix
    base_color = base_color_network(x_fg, feature, grad_x, params, conf)

    roughness, std_roughness = roughness_network(x_fg, feature, grad_x, params, conf)
    roughness_pix = VR(roughness)
    spec_refl, std_spec_refl = specular_reflectance_network(x_fg, feature, grad_x, params, conf)
    spec_refl_pix = VR(spec_refl)

    if conf.specular_brdf.sampling == "importance":
        imp_dir = sample_directions(normal_pixel, rand["specular_cdf_the"], rand["specular_cdf_phi"],
                                    roughness_pix)
    else:
        imp_dir = sample_directions(normal_pixel, rand["specular_cdf_the"], rand["specular_cdf_phi"])
    sBRDF, cos = specular_brdf_model(normal_pixel, view_dir, imp_dir, roughness_pix, spec_refl_pix, conf)
    env = environment_light_network(imp_dir, params, conf)
    soft_vis = soft_visibility_light_network(x_pix, imp_dir, f_pix, n_bc, params, conf)
    if conf.specular_brdf.use_split_sum:
        spec_pix = (soft_vis * env).mean(dim=2) * (sBRDF * cos).mean(dim=2)
    else:
        spec_pix = (sBRDF * soft_vis * env * cos).mean(dim=2)
    if conf.implicit_illumination_network.use_me and conf.implicit_illumination_network.use_me_on_specular:
        spec_pix = spec_pix + (sBRDF * implicit_pix[:, :, :, None]).mean(dim=2)
    spec_pix = conf.specular_brdf.weight * spec_pix

    if conf.photogrammetric_light_network.use_me:
        photo = photogrammetric_light_network(x_fg, camloc, view_dir, feature, grad_x, params, conf)
        photo_pix = VR(photo)
        if conf.diffuse_brdf.entangle:
            color_fg = VR(base_color * photo) * diffuse_light_pix + photo_pix * spec_pix
        else:
            color_fg = photo_pix * (VR(base_color) * diffuse_light_pix + spec_pix)
    else:
        color_fg = VR(base_color) + spec_pix

    color_pixel = color_fg + VR(color_bg, weights_bg)

    obj_mask_pred = torch.zeros((), dtype=x_fg.dtype)
    if conf.train.mask_weight > 0.0:
        obj_mask_pred = (alpha_fg * trans_fg).sum(dim=2)

    G = conf.geometric_network.voxel.grid_size
    r = conf.renderer.bounding_sphere_radius
    x_ptb = x_fg + rand["noise"] * (math.sqrt(3) * 2 * r / G)
    _, feature_ptb, _ = geometric_network(x_ptb, params, conf)
    base_color_ptb = base_color_network(x_ptb, feature_ptb, None, params, conf)

    return dict(color_pixel=color_pixel, sdf_x_fg=sdf, grad_x_fg=grad_x, alpha_fg=alpha_fg,
                trans_fg=trans_fg, obj_mask_pred=obj_mask_pred, base_color=base_color,
                base_color_ptb=base_color_ptb, roughness=roughness, specular_reflectance=spec_refl,
                std_roughness=std_roughness, std_specular_reflectance=std_spec_refl,
                normal_pixel=normal_pixel, weights_fg=weights_fg)


# ----------------------------------------------------------------------------------------------
# loss.py
# ----------------------------------------------------------------------------------------------
_TV = {"voxel_feature": C.tv_loss_on_voxel, "triplane_feature": C.tv_loss_on_triplane,
       "triline_feature": C.tv_loss_on_triline}


def total_loss(camloc, raydir, color_gt, obj_mask, cos_anneal_ratio, rand, params, conf, record=None,
               samples=None):
    """loss.py:27-192.  `rand` additionally holds stratified_sample (B,R,N0,1) and
    background_sample (B,R,Nb+1,1).  Returns dict of scalars (+ 'render' outputs).
    `samples` (x_fg, t_fg, x_bg, t_bg, mask) bypasses the sampler (renderer-only parity tests)."""
    B, R, _ = color_gt.shape
    if samples is not None:
        x_fg, t_fg, x_bg, t_bg, mask = samples
    else:
        x_fg, t_fg, x_bg, t_bg, mask = sample_points(camloc, raydir, rand["stratified_sample"],
                                                     rand["background_sample"], params, conf, record)
    x_fg = x_fg.detach().requires_grad_(True)
    res = pb_render(x_fg, t_fg, x_bg, t_bg, camloc, raydir, mask, cos_anneal_ratio, rand, params, conf)
    tr = conf.train
    zero = torch.zeros((), dtype=x_fg.dtype)
    N = x_fg.shape[2]

    err = (res["color_pixel"] - color_gt).abs() if tr.rgb_loss == "l1" else (res["color_pixel"] - color_gt) ** 2
    if tr.mask_weight > 0.0:
        loss_rgb = (err * obj_mask).sum() / (obj_mask.sum() + 1e-5)
    else:
        loss_rgb = err.sum() / (B * R)

    denorm = mask.sum() * N + 1e-5
    # loss.py:36 binds N = n_samples0; :72 rebinds N to the sample count only inside `if eikonal_weight > 0` (the
    # nested compute_tv_loss has its own N); :118 forms the priors' normaliser from whichever N is bound by then
    denorm_prior = mask.sum() * (N if tr.eikonal_weight > 0.0 else conf.renderer.n_samples0) + 1e-5
    loss_eikonal = zero
    if tr.eikonal_weight > 0.0:
        gx = res["grad_x_fg"]
        gn = torch.sqrt((gx * gx).sum(-1, keepdim=True))
        loss_eikonal = (((gn - 1) * mask) ** 2.0).sum() / denorm

    loss_tv = zero
    if conf.geometric_network.voxel.type != "none" and tr.tv_weight > 0.0:
        for name, p in params.items():
            if not name.endswith("feature/F"):
                continue
            fn = _TV[name.split("/")[-2]]
            tv = fn(x_fg.detach().reshape(-1, 3), p, sym_backward=tr.tv_sym_backward)
            tv = tv.reshape(x_fg.shape[:-1] + (-1,))
            loss_tv = loss_tv + (tv * mask).sum() / denorm

    loss_mask = zero
    if tr.mask_weight > 0.0:
        pred = res["obj_mask_pred"].clamp(1e-3, 1.0 - 1e-3)
        bce = -(obj_mask * torch.log(pred) + (1 - obj_mask) * torch.log(1 - pred))
        loss_mask = bce.sum() / (mask.sum() + 1e-5)

    prior_base_color = zero
    if tr.base_color_prior_weight > 0.0:
        bc = res["base_color"] if tr.base_color_prior_sym_backward else res["base_color"].detach()
        prior_base_color = ((bc - res["base_color_ptb"]).abs() * mask).sum() / denorm_prior

    prior_roughness = reg_std_roughness = zero
    if tr.roughness_prior_weight > 0.0:
        pr = (res["roughness"] - conf.roughness_network.prior_value).abs() / res["std_roughness"]
        prior_roughness = (pr * mask).sum() / denorm_prior
        reg_std_roughness = (torch.log(res["std_roughness"]).clamp(1e-5, 1e5) * mask).sum() / denorm_prior

    prior_spec = reg_std_spec = zero
    if tr.specular_reflectance_prior_weight > 0.0:
        ps = (res["specular_reflectance"] - conf.specular_reflectance_network.prior_value).abs() \
            / res["std_specular_reflectance"]
        prior_spec = (ps * mask).sum() / denorm_prior
        reg_std_spec = (torch.log(res["std_specular_reflectance"]).clamp(1e-5, 1e5) * mask).sum() / denorm_prior

    loss = (loss_rgb + tr.eikonal_weight * loss_eikonal + tr.tv_weight * loss_tv
            + tr.mask_weight * loss_mask + tr.base_color_prior_weight * prior_base_color
            + tr.roughness_prior_weight * prior_roughness
            + tr.specular_reflectance_prior_weight * prior_spec
            + tr.roughness_prior_weight * reg_std_roughness
            + tr.specular_reflectance_prior_weight * reg_std_spec)
    return dict(loss=loss, loss_rgb=loss_rgb, loss_eikonal=loss_eikonal, loss_tv=loss_tv,
                loss_mask=loss_mask, prior_base_color=prior_base_color,
                prior_roughness=prior_roughness, prior_specular_reflectance=prior_spec,
                reg_std_roughness=reg_std_roughness, reg_std_specular_reflectance=reg_std_spec,
                render=res, x_fg=x_fg, t_fg=t_fg, x_bg=x_bg, t_bg=t_bg, mask=mask)


# ----------------------------------------------------------------------------------------------
# extract_by_mc.py
# ----------------------------------------------------------------------------------------------
def compute_pts_vol(mins, maxs, grid_size, params, conf, batch_size=50000):
    """extract_by_mc.py:46-74, step by step: meshgrid (default 'xy' indexing), batches through
    geometric_network(...)[0], reshape to (y, x, z), transpose to (x, y, z)."""
    import numpy as np
    x = np.linspace(mins[0], maxs[0], grid_size).astype(np.float32)
    y = np.linspace(mins[1], maxs[1], grid_size).astype(np.float32)
    z = np.linspace(mins[2], maxs[2], grid_size).astype(np.float32)
    X, Y, Z = np.meshgrid(x, y, z)
    pts = np.stack((X.reshape(-1), Y.reshape(-1), Z.reshape(-1)), axis=1)
    dtype = next(iter(params.values())).dtype
    vol = []
    with torch.no_grad():
        for b in range(0, pts.shape[0], batch_size):
            p = torch.from_numpy(pts[b:b + batch_size]).to(dtype)
            vol.append(geometric_network(p, params, conf)[0].reshape(-1).numpy())
    vol = np.concatenate(vol).reshape((y.size, x.size, z.size)).transpose((1, 0, 2))
    return pts, vol


# ----------------------------------------------------------------------------------------------
# helper.py ray generation + renderer.py render_image
# ----------------------------------------------------------------------------------------------
def generate_all_pixels(W, H):
    """helper.py:75-81: (H*W, 2) pixel coordinates (x, y), y-major."""
    import numpy as np
    xx, yy = np.meshgrid(np.arange(0, W), np.arange(0, H))
    return np.asarray([xx.flatten(), yy.flatten()]).T


def generate_raydir_camloc(pose, intrinsic, xy):
    """helper.py:44-73 in numpy float64: x_w = R_c2w K^-1 (x, y, 1)^T, normalised."""
    import numpy as np
    B, R, _ = xy.shape
    R_c2w = pose[:, np.newaxis, :3, :3]
    camloc = pose[:, np.newaxis, :3, 3:4]
    K_inv = np.linalg.inv(intrinsic[:, np.newaxis, :, :])
    pix = np.concatenate([xy, np.ones([B, R, 1])], axis=-1)[:, :, :, np.newaxis]
    world = np.matmul(R_c2w, np.matmul(K_inv, pix)).reshape((B, R, 3))
    return world / np.sqrt(np.sum(world ** 2, axis=-1, keepdims=True)), camloc.reshape((B, 3))


def render_image(pose, intrinsic, resolution, rand, params, conf):
    """renderer.py:212-272: intrinsics scaled by 2^-n_down_samples, all pixels y-major, tile size
    P = valid.n_rays - mod(W*H, valid.n_rays), per tile {host rays in float64 -> fp32, sample_points,
    pb_render with cos_anneal_ratio = 1, color_pixel}, reshape to NCHW, clip to [0,1].
    `rand`: the P-ray random tensors of one tile, reused for every tile (explicit inputs; the reference's
    F.rand nodes are graph constants of the one graph it forwards per tile).  Where the reference's
    `xy[:, p:p+P].reshape((1, P, 2))` would raise on a short last tile, the tile is padded with the last
    pixel and the padding dropped."""
    import numpy as np
    scale = 1.0 / 2 ** conf.valid.n_down_samples
    W, H = resolution
    W, H = int(W * scale), int(H * scale)
    P = conf.valid.n_rays
    intrinsic = intrinsic.copy()
    for (i, j) in ((0, 0), (1, 1), (0, 2), (1, 2), (0, 1)):
        intrinsic[:, i, j] = intrinsic[:, i, j] * scale
    xy = generate_all_pixels(W, H).reshape((1, H * W, 2))
    _, m = divmod(W * H, P)
    P = P - m
    dtype = next(iter(params.values())).dtype
    car = torch.ones(1, dtype=dtype)
    rimage = np.zeros([1, H * W, 3])
    for p in range(0, H * W, P):
        xy_b = xy[:, p:p + P, :]
        n = xy_b.shape[1]
        if n < P:
            xy_b = np.concatenate([xy_b, np.repeat(xy_b[:, -1:, :], P - n, axis=1)], axis=1)
        raydir, camloc = generate_raydir_camloc(pose, intrinsic, xy_b)
        raydir = torch.from_numpy(raydir.astype(np.float32)).to(dtype)
        camloc = torch.from_numpy(camloc.astype(np.float32)).to(dtype)
        x_fg, t_fg, x_bg, t_bg, mask = sample_points(camloc, raydir, rand["stratified_sample"],
                                                     rand["background_sample"], params, conf)
        x_fg = x_fg.detach().requires_grad_(True)
        res = pb_render(x_fg, t_fg, x_bg, t_bg, camloc, raydir, mask, car, rand, params, conf)
        rimage[0, p:p + n, :] = res["color_pixel"].detach().reshape(P, 3)[:n].double().numpy()
    rimage = rimage.reshape((1, H, W, 3)).transpose((0, 3, 1, 2))
    return np.clip(rimage, 0.0, 1.0)
