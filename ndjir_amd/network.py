"""Networks of the hot path with the reference's function signatures.

Reference: python/network.py -- positional_encoding :96-117, query_on_grid :120-151,
geometric_network :154-232, base_color_network :235-263, environment_light_network :266-297,
implicit_illumination_network :300-336, soft_visibility_light_network :339-377,
photogrammetric_light_network :380-424, roughness_network :427-464,
specular_reflectance_network :467-509, background_network :512-561.

Parameters live in `ndjir_amd.parameter` under the reference's nnabla scope names (W is (in, out),
y = x @ W + b).  Grid queries go to the HIP ops of `ndjir_amd.grid_feature`; dense layers run as
device GEMMs (see DESIGN.md for the hand-written MFMA path that replaces them).
"""
import math

import os

import numpy as np
import torch
import torch.nn.functional as TF

from . import lib
from . import parameter as P
from . import parametric_functions as PF
# these imports attach the ops to the F / PF namespaces (network.py:25-33)
from .grid_feature import (cosine_triline_feature, cosine_triplane_feature, cosine_voxel_feature,  # noqa: F401
                           lanczos_triline_feature, lanczos_triplane_feature, lanczos_voxel_feature,
                           triline_feature, triplane_feature, voxel_feature)

prng = np.random.RandomState(313)  # nnabla.random.prng default seed


def seed(s):
    global prng
    prng = np.random.RandomState(s)


class GeometricInitializer:
    """network.py:36-56: SAL/IDR-style geometric initialisation (sphere of radius r)."""

    def __init__(self, Di, Do, sigma, zero_start=None, last=False):
        self.Di, self.Do, self.sigma, self.zero_start, self.last = Di, Do, sigma, zero_start, last

    def __call__(self, shape):
        w = np.sqrt(self.sigma) * prng.randn(self.Di, self.Do)
        if self.zero_start is not None:
            w[self.zero_start:, :] = 0.0
        if self.last:
            w[:, 0] = np.sqrt(np.pi / self.Di) * np.ones([self.Di]) + prng.randn(self.Di) * 1e-4
        return w


def _glorot_uniform(shape):
    """nnabla PF.affine default: UniformInitializer(calc_uniform_lim_glorot(in, out))."""
    lim = math.sqrt(6.0 / (shape[0] + shape[1]))
    return prng.uniform(-lim, lim, size=shape)


# Dense layers run through the fused MFMA chain kernel (ndjir_amd/mlp.py) wherever only
# first-order gradients are needed; the geometric network's main pass (differentiated twice through
# nn.grad) is one fused operator (ndjir_amd/geometric.py) for every shipped configuration and runs layer by
# layer otherwise (use_ste, relu, no geometric init).  `USE_FUSED = False` forces the layer-by-layer path
# everywhere (debugging / A-B timing only).  Both paths run on this package's own kernels: a single layer is
# `mlp.linear` (one-layer chain launch; weight-gradient and column-sum kernels in backward), never a library GEMM.
USE_FUSED = True


def affine_params(Din, D, use_wn=False, w_init=None, b_init=None, name=None):
    """Parameters of PF.affine inside scope `name`/affine (network.py:88-93)."""
    assert not use_wn, "weight normalisation (use_wn) is off in every shipped config"
    with P.parameter_scope(name), P.parameter_scope("affine"):
        W = P.get_parameter_or_create("W", (Din, D), w_init if w_init is not None else _glorot_uniform, True)
        b = P.get_parameter_or_create("b", (D,), b_init, True)
    return W, b


def affine(h, D, use_wn=False, w_init=None, b_init=None, name=None):
    """network.py:88-93: PF.affine on the last axis inside parameter scope `name`/affine."""
    Din = h.shape[-1]
    W, b = affine_params(Din, D, use_wn, w_init, b_init, name)
    from .mlp import linear
    return linear(h, W, b)


def _run_mlp(h, Ws, bs, act, skip_layer=-1, skip_scale=1.0, inputs=None, pack=None):
    """softplus-MLP on the last axis: fused chain kernel, or layer by layer."""
    if USE_FUSED and act is softplus:
        from .mlp import fused_mlp
        return fused_mlp(h, Ws, bs, 100.0, skip_layer, skip_scale, pack=pack)
    assert pack is None
    from .mlp import linear
    for j, (W, b) in enumerate(zip(Ws, bs)):
        h = linear(h, W, b)          # own one-layer kernels (no library GEMM), see ndjir_amd/mlp.py
        if j < len(Ws) - 1:
            h = act(h)
            if j == skip_layer:
                h = torch.cat([h, inputs], dim=-1) * skip_scale
    return h


def _sdf_gain(p):
    """network.py:229-231: clip(exp(10 gain), 1e-6, 5e4)."""
    from .volume import sdf_gain
    return sdf_gain(p, 10.0, 1e-6, 5e4)


def softplus(x, beta=100):
    return TF.softplus(x, beta=beta)


def _act(name):
    return {"relu": torch.relu, "softplus": softplus}[name]


class _PosEnc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, M, include_input):
        C = x.shape[-1]
        xc = x.detach().contiguous()
        P_ = xc.numel() // C
        out = torch.empty(x.shape[:-1] + ((C if include_input else 0) + 2 * C * M,), device=x.device, dtype=torch.float32)
        lib.call("positional_encoding", P_, C, M, int(include_input), xc, out)
        ctx.save_for_backward(xc)
        ctx.cfg = (P_, C, M, int(include_input))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        P_, C, M, inc = ctx.cfg
        gx = torch.empty_like(xc)
        lib.call("positional_encoding_backward", P_, C, M, inc, xc, g.contiguous(), gx)
        return gx, None, None


def positional_encoding(x, M=6, include_input=True):
    """network.py:96-117: [x, cos(x_i 2^k), sin(x_i 2^k)] with the band index fastest (one HIP launch)."""
    # inputs that carry a gradient stay on the stock-op composite: the geometric network differentiates
    # THROUGH this function twice (nn.grad, renderer.py:52), the kernel's backward is first-order only
    if x.is_cuda and x.dtype == torch.float32 and not x.requires_grad:
        return _PosEnc.apply(x, int(M), bool(include_input))
    bands = 2.0 ** torch.arange(0, M, dtype=x.dtype, device=x.device)
    b = (bands.reshape((1,) * x.dim() + (M,)) * x.unsqueeze(-1)).reshape(x.shape[:-1] + (-1,))
    g = [x, torch.cos(b), torch.sin(b)] if include_input else [torch.cos(b), torch.sin(b)]
    return torch.cat(g, dim=-1)


def _big_normal_init(std):
    """Device-side N(0, std) initialiser for grids too large to stage through numpy."""
    def init(shape):
        gen = torch.Generator(device=P.get_device())
        gen.manual_seed(int(prng.randint(0, 2 ** 31 - 1)))
        return torch.randn(tuple(shape), generator=gen, device=P.get_device(), dtype=torch.float32) * std
    return init


def query_on_grid(x, G, D, use_ste, type, parts=False):
    """network.py:120-151.  parts=True: the list of the grid's feature tensors instead of their concatenation (the tri-plane +
    tri-line pair: a caller that packs them into a wider row itself saves the torch.cat)."""
    if type == "none":
        return None
    n = G ** 3 * D if type.endswith("voxel") else 3 * G * G * D
    f_init = _big_normal_init(1e-3) if n > (1 << 24) else None
    if type.endswith("triplaneline"):
        pre = type[:-len("triplaneline")]
        feat0 = getattr(PF, pre + "query_on_triplane")(x, G, D, use_ste=use_ste, f_init=f_init)
        feat1 = getattr(PF, pre + "query_on_triline")(x, G, D, use_ste=use_ste)
        return [feat0, feat1] if parts else torch.cat([feat0, feat1], dim=-1)
    pre, topo = ("", type) if "_" not in type else type.split("_", 1)
    pre = pre + "_" if pre else ""
    out = getattr(PF, f"{pre}query_on_{topo}")(x, G, D, use_ste=use_ste, f_init=f_init)
    return [out] if parts else out


# grid type -> [(grid-feature family, parameter scope)] in concatenation order (python/network.py:120-151; the cosine /
# Lanczos variants keep the linear ones' parameter scopes)
_FUSED_GRIDS = {"none": []}
for _pre in ("", "cosine_", "lanczos_"):
    _FUSED_GRIDS[_pre + "voxel"] = [(_pre + "voxel", "voxel_feature")]
    _FUSED_GRIDS[_pre + "triplane"] = [(_pre + "triplane", "triplane_feature")]
    _FUSED_GRIDS[_pre + "triline"] = [(_pre + "triline", "triline_feature")]
    _FUSED_GRIDS[_pre + "triplaneline"] = [(_pre + "triplane", "triplane_feature"), (_pre + "triline", "triline_feature")]


def uses_fused_geometric(conf):
    """True when `geometric_network_with_grad` runs as fused chains that produce d(sdf)/dx themselves (no autograd
    graph needed for a forward-only render)."""
    g = conf.geometric_network
    return bool(USE_FUSED and g.geometric_init and g.act == "softplus" and g.voxel.type in _FUSED_GRIDS)


def geometric_network_with_grad(x, conf, packed=False):
    """`sdf, feature, gain = geometric_network(x, conf); grad_x = nn.grad([sdf], [x])[0]`
    (python/renderer.py:51-52) as one operator.  Dense-voxel / no-grid configurations with the
    geometric initialisation run as fused MFMA chains with the hand-derived double backward
    (ndjir_amd/geometric.py); other configurations run layer by layer through autograd.
    packed=True: a fifth result, the packed sample inputs Z (..., ld) = [x | feature | grad_x | spare] of the fused path
    (None otherwise) -- what `material_nets_raw` reads instead of a concatenation."""
    g = conf.geometric_network
    v = g.voxel
    fused_grids = _FUSED_GRIDS
    if uses_fused_geometric(conf):
        from .geometric import geometric_main
        _ensure_geometric_params(x, conf)
        Ws, bs, skip_at, scale = _geometric_param_lists(conf)
        params = P.get_parameters()
        grids = [(f, params[f"geometric-network/{scope}/F"]) for f, scope in fused_grids[v.type]]
        sdf, feat, grad_x, Z = geometric_main(x, grids, Ws, bs, g.pe_bands, skip_at, scale, use_ste=v.use_ste)
        with P.parameter_scope("geometric-network"):
            gain = P.get_parameter_or_create("gain", (1,), np.asarray([conf.train.sigmoid_gain]), True)
        gain = _sdf_gain(gain)
        return (sdf, feat, gain, grad_x, Z) if packed else (sdf, feat, gain, grad_x)
    from .grid_feature import grad as nn_grad
    sdf, feat, gain = geometric_network(x, conf)
    grad_x = nn_grad([sdf], [x])[0]
    return (sdf, feat, gain, grad_x, None) if packed else (sdf, feat, gain, grad_x)


def _ensure_geometric_params(x, conf):
    """Create the geometric network's parameters (lazily, in the reference's order) if missing."""
    if "geometric-network/affine-last/affine/W" not in P.get_parameters():
        with torch.no_grad():
            geometric_network(x.detach()[..., :1, :] if x.dim() > 2 else x.detach()[:1], conf, first_order_only=True)


def _geometric_param_lists(conf):
    g = conf.geometric_network
    L = g.layers
    params = P.get_parameters()
    names = [f"affine-{l:02d}" for l in range(L - 1)] + ["affine-last"]
    Ws = [params[f"geometric-network/{n}/affine/W"] for n in names]
    bs = [params[f"geometric-network/{n}/affine/b"] for n in names]
    skip_layers = list(g.skip_layers)
    skip_at = -1
    for l in range(1, L - 1):
        if l not in skip_layers and (l + 1) in skip_layers:
            skip_at = l
    scale = 1.0 / np.sqrt(2) if g.use_inv_square else 1.0
    return Ws, bs, skip_at, scale


_TRI_FUSED_MAX_POINTS = 16384     # (ndjir_amd/geometric.py TRI_FUSED_MAX_POINTS)
_NO_QUERY_ENCODE = bool(os.environ.get("NDJIR_NO_QUERY_ENCODE"))      # A/B: the voxel query and the input encoding as two launches


def geometric_network(x, conf, first_order_only=False, sdf_only=False, packed=False):
    """network.py:154-232.  x (..., 3) -> sdf (..., 1), feature (..., 256), gain (1,).

    `first_order_only`: the caller will differentiate the outputs at most once w.r.t. the
    parameters (sampler rounds, base-colour perturbation pass); the whole net then runs as one
    fused MFMA chain.  The main pass of pb_render goes through `nn.grad` (second-order terms) and
    runs layer by layer.  `sdf_only` (sampler) skips the 256 feature columns of the last layer --
    the reference computes and discards them (sampler.py:193).
    `packed` (with first_order_only, fused path): a fourth result Zp (..., ld) = [x | feature | spare], the packed input of the
    nets that read cat(x, feature) (feature is then a view of it and the sdf is not returned); None when not available."""
    with P.parameter_scope("geometric-network"):
        g = conf.geometric_network
        D, L, M = g.feature_size, g.layers, g.pe_bands
        act = _act(g.act)
        use_wn = conf.use_wn
        skip_layers = list(g.skip_layers)
        v = g.voxel

        # one dense voxel grid and a query that needs no gradient (the sampler's rounds, mesh extraction, the perturbed pass of
        # python/renderer.py:186-193): query + encoding in one launch (differentiable w.r.t. the grid)
        vfeat = None
        fused_in = None
        fparam = P.get_parameters().get("geometric-network/voxel_feature/F") if v.type.endswith("voxel") else None
        if (fparam is not None and x.is_cuda and x.dtype == torch.float32 and M > 0 and x.shape[-1] == 3 and g.geometric_init
                and not x.requires_grad and fparam.dim() == 4 and not _NO_QUERY_ENCODE):
            from .grid_feature import _core
            fused_in = _core.query_encode(v.type, x, fparam, M)
        elif (v.type.endswith("triplaneline") and x.is_cuda and x.dtype == torch.float32 and M > 0 and x.shape[-1] == 3
              and g.geometric_init and not x.requires_grad and not torch.is_grad_enabled() and not _NO_QUERY_ENCODE
              and x.numel() // 3 <= _TRI_FUSED_MAX_POINTS
              and "geometric-network/triplane_feature/F" in P.get_parameters() and "geometric-network/triline_feature/F" in P.get_parameters()):
            # tri-plane + tri-line, nothing to differentiate (the sampler's rounds, mesh extraction): both queries and the encoding in
            # one launch (ndjir_triplaneline_query_encode)
            from .grid_feature import _core
            Fp = P.get_parameters()["geometric-network/triplane_feature/F"]
            Fl = P.get_parameters()["geometric-network/triline_feature/F"]
            fam = _core.FAMILIES[v.type[:-len("triplaneline")] + "triplane"]
            Kin = 3 + 6 * M + 3 * Fp.shape[-1] + 3 * Fl.shape[-1]
            fused_in = torch.empty(x.shape[:-1] + (Kin,), device=x.device, dtype=torch.float32)
            lib.call("triplaneline_query_encode", fused_in.numel() // Kin, M, x.contiguous(), Fp.detach().contiguous(), Fp.shape[1],
                     Fp.shape[-1], Fl.detach().contiguous(), Fl.shape[1], Fl.shape[-1], [-1.0] * 3, [1.0] * 3, _core.interp_code(fam),
                     fused_in, Kin)
        else:
            vfeat = query_on_grid(x, v.grid_size, v.feature_size, v.use_ste, v.type, parts=True)
        if fused_in is not None:
            inputs, pe_x = fused_in, None
        elif (vfeat is not None and x.is_cuda and x.dtype == torch.float32 and M > 0 and x.shape[-1] == 3 and g.geometric_init
                and not any(t.requires_grad for t in vfeat) and not x.requires_grad):
            # nothing to differentiate (sampler rounds, mesh extraction): encoding and grid features side by side in one launch
            # (the tri-plane and tri-line features as two segments: no concatenation in between)
            Cs = [t.shape[-1] for t in vfeat]
            Kin = 3 + 6 * M + sum(Cs)
            inputs = torch.empty(x.shape[:-1] + (Kin,), device=x.device, dtype=torch.float32)
            lib.call("geo_encode", inputs.numel() // Kin, M, x.contiguous(), len(vfeat), [t.contiguous() for t in vfeat], Cs, inputs, Kin)
            pe_x = None
        else:
            pe_x = positional_encoding(x, M) if M > 0 else x
            inputs = torch.cat([pe_x] + list(vfeat), dim=-1) if vfeat is not None else pe_x
        h = inputs

        if not g.geometric_init:
            for l in range(L - 1):
                h = affine(h, D, use_wn, name=f"affine-{l:02d}")
                h = torch.cat([h, pe_x], dim=-1) if l in skip_layers else h
                h = act(h)
            h = affine(h, D + 1, use_wn, name=f"affine-{L - 1:02d}")
        else:
            # parameters, created in the reference's order with its initialisers (network.py:195-224)
            r0 = g.initial_sphere_radius
            Dx = x.shape[-1]
            Dinputs = inputs.shape[-1]
            Ws, bs = [], []
            Din = Dinputs
            skip_at = -1
            for l in range(L):
                if l == 0:
                    W, b = affine_params(Din, D, use_wn, GeometricInitializer(Din, D, 2 / D, Dx), name=f"affine-{l:02d}")
                    Din = D
                elif l in skip_layers:
                    W, b = affine_params(D, D, use_wn, GeometricInitializer(D, D, 2 / (D - Dinputs), -Dinputs),
                                         name=f"affine-{l:02d}")
                    Din = D
                elif l == L - 1:
                    Do = 1 + D
                    W, b = affine_params(D, Do, use_wn, GeometricInitializer(D, Do, 2 / Do, last=True),
                                         np.full((Do,), -r0), name="affine-last")
                else:
                    Do = D - Dinputs if l + 1 in skip_layers else D
                    W, b = affine_params(Din, Do, use_wn, GeometricInitializer(Din, Do, 2 / Do), name=f"affine-{l:02d}")
                    Din = Do
                    if l + 1 in skip_layers:
                        skip_at = l
                        Din = Do + Dinputs
                Ws.append(W)
                bs.append(b)
            scale = 1.0 / np.sqrt(2) if g.use_inv_square else 1.0
            if sdf_only:
                Ws = Ws[:-1] + [_sdf_column(Ws[-1])]
                bs = bs[:-1] + [_sdf_column(bs[-1])]
            Zp = None
            if first_order_only and USE_FUSED and act is softplus:
                if packed and not sdf_only and x.is_cuda and not os.environ.get("NDJIR_NO_PACKED_INPUT"):
                    Zp = _run_mlp(inputs, Ws, bs, act, skip_at, scale, pack=x)
                    h = None
                else:
                    h = _run_mlp(inputs, Ws, bs, act, skip_at, scale)
            else:
                from .mlp import linear
                for l in range(L):
                    h = linear(h, Ws[l], bs[l])
                    if l < L - 1:
                        h = act(h)
                        if l == skip_at:
                            h = torch.cat([h, inputs], dim=-1) * scale
        if g.geometric_init and Zp is not None:
            sdf, feature = None, Zp[..., x.shape[-1]:x.shape[-1] + D]
        else:
            sdf, feature = h[..., 0:1], h[..., 1:]
        gain = P.get_parameter_or_create("gain", (1,), np.asarray([conf.train.sigmoid_gain]), True)
        # (the sampler's and the perturbation pass's callers ignore the gain: three launches saved per call)
        gain = None if (sdf_only or first_order_only) else _sdf_gain(gain)
    if packed:
        return sdf, feature, gain, (Zp if g.geometric_init else None)
    return sdf, feature, gain


_SDF_COL_CACHE = {}


def _sdf_column(t):
    """Column 0 of the last layer (weights (D, 1+D) -> (D, 1), bias (1+D,) -> (1,)) as a stable
    tensor object (so that its packed copy is cached); refreshed when the parameter changes."""
    from . import mlp as _mlp
    if t.dim() == 2 and _mlp._TRACK is not None and t.is_cuda:
        return t.detach()[..., 0:1]          # tracked weights pack straight from the parameter (row stride = its width)
    key = (t.data_ptr(), t._version)
    hit = _SDF_COL_CACHE.get(key)
    if hit is None or hit[0] is not t:
        if len(_SDF_COL_CACHE) > 16:
            _SDF_COL_CACHE.clear()
        col = t.detach()[..., 0:1].contiguous()
        hit = (t, col)
        _SDF_COL_CACHE[key] = hit
    return hit[1]


def _last_act(name, beta):
    return {"softplus": lambda v: TF.softplus(v, beta=beta), "relu": torch.relu, "sigmoid": torch.sigmoid}[name]


def _mlp_params(Din, D, L, Dout, use_wn, shift=0):
    """Parameters of an MLP: L-1 hidden layers of width D named affine-{l-shift:02d} + output layer affine-{L-1:02d}."""
    Ws, bs = [], []
    for l in range(L - 1):
        W, b = affine_params(Din, D, use_wn, name=f"affine-{l - shift:02d}")
        Ws.append(W); bs.append(b)
        Din = D
    W, b = affine_params(Din, Dout, use_wn, name=f"affine-{L - 1:02d}")
    Ws.append(W); bs.append(b)
    return Ws, bs


def _mlp(h, D, L, Dout, act, use_wn, shift=0):
    Ws, bs = _mlp_params(h.shape[-1], D, L, Dout, use_wn, shift)
    return _run_mlp(h, Ws, bs, act)


def _cat_inputs(x, feature, normal, c):
    if c.use_normal and normal is None:
        # the reference concatenates None here and fails the same way (python/network.py:251-254 called from the perturbed
        # pass, python/renderer.py:193 "normal is not used"): a net evaluated there cannot have use_normal set
        raise ValueError("use_normal=true for a network that is also evaluated without a normal (perturbed base-colour pass)")
    inputs = [x] + ([feature] if c.use_geometric_feature else []) + ([normal] if c.use_normal else [])
    return torch.cat(inputs, dim=-1) if len(inputs) > 1 else x


def material_nets_raw(x, feature, normal, conf, packed=None, photo=None):
    """The four per-sample material nets that share the input cat(x, feature, normal) -- implicit illumination
    (network.py:300-336), base colour (:235-263), roughness (:427-464), specular reflectance (:467-509) -- evaluated
    as one operator on one input (ndjir_amd.mlp.MultiMLP): their raw (pre-activation) outputs, in that order.
    Returns None when the configuration does not allow it (different inputs, activations or weight normalisation);
    parameters are created in the order the separate calls create them.

    packed: the geometric pass's Z = [x | feature | normal | spare] (geometric_network_with_grad(packed=True)); the nets
    read its leading columns, no concatenation is built and one gradient flows back to the geometric node.
    photo = (camloc, view): also evaluate the photogrammetric light net (:380-424) in the same operator -- its per-ray
    inputs (the encoded view direction) enter as a row term, the inverse squared distance goes to Z's first spare
    column; a fifth output (raw) and the net's gain are appended: (imp, bc, rough, spec, photo_raw, photo_gain)."""
    cs = [conf.implicit_illumination_network, conf.base_color_network, conf.roughness_network, conf.specular_reflectance_network]
    if not (USE_FUSED and all(c.act == "softplus" for c in cs) and not conf.use_wn and cs[0].use_me and not cs[3].fixme) \
            or os.environ.get("NDJIR_NO_MULTI_MLP"):
        return None
    from .mlp import multi_mlp
    nx, nf, nn_ = x.shape[-1], feature.shape[-1], normal.shape[-1]
    # with the packed input every net reads a prefix of [x | feature | normal]; a concatenation must be the same for all
    prefix = all(c.use_geometric_feature or not c.use_normal for c in cs)
    if packed is not None and (not prefix or os.environ.get("NDJIR_NO_PACKED_INPUT")):
        packed = None
    if packed is None and not all(c.use_geometric_feature == cs[0].use_geometric_feature and c.use_normal == cs[0].use_normal
                                  for c in cs):
        return None
    inp = packed if packed is not None else _cat_inputs(x, feature, normal, cs[0])
    Dins = [nx + (nf if c.use_geometric_feature else 0) + (nn_ if c.use_normal else 0) for c in cs]
    Din = nx + nf + nn_                           # the packed columns [x | feature | normal]
    nets = []
    for scope, c, Dout, shift, Di in (("implicit-illumination-network", cs[0], cs[0].channels, 0, Dins[0]),
                                      ("base-color-network", cs[1], 3, 0, Dins[1]),
                                      ("roughness-network", cs[2], 2, 1, Dins[2]),
                                      ("specular-reflectance-network", cs[3], cs[3].channels * 2, 1, Dins[3])):
        with P.parameter_scope(scope):
            nets.append(_mlp_params(Di, c.feature_size, c.layers, Dout, conf.use_wn, shift))
    widths, rows = list(Dins), [None] * 4
    pc = conf.photogrammetric_light_network
    with_photo = (photo is not None and packed is not None and pc.use_me and pc.act == "softplus" and pc.layers >= 2
                  and Din + (1 if pc.use_inverse_distance else 0) <= packed.shape[-1])
    if with_photo:
        camloc, view = photo
        B, R, N, _ = x.shape
        with P.parameter_scope("photogrammetric-light-network"):
            view_ray = view.reshape(B, R, 3)
            pe_view = positional_encoding(view_ray, pc.pe_bands) if pc.pe_bands > 0 else view_ray
            nx, npe = x.shape[-1], pe_view.shape[-1]
            extra = 1 if pc.use_inverse_distance else 0
            Ws, bs = _mlp_params(Din + npe + extra, pc.feature_size, pc.layers, pc.channels, conf.use_wn)
            pgain = P.get_parameter_or_create("gain", (1,), np.asarray([conf.train.sigmoid_gain_lv_start]), False)
        from .mlp import linear, rows_except
        # reference input order [x, pe(view), feature, normal, 1/d^2]: the pe rows of the first layer act per ray
        row_term = linear(pe_view.reshape(B * R, npe), Ws[0][nx:nx + npe], bs[0])
        W0 = rows_except(Ws[0], nx, nx + npe)
        if extra:
            Z2 = packed.detach().view(-1, packed.shape[-1])
            lib.call("inverse_squared_distance", Z2.shape[0], R * N, Z2, Z2.shape[1], camloc.detach().reshape(B, 3).contiguous(),
                     _ColumnView(Z2, Din), Z2.shape[1])
        nets.append(([W0] + Ws[1:], [None] + bs[1:]))
        widths.append(Din + extra)
        rows.append((row_term, N))
    outs = multi_mlp(inp, nets, widths=widths if packed is not None else None, row_terms=rows if with_photo else None,
                     lazy_pad=packed is not None)
    return tuple(outs) + ((pgain,) if with_photo else ())


class _ColumnView:
    """Raw pointer to column `c` of a row-major GPU matrix (for kernels that take a pointer and a row stride)."""
    is_cuda, dtype = True, torch.float32

    def __init__(self, t, c):
        self.t, self.c = t, c

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return self.t.data_ptr() + 4 * self.c


def base_color_network(x, feature, normal, conf, raw=False, packed=None):
    """network.py:235-263.  raw=True: the net's output before the sigmoid (for volume.material_head).
    packed: a tensor whose leading columns are cat(x, feature[, normal]) (geometric_network(packed=True)): read in place."""
    with P.parameter_scope("base-color-network"):
        c = conf.base_color_network
        if (packed is not None and USE_FUSED and c.act == "softplus" and not conf.use_wn and c.use_geometric_feature
                and (normal is not None or not c.use_normal)):
            from .mlp import multi_mlp
            Din = x.shape[-1] + feature.shape[-1] + (normal.shape[-1] if c.use_normal else 0)
            h = multi_mlp(packed, [_mlp_params(Din, c.feature_size, c.layers, 3, conf.use_wn)], widths=[Din], lazy_pad=True)[0]
        else:
            h = _mlp(_cat_inputs(x, feature, normal, c), c.feature_size, c.layers, 3, _act(c.act), conf.use_wn)
        return h if raw else torch.sigmoid(h)


def environment_light_network(light_dirs, conf, raw=False):
    """network.py:266-297.  raw=True: the net's output before `act_last` and the upper-bound clamp (volume.direct_light
    applies both)."""
    with P.parameter_scope("environment-light-network"):
        c = conf.environment_light_network
        h = positional_encoding(light_dirs, c.pe_bands) if c.pe_bands > 0 else light_dirs
        h = _mlp(h, c.feature_size, c.layers, c.channels, _act(c.act), conf.use_wn)
        if raw:
            return h
        out = _last_act(c.act_last, c.inverse_black_degree)(h)
        if c.upper_bound > 0:
            out = out.clamp(0.0, c.upper_bound)
        return out


def implicit_illumination_network(x, feature, normal, conf, raw=False):
    """network.py:300-336."""
    with P.parameter_scope("implicit-illumination-network"):
        c = conf.implicit_illumination_network
        if not c.use_me:
            return torch.zeros(x.shape[:-1] + (1,), dtype=x.dtype, device=x.device)
        h = _mlp(_cat_inputs(x, feature, normal, c), c.feature_size, c.layers, c.channels, _act(c.act), conf.use_wn)
        return h if raw else _last_act(c.act_last, c.inverse_black_degree)(h)


def soft_visibility_light_network(x, light_dirs, feature, normal, conf, raw=False):
    """network.py:339-377.  raw=True: the net's output before `act_last`."""
    with P.parameter_scope("soft-visibility-light-network"):
        c = conf.soft_visibility_light_network
        pe = positional_encoding(light_dirs, c.pe_bands) if c.pe_bands > 0 else light_dirs
        per_ray = [x] + ([feature] if c.use_geometric_feature else []) + ([normal] if c.use_normal else [])
        act = _act(c.act)
        if (USE_FUSED and act is softplus and c.layers >= 2 and pe.dim() == 4
                and all(t.dim() == 4 and (t.shape[2] == 1 or t.stride(2) == 0) for t in per_ray)):
            # All inputs but the light direction are per-ray constants broadcast over the M lights.  The first
            # affine is linear: their share  [x, feature, normal] W_0[rows] + b_0  is computed once per RAY and
            # enters the fused chain as a per-row-group term; the (B,R,M,301) concatenation is never built and
            # the first layer multiplies 39 instead of 301 columns per light.
            B, R, M, _ = pe.shape
            nx, npe = x.shape[-1], pe.shape[-1]
            Ws, bs = _mlp_params(nx + npe + sum(t.shape[-1] for t in per_ray[1:]), c.feature_size, c.layers, c.channels,
                                 conf.use_wn)
            # (a (B, R, 1, C) tensor loses its unit axis by a view: `select`'s backward would fill a zero tensor and copy into it)
            ray_in = torch.cat([t.reshape(t.shape[0], t.shape[1], t.shape[3]) if t.shape[2] == 1 else t[:, :, 0, :] for t in per_ray],
                               dim=-1)
            from .mlp import fused_mlp, linear, rows_except
            W0_ray = rows_except(Ws[0], nx, nx + npe)
            row_term = linear(ray_in.reshape(B * R, -1), W0_ray, bs[0]).view(B, R, -1)
            h = fused_mlp(pe, [Ws[0][nx:nx + npe]] + Ws[1:], [None] + bs[1:], 100.0, row_bias=row_term, row_bias_div=M)
        else:
            per_ray = [t.expand(t.shape[0], t.shape[1], pe.shape[2], t.shape[3]) for t in per_ray]
            inputs = [per_ray[0], pe] + per_ray[1:]
            h = _mlp(torch.cat(inputs, dim=-1), c.feature_size, c.layers, c.channels, act, conf.use_wn)
        return h if raw else _last_act(c.act_last, c.inverse_black_degree)(h)


def photogrammetric_light_network(x, camloc, view, feature, normal, conf, raw=False):
    """network.py:380-424."""
    with P.parameter_scope("photogrammetric-light-network"):
        c = conf.photogrammetric_light_network
        B, R, N, _ = x.shape
        view = view.expand(B, R, N, 3)
        pe_view = positional_encoding(view, c.pe_bands) if c.pe_bands > 0 else view
        inputs = [x, pe_view, feature, normal]
        if c.use_inverse_distance:
            d = x - camloc.reshape(B, 1, 1, 3)
            dist2 = torch.sqrt((d * d).sum(-1, keepdim=True)) ** 2
            inputs.append(1.0 / (dist2 + 1e-5))
        h = _mlp(torch.cat(inputs, dim=-1), c.feature_size, c.layers, c.channels, _act(c.act), conf.use_wn)
        gain = P.get_parameter_or_create("gain", (1,), np.asarray([conf.train.sigmoid_gain_lv_start]), False)
        if raw:
            return h, gain
        return torch.sigmoid(gain.reshape((1,) * h.dim()) * h)


def roughness_network(x, feature, normal, conf, raw=False):
    """network.py:427-464 (hidden layers are named affine--1, affine-00, affine-01; :450-454)."""
    with P.parameter_scope("roughness-network"):
        c = conf.roughness_network
        h = _mlp(_cat_inputs(x, feature, normal, c), c.feature_size, c.layers, 2, _act(c.act), conf.use_wn, shift=1)
        if raw:
            return h
        h0, h1 = h[..., 0:1], h[..., 1:2]
        std = TF.softplus(h1)
        r = torch.sigmoid(h0)
        if conf.specular_brdf.model == "filament" and conf.specular_brdf.remap:
            r = r ** 2
        return r.clamp(c.lower_bound, 1.0), std


def specular_reflectance_network(x, feature, normal, conf, raw=False):
    """network.py:467-509."""
    with P.parameter_scope("specular-reflectance-network"):
        c = conf.specular_reflectance_network
        Do = c.channels
        if c.fixme:
            return torch.full(x.shape[:-1] + (Do,), 0.04, dtype=x.dtype, device=x.device), None
        h = _mlp(_cat_inputs(x, feature, normal, c), c.feature_size, c.layers, Do * 2, _act(c.act), conf.use_wn, shift=1)
        if raw:
            return h
        h0, h1 = h[..., :-Do], h[..., Do:]
        std = TF.softplus(h1)
        s = torch.sigmoid(h0)
        if conf.specular_brdf.model == "filament" and conf.specular_brdf.remap:
            s = 0.16 * (s ** 2)
        else:
            s = c.upper_bound_scale * s
        return s, std


def background_network(x, view, delta, conf):
    """network.py:512-561.  x (B,R,N,4) inverted-sphere coordinates."""
    with P.parameter_scope("background-network"):
        c = conf.background_network
        B, R, N, _ = x.shape
        act = _act(c.act)
        fused = (USE_FUSED and act is softplus and not conf.use_wn and c.layers1 >= 2 and x.is_cuda and view.shape[2] == 1
                 and not x.requires_grad and not delta.requires_grad and not os.environ.get("NDJIR_NO_BACKGROUND_HEAD"))
        with P.parameter_scope("geometric-network"):
            h = positional_encoding(x, c.pe_bands0) if c.pe_bands0 > 0 else x
            h = _mlp(h, c.feature_size0, c.layers0, c.feature_size0 + 1, act, conf.use_wn)
            if fused:
                # density, alpha and the lighting net's per-sample input [x | feature] in one launch (volume.background_head)
                from .volume import background_head
                alpha, inp = background_head(h, x, delta)
            else:
                density, feature = softplus(h[..., 0:1], 100), h[..., 1:]
                alpha = 1 - torch.exp(-density * delta)
        with P.parameter_scope("lighting-network"):
            if fused:
                # reference input order [x, feature, view, pe(view)]: the view rows of the first layer act per RAY -- their
                # share of the first affine (+ its bias) enters the fused chain as a row term, the (B,R,N,287) concatenation
                # is never built
                view_ray = view.reshape(B, R, 3)
                per_ray = torch.cat([view_ray, positional_encoding(view_ray, c.pe_bands1)], dim=-1) if c.pe_bands1 > 0 else view_ray
                ns = inp.shape[-1]
                Ws, bs = _mlp_params(ns + per_ray.shape[-1], c.feature_size1, c.layers1, 3, conf.use_wn)
                from .mlp import fused_mlp, linear
                row_term = linear(per_ray.reshape(B * R, -1), Ws[0][ns:], bs[0]).view(B, R, -1)
                raw = fused_mlp(inp, [Ws[0][:ns]] + Ws[1:], [None] + bs[1:], 100.0, row_bias=row_term, row_bias_div=N)
                color = torch.sigmoid(raw)
            else:
                view = view.expand(B, R, N, 3)
                if c.pe_bands1 > 0:
                    h = torch.cat([x, feature, view, positional_encoding(view, c.pe_bands1)], dim=-1)
                else:
                    h = torch.cat([x, feature, view], dim=-1)
                color = torch.sigmoid(_mlp(h, c.feature_size1, c.layers1, 3, act, conf.use_wn))
    return alpha, color
