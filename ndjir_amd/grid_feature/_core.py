"""Operator layer of the grid-feature ops on torch tensors, backed by the HIP C ABI.

Mirrors the reference's `nnabla.function.PythonFunction` classes
(python/grid_feature/voxel_feature.py:27-140 forward op, :171-282 grad-query op, :285-380
grad-feature op, :383-399 registered backward) as `torch.autograd.Function`s:

  Query          forward: <family>.query_*            backward: d/dquery -> GradQuery (a graph
                                                       node, so it is differentiable again, like
                                                       the reference's registered backward);
                                                       d/dfeature -> grad_feature
  GradQuery      forward: grad_query                   backward: grad_query_grad_grad_output,
                                                       grad_query_grad_query (linear voxel only),
                                                       grad_query_grad_feature
  GradFeature    forward: grad_feature                 backward (linear voxel only):
                                                       grad_feature_grad_grad_output, _grad_query

Gradients w.r.t. a grid can be accumulated IN PLACE into a caller-owned buffer
(`set_grad_buffer`), the torch analogue of nnabla's `accum` protocol -- a 512^3 x 4 grid is 2 GiB
and must not be materialised once per backward call.
"""
import weakref

import torch
from torch.autograd import Function

from .. import lib

from ..registry import REG

_GRAD_BUFFERS = REG.grid_grad_buffers      # (ndjir_amd/registry.py)

# nnabla distinguishes `nn.grad` (uses the *registered backward functions*, a differentiable graph)
# from `.backward()` (uses `backward_impl`).  torch has one backward, so the distinction is a mode:
#   with nn_grad(): torch.autograd.grad(sdf, x, create_graph=True)   # == nn.grad([sdf], [x])
_MODE = {"nn_grad": False, "backward_computes_grad_query": False}


class nn_grad:
    """Context manager: backward calls inside behave like nnabla's `nn.grad` (renderer.py:52)."""

    def __enter__(self):
        self._prev = _MODE["nn_grad"]
        _MODE["nn_grad"] = True
        return self

    def __exit__(self, *exc):
        _MODE["nn_grad"] = self._prev
        return False


def grad(outputs, inputs, grad_outputs=None):
    """nn.grad(outputs, inputs, grad_outputs): differentiable first-order gradients."""
    if grad_outputs is None:
        grad_outputs = [torch.ones_like(o) for o in outputs]
    with nn_grad():
        return list(torch.autograd.grad(outputs, inputs, grad_outputs, create_graph=True, allow_unused=True))


def set_grad_buffer(feature, buf):
    """Register `buf` (same shape as `feature`, fp32, GPU) as the accumulate-in-place gradient
    buffer of `feature`.  Ops then scatter-add into `buf` and return no dense gradient.
    The registry is keyed by the feature's address (the operators see detached views of the parameter, not the object):
    a registration therefore ends with the registered tensor -- when `feature` is collected its entry goes with it, so that
    a later tensor the allocator places at the same address never inherits a dead step's buffer (and with it a gradient of
    None).  `parameter.clear_parameters` drops every registration."""
    ptr = feature.data_ptr()
    old = _GRAD_BUFFERS.pop(ptr, None)
    if old is not None:
        old[1].detach()                      # (its finalizer must not remove the entry that replaces it)
    if buf is not None:
        assert buf.shape == feature.shape and buf.is_contiguous()
        _GRAD_BUFFERS[ptr] = (buf, weakref.finalize(feature, _GRAD_BUFFERS.pop, ptr, None))


def clear_grad_buffers():
    for _, fin in _GRAD_BUFFERS.values():
        fin.detach()
    _GRAD_BUFFERS.clear()


def zero_touched(buf, query, min_=(-1, -1, -1), max_=(1, 1, 1), interp="linear"):
    """Re-arm the accumulate-in-place gradient buffer `buf` (G0, G1, G2, D) of a dense voxel grid: zero only the
    cells the `query` points (..., 3) touched -- every tap of the family's stencil (linear / cosine: the 8 corners,
    which also hold the TV backward's cells; Lanczos: 4 x 4 x 4) -- instead of the whole buffer."""
    q = query.detach().reshape(-1, 3).contiguous()
    code = {"linear": 0, "cosine": 1, "lanczos": 2}[interp]
    if code == 0:
        lib.call("voxel_feature_zero_touched", q.shape[0], buf, q, list(buf.shape[:3]), buf.shape[3], list(min_), list(max_))
    else:
        lib.call("voxel_feature_zero_touched_interp", q.shape[0], buf, q, list(buf.shape[:3]), buf.shape[3], list(min_),
                 list(max_), code)


def get_grad_buffer(feature):
    e = _GRAD_BUFFERS.get(feature.data_ptr())
    return e[0] if e is not None else None


class Family:
    """One native module of the reference (e.g. voxel_feature_cuda) and its layouts."""

    def __init__(self, prefix, topo, fwd):
        self.prefix, self.topo, self.fwd = prefix, topo, fwd
        self.linear_voxel = prefix == "voxel_feature"

    def check(self, query, feature, hcfg):
        assert query.dim() > 1 and query.shape[-1] == 3, "Query shape must be (B1, ..., Bn, 3)."
        if self.topo == "voxel":
            assert feature.dim() == 4
        elif self.topo == "triplane":
            assert feature.dim() == 4 and feature.shape[0] == 3 and feature.shape[1] == feature.shape[2]
        elif self.topo == "triline":
            assert feature.dim() == 3 and feature.shape[0] == 3
        else:
            assert feature.dim() == 1 and hcfg is not None

    def channels(self, fshape, hcfg):
        if self.topo == "voxel":
            return fshape[-1]
        if self.topo in ("triplane", "triline"):
            return fshape[-1] * 3
        return hcfg[4] * hcfg[3]

    def shape_args(self, fshape, hcfg):
        if self.topo == "voxel":
            return [list(fshape[:3]), fshape[-1]]
        if self.topo in ("triplane", "triline"):
            return [fshape[1], fshape[-1]]
        return list(hcfg)

    def n(self, P, fshape, hcfg):
        if self.topo == "hash":
            return hcfg[3] * P
        return P * self.channels(fshape, hcfg)

    # hash kernels use the reference's (D, L, P) layout; wrapper level is (P, D*L)
    def to_native(self, t):
        return t.t().contiguous() if self.topo == "hash" else t.contiguous()

    def from_native(self, t, P):
        return t.view(-1, P).t().contiguous() if self.topo == "hash" else t


def interp_code(fam):
    """0 linear / 1 cosine / 2 Lanczos: the interpolation of a family, as the entry points that take it as an argument count it
    (ndjir_grid_pack_rows, ndjir_voxel_feature_query_encode)."""
    return 1 if fam.prefix.startswith("cosine_") else 2 if fam.prefix.startswith("lanczos_") else 0


def _flat(query):
    return query.detach().reshape(-1, 3).contiguous()


class QueryEncode(Function):
    """Rows [x | cos(x_d 2^k) | sin(x_d 2^k) | feature(x)] of the geometric network's input (python/network.py:96-117 + 120-151)
    for a dense voxel family and a query that needs NO gradient, as ONE launch (`ndjir_voxel_feature_query_encode`, bit-identical
    to the family's query followed by the encoding).  Backward: d/dfeature of the feature columns, as `Query`'s."""

    @staticmethod
    def forward(ctx, query, feature, fam, M, min_, max_):
        assert fam.topo == "voxel" and feature.dim() == 4 and query.shape[-1] == 3
        q = _flat(query)
        f = feature.detach().contiguous()
        P, D = q.shape[0], feature.shape[-1]
        W = 3 + 6 * int(M) + D
        out = torch.empty((P, W), device=q.device, dtype=torch.float32)
        lib.call("voxel_feature_query_encode", P, int(M), q, f, list(f.shape[:3]), D, list(min_), list(max_), interp_code(fam), out, W)
        ctx.save_for_backward(query, feature)
        ctx.cfg = (fam, int(M), tuple(min_), tuple(max_))
        return out.view(query.shape[:-1] + (W,))

    @staticmethod
    def backward(ctx, grad_output):
        query, feature = ctx.saved_tensors
        fam, M, min_, max_ = ctx.cfg
        gf = None
        if ctx.needs_input_grad[1] and not _MODE["nn_grad"]:      # (nn.grad: the registered backward never touches the feature)
            go = grad_output[..., 3 + 6 * M:].contiguous()
            gf = GradFeature.apply(go, query, feature, fam, min_, max_, False, None)
        return None, gf, None, None, None, None


def query_encode(family, query, feature, M, min_=(-1, -1, -1), max_=(1, 1, 1)):
    return QueryEncode.apply(query, feature, FAMILIES[family], int(M), tuple(float(v) for v in min_), tuple(float(v) for v in max_))


class Query(Function):
    @staticmethod
    def forward(ctx, query, feature, fam, min_, max_, use_ste, boundary_check, hcfg):
        fam.check(query, feature, hcfg)
        q = _flat(query)
        f = feature.detach().contiguous()
        P = q.shape[0]
        C = fam.channels(feature.shape, hcfg)
        out = torch.empty((C, P) if fam.topo == "hash" else (P, C), device=q.device, dtype=torch.float32)
        lib.call(f"{fam.prefix}_{fam.fwd}", fam.n(P, feature.shape, hcfg), out, q, f,
                 *fam.shape_args(feature.shape, hcfg), min_, max_, int(boundary_check))
        ctx.save_for_backward(query, feature)
        ctx.cfg = (fam, min_, max_, use_ste, boundary_check, hcfg)
        return fam.from_native(out, P).view(query.shape[:-1] + (C,))

    @staticmethod
    def backward(ctx, grad_output):
        query, feature = ctx.saved_tensors
        fam, min_, max_, use_ste, boundary_check, hcfg = ctx.cfg
        gq = gf = None
        if _MODE["nn_grad"]:
            # nn.grad path = the reference's registered backward function: returns
            # (grad_query graph node, None) and never touches the feature (voxel_feature.py:383-399)
            if ctx.needs_input_grad[0] and not use_ste:
                gq = GradQuery.apply(grad_output, query, feature, fam, min_, max_, boundary_check, hcfg)
            return gq, None, None, None, None, None, None, None
        # .backward() path = the reference's backward_impl: d/dfeature only; d/dquery is
        # deliberately not computed there ("Do not call for optimization", voxel_feature.py:108-116)
        if ctx.needs_input_grad[0] and not use_ste and _MODE["backward_computes_grad_query"]:
            gq = GradQuery.apply(grad_output, query, feature, fam, min_, max_, boundary_check, hcfg)
        if ctx.needs_input_grad[1]:
            gf = GradFeature.apply(grad_output, query, feature, fam, min_, max_, boundary_check, hcfg)
        return gq, gf, None, None, None, None, None, None


class GradQuery(Function):
    """grad_query(grad_output, query, feature) -> (.., 3)  (voxel_feature.py:171-282)."""

    @staticmethod
    def forward(ctx, grad_output, query, feature, fam, min_, max_, boundary_check, hcfg):
        q = _flat(query)
        f = feature.detach().contiguous()
        P = q.shape[0]
        go = fam.to_native(grad_output.detach().reshape(P, -1))
        gq = torch.empty((P, 3), device=q.device, dtype=torch.float32)
        lib.call(f"{fam.prefix}_grad_query", fam.n(P, feature.shape, hcfg), gq, go, q, f,
                 *fam.shape_args(feature.shape, hcfg), min_, max_, int(boundary_check), 0)
        ctx.save_for_backward(grad_output, query, feature)
        ctx.cfg = (fam, min_, max_, boundary_check, hcfg)
        return gq.view(query.shape)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg_query):
        grad_output, query, feature = ctx.saved_tensors
        fam, min_, max_, boundary_check, hcfg = ctx.cfg
        q = _flat(query)
        f = feature.detach().contiguous()
        P = q.shape[0]
        C = fam.channels(feature.shape, hcfg)
        ggq = gg_query.reshape(P, 3).contiguous()
        sargs = fam.shape_args(feature.shape, hcfg)
        N = fam.n(P, feature.shape, hcfg)
        g_go = g_q = g_f = None
        if ctx.needs_input_grad[0]:
            ggo = torch.empty((C, P) if fam.topo == "hash" else (P, C), device=q.device, dtype=torch.float32)
            lib.call(f"{fam.prefix}_grad_query_grad_grad_output", N, ggo, ggq, q, f, *sargs, min_, max_,
                     int(boundary_check), 0)
            g_go = fam.from_native(ggo, P).view(grad_output.shape)
        go = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            go = fam.to_native(grad_output.detach().reshape(P, -1))
        if ctx.needs_input_grad[1] and fam.linear_voxel:
            # only the linear dense voxel grid implements this term (voxel_feature.py:248-255)
            g_q = torch.zeros((P, 3), device=q.device, dtype=torch.float32)
            lib.call("voxel_feature_grad_query_grad_query", N, g_q, ggq, go, q, f, *sargs, min_, max_,
                     int(boundary_check), 1)
            g_q = g_q.view(query.shape)
        if ctx.needs_input_grad[2]:
            buf = get_grad_buffer(feature)
            dst = buf if buf is not None else torch.zeros_like(f)
            lib.call(f"{fam.prefix}_grad_query_grad_feature", N, dst, ggq, go, q, *sargs, min_, max_,
                     int(boundary_check), 1)
            g_f = None if buf is not None else dst
        return g_go, g_q, g_f, None, None, None, None, None


class GradFeature(Function):
    """grad_feature(grad_output, query) -> feature-shaped gradient (voxel_feature.py:285-380).
    With a registered grad buffer the scatter-add happens in place and None is returned upstream."""

    @staticmethod
    def forward(ctx, grad_output, query, feature, fam, min_, max_, boundary_check, hcfg):
        q = _flat(query)
        P = q.shape[0]
        go = fam.to_native(grad_output.detach().reshape(P, -1))
        buf = get_grad_buffer(feature)
        ctx.in_place = buf is not None
        dst = buf if buf is not None else torch.empty_like(feature, memory_format=torch.contiguous_format)
        lib.call(f"{fam.prefix}_grad_feature", fam.n(P, feature.shape, hcfg), dst, go, q,
                 *fam.shape_args(feature.shape, hcfg), min_, max_, int(boundary_check), 1 if ctx.in_place else 0)
        ctx.save_for_backward(grad_output, query)
        ctx.cfg = (fam, min_, max_, boundary_check, hcfg, tuple(feature.shape))
        if ctx.in_place:
            ctx.mark_non_differentiable()
            return None
        return dst

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg_feature):
        grad_output, query = ctx.saved_tensors
        fam, min_, max_, boundary_check, hcfg, fshape = ctx.cfg
        if not fam.linear_voxel:
            raise NotImplementedError(
                f"second-order terms of {fam.prefix}.grad_feature are not implemented by the reference either")
        q = _flat(query)
        P = q.shape[0]
        ggf = gg_feature.contiguous()
        sargs = fam.shape_args(fshape, hcfg)
        N = fam.n(P, fshape, hcfg)
        g_go = g_q = None
        if ctx.needs_input_grad[0]:
            g_go = torch.empty((P, fshape[-1]), device=q.device, dtype=torch.float32)
            lib.call("voxel_feature_grad_feature_grad_grad_output", N, g_go, ggf, q, *sargs, min_, max_,
                     int(boundary_check), 0)
            g_go = g_go.view(grad_output.shape)
        if ctx.needs_input_grad[1]:
            g_q = torch.zeros((P, 3), device=q.device, dtype=torch.float32)
            lib.call("voxel_feature_grad_feature_grad_query", N, g_q, ggf, grad_output.detach().reshape(P, -1).contiguous(),
                     q, *sargs, min_, max_, int(boundary_check), 1)
            g_q = g_q.view(query.shape)
        return g_go, g_q, None, None, None, None, None, None


class TVLoss(Function):
    """tv_loss_on_*(query, feature) -> per-sample TV (total_variation_loss.py:22-130).
    Backward reaches the feature only (query gets none, :92-93)."""

    @staticmethod
    def forward(ctx, query, feature, fam, tvname, min_, max_, sym_backward, boundary_check, hcfg):
        fam.check(query, feature, hcfg)
        q = _flat(query)
        f = feature.detach().contiguous()
        P = q.shape[0]
        C = fam.channels(feature.shape, hcfg)
        out = torch.empty((C, P) if fam.topo == "hash" else (P, C), device=q.device, dtype=torch.float32)
        lib.call(tvname, fam.n(P, feature.shape, hcfg), out, q, f, *fam.shape_args(feature.shape, hcfg),
                 min_, max_, int(boundary_check))
        ctx.save_for_backward(query, feature)
        ctx.cfg = (fam, tvname, min_, max_, sym_backward, boundary_check, hcfg)
        return fam.from_native(out, P).view(query.shape[:-1] + (C,))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        query, feature = ctx.saved_tensors
        fam, tvname, min_, max_, sym_backward, boundary_check, hcfg = ctx.cfg
        if not ctx.needs_input_grad[1]:
            return (None,) * 9
        q = _flat(query)
        f = feature.detach().contiguous()
        P = q.shape[0]
        go = fam.to_native(grad_output.reshape(P, -1))
        buf = get_grad_buffer(feature)
        dst = buf if buf is not None else torch.zeros_like(f)
        lib.call(tvname + "_backward", fam.n(P, feature.shape, hcfg), dst, go, q, f,
                 *fam.shape_args(feature.shape, hcfg), min_, max_, int(sym_backward), int(boundary_check), 1)
        return None, (None if buf is not None else dst), None, None, None, None, None, None, None


FAMILIES = {
    "voxel": Family("voxel_feature", "voxel", "query_on_voxel"),
    "cosine_voxel": Family("cosine_voxel_feature", "voxel", "query_on_voxel"),
    "lanczos_voxel": Family("lanczos_voxel_feature", "voxel", "query_on_voxel"),
    "triplane": Family("triplane_feature", "triplane", "query_on_triplane"),
    "cosine_triplane": Family("cosine_triplane_feature", "triplane", "query_on_triplane"),
    "lanczos_triplane": Family("lanczos_triplane_feature", "triplane", "query_on_triplane"),
    "triline": Family("triline_feature", "triline", "query_on_triline"),
    "cosine_triline": Family("cosine_triline_feature", "triline", "query_on_triline"),
    "lanczos_triline": Family("lanczos_triline_feature", "triline", "query_on_triline"),
    "voxel_hash": Family("voxel_hash_feature", "hash", "voxel_hash_feature"),
    "lanczos_voxel_hash": Family("lanczos_voxel_hash_feature", "hash", "voxel_hash_feature"),
}

TV_NAMES = {
    "voxel": "total_variation_loss_tv_loss_on_voxel",
    "triplane": "total_variation_loss_on_triplane_tv_loss_on_triplane",
    "triline": "total_variation_loss_on_triline_tv_loss_on_triline",
    "hash": "total_variation_loss_on_voxel_hash_tv_loss_on_voxel_hash",
}


def query(family, query, feature, min_=(-1, -1, -1), max_=(1, 1, 1), use_ste=False, boundary_check=False, hcfg=None):
    return Query.apply(query, feature, FAMILIES[family], tuple(min_), tuple(max_), use_ste, boundary_check, hcfg)


def grad_query(family, grad_output, query, feature, min_=(-1, -1, -1), max_=(1, 1, 1), boundary_check=False, hcfg=None):
    return GradQuery.apply(grad_output, query, feature, FAMILIES[family], tuple(min_), tuple(max_), boundary_check, hcfg)


def grad_feature(family, grad_output, query, feature, min_=(-1, -1, -1), max_=(1, 1, 1), boundary_check=False, hcfg=None):
    return GradFeature.apply(grad_output, query, feature, FAMILIES[family], tuple(min_), tuple(max_), boundary_check, hcfg)


def tv_loss(topo_family, query, feature, min_=(-1, -1, -1), max_=(1, 1, 1), sym_backward=False, boundary_check=False,
            hcfg=None):
    fam = FAMILIES[topo_family]
    return TVLoss.apply(query, feature, fam, TV_NAMES[fam.topo], tuple(min_), tuple(max_), sym_backward,
                        boundary_check, hcfg)
