"""Volume-rendering stage of pb_render as fused HIP operators (ndjir_amd/csrc/render.hip).

Reference: python/renderer.py:55-67 (foreground alpha), :79-87 (transmittance, weights, the `VR`
integrals).  `alpha_weights` and `integrate` are drop-ins for those lines with hand-derived
backward kernels; tests/test_gpu_render.py checks both against the stock-op composite (autograd).
"""
import torch
from torch.autograd import Function

from . import lib
from .mlp import _Strided



def _c(t):
    return t.detach().contiguous()


class AlphaWeights(Function):
    """sdf (B,R,N,1), n = d sdf/dx (B,R,N,3), raydir (B,R,3), t_fg (B,R,N+1,1), gain (1,),
    cos_anneal_ratio (1,), mask (B,R,1,1), alpha_bg (B,R,Nb,1)
    -> alpha_fg (B,R,N,1), trans (B,R,N+Nb,1), weights (B,R,N+Nb,1)."""

    @staticmethod
    def forward(ctx, sdf, n, raydir, t_fg, gain, car, mask, alpha_bg):
        B, R, N, _ = sdf.shape
        Nb = alpha_bg.shape[2]
        dev = sdf.device
        args = [_c(sdf), _c(n), _c(raydir), _c(t_fg), _c(gain), _c(car), _c(mask), _c(alpha_bg)]
        alpha_fg = torch.empty((B, R, N, 1), device=dev, dtype=torch.float32)
        trans = torch.empty((B, R, N + Nb, 1), device=dev, dtype=torch.float32)
        weights = torch.empty((B, R, N + Nb, 1), device=dev, dtype=torch.float32)
        lib.call("render_alpha_weights", B * R, N, Nb, *args, alpha_fg, trans, weights)
        ctx.save_for_backward(*args, trans)
        ctx.dims = (B, R, N, Nb)
        ctx.set_materialize_grads(False)      # (an unused output's gradient arrives as None -- the kernel takes null --, not as zeros)
        return alpha_fg, trans, weights

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_alpha, g_trans, g_weights):
        B, R, N, Nb = ctx.dims
        sdf, n, raydir, t_fg, gain, car, mask, alpha_bg, trans = ctx.saved_tensors
        dev = sdf.device
        g_sdf = torch.empty((B, R, N, 1), device=dev, dtype=torch.float32)
        g_n = torch.empty((B, R, N, 3), device=dev, dtype=torch.float32)
        g_gain_ray = torch.empty((B * R,), device=dev, dtype=torch.float32)
        g_bg = torch.empty((B, R, Nb, 1), device=dev, dtype=torch.float32) if ctx.needs_input_grad[7] else None
        opt = lambda g: None if g is None else g.contiguous()
        lib.call("render_alpha_weights_backward", B * R, N, Nb, sdf, n, raydir, t_fg, gain, car, mask, alpha_bg, trans,
                 opt(g_alpha), opt(g_trans), opt(g_weights), g_sdf, g_n, g_gain_ray, g_bg)
        g_gain = g_gain_ray.sum().reshape(gain.shape) if ctx.needs_input_grad[4] else None
        return g_sdf, g_n, None, None, g_gain, None, None, g_bg


def alpha_weights(sdf, n, raydir, t_fg, gain, cos_anneal_ratio, mask, alpha_bg):
    return AlphaWeights.apply(sdf, n, raydir, t_fg, gain.reshape(-1), cos_anneal_ratio.reshape(-1), mask, alpha_bg)


class Integrate(Function):
    """VR (renderer.py:84-87): out (B,R,C) = sum_i weights[:, :, off+i] x[:, :, i, :] for the S
    samples of x (B,R,S,C); `weights` is the full (B,R,S_all,1) tensor, `off` selects the
    foreground (0) or background (N) part without a copy."""

    @staticmethod
    def forward(ctx, weights, off, x):
        B, R, S, C = x.shape
        S_all = weights.shape[2]
        w = _c(weights)
        xc = x.detach()
        ld = xc.stride(2)
        if not (xc.stride(3) == 1 and ld >= C and xc.stride(1) == S * ld and xc.stride(0) == R * S * ld):
            xc = xc.contiguous()      # anything but a column slice of a row-major (B,R,S,ld) array
            ld = C
        out = torch.empty((B, R, C), device=x.device, dtype=torch.float32)
        lib.call("render_integrate", B * R, S, C, _Strided(w.view(B * R, S_all)[:, off:]), S_all, _Strided(xc), ld, out)
        ctx.save_for_backward(w, xc)
        ctx.off, ctx.ldx = off, ld
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        w, x = ctx.saved_tensors
        B, R, S, C = x.shape
        S_all = w.shape[2]
        off = ctx.off
        need_w, need_x = ctx.needs_input_grad[0], ctx.needs_input_grad[2]
        gx = torch.empty((B, R, S, C), device=x.device, dtype=torch.float32) if need_x else None
        gw = None
        if need_w:
            gw = torch.zeros_like(w) if S != S_all else torch.empty_like(w)
        lib.call("render_integrate_backward", B * R, S, C, _Strided(w.view(B * R, S_all)[:, off:]), S_all, _Strided(x), ctx.ldx,
                 g.contiguous(),
                 gx, _Strided(gw.view(B * R, S_all)[:, off:]) if need_w else None, S_all)
        return gw, None, gx


def integrate(weights, x, off=0):
    return Integrate.apply(weights, off, x)


class PixelNormal(Function):
    """renderer.py:90-91: normal_pixel = (VR(grad) + eps) / |VR(grad) + eps| (one launch each way instead of 5 + 8)."""

    @staticmethod
    def forward(ctx, grad_pixel, eps):
        g = _c(grad_pixel)
        n = torch.empty_like(g)
        lib.call("render_pixel_normal", g.numel() // 3, float(eps), g, n)
        ctx.save_for_backward(g)
        ctx.eps = float(eps)
        return n

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gn):
        (g,) = ctx.saved_tensors
        gg = torch.empty_like(g)
        lib.call("render_pixel_normal_backward", g.numel() // 3, ctx.eps, g, gn.contiguous(), gg)
        return gg, None


def pixel_normal(grad_pixel, eps):
    return PixelNormal.apply(grad_pixel, eps)


class IntegrateMany(Function):
    """Several VR integrals (renderer.py:84-87) of the same weights in one launch each way: apply(weights, offs, *xs) with
    xs[k] (B,R,S_k,C_k) against weights[:, :, offs[k]:offs[k]+S_k] -> one (B,R,C_k) per x.  The weight gradient is formed
    once, inside the backward launch, instead of one tensor per integral summed by autograd."""

    @staticmethod
    def forward(ctx, weights, offs, *xs):
        B, R, S_all, _ = weights.shape
        w = _c(weights)
        xcs, lds = [], []
        for x in xs:
            _, _, S, C = x.shape
            xc = x.detach()
            ld = xc.stride(2)
            if not (xc.stride(3) == 1 and ld >= C and xc.stride(1) == S * ld and xc.stride(0) == R * S * ld):
                xc = xc.contiguous()
                ld = C
            xcs.append(xc)
            lds.append(ld)
        outs = [torch.empty((B, R, x.shape[3]), device=w.device, dtype=torch.float32) for x in xs]
        lib.call("render_integrate_many", B * R, S_all, w, len(xs), [_Strided(x) for x in xcs], lds, [x.shape[3] for x in xs],
                 [x.shape[2] for x in xs], list(offs), outs)
        ctx.save_for_backward(w, *xcs)
        ctx.cfg = (tuple(offs), tuple(lds))
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        w, *xs = ctx.saved_tensors
        offs, lds = ctx.cfg
        B, R, S_all, _ = w.shape
        need_w = ctx.needs_input_grad[0]
        gxs = [torch.empty(x.shape, device=w.device, dtype=torch.float32) if (ctx.needs_input_grad[2 + k] and gs[k] is not None)
               else None for k, x in enumerate(xs)]
        gw = torch.empty_like(w) if need_w else None
        lib.call("render_integrate_many_backward", B * R, S_all, w, len(xs), [_Strided(x) for x in xs], list(lds),
                 [x.shape[3] for x in xs], [x.shape[2] for x in xs], list(offs),
                 [g.contiguous() if g is not None else None for g in gs], gxs, gw)
        return (gw, None, *gxs)


def integrate_many(weights, xs, offs):
    return IntegrateMany.apply(weights, tuple(int(o) for o in offs), *xs)


class DiffuseLight(Function):
    """renderer.py:117-118: mean over the M light directions of soft_vis * env * clamp(n.l, eps).
    normal (B,R,3), light_dir (B,R,M,3) [no grad], soft_vis (B,R,M,1), env (B,R,M,C) -> (B,R,C)."""

    @staticmethod
    def forward(ctx, normal, light_dir, soft_vis, env, eps_dot):
        B, R, M, C = env.shape
        args = [_c(normal), _c(light_dir), _c(soft_vis), _c(env)]
        out = torch.empty((B, R, C), device=env.device, dtype=torch.float32)
        lib.call("render_diffuse_light", B * R, M, C, *args, float(eps_dot), out)
        ctx.save_for_backward(*args)
        ctx.cfg = (B, R, M, C, float(eps_dot))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        B, R, M, C, eps = ctx.cfg
        normal, light, sv, env = ctx.saved_tensors
        gn = torch.empty_like(normal)
        gsv = torch.empty_like(sv)
        genv = torch.empty_like(env)
        lib.call("render_diffuse_light_backward", B * R, M, C, normal, light, sv, env, eps, g.contiguous(), gn, gsv, genv)
        return gn, None, gsv, genv, None


def diffuse_light(normal, light_dir, soft_vis, env, eps_dot):
    return DiffuseLight.apply(normal, light_dir, soft_vis, env, eps_dot)


class SpecularLightFilament(Function):
    """renderer.py:136-161 for model=filament, sampling=importance, use_split_sum=false:
    weight * mean_m( sBRDF * soft_vis * env * clamp(n.l, eps) ), BRDF per specular_brdf.py:40-118.
    normal, view_dir (B,R,3); light_dir (B,R,M,3); roughness (B,R,1); specular_color (B,R,3);
    soft_vis (B,R,M,1); env (B,R,M,C) with C = 1 or 3 -> (B,R,3)."""

    @staticmethod
    def forward(ctx, normal, view_dir, light_dir, roughness, specular_color, soft_vis, env, eps_dot, weight):
        B, R, M, C = env.shape
        args = [_c(normal), _c(view_dir), _c(light_dir), _c(roughness), _c(specular_color), _c(soft_vis), _c(env)]
        out = torch.empty((B, R, 3), device=env.device, dtype=torch.float32)
        lib.call("render_specular_light_filament", B * R, M, C, *args, float(eps_dot), float(weight), out)
        ctx.save_for_backward(*args)
        ctx.cfg = (B, R, M, C, float(eps_dot), float(weight))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        B, R, M, C, eps, weight = ctx.cfg
        normal, view, light, rough, spec, sv, env = ctx.saved_tensors
        gn, gr, gs = torch.empty_like(normal), torch.empty_like(rough), torch.empty_like(spec)
        gsv, genv = torch.empty_like(sv), torch.empty_like(env)
        lib.call("render_specular_light_filament_backward", B * R, M, C, normal, view, light, rough, spec, sv, env, eps, weight,
                 g.contiguous(), gn, gr, gs, gsv, genv)
        return gn, None, None, gr, gs, gsv, genv, None, None


def specular_light_filament(normal, view_dir, light_dir, roughness, specular_color, soft_vis, env, eps_dot, weight):
    return SpecularLightFilament.apply(normal, view_dir, light_dir, roughness, specular_color, soft_vis, env, eps_dot, weight)


class SpecularLight(Function):
    """renderer.py:141-161 for every other branch of the specular BRDF (python/specular_brdf.py:40-199): model filament | ue4,
    sampling importance | uniform, with or without the split sum -- BRDF algebra and the light integral(s) in one launch each
    way (csrc/render.hip k_specular_light_g).  Shapes as SpecularLightFilament; cfg = (model, sampling, split)."""

    @staticmethod
    def forward(ctx, normal, view_dir, light_dir, roughness, specular_color, soft_vis, env, eps_dot, weight, cfg):
        B, R, M, C = env.shape
        args = [_c(normal), _c(view_dir), _c(light_dir), _c(roughness), _c(specular_color), _c(soft_vis), _c(env)]
        out = torch.empty((B, R, 3), device=env.device, dtype=torch.float32)
        lib.call("render_specular_light", B * R, M, C, *cfg, *args, float(eps_dot), float(weight), out)
        ctx.save_for_backward(*args)
        ctx.cfg = (B, R, M, C, float(eps_dot), float(weight), tuple(cfg))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        B, R, M, C, eps, weight, cfg = ctx.cfg
        normal, view, light, rough, spec, sv, env = ctx.saved_tensors
        gn, gr, gs = torch.empty_like(normal), torch.empty_like(rough), torch.empty_like(spec)
        gsv, genv = torch.empty_like(sv), torch.empty_like(env)
        lib.call("render_specular_light_backward", B * R, M, C, *cfg, normal, view, light, rough, spec, sv, env, eps, weight,
                 g.contiguous(), gn, gr, gs, gsv, genv)
        return gn, None, None, gr, gs, gsv, genv, None, None, None


def specular_light(normal, view_dir, light_dir, roughness, specular_color, soft_vis, env, eps_dot, weight, model, sampling,
                   use_split_sum):
    cfg = ({"filament": 0, "ue4": 1}[model], {"importance": 0, "uniform": 1}[sampling], 1 if use_split_sum else 0)
    return SpecularLight.apply(normal, view_dir, light_dir, roughness, specular_color, soft_vis, env, eps_dot, weight, cfg)


class MaterialHead(Function):
    """Output activations of the per-sample material nets and the prior integrands in one launch
    (csrc/render.hip; network.py:262, 335, 423, 456-463, 498-508 and loss.py:117-166).
    Raw net outputs -> V (B,R,N,9) = [implicit, roughness, specular x3, photo, base colour (x photo)],
    aux (B,R,N,10) = [base x3, base_ptb x3, std_roughness, std_specular x3] (no gradient),
    prior (B,R,5) per-ray sums of the five prior / regulariser integrands."""

    @staticmethod
    def forward(ctx, raw_bc, raw_ptb, raw_imp, raw_photo, photo_gain, raw_rough, raw_spec, cfg):
        B, R, N, _ = raw_bc.shape
        args = [_c(raw_bc), _c(raw_ptb), _c(raw_imp), _c(raw_photo), _c(photo_gain).reshape(-1), _c(raw_rough), _c(raw_spec)]
        dev = raw_bc.device
        V = torch.empty((B, R, N, 9), device=dev, dtype=torch.float32)
        aux = torch.empty((B, R, N, 10), device=dev, dtype=torch.float32)
        prior = torch.empty((B, R, 5), device=dev, dtype=torch.float32)
        lib.call("render_material_head", B * R, N, *args, *cfg, V, aux, prior)
        ctx.save_for_backward(*args)
        ctx.cfg = (B, R, N, cfg)
        ctx.mark_non_differentiable(aux)
        ctx.set_materialize_grads(False)      # (no zero tensor for aux's "gradient")
        return V, aux, prior

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gV, _g_aux, g_prior):
        B, R, N, cfg = ctx.cfg
        raw_bc, raw_ptb, raw_imp, raw_photo, gain, raw_rough, raw_spec = ctx.saved_tensors
        outs = [torch.empty_like(t) for t in (raw_bc, raw_ptb, raw_imp, raw_photo, raw_rough, raw_spec)]
        if gV is None:
            gV = torch.zeros((B, R, N, 9), device=raw_bc.device, dtype=torch.float32)
        lib.call("render_material_head_backward", B * R, N, raw_bc, raw_ptb, raw_imp, raw_photo, gain, raw_rough, raw_spec, *cfg,
                 gV.contiguous(), None if g_prior is None else g_prior.contiguous(), *outs)
        return outs[0], outs[1], outs[2], outs[3], None, outs[4], outs[5], None


def material_head(raw_bc, raw_ptb, raw_imp, raw_photo, photo_gain, raw_rough, raw_spec, remap, entangle, sym_backward,
                  roughness_lower_bound, specular_scale, roughness_prior, specular_prior):
    cfg = (int(bool(remap)), int(bool(entangle)), int(bool(sym_backward)), float(roughness_lower_bound), float(specular_scale),
           float(roughness_prior), float(specular_prior))
    return MaterialHead.apply(raw_bc, raw_ptb, raw_imp, raw_photo, photo_gain, raw_rough, raw_spec, cfg)


class PixelCompose(Function):
    """renderer.py:163-178 for the fused material head: diffuse = env + implicit; color = base * diffuse + photo * spec
    (entangle) or photo * (base * diffuse + spec); + background.  pix (B,R,9) = VR of the head's V, env (B,R,1|3),
    spec (B,R,3), bg (B,R,3) -> color_pixel (B,R,3).  One launch each way (csrc/loss.hip)."""

    @staticmethod
    def forward(ctx, pix, env, spec, bg, entangle):
        B, R, _ = pix.shape
        args = [_c(pix), _c(env), _c(spec)]
        color = torch.empty((B, R, 3), device=pix.device, dtype=torch.float32)
        lib.call("render_pixel_compose", B * R, env.shape[-1], int(entangle), *args, _c(bg), color)
        ctx.save_for_backward(*args)
        ctx.cfg = (B, R, env.shape[-1], int(entangle))
        return color

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        B, R, Ce, entangle = ctx.cfg
        pix, env, spec = ctx.saved_tensors
        g_pix, g_env, g_spec = torch.empty_like(pix), torch.empty_like(env), torch.empty_like(spec)
        g_bg = torch.empty((B, R, 3), device=pix.device, dtype=torch.float32) if ctx.needs_input_grad[3] else None
        lib.call("render_pixel_compose_backward", B * R, Ce, entangle, pix, env, spec, g.contiguous(), g_pix, g_env, g_spec, g_bg)
        return g_pix, g_env, g_spec, g_bg, None


def pixel_compose(pix, env, spec, bg, entangle):
    return PixelCompose.apply(pix, env, spec, bg, bool(entangle))


class BackgroundHead(Function):
    """python/network.py:543-556 between the background model's two nets, one launch each way (csrc/render.hip):
    h (B,R,N,1+F), x (B,R,N,nx) [no grad], delta (B,R,N,1) [no grad] -> alpha (B,R,N,1) = 1 - exp(-softplus_100(h_0) delta) and
    the lighting net's per-sample input [x | h_1..F] (B,R,N,nx+F)."""

    @staticmethod
    def forward(ctx, h, x, delta):
        hc, xc, dc = _c(h), _c(x), _c(delta)
        F, nx = hc.shape[-1] - 1, xc.shape[-1]
        P = hc.numel() // (F + 1)
        alpha = torch.empty(hc.shape[:-1] + (1,), device=hc.device, dtype=torch.float32)
        inp = torch.empty(hc.shape[:-1] + (nx + F,), device=hc.device, dtype=torch.float32)
        lib.call("render_background_head", P, nx, F, hc, xc, dc, alpha, inp)
        ctx.save_for_backward(hc, dc)
        ctx.cfg = (P, nx, F)
        return alpha, inp

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_alpha, g_inp):
        hc, dc = ctx.saved_tensors
        P, nx, F = ctx.cfg
        g_h = torch.empty_like(hc)
        lib.call("render_background_head_backward", P, nx, F, hc, dc, None if g_alpha is None else g_alpha.contiguous(),
                 None if g_inp is None else g_inp.contiguous(), g_h)
        return g_h, None, None


def background_head(h, x, delta):
    return BackgroundHead.apply(h, x, delta)


class Gain(Function):
    """clamp(exp(scale p), lo, hi) of the geometric network's scalar gain parameter (python/network.py:229-231): one launch
    each way instead of mul / exp / clamp and their eight backward launches."""

    @staticmethod
    def forward(ctx, p, scale, lo, hi):
        pc = _c(p)
        out = torch.empty_like(pc)
        lib.call("render_gain", pc.numel(), pc, scale, lo, hi, out)
        ctx.save_for_backward(pc)
        ctx.cfg = (scale, lo, hi)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (pc,) = ctx.saved_tensors
        gp = torch.empty_like(pc)
        lib.call("render_gain_backward", pc.numel(), pc, *ctx.cfg, g.contiguous(), gp)
        return gp, None, None, None


def sdf_gain(p, scale=10.0, lo=1e-6, hi=5e4):
    if p.is_cuda and p.dtype == torch.float32:
        return Gain.apply(p, float(scale), float(lo), float(hi))
    return torch.exp(p * scale).clamp(lo, hi)


LIGHT_ACTS = {"identity": 0, "softplus": 1, "sigmoid": 2, "relu": 3}


class DirectLight(Function):
    """renderer.py:105-178, default branch, one launch each way (csrc/render.hip k_direct_light): output activations of the
    environment-light / soft-visibility nets, the diffuse and the filament-specular light integrals and the pixel
    composition.  normal, view_dir (B,R,3); light_dirs (B,R,2M,3) [diffuse | specular, no grad]; raw_soft_vis (B,R,2M,1),
    raw_env (B,R,2M,C): the nets' outputs BEFORE `act_last`; pix (B,R,9) = VR of the material head's V; bg (B,R,3) or None
    -> color_pixel (B,R,3).  cfg = (acts (sv, env), params (beta_sv, beta_env, ub_env, eps_dot, weight), entangle)."""

    @staticmethod
    def forward(ctx, normal, view_dir, light_dirs, raw_sv, raw_env, pix, bg, cfg):
        B, R, M2, C = raw_env.shape
        acts, params, entangle = cfg
        args = [_c(normal), _c(view_dir), _c(light_dirs), _c(raw_sv), _c(raw_env), _c(pix)]
        dev = pix.device
        color = torch.empty((B, R, 3), device=dev, dtype=torch.float32)
        env_pix = torch.empty((B, R, C), device=dev, dtype=torch.float32)
        spec_pix = torch.empty((B, R, 3), device=dev, dtype=torch.float32)
        lib.call("render_direct_light", B * R, M2 // 2, C, acts, params, int(entangle), *args, None if bg is None else _c(bg),
                 color, env_pix, spec_pix)
        ctx.save_for_backward(*args, env_pix, spec_pix)
        ctx.cfg = (B, R, M2 // 2, C, cfg, bg is not None)
        return color

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        B, R, M, C, (acts, params, entangle), has_bg = ctx.cfg
        normal, view, dirs, raw_sv, raw_env, pix, env_pix, spec_pix = ctx.saved_tensors
        gn, gsv, genv, gpix = torch.empty_like(normal), torch.empty_like(raw_sv), torch.empty_like(raw_env), torch.empty_like(pix)
        gbg = torch.empty((B, R, 3), device=pix.device, dtype=torch.float32) if has_bg and ctx.needs_input_grad[6] else None
        lib.call("render_direct_light_backward", B * R, M, C, acts, params, int(entangle), normal, view, dirs, raw_sv, raw_env, pix,
                 env_pix, spec_pix, g.contiguous(), gn, gsv, genv, gpix, gbg)
        return gn, None, None, gsv, genv, gpix, gbg, None


def direct_light(normal, view_dir, light_dirs, raw_soft_vis, raw_env, pix, bg, acts, params, entangle):
    cfg = (tuple(int(a) for a in acts), tuple(float(p) for p in params), bool(entangle))
    return DirectLight.apply(normal, view_dir, light_dirs, raw_soft_vis, raw_env, pix, bg, cfg)


LOSS_TERM_NAMES = ("loss", "loss_rgb", "loss_eikonal", "loss_tv", "prior_base_color", "prior_roughness", "reg_std_roughness",
                   "prior_specular_reflectance", "reg_std_specular_reflectance")


class LossTerms(Function):
    """python/loss.py:59-178 without the mask term: RGB error, eikonal, sampled TV, the five prior / regulariser sums and
    their weighted total in one pass over the rays plus a fixed-order final reduction (csrc/loss.hip).
    color, color_gt (B,R,3); mask (B,R,1,1); grad_x (B,R,N,3) or None; prior (B,R,5) or None (per-ray sums of the material
    head); mask_sum_global 0-d tensor or None; cfg = (N, inv_rays, (w_eikonal, w_tv, w_base_color, w_roughness,
    w_specular), l2, N_prior); tvs: up to two (B,R,N,D) sampled TV tensors.
    -> terms (12,): see LOSS_TERM_NAMES for [0..8]; only terms[0] (the total) carries a gradient -- take the other
    entries detached (`total_loss` does)."""

    @staticmethod
    def forward(ctx, color, color_gt, mask, grad_x, prior, mask_sum_global, cfg, *tvs):
        N, inv_rays, weights, l2, N_prior = cfg
        R = color.shape[0] * color.shape[1]
        dev = color.device
        assert len(tvs) <= 2
        tv = [_c(t) for t in tvs] + [None] * (2 - len(tvs))
        D = [t.shape[-1] if t is not None else 0 for t in tv]
        args = [_c(color), _c(color_gt), _c(mask).reshape(-1), None if grad_x is None else _c(grad_x)]
        ws = torch.empty(lib.load().ndjir_loss_terms_workspace(R), device=dev, dtype=torch.float32)
        terms = torch.empty(12, device=dev, dtype=torch.float32)
        lib.call("loss_terms", R, N, N_prior, *args, tv[0], D[0], tv[1], D[1], None if prior is None else _c(prior),
                 None if mask_sum_global is None else _c(mask_sum_global).reshape(1), float(inv_rays), [float(w) for w in weights],
                 int(l2), ws, terms)
        ctx.save_for_backward(args[0], args[1], args[2], *([args[3]] if args[3] is not None else []), terms)
        ctx.cfg = (R, N, float(inv_rays), [float(w) for w in weights], int(l2), D, grad_x is not None, prior is not None,
                   [t.shape if t is not None else None for t in tv], tuple(color.shape), None if grad_x is None else tuple(grad_x.shape))
        return terms

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_terms):
        R, N, inv_rays, weights, l2, D, has_gx, has_prior, tv_shapes, cshape, gxshape = ctx.cfg
        saved = ctx.saved_tensors
        color, gt, mask = saved[0], saved[1], saved[2]
        grad_x = saved[3] if has_gx else None
        terms = saved[-1]
        dev = color.device
        g_color = torch.empty(cshape, device=dev, dtype=torch.float32)
        g_gx = torch.empty(gxshape, device=dev, dtype=torch.float32) if (has_gx and ctx.needs_input_grad[3]) else None
        g_prior = torch.empty((cshape[0], cshape[1], 5), device=dev, dtype=torch.float32) if (has_prior and ctx.needs_input_grad[4]) else None
        g_tv = [torch.empty(s, device=dev, dtype=torch.float32) if (s is not None and ctx.needs_input_grad[7 + i]) else None
                for i, s in enumerate(tv_shapes)]
        lib.call("loss_terms_backward", R, N, color, gt, mask, grad_x, D[0], D[1], terms, g_terms[0:1].contiguous(), inv_rays,
                 weights, l2, g_color, g_gx, g_tv[0], g_tv[1], g_prior)
        n_tv = sum(s is not None for s in tv_shapes)
        return (g_color, None, None, g_gx, g_prior, None, None, *g_tv[:n_tv])


def loss_terms(color, color_gt, mask, grad_x, prior, mask_sum_global, N, inv_rays, weights, l2, tvs, N_prior=None):
    """N_prior: the N that divides the five prior sums (python/loss.py:118); None = N."""
    return LossTerms.apply(color, color_gt, mask, grad_x, prior, mask_sum_global,
                           (int(N), float(inv_rays), tuple(weights), bool(l2), int(N if N_prior is None else N_prior)), *tvs)
