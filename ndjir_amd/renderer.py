"""Physically-based volume renderer.

Reference: python/renderer.py -- pb_render :32-209, render_image :212-272.

The reference draws its light-sampling CDF tables and the base-colour perturbation with fixed
seeds inside the graph (`F.rand(seed=412/124/810/108)` :97-98,131-132; `F.randn(seed=913)` :191),
i.e. they are constants of the step.  Here they are explicit inputs (`rand` dict:
diffuse_cdf_the, diffuse_cdf_phi, specular_cdf_the, specular_cdf_phi, noise) so that every
backend sees identical values; `make_rand` draws them with the configured seeds.
"""
import math
import os

import numpy as np
import torch

from . import parameter as P
from .grid_feature import grad as nn_grad
from .helper import generate_all_pixels, generate_raydir_camloc
from .network import (background_network, base_color_network, material_nets_raw, environment_light_network, geometric_network,
                      geometric_network_with_grad,
                      implicit_illumination_network, photogrammetric_light_network, roughness_network,
                      soft_visibility_light_network, specular_reflectance_network)
from .sampler import sample_importance_directions, sample_points, sample_uniform_directions
from .specular_brdf import dot, specular_brdf_model
from .volume import (LIGHT_ACTS, alpha_weights, diffuse_light, direct_light, integrate, integrate_many, material_head, pixel_compose, specular_light,
                     pixel_normal, specular_light_filament)


def make_rand(B, R, conf, device, n_fg=None, include_samples=True):
    """Draw the step's random tensors with the reference's seeds (default.yaml:115-120, 146)."""
    r = conf.renderer
    N0, Nb, nt = r.n_samples0, r.n_bg_samples, r.n_thetas
    N = n_fg if n_fg is not None else r.n_samples0 + r.n_samples1 * r.n_upsamples

    def uni(seed, shape, lo=0.0, hi=1.0):
        return torch.from_numpy(np.random.RandomState(seed).uniform(lo, hi, size=shape).astype(np.float32)).to(device)

    out = dict(
        diffuse_cdf_the=uni(r.diffuse_cdf_the_seed, (B, R, nt)),
        diffuse_cdf_phi=uni(r.diffuse_cdf_phi_seed, (B, R, 2 * nt)),
        specular_cdf_the=uni(r.specular_cdf_the_seed, (B, R, nt)),
        specular_cdf_phi=uni(r.specular_cdf_phi_seed, (B, R, 2 * nt)),
        noise=torch.from_numpy(np.random.RandomState(conf.train.base_color_perturb_seed)
                               .randn(B, R, N, 3).astype(np.float32)).to(device),
    )
    if include_samples:
        out["stratified_sample"] = uni(r.stratified_sample_seed, (B, R, N0, 1))
        out["background_sample"] = uni(r.background_sample_seed, (B, R, Nb + 1, 1), 1e-5, 1.0)
    return out


def redraw_rand(rand, generator=None):
    """Next iteration's random tensors, drawn in place on the device (the reference's `F.rand` / `F.randn` nodes produce
    new numbers at every forward: python/loss.py:40-41, python/renderer.py:97-98, 131-132, 191).  In place, so that a
    captured graph keeps reading the same tensors; call it between replays."""
    for k, t in rand.items():
        if k == "noise":
            t.normal_(generator=generator)
        elif k == "background_sample":
            t.uniform_(1e-5, 1.0, generator=generator)
        else:
            t.uniform_(0.0, 1.0, generator=generator)
    return rand


def pb_render(x_fg, t_fg, x_bg, t_bg, camloc, raydir, mask, cos_anneal_ratio, conf, rand, render_only=False):
    """renderer.py:32-209.
      x_fg (B,R,N,3) requires grad; t_fg (B,R,N+1,1); x_bg (B,R,Nb,4); t_bg (B,R,Nb+1,1);
      camloc (B,3); raydir (B,R,3); mask (B,R,1,1); cos_anneal_ratio (1,)."""
    B, R, N, _ = x_fg.shape
    raydir = raydir.reshape(B, R, 1, 3)
    view_dir = -raydir
    eps_normal = conf.renderer.eps_normal

    # Geometric network and its spatial gradient (nn.grad, renderer.py:51-52)
    sdf_x_fg, feature_x_fg, gain, grad_x_fg, packed_fg = geometric_network_with_grad(x_fg, conf, packed=True)

    # Background alpha (renderer.py:70-76)
    if conf.background_modeling:
        delta_bg = (t_bg[:, :, 1:, :] - t_bg[:, :, :-1, :]).detach()
        alpha_bg, color_bg = background_network(x_bg, view_dir, delta_bg, conf)
    else:
        alpha_bg = torch.ones(B, R, 1, 1, dtype=x_fg.dtype, device=x_fg.device)
        color_bg = torch.full((B, R, 1, 3), conf.background_color, dtype=x_fg.dtype, device=x_fg.device)

    # Foreground alpha, transmittance and weights (renderer.py:55-67, 79-87): one fused launch
    # (csrc/render.hip) instead of ~25 elementwise functions and an exclusive cumprod
    alpha_fg, trans, weights = alpha_weights(sdf_x_fg, grad_x_fg, raydir.reshape(B, R, 3), t_fg, gain, cos_anneal_ratio,
                                             mask, alpha_bg)
    trans_fg = trans[:, :, :N, :]

    def VR(x, off=0):
        """sum_i weights_i x_i over the foreground (off=0) or background (off=N) samples"""
        return integrate(weights, x, off)

    n_thetas = conf.renderer.n_thetas
    M = n_thetas * 2 * n_thetas
    D = feature_x_fg.shape[-1]
    if not any(k.startswith("environment-light-network/") for k in P.get_parameters()):
        # first call only: create the two light nets' parameters HERE, i.e. in the reference's creation order
        # (renderer.py:105-110, before the material nets), so that seeded initialisation does not depend on the evaluation
        # order further down; only the input widths matter
        with torch.no_grad():
            z = torch.zeros((B, 1, 1, 3), dtype=x_fg.dtype, device=x_fg.device)
            environment_light_network(z, conf)
            soft_visibility_light_network(z, z, torch.zeros((B, 1, 1, D), dtype=x_fg.dtype, device=x_fg.device), z, conf)

    # Material nets of the foreground samples (renderer.py:113-128, 164, 186-193).  Default configuration:
    # their output activations, the products feeding the VR integrals and the prior integrands of
    # loss.py:117-166 are ONE fused launch (volume.material_head) followed by ONE VR integral.
    ii = conf.implicit_illumination_network
    pl = conf.photogrammetric_light_network
    sr = conf.specular_reflectance_network
    use_head = (ii.use_me and ii.channels == 1 and ii.act_last == "sigmoid" and pl.use_me and pl.channels == 1
                and not sr.fixme and sr.channels == 3)
    G = conf.geometric_network.voxel.grid_size
    rad = conf.renderer.bounding_sphere_radius
    # (detached: the sample positions carry no parameter gradient, python/sampler.py -- the perturbed pass is first-order in
    # the parameters only, so its encoding and grid query need no d/dx)
    x_fg_ptb = torch.add(x_fg.detach(), rand["noise"], alpha=math.sqrt(3) * 2 * rad / G)
    prior_partials = None
    bg_pixel = None
    if use_head:
        raws = material_nets_raw(x_fg, feature_x_fg, grad_x_fg, conf, packed=packed_fg, photo=(camloc, view_dir))
        raw_photo = None
        if raws is not None and len(raws) == 6:
            raw_imp, raw_bc, raw_rough, raw_spec, raw_photo, photo_gain = raws
        elif raws is not None:
            raw_imp, raw_bc, raw_rough, raw_spec = raws
        else:
            raw_imp = implicit_illumination_network(x_fg, feature_x_fg, grad_x_fg, conf, raw=True)
            raw_bc = base_color_network(x_fg, feature_x_fg, grad_x_fg, conf, raw=True)
            raw_rough = roughness_network(x_fg, feature_x_fg, grad_x_fg, conf, raw=True)
            raw_spec = specular_reflectance_network(x_fg, feature_x_fg, grad_x_fg, conf, raw=True)
        if raw_photo is None:
            raw_photo, photo_gain = photogrammetric_light_network(x_fg, camloc, view_dir, feature_x_fg, grad_x_fg, conf, raw=True)
        if render_only:
            # `render_image` evaluates `color_pixel` only: the reference's graph executor never runs the base-colour
            # perturbation branch there (it feeds the prior term of the loss, python/loss.py:108-115)
            raw_ptb = raw_bc
        else:
            _, feature_ptb, _, packed_ptb = geometric_network(x_fg_ptb, conf, first_order_only=True, packed=True)
            raw_ptb = base_color_network(x_fg_ptb, feature_ptb, None, conf, raw=True, packed=packed_ptb)
        remap = conf.specular_brdf.model == "filament" and conf.specular_brdf.remap
        V, aux, prior_partials = material_head(
            raw_bc, raw_ptb, raw_imp, raw_photo, photo_gain, raw_rough, raw_spec, remap, conf.diffuse_brdf.entangle,
            conf.train.base_color_prior_sym_backward, conf.roughness_network.lower_bound, sr.upper_bound_scale,
            conf.roughness_network.prior_value, sr.prior_value)
        # every VR integral of the ray in one launch (normal, position, feature, material products, background colour)
        if os.environ.get("NDJIR_NO_FUSED_VR"):
            grad_vr, x_vr, feat_vr, pix = VR(grad_x_fg), VR(x_fg), VR(feature_x_fg), VR(V)
        else:
            grad_vr, x_vr, feat_vr, pix, bg_pixel = integrate_many(weights, [grad_x_fg, x_fg, feature_x_fg, V, color_bg],
                                                                   [0, 0, 0, 0, N])
        implicit_pixel, roughness_pixel, spec_refl_pixel = pix[..., 0:1], pix[..., 1:2], pix[..., 2:5]
        photo_pixel, base_term_pixel = pix[..., 5:6], pix[..., 6:9]
        base_color, base_color_ptb = aux[..., 0:3], aux[..., 3:6]
        roughness, spec_refl = V[..., 1:2], V[..., 2:5]
        std_roughness, std_spec_refl = aux[..., 6:7], aux[..., 7:10]
    else:
        grad_vr, x_vr, feat_vr = VR(grad_x_fg), VR(x_fg), VR(feature_x_fg)

    # Normal (renderer.py:90-91)
    if grad_vr.is_cuda and not os.environ.get("NDJIR_NO_FUSED_TAIL"):
        normal_pixel = pixel_normal(grad_vr, eps_normal)
    else:
        grad_pixel = grad_vr + eps_normal
        normal_pixel = grad_pixel / torch.sqrt((grad_pixel * grad_pixel).sum(-1, keepdim=True))
    x_fg_pixel = x_vr.reshape(B, R, 1, 3)
    feature_pixel = feat_vr.reshape(B, R, 1, D)
    normal_bc = normal_pixel[:, :, None, :]

    # Direct light + visibility (renderer.py:103-110)
    uniform_light_dir = sample_uniform_directions(normal_pixel, rand["diffuse_cdf_the"], rand["diffuse_cdf_phi"])

    if not use_head:
        # Implicit light (renderer.py:113-114)
        implicit = implicit_illumination_network(x_fg, feature_x_fg, grad_x_fg, conf)
        implicit_pixel = VR(implicit)
        base_color = base_color_network(x_fg, feature_x_fg, grad_x_fg, conf)
        # Roughness / specular reflectance (renderer.py:123-128)
        roughness, std_roughness = roughness_network(x_fg, feature_x_fg, grad_x_fg, conf)
        roughness_pixel = VR(roughness)
        spec_refl, std_spec_refl = specular_reflectance_network(x_fg, feature_x_fg, grad_x_fg, conf)
        spec_refl_pixel = VR(spec_refl)

    # Light directions of the specular term (renderer.py:131-140)
    if conf.specular_brdf.sampling == "importance":
        imp_dir = sample_importance_directions(normal_pixel, rand["specular_cdf_the"], rand["specular_cdf_phi"],
                                               roughness_pixel)
    else:
        imp_dir = sample_uniform_directions(normal_pixel, rand["specular_cdf_the"], rand["specular_cdf_phi"])

    # Environment light and soft visibility (renderer.py:105-110 and :143-150): the reference evaluates both nets
    # once for the diffuse and once for the specular directions; the 2 M directions go through them in one pass
    # here (row-independent nets: same values; their weight gradients come from one reduction over 2 M lights)
    dirs_all = torch.cat([uniform_light_dir, imp_dir], dim=2)
    sb = conf.specular_brdf
    el, sl = conf.environment_light_network, conf.soft_visibility_light_network
    color_pixel = None
    fused_lights = (use_head and sb.model == "filament" and sb.sampling == "importance" and not sb.use_split_sum
                    and not (ii.use_me and ii.use_me_on_specular) and el.channels in (1, 3) and sl.channels == 1
                    and el.act_last in LIGHT_ACTS and sl.act_last in LIGHT_ACTS and dirs_all.is_cuda
                    and not os.environ.get("NDJIR_NO_FUSED_TAIL") and not os.environ.get("NDJIR_NO_FUSED_LIGHTS"))
    if fused_lights:
        # the default configuration: both nets hand over their raw outputs; output activations, the two light integrals and
        # the pixel composition are one launch each way (csrc/render.hip k_direct_light) -- no (B,R,2M,*) slice, activation
        # or gradient sum exists as a separate launch
        raw_env = environment_light_network(dirs_all, conf, raw=True)
        raw_sv = soft_visibility_light_network(x_fg_pixel, dirs_all, feature_pixel, normal_bc, conf, raw=True)
        color_pixel = direct_light(normal_pixel, view_dir.reshape(B, R, 3), dirs_all, raw_sv, raw_env, pix,
                                   bg_pixel if bg_pixel is not None else VR(color_bg, N),
                                   (LIGHT_ACTS[sl.act_last], LIGHT_ACTS[el.act_last]),
                                   (sl.inverse_black_degree, el.inverse_black_degree, el.upper_bound, conf.renderer.eps_dot, sb.weight),
                                   conf.diffuse_brdf.entangle)
    else:
        env_all = environment_light_network(dirs_all, conf)
        # (per-ray inputs are passed un-broadcast, (B,R,1,*): the net folds them into a per-ray first-layer term)
        soft_vis_all = soft_visibility_light_network(x_fg_pixel, dirs_all, feature_pixel, normal_bc, conf)
        env_d, env = env_all[:, :, :M], env_all[:, :, M:]
        soft_vis_d, soft_vis = soft_vis_all[:, :, :M], soft_vis_all[:, :, M:]

        # Diffuse colour (renderer.py:117-120)
        # mean_m soft_vis * env * clamp(n.l): one fused launch (csrc/render.hip) instead of dot/clamp/mul/mean
        env_pixel = diffuse_light(normal_pixel, uniform_light_dir, soft_vis_d, env_d, conf.renderer.eps_dot)

        # Specular colour (renderer.py:141-161)
        if (sb.model == "filament" and sb.sampling == "importance" and not sb.use_split_sum
                and not (ii.use_me and ii.use_me_on_specular) and env.shape[-1] in (1, 3) and spec_refl_pixel.shape[-1] == 3):
            # BRDF algebra and the light integral fused (csrc/render.hip)
            spec_pixel = specular_light_filament(normal_pixel, view_dir.reshape(B, R, 3), imp_dir, roughness_pixel,
                                                 spec_refl_pixel, soft_vis, env, conf.renderer.eps_dot, sb.weight)
        elif (sb.model in ("filament", "ue4") and sb.sampling in ("importance", "uniform") and imp_dir.is_cuda
              and not (ii.use_me and ii.use_me_on_specular) and env.shape[-1] in (1, 3) and spec_refl_pixel.shape[-1] == 3
              and not os.environ.get("NDJIR_NO_FUSED_BRDF")):
            # ue4 model, uniform sampling, split sum: the same fusion, one templated kernel pair (csrc/render.hip k_specular_light_g)
            spec_pixel = specular_light(normal_pixel, view_dir.reshape(B, R, 3), imp_dir, roughness_pixel, spec_refl_pixel,
                                        soft_vis, env, conf.renderer.eps_dot, sb.weight, sb.model, sb.sampling, sb.use_split_sum)
        else:
            # (stock-op composite: the implicit-illumination term on the specular lobe, 1- or 2-channel reflectance, CPU)
            sBRDF, cos = specular_brdf_model(normal_pixel, view_dir, imp_dir, roughness_pixel, spec_refl_pixel, conf)
            if sb.use_split_sum:
                spec_pixel = (soft_vis * env).mean(dim=2) * (sBRDF * cos).mean(dim=2)
            else:
                spec_pixel = (sBRDF * soft_vis * env * cos).mean(dim=2)
            if ii.use_me and ii.use_me_on_specular:
                spec_pixel = spec_pixel + (sBRDF * implicit_pixel[:, :, :, None]).mean(dim=2)
            spec_pixel = sb.weight * spec_pixel

    # Diffuse + specular composition (renderer.py:163-176)
    if color_pixel is not None:
        pass
    elif use_head and env_pixel.shape[-1] in (1, 3) and spec_pixel.shape[-1] == 3 and not os.environ.get("NDJIR_NO_FUSED_TAIL"):
        # one launch: diffuse = env + implicit, the entangled / disentangled product, + VR(color_bg)
        color_pixel = pixel_compose(pix, env_pixel, spec_pixel, bg_pixel if bg_pixel is not None else VR(color_bg, N),
                                    conf.diffuse_brdf.entangle)
    elif use_head:
        diffuse_light_pixel = env_pixel + implicit_pixel
        if conf.diffuse_brdf.entangle:
            color_fg_pixel = base_term_pixel * diffuse_light_pixel + photo_pixel * spec_pixel
        else:
            color_fg_pixel = photo_pixel * (base_term_pixel * diffuse_light_pixel + spec_pixel)
    elif conf.photogrammetric_light_network.use_me:
        diffuse_light_pixel = env_pixel + implicit_pixel
        photo = photogrammetric_light_network(x_fg, camloc, view_dir, feature_x_fg, grad_x_fg, conf)
        photo_pixel = VR(photo)
        if conf.diffuse_brdf.entangle:
            color_fg_pixel = VR(base_color * photo) * diffuse_light_pixel + photo_pixel * spec_pixel
        else:
            color_fg_pixel = photo_pixel * (VR(base_color) * diffuse_light_pixel + spec_pixel)
    else:
        color_fg_pixel = VR(base_color) + spec_pixel

    if color_pixel is None:
        color_pixel = color_fg_pixel + (bg_pixel if bg_pixel is not None else VR(color_bg, N))

    obj_mask_pred = torch.zeros((), dtype=x_fg.dtype, device=x_fg.device)
    if conf.train.mask_weight > 0.0:
        obj_mask_pred = (alpha_fg * trans_fg).sum(dim=2)

    # Base-colour perturbation (renderer.py:186-193)
    if not use_head:
        if render_only:
            base_color_ptb = base_color
        else:
            _, feature_ptb, _ = geometric_network(x_fg_ptb, conf, first_order_only=True)
            base_color_ptb = base_color_network(x_fg_ptb, feature_ptb, None, conf)

    return dict(prior_partials=prior_partials, color_pixel=color_pixel, sdf_x_fg=sdf_x_fg, grad_x_fg=grad_x_fg, alpha_fg=alpha_fg,
                trans_fg=trans_fg, obj_mask_pred=obj_mask_pred, base_color=base_color,
                base_color_ptb=base_color_ptb, roughness=roughness, specular_reflectance=spec_refl,
                std_roughness=std_roughness, std_specular_reflectance=std_spec_refl)


def render_image(pose, intrinsic, resolution, conf, device=None, progress=None, rank=0, world=1, reduce=True):
    """renderer.py:212-272: tiled forward render of one view.
    pose (1,4,4), intrinsic (1,3,3) numpy; resolution (W, H).  Returns (1,3,H,W) in [0,1].
    world > 1: tiles go round-robin to the ranks (no data-path collective while rendering); `reduce` sums the
    partial images over the ranks at the end (torch.distributed), otherwise the rank's partial image is returned."""
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    scale = 1.0 / 2 ** conf.valid.n_down_samples
    W, H = resolution
    W, H = int(W * scale), int(H * scale)
    P = conf.valid.n_rays
    intrinsic = intrinsic.copy()
    for (i, j) in ((0, 0), (1, 1), (0, 2), (1, 2), (0, 1)):
        intrinsic[:, i, j] = intrinsic[:, i, j] * scale
    _, m = divmod(W * H, P)
    P = P - m   # renderer.py:237-241 (sic: makes P a divisor only in the reference's use cases)

    # The reference builds every tile's rays on the host and copies each tile's colours back; here the camera lives on
    # the device (float64 like the reference's numpy), rays come from the pixel index (ndjir_generate_raydir_camloc),
    # the image is assembled on the device and copied once.  With the fused geometric pass the normals come out of
    # the forward chains themselves, so no autograd graph is recorded.
    import contextlib
    from .helper import generate_raydir_camloc_device
    from .network import uses_fused_geometric
    rand = make_rand(1, P, conf, device)
    cos_anneal_ratio = torch.ones(1, device=device)
    pose_d = torch.from_numpy(np.ascontiguousarray(pose, dtype=np.float64)).to(device)
    K_d = torch.from_numpy(np.ascontiguousarray(intrinsic, dtype=np.float64)).to(device)
    image = torch.zeros((H * W, 3), device=device)
    guard = torch.no_grad() if uses_fused_geometric(conf) else contextlib.nullcontext()
    for p in range(rank * P, H * W, world * P):
        n = min(P, H * W - p)
        idx = torch.arange(p, p + P, device=device, dtype=torch.int32).clamp_(max=H * W - 1).reshape(1, P)   # last tile: padded
        raydir, camloc = generate_raydir_camloc_device(pose_d, K_d, pixel_index=idx, width=W)
        with guard:
            x_fg, t_fg, x_bg, t_bg, mask = sample_points(camloc, raydir, rand["stratified_sample"],
                                                         rand["background_sample"], conf)
            if not isinstance(guard, torch.no_grad):
                x_fg = x_fg.requires_grad_(True)
            res = pb_render(x_fg, t_fg, x_bg, t_bg, camloc, raydir, mask, cos_anneal_ratio, conf, rand, render_only=True)
        image[p:p + n] = res["color_pixel"].detach().reshape(P, 3)[:n]
        if progress is not None:
            progress(p, H * W)
    if world > 1 and reduce:
        import torch.distributed as dist
        dist.all_reduce(image)
    rimage = image.cpu().numpy().astype(np.float64).reshape((1, H * W, 3))
    rimage = rimage.reshape((1, H, W, 3)).transpose((0, 3, 1, 2))
    return np.clip(rimage, 0.0, 1.0)
