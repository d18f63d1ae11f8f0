"""Training loss of the hot path.

Reference: python/loss.py:27-192 (total_loss).  The stratified / background random samples that
the reference draws with fixed seeds (:40-41) are explicit inputs in `rand` (see
renderer.make_rand).
"""
import torch

from . import functions as F
from . import parameter as P
from .grid_feature import (total_variation_loss, total_variation_loss_on_triline,  # noqa: F401
                           total_variation_loss_on_triplane, total_variation_loss_on_voxel_hash)
from .renderer import pb_render
from .sampler import sample_points


import os
_NO_FUSED_TAIL = bool(os.environ.get("NDJIR_NO_FUSED_TAIL"))       # A/B switch: the stock-op formulation of the loss terms


def total_loss(camloc, raydir, color_gt, obj_mask, cos_anneal_ratio, conf, rand, record=None, ray_shards=1,
               mask_sum_global=None, obj_mask_sum_global=None):
    """camloc (B,3), raydir (B,R,3), color_gt (B,R,3), obj_mask (B,R,1) or None, cos_anneal_ratio (1,).

    `ray_shards` > 1: this process holds one of `ray_shards` equal slices of the ray batch
    (torch.distributed initialised).  The loss normalisers -- B*R and sum(mask) -- are then taken
    over ALL shards (one scalar all-reduce before backward), so the sum over ranks of the returned
    losses, and of their gradients, equals the single-process loss / gradient of the whole batch.
    `mask_sum_global` / `obj_mask_sum_global`: the all-reduced sum(mask) / sum(obj_mask), if the caller has them
    already (0-d tensors)."""
    B, R, _ = color_gt.shape
    tr = conf.train

    # Points on ray (loss.py:40-43)
    x_fg, t_fg, x_bg, t_bg, mask = sample_points(camloc, raydir, rand["stratified_sample"],
                                                 rand["background_sample"], conf, record)
    x_fg = x_fg.requires_grad_(True)

    res = pb_render(x_fg, t_fg, x_bg, t_bg, camloc, raydir, mask, cos_anneal_ratio, conf, rand)
    zero = torch.zeros((), dtype=x_fg.dtype, device=x_fg.device)
    N = x_fg.shape[2]
    # python/loss.py:36 binds N = n_samples0 and :72 rebinds it to the sample count only inside `if eikonal_weight > 0`;
    # the priors' normaliser (:118) uses whichever N is bound by then
    N_prior = N if tr.eikonal_weight > 0.0 else conf.renderer.n_samples0

    # The default structure (material head with its per-ray prior sums, no object-mask term): every term below and the
    # weighted total are ONE pass over the rays plus a fixed-order final reduction (csrc/loss.hip), and as much in
    # backward, instead of ~40 elementwise / reduction launches each way.
    tv_feats = [(name, f) for name, f in P.get_parameters().items() if name.endswith("feature/F")] \
        if (conf.geometric_network.voxel.type != "none" and tr.tv_weight > 0.0) else []
    if (x_fg.is_cuda and tr.mask_weight == 0.0 and res.get("prior_partials") is not None and len(tv_feats) <= 2
            and tr.rgb_loss in ("l1", "l2") and not _NO_FUSED_TAIL):
        from .volume import LOSS_TERM_NAMES, loss_terms
        tv_loss_map = dict(voxel_feature=F.tv_loss_on_voxel, voxel_hash_feature=F.tv_loss_on_voxel_hash,
                           triplane_feature=F.tv_loss_on_triplane, triline_feature=F.tv_loss_on_triline)
        tvs = [tv_loss_map[name.split("/")[-2]](x_fg.detach(), f, sym_backward=tr.tv_sym_backward) for name, f in tv_feats]
        msg = None
        if mask_sum_global is not None:
            msg = mask_sum_global
        elif ray_shards > 1:
            import torch.distributed as dist
            msg = mask.sum()
            dist.all_reduce(msg)
        terms = loss_terms(res["color_pixel"], color_gt, mask, res["grad_x_fg"] if tr.eikonal_weight > 0.0 else None,
                           res["prior_partials"], msg, N, 1.0 / (B * R * ray_shards),
                           (tr.eikonal_weight, tr.tv_weight, tr.base_color_prior_weight, tr.roughness_prior_weight,
                            tr.specular_reflectance_prior_weight), tr.rgb_loss == "l2", tvs, N_prior=N_prior)
        # only the total carries a gradient (the kernel pair differentiates terms[0]); the reported terms are detached
        out = {k: (terms[i] if i == 0 else terms[i].detach()) for i, k in enumerate(LOSS_TERM_NAMES)}
        # (a term whose weight is zero is reported as zero, as the reference does not evaluate it)
        for k, w in (("loss_eikonal", tr.eikonal_weight), ("prior_base_color", tr.base_color_prior_weight),
                     ("prior_roughness", tr.roughness_prior_weight), ("reg_std_roughness", tr.roughness_prior_weight),
                     ("prior_specular_reflectance", tr.specular_reflectance_prior_weight),
                     ("reg_std_specular_reflectance", tr.specular_reflectance_prior_weight)):
            if not w > 0.0:
                out[k] = zero
        if not tvs:
            out["loss_tv"] = zero
        out.update(loss_mask=zero, render=res, samples=dict(x_fg=x_fg, t_fg=t_fg, x_bg=x_bg, t_bg=t_bg, mask=mask))
        return out

    # RGB loss (loss.py:59-66)
    color = res["color_pixel"]
    err = (color - color_gt).abs() if tr.rgb_loss == "l1" else (color - color_gt) ** 2
    if tr.mask_weight > 0.0:
        if obj_mask is None:
            raise ValueError("train.mask_weight > 0 needs obj_mask (python/loss.py:62)")
        obj_sum = obj_mask.sum()
        if obj_mask_sum_global is not None:
            obj_sum = obj_mask_sum_global.reshape(()).to(obj_sum.dtype)
        elif ray_shards > 1:       # normalise by the GLOBAL count, like every other term, so that per-rank gradients add
            import torch.distributed as dist
            obj_sum = obj_sum.clone()
            dist.all_reduce(obj_sum)
        loss_rgb = (err * obj_mask).sum() / (obj_sum + 1e-5)
    else:
        loss_rgb = err.sum() / (B * R * ray_shards)

    mask_sum = mask.sum()
    if mask_sum_global is not None:
        # the caller already summed the ray masks over all shards (keeps this function free of collectives, so
        # that a whole step can be captured into a HIP graph; the mask depends on the rays only)
        mask_sum = mask_sum_global.reshape(()).to(mask_sum.dtype)
    elif ray_shards > 1:
        import torch.distributed as dist
        mask_sum = mask_sum.clone()
        dist.all_reduce(mask_sum)
    denorm = mask_sum * N + 1e-5
    denorm_prior = mask_sum * N_prior + 1e-5

    # Eikonal loss (loss.py:68-76)
    loss_eikonal = zero
    if tr.eikonal_weight > 0.0:
        g = res["grad_x_fg"]
        gn = torch.sqrt((g * g).sum(-1, keepdim=True))
        loss_eikonal = (((gn - 1) * mask) ** 2.0).sum() / denorm

    # TV loss over every `.../<grid>_feature/F` parameter (loss.py:78-105)
    loss_tv = zero
    tv_loss_map = dict(voxel_feature=F.tv_loss_on_voxel, voxel_hash_feature=F.tv_loss_on_voxel_hash,
                       triplane_feature=F.tv_loss_on_triplane, triline_feature=F.tv_loss_on_triline)
    if conf.geometric_network.voxel.type != "none" and tr.tv_weight > 0.0:
        for name, feature in P.get_parameters().items():
            if not name.endswith("feature/F"):
                continue
            fn = tv_loss_map[name.split("/")[-2]]
            tv = fn(x_fg.detach(), feature, sym_backward=tr.tv_sym_backward)
            loss_tv = loss_tv + (tv * mask).sum() / denorm

    # Mask loss (loss.py:107-115)
    loss_mask = zero
    if tr.mask_weight > 0.0:
        pred = res["obj_mask_pred"].clamp(1e-3, 1.0 - 1e-3)
        bce = -(obj_mask * torch.log(pred) + (1 - obj_mask) * torch.log(1 - pred))
        loss_mask = bce.sum() / (mask_sum + 1e-5)

    # Priors (loss.py:117-166)
    pp = res.get("prior_partials")
    if pp is not None:
        # per-ray sums of the five integrands from the fused material head (volume.material_head)
        pm = (pp * mask.reshape(B, R, 1)).sum(dim=(0, 1)) / denorm_prior
    prior_base_color = zero
    if pp is not None:
        if tr.base_color_prior_weight > 0.0:
            prior_base_color = pm[0]
    elif tr.base_color_prior_weight > 0.0:
        bc = res["base_color"] if tr.base_color_prior_sym_backward else res["base_color"].detach()
        prior_base_color = ((bc - res["base_color_ptb"]).abs() * mask).sum() / denorm_prior

    prior_roughness = reg_std_roughness = zero
    if pp is not None:
        if tr.roughness_prior_weight > 0.0:
            prior_roughness, reg_std_roughness = pm[1], pm[2]
    elif tr.roughness_prior_weight > 0.0:
        pr = (res["roughness"] - conf.roughness_network.prior_value).abs() / res["std_roughness"]
        prior_roughness = (pr * mask).sum() / denorm_prior
        reg_std_roughness = (torch.log(res["std_roughness"]).clamp(1e-5, 1e5) * mask).sum() / denorm_prior

    prior_spec = reg_std_spec = zero
    if pp is not None:
        if tr.specular_reflectance_prior_weight > 0.0:
            prior_spec, reg_std_spec = pm[3], pm[4]
    elif tr.specular_reflectance_prior_weight > 0.0:
        ps = (res["specular_reflectance"] - conf.specular_reflectance_network.prior_value).abs() \
            / res["std_specular_reflectance"]
        prior_spec = (ps * mask).sum() / denorm_prior
        reg_std_spec = (torch.log(res["std_specular_reflectance"]).clamp(1e-5, 1e5) * mask).sum() / denorm_prior

    loss = (loss_rgb + tr.eikonal_weight * loss_eikonal + tr.tv_weight * loss_tv + tr.mask_weight * loss_mask
            + tr.base_color_prior_weight * prior_base_color + tr.roughness_prior_weight * prior_roughness
            + tr.specular_reflectance_prior_weight * prior_spec + tr.roughness_prior_weight * reg_std_roughness
            + tr.specular_reflectance_prior_weight * reg_std_spec)

    return dict(loss=loss, loss_rgb=loss_rgb, loss_eikonal=loss_eikonal, loss_tv=loss_tv, loss_mask=loss_mask,
                prior_base_color=prior_base_color, prior_roughness=prior_roughness,
                prior_specular_reflectance=prior_spec, reg_std_roughness=reg_std_roughness,
                reg_std_specular_reflectance=reg_std_spec, render=res,
                samples=dict(x_fg=x_fg, t_fg=t_fg, x_bg=x_bg, t_bg=t_bg, mask=mask))
