"""Point and light-direction sampling.

Reference: python/sampler.py -- SamplePoints :24-314 (t_near_far :71-138, stratified :140-165,
hierarchical SDF-guided importance sampling :167-242, background :244-254, forward :265-299),
SampleDirections :317-408.  Neither op has a gradient (`backward_impl: pass`, :301-302, :391-392).

Random tensors (`stratified_sample`, `background_sample`, cdf tables) are explicit inputs, exactly
as in the reference.
"""
import os

import torch

from . import lib
from .intersection.ray_aabb_intersection import ray_aabb_intersection
from .intersection.ray_sphere_intersection import ray_sphere_intersection
from .network import geometric_network


class SamplePoints:
    """sampler.py:24-314.  `SamplePoints(conf)(camloc, raydir, stratified_sample, background_sample)`
    -> x_fg (B,R,N,3), t_fg (B,R,N+1,1), x_bg (B,R,Nb,4), t_bg (B,R,Nb+1,1), mask (B,R,1,1)."""

    def __init__(self, conf):
        self.conf = conf
        self.record = None  # optional dict filled with per-round (t, sdf, idx) for parity tests

    # -- t_near / t_far --------------------------------------------------------------------
    def t_near_far(self, camloc, raydir):
        m = self.conf.renderer.t_near_far_method
        if m == "intersect_with_r_sphere":
            return self._intersect_with_r_sphere(camloc, raydir)
        if m == "intersect_with_aabb":
            return self._intersect_with_aabb(camloc, raydir)
        if m == "intersect_with_midpoint":
            return self._intersect_with_midpoint(camloc, raydir)
        if m == "intersect_with_camloc_dists":
            return self._intersect_with_camloc_dists(camloc, raydir)
        raise ValueError(f"{m} is not supported.")

    def _intersect_with_r_sphere(self, camloc, raydir):
        t_near, t_far, n_hits = ray_sphere_intersection(camloc, raydir, self.conf.renderer.bounding_sphere_radius)
        return t_near, t_far, (n_hits > 1.0).to(t_near.dtype)

    def _intersect_with_aabb(self, camloc, raydir):
        r = self.conf.renderer.bounding_sphere_radius
        t_near, t_far, n_hits = ray_aabb_intersection(camloc, raydir, [-r, -r, -r], [r, r, r])
        return t_near, t_far, (n_hits > 1.0).to(t_near.dtype)

    def _intersect_with_midpoint(self, camloc, raydir):
        r = self.conf.renderer.bounding_sphere_radius
        B, R, _ = raydir.shape
        b = 2.0 * (camloc.reshape(B, 1, 3) * raydir).sum(-1, keepdim=True)
        mid = -b / 2.0
        return (mid - r).clamp(min=0), mid + r, torch.ones(B, R, 1, dtype=raydir.dtype, device=raydir.device)

    def _intersect_with_camloc_dists(self, camloc, raydir):
        r = self.conf.renderer.bounding_sphere_radius
        B, R, _ = raydir.shape
        d = torch.sqrt((camloc * camloc).sum(-1, keepdim=True))
        t_near = (d - r).reshape(B, 1, 1).expand(B, R, 1)
        t_far = (d + r).reshape(B, 1, 1).expand(B, R, 1)
        return t_near, t_far, torch.ones(B, R, 1, dtype=raydir.dtype, device=raydir.device)

    # -- distances ----------------------------------------------------------------------------
    def sample_stratified_dists(self, t_near, t_far, stratified_sample):
        """sampler.py:140-165: t = tn + (tf - tn)/N (i + u_i)."""
        B, R, _ = t_far.shape
        N = self.conf.renderer.n_samples0
        tn, tf = t_near.reshape(B, R, 1, 1), t_far.reshape(B, R, 1, 1)
        step = (tf - tn) / N
        i = torch.arange(0, N, dtype=tn.dtype, device=tn.device).reshape(1, 1, N, 1)
        return tn + step * (i + stratified_sample)

    def importance_round(self, t, sdf, t_near, t_far, gain, M, with_source=False):
        """One up-sampling round (sampler.py:194-240) given the SDF at the current samples: one HIP
        launch (csrc/sampler.hip), one wave per ray, scan orders fixed by include/ndjir_math.h.
        with_source: also return the M new distances and, per merged position, the slot it came from."""
        B, R, N, _ = t.shape
        t_out = torch.empty((B, R, N + M, 1), device=t.device, dtype=torch.float32)
        idx = torch.empty((B, R, M), device=t.device, dtype=torch.int32)
        src = torch.empty((B, R, N + M), device=t.device, dtype=torch.int32) if with_source else None
        t_new = torch.empty((B, R, M, 1), device=t.device, dtype=torch.float32) if with_source else None
        lib.call("sampler_importance_round", B * R, N, M, float(gain), t.contiguous(), sdf.contiguous(),
                 t_near.expand(B, R, 1, 1).contiguous(), t_far.expand(B, R, 1, 1).contiguous(), t_out, idx, src, t_new)
        if with_source:
            return t_out, idx.long(), src, t_new
        return t_out, idx.long()

    def importance_round_stock(self, t, sdf, t_near, t_far, gain, M):
        """The same round with stock device ops (kept for A/B debugging; not on the product path)."""
        B, R, N, _ = t.shape
        dev, dt = t.device, t.dtype
        ts_end = t[:, :, N - 1:N, :]
        sdf0, sdf1 = sdf[:, :, :-1, :], sdf[:, :, 1:, :]
        t0, t1 = t[:, :, :-1, :], t[:, :, 1:, :]
        sdfm = (sdf0 + sdf1) * 0.5
        cos_val1 = (sdf1 - sdf0) / (t1 - t0 + 1e-5)
        cos_val0 = torch.cat([torch.ones(B, R, 1, 1, dtype=dt, device=dev), cos_val1[:, :, :-1, :]], dim=2)
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
        u = (torch.arange(0, M, dtype=torch.float32, device=dev) / (M - 1 + 1 / M)).to(dt)
        u = u.reshape(1, 1, M).expand(B, R, M).contiguous()
        idx = torch.searchsorted(cumsum_w.contiguous(), u, right=False)
        cumsum_w0 = torch.cat([torch.zeros(B, R, 1, dtype=dt, device=dev), cumsum_w], dim=2)
        denorm = torch.gather(weights, 2, idx.clamp(max=N - 2))
        lower = torch.gather(cumsum_w0, 2, idx)
        ratio = ((u - lower) / denorm).reshape(B, R, M, 1)
        steps = torch.cat([t[:, :, 1:, :] - t[:, :, :-1, :], t_far - ts_end], dim=2)
        steps_idx = torch.gather(steps, 2, idx.unsqueeze(-1))
        ts_idx = torch.gather(t, 2, idx.unsqueeze(-1))
        t_new = torch.maximum(torch.minimum(ts_idx + steps_idx * ratio, t_far), t_near)
        t_all, _ = torch.sort(torch.cat([t, t_new], dim=2), dim=2)
        return t_all, idx

    def sample_importance_dists(self, camloc, raydir, t_near, t_far, t):
        """sampler.py:167-242: U rounds of {SDF at samples -> weights -> inverse-transform -> merge}."""
        B, R, N, _ = t.shape
        M, U = self.conf.renderer.n_samples1, self.conf.renderer.n_upsamples
        c = camloc.reshape(B, 1, 1, 3)
        d = raydir.reshape(B, R, 1, 3)
        tn, tf = t_near.reshape(B, R, 1, 1), t_far.reshape(B, R, 1, 1)
        # The reference re-evaluates the SDF at ALL current samples in every round (sampler.py:192-193).
        # Old samples keep their positions (x = c + t d, same t) and the network is row-independent and
        # deterministic, so their values cannot change: evaluate only the M samples the previous round
        # added and merge them by the round kernel's source map.  Bit-identical, ~4x fewer points.
        sdf, src, t_new = None, None, None
        for u in range(U):
            if sdf is None:
                sdf, _, _ = geometric_network(c + t * d, self.conf, first_order_only=True, sdf_only=True)
            else:
                sdf_new, _, _ = geometric_network(c + t_new * d, self.conf, first_order_only=True, sdf_only=True)
                sdf = torch.gather(torch.cat([sdf, sdf_new], dim=2), 2, src.long().unsqueeze(-1))
            gain = self.conf.renderer.sampling_sigmoid_gain * 2 ** u
            t_in = t
            t, idx, src, t_new = self.importance_round(t, sdf, tn, tf, gain, M, with_source=True)
            if self.record is not None:
                self.record.setdefault("t_in", []).append(t_in.clone())
                self.record.setdefault("sdf", []).append(sdf.clone())
                self.record.setdefault("idx", []).append(idx.clone())
                self.record.setdefault("t_out", []).append(t.clone())
        return t

    def sample_outside_dists(self, t_base, background_sample):
        """sampler.py:244-254."""
        B, R, _ = t_base.shape
        t, _ = torch.sort(t_base.reshape(B, R, 1, 1) / background_sample, dim=2)
        return t

    def _fused_ok(self, raydir):
        """The kernel formulation of the glue around the round kernel (csrc/sampler.hip): device tensors, the fused
        geometric chain for the sdf-only passes, at most 64 background samples; a parity recorder wants the per-round
        tensors of the step-by-step path."""
        from .network import uses_fused_geometric
        c = self.conf
        return (raydir.is_cuda and self.record is None and uses_fused_geometric(c) and c.renderer.n_bg_samples + 1 <= 64
                and c.renderer.n_samples1 <= 32
                and c.renderer.n_samples0 + c.renderer.n_samples1 * c.renderer.n_upsamples <= 256
                and not os.environ.get("NDJIR_NO_FUSED_SAMPLER"))

    def _call_fused(self, camloc, raydir, stratified_sample, background_sample):
        """SamplePoints._forward_impl (sampler.py:265-299) as: intersection, `sampler_begin`, per round {sdf-only chain on
        the new points, `sampler_round_fused`}, `sampler_finish` -- the same arithmetic, expression by expression, as the
        step-by-step path below (tests/test_gpu_sampler.py compares the two bit for bit)."""
        conf = self.conf
        r = conf.renderer
        B, R, _ = raydir.shape
        dev = raydir.device
        N0, M, U, Nb = r.n_samples0, r.n_samples1, r.n_upsamples, r.n_bg_samples
        method = r.t_near_far_method
        if method == "intersect_with_r_sphere":
            t_near, t_far, n_hits = ray_sphere_intersection(camloc, raydir, r.bounding_sphere_radius)
        elif method == "intersect_with_aabb":
            rad = r.bounding_sphere_radius
            t_near, t_far, n_hits = ray_aabb_intersection(camloc, raydir, [-rad] * 3, [rad] * 3)
        else:
            t_near, t_far, _ = self.t_near_far(camloc, raydir)
            t_near, t_far, n_hits = t_near.contiguous(), t_far.contiguous(), None
        cl, rd = camloc.contiguous(), raydir.contiguous()
        BR = B * R
        f = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
        mask, t, x = f(B, R, 1, 1), f(B, R, N0, 1), f(B, R, N0, 3)
        lib.call("sampler_begin", BR, N0, R, cl, rd, t_near, t_far, n_hits, stratified_sample.contiguous(), mask, t, x)
        sdf_prev, sdf_new, src, N = None, None, None, N0
        for u in range(U):
            s_new, _, _ = geometric_network(x, conf, first_order_only=True, sdf_only=True)     # (B,R,N0|M,1)
            gain = r.sampling_sigmoid_gain * 2 ** u
            t_out, idx = f(B, R, N + M, 1), torch.empty((B, R, M), device=dev, dtype=torch.int32)
            src_out, t_new, x = torch.empty((B, R, N + M), device=dev, dtype=torch.int32), f(B, R, M, 1), f(B, R, M, 3)
            if u == 0:
                sdf_cur = s_new
                lib.call("sampler_round_fused", BR, N, M, float(gain), t, sdf_cur, N, None, None, None, t_near, t_far, cl, rd, R,
                         t_out, idx, src_out, t_new, x)
            else:
                sdf_cur = f(B, R, N, 1)
                lib.call("sampler_round_fused", BR, N, M, float(gain), t, sdf_prev, N - M, s_new, src, sdf_cur, t_near, t_far, cl, rd,
                         R, t_out, idx, src_out, t_new, x)
            sdf_prev, src, t, N = sdf_cur, src_out, t_out, N + M
        x_fg, t_fg = f(B, R, N, 3), f(B, R, N + 1, 1)
        if conf.background_modeling:
            x_bg, t_bg = f(B, R, Nb, 4), f(B, R, Nb + 1, 1)
            lib.call("sampler_finish", BR, N, Nb, R, float(r.bounding_sphere_radius), cl, rd, t, t_far, mask,
                     background_sample.contiguous(), x_fg, t_fg, x_bg, t_bg)
        else:
            lib.call("sampler_finish", BR, N, Nb, R, float(r.bounding_sphere_radius), cl, rd, t, t_far, None, None, x_fg, t_fg,
                     None, None)
            x_bg = torch.ones(B, R, Nb, 4, dtype=t.dtype, device=dev)
            t_bg = torch.ones(B, R, Nb + 1, 1, dtype=t.dtype, device=dev)
        return x_fg, t_fg, x_bg, t_bg, mask

    def __call__(self, camloc, raydir, stratified_sample, background_sample):
        with torch.no_grad():
            if self._fused_ok(raydir):
                return self._call_fused(camloc, raydir, stratified_sample, background_sample)
            conf = self.conf
            B, R, _ = raydir.shape
            t_near, t_far, mask = self.t_near_far(camloc, raydir)
            t = self.sample_stratified_dists(t_near, t_far, stratified_sample)
            t = self.sample_importance_dists(camloc, raydir, t_near, t_far, t)
            c = camloc.reshape(B, 1, 1, 3)
            d = raydir.reshape(B, R, 1, 3)
            x_fg = c + t * d
            t_fg = torch.cat([t, t_far.reshape(B, R, 1, 1)], dim=2)
            Nb = conf.renderer.n_bg_samples
            if conf.background_modeling:
                tn_bg, _, _ = self._intersect_with_camloc_dists(camloc, raydir)
                t_base = t_far * mask + tn_bg * (1 - mask)
                t_bg = self.sample_outside_dists(t_base, background_sample)
                xb = c + t_bg[:, :, :-1, :] * d
                dists = torch.sqrt((xb * xb).sum(-1, keepdim=True)) + 1e-6
                x_bg = torch.cat([xb / dists, 1.0 / dists], dim=-1)
            else:
                x_bg = torch.ones(B, R, Nb, 4, dtype=t.dtype, device=t.device)
                t_bg = torch.ones(B, R, Nb + 1, 1, dtype=t.dtype, device=t.device)
            return x_fg, t_fg, x_bg, t_bg, mask.reshape(B, R, 1, 1)


def sample_points(camloc, raydir, stratified_sample, background_sample, conf, record=None):
    func = SamplePoints(conf)
    func.record = record
    return func(camloc, raydir, stratified_sample, background_sample)


class SampleDirections:
    """sampler.py:317-408: hemisphere directions around the pixel normal, uniform-cos(theta) or
    GGX-importance (when `alpha` is given); kernels of csrc/sampling/inverse_transform_cuda.cu."""

    def __init__(self, eps=0.0):
        self._eps = eps

    def __call__(self, normal, cdf_the, cdf_phi, alpha=None):
        B, R, _ = normal.shape
        n_thes, n_phis = cdf_the.shape[-1], cdf_phi.shape[-1]
        M = n_thes * n_phis
        n = normal.detach().contiguous()
        ct = cdf_the.detach().contiguous()
        cp = cdf_phi.detach().contiguous()
        out = torch.empty((B, R, M, 3), device=n.device, dtype=torch.float32)
        if alpha is None:
            lib.call("inverse_transform_sample_uniform_directions", B * R * M, out, n, ct, cp,
                     B * R, M, n_thes, n_phis, self._eps)
        else:
            lib.call("inverse_transform_sample_importance_directions", B * R * M, out, n, ct, cp,
                     alpha.detach().contiguous(), B * R, M, n_thes, n_phis, self._eps)
        return out


def sample_uniform_directions(normal, cdf_the, cdf_phi, eps=0.0):
    return SampleDirections(eps)(normal, cdf_the, cdf_phi)


def sample_importance_directions(normal, cdf_the, cdf_phi, alpha, eps=0.0):
    return SampleDirections(eps)(normal, cdf_the, cdf_phi, alpha)
