"""Optimizer step of the training loop on the HIP solver kernels.

Mirror of the reference's `Solvers` (python/solver.py:20-119): two Adam solvers -- grid features
(parameter names ending in "feature/F") and everything else -- with the same method names, the
learning-rate / cos-anneal / light-visibility-gain schedules, and the call order of
python/train.py:136-148.  `Adam` mirrors the subset of nnabla's `S.Adam` the reference uses
(nnabla 1.29.0 is not vendored; update rule as published, see include/ndjir_hip.h).

What is different from the reference, with identical results:
  * `weight_decay()` does not run a dense `g += d w` pass over the 2 GiB grid: the rate is folded into
    the update kernel (g_total = dL/dw + d w).  `clip_grad_by_norm()` -- which the reference applies to the
    decay-only gradient, python/train.py:138-139 -- becomes a scale of that rate from one ||w|| reduction.
  * `zero_grad()` of a grid is folded into the same kernel (the accumulate-in-place gradient buffer is
    cleared while it is read), so a training step reads and writes every grid cell exactly once.
  * The guard (`check_inf_or_nan_grad`) inspects the MLP gradients and the grid cells the step touched and
    leaves its verdict on the device; learning rate, nnabla's step counter t and alpha_t live there too
    (`ndjir_adam_state`), so `guarded_update()` needs no host synchronisation and can be captured in a HIP
    graph.  `check_inf_or_nan_grad()` (a host bool, as in the reference) is kept for API parity.
There is no CPU path: the kernels come from libndjir_hip.so.
"""
import math

import numpy as np
import torch

from . import lib
from . import parameter as P
from .grid_feature import get_grad_buffer


class Adam:
    """nnabla `S.Adam(alpha, beta1, beta2, eps)` on torch GPU tensors."""

    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        self.beta1, self.beta2, self.eps = float(beta1), float(beta2), float(eps)
        self._alpha = float(alpha)
        self.names, self.params, self.m, self.v = [], [], [], []
        self.state = None            # device ndjir_adam_state {alpha, t, alpha_t, skipped} as 4 x int32/float32
        self.flag = None             # device int: the guard found an inf / nan
        self._decay = 0.0
        self._decay_scales = None    # per-parameter factors from clip_grad_by_norm (None: no clipping)
        self._touched = {}           # name -> list of query tensors whose cells hold this step's grid gradient
        self._bitmaps = {}           # name -> touched-cell bitmap of a voxel grid (all zero between updates)
        self._grads = None

    # -- nnabla API ------------------------------------------------------------------------------------------------
    def set_parameters(self, params):
        for k, p in params.items():
            if k in self.names:
                continue
            assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous(), k
            self.names.append(k)
            self.params.append(p)
            self.m.append(torch.zeros_like(p))
            self.v.append(torch.zeros_like(p))
        if self.state is None and self.params:
            dev = self.params[0].device
            self.state = torch.zeros(4, dtype=torch.float32, device=dev)
            self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
            self.set_learning_rate(self._alpha)

    def get_parameters(self):
        return dict(zip(self.names, self.params))

    def set_learning_rate(self, lr):
        self._alpha = float(lr)
        if self.state is not None:
            self.state[0:1].fill_(float(np.float32(lr)))

    def learning_rate(self):
        return self._alpha

    def weight_decay(self, rate):
        """Reference: g += rate * w, immediately.  Here: folded into the next update."""
        self._decay = float(rate)

    def clip_grad_by_norm(self, clip):
        """Called between weight_decay() and backward (python/train.py:138-139), so what it clips is the decay-only
        gradient rate * w of each parameter: g *= clip / ||g|| when ||g|| > clip.  Returned as per-parameter scales
        of the decay rate (host-synchronising; the shipped configs keep clip_grad_norm = 0)."""
        scales = []
        for p in self.params:
            acc = torch.zeros(1, dtype=torch.float64, device=p.device)
            lib.call("solver_sum_squares", p.numel(), p.detach(), acc)
            norm = abs(self._decay) * math.sqrt(float(acc.item()))
            scales.append(clip / norm if norm > clip else 1.0)
        self._decay_scales = scales

    def zero_grad(self):
        """Dense gradients are produced fresh by autograd each step; grid gradient buffers are cleared by the
        update kernel itself.  Only forgets the previous step's gradient references."""
        self._grads = None
        self._touched = {}
        self._decay = 0.0
        self._decay_scales = None

    def set_gradients(self, grads, touched=None):
        """grads: {name: tensor or None} for dense parameters (what backward produced); grid parameters use
        their accumulate-in-place buffer (`grid_feature.set_grad_buffer`).  touched: {name: [query tensors]} -- the
        points whose cells can hold gradient (restricts the guard and the gradient read of the update to the 8 corner
        cells of each point: LINEAR dense voxel grids only -- a cosine / Lanczos grid scatters into more cells and must
        be passed without `touched`; absent = dense check and dense update)."""
        self._grads = grads
        self._touched = touched or {}

    def _grad_of(self, k, p):
        buf = get_grad_buffer(p)
        if buf is not None:
            return buf, True
        g = self._grads.get(k) if self._grads is not None else p.grad
        return (g.contiguous() if g is not None else None), False

    def _check(self):
        if not self.params:          # (a configuration without this solver's parameter group: `no_voxel` has no feature grid)
            return
        self.flag.zero_()
        dense, numel = [], []
        for k, p in zip(self.names, self.params):
            g, is_buf = self._grad_of(k, p)
            if g is None:
                continue
            if is_buf and k in self._touched and p.dim() == 4:
                for q in self._touched[k]:
                    q = q.detach().reshape(-1, 3).contiguous()
                    lib.call("voxel_feature_check_touched", q.shape[0], g, q, list(g.shape[:3]), g.shape[3],
                             [-1, -1, -1], [1, 1, 1], self.flag)
            elif g.numel() >= (1 << 20):
                lib.call("solver_check_inf_or_nan", g.numel(), g, self.flag)
            else:
                dense.append(g)
                numel.append(g.numel())
        if dense:
            lib.call("solver_check_inf_or_nan_multi", len(dense), dense, numel, self.flag)

    def check_inf_or_nan_grad(self):
        self._check()
        return bool(self.flag.item())

    def update(self, guard_flags=None, repack=True):
        """One Adam step.  guard_flags: (flag_a, flag_b) device ints -- the update is skipped on the device when both
        are raised (python/solver.py:67-69); None = unconditional, as nnabla's `update()`.  repack: refresh the tracked
        packed weights afterwards (`Solvers` does it once for both of its solvers instead: a hash grid's 2-D feature table
        would otherwise trigger a second launch)."""
        if not self.params:
            return
        fa, fb = guard_flags if guard_flags is not None else (None, None)
        lib.call("solver_adam_begin", self.state, self.beta1, self.beta2, fa, fb)
        scales = self._decay_scales
        small = []
        for i, (k, p) in enumerate(zip(self.names, self.params)):
            g, is_buf = self._grad_of(k, p)
            decay = self._decay * (scales[i] if scales else 1.0)
            if g is not None and is_buf and k in self._touched and p.dim() == 4 and p.shape[-1] == 4 and p.numel() % 128 == 0:
                # voxel grid whose gradient lives in the cells this step's samples touched: mark them, then the update
                # reads / clears g only there
                bm = self._bitmaps.get(k)
                if bm is None:
                    bm = self._bitmaps[k] = torch.zeros(p.numel() // 128, dtype=torch.int32, device=p.device)
                for q in self._touched[k]:
                    q = q.detach().reshape(-1, 3).contiguous()
                    lib.call("voxel_feature_mark_touched", q.shape[0], q, list(p.shape[:3]), 4, [-1, -1, -1], [1, 1, 1], bm)
                lib.call("solver_adam_touched", p.numel(), p.detach(), g, self.m[i], self.v[i], 0.0, self.beta1, self.beta2,
                         self.eps, decay, bm, self.state)
            elif g is not None and (is_buf or p.numel() >= (1 << 20)) and p.numel() % 4 == 0:
                lib.call("solver_adam", p.numel(), p.detach(), g, self.m[i], self.v[i], 0.0, self.beta1, self.beta2,
                         self.eps, decay, 1 if is_buf else 0, self.state)
            else:
                small.append((p, g, self.m[i], self.v[i], decay))
        # dense small tensors: grouped by decay value (one value unless clipping is on)
        for d in sorted({s[4] for s in small}):
            grp = [s for s in small if s[4] == d]
            lib.call("solver_adam_multi", len(grp), [s[0].detach() for s in grp], [s[1] for s in grp],
                     [s[2] for s in grp], [s[3] for s in grp], [s[0].numel() for s in grp], 0.0, self.beta1,
                     self.beta2, self.eps, d, self.state)
        for p in self.params:            # kernels wrote the parameters behind autograd's back: packed-weight caches etc.
            torch.autograd.graph.increment_version(p)
        if repack and any(p.dim() == 2 for p in self.params):
            # the MLP weights' packed (MFMA fragment order) copies: ONE launch over every tracked (weight, orientation)
            # instead of one pack launch per weight and orientation at their next use (ndjir_amd/mlp.py `track_weights`)
            from . import mlp
            mlp.repack_tracked()

    # -- introspection ---------------------------------------------------------------------------------------------
    def step_count(self):
        return int(self.state.view(torch.int32)[1].item())

    def skipped(self):
        return bool(self.state.view(torch.int32)[3].item())


class Solvers:
    """python/solver.py:20-119."""

    def __init__(self, conf, **adam):
        B, R = conf.train.batch_size, conf.train.n_rays
        self.learning_rate_weight = conf.train.base_learning_rate_weight * (B * R) / (1 * 512)
        self.learning_rate_feat = conf.train.base_learning_rate_feat * (B * R) / (1 * 512)
        self.solver_weight = Adam(0, **adam)
        self.solver_feat = Adam(0, **adam)
        self.conf = conf

    def set_parameters(self):
        params = P.get_parameters(grad_only=True)
        self.solver_weight.set_parameters({k: p for k, p in params.items() if not k.endswith("feature/F")})
        self.solver_feat.set_parameters({k: p for k, p in params.items() if k.endswith("feature/F")})

    def weight_decay(self):
        self.solver_weight.weight_decay(self.conf.train.weight_decay)
        self.solver_feat.weight_decay(self.conf.train.weight_decay)

    def clip_grad_by_norm(self):
        if self.conf.train.clip_grad_norm <= 0:
            return
        self.solver_weight.clip_grad_by_norm(self.conf.train.clip_grad_norm)
        self.solver_feat.clip_grad_by_norm(self.conf.train.clip_grad_norm)

    def set_gradients(self, grads, touched=None):
        self.solver_weight.set_gradients(grads)
        self.solver_feat.set_gradients(grads, touched)

    def update(self):
        self.solver_weight.update(repack=False)
        self.solver_feat.update(repack=False)
        self._repack()

    @staticmethod
    def _repack():
        """ONE re-pack launch for the tracked packed weights after both solvers' updates (ndjir_amd/mlp.py `repack_tracked`)."""
        from . import mlp
        mlp.repack_tracked()

    def guarded_update(self, loss=None):
        """`if check_inf_or_nan_grad(): continue`, `if isnan(loss): continue`, `update()` (python/train.py:141-148)
        without leaving the stream.  loss: the step's loss as a 1-element device tensor (None: gradient guard only)."""
        self.solver_weight._check()
        self.solver_feat._check()
        flags = [self.solver_weight.flag, self.solver_feat.flag]
        if flags[0] is None or flags[1] is None:
            # A solver without parameters finds no inf / nan (the reference's check over an empty set, python/solver.py:67-69):
            # its flag is a zero of our own, on the other solver's device.
            live = flags[0] if flags[0] is not None else flags[1]
            if live is None:
                return
            if getattr(self, "_zero_flag", None) is None or self._zero_flag.device != live.device:
                self._zero_flag = torch.zeros(1, dtype=torch.int32, device=live.device)
            self._zero_flag.zero_()
            flags = [f if f is not None else self._zero_flag for f in flags]
        flags = tuple(flags)
        if loss is not None:        # a NaN loss vetoes the step whatever the gradients look like: raise both flags
            lib.call("solver_veto_if_nan", 1, loss.detach().reshape(1), flags[0], flags[1])
        self.solver_weight.update(flags, repack=False)
        self.solver_feat.update(flags, repack=False)
        self._repack()

    def zero_grad(self):
        self.solver_weight.zero_grad()
        self.solver_feat.zero_grad()

    def check_inf_or_nan_grad(self):
        # sic: `and` (python/solver.py:67-69)
        return self.solver_weight.check_inf_or_nan_grad() and self.solver_feat.check_inf_or_nan_grad()

    def update_learning_rate(self, i):
        self.solver_weight.set_learning_rate(self.compute_learning_rate(i, self.learning_rate_weight))
        self.solver_feat.set_learning_rate(self.compute_learning_rate(i, self.learning_rate_feat))
        self.update_cos_anneal_ratio(i)
        self.update_light_visibility_gain(i)

    def compute_learning_rate(self, i, lr):
        """Linear warm-up over int(epoch * warmup_term_ratio) epochs, then a cosine down to
        learning_rate_end_ratio * lr at the last epoch (python/solver.py:82-98)."""
        t = self.conf.train
        warm = int(t.epoch * t.warmup_term_ratio)
        if warm < 1:
            warm = 0
        if i < warm:
            return lr * i / warm
        phase = np.pi * (i - warm) / (t.epoch - warm)
        amp = (1 - t.learning_rate_end_ratio) * lr / (1 + np.cos(np.pi * warm / t.epoch))
        return np.cos(phase) * amp + (amp + t.learning_rate_end_ratio * lr)

    def update_cos_anneal_ratio(self, i):
        """python/solver.py:100-108 (parameter "cos_anneal_ratio", not trainable)."""
        t = self.conf.train
        x = i / (t.epoch * t.cos_anneal_term_ratio)
        ratio = 0.5 * np.cos(np.pi * x) + 0.5 if x < 1.0 else 1.0
        car = P.get_parameter_or_create("cos_anneal_ratio", (1,), np.asarray([0.0]), False)
        car.fill_(float(ratio))

    def update_light_visibility_gain(self, i):
        """python/solver.py:110-119 (parameter "photogrammetric-light-network/gain", not trainable)."""
        t = self.conf.train
        hi = t.sigmoid_gain_lv_end
        b = (hi + 1) * 0.5
        g = (1 - b) * np.cos(np.pi * i / t.epoch) + b
        gain = P.get_parameter_or_create("photogrammetric-light-network/gain", (1,), np.asarray([1.0]), False)
        gain.fill_(float(g))
