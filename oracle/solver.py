"""Numpy restatement of the reference's optimizer step.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  PARITY UNPINNED: the reference holds no test or
golden value for its solver, and the arithmetic lives in a dependency that is not vendored
(nnabla 1.29.0, `README.md:29`, `docker/Dockerfile:74-75`).  What is restated:

  * nnabla `S.Adam` as published in its solver documentation
        m_t = b1 m_{t-1} + (1 - b1) g_t
        v_t = b2 v_{t-1} + (1 - b2) g_t^2
        alpha_t = alpha sqrt(1 - b2^t) / (1 - b1^t)
        w_{t+1} = w_t - alpha_t m_t / (sqrt(v_t) + eps)            (eps OUTSIDE the bias correction)
    defaults alpha=1e-3, b1=0.9, b2=0.999, eps=1e-8; t is a per-solver counter starting at 0 and
    incremented by every `update()`; `weight_decay(d)`: g += d w; `clip_grad_by_norm(c)`: per
    parameter, g *= c / ||g|| when ||g|| > c; `check_inf_or_nan_grad()`: any inf / nan.
  * the reference's use of them: python/solver.py:20-69 (two solvers: names ending in "feature/F"
    vs everything else; learning rates scaled by B R / 512), the schedules python/solver.py:71-119,
    and the order of calls in python/train.py:136-148:
        forward; zero_grad; weight_decay; clip_grad_by_norm; backward (accumulates); guard; update
    i.e. the decay term is written into the zeroed gradient first, clipped alone, and the loss
    gradient is added on top.

Cross-check available without nnabla (tests/test_oracle_solver_cpu.py): with eps = 0 the rule
coincides with torch.optim.Adam's (whose eps sits inside the bias correction), which pins the
recurrences and the bias correction independently.

Expression order is fixed (left to right, no fused multiply-add) so that the HIP kernel
(ndjir_amd/csrc/solver.hip, compiled with contraction off) can be compared bit for bit in fp32.
"""
import math

import numpy as np


class Adam:
    """nnabla `S.Adam` on numpy arrays (dtype of the arrays: float32 or float64)."""

    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        self.alpha, self.beta1, self.beta2, self.eps = alpha, beta1, beta2, eps
        self.params, self.grads, self.m, self.v = {}, {}, {}, {}
        self.t = 0

    def set_parameters(self, params):
        for k, p in params.items():
            self.params[k] = p
            self.grads[k] = np.zeros_like(p)
            self.m[k] = np.zeros_like(p)
            self.v[k] = np.zeros_like(p)

    def set_learning_rate(self, lr):
        self.alpha = lr

    def zero_grad(self):
        for g in self.grads.values():
            g[...] = 0

    def weight_decay(self, rate):
        for k, p in self.params.items():
            dt = p.dtype.type
            self.grads[k] += dt(rate) * p

    def clip_grad_by_norm(self, clip):
        for k, g in self.grads.items():
            norm = math.sqrt(float(np.sum(g.astype(np.float64) ** 2)))
            if norm > clip:
                g *= g.dtype.type(clip / norm)

    def check_inf_or_nan_grad(self):
        return any(not np.all(np.isfinite(g)) for g in self.grads.values())

    def alpha_t(self):
        # double arithmetic (std::pow(float, uint32) promotes), one rounding to the working type
        b1, b2 = float(np.float32(self.beta1)), float(np.float32(self.beta2))
        return float(np.float32(self.alpha)) * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)

    def update(self):
        self.t += 1
        for k, p in self.params.items():
            dt = p.dtype.type
            b1, b2, eps, a_t = dt(np.float32(self.beta1)), dt(np.float32(self.beta2)), dt(np.float32(self.eps)), dt(self.alpha_t())
            if p.dtype == np.float32:
                a_t = np.float32(self.alpha_t())
            g, m, v = self.grads[k], self.m[k], self.v[k]
            m[...] = b1 * m + (dt(1) - b1) * g
            v[...] = b2 * v + (dt(1) - b2) * g * g
            p[...] = p - a_t * m / (np.sqrt(v) + eps)


def compute_learning_rate(conf_train, i, lr):
    """python/solver.py:82-98: linear warm-up, then a cosine that ends at lr_end_ratio * lr."""
    epoch = conf_train["epoch"]
    warmup_term = int(epoch * conf_train["warmup_term_ratio"])
    warmup_term = 0 if warmup_term < 1 else warmup_term
    r = conf_train["learning_rate_end_ratio"]
    if i < warmup_term:
        return lr * i / warmup_term
    x = np.pi * (i - warmup_term) / (epoch - warmup_term)
    a = (1 - r) * lr / (1 + np.cos(np.pi * warmup_term / epoch))
    b = a + r * lr
    return np.cos(x) * a + b


def cos_anneal_ratio(conf_train, i):
    """python/solver.py:100-108."""
    x = i / (conf_train["epoch"] * conf_train["cos_anneal_term_ratio"])
    return 0.5 * np.cos(np.pi * x) + 0.5 if x < 1.0 else 1.0


def light_visibility_gain(conf_train, i):
    """python/solver.py:110-119."""
    M = conf_train["sigmoid_gain_lv_end"]
    b = (M + 1) * 0.5
    a = 1 - b
    return a * np.cos(np.pi * i / conf_train["epoch"]) + b


class Solvers:
    """python/solver.py:20-80 on the numpy `Adam` above."""

    def __init__(self, conf_train, **adam):
        B, R = conf_train["batch_size"], conf_train["n_rays"]
        self.learning_rate_weight = conf_train["base_learning_rate_weight"] * (B * R) / (1 * 512)
        self.learning_rate_feat = conf_train["base_learning_rate_feat"] * (B * R) / (1 * 512)
        self.solver_weight = Adam(0, **adam)
        self.solver_feat = Adam(0, **adam)
        self.conf = conf_train

    def set_parameters(self, params):
        self.solver_weight.set_parameters({k: p for k, p in params.items() if not k.endswith("feature/F")})
        self.solver_feat.set_parameters({k: p for k, p in params.items() if k.endswith("feature/F")})

    def weight_decay(self):
        self.solver_weight.weight_decay(self.conf["weight_decay"])
        self.solver_feat.weight_decay(self.conf["weight_decay"])

    def clip_grad_by_norm(self):
        if self.conf["clip_grad_norm"] <= 0:
            return
        self.solver_weight.clip_grad_by_norm(self.conf["clip_grad_norm"])
        self.solver_feat.clip_grad_by_norm(self.conf["clip_grad_norm"])

    def update(self):
        self.solver_weight.update()
        self.solver_feat.update()

    def zero_grad(self):
        self.solver_weight.zero_grad()
        self.solver_feat.zero_grad()

    def check_inf_or_nan_grad(self):
        return self.solver_weight.check_inf_or_nan_grad() and self.solver_feat.check_inf_or_nan_grad()

    def update_learning_rate(self, i):
        self.solver_weight.set_learning_rate(compute_learning_rate(self.conf, i, self.learning_rate_weight))
        self.solver_feat.set_learning_rate(compute_learning_rate(self.conf, i, self.learning_rate_feat))

    def step(self, loss_grads):
        """python/train.py:136-148 after the forward pass: `loss_grads` {name: dL/dw} is what backward
        accumulates on top of the (clipped) decay term.  Returns False when the guard skipped the update."""
        self.zero_grad()
        self.weight_decay()
        self.clip_grad_by_norm()
        for s in (self.solver_weight, self.solver_feat):
            for k in s.params:
                if loss_grads.get(k) is not None:
                    s.grads[k] += loss_grads[k]
        if self.check_inf_or_nan_grad():
            return False
        self.update()
        return True
