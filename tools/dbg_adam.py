import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ndjir_amd import lib
from oracle import solver as OS
dev = torch.device("cuda:0")
for lr in [5e-4, 1e-3]:
    st = torch.zeros(4, device=dev); st[0] = float(np.float32(lr))
    o = OS.Adam(alpha=lr)
    for t in range(1, 8):
        lib.call("solver_adam_begin", st, 0.9, 0.999, None, None)
        o.t = t
        print(lr, t, float(st[2]).hex(), float(np.float32(o.alpha_t())).hex(), o.alpha_t().hex())
n = 64
rng = np.random.RandomState(0)
w0 = (rng.randn(n) * 1e-3).astype(np.float32); g0 = rng.randn(n).astype(np.float32)
o = OS.Adam(alpha=5e-4); wo = w0.copy(); o.set_parameters({"w": wo}); o.grads["w"] += g0; o.update()
w, g, m, v = [torch.from_numpy(a.copy()).to(dev) for a in (w0, g0, np.zeros(n, np.float32), np.zeros(n, np.float32))]
lib.call("solver_adam", n, w, g, m, v, float(np.float32(o.alpha_t())), 0.9, 0.999, 1e-8, 0.0, 0, None)
print("w mism", int((w.cpu().numpy() != wo).sum()), "m", int((m.cpu().numpy() != o.m["w"]).sum()), "v", int((v.cpu().numpy() != o.v["w"]).sum()))
# which op: recompute pieces with torch on the GPU
a_t = np.float32(o.alpha_t())
num = (a_t * o.m["w"]); den = np.sqrt(o.v["w"]) + np.float32(1e-8)
tn = torch.from_numpy(num).to(dev) / torch.from_numpy(den).to(dev)
print("torch div vs numpy", int((tn.cpu().numpy() != num / den).sum()))
print("sqrt", int((torch.sqrt(v).cpu().numpy() != np.sqrt(o.v["w"])).sum()))
