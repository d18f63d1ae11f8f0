"""Where the stock (non-ndjir) launches of one fwd+bwd step are issued: wraps the handful of torch functions they come from and
prints, per (op, shapes), the ndjir_amd source line of the caller.  usage: python tools/stock_sources.py [overrides]"""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ndjir_amd import config as cfg  # noqa: E402

dev = torch.device("cuda", 0)
conf = cfg.load("default", sys.argv[1:])
step = bench.Step(conf, 512, dev, 0, 1)
for _ in range(2):
    step.forward_backward()
torch.cuda.synchronize()
seen = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "ndjir_amd/" in fr.filename and "stock_sources" not in fr.filename:
            return f"{fr.filename.split('ndjir_amd/')[-1]}:{fr.lineno}"
    return "?"


def wrap(obj, name):
    orig = getattr(obj, name)

    def f(*a, **k):
        shapes = [tuple(t.shape) for t in a if torch.is_tensor(t)] + [tuple(t.shape) for x in a if isinstance(x, (list, tuple)) for t in x if torch.is_tensor(t)]
        if any(s for s in shapes):
            seen[(f"{getattr(obj, '__name__', 'Tensor')}.{name}", str(shapes)[:80], site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


for n in ("cat", "sigmoid", "neg", "sub", "add", "sum", "zeros", "ones", "full", "zeros_like", "empty_like", "where"):
    wrap(torch, n)
for n in ("copy_", "add_", "fill_", "zero_", "sum", "contiguous", "clone", "__sub__", "__add__", "__neg__", "__mul__", "mul"):
    wrap(torch.Tensor, n)
step.forward_backward()
torch.cuda.synchronize()
for (op, shapes, where), n in sorted(seen.items(), key=lambda kv: kv[0][2]):
    print(f"{n:3d}  {op:24s} {shapes:82s} {where}")
