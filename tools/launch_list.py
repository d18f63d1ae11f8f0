"""Every stock (non-ndjir) device launch of one fwd+bwd step with the aten op, its input shapes and the ndjir_amd source
line that issued it (forward) or the autograd node that ran it (backward).  usage: python tools/launch_list.py [overrides]"""
import collections, os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ndjir_amd import config as cfg  # noqa: E402

dev = torch.device("cuda", 0)
conf = cfg.load(os.environ.get("LAUNCH_LIST_CONFIG", "default"), sys.argv[1:])
step = bench.Step(conf, 512, dev, 0, 1)
for _ in range(2):
    step.forward_backward()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step.forward_backward()
    torch.cuda.synchronize()

evs = [e for e in prof.events()]
rows = collections.defaultdict(lambda: [0, 0.0])
own = [0, 0.0]
for e in evs:
    if str(e.device_type).endswith("CUDA"):
        continue
    kern = [k for k in e.kernels] if hasattr(e, "kernels") else []
    if not kern:
        continue
    names = [k.name for k in kern]
    if e.cpu_children and any(c.kernels for c in e.cpu_children if hasattr(c, "kernels")):
        # count kernels at the innermost op only
        inner = set()
        for c in e.cpu_children:
            for k in getattr(c, "kernels", []):
                inner.add(id(k))
        kern = [k for k in kern if id(k) not in inner]
        if not kern:
            continue
    t = sum(k.duration for k in kern)
    if all("ndjir" in k.name for k in kern):
        own[0] += len(kern); own[1] += t
        continue
    frame = ""
    for s in (e.stack or []):
        if "ndjir_amd/" in s or "bench.py" in s:
            frame = s.split("ndjir_amd/")[-1] if "ndjir_amd/" in s else s.split("/")[-1]
            break
    if not frame:
        p = e.cpu_parent
        while p is not None:
            if "Backward" in p.name or "evaluate_function" in p.name:
                frame = "bwd " + p.name.replace("autograd::engine::evaluate_function: ", "")
                break
            p = p.cpu_parent
    shapes = str([s for s in (e.input_shapes or []) if s])[:70]
    key = (e.name[:40], shapes, frame[:90])
    rows[key][0] += len(kern); rows[key][1] += t
tot = sum(v[0] for v in rows.values()); tt = sum(v[1] for v in rows.values())
print(f"stock launches {tot}  device time {tt / 1e3:.3f} ms | ndjir launches {own[0]} {own[1] / 1e3:.3f} ms")
for (name, shapes, frame), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d} {t:8.1f} us  {name:40s} {shapes:70s} {frame}")
