#!/bin/bash
# one bench run with per-kernel HIP-event detail; summary table -> gpurun_out/detail.txt
mkdir -p gpurun_out
NDJIR_BENCH_DETAIL=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/detail.json
python - <<'PY' > gpurun_out/detail.txt
import json
d=json.load(open("gpurun_out/detail.json"))
print(d["value"], d["ms_per_step"], d["execution"][:40], d.get("eager"))
print(json.dumps(d["roofline"]))
ks=d.get("kernels") or {}
rows=ks if isinstance(ks,list) else [dict(name=k,**v) if isinstance(v,dict) else dict(name=k,v=v) for k,v in ks.items()]
for r in rows: print(r)
kd=d.get("kernel_detail")
if kd:
    for r in (kd if isinstance(kd,list) else kd.items()): print(r)
PY
