#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   -> bench line (rays/s, ms/step, wgrad avg us) per value
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})" "$var=$v"
done
