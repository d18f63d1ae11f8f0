#!/bin/bash
# same-box A/B of the default bench under different environment settings: bash tools/ab.sh "A=1" "B=2 C=3" ...  ("-" = none)
for rep in 1 2; do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
    r=$(env $envs python bench.py --no-cpu-baseline --train-steps 0 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']))")
    echo "[$cfg] $r"
  done
done
