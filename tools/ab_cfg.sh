#!/bin/bash
# same-box A/B of a bench configuration under environment settings: bash tools/ab_cfg.sh <config> "A=1" "-" ...
CFG=$1; shift
for rep in 1 2; do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
    r=$(env $envs python bench.py --config $CFG --no-cpu-baseline --no-extra-legs --train-steps 0 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']), d.get('execution','')[:20])")
    echo "[$CFG $cfg] $r"
  done
done
