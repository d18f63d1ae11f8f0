#!/bin/bash
# same-box A/B of the default bench under environment settings: bash tools/ab_env.sh reps "A=1" "B=2 C=3" ...   ("-" = none)
REPS=$1; shift
run() {
  env $2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --train-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('[$1]', 'ms/step %.3f' % d['ms_per_step'], ' '.join('%s %.1f' % (n, k[n]['avg_us']) for n in ('chain_fwd', 'chain_bwd', 'chain_tan', 'wgrad') if n in k))
"
}
for rep in $(seq $REPS); do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then run "-" ""; else run "$cfg" "$cfg"; fi
  done
done
