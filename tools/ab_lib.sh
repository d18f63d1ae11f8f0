#!/bin/bash
# same-box A/B of the in-tree library against variants: bash tools/ab_lib.sh <variant.so> [reps]   (bench ms/step + chain kernel averages)
V=$1; REPS=${2:-3}
run() {
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --train-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('$1', 'ms/step %.3f' % d['ms_per_step'], ' '.join('%s %.1f' % (n, k[n]['avg_us']) for n in ('chain_fwd', 'chain_bwd', 'chain_tan', 'wgrad') if n in k))
"
}
for rep in $(seq $REPS); do
  unset NDJIR_HIP_LIB; run cur
  export NDJIR_HIP_LIB=$PWD/$V; run "$(basename $V)"
done
