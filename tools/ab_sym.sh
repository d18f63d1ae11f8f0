#!/bin/bash
# same-box A/B of the in-tree library against a variant, per kernel symbol: bash tools/ab_sym.sh <variant.so> <symbol substring> [reps]
V=$1; SYM=$2; REPS=${3:-3}
run() {
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --train-steps 0 2>/dev/null | SYM="$SYM" python -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels_by_symbol']
print('$1', 'ms/step %.3f' % d['ms_per_step'], ' | '.join('%s x%.0f %.1f us' % (n, v['launches_per_step'], v['avg_us']) for n, v in k.items() if os.environ['SYM'] in n))
"
}
for rep in $(seq $REPS); do
  unset NDJIR_HIP_LIB; run cur
  export NDJIR_HIP_LIB=$PWD/$V; run "$(basename $V)"
done
