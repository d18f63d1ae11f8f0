#!/bin/bash
# rocprofv3 kernel stats of a short eager bench run under the given math mode: bash tools/quick_prof.sh <tag> [env ...]
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/qp_$TAG; rm -rf $O; mkdir -p $O
env "$@" true
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats -d $O/ks -o run --output-format csv -- python3 bench.py --exec eager --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/bench.json 2> $O/err.txt
f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py "$f" 7 30 > $O/summary.txt
rm -rf $O/ks
cat $O/summary.txt
