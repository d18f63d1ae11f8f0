#!/bin/bash
# rocprofv3 kernel stats of the training iteration (fwd+bwd + guarded Adam): bash tools/train_prof.sh <tag>
export TMPDIR=/tmp
O=gpurun_out/tp_$1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/ks -o run --output-format csv -- python3 bench.py --exec eager --steps 2 --warmup 1 --no-cpu-baseline --train-steps 8 > $O/bench.json 2> $O/err.txt
f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py "$f" 1 400 > $O/summary.txt
rm -rf $O/ks
grep -i "adam\|pack\|nonfinite\|decay\|zero\|mark\|veto\|check\|sum_sq" $O/summary.txt
