#!/bin/bash
# rocprofv3 kernel stats of a short eager bench run of a named configuration: bash tools/quick_prof_cfg.sh <config>
export TMPDIR=/tmp
C=$1
O=gpurun_out/qp_$C; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/ks -o run --output-format csv -- python3 bench.py --config $C --exec eager --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/bench.json 2> $O/err.txt
f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py "$f" 7 40 > $O/summary.txt
rm -rf $O/ks
cat $O/summary.txt
