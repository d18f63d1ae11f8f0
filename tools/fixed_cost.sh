#!/bin/bash
# Per-kernel-symbol fit t(R) = a + b*R of the fwd+bwd step:
#   bash tools/fixed_cost.sh <tag> [configs...]      -> gpurun_out/fc_<tag>/{<config>_R<R>.csv, <config>_R<R>.json, fit.txt}
export TMPDIR=/tmp
TAG=$1; shift
CONFIGS=${@:-default no_voxel}
O=gpurun_out/fc_$TAG; rm -rf $O; mkdir -p $O
SA=4; SB=12; WARM=1     # two eager runs per point: per-step kernel time = (run B - run A) / (SB - SA), free of one-time launches
for c in $CONFIGS; do
  for R in 256 512 1024 2048 4096; do
    # whole-step time from graph replays (what the bench line reports)
    python3 bench.py --config $c --rays $R --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/${c}_R$R.json 2> $O/${c}_R$R.err
    # per-kernel durations from an eager run under the kernel trace
    for S in $SA $SB; do
      rocprofv3 --kernel-trace --stats -d $O/ks -o run --output-format csv -- python3 bench.py --config $c --rays $R --exec eager --steps $S --warmup $WARM --no-cpu-baseline --no-extra-legs --train-steps 0 > /dev/null 2>> $O/${c}_R$R.err
      f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
      cp "$f" $O/${c}_R${R}_s$S.csv
      rm -rf $O/ks
    done
  done
done
python3 tools/fixed_cost.py $O $SA $SB $CONFIGS > $O/fit.txt
cat $O/fit.txt
