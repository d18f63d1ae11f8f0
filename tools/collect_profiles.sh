#!/bin/bash
# Collects the round's judged profiles on the GPU box into gpurun_out/profiles_new/ (copy into profiles/ afterwards).
# 1) rocprofv3 kernel stats of the default bench command in eager mode (+3 training iterations so that the solver
#    kernels appear), 2) FETCH_SIZE and WRITE_SIZE in separate --pmc passes, 3) the default bench line.
export TMPDIR=/tmp
O=gpurun_out/profiles_new; rm -rf $O; mkdir -p $O
CMD="python3 bench.py --exec eager --steps 10 --warmup 3 --no-cpu-baseline --train-steps 3"
rocprofv3 --kernel-trace --stats -d $O/ks -o run --output-format csv -- $CMD > $O/bench_under_rocprof.json 2> $O/ks_err.txt
f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
cp "$f" $O/bench_kernel_stats.csv
python tools/prof_summary.py $O/bench_kernel_stats.csv 1 60 > $O/bench_kernel_summary.txt
rm -rf $O/ks
PCMD="python3 bench.py --exec eager --steps 3 --warmup 1 --no-cpu-baseline --train-steps 2"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_$c -o run --output-format csv -- $PCMD > /dev/null 2> $O/pmc_${c}_err.txt
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1)
  echo "## $c" >> $O/pmc_hbm_bench.txt
  python tools/pmc_table.py "$f" ndjir >> $O/pmc_hbm_bench.txt
  rm -rf $O/pmc_$c
done
python bench.py > $O/bench_default.json 2> $O/bench_default_err.txt
tail -c 600 $O/bench_default.json
