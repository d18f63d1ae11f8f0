#!/bin/bash
# Collects the round's judged profiles on the GPU box into gpurun_out/profiles_new/ (copy into profiles/ afterwards).
# 1) rocprofv3 kernel stats of the default bench command in eager mode (+3 training iterations so that the solver
#    kernels appear), 2) FETCH_SIZE, WRITE_SIZE and the MFMA-busy / wave-cycle counters in separate --pmc passes
#    (no trace domains beside --pmc), 3) the default bench line, 4) per-net kernel durations of both split engines.
export TMPDIR=/tmp
O=gpurun_out/profiles_new; rm -rf $O; mkdir -p $O
CMD="python3 bench.py --exec eager --steps 10 --warmup 3 --no-cpu-baseline --train-steps 3"
rocprofv3 --kernel-trace --stats -d $O/ks -o run --output-format csv -- $CMD > $O/bench_under_rocprof.json 2> $O/ks_err.txt
f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
cp "$f" $O/bench_kernel_stats.csv
# steps executed under the profiler: the chain-forward launch count / 13 launches per step
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
# MFMA utilisation of the MLP kernels: SQ_VALU_MFMA_BUSY_CYCLES counts cycles (= 32 x MFMAs for 32x32x16 16-bit),
# SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE the kernel's cycles; SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY in quad-cycles
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '+')
  rocprofv3 --pmc $c -d $O/pmc_m -o run --output-format csv -- $PCMD > /dev/null 2> $O/pmc_mfma_err.txt
  f=$(find $O/pmc_m -name "*counter_collection.csv" | head -1)
  echo "## $tag" >> $O/pmc_mfma_bench.txt
  python tools/pmc_table.py "$f" "ndjir::x" >> $O/pmc_mfma_bench.txt
  python tools/pmc_table.py "$f" "k_wgrad" >> $O/pmc_mfma_bench.txt
  rm -rf $O/pmc_m
done
python bench.py > $O/bench_default.json 2> $O/bench_default_err.txt
bash tools/chain_shapes.sh f16x3 bf16x6 > $O/chain_shapes.txt 2>&1
tail -c 600 $O/bench_default.json
