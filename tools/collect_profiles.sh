#!/bin/bash
# Collects the round's judged profiles on the GPU box into gpurun_out/profiles_new/ (copy into profiles/ afterwards).
# 1) rocprofv3 kernel stats of the default bench command in eager mode (+3 training iterations so that the solver
#    kernels appear), 2) FETCH_SIZE, WRITE_SIZE and the MFMA-busy / wave-cycle counters in separate --pmc passes
#    (no trace domains beside --pmc), 3) the default bench line, 4) per-net kernel durations of both split engines.
export TMPDIR=/tmp
O=gpurun_out/profiles_new; rm -rf $O; mkdir -p $O
# per-step kernel table of the fwd+bwd step: two eager runs of 4 and 12 timed steps, differenced (one-time launches cancel)
for S in 4 12; do
  rocprofv3 --kernel-trace --stats -d $O/ks -o run --output-format csv -- python3 bench.py --exec eager --steps $S --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/bench_under_rocprof.json 2> $O/ks_err.txt
  f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/bench_kernel_stats_s$S.csv
  rm -rf $O/ks
done
python tools/diff_summary.py $O/bench_kernel_stats_s4.csv $O/bench_kernel_stats_s12.csv 4 12 70 > $O/bench_kernel_summary.txt
# ... and of a run that also holds 3 training iterations (solver kernels; totals, not per step)
rocprofv3 --kernel-trace --stats -d $O/ks -o run --output-format csv -- python3 bench.py --exec eager --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 3 > /dev/null 2>> $O/ks_err.txt
f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py "$f" 1 25 > $O/train_kernel_totals.txt
rm -rf $O/ks
mv $O/bench_kernel_stats_s12.csv $O/bench_kernel_stats.csv; rm -f $O/bench_kernel_stats_s4.csv
PCMD="python3 bench.py --exec eager --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 2"
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
  python tools/pmc_table.py "$f" "k_wgrad_group" >> $O/pmc_mfma_bench.txt
  rm -rf $O/pmc_m
done
# 5) the grid-feature micro-benchmark (the reference authors' shape: 2^19 points) with FETCH_SIZE / WRITE_SIZE per launch
{
  echo "# python tools/grid_bench.py 10   (1x MI355X; the reference authors' micro-benchmark shape, scripts/bench_voxel_hash.py / bench_lanczos_voxel.py: P = 2^19 points)"
  python tools/grid_bench.py 10 2> $O/grid_bench_err.txt
  echo
  echo "# HBM traffic of the same launches: rocprofv3 --pmc FETCH_SIZE -- python3 tools/grid_bench.py 2 ; rocprofv3 --pmc WRITE_SIZE -- python3 tools/grid_bench.py 2 (separate passes)"
  echo "# mean per launch over the uniform and the ray-coherent point sets, KB (FETCH_SIZE uncorrected: these are 16-byte gathers, not wide streaming reads)"
  echo "# kernel template arguments: <topology 0 voxel / 1 triplane / 2 triline / 3 hash, interpolation 0 linear / 2 lanczos, ...>"
} > $O/grid_bench.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_g -o run --output-format csv -- python3 tools/grid_bench.py 2 > /dev/null 2> $O/pmc_grid_err.txt
  f=$(find $O/pmc_g -name "*counter_collection.csv" | head -1)
  echo "## $c" >> $O/grid_bench.txt
  python tools/pmc_table.py "$f" ndjir >> $O/grid_bench.txt
  rm -rf $O/pmc_g
done
python bench.py > $O/bench_default.json 2> $O/bench_default_err.txt
bash tools/chain_shapes.sh f16x3 > $O/chain_shapes.txt 2>&1
tail -c 600 $O/bench_default.json
