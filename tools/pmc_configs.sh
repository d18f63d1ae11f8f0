#!/bin/bash
# HBM traffic per kernel symbol of the grid kernels in the `custom` (Lanczos voxel) and `triplaneline` steps: separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE; no trace domains beside --pmc) of a short eager bench run -> gpurun_out/profiles_new/pmc_<config>.txt
export TMPDIR=/tmp
O=gpurun_out/profiles_new; mkdir -p $O
for c in custom triplaneline; do
  : > $O/pmc_$c.txt
  echo "# rocprofv3 --pmc <counter> -- python3 bench.py --config $c --exec eager --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 ; mean per launch, KB (tools/pmc_table.py); grid kernels only" >> $O/pmc_$c.txt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr -d $O/pmc_c -o run --output-format csv -- python3 bench.py --config $c --exec eager --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 > /dev/null 2> $O/pmc_${c}_err.txt
    f=$(find $O/pmc_c -name "*counter_collection.csv" | head -1)
    echo "## $ctr" >> $O/pmc_$c.txt
    python tools/pmc_table.py "$f" "ndjir::k_" | grep -A1 -E "scatter|query|tv|zero_touched|pack_rows" | grep -v "^--" >> $O/pmc_$c.txt
    rm -rf $O/pmc_c
  done
done
