#!/bin/bash
export TMPDIR=/tmp
for m in "$@"; do
  O=gpurun_out/cs_$m; rm -rf $O; mkdir -p $O
  export NDJIR_MLP_MATH=$m
  rocprofv3 --kernel-trace -d $O/kt -o run --output-format csv -- python3 tools/chain_shapes.py > $O/out.txt 2> $O/err.txt
  f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
  echo "== $m"; python tools/chain_shapes_report.py "$f" | tee $O/report.txt
  rm -rf $O/kt
done
