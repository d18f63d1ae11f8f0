#!/bin/bash
# Timeline (workgroup 0, geometric net forward, point-blocked side tensors) of the pipelined chain kernel and of its timing-only
# experiment builds (tools/build_variant.sh <name> mlp3p.hip -DNDJIR_CHAINP_X_...): bash tools/chainp_variants.sh [name ...]
for v in main "$@"; do
  if [ "$v" = main ]; then unset NDJIR_HIP_LIB; else export NDJIR_HIP_LIB=$PWD/ndjir_amd/_lib/variants/$v.so; fi
  echo "=== $v"
  TIMELINE_BLOCKED=1 TIMELINE_RAW=2 timeout 200 python tools/chain_timeline.py ${MODE:-fwd} 2>&1 | grep -E "launches|^  [0-9] |total|wave 0|wave 4"
done
