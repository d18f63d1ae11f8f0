#!/bin/bash
# Same-box A/B of weight-gradient variants (the pool's boxes differ by +-3 %: only pairs inside one gpurun call compare):
#   bash tools/wg_ab.sh <variant.so | -> [ENV=VALUE ...]
# e.g.  tools/build_variant.sh wgp_f2off wgrad.hip -DWGP_FRAG2_OFF; gpurun -- 'bash tools/wg_ab.sh ndjir_amd/_lib/variants/wgp_f2off.so'
#       gpurun -- 'bash tools/wg_ab.sh - NDJIR_WGRAD_NO_WIDE=1'
# Prints three interleaved pairs of the default step (base, then variant) and the 8 x (256 x 256 x 65536) microbenchmark of both.
V=$1; shift
run() { if [ "$V" != "-" ]; then NDJIR_HIP_LIB=$PWD/$V "$@"; else env "${EXTRA[@]}" "$@"; fi; }
EXTRA=("$@"); [ ${#EXTRA[@]} -eq 0 ] && EXTRA=(NDJIR_AB_NOOP=1)
for rep in 1 2 3; do
  echo -n "base    "; python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 | cut -c60-95,155-190
  echo -n "variant "; run python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 | cut -c60-95,155-190
done
echo "== base, blocked";    WGT_ONE=1 WGT_BLOCKED=3 python tools/wgrad_group_time.py 8 2>/dev/null
echo "== variant, blocked"; WGT_ONE=1 WGT_BLOCKED=3 run python tools/wgrad_group_time.py 8 2>/dev/null
