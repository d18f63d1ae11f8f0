for v in cur s16 s16i; do
  if [ $v = cur ]; then unset NDJIR_HIP_LIB; else export NDJIR_HIP_LIB=$PWD/ndjir_amd/_lib/variants/wgp_$v.so; fi
  for w in wide nowide; do
    if [ $w = wide ]; then unset NDJIR_WGRAD_NO_WIDE; else export NDJIR_WGRAD_NO_WIDE=1; fi
    echo "== $v $w blocked"; WGT_ONE=1 WGT_BLOCKED=3 python tools/wgrad_group_time.py 8
  done
done
unset NDJIR_WGRAD_NO_WIDE
NDJIR_HIP_LIB=$PWD/ndjir_amd/_lib/variants/wg_tl.so python tools/wgrad_timeline.py
