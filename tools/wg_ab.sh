python -m pytest tests/test_gpu_mlp.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -3
for rep in 1 2; do
for v in cur split16; do
  if [ $v = cur ]; then unset NDJIR_HIP_LIB; else export NDJIR_HIP_LIB=$PWD/ndjir_amd/_lib/variants/wgp_$v.so; fi
  echo "== $v blocked"; WGT_ONE=1 WGT_BLOCKED=3 python tools/wgrad_group_time.py 8
  python bench.py --steps 20 --warmup 5 --no-extra-legs 2>/dev/null | tail -1 | cut -c1-200
done
done
