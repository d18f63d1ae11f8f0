python -m pytest tests/test_gpu_mlp.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -3
for v in cur frag2; do
  if [ $v = cur ]; then unset NDJIR_HIP_LIB; else export NDJIR_HIP_LIB=$PWD/ndjir_amd/_lib/variants/wgp_$v.so; fi
  echo "== $v row-major"; WGT_ONE=1 python tools/wgrad_group_time.py 8
  echo "== $v blocked"; WGT_BLOCKED=3 python tools/wgrad_group_time.py 8
done
NDJIR_HIP_LIB=$PWD/ndjir_amd/_lib/variants/wg_tl.so python tools/wgrad_timeline.py
unset NDJIR_HIP_LIB
python bench.py --steps 20 --warmup 5 --no-extra-legs 2>/dev/null | tail -1 | cut -c1-300
