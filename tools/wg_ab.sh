python -m pytest tests/test_gpu_mlp.py -x -q -m gpu -k "wgrad" 2>&1 | tail -2
for rep in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 | cut -c60-200
  NDJIR_HIP_LIB=$PWD/ndjir_amd/_lib/variants/wgp_f2off.so python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 | cut -c60-200
done
echo "== f2 wide blocked"; WGT_ONE=1 WGT_BLOCKED=3 python tools/wgrad_group_time.py 8
echo "== f2off wide blocked"; NDJIR_HIP_LIB=$PWD/ndjir_amd/_lib/variants/wgp_f2off.so WGT_ONE=1 WGT_BLOCKED=3 python tools/wgrad_group_time.py 8
