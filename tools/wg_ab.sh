python -m pytest tests/test_gpu_mlp.py tests/test_gpu_ops.py tests/test_gpu_step.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
  NDJIR_HIP_LIB=$PWD/ndjir_amd/_lib/variants/wgp_s16i.so python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
done
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_full.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_full.json').read()); print(d['value'], d['ms_per_step'], 'redraw', d['redraw'].get('loss'), d['redraw'].get('ms_per_step'), 'fp32', d['fp32_engine'].get('ms_per_step'), d['fp32_engine'].get('loss'), 'train', d['train_step']['ms_per_step'], 'b4', d['b4'].get('rays_per_s'))"
