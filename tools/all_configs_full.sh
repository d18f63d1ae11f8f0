for c in triplaneline custom ste no_voxel; do
  python bench.py --config $c --no-cpu-baseline > gpurun_out/full_$c.json 2> gpurun_out/full_$c.err; echo "$c rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/full_$c.json').read().strip().splitlines()[-1])
    print('  ', round(d['value']), d['ms_per_step'], 'redraw', d.get('redraw',{}).get('loss', d.get('redraw')), 'train', (d.get('train_step') or {}).get('ms_per_step'), 'b4', (d.get('b4') or {}).get('rays_per_s', d.get('b4')), 'fp32', (d.get('fp32_engine') or {}).get('ms_per_step', d.get('fp32_engine')))
except Exception as e:
    print('  parse failed', e); print(open('gpurun_out/full_$c.err').read()[-600:])
PY
done
