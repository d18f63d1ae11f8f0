#!/bin/bash
# The round's secondary measurements -> gpurun_out/profiles_new/: other configurations, strong-scaling mode on one GPU, the N > 1 code
# path against RCCL with a 1-rank group, the exchange kernels at the 8-rank size, the forward-only render bench, the stock launch list.
O=gpurun_out/profiles_new; mkdir -p $O
{
  echo "# bench.py --config <c> --no-cpu-baseline --train-steps 0 --steps 20 --warmup 3 (512 rays, one box): ms/step, rays/s"
  for c in default triplaneline custom no_voxel ste; do bash tools/ab_cfg.sh $c - 2>/dev/null | tail -1; done
  echo "# bench.py --scaling strong --total-rays 4096 --config no_voxel (N = 1 point of BASELINE config 4's curve)"
  python bench.py --scaling strong --total-rays 4096 --config no_voxel --no-cpu-baseline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms', round(d['value']), 'rays/s'); print('projected 8-GPU strong scaling:', d.get('projected_strong_scaling_8'))"
} > $O/configs.txt 2>&1
{
  echo "# NDJIR_BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0   (1-rank RCCL group: the N > 1 code path on one GPU)"
  NDJIR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','execution','exchange','loss')}); print(d.get('eager'))"
  echo "# per-step split (NDJIR_BENCH_TRACE=1): pre-exchange (mask all-reduce) / graph replay / exchange (bucket all-reduce + sparse grid exchange), host-synchronised around each part"
  NDJIR_BENCH_TRACE=1 NDJIR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29572 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --train-steps 0 2>&1 | grep "^\[rank" | tail -6
  echo "# python tools/exchange_bench.py   (GPU time of one rank's share of an 8-rank exchange at the bench size)"
  python tools/exchange_bench.py 2>/dev/null | tail -3
} > $O/force_dist_1rank_rccl.txt 2>&1
# per-kernel tables of BASELINE config 3 (triplaneline) and of `custom` (Lanczos voxel): differenced 4- and 12-step eager runs, as for the default step
export TMPDIR=/tmp
for c in triplaneline custom; do
  for S in 4 12; do
    rocprofv3 --kernel-trace --stats -d $O/ks -o run --output-format csv -- python3 bench.py --config $c --exec eager --steps $S --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 > /dev/null 2> $O/ks_${c}_err.txt
    f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
    cp "$f" $O/ks_${c}_s$S.csv
    rm -rf $O/ks
  done
  python tools/diff_summary.py $O/ks_${c}_s4.csv $O/ks_${c}_s12.csv 4 12 45 > $O/${c}_kernel_summary.txt
  rm -f $O/ks_${c}_s4.csv $O/ks_${c}_s12.csv
done
python tools/render_bench.py > $O/render_bench.txt 2>/dev/null
python tools/launch_list.py 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $O/stock_launches.txt
tail -3 $O/configs.txt
