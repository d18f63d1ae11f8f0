export TMPDIR=/tmp
C=$1
O=gpurun_out/aggtrace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/ks -o run --output-format csv -- python3 bench.py --config $C --exec eager --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/bench.json 2> $O/err.txt
f=$(find $O/ks -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections, os
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if any(x in n for x in os.environ.get('TRACE_KERNELS', 'k_scatter,k_tv').split(',')):
        d[n.split('(')[0].replace('void ndjir::','')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in d.items():
    v = v[len(v)//5:]
    print("  %-34s n=%d  sum/step %.1f us   launches: %s" % (n, len(v), sum(v) / 4, ' '.join('%.0f' % x for x in v[-8:])))
PY
rm -rf $O/ks
