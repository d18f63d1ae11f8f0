export TMPDIR=/tmp
O=gpurun_out/gp; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/ks -o run --output-format csv -- python3 tools/grid_bench.py 3 > $O/out.txt 2> $O/err.txt
f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py "$f" 1 60 | grep -E "plane|hash|scatter|k_zero|kernel " 
