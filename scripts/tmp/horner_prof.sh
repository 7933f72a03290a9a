export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/hprof" -o h -- python "$R/bench.py" --workload horner --log2n 24 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> "$R/gpurun_out/hprof.err" )
f=$(find $R/gpurun_out/hprof -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 "$f" | sed 's/void nmx:://' | cut -c1-120 | head -14
rm -rf $R/gpurun_out/hprof
