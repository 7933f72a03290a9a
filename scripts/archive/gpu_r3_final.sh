#!/bin/bash
# Round 3 final evidence: the GPU suite, the default bench line, the kernel trace of the headline command (final build).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3final}
mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout 1500 python -m pytest tests -q -m gpu --maxfail=5 -x > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -2
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== default line"; timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"; cut -c1-260 "$OUT/bench_default.json"
echo "== kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -o msm -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/trace.err" )
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4 "$f" | sed 's/void nmx:://; s/(nmx::[^"]*"/"/; s/(nmx::.*)//' | cut -c1-110 | head -20
echo "== done"
