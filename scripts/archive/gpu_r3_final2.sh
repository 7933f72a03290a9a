#!/bin/bash
# Round 3 final evidence (second pass: fused tree placement fix, item-list big-bucket pass, worker pool):
# GPU suite, default line, kernel trace of the headline command, scalar distributions.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3final2}
mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout 1500 python -m pytest tests -q -m gpu --maxfail=5 -x > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -2
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== default line"; timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"; cut -c1-200 "$OUT/bench_default.json"
echo "== kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -o msm -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/trace.err" )
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && cut -d, -f1-4 "$f" | sed 's/void nmx:://; s/(nmx::[^"]*"/"/; s/(nmx::.*)//' | cut -c1-110 | head -20
echo "== distributions"
for dist in random u64 u32 u16 u10 u1 equal zero_rm1; do
  timeout 300 python bench.py --log2n 20 --dist $dist --steps 20 --warmup 5 --no-extras > "$OUT/dist_$dist.json" 2> "$OUT/dist_$dist.err"
  python - "$OUT/dist_$dist.json" $dist <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:9s} {d['ms_per_step']:.4f} ms  {d['stages_ms']}  matches_cpu={d.get('cpu_baseline',{}).get('gpu_matches_cpu')}")
PY
done
echo "== done"
