#!/bin/bash
# bench + rocprof kernel trace only.  usage: gpu_bench.sh <tag> [bench args]
set -u
export TMPDIR=/tmp
TAG=${1:-run}; shift
OUT=gpurun_out/$(date +%H%M%S)_$TAG
mkdir -p "$OUT"
echo "== bench $*"
timeout 900 python bench.py --steps 10 --warmup 3 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"; cat "$OUT/bench.json"; tail -5 "$OUT/bench.err"
echo "== rocprof kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o msm -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline "$@" > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4,8 "$f" | head -16
echo "== done"
