#!/bin/bash
# Round 3 closing run: launch-chain (stream vs hipGraph) and point-addition latency micro-benchmarks, the GPU suite and the
# default bench line on the final build (after the big-bucket grid fix).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3closing}
mkdir -p "$OUT"
echo "== graph gap"; for n in 8 24; do timeout 120 bench/graph_gap $n 200; done | tee "$OUT/graph_gap.jsonl"
echo "== add latency"; timeout 300 bench/lat_test | tee "$OUT/lat_test.json"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -q -m gpu --maxfail=5 -x > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -2
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== default line"; timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"; cut -c1-260 "$OUT/bench_default.json"
echo "== done"
