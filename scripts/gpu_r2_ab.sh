#!/bin/bash
# A/B of the hand-written partition against the rocPRIM path + kernel trace.  Outputs under gpurun_out/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-ab}
mkdir -p "$OUT"
echo "== pytest gpu (quick subset first)"
timeout 900 python -m pytest tests -q -m gpu --maxfail=5 -x ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.txt" 2>&1; tail -15 "$OUT/pytest_gpu.txt"
for np in 1 0; do
  for lg in 20 16 13; do
    echo "== bench no_partition=$np log2n=$lg"
    NMX_TUNE_NO_PARTITION=$np timeout 300 python bench.py --steps 20 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/bench_np${np}_$lg.json" 2> "$OUT/bench_np${np}_$lg.err"
    python - "$OUT/bench_np${np}_$lg.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(round(d["ms_per_step"],4), d["stages_ms"])
PY
  done
done
for dist in u1 u16 u64; do
  echo "== bench dist=$dist"
  timeout 300 python bench.py --steps 10 --warmup 3 --dist $dist --no-extras --no-cpu-baseline > "$OUT/bench_$dist.json" 2>/dev/null
  python - "$OUT/bench_$dist.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(round(d["ms_per_step"],4), d["stages_ms"])
PY
done
echo "== rocprof kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o msm -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4 "$f" | sed 's/void nmx:://; s/(nmx::.*,/,/' | cut -c1-150 | head -24
echo "== done"
