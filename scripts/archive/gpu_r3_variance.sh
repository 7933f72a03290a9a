#!/bin/bash
# One lease of the box-variance series: the headline command three times (+ optionally the full suite and the full default line).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-var}
mkdir -p "$OUT"
if [ "${WITH_TESTS:-0}" = "1" ]; then
  echo "== pytest gpu"; timeout 1500 python -m pytest tests -q -m gpu --maxfail=5 -x > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -2
fi
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$OUT/bench_$i.json" 2> "$OUT/bench_$i.err"
  python - "$OUT/bench_$i.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("VARIANCE", round(d["ms_per_step"],4), round(d["value"]/1e6,1), d["stages_ms"], d["per_call_ms"])
PY
done
if [ "${WITH_DEFAULT:-0}" = "1" ]; then
  echo "== default line"; timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"; cut -c1-300 "$OUT/bench_default.json"
  echo "== in-process --gpus 2 (fallback)"; timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 > "$OUT/bench_inproc.json" 2> "$OUT/bench_inproc.err"; echo "rc=$?"; tail -1 "$OUT/bench_inproc.err"; cut -c1-260 "$OUT/bench_inproc.json"
fi
echo "== done"
