#!/bin/bash
# A/B runs of the round-2 pipeline stages + kernel trace.  Outputs under gpurun_out/.
#   A  NMX_TUNE_NO_PARTITION=1                   round-1 pipeline: DigitsFn + rocPRIM + BoundsFn, task accumulate
#   B  NMX_TUNE_SEG_MIN_TOTAL=0xffffffff         hand-written partition, task accumulate
#   C  (default)                                 hand-written partition + segment-balanced accumulate
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-ab}
mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(round(d["ms_per_step"],4), d["stages_ms"])
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-400:])
PY
}
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest gpu"
timeout 900 python -m pytest tests -q -m gpu --maxfail=5 -x ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.txt" 2>&1; tail -8 "$OUT/pytest_gpu.txt"
fi
for lg in 20 21 18 16 13; do
  for mode in A B C; do
    case $mode in A) envs="NMX_TUNE_NO_PARTITION=1";; B) envs="NMX_TUNE_SEG_MIN_TOTAL=0xffffffff";; C) envs="NMX_X=0";; esac
    echo "== $mode log2n=$lg"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/bench_${mode}_$lg.json" 2> "$OUT/bench_${mode}_$lg.err"
    show "$OUT/bench_${mode}_$lg.json"
  done
done
for dist in u1 u16 u64 equal zero_rm1; do
  for mode in A C; do
    case $mode in A) envs="NMX_TUNE_NO_PARTITION=1";; C) envs="NMX_X=0";; esac
    echo "== $mode dist=$dist"
    env $envs timeout 300 python bench.py --steps 10 --warmup 3 --dist $dist --no-extras --no-cpu-baseline > "$OUT/bench_${mode}_$dist.json" 2> "$OUT/bench_${mode}_$dist.err"
    show "$OUT/bench_${mode}_$dist.json"
  done
done
echo "== seg tuning at 2^20"
for v in "NMX_TUNE_SEG_MIN_LEN=4" "NMX_TUNE_NO_QUAD_FINAL=1" "NMX_TUNE_SEG_LANES=131072" "NMX_TUNE_SEG_LANES=393216"; do
  echo "-- $v"
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$OUT/bench_tune.json" 2> "$OUT/bench_tune.err"; show "$OUT/bench_tune.json"
done
echo "== rocprof kernel trace (C)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o msm -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4 "$f" | sed 's/void nmx:://; s/(nmx::[^"]*"/"/; s/(nmx::.*)//' | cut -c1-120 | head -26
echo "== done"
