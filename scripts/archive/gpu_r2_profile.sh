#!/bin/bash
# Round-2 measurement session: default bench line, prefetch experiment, kernel trace, PMC passes.  Outputs under gpurun_out/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-prof}
mkdir -p "$OUT"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest gpu"
timeout 1200 python -m pytest tests -q -m gpu --maxfail=5 > "$OUT/pytest_gpu.txt" 2>&1; tail -6 "$OUT/pytest_gpu.txt"
fi
echo "== default bench (with extras)"
t0=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "rc=$? wall=$(( $(date +%s) - t0 ))s"; cat "$OUT/bench.json"
for v in "NMX_TUNE_ACCUM_PF=2"; do
  echo "== $v"
  for lg in 20 21; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/b.json" 2> "$OUT/b.err"
  python - "$OUT/b.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(round(d["ms_per_step"],4), d["stages_ms"])
PY
  done
done
echo "== rocprof kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o msm -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && cut -d, -f1-4 "$f" | sed 's/void nmx:://; s/(nmx::[^"]*"/"/' | cut -c1-110 | head -22
rm -rf "$OUT/prof"
echo "== pmc passes"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/p$i" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> "$GRAFT_REPO_ROOT/$OUT/p$i.err" )
  find "$OUT/p$i" -name "*counter_collection.csv" | head -1
done
python scripts/pmc_summary.py "$OUT" "$OUT/pmc_traffic.json" | tee "$OUT/pmc_passes.txt"
for i in 1 2 3 4; do f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c "$f" > "$OUT/pmc_pass$i.csv.gz"; rm -rf "$OUT/p$i"; done
echo "== done"
