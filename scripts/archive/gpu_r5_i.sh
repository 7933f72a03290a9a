#!/bin/bash
# Round 5, call I: why is the reduce stage 0.31 ms at 2^21 and 0.15 ms at 2^20 with the same 2^16 buckets?  kernel traces of both
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5i
mkdir -p "$OUT"
for lg in 20 21; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/p$lg" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --log2n $lg --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/$OUT/bench_$lg.json" 2>/dev/null )
  echo "== 2^$lg"; python -c "import json; d=json.loads(open('$OUT/bench_$lg.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'])"
  f=$(find "$OUT/p$lg" -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
PY
done
echo "== done"
