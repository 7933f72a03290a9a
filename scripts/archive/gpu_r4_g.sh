#!/bin/bash
# Round 4: HyperKZG replay: the three openings as one batch_commit / three threads / serial; stream-ordered folds
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4g}
mkdir -p "$OUT"
for rep in 1 2; do
for lg in 20 16 14; do
for mode in "--opens batch" "--opens threads" "--opens serial" "--opens threads --sync-field-ops"; do
  tag=$(echo "$mode" | tr -d ' -')
  timeout 600 python bench.py --workload hyperkzg_replay --log2n $lg --steps 5 --warmup 2 $mode > "$OUT/hk_${lg}_$tag.json" 2> "$OUT/hk.err"
  python - "$OUT/hk_${lg}_$tag.json" "2^$lg $mode" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"],4), d["cpu_baseline"]["gpu_matches_cpu"])
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done
done
done
tail -3 "$OUT/hk.err"
echo "== done"
