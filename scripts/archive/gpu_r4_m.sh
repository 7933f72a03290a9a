#!/bin/bash
# Round 4: the final pass (FinalSegFn: one lane per bucket) against its four-lane form at 2^16 / 2^19 buckets (c = 17 / 20 tables)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4m}
mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d.get("stages_ms"))
except Exception as e:
    print("ERR", e)
PY
}
for rep in 1 2 3; do
for q in 65536 131072 1048576; do
  for lg in 20 21 22; do
    NMX_TUNE_QUAD_FINAL_BELOW=$q timeout 300 python bench.py --steps 20 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/q${q}_$lg.json" 2> "$OUT/t.err"; echo -n "quad_below=$q 2^$lg: "; show "$OUT/q${q}_$lg.json"
  done
done
done
echo "== done"
