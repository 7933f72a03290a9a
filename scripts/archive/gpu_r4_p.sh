#!/bin/bash
# Round 4: block size / grid of the partition's counting pass (k_hist_hi), same box
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4p}
mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d.get("stages_ms"))
except Exception as e:
    print("ERR", e)
PY
}
for rep in 1 2; do
for cfg in "0:0" "256:0" "256:2048" "256:1024" "512:0" "128:0" "256:512"; do
  bs=${cfg%%:*}; gr=${cfg##*:}
  for lg in 20 22; do
    NMX_TUNE_HIST_BS=$bs NMX_TUNE_HIST_GRID=$gr timeout 300 python bench.py --steps 20 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/h_${bs}_${gr}_$lg.json" 2> "$OUT/t.err"; echo -n "hist_bs=$bs grid=$gr 2^$lg: "; show "$OUT/h_${bs}_${gr}_$lg.json"
  done
done
done
echo "== done"
