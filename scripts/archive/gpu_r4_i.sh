#!/bin/bash
# Round 4: same-box A/B of the digit-extraction and quad_pick changes (variant libraries built with -DNMX_AB_OLD_*)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4i}
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
for v in new olddig oldpick; do
  so=""; [ $v != new ] && so="$GRAFT_REPO_ROOT/nova_amd/libnova_mi355x_$v.so"
  for lg in 20 13; do
    NMX_SO=$so timeout 300 python bench.py --steps 20 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/${v}_$lg.json" 2> "$OUT/t.err"; echo -n "$v tables 2^$lg: "; show "$OUT/${v}_$lg.json"
  done
  NMX_SO=$so timeout 300 python bench.py --steps 20 --warmup 5 --log2n 20 --no-tables --no-extras --no-cpu-baseline > "$OUT/${v}_p20.json" 2> "$OUT/p.err"; echo -n "$v plain 2^20: "; show "$OUT/${v}_p20.json"
done
done
echo "== done"
