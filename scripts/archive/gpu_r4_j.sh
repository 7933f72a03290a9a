#!/bin/bash
# Round 4: same-box A/B of the one-block final sum (batched 16-byte loads against the simple loop)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4j}
mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline") or {}; print(round(d["ms_per_step"],4), "kernel_ms", round(d.get("kernel_ms",0),4), "frac", round(r.get("frac",0),4))
except Exception as e:
    print("ERR", e)
PY
}
for rep in 1 2 3; do
for v in new simple; do
  so=""; [ $v != new ] && so="$GRAFT_REPO_ROOT/nova_amd/libnova_mi355x_$v.so"
  for wl in mle_eval:20 mle_eval:24 sumcheck3:24 sumcheck3:20 quad_prod:24; do
    name=${wl%%:*}; lg=${wl##*:}
    NMX_SO=$so timeout 300 python bench.py --workload $name --log2n $lg --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${v}_${name}_$lg.json" 2> "$OUT/fv.err"; echo -n "$v $name 2^$lg: "; show "$OUT/${v}_${name}_$lg.json"
  done
done
done
echo "== done"
