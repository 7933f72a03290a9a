#!/bin/bash
# Round 4: same-box A/B: last-block final sums against the two-launch form; Horner block tickets against block ids
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4e}
mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline") or {}; print(round(d["ms_per_step"],4), "kernel_ms", round(d.get("kernel_ms",0),4), "frac", round(r.get("frac",0),4))
except Exception as e:
    print("ERR", e)
PY
}
for rep in 1 2; do
for ff in 0 1; do
for wl in sumcheck3:24 mle_eval:24 mle_eval:20 quad_prod:24 round3:24 sumcheck3:20; do
  name=${wl%%:*}; lg=${wl##*:}
  NMX_TUNE_FUSED_FINAL=$ff timeout 300 python bench.py --workload $name --log2n $lg --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${name}_${lg}_f$ff.json" 2> "$OUT/fv.err"; echo -n "fused_final=$ff $name 2^$lg: "; show "$OUT/${name}_${lg}_f$ff.json"
done
done
for ord in 0 1; do
  for lg in 20 22; do
    NMX_TUNE_HORNER_ORDER=$ord timeout 300 python bench.py --workload horner --log2n $lg --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/horner_o${ord}_$lg.json" 2> "$OUT/horner.err"; echo -n "horner order=$ord 2^$lg: "; show "$OUT/horner_o${ord}_$lg.json"
  done
done
done
echo "== done"
