#!/bin/bash
# Round 4: A/B of the Horner tile order and the last-block sums, fieldvec tests
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4d}
mkdir -p "$OUT"
echo "== pytest fieldvec"; timeout 900 python -m pytest tests/test_gpu_fieldvec.py tests/test_gpu_fieldvec_large.py -q --maxfail=8 > "$OUT/pytest_fv.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_fv.txt" | tail -5
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline") or {}; print(round(d["ms_per_step"],4), "kernel_ms", round(d.get("kernel_ms",0),4), "frac", round(r.get("frac",0),4))
except Exception as e:
    print("ERR", e)
PY
}
for rep in 1 2; do
for ord in 0 1; do
  for lg in 20 22 24; do
    NMX_TUNE_HORNER_ORDER=$ord timeout 300 python bench.py --workload horner --log2n $lg --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/horner_o${ord}_$lg.json" 2> "$OUT/horner.err"; echo -n "horner order=$ord 2^$lg: "; show "$OUT/horner_o${ord}_$lg.json"
  done
done
done
for wl in sumcheck3:24 mle_eval:24 mle_eval:20 quad_prod:24 round3:24; do
  name=${wl%%:*}; lg=${wl##*:}
  timeout 300 python bench.py --workload $name --log2n $lg --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${name}_$lg.json" 2> "$OUT/fv.err"; echo -n "$name 2^$lg: "; show "$OUT/${name}_$lg.json"
done
echo "== done"
