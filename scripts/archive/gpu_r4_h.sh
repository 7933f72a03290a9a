#!/bin/bash
# Round 4: digit extraction without the array in memory (funnel shift / word walk), quad_pick without scratch: MSM sizes, plain keys
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4h}
mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d.get("stages_ms"))
except Exception as e:
    print("ERR", e)
PY
}
echo "== pytest parity + pipeline variants + large"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline_variants.py tests/test_gpu_large.py tests/test_gpu_batch_fused.py -q --maxfail=5 > "$OUT/pytest.txt" 2>&1; tail -3 "$OUT/pytest.txt"
for rep in 1 2; do
for lg in 20 13 16 18 22; do
  timeout 300 python bench.py --steps 20 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/t_$lg.json" 2> "$OUT/t.err"; echo -n "tables 2^$lg: "; show "$OUT/t_$lg.json"
done
for lg in 20 16; do
  timeout 300 python bench.py --steps 20 --warmup 5 --log2n $lg --no-tables --no-extras --no-cpu-baseline > "$OUT/p_$lg.json" 2> "$OUT/p.err"; echo -n "plain 2^$lg: "; show "$OUT/p_$lg.json"
done
done
echo "== done"
