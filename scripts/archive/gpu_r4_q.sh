#!/bin/bash
# Round 4: single-bin partitions (c = 8 keys) without the counting pass: parity + small-MSM timing + prove_step replay
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4q}
mkdir -p "$OUT"
echo "== pytest"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline_variants.py tests/test_gpu_batch_fused.py tests/test_gpu_fuzz.py tests/test_gpu_large.py -q --maxfail=3 2>&1 | grep -E "passed|failed|^E " | tail -4
echo "== fuzz"; timeout 600 python scripts/gpu_fuzz.py 400 31 2>&1 | tail -1
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d.get("stages_ms"))
except Exception as e:
    print("ERR", e)
PY
}
for rep in 1 2; do
for lg in 10 12 13; do
  timeout 300 python bench.py --steps 30 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/t_$lg.json" 2> "$OUT/t.err"; echo -n "tables 2^$lg: "; show "$OUT/t_$lg.json"
done
timeout 300 python bench.py --workload prove_step_replay --iters 65536 --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prove_step 65536', round(d['value'],4), d['cpu_baseline']['gpu_matches_cpu'], d['breakdown_ms'])"
done
echo "== done"
