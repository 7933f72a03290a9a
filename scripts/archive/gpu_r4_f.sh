#!/bin/bash
# Round 4: NMX_ASYNC field calls in the prove_step replay (A/B against synchronous calls), the async chain test
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4f}
mkdir -p "$OUT"
echo "== pytest async + horner"; timeout 900 python -m pytest tests/test_gpu_fieldvec.py -q --maxfail=8 -k "async" > "$OUT/pytest.txt" 2>&1; tail -4 "$OUT/pytest.txt"
for rep in 1 2; do
for mode in "" "--separate-field-ops" "--sync-field-ops"; do
  for it in 65536 1024; do
  timeout 300 python bench.py --workload prove_step_replay --iters $it --steps 10 --warmup 3 $mode > "$OUT/ps_${it}_${mode:-async}.json" 2> "$OUT/ps.err"
  python - "$OUT/ps_${it}_${mode:-async}.json" "$it ${mode:-async}" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"],4), d["cpu_baseline"]["gpu_matches_cpu"], d["breakdown_ms"])
PY
  done
done
done
echo "== done"
