#!/bin/bash
# Round 4: prefix tables on wide-table keys: tests, and the HyperKZG replay at 2^22 with / without them
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4n}
mkdir -p "$OUT"
echo "== pytest"; timeout 1200 python -m pytest tests/test_gpu_large.py tests/test_gpu_multidev.py tests/test_gpu_batch_fused.py -q -x --maxfail=3 > "$OUT/pytest.txt" 2>&1; grep -E "passed|failed|Error|^E " "$OUT/pytest.txt" | tail -8
for rep in 1 2; do
for pt in 1 0; do
  NMX_TUNE_PREFIX_TABLES=$pt timeout 900 python bench.py --workload hyperkzg_replay --log2n 22 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefix_tables', $pt, 'hkzg 2^22', round(d['value'],3), d['cpu_baseline']['gpu_matches_cpu'])"
done
done
echo "== done"
