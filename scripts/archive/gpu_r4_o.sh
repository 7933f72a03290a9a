#!/bin/bash
# Round 4: the prefix chain for batches (prefix_tables = 2) against the wide-key-only form (1): tests under 2, HyperKZG replays A/B
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4o}
mkdir -p "$OUT"
echo "== pytest under prefix_tables = 2"; NMX_TUNE_PREFIX_TABLES=2 timeout 1200 python -m pytest tests/test_gpu_batch_fused.py tests/test_gpu_large.py tests/test_gpu_multidev.py tests/test_gpu_parity.py tests/test_gpu_slice_cache.py -q --maxfail=3 > "$OUT/pytest.txt" 2>&1; grep -E "passed|failed|Error|^E " "$OUT/pytest.txt" | tail -6
echo "== fuzz under 2"; NMX_TUNE_PREFIX_TABLES=2 timeout 600 python scripts/gpu_fuzz.py 300 21 2>&1 | tail -1
for rep in 1 2 3; do
for pt in 1 2; do
  for lg in 14 16 18 20; do
  NMX_TUNE_PREFIX_TABLES=$pt timeout 900 python bench.py --workload hyperkzg_replay --log2n $lg --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefix_tables', $pt, 'hkzg 2^$lg', round(d['value'],3))"
  done
done
done
echo "== done"
