#!/bin/bash
# Round 4, second pass: the sharded-scalar fix (pieces cut by the REGISTERED key's layout), piece guard, big sharded sizes.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4b}
mkdir -p "$OUT"
echo "== pytest multidev"; timeout 900 python -m pytest tests/test_gpu_multidev.py -q -x > "$OUT/pytest_multidev.txt" 2>&1; tail -5 "$OUT/pytest_multidev.txt"
echo "== diag"; timeout 600 python scripts/gpu_r4_shard_diag.py 2>&1 | grep -v "amdgpu.ids\|^  \[" > "$OUT/shard_diag.txt"; cat "$OUT/shard_diag.txt"
echo "== in-process, 2 logical devices oversubscribed, RCCL required, 2^22 total"
NMX_BENCH_OVERSUB=1 NMX_BENCH_COMBINE=2 timeout 600 python bench.py --gpus 2 --total-log2n 22 --steps 5 --warmup 2 > "$OUT/inproc_oversub2.json" 2> "$OUT/inproc_oversub2.err"; echo "rc=$?"
python - "$OUT/inproc_oversub2.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["rccl_ranks"], d["combine_ms"], d.get("scalars_on_gpu0"), d["cpu_baseline"]["gpu_matches_cpu"])
PY
tail -3 "$OUT/inproc_oversub2.err"
echo "== done"
