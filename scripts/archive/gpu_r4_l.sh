#!/bin/bash
# Round 4: end-of-call wait: polling (NMX_SYNC_SPIN_US) against the blocking wait, on the round-4 replays (same box)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4l}
mkdir -p "$OUT"
for rep in 1 2 3; do
for spin in 0 3000; do
  for it in 65536 1024; do
    NMX_SYNC_SPIN_US=$spin timeout 300 python bench.py --workload prove_step_replay --iters $it --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spin', $spin, 'prove_step', $it, round(d['value'],4), d['breakdown_ms'])"
  done
  NMX_SYNC_SPIN_US=$spin timeout 300 python bench.py --workload hyperkzg_replay --log2n 16 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spin', $spin, 'hkzg 2^16', round(d['value'],4))"
  NMX_SYNC_SPIN_US=$spin timeout 300 python bench.py --steps 20 --warmup 5 --log2n 13 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spin', $spin, 'msm 2^13', round(d['ms_per_step'],4))"
done
done
echo "== done"
