#!/bin/bash
# round 6, lease d: the resident multi-round sum-check kernel -- parity, then A/B against a pass per round
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6d
mkdir -p "$OUT"
timeout 1200 python -m pytest tests/test_gpu_spartan.py -x -q -m gpu 2>&1 | tail -40 | tee "$OUT/pytest_spartan.txt"
for l in 14 20; do
  for res in 1 0 1 0; do
    NMX_SC_RESIDENT=$res timeout 300 python bench.py --workload spartan_replay --log2n $l --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/spartan_${l}_res${res}.json" 2>> "$OUT/err.txt"
    python - "$OUT/spartan_${l}_res${res}.json" $l $res <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"2^{sys.argv[2]} resident {sys.argv[3]}: {d['value']:.3f} ms", {k: v for k, v in d['breakdown_ms'].items() if k.startswith('sumcheck')},
      {k: (v['wait_ms'], v['host_algebra_ms'], v['launches']) for k, v in d['provers'].items()})
PY
  done
done | tee "$OUT/spartan_resident_ab.txt"
