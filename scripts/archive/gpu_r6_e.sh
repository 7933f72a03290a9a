#!/bin/bash
# round 6, lease e: evaluations through the mailbox, the three (transposed) products in one call -- parity, then the Spartan replay A/B
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6e
mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_spartan.py tests/test_gpu_fieldvec.py -x -q -m gpu 2>&1 | tail -12 | tee "$OUT/pytest.txt"
for l in 14 20; do
  for sep in 0 1 0 1; do
    extra=""; [ $sep = 1 ] && extra="--separate-spmv"
    timeout 300 python bench.py --workload spartan_replay --log2n $l --steps 10 --warmup 3 --no-cpu-baseline $extra > "$OUT/spartan_${l}_sep${sep}.json" 2>> "$OUT/err.txt"
    python - "$OUT/spartan_${l}_sep${sep}.json" $l $sep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"2^{sys.argv[2]} separate_spmv {sys.argv[3]}: {d['value']:.3f} ms", d['breakdown_ms'])
PY
  done
done | tee "$OUT/spartan_ab.txt"
