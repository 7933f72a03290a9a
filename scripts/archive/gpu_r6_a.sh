#!/bin/bash
# round 6, lease a: the safety fixes (challenge-line checksum, batch fall-back beside pre-launched passes, begun commitment over a
# sharded key) and the new replay shapes (Grumpkin-field Spartan, Pasta prove_step), then the Spartan replay with the checksum in.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6a
mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_spartan.py tests/test_gpu_commit_overlap.py -x -q -m gpu 2>&1 | tail -15 | tee "$OUT/pytest_spartan.txt"
timeout 900 python -m pytest tests/test_gpu_multidev.py tests/test_gpu_large.py -x -q -m gpu -k "begun or pasta or prove_step" 2>&1 | tail -15 | tee "$OUT/pytest_new.txt"
for l in 14 20; do
  for pre in 1 0 1 0; do
    NMX_SC_PRELAUNCH=$pre timeout 300 python bench.py --workload spartan_replay --log2n $l --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/spartan_${l}_pre${pre}.json" 2>> "$OUT/err.txt"
    python - "$OUT/spartan_${l}_pre${pre}.json" $l $pre <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"2^{sys.argv[2]} prelaunch {sys.argv[3]}: {d['value']:.3f} ms", {k: v for k, v in d['breakdown_ms'].items() if k.startswith('sumcheck')})
PY
  done
done | tee "$OUT/spartan_prelaunch_ab.txt"
