#!/bin/bash
# round 6, lease b: config 5 as one chained replay (+ trait-only form), sizes 14 and 20
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6b
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_large.py -x -q -m gpu -k "compressed or trait_only" 2>&1 | tail -15 | tee "$OUT/pytest.txt"
for l in 14 20; do
  timeout 900 python bench.py --workload compressed_snark_replay --log2n $l --steps 5 --warmup 2 > "$OUT/csnark_$l.json" 2> "$OUT/csnark_$l.err"
  tail -3 "$OUT/csnark_$l.err"
  python - "$OUT/csnark_$l.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"][:60], "->", round(d["value"], 3), "ms; cpu", round(d["cpu_baseline"]["value"], 1), "ms; matches", d["cpu_baseline"]["gpu_matches_cpu"])
print("  groups", d["groups_ms"])
print("  trait_only", {k: v for k, v in d["trait_only"].items() if k != "what"})
print("  breakdown", d["breakdown_ms"])
PY
done | tee "$OUT/summary.txt"
timeout 600 python bench.py --workload prove_step_replay --steps 5 --warmup 2 > "$OUT/prove_step.json" 2> "$OUT/prove_step.err"; tail -c 1500 "$OUT/prove_step.json"
