#!/bin/bash
# round 6, lease f: the chained replay with the overlapped random-instance commitments, the C++ driver beside it
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6f
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_large.py tests/test_cpp_mirror.py -x -q -m gpu -k "compressed or trait_only or cpp" 2>&1 | tail -15 | tee "$OUT/pytest.txt"
for l in 14 17 20; do
  timeout 900 python bench.py --workload compressed_snark_replay --log2n $l --steps 5 --warmup 2 > "$OUT/csnark_$l.json" 2> "$OUT/csnark_$l.err"
  tail -3 "$OUT/csnark_$l.err" | grep -v amdgpu.ids
  python - "$OUT/csnark_$l.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"][:64], "->", round(d["value"], 3), "ms; cpu", round(d["cpu_baseline"]["value"], 1), "ms; matches", d["cpu_baseline"]["gpu_matches_cpu"])
print("  groups", d["groups_ms"])
print("  cpp_driver", {k: v for k, v in d["cpp_driver"].items() if k != "what"})
print("  trait_only", {k: v for k, v in d["trait_only"].items() if k in ("ms", "calls", "gpu_matches_cpu")})
PY
done | tee "$OUT/summary.txt"
