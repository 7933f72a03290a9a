#!/bin/bash
# round 6: the chained replay with the secondary's inner-product argument in it
set -u
OUT=gpurun_out/r6ipa
mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_large.py -x -q -k "compressed_snark_replay" 2>&1 | tail -8 | tee "$OUT/pytest_chain.txt"
timeout 900 python bench.py --workload compressed_snark_replay --log2n 20 --steps 5 --warmup 2 > "$OUT/csnark_20.json" 2> "$OUT/csnark_20.err"
tail -2 "$OUT/csnark_20.err" | grep -v amdgpu.ids
python - "$OUT/csnark_20.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms", d["value"], "cpu", d["cpu_baseline"]["value"], "match", d["cpu_baseline"]["gpu_matches_cpu"], "groups", d["groups_ms"])
print({k: v for k, v in d["breakdown_ms"].items() if k.startswith("S.ee")})
c = d.get("cpp_driver", {})
print("cpp", {k: c.get(k) for k in ("ms", "median_ms", "gpu_matches_cpu", "failed", "groups_ms", "error")})
PY
