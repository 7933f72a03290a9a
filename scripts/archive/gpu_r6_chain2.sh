#!/bin/bash
set -u
OUT=gpurun_out/r6ipa2
mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_large.py -x -q -k "compressed_snark_replay" 2>&1 | tail -4 | tee "$OUT/pytest_chain.txt"
for lg in 20 14; do
timeout 900 python bench.py --workload compressed_snark_replay --log2n $lg --steps 7 --warmup 3 > "$OUT/csnark_$lg.json" 2> "$OUT/csnark_$lg.err"
python - "$OUT/csnark_$lg.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms", round(d["value"], 3), "cpu", round(d["cpu_baseline"]["value"], 1), "match", d["cpu_baseline"]["gpu_matches_cpu"], "groups", d["groups_ms"])
print({k: v for k, v in d["breakdown_ms"].items() if k.startswith("S.ee")})
c = d.get("cpp_driver", {})
print("cpp", {k: c.get(k) for k in ("ms", "median_ms", "gpu_matches_cpu", "failed", "groups_ms", "error")})
PY
done
timeout 900 python bench.py --workload compressed_snark_replay --log2n 20 --steps 7 --warmup 3 --serial-snarks > "$OUT/csnark_20_serial.json" 2> "$OUT/csnark_20_serial.err"
python - "$OUT/csnark_20_serial.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("serial: ms", round(d["value"], 3), "groups", d["groups_ms"])
c = d.get("cpp_driver", {})
print("serial cpp", {k: c.get(k) for k in ("ms", "median_ms", "gpu_matches_cpu", "failed", "groups_ms", "error")})
PY
