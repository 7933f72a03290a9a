#!/bin/bash
# round 6: the whole GPU suite, smoke, and the default bench line (what the driver runs at round end)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6full
mkdir -p "$OUT"
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) 2>&1 | tee "$OUT/pytest_gpu.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.txt"
( time timeout 1200 python bench.py > "$OUT/bench_default_line.json" 2> "$OUT/bench.err" ) 2>&1 | tail -4 | tee "$OUT/bench_time.txt"
tail -3 "$OUT/bench.err" | grep -v amdgpu.ids
python - "$OUT/bench_default_line.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic")})
for k in ("prove_step_replay_ms", "hyperkzg_replay_ms", "spartan_replay_ms", "compressed_snark_replay_ms"):
    v = d.get(k, {})
    print(k, {q: v.get(q) for q in ("ms", "cpu_ms", "gpu_matches_cpu", "error")}, "trait_only", (v.get("trait_only") or {}).get("ms"))
print("keys", sorted(d))
PY
