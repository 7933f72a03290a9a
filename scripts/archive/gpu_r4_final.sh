#!/bin/bash
# Round 4: validation of the final build -- the whole GPU suite, smoke, the default line
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4final}
mkdir -p "$OUT"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -q -m gpu --maxfail=8 > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -3
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== default line"; timeout 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default_line.json" 2> "$OUT/bench_default.err"; echo "rc=$?"
python - "$OUT/bench_default_line.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], "stages", d["stages_ms"]); print("trait_form", d["trait_form"]["ms"], "prove", d["prove_step_replay_ms"]["ms"], d["prove_step_replay_ms"]["breakdown_ms"], "hkzg", d["hyperkzg_replay_ms"]["ms"])
print({k:(v["frac"], v["kernel_ms"], v.get("valu_floor_ms")) for k,v in d["fieldvec"].items() if isinstance(v,dict)})
print(d["fieldvec"]["_min_frac"], d["cpu_baseline"]["gpu_matches_cpu"], d["prove_step_replay_ms"]["gpu_matches_cpu"], d["hyperkzg_replay_ms"]["gpu_matches_cpu"])
PY
echo "== done"
