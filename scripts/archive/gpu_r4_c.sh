#!/bin/bash
# Round 4, third pass: last-block final sums (one launch per reduction pass), both eq tables in one launch, Horner ticket ids /
# canonical stores / overlap rejection, prove_step breakdown.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4c}
mkdir -p "$OUT"
echo "== pytest fieldvec + multidev + emul-free gpu subset"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 > "$OUT/pytest_fv.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_fv.txt" | tail -12
echo "== default line"; timeout 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"
python - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], "stages", d["stages_ms"]); print("trait_form", d.get("trait_form",{}).get("ms")); print("prove", d.get("prove_step_replay_ms")); print("hkzg", d.get("hyperkzg_replay_ms",{}).get("ms"))
print({k:((v.get("frac"), v.get("kernel_ms")) if isinstance(v,dict) else v) for k,v in d.get("fieldvec",{}).items() if not k.startswith("_")})
print(d.get("fieldvec",{}).get("_min_frac"), d.get("fieldvec",{}).get("error"))
PY
tail -3 "$OUT/bench_default.err"
echo "== done"
