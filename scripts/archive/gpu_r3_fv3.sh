#!/bin/bash
# Round 3: field-vector kernels after the dot-product / mul_add changes: parity tests, then the default driver line.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-fv3}
mkdir -p "$OUT"
echo "== pytest fieldvec"
timeout 900 python -m pytest tests/test_gpu_fieldvec.py tests/test_gpu_fieldvec_large.py tests/test_gpu_large.py -q -m gpu -x > "$OUT/pytest_fv.txt" 2>&1; tail -3 "$OUT/pytest_fv.txt"
grep -q "failed\|Error" "$OUT/pytest_fv.txt" && tail -40 "$OUT/pytest_fv.txt"
echo "== default line"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"
python - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], d["stages_ms"])
for k,v in d["fieldvec"].items():
    if isinstance(v, dict): print(f"  {k:14s} 2^{v.get('log2n')} {v.get('kernel_ms')} ms  frac {v.get('frac')}  ok={v.get('gpu_matches_cpu')}")
print("prove_step", d.get("prove_step_replay_ms",{}).get("ms"), "hkzg", (d.get("hyperkzg_replay_ms") or {}).get("ms"))
PY
echo "== done"
