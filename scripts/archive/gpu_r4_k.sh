#!/bin/bash
# Round 4: fuzz on the final build (plain-key partition, digit walk, batches), then the evidence refresh of the default line
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4k}
mkdir -p "$OUT"
echo "== fuzz (3 seeds x 300 iterations)"
for seed in 11 12 13; do timeout 600 python scripts/gpu_fuzz.py 300 $seed 2>&1 | tail -2; done
echo "== pytest: fieldvec + parity + multidev"; timeout 900 python -m pytest tests/test_gpu_fieldvec.py tests/test_gpu_parity.py tests/test_gpu_multidev.py tests/test_gpu_large.py -q --maxfail=5 2>&1 | grep -E "passed|failed" | tail -2
echo "== default line"; timeout 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default_line.json" 2> "$OUT/bench_default.err"; echo "rc=$?"
python - "$OUT/bench_default_line.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], "stages", d["stages_ms"]); print("trait_form", d["trait_form"]["ms"], "prove", d["prove_step_replay_ms"]["ms"], "hkzg", d["hyperkzg_replay_ms"]["ms"])
print({k:(v["frac"], v["kernel_ms"]) for k,v in d["fieldvec"].items() if isinstance(v,dict)})
print(d["fieldvec"]["_min_frac"], d["cpu_baseline"]["gpu_matches_cpu"], d["prove_step_replay_ms"]["gpu_matches_cpu"], d["hyperkzg_replay_ms"]["gpu_matches_cpu"])
PY
for lg in 20 16; do timeout 300 python bench.py --workload mle_eval --log2n $lg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mle_eval', $lg, round(d['kernel_ms'],4), round(d['roofline']['frac'],4))"; done
echo "== done"
