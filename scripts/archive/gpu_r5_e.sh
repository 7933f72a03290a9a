#!/bin/bash
# Round 5, call E: host_split (trait form), DPF ubench, tests of the round's new paths, the default bench line
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5e
mkdir -p "$OUT"
echo "== tests"; timeout 1800 python -m pytest tests/test_gpu_pipeline_variants.py tests/test_gpu_spartan.py -q --maxfail=6 > "$OUT/pytest_a.txt" 2>&1; tail -5 "$OUT/pytest_a.txt"
echo "== dpf ubench"; timeout 300 bench/dpf_ubench | tee "$OUT/dpf_ubench.jsonl"
echo "== trait form / host_split"; timeout 900 python scripts/gpu_trait_form.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/host_split.txt"
echo "== default line"; timeout 1500 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default_line.json" 2> "$OUT/bench_default.err"; echo "rc=$?"; tail -3 "$OUT/bench_default.err"
python - "$OUT/bench_default_line.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], "stages", d["stages_ms"]); print("trait_form", d["trait_form"]["ms"], "incl_h2d", d["incl_h2d"]["ms"], "prove", d["prove_step_replay_ms"]["ms"], d["prove_step_replay_ms"]["breakdown_ms"], "hkzg", d["hyperkzg_replay_ms"]["ms"])
print("spartan", d["spartan_replay_ms"])
print({k:(v["frac"], v["kernel_ms"]) for k,v in d["fieldvec"].items() if isinstance(v,dict) and "frac" in v})
print("at 2^20", {k:(v["frac"], v["kernel_ms"], v["gpu_matches_cpu"]) for k,v in d["fieldvec"]["at_2p20"].items() if isinstance(v,dict)})
print(d["fieldvec"]["_min_frac"], d["cpu_baseline"]["gpu_matches_cpu"], d["prove_step_replay_ms"]["gpu_matches_cpu"], d["hyperkzg_replay_ms"]["gpu_matches_cpu"])
PY
echo "== done"
