#!/bin/bash
# Round 5, call G: the whole GPU suite, smoke, evidence runs (rocprof summaries of the Spartan replay and of a small MSM), default line
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5g
mkdir -p "$OUT"
echo "== pytest gpu"; timeout 2400 python -m pytest tests -q -m gpu --maxfail=8 > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -3
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== spartan replay"; for l in 14 17 20; do timeout 900 python bench.py --workload spartan_replay --log2n $l --steps 5 --warmup 2 > "$OUT/spartan_$l.json" 2>/dev/null; python -c "import json,sys; d=json.loads(open('$OUT/spartan_$l.json').read().strip().splitlines()[-1]); print('spartan', $l, round(d['value'],3), 'cpu', round(d['cpu_baseline']['value'],1), d['cpu_baseline']['gpu_matches_cpu'], d['breakdown_ms'])"; done
echo "== rocprof spartan 2^20"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_spartan20" -o sp20 -- python "$GRAFT_REPO_ROOT/bench.py" --workload spartan_replay --log2n 20 --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$OUT/prof_spartan20.log" 2>&1 ); echo "rc=$?"
echo "== rocprof small msm"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_smallmsm" -o sm -- python "$GRAFT_REPO_ROOT/scripts/gpu_small_msm_stages.py" 10 > "$GRAFT_REPO_ROOT/$OUT/prof_smallmsm.log" 2>&1 ); echo "rc=$?"
echo "== rocprof headline"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_msm" -o msm -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof_msm.err" ); echo "rc=$?"
echo "== default line"; timeout 1500 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default_line.json" 2> "$OUT/bench_default.err"; echo "rc=$?"
python - "$OUT/bench_default_line.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], "cpu", d["cpu_baseline"])
print("trait_form", d["trait_form"]["ms"], "incl_h2d", d["incl_h2d"]["ms"], "prove", d["prove_step_replay_ms"]["ms"], "hkzg", d["hyperkzg_replay_ms"]["ms"], "spartan", d["spartan_replay_ms"]["ms"], d["spartan_replay_ms"]["gpu_matches_cpu"])
PY
echo "== done"
