#!/bin/bash
# Round 5, call C: one-launch sum-check rounds (k_sc_pass) A/B; small-MSM block path: tests, stages, prove_step replay A/B
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5c
mkdir -p "$OUT"
echo "== spartan tests"; timeout 1500 python -m pytest tests/test_gpu_spartan.py -q --maxfail=6 > "$OUT/pytest_spartan.txt" 2>&1; tail -8 "$OUT/pytest_spartan.txt"
echo "== small msm variants"; timeout 1500 python -m pytest tests/test_gpu_pipeline_variants.py -q --maxfail=6 -k small_msm > "$OUT/pytest_small.txt" 2>&1; tail -8 "$OUT/pytest_small.txt"
echo "== parity + batch + slice cache (default path changed for small keys)"; timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch_fused.py tests/test_gpu_slice_cache.py tests/test_gpu_public_kats.py -q --maxfail=6 > "$OUT/pytest_parity.txt" 2>&1; tail -4 "$OUT/pytest_parity.txt"
echo "== spartan replay fused 1 / 0"
for f in 1 0 1 0; do
  for l in 14 20; do
    NMX_SC_FUSED_SUM=$f timeout 900 python bench.py --workload spartan_replay --log2n $l --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/spartan_${l}_fused${f}.json" 2> "$OUT/spartan_${l}_fused${f}.err"
    python - "$OUT/spartan_${l}_fused${f}.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split("/")[-1], round(d["value"],3), {k:(v["ms"],v["wait_ms"],v["host_algebra_ms"],v["launches"]) for k,v in d["provers"].items()}, all(d["proof_verifies"].values()))
PY
  done
done
echo "== small MSM stages: block path (default) / task path"
timeout 600 python scripts/gpu_small_msm_stages.py 30 > "$OUT/small_msm_stages_blocks.txt" 2>&1; cat "$OUT/small_msm_stages_blocks.txt"
NMX_TUNE_SMALL_BLOCKS=0 timeout 600 python scripts/gpu_small_msm_stages.py 30 > "$OUT/small_msm_stages_tasks.txt" 2>&1; cat "$OUT/small_msm_stages_tasks.txt"
for q in 4 16; do NMX_TUNE_SMALL_BLOCKS=$q timeout 600 python scripts/gpu_small_msm_stages.py 30 2>&1 | head -3; done
echo "== prove_step replay A/B"
for sb in 8 0 8 0; do
  NMX_TUNE_SMALL_BLOCKS=$sb timeout 600 python bench.py --workload prove_step_replay --iters 65536 --steps 10 --warmup 3 > "$OUT/prove_step_sb${sb}.json" 2> "$OUT/prove_step_sb${sb}.err"
  python - "$OUT/prove_step_sb${sb}.json" $sb <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("small_blocks", sys.argv[2], "prove_step ms", round(d["value"],4), d["cpu_baseline"]["gpu_matches_cpu"], d["breakdown_ms"])
PY
done
echo "== hyperkzg replay 2^14 / 2^20 (batch commits of short vectors take the small path)"
for l in 14 20; do timeout 600 python bench.py --workload hyperkzg_replay --log2n $l --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hkzg', $l, round(d['value'],3), d['cpu_baseline']['gpu_matches_cpu'])"; done
echo "== done"
