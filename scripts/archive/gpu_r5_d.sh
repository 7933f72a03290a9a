#!/bin/bash
# Round 5, call D: small-MSM block path with one partial per quad
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5d
mkdir -p "$OUT"
echo "== small msm variants"; timeout 1500 python -m pytest tests/test_gpu_pipeline_variants.py -q --maxfail=6 -k small_msm > "$OUT/pytest_small.txt" 2>&1; tail -4 "$OUT/pytest_small.txt"
for q in 8 4 16 2; do echo "-- small_blocks $q"; NMX_TUNE_SMALL_BLOCKS=$q timeout 600 python scripts/gpu_small_msm_stages.py 30 2>&1 | grep -v amdgpu.ids | head -5 | tee -a "$OUT/small_msm_stages_q$q.txt"; done
echo "== prove_step replay"
for sb in 8 0 8; do
  NMX_TUNE_SMALL_BLOCKS=$sb timeout 600 python bench.py --workload prove_step_replay --iters 65536 --steps 10 --warmup 3 > "$OUT/prove_step_sb${sb}.json" 2> "$OUT/prove_step_sb${sb}.err"
  python - "$OUT/prove_step_sb${sb}.json" $sb <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("small_blocks", sys.argv[2], "prove_step ms", round(d["value"],4), d["cpu_baseline"]["gpu_matches_cpu"], d["breakdown_ms"])
PY
done
echo "== done"
