#!/bin/bash
# round 5, call l: nmx_commit_begin / finish parity + the prove_step replay serial vs overlapped through the ticket API
mkdir -p gpurun_out/r5l
timeout 900 python -m pytest tests/test_gpu_commit_overlap.py tests/test_cpp_mirror.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r5l/pytest_commit_overlap.txt
for ov in 0 1 2 0 1 2; do
  timeout 300 python bench.py --workload prove_step_replay --steps 30 --warmup 5 --overlap-commits $ov 2>/dev/null | tail -1 > gpurun_out/r5l/ps_ov${ov}.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r5l/ps_ov${ov}.json"))
print("overlap ${ov}: %.4f ms  matches %s" % (d["value"], d.get("cpu_baseline", {}).get("gpu_matches_cpu")))
PY
done 2>&1 | tee gpurun_out/r5l/overlap.txt
