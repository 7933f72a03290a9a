#!/bin/bash
# round 5, call k: prove_step replay with commit(W) beside cross term + commit(T) (two host threads)
mkdir -p gpurun_out/r5k
for ov in 0 1 2 0 1 2; do
  timeout 300 python bench.py --workload prove_step_replay --steps 30 --warmup 5 --overlap-commits $ov 2>/dev/null | tail -1 > gpurun_out/r5k/ps_ov${ov}.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r5k/ps_ov${ov}.json"))
print("overlap ${ov}: %.4f ms  matches %s" % (d["value"], d.get("cpu_baseline", {}).get("gpu_matches_cpu")), d["breakdown_ms"])
PY
done 2>&1 | tee gpurun_out/r5k/overlap.txt
