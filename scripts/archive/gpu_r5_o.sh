#!/bin/bash
# round 5, call o: batch prover claims on side streams: parity + spartan replay on / off
mkdir -p gpurun_out/r5o
timeout 1200 python -m pytest tests/test_gpu_spartan.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r5o/pytest_spartan.txt
for side in 1 0 1 0; do
  NMX_SC_SIDE_STREAMS=$side timeout 600 python bench.py --workload spartan_replay --log2n 20 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5o/sp_$side.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r5o/sp_$side.json"))
print("side_streams $side: %.3f ms" % d["value"], {k: v for k, v in d["breakdown_ms"].items() if k.startswith("sumcheck")}, d["provers"]["sumcheck_batch"])
PY
done 2>&1 | tee gpurun_out/r5o/side_streams.txt
