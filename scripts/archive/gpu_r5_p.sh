#!/bin/bash
# round 5, call p: the round's host inversion prepared under the device pass: parity + spartan replay
mkdir -p gpurun_out/r5t
timeout 1200 python -m pytest tests/test_gpu_spartan.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r5t/pytest_spartan.txt
for i in 1 2 3; do
  timeout 600 python bench.py --workload spartan_replay --log2n 20 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5t/sp_$i.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r5t/sp_$i.json"))
print("run $i: %.3f ms" % d["value"], {k: v for k, v in d["breakdown_ms"].items() if k.startswith("sumcheck")})
for k, v in d["provers"].items(): print("   ", k, v)
PY
done 2>&1 | tee gpurun_out/r5t/prepare_inv.txt
for l in 14 17; do
timeout 600 python bench.py --workload spartan_replay --log2n $l --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2^$l: %.3f ms' % d['value'])"
done 2>&1 | tee -a gpurun_out/r5t/prepare_inv.txt
