#!/bin/bash
# round 5, call y: pre-launched passes, third form (challenge line in uncached device memory, written through the BAR): parity + spartan replay on / off
mkdir -p gpurun_out/r5za
timeout 1500 python -m pytest tests/test_gpu_spartan.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r5za/pytest_spartan.txt
for q in 1 0 1 0; do
  for l in 20 14; do
  NMX_SC_PRELAUNCH=$q timeout 600 python bench.py --workload spartan_replay --log2n $l --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('prelaunch $q 2^$l: %.3f ms' % d['value'], {k: v for k, v in d['breakdown_ms'].items() if k.startswith('sumcheck')}); print('    ', {k: (v['wait_ms'], v['host_algebra_ms'], v['launches']) for k, v in d['provers'].items()})"
  done
done 2>&1 | tee gpurun_out/r5za/prelaunch.txt
