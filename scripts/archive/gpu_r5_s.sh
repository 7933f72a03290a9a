#!/bin/bash
# round 5, call s: host-tail threshold again, now that a host round's inversion costs 2-3 us
mkdir -p gpurun_out/r5s
for t in 6 7 8 6 7 8; do
  for l in 20 14; do
  NMX_SC_HOST_TAIL=$t timeout 600 python bench.py --workload spartan_replay --log2n $l --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tail $t 2^$l: %.3f ms' % d['value'], {k: v for k, v in d['breakdown_ms'].items() if k.startswith('sumcheck')})"
  done
done 2>&1 | tee gpurun_out/r5s/tail.txt
