#!/bin/bash
set -u
OUT=gpurun_out/$(date +%H%M%S)_fv; mkdir -p $OUT
echo "== pytest fieldvec"; timeout 900 python -m pytest tests/test_gpu_fieldvec.py -x -q -m gpu 2>&1 | tail -8
for w in axpy cross_term bind; do
  for l in 20 24; do
    echo "== $w 2^$l"; timeout 300 python bench.py --workload $w --log2n $l --steps 20 --warmup 3 > $OUT/bench_${w}_$l.json 2>$OUT/err_${w}_$l.txt; python -c "
import json; d=json.load(open('$OUT/bench_${w}_$l.json')); print(round(d['kernel_ms'],4),'ms kernel', round(d['roofline']['achieved'],1),'GB/s frac',round(d['roofline']['frac'],3), 'wall ms', round(d['ms_per_step'],3), d.get('cpu_baseline',{}).get('gpu_matches_cpu'), d.get('cpu_baseline',{}).get('value'))" || tail -3 $OUT/err_${w}_$l.txt
  done
done
