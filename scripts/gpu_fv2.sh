#!/bin/bash
# roofline of the remaining field-vector kernels + small-MSM stage profile + full GPU test suite
set -u
OUT=gpurun_out/fv2; mkdir -p $OUT
for w in quad_prod lincomb8 horner mle_eval spmv sumcheck3; do
  for l in 20 22; do
    echo "== $w 2^$l"; timeout 200 python bench.py --workload $w --log2n $l --steps 10 --warmup 2 > $OUT/bench_${w}_$l.json 2>$OUT/err_${w}_$l.txt; python -c "
import json; d=json.load(open('$OUT/bench_${w}_$l.json')); print(round(d['kernel_ms'],4),'ms kernel', round(d['roofline']['achieved'],1),'GB/s frac',round(d['roofline']['frac'],3), 'wall ms', round(d['ms_per_step'],3), d.get('cpu_baseline',{}).get('gpu_matches_cpu'), d.get('cpu_baseline',{}).get('value'))" || tail -3 $OUT/err_${w}_$l.txt
  done
done
for l in 10 12 13 14 16; do
  echo "== msm 2^$l"; timeout 100 python bench.py --log2n $l --steps 30 --warmup 5 --no-cpu-baseline > $OUT/msm_$l.json 2>/dev/null; python scripts/show.py $OUT/msm_$l.json
done
echo "== full gpu suite"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
