cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pipeline_variants.py tests/test_gpu_large.py tests/test_gpu_batch_fused.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for rep in 1 2; do for h in 0 1; do
export NMX_TUNE_FINAL_TREE=$h
for lg in 18 19 20 21; do
  timeout 600 python bench.py --log2n $lg --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tree=$h 2^$lg', round(d['ms_per_step'],3), 'ms', d['stages_ms'])"
done; done; done
