cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_batch_fused.py tests/test_gpu_large.py -x -q 2>&1 | tail -2
for q in 0 1 0 1; do
export NMX_TUNE_REDUCE_QUAD_X2=$q
for lg in 20; do
  timeout 600 python bench.py --log2n $lg --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('quad_x2=$q 2^$lg', round(d['ms_per_step'],3), 'ms', d['stages_ms'])"
done; done
unset NMX_TUNE_REDUCE_QUAD_X2
echo "== parts (fused, c=17 key, 16 per run)"
timeout 300 python scripts/tmp/hkzg_parts.py 2>&1 | grep -v amdgpu.ids | grep -E "batch_commit"
timeout 300 python bench.py --workload hyperkzg_replay --log2n 20 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hkzg 2^20', round(d['value'],3))"
