cd $GRAFT_REPO_ROOT
for q in 0 1; do
export NMX_TUNE_REDUCE_QUAD_X2=$q
for lg in 20 21; do
  timeout 600 python bench.py --log2n $lg --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('quad_x2=$q 2^$lg', round(d['ms_per_step'],3), 'ms', d['stages_ms'])"
done; done
unset NMX_TUNE_REDUCE_QUAD_X2
echo "== parts (fused, c=17 key)"
timeout 300 python scripts/tmp/hkzg_parts.py 2>&1 | grep -v amdgpu.ids | grep -E "batch_commit|3 opens|commit n-1"
bash scripts/gpu_r2_check.sh c17
