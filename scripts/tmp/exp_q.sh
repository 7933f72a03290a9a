cd $GRAFT_REPO_ROOT
for q in 0 1 0 1; do
export NMX_TUNE_QUAD_INCLUSIVE=$q
for lg in 20 21; do
  timeout 600 python bench.py --log2n $lg --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('quad_incl=$q 2^$lg', round(d['ms_per_step'],3), 'ms', d['stages_ms'], d['cpu_baseline'] if 'cpu_baseline' in d else '')"
done; done
