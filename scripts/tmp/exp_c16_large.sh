set -u
cd $GRAFT_REPO_ROOT
for lg in 22 23 24; do for c in 16 0; do
  timeout 600 python bench.py --log2n $lg --window-bits $c --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^$lg c=$c', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1),'M/s', d['stages_ms'])"
done; done
