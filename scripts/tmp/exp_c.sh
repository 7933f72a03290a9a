cd $GRAFT_REPO_ROOT
for lg in 20 21 22; do for c in 16 17 18 19 20; do
  timeout 600 python bench.py --log2n $lg --window-bits $c --steps 8 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^$lg c=$c', round(d['ms_per_step'],3), 'ms', d['stages_ms'])"
done; done
