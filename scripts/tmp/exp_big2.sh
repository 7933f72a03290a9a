set -u
cd $GRAFT_REPO_ROOT
for lg in 22 24; do for v in "NMX_X=1" "NMX_TUNE_ACCUM_PF=2" "NMX_TUNE_SEG_LANES=196608" "NMX_TUNE_SEG_LANES=393216"; do
  env $v timeout 600 python bench.py --log2n $lg --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^$lg $v', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1),'M/s', d['stages_ms'])"
done; done
