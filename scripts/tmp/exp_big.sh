set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -k "large or variants or parity or fuzz or slice or emul" 2>&1 | tail -4
for lg in 20 22 23 24; do for np in 0 1; do
  NMX_TUNE_NO_PARTITION=$np timeout 600 python bench.py --log2n $lg --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^$lg no_partition=$np', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1),'M/s', d['stages_ms'])"
done; done
for v in 589824 1179648 2359296; do
NMX_TUNE_SEG_LANES=$v timeout 600 python bench.py --log2n 24 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^24 seg_lanes=$v', round(d['ms_per_step'],3), 'ms', d['stages_ms'])"
done
