#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for the MSM bench.  usage: gpu_pmc.sh <tag> [bench args]
set -u
export TMPDIR=/tmp
TAG=${1:-pmc}; shift
OUT=gpurun_out/$(date +%H%M%S)_$TAG; mkdir -p $OUT
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/p$i" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > /dev/null 2> "$GRAFT_REPO_ROOT/$OUT/p$i.err" )
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $ctrs -> $f"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if any(t in k for t in ('Accum', 'ReducePair', 'Fold', 'FinalSeg', 'radix', 'DigitsFn', 'k_hist', 'k_part', 'k_tiles')) or 'axpy' in k.lower():
        agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print('   ', c, 'n=%d' % len(v), 'mean=%.4g' % (sum(v)/len(v)), 'max=%.4g' % max(v))
PY
done
