#!/bin/bash
# Round 3, fourth batch: segment path at small sizes (threshold sweep), batched-affine ubench (fixed operands), prove_step / HyperKZG replays.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3d}
mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d.get("stages_ms"))
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
for lg in ${SIZES:-12 13 14 15 16 17 18}; do
  for thr in ${THRS:-4194304 0}; do
    echo "== log2n=$lg seg_min_total=$thr"
    NMX_TUNE_SEG_MIN_TOTAL=$thr timeout 300 python bench.py --steps 30 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/bench_${lg}_thr$thr.json" 2> "$OUT/bench_${lg}_thr$thr.err"
    show "$OUT/bench_${lg}_thr$thr.json"
  done
done
echo "== batched-affine ubench"
timeout 300 bench/affine_batch > "$OUT/affine_batch.jsonl" 2> "$OUT/affine_batch.err"; cat "$OUT/affine_batch.jsonl"; tail -2 "$OUT/affine_batch.err"
for thr in 4194304 0; do
echo "== prove_step replay seg_min_total=$thr"
NMX_TUNE_SEG_MIN_TOTAL=$thr timeout 300 python bench.py --workload prove_step_replay --steps 10 --warmup 3 > "$OUT/replay_thr$thr.json" 2> "$OUT/replay_thr$thr.err"; python -c "
import json,sys; d=json.loads(open('$OUT/replay_thr$thr.json').read().strip().splitlines()[-1]); print(d['value'], d.get('cpu_baseline',{}).get('gpu_matches_cpu'))"
done
echo "== done"
