#!/bin/bash
# Round 3, second batch: full GPU suite, segment lane multiples (pieces per bucket for the final pass), in-process multi-GPU mode.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3b}
mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d.get("stages_ms"))
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
echo "== pytest gpu"
timeout 1500 python -m pytest tests -q -m gpu --maxfail=5 -x -s ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.txt" 2>&1; tail -12 "$OUT/pytest_gpu.txt"; grep -h "IPA-shaped" "$OUT/pytest_gpu.txt"
for lanes in ${LANES:-0 196608 294912 589824}; do
  for lg in ${SIZES:-20 21 18}; do
    echo "== seg_lanes=$lanes log2n=$lg"
    NMX_TUNE_SEG_LANES=$lanes timeout 300 python bench.py --steps 20 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/bench_lanes${lanes}_$lg.json" 2> "$OUT/bench_lanes${lanes}_$lg.err"
    show "$OUT/bench_lanes${lanes}_$lg.json"
  done
done
for lg in ${BIG:-22 24}; do
  echo "== log2n=$lg (default lanes)"
  timeout 600 python bench.py --steps 5 --warmup 2 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/bench_big_$lg.json" 2> "$OUT/bench_big_$lg.err"
  show "$OUT/bench_big_$lg.json"
done
echo "== in-process multi-GPU mode (fallback expected on a 1-GPU box)"
timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --total-log2n 22 > "$OUT/bench_inproc_gpus2.json" 2> "$OUT/bench_inproc_gpus2.err"; echo "rc=$?"; tail -1 "$OUT/bench_inproc_gpus2.err"; show "$OUT/bench_inproc_gpus2.json"
echo "== default bench line"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"; show "$OUT/bench_default.json"
python - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k,v in (d.get("fieldvec") or {}).items(): print("  fieldvec", k, v)
for k in ("incl_h2d","trait_form","anchor_2p24_single_gpu","prove_step_replay_ms","hyperkzg_replay_ms","cpu_baseline"): print(" ", k, d.get(k))
PY
echo "== done"
