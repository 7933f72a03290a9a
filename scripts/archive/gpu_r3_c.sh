#!/bin/bash
# Round 3, third batch: full GPU suite, same-box A/B of the segment lane rounds, the default bench line, the batched-affine ubench.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3c}
mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d.get("stages_ms"))
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest gpu"
timeout 1500 python -m pytest tests -q -m gpu --maxfail=5 -x -s ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.txt" 2>&1; tail -5 "$OUT/pytest_gpu.txt"; grep -h "IPA-shaped" "$OUT/pytest_gpu.txt"
fi
for rep in 1 2; do
  for lanes in 196608 393216 589824; do
    for lg in ${SIZES:-20}; do
      echo "== rep $rep seg_lanes=$lanes log2n=$lg"
      NMX_TUNE_SEG_LANES=$lanes timeout 300 python bench.py --steps 30 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/bench_lanes${lanes}_${lg}_$rep.json" 2> "$OUT/bench_lanes${lanes}_${lg}_$rep.err"
      show "$OUT/bench_lanes${lanes}_${lg}_$rep.json"
    done
  done
done
echo "== batched-affine ubench"
timeout 300 bench/affine_batch > "$OUT/affine_batch.jsonl" 2> "$OUT/affine_batch.err"; cat "$OUT/affine_batch.jsonl"; tail -2 "$OUT/affine_batch.err"
echo "== default bench line"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"; show "$OUT/bench_default.json"
python - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k,v in (d.get("fieldvec") or {}).items(): print("  fieldvec", k, v)
for k in ("incl_h2d","trait_form","anchor_2p24_single_gpu","prove_step_replay_ms","hyperkzg_replay_ms"): print(" ", k, d.get(k))
PY
echo "== done"
