#!/bin/bash
# window-width sweep of the table path with the hand-written partition (sizes of prove_step / HyperKZG batch commits)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-csweep}
mkdir -p "$OUT"
for lg in 13 14 15 16 17 18 19; do
  for c in 8 9 10 11 12 13 14 15 16; do
    NMX_TUNE_PRECOMP_MIN_N=2 timeout 120 python bench.py --steps 20 --warmup 5 --log2n $lg --window-bits $c --no-extras --no-cpu-baseline > "$OUT/b.json" 2> "$OUT/b.err"
    python - "$OUT/b.json" $lg $c <<'PY' | tee -a "$OUT/window_width_sweep_partition.txt"
import json,sys
try:
    d=json.load(open(sys.argv[1])); s=d["stages_ms"]
    print(f"2^{sys.argv[2]} c={sys.argv[3]:>2}: {d['ms_per_step']:.4f} ms  " + " ".join(f"{k}={v:.3f}" for k,v in s.items()))
except Exception as e:
    print(f"2^{sys.argv[2]} c={sys.argv[3]}: ERR {e}")
PY
  done
done
echo "== seg lanes sweep at 2^20 / 2^21"
for lg in 20 21; do for l in 131072 196608 294912 393216 589824 786432; do
  NMX_TUNE_SEG_LANES=$l timeout 120 python bench.py --steps 20 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/b.json" 2> "$OUT/b.err"
  python - "$OUT/b.json" $lg $l <<'PY' | tee -a "$OUT/seg_lanes_sweep.txt"
import json,sys
d=json.load(open(sys.argv[1])); s=d["stages_ms"]
print(f"2^{sys.argv[2]} lanes={sys.argv[3]:>7}: {d['ms_per_step']:.4f} ms  accum={s['accum']:.4f} fold={s['fold']:.4f}")
PY
done; done
echo "== seg threshold: 2^17..2^19 with seg forced on / off"
for lg in 17 18 19; do for v in 0 0xffffffff; do
  NMX_TUNE_SEG_MIN_TOTAL=$v timeout 120 python bench.py --steps 20 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/b.json" 2> "$OUT/b.err"
  python - "$OUT/b.json" $lg $v <<'PY' | tee -a "$OUT/seg_threshold.txt"
import json,sys
d=json.load(open(sys.argv[1])); s=d["stages_ms"]
print(f"2^{sys.argv[2]} seg_min_total={sys.argv[3]:>10}: {d['ms_per_step']:.4f} ms  accum={s['accum']:.4f} fold={s['fold']:.4f}")
PY
done; done
echo "== done"
