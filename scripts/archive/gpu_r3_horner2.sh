#!/bin/bash
# Round 3: single-pass suffix Horner, sub-tiles per wave (NMX_TUNE_HORNER_SUB = 1 / 2 / 4) against the two-pass kernels.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-horner2}
mkdir -p "$OUT"
echo "== pytest horner (default), then with 2 and 4 sub-tiles per wave at every size"
timeout 900 python -m pytest tests/test_gpu_fieldvec.py -q -m gpu -k "horner" -x > "$OUT/pytest_horner.txt" 2>&1; tail -2 "$OUT/pytest_horner.txt"
for sub in 2 4; do
NMX_TUNE_HORNER_SUB=$sub timeout 900 python -m pytest tests/test_gpu_fieldvec.py -q -m gpu -k "horner" -x > "$OUT/pytest_horner_sub$sub.txt" 2>&1; tail -2 "$OUT/pytest_horner_sub$sub.txt"
done
grep -l "failed\|Error" "$OUT"/pytest_horner*.txt | while read f; do tail -30 "$f"; done
echo "== kernel time (ms)"
for lg in ${LGS:-16 18 20 22 24}; do
  for cfg in "0 1" "0 2" "0 4" "8 0"; do
    set -- $cfg; top=$1; sub=$2
    NMX_TUNE_HORNER_TOP=$top NMX_TUNE_HORNER_SUB=$sub timeout 300 python bench.py --workload horner --log2n $lg --steps 20 --warmup 3 $([ $lg -ge 24 ] && echo --no-cpu-baseline) > "$OUT/h_${lg}_${top}_${sub}.json" 2> "$OUT/h_${lg}_${top}_${sub}.err" || { echo "rc=$? lg=$lg top=$top sub=$sub"; tail -5 "$OUT/h_${lg}_${top}_${sub}.err"; }
    python - "$OUT/h_${lg}_${top}_${sub}.json" $lg $top $sub <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"2^{sys.argv[2]} top={sys.argv[3]} sub={sys.argv[4]}  kernel {d['kernel_ms']:.4f} ms  frac {d['roofline']['frac']:.3f}  call {d['ms_per_step']:.4f} ms  matches={d.get('cpu_baseline',{}).get('gpu_matches_cpu')}")
except Exception as e:
    print("no result", sys.argv[1:], e)
PY
  done
done
echo "== done"
