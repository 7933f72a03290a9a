#!/bin/bash
# Round 3: single-pass suffix Horner (k_horner_scan) -- parity tests first, then kernel time against the two-pass kernels.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-horner}
mkdir -p "$OUT"
echo "== pytest horner"
timeout 900 python -m pytest tests/test_gpu_fieldvec.py -q -m gpu -k "horner" -x > "$OUT/pytest_horner.txt" 2>&1; tail -3 "$OUT/pytest_horner.txt"
if ! grep -q " passed" "$OUT/pytest_horner.txt" || grep -q "failed" "$OUT/pytest_horner.txt"; then tail -40 "$OUT/pytest_horner.txt"; fi
echo "== kernel time (ms), scan vs two-pass (NMX_TUNE_HORNER_TOP=8)"
for lg in 16 20 22 24; do
  for top in 0 8; do
    NMX_TUNE_HORNER_TOP=$top timeout 300 python bench.py --workload horner --log2n $lg --steps 20 --warmup 3 $([ $lg -ge 24 ] && echo --no-cpu-baseline) > "$OUT/h_${lg}_${top}.json" 2> "$OUT/h_${lg}_${top}.err" || { echo "rc=$? lg=$lg top=$top"; tail -5 "$OUT/h_${lg}_${top}.err"; }
    python - "$OUT/h_${lg}_${top}.json" $lg $top <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"2^{sys.argv[2]} top={sys.argv[3]}  kernel {d['kernel_ms']:.4f} ms  frac {d['roofline']['frac']:.3f}  call {d['ms_per_step']:.4f} ms  matches={d.get('cpu_baseline',{}).get('gpu_matches_cpu')}")
except Exception as e:
    print("no result", sys.argv[1:], e)
PY
  done
done
echo "== large compare"
timeout 900 python -m pytest tests/test_gpu_fieldvec_large.py -q -m gpu -x > "$OUT/pytest_large.txt" 2>&1; tail -2 "$OUT/pytest_large.txt"
echo "== done"
