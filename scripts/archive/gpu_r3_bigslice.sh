#!/bin/bash
# Big-bucket pass: block size x pieces per item (slice) swept on the inputs that make big buckets.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3slice}
mkdir -p "$OUT"
echo "== variants test"; timeout 900 python -m pytest tests/test_gpu_pipeline_variants.py -q -x -m gpu 2>&1 | tail -2
for bt in ${BTS:-128 256 512}; do for sl in ${SLS:-256 512 1024 2048}; do
for dist in u1 equal zero_rm1 u10; do
  NMX_TUNE_BIG_THREADS=$bt NMX_TUNE_BIG_SLICE=$sl timeout 300 python bench.py --log2n ${LG:-20} --dist $dist --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$OUT/b_${bt}_${sl}_$dist.json" 2> "$OUT/b_${bt}_${sl}_$dist.err"
  python - "$OUT/b_${bt}_${sl}_$dist.json" $bt $sl $dist <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stages_ms']
print(f"threads {sys.argv[2]:4s} slice {sys.argv[3]:5s} {sys.argv[4]:9s} {d['ms_per_step']:.4f} ms  fold {s['fold']:.4f}")
PY
done; done; done
echo "== done"
