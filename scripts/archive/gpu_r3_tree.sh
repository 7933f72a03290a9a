#!/bin/bash
# Fused reduction tree with wave-interleaved roles: per level (T1) against fused in 512- (F512) and 256-thread blocks (F256).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3tree}
mkdir -p "$OUT"
echo "== variants test"; timeout 900 python -m pytest tests/test_gpu_pipeline_variants.py tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -2
one() { # label log2n env...
  local label=$1 lg=$2; shift 2
  env "$@" timeout 300 python bench.py --log2n $lg --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$OUT/bench_${label}_$lg.json" 2> "$OUT/bench_${label}_$lg.err"
  python - "$OUT/bench_${label}_$lg.json" "$label" $lg <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stages_ms']
print(f"{sys.argv[2]:5s} 2^{sys.argv[3]:3s} {d['ms_per_step']:.4f} ms  fold {s['fold']:.4f} reduce {s['reduce']:.4f}  tree={d.get('reduction_tree')}")
PY
}
for rep in 1 2; do
for lg in 20 16 13 22; do
  one T1 $lg NMX_TUNE_NO_TREE_FUSE=1
  one F512 $lg NMX_TUNE_NO_TREE_FUSE=2 NMX_TUNE_TREE_THREADS=512
  one F256 $lg NMX_TUNE_NO_TREE_FUSE=2 NMX_TUNE_TREE_THREADS=256
done; done
echo "== done"
