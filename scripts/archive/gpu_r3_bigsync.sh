#!/bin/bash
# Big-bucket pass: what the hand-over between blocks costs.  default = plain stores + agent fences (L2 write-back / invalidate),
# nofence = timing only (wrong), coh = agent-coherent (sc1) accesses + workgroup fences.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3bigsync}
mkdir -p "$OUT"
echo "== correctness of the coherent variant"; NMX_SO=$PWD/nova_amd/libnova_mi355x_coh.so timeout 900 python -m pytest tests/test_gpu_pipeline_variants.py tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -2
for cfg in "128 512" "256 1024" "512 2048"; do set -- $cfg; bt=$1; sl=$2
for v in default nofence coh; do
  so=$PWD/nova_amd/libnova_mi355x.so; [ $v != default ] && so=$PWD/nova_amd/libnova_mi355x_$v.so
  for dist in u1 equal u10; do
    NMX_SO=$so NMX_TUNE_BIG_THREADS=$bt NMX_TUNE_BIG_SLICE=$sl timeout 300 python bench.py --log2n 20 --dist $dist --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$OUT/b_${v}_${bt}_$dist.json" 2> "$OUT/b_${v}_${bt}_$dist.err"
    python - "$OUT/b_${v}_${bt}_$dist.json" $v $bt $sl $dist <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stages_ms']
print(f"{sys.argv[2]:8s} threads {sys.argv[3]:4s} slice {sys.argv[4]:5s} {sys.argv[5]:6s} {d['ms_per_step']:.4f} ms  fold {s['fold']:.4f}")
PY
  done
done; done
echo "== done"
