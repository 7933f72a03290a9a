#!/bin/bash
# kernel trace of the fold stage on inputs with big buckets
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3bigtrace}
mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
for sl in 4096 1024; do for dist in u1 equal u10; do
  ( cd /tmp && NMX_TUNE_BIG_SLICE=$sl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/t_${sl}_$dist" -o t -- python "$R/bench.py" --log2n 20 --dist $dist --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$R/$OUT/b_${sl}_$dist.json" 2> "$R/$OUT/t_${sl}_$dist.err" )
  f=$(find "$OUT/t_${sl}_$dist" -name "*kernel_stats.csv" | head -1)
  echo "== slice $sl $dist"; [ -n "$f" ] && cut -d, -f1-4 "$f" | sed 's/void nmx:://; s/(nmx::[^"]*"/"/; s/(nmx::.*)//' | grep -v "Precomp\|GenFn\|Identity" | cut -c1-100 | head -14
done; done
