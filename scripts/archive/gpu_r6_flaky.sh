#!/bin/bash
# round 6: the timing-sensitive suites several times over (resident kernel, mailbox evaluations, side streams, several host threads)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6flaky
mkdir -p "$OUT"
for i in 1 2 3 4; do
  timeout 1200 python -m pytest tests/test_gpu_spartan.py tests/test_gpu_stress.py tests/test_gpu_commit_overlap.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
done | tee "$OUT/loops.txt"
for i in 1 2 3 4 5 6; do
  timeout 600 python -m pytest tests/test_gpu_spartan.py -x -q -m gpu -k "several_threads or resident or torn or fallback_round" -p no:cacheprovider 2>&1 | tail -2
done | tee -a "$OUT/loops.txt"
