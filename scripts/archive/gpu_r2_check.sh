#!/bin/bash
# Round-2 GPU-box session: smoke -> parity tests (all) -> default bench line (with extras).  Outputs under gpurun_out/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r2check}
mkdir -p "$OUT"
rocminfo | grep -E "Marketing|gfx9" | head -2
nproc > "$OUT/nproc.txt"; free -g >> "$OUT/nproc.txt"
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest gpu"
timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -q -m gpu --maxfail=10 --durations=25 ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.txt" 2>&1
tail -60 "$OUT/pytest_gpu.txt"
echo "== bench"
t0=$(date +%s)
timeout 900 python bench.py --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "rc=$? wall=$(( $(date +%s) - t0 ))s"; cat "$OUT/bench.json"; tail -5 "$OUT/bench.err"
echo "== done"
