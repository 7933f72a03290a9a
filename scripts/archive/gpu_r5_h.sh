#!/bin/bash
# Round 5, call H: C++ mirror incl. the sum-check provers, host-table provers, window sweep at 2^21 (the 8-GPU shard size)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5h
mkdir -p "$OUT"
echo "== cpp mirror + spartan"; timeout 1500 python -m pytest tests/test_cpp_mirror.py tests/test_gpu_spartan.py -q -m gpu --maxfail=5 > "$OUT/pytest.txt" 2>&1; tail -5 "$OUT/pytest.txt"
echo "== 2^21 window sweep"
for c in 0 16 17 18 19; do
  timeout 600 python bench.py --log2n 21 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --window-bits $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c=$c', round(d['ms_per_step'],4), d.get('stages_ms'))"
done
echo "== done"
