#!/bin/bash
# One GPU-box session: microbench -> parity tests -> bench -> rocprof kernel trace.  Outputs under gpurun_out/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-run}
mkdir -p "$OUT"
echo "== rocminfo" ; rocminfo | grep -E "Marketing|gfx9" | head -4
nproc > "$OUT/nproc.txt"; lscpu | grep -E "Model name|^CPU\(s\)" >> "$OUT/nproc.txt"
echo "== ubench"
timeout 120 bench/ubench > "$OUT/ubench.jsonl" 2>&1; cat "$OUT/ubench.jsonl"
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest gpu"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee "$OUT/pytest_gpu.txt"
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"; cat "$OUT/bench.json"; tail -5 "$OUT/bench.err"
echo "== host-scalar (PCIe-inclusive) rate"
timeout 300 python - <<'PY' | tee "$OUT/pcie_inclusive.txt"
import time, numpy as np, nova_amd
from nova_amd import _lib
from tests import util
L=_lib.lib(); L.nmx_init(0)
g=nova_amd.DlogGroup(0); n=1<<20
ck=nova_amd.CommitmentKey.generate(0, n)
s=util.random_scalars(0,n)
for _ in range(3): g.vartime_multiscalar_mul(s, ck)
t=time.perf_counter()
for _ in range(10): g.vartime_multiscalar_mul(s, ck)
dt=(time.perf_counter()-t)/10
print(f"BN254 2^20, scalars in pageable host memory (H2D inside the call): {dt*1e3:.3f} ms/MSM, {n/dt/1e6:.1f} M pairs/s")
PY
echo "== rocprof kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o msm -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
find "$OUT/prof" -name "*stats*" | head; f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4,8 "$f" | head -16
echo "== done"
