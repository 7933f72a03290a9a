#!/bin/bash
# round 6: where do the 2.0 ms of HyperKZG's batch_commit (19 vectors, 2^20 - 2 pairs in all) go against 1.5 ms for one 2^20 MSM?
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6bt
mkdir -p "$OUT"
python scripts/archive/gpu_r6_batch_timeline.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/parts.txt"
cd /tmp && export TMPDIR=/tmp
TRACE_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/bt_trace -- python $GRAFT_REPO_ROOT/scripts/archive/gpu_r6_batch_timeline.py > /tmp/bt_trace.log 2>&1
f=$(find /tmp/bt_trace -name "*kernel_trace.csv" | head -1)
echo "trace: $f"; wc -l "$f"
python - "$f" <<'PY' | tee "$OUT/timeline.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows)
# the last batch call: kernels after the last gap > 300 us
cut = 0
for i in range(1, len(ks)):
    if ks[i][0] - max(k[1] for k in ks[max(0, i - 8):i]) > 300_000: cut = i
last = ks[cut:]
t0 = last[0][0]
print("kernels of the last batch_commit: %d, span %.1f us" % (len(last), (max(k[1] for k in last) - t0) / 1e3))
for st, en, nm, q, s in last:
    short = nm.replace("void nmx::", "").split("(")[0][:70]
    print(f"q{q:>3} s{s:>3}  start {(st - t0) / 1e3:8.1f}  end {(en - t0) / 1e3:8.1f}  run {(en - st) / 1e3:7.1f} us  {short}")
PY
