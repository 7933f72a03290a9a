#!/bin/bash
# round 6: the fused-round form of nmx_ipa_prove: tests, timing, kernel timeline at 2^14
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6ipa2
mkdir -p "$OUT"
timeout 1200 python -m pytest tests/test_gpu_ipa.py -x -q 2>&1 | tail -8 | tee "$OUT/pytest.txt"
python scripts/archive/gpu_r6_ipa.py 10 12 14 16 17 2>&1 | grep -v amdgpu.ids | tee "$OUT/timing.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ipa_trace -- python $GRAFT_REPO_ROOT/scripts/archive/gpu_r6_ipa.py 14 > /tmp/ipa_trace.log 2>&1
f=$(find /tmp/ipa_trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee "$OUT/timeline_2p14.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
idx = [i for i, k in enumerate(ks) if "k_ipa_round" in k[2]]
start = idx[-14]
last = ks[start:]
t0 = last[0][0]
print("kernels of the last proof: %d, span %.1f us" % (len(last), (max(k[1] for k in last) - t0) / 1e3))
prev = None
for st, en, nm, q in last[:45]:
    short = nm.replace("void nmx::", "").split("(")[0][:60]
    gap = (st - prev) / 1e3 if prev else 0.0
    print(f"q{q:>3} start {(st - t0) / 1e3:8.1f}  gap {gap:7.1f}  run {(en - st) / 1e3:7.1f} us  {short}")
    prev = en
PY
