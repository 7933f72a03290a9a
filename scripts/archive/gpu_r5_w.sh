#!/bin/bash
# round 5, call w: kernel timeline of the Spartan replay (start / end of every kernel): where a round's 35-50 us go
mkdir -p gpurun_out/r5w
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/sp_trace -- python $GRAFT_REPO_ROOT/bench.py --workload spartan_replay --log2n 20 --steps 3 --warmup 2 --no-cpu-baseline > /tmp/sp_trace.log 2>&1
f=$(find /tmp/sp_trace -name "*kernel_trace.csv" | head -1)
echo "trace: $f" ; wc -l $f
python - "$f" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r5w/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ks.sort()
# the last replay step: find the last occurrence of the z_concat / first kernel pattern: take the final 1/5 of the kernels
names = [k[2] for k in ks]
# locate starts of replays by the k_sc_pass<1, 3> first occurrence after a gap; simpler: print the last 140 kernels with gaps
tail = ks[-150:]
prev_end = None
for st, en, nm in tail:
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    short = nm.replace("void nmx::", "").split("(")[0][:60]
    print(f"gap {gap:8.1f} us   run {(en - st) / 1e3:8.1f} us   {short}")
    prev_end = en
PY
