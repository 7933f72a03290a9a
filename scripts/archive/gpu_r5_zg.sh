#!/bin/bash
# round 5, call zg: kernel timeline of a 10 538-pair Grumpkin commitment (the small MSM of prove_step)
mkdir -p gpurun_out/r5zg
cat > /tmp/small_msm.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, nova_amd
from tests import util
ce = nova_amd.CommitmentEngine(1)
ck = ce.setup_synthetic(10538, k0=3)
d = torch.from_numpy(util.random_scalars(1, 10538, seed=5)).cuda()
for _ in range(12):
    ce.commit(ck, d)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/sm_trace -- python /tmp/small_msm.py > /tmp/sm_trace.log 2>&1
f=$(find /tmp/sm_trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r5zg/small_msm_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
tail = ks[-26:]
prev = None
tot_run = tot_gap = 0
for st, en, nm in tail:
    gap = (st - prev) / 1e3 if prev else 0.0
    print(f"gap {gap:7.1f} us   run {(en - st) / 1e3:7.1f} us   {nm.replace('void nmx::', '').split('(')[0][:70]}")
    prev = en
PY
