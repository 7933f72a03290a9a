#!/bin/bash
# lmax sweep for the precomputed-table path
for l in 16 24 32 48 64; do
  echo "== lmax $l"
  NMX_TUNE_LMAX=$l timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['stages_ms'])"
done
echo "== plain path (no tables), default"
timeout 300 python - <<'PY'
import time, numpy as np, torch, nova_amd
from nova_amd import _lib
from tests import util
L=_lib.lib(); L.nmx_init(0)
g=nova_amd.DlogGroup(0)
for logn in (14, 16, 18, 20, 22):
    n=1<<logn
    for pre in (True, False):
        ck=nova_amd.CommitmentKey.generate(0, n, precompute=pre)
        s=torch.from_numpy(util.random_scalars(0,n).copy()).cuda()
        for _ in range(3): g.vartime_multiscalar_mul(s, ck)
        t=time.perf_counter()
        for _ in range(5): g.vartime_multiscalar_mul(s, ck)
        dt=(time.perf_counter()-t)/5
        print(f"2^{logn} precompute={pre}: {dt*1e3:.3f} ms  {n/dt/1e6:.1f} Mpairs/s", flush=True)
        ck.close()
PY
