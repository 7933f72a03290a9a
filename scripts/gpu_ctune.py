"""Window-width sweep for large registered keys (tables built with the forced width)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes
import nova_amd
from nova_amd import _lib
from tests import util
L = _lib.lib(); assert L.nmx_init(0) == 0
g = nova_amd.DlogGroup(0)
prof = (ctypes.c_float * 16)()
for logn in (19, 20, 21, 22):
    n = 1 << logn
    d = torch.from_numpy(util.random_scalars(0, n, seed=logn)).cuda()
    ref = None
    for c in (15, 16, 17, 18, 19, 20):
        L.nmx_set_window_bits(c)
        ck = nova_amd.CommitmentKey.generate(0, n, k0=1)
        for _ in range(2): r = g.vartime_multiscalar_mul(d, ck)
        L.nmx_set_profiling(1)
        t = time.perf_counter()
        for _ in range(8): r = g.vartime_multiscalar_mul(d, ck)
        dt = (time.perf_counter() - t) / 8
        k = L.nmx_profile_last(prof, 16)
        L.nmx_set_profiling(0)
        if ref is None: ref = r.xy
        print(f"2^{logn} c={c}: {dt*1e3:8.3f} ms  {n/dt/1e6:6.0f} M pairs/s  same={r.xy==ref}  stages={[round(x,3) for x in prof[:k]]}", flush=True)
        ck.close()
L.nmx_set_window_bits(0)
