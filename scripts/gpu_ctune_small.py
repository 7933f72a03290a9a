"""Window-width sweep for the prove_step-sized MSMs (10 k - 200 k pairs, registered keys with tables)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes
import nova_amd
from nova_amd import _lib
from tests import util
L = _lib.lib(); assert L.nmx_init(0) == 0
g = nova_amd.DlogGroup(0)
prof = (ctypes.c_float * 16)()
for n in (4096, 10538, 13058, 32768, 65536, 131072, 206594, 262144, 524288):
    d = torch.from_numpy(util.random_scalars(0, n, seed=n)).cuda()
    ref = None
    L.nmx_set_window_bits(0)
    best = None
    for c in (0, 8, 9, 10, 11, 12, 13, 14, 15, 16):
        if c and (1 << c) > 8 * n: continue
        L.nmx_set_window_bits(c)
        ck = nova_amd.CommitmentKey.generate(0, n, k0=1)
        for _ in range(3): r = g.vartime_multiscalar_mul(d, ck)
        L.nmx_set_profiling(1)
        t = time.perf_counter()
        for _ in range(10): r = g.vartime_multiscalar_mul(d, ck)
        dt = (time.perf_counter() - t) / 10
        k = L.nmx_profile_last(prof, 16)
        L.nmx_set_profiling(0)
        if ref is None: ref = r.xy
        print(f"n={n} c={c or 'auto'}: {dt*1e3:7.3f} ms  same={r.xy==ref}  stages={[round(x,3) for x in prof[:k]]}", flush=True)
        ck.close()
L.nmx_set_window_bits(0)
