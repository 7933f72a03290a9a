"""Window-width sweep of the table path on the CURRENT pipeline at the shard sizes of BASELINE configs[2]:
c in {16..20} x n in {2^20 .. 2^23}, two alternating passes on one lease; per point the wall time of a full MSM call and the stage split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes
import nova_amd
from nova_amd import _lib
from tests import util
L = _lib.lib(); assert L.nmx_init(0) == 0
g = nova_amd.DlogGroup(0)
prof = (ctypes.c_float * 16)()
for rep in range(2):
    for logn in (20, 21, 22, 23):
        n = 1 << logn
        d = [torch.from_numpy(util.random_scalars(0, n, seed=logn + 100 * j)).cuda() for j in range(2)]
        ref = None
        for c in (16, 17, 18, 19, 20):
            if logn == 20 and c > 18:
                continue
            L.nmx_set_window_bits(c)
            try:
                ck = nova_amd.CommitmentKey.generate(0, n, k0=1)
            except nova_amd.NmxError as e:
                print(f"2^{logn} c={c}: {e}", flush=True)
                continue
            for j in range(3): r = g.vartime_multiscalar_mul(d[j & 1], ck)
            L.nmx_set_profiling(1)
            ts = []
            for j in range(10):
                t = time.perf_counter()
                r = g.vartime_multiscalar_mul(d[j & 1], ck)
                ts.append(time.perf_counter() - t)
            k = L.nmx_profile_last(prof, 16)
            L.nmx_set_profiling(0)
            if ref is None: ref = r.xy
            print(f"pass {rep} 2^{logn} c={c}: median {np.median(ts)*1e3:7.3f} ms  min {min(ts)*1e3:7.3f}  {n/np.median(ts)/1e6:6.0f} M pairs/s  same={r.xy==ref}  "
                  f"stages={[round(x,3) for x in prof[:k]]}", flush=True)
            ck.close()
        del d
L.nmx_set_window_bits(0)
