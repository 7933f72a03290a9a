"""Where a small MSM's time goes: stage times (nmx_profile_last) and wall clock for the prove_step shapes (Grumpkin 10 538,
BN254 13 058 / 206 594) and a few powers of two, device-resident scalars, registered keys with tables."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from tests import util
L = _lib.lib(); assert L.nmx_init(0) == 0
names = ["digits", "sort", "bounds", "accum", "fold", "reduce", "tail"]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for cid, n in ((1, 10538), (0, 13058), (0, 206594), (0, 1 << 10), (0, 1 << 13), (0, 1 << 14), (0, 1 << 16)):
    g = nova_amd.DlogGroup(cid)
    ck = nova_amd.CommitmentKey.generate(cid, n, k0=3)
    d = torch.from_numpy(util.random_scalars(cid, n, seed=n)).cuda()
    for _ in range(5): g.vartime_multiscalar_mul(d, ck)
    t = time.perf_counter()
    for _ in range(reps): g.vartime_multiscalar_mul(d, ck)
    wall = (time.perf_counter() - t) / reps * 1e3
    L.nmx_set_profiling(1)
    acc = np.zeros(12)
    for _ in range(reps):
        g.vartime_multiscalar_mul(d, ck)
        buf = (ctypes.c_float * 12)()
        k = L.nmx_profile_last(buf, 12)
        acc[:k] += np.array(buf[:k])
    L.nmx_set_profiling(0)
    st = {names[i]: round(acc[i] / reps, 4) for i in range(7)}
    print(f"curve {cid} n={n}: wall {wall:.4f} ms  stages {st}  sum {sum(st.values()):.4f}", flush=True)
    ck.close()
