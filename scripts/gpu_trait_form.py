"""trait form (nmx_msm slice form, host scalars in halo2curves' layout, bases through the slice cache) and the handle form with
host scalars, per host_split setting, at 2^20 (and 2^19 / 2^21): ms per call, all results compared with the HBM-resident run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from tests import util
L = _lib.lib(); assert L.nmx_init(0) == 0
g = nova_amd.DlogGroup(0)
for lg in (20, 19, 21):
    n = 1 << lg
    ck = nova_amd.CommitmentKey.generate(0, n, k0=1)
    sc = util.random_scalars(0, n, seed=lg)
    d = torch.from_numpy(sc).cuda()
    ref = g.vartime_multiscalar_mul(d, ck)
    t = time.perf_counter()
    for _ in range(5): g.vartime_multiscalar_mul(d, ck)
    base = (time.perf_counter() - t) / 5 * 1e3
    row = [f"2^{lg}: resident {base:.3f} ms"]
    for k in (0, 255, 2, 3, 4, 255, 0):
        assert L.nmx_set_option(b"host_split", k) == 0
        for _ in range(2): r = g.vartime_multiscalar_mul(sc, ck)
        ts = []
        for _ in range(7):
            t = time.perf_counter(); r = g.vartime_multiscalar_mul(sc, ck); ts.append(time.perf_counter() - t)
        row.append(f"split {k}: {np.median(ts)*1e3:.3f} ms ok={r == ref}")
    print(" | ".join(row), flush=True)
    ck.close()
L.nmx_set_option(b"host_split", 255)
