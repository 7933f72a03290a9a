"""Small MSMs: own small keys, and prefixes of a 2^20-point key (the HyperKZG batch_commit shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from tests import util
L = _lib.lib(); assert L.nmx_init(0) == 0
g = nova_amd.DlogGroup(0)
big = nova_amd.CommitmentKey.generate(0, 1 << 20, k0=1)
for lg in (1, 4, 6, 8, 9, 10, 11, 12, 13):
    n = 1 << lg
    d = torch.from_numpy(util.random_scalars(0, n, seed=lg)).cuda()
    own = nova_amd.CommitmentKey.generate(0, n, k0=1)
    res = []
    for ck in (own, big):
        for _ in range(3): r = g.vartime_multiscalar_mul(d, ck)
        t = time.perf_counter()
        for _ in range(20): r = g.vartime_multiscalar_mul(d, ck)
        res.append((time.perf_counter() - t) / 20 * 1e3)
        res.append(r.xy[:4].hex())
    print(f"n=2^{lg}: own key {res[0]:.3f} ms  prefix of 2^20 key {res[2]:.3f} ms  same={res[1]==res[3]}", flush=True)
    own.close()
