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

# slice form through the cache (what the reference's signature reaches) against the CPU oracle on ONE thread: the basis of
# nmx_min_gpu_n (the shim keeps MSMs below it on the CPU)
from oracle import cref
cref.set_threads(1)
print("slice form (cached key, host scalars) vs the CPU oracle on one thread:")
for lg in (1, 2, 4, 5, 6, 7, 8, 10, 12):
    n = 1 << lg
    host_bases = big.read(0, n)
    sc = util.random_scalars(0, n, seed=100 + lg)
    for _ in range(3): r = g.vartime_multiscalar_mul(sc, host_bases)
    t = time.perf_counter()
    for _ in range(20): r = g.vartime_multiscalar_mul(sc, host_bases)
    gpu = (time.perf_counter() - t) / 20 * 1e3
    t = time.perf_counter()
    reps = 5
    for _ in range(reps): e = cref.msm(0, sc, host_bases, n)
    cpu = (time.perf_counter() - t) / reps * 1e3
    print(f"n=2^{lg}: GPU slice form {gpu:.3f} ms   CPU oracle (1 thread) {cpu:.3f} ms   match={(r.xy, int(r.is_inf)) == e}", flush=True)
