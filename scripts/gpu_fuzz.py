"""Randomised differential test on the GPU box: MSM entry points against the oracle over random curves, sizes, scalar
distributions, key registration modes and offsets.  usage: gpu_fuzz.py [iterations] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from oracle import cref
from oracle import pyref as R
from tests import util
L = _lib.lib(); assert L.nmx_init(0) == 0
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
curves = [R.BN254_G1, R.GRUMPKIN, R.PALLAS, R.VESTA]
kinds = ["random", "equal", "zero_rm1", "pm_small", "u1", "u10", "u16", "u64"]
keys = {}
t0 = time.time()
for it in range(iters):
    c = curves[rng.integers(0, 4)]
    g = nova_amd.DlogGroup(c.cid)
    nk = int(rng.choice([1, 2, 17, 100, 1000, 4095, 4096, 5000, 20000, 70000]))
    if (c.cid, nk) not in keys:
        host = cref.sequential_bases(c, int(rng.integers(1, 1000)), nk).copy()
        for j in rng.integers(0, nk, size=min(3, nk)):   # a few identity bases and duplicates
            host[j] = 0 if rng.integers(0, 2) else host[0]
        keys[(c.cid, nk)] = (host, nova_amd.CommitmentKey.from_host(c.cid, host, precompute=bool(rng.integers(0, 2))))
    host, ck = keys[(c.cid, nk)]
    if it % 5 == 4:   # a ragged batch (fused runs on keys with tables, one MSM per vector otherwise)
        kv = int(rng.choice([1, 2, 3, 7, 16, 17, 40]))
        lens = [int(rng.integers(0, nk + 1)) if rng.integers(0, 3) else int(rng.integers(0, min(nk, 9) + 1)) for _ in range(kv)]
        fk = [kk for kk in kinds if not kk.startswith("u")]
        vecs = [util.scalar_set(c.cid, max(m, 1), fk[rng.integers(0, len(fk))], seed=int(rng.integers(0, 1 << 30)))[:m] for m in lens]
        exp = [cref.msm(c.cid, v, host[:len(v)], len(v)) if len(v) else (bytes(64), 1) for v in vecs]
        mode = int(rng.integers(0, 3))
        if mode == 0:
            got = g.batch_vartime_multiscalar_mul(vecs, ck)
        elif mode == 1:
            got = g.batch_vartime_multiscalar_mul([torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in vecs], ck)
        else:
            got = g.batch_vartime_multiscalar_mul(vecs, host)
        if [(x.xy, int(x.is_inf)) for x in got] != exp:
            print(f"MISMATCH (batch) it={it} curve={c.name} nk={nk} lens={lens} mode={mode}", flush=True)
            sys.exit(1)
        continue
    n = int(rng.integers(0, nk + 1))
    off = int(rng.integers(0, nk - n + 1))
    kind = kinds[rng.integers(0, len(kinds))]
    mode = int(rng.integers(0, 3))
    if kind.startswith("u"):
        bits = {"u1": 1, "u10": 10, "u16": 16, "u64": 64}[kind]
        s = util.small_scalars(max(n, 1), bits)[:n]
        exp = cref.msm_u64(c.cid, s, host[off:off + n], n) if n else (bytes(64), 1)
        if off == 0 and mode != 2:
            got = g.vartime_multiscalar_mul_small(s, ck)
        else:
            got = g.vartime_multiscalar_mul_small(s, np.ascontiguousarray(host[off:off + n]))
    else:
        s = util.scalar_set(c.cid, max(n, 1), kind, seed=int(rng.integers(0, 1 << 30)))[:n]
        exp = cref.msm(c.cid, s, host[off:off + n], n) if n else (bytes(64), 1)
        if mode == 0:
            got = g.vartime_multiscalar_mul(s, ck, offset=off)
        elif mode == 1:
            got = g.vartime_multiscalar_mul(torch.from_numpy(np.ascontiguousarray(s)).cuda() if n else s, ck, offset=off)
        else:
            got = g.vartime_multiscalar_mul(s, np.ascontiguousarray(host[off:off + n]))
    if (got.xy, int(got.is_inf)) != exp:
        print(f"MISMATCH it={it} curve={c.name} nk={nk} n={n} off={off} kind={kind} mode={mode}", flush=True)
        sys.exit(1)
print(f"fuzz ok: {iters} cases in {time.time() - t0:.1f}s", flush=True)
