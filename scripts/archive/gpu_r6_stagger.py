"""batch_stagger on / off, alternating on one lease: three 2^20-1 openings as one batch (HyperKZG kzg_open), the 19-vector batch_commit,
four and two equal 2^20 vectors, and the HyperKZG replay."""
import os, sys, time, argparse
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from tests import util
L = _lib.lib(); assert L.nmx_init(0) == 0
cid, n = 0, 1 << 20
ce = nova_amd.CommitmentEngine(cid)
ck = ce.setup_synthetic(n, k0=5)
vecs = [torch.from_numpy(util.random_scalars(cid, n, seed=40 + j)).cuda() for j in range(4)]
polys = [torch.from_numpy(util.random_scalars(cid, n >> i, seed=50 + i)).cuda() for i in range(1, 20)]
def T(f, reps=9):
    f(); f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), min(ts), r
cases = {"3 openings of 2^20-1": lambda: ce.batch_commit(ck, [v[:n - 1] for v in vecs[:3]]),
         "2 x 2^20": lambda: ce.batch_commit(ck, vecs[:2]),
         "4 x 2^20": lambda: ce.batch_commit(ck, vecs),
         "19 vectors 2^19..2": lambda: ce.batch_commit(ck, polys),
         "one 2^20": lambda: ce.commit(ck, vecs[0])}
ref = {}
for rnd in range(2):
    for on in (0, 1):
        assert L.nmx_set_option(b"batch_stagger", on) == 0
        for name, f in cases.items():
            m, lo, r = T(f)
            key = [(c.xy, c.is_inf) for c in r] if isinstance(r, list) else (r.xy, r.is_inf)
            assert ref.setdefault(name, key) == key, name
            print("pass %d  stagger=%d  %-22s median %.3f  min %.3f ms" % (rnd, on, name, m, lo), flush=True)
import bench
for rnd in range(2):
    for on in (0, 1):
        L.nmx_set_option(b"batch_stagger", on)
        a = argparse.Namespace(log2n=20, steps=7, warmup=2, no_cpu_baseline=True, separate_folds=False)
        out = bench.hyperkzg_replay(a, torch, ck=ck)
        print("pass %d  stagger=%d  hyperkzg_replay 2^20: %.3f ms" % (rnd, on, out["value"]), flush=True)
