"""Stage times of fused batch runs (one job -> runs on the calling thread, so nmx_profile_last sees it)."""
import os, sys, time, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from tests import util
L = _lib.lib()
cid = 0
ce = nova_amd.CommitmentEngine(cid)
names = ["digits", "sort", "bounds_plan", "accum", "fold", "reduce", "tail"]
def run(ck, lens, reps=5):
    vecs = [torch.from_numpy(util.random_scalars(cid, m, seed=3 + j)).cuda() for j, m in enumerate(lens)]
    ce.batch_commit(ck, vecs); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): ce.batch_commit(ck, vecs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    L.nmx_set_profiling(1)
    ce.batch_commit(ck, vecs)
    prof = (ctypes.c_float * 16)()
    k = L.nmx_profile_last(prof, 16)
    L.nmx_set_profiling(0)
    st = {names[i]: round(prof[i], 4) for i in range(min(k, 7))}
    print("lens", [len(v) for v in vecs][:4], "... k=%d sum=%d: %.3f ms  stages %s" % (len(lens), sum(lens), dt, st), flush=True)
for lg in (20, 16, 12):
    n = 1 << lg
    ck = ce.setup_synthetic(n, k0=5)
    print("== key 2^%d" % lg)
    run(ck, [n >> i for i in range(8, min(lg, 20))])        # the tiny tail
    run(ck, [n >> i for i in range(4, min(lg, 20))][:16])
    run(ck, [n >> 1, n >> 2, n >> 3])
    run(ck, [n >> i for i in range(1, 8)])
    run(ck, [n >> 3] * 8)
    run(ck, [n >> 3])
    ck.close()
