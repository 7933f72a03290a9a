"""HyperKZG batch_commit at 2^20 (19 vectors 2^19 .. 2): whole call against its jobs alone and against other groupings."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import nova_amd
from nova_amd import fieldvec as fv
from tests import util
ell = 20; n = 1 << ell; cid = 0
ce = nova_amd.CommitmentEngine(cid)
ck = ce.setup_synthetic(n, k0=5)
polys = [torch.from_numpy(util.random_scalars(cid, n >> i, seed=50 + i)).cuda() for i in range(1, ell)]
def T(f, reps=int(os.environ.get("REPS", "7"))):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), min(ts)
if os.environ.get("TRACE_ONLY"):
    for _ in range(3): ce.batch_commit(ck, polys)
    torch.cuda.synchronize(); sys.exit(0)
print("batch_commit 19 vectors        median %.3f  min %.3f ms" % T(lambda: ce.batch_commit(ck, polys)))
print("commit 2^19 alone              median %.3f  min %.3f ms" % T(lambda: ce.commit(ck, polys[0])))
print("commit 2^18 alone              median %.3f  min %.3f ms" % T(lambda: ce.commit(ck, polys[1])))
print("commit 2^17 alone              median %.3f  min %.3f ms" % T(lambda: ce.commit(ck, polys[2])))
print("batch 16 short (fused) alone   median %.3f  min %.3f ms" % T(lambda: ce.batch_commit(ck, polys[3:])))
print("batch [2^19, 2^18, 2^17]       median %.3f  min %.3f ms" % T(lambda: ce.batch_commit(ck, polys[:3])))
print("batch [2^19, 2^18]             median %.3f  min %.3f ms" % T(lambda: ce.batch_commit(ck, polys[:2])))
print("batch [2^18 .. 2] (18 vectors) median %.3f  min %.3f ms" % T(lambda: ce.batch_commit(ck, polys[1:])))
def seq():
    ce.commit(ck, polys[0]); ce.commit(ck, polys[1]); ce.commit(ck, polys[2]); ce.batch_commit(ck, polys[3:])
print("the four jobs one after another median %.3f  min %.3f ms" % T(seq))
full = torch.from_numpy(util.random_scalars(cid, n, seed=7)).cuda()
print("commit 2^20 (one MSM)          median %.3f  min %.3f ms" % T(lambda: ce.commit(ck, full)))
print("batch_commit 19 vectors again  median %.3f  min %.3f ms" % T(lambda: ce.batch_commit(ck, polys)))
