"""Per-part timing of the HyperKZG replay (bench.py hyperkzg_replay) at 2^20: where do the 10.4 ms go?"""
import os, sys, time, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import nova_amd
from nova_amd import fieldvec as fv
from tests import util
ell = int(os.environ.get("LOG2N", "20")); n = 1 << ell; cid = 0
fid = fv.SCALAR_FIELD_OF_CURVE[cid]
ce = nova_amd.CommitmentEngine(cid)
ck = ce.setup_synthetic(n, k0=5)
hP = util.random_scalars(cid, n, seed=41); xs = util.random_scalars(cid, ell, seed=42)
us = util.random_scalars(cid, 3, seed=43); qs = util.random_scalars(cid, ell, seed=44)
dP = torch.from_numpy(hP).cuda()
def T(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r
def folds():
    polys, cur = [dP], dP
    for i in range(ell - 1):
        cur = fv.fold_pairs(fid, cur, xs[ell - i - 1]); polys.append(cur)
    return polys
t, polys = T(folds); print("folds            %.3f ms" % t)
t, _ = T(lambda: ce.batch_commit(ck, polys[1:])); print("batch_commit all %.3f ms" % t)
for lo, hi in ((1, 2), (2, 3), (3, 5), (5, 8), (8, 20), (1, 8)):
    t, _ = T(lambda: ce.batch_commit(ck, polys[lo:hi])); print("batch_commit polys[%d:%d] (len %s) %.3f ms" % (lo, hi, [len(p) for p in polys[lo:hi]][:4], t))
for j in (1, 2, 4, 6, 8, 10, 14, 19):
    t, _ = T(lambda: ce.commit(ck, polys[j])); print("commit len %7d  %.3f ms" % (len(polys[j]), t))
t, _ = T(lambda: fv.poly_eval_multi(fid, polys, us)); print("poly_eval_multi  %.3f ms" % t)
t, B = T(lambda: fv.lincomb_powers(fid, polys, qs[0])); print("lincomb_powers   %.3f ms" % t)
t, h = T(lambda: fv.div_by_monomial(fid, B, us[0]).contiguous()); print("div_by_monomial  %.3f ms" % t)
t, _ = T(lambda: ce.commit(ck, h)); print("commit n-1       %.3f ms" % t)
def opens():
    out = [None] * 3
    def o(j): out[j] = ce.commit(ck, fv.div_by_monomial(fid, B, us[j]).contiguous())
    ths = [threading.Thread(target=o, args=(j,)) for j in range(3)]
    [t.start() for t in ths]; [t.join() for t in ths]
    return out
t, _ = T(opens); print("3 opens threaded %.3f ms" % t)
