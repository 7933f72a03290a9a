"""nmx_ipa_prove at 2^14 on Grumpkin alone and BESIDE another thread's 2^20 BN254 commitments (what S2's evaluation argument meets when
S1's HyperKZG runs next to it, nova/mod.rs:862-881), option ipa_priority off / on, alternating."""
import os, sys, time, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from oracle import pyref as R
from tests import ipa_common as ic, util, standin
L = _lib.lib(); assert L.nmx_init(0) == 0
curve, lg = R.GRUMPKIN, 14
n = 1 << lg
ck, ckc, a, b = ic.make_instance(curve, n, 3)
K = nova_amd.CommitmentKey.from_host(curve.cid, ck)
da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
big = nova_amd.CommitmentEngine(0)
bk = big.setup_synthetic(1 << 20, k0=5)
bs = torch.from_numpy(util.random_scalars(0, 1 << 20, seed=9)).cuda()
def run():
    tr = standin.Transcript(seed=5)
    return nova_amd.ipa_prove(K, ckc, da, db, tr.fn_ipa(_lib.IPA_TRANSCRIPT_FN), ctx=tr.ctx)
def timed(reps=9):
    run()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); run(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), min(ts)
stop = False
count = [0]
def load():
    while not stop:
        big.commit(bk, bs); count[0] += 1
ref = run()
for rnd in range(2):
    for prio in (0, 1):
        L.nmx_set_option(b"ipa_priority", prio)
        m, lo = timed()
        print("pass %d  ipa_priority=%d  alone:            median %.3f  min %.3f ms" % (rnd, prio, m, lo), flush=True)
    for prio in (0, 1):
        L.nmx_set_option(b"ipa_priority", prio)
        stop = False; count[0] = 0
        th = threading.Thread(target=load); th.start()
        time.sleep(0.05)
        t0 = time.perf_counter()
        m, lo = timed()
        dt = time.perf_counter() - t0
        stop = True; th.join()
        assert run() == ref
        print("pass %d  ipa_priority=%d  beside 2^20 MSMs: median %.3f  min %.3f ms   (the other thread: %.3f ms per commitment)" % (
            rnd, prio, m, lo, dt * 1e3 / max(count[0], 1)), flush=True)
