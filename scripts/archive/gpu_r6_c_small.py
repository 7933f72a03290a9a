"""Window width of the tables of a 2^14-point (and 2^12 / 2^16) key: one commitment of full-width scalars, and the inner-product argument
(14 rounds of a fused two-vector MSM with n / 2 non-zero scalars each), c forced at registration (nmx_set_window_bits)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from oracle import pyref as R
from tests import ipa_common as ic, util, standin
L = _lib.lib(); assert L.nmx_init(0) == 0
curve = R.GRUMPKIN
ce = nova_amd.CommitmentEngine(curve.cid)
def med(f, reps=9):
    f(); f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
for lg in (12, 14, 16):
    n = 1 << lg
    ck_host, ckc, a, b = ic.make_instance(curve, n, 3)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    ref = None
    for rnd in range(2):
        for c in (0, 8, 10, 11, 12, 13, 14, 15, 16):
            L.nmx_set_window_bits(c)
            K = nova_amd.CommitmentKey.from_host(curve.cid, ck_host)
            L.nmx_set_window_bits(0)
            def ipa():
                tr = standin.Transcript(seed=5)
                return nova_amd.ipa_prove(K, ckc, da, db, tr.fn_ipa(_lib.IPA_TRANSCRIPT_FN), ctx=tr.ctx)
            got = ipa()
            if ref is None: ref = got
            assert got == ref
            t_ipa = med(ipa)
            t_commit = med(lambda: ce.commit(K, da))
            print("pass %d  2^%d  c=%2d  commit %.3f ms   ipa %.3f ms (%.3f per round)" % (rnd, lg, c, t_commit, t_ipa, t_ipa / lg), flush=True)
            K.close()
