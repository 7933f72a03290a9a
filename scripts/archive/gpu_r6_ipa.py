"""nmx_ipa_prove timing (native stand-in transcript would be nicer; the Python callback costs ~10 us per round) against the oracle's
key-folding restatement: usage gpu_r6_ipa.py [log2n ...]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from oracle import cref, pyref as R
from tests import ipa_common as ic
L = _lib.lib(); assert L.nmx_init(0) == 0
curve = R.GRUMPKIN
for lg in [int(x) for x in sys.argv[1:]] or [10, 14, 16]:
    n = 1 << lg
    ck, ckc, a, b = ic.make_instance(curve, n, 3)
    K = nova_amd.CommitmentKey.from_host(curve.cid, ck)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    def run():
        tr = ic.IpaTranscript(curve.r)
        return nova_amd.ipa_prove(K, ckc, da, db, tr), tr
    for _ in range(3): got, tg = run()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); run(); ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    tw = ic.IpaTranscript(curve.r)
    want = cref.ipa_prove(curve.cid, ck, ckc, a, b, n, cref.make_ipa_transcript(tw))
    t_cpu = (time.perf_counter() - t0) * 1e3
    print("2^%d  gpu median %.3f ms  min %.3f  (%.3f ms per round)   oracle (%d threads, key fold) %.1f ms   same=%s" % (
        lg, float(np.median(ts)), min(ts), float(np.median(ts)) / lg, cref.get_threads(), t_cpu, got == want and tg.rs == tw.rs), flush=True)
    K.close()
