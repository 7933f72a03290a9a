"""Randomised differential test of nmx_ipa_prove on the GPU box: random curves, sizes, vector placement, layouts, key forms, sparse /
structured witnesses and forced challenges -- every proof equal to the oracle's key-folding restatement (same transcript) and, every
few iterations, checked against the reference's verifier.  usage: gpu_fuzz_ipa.py [iterations] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nova_amd
from nova_amd import _lib
from oracle import cref
from oracle import pyref as R
from tests import ipa_common as ic
from tests.test_gpu_ipa import gpu_prove
L = _lib.lib(); assert L.nmx_init(0) == 0
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
curves = [R.BN254_G1, R.GRUMPKIN, R.PALLAS, R.VESTA]
t0 = time.time()
for it in range(iters):
    curve = curves[rng.integers(0, 4)]
    n = 1 << int(rng.choice([0, 1, 2, 3, 5, 7, 8, 10, 11, 12], p=[.05, .1, .1, .1, .1, .1, .1, .15, .1, .1]))
    kw = dict(device=bool(rng.integers(0, 2)), mont=bool(rng.integers(0, 2)), precompute=bool(rng.integers(0, 4)),
              key_extra=int(rng.choice([0, 0, 1, 37, 5000])))
    ck, ckc, a, b = ic.make_instance(curve, n, int(rng.integers(0, 200)))
    kind = rng.integers(0, 6)
    if kind == 1: a[rng.integers(0, n, size=max(n // 2, 1))] = 0          # sparse witness
    if kind == 2: a[:] = a[0]                                              # all equal
    if kind == 3: b[:] = 0                                                 # c_L = c_R = 0 every round
    if kind == 4: a[: n // 2] = 0                                          # R's MSM part is empty in round 0
    force = {}
    if rng.integers(0, 3) == 0 and n > 1:
        force[int(rng.integers(0, n.bit_length() - 1))] = int(rng.choice([1, 2, curve.r - 1, curve.r - 2]))
    tg, tw = ic.IpaTranscript(curve.r, force=force), ic.IpaTranscript(curve.r, force=force)
    got = gpu_prove(nova_amd, curve, **kw)(ck, ckc, a, b, n, tg)
    want = cref.ipa_prove(curve.cid, ck, ckc, a, b, n, cref.make_ipa_transcript(tw))
    assert got == want and tg.rs == tw.rs, (it, curve.name, n, kw, kind, force)
    if it % 7 == 0:
        assert ic.verify(curve, ck, ckc, a, b, n, *got, tg.rs), (it, "verifier")
    if it % 20 == 19:
        print("iteration %d ok (%.0f s)" % (it + 1, time.time() - t0), flush=True)
print("ipa fuzz ok: %d iterations in %.0f s" % (iters, time.time() - t0))
