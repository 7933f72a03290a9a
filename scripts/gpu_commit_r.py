"""commit(v, r) with r = 0 against r != 0: the cost of the host-side h * r."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from tests import util
L = _lib.lib(); assert L.nmx_init(0) == 0
for cid in (0, 2):
    ce = nova_amd.CommitmentEngine(cid)
    for lg in (13, 20):
        n = 1 << lg
        ck = nova_amd.CommitmentKey.generate(cid, n, k0=1)
        d = torch.from_numpy(util.random_scalars(cid, n, seed=lg)).cuda()
        for name, r in (("r=0", bytes(32)), ("r!=0", util.random_scalars(cid, 1, seed=7))):
            for _ in range(3): c = ce.commit(ck, d, r)
            t = time.perf_counter()
            for _ in range(20): c = ce.commit(ck, d, r)
            print(f"curve {cid} n=2^{lg} {name}: {(time.perf_counter()-t)/20*1e3:.3f} ms", flush=True)
        ck.close()
