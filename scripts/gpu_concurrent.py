"""Do independent MSMs issued from several host threads overlap on the GPU (each call has its own stream)?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from tests import util
L = _lib.lib(); assert L.nmx_init(0) == 0
g = nova_amd.DlogGroup(0)
for logn in (14, 17, 20):
    n = 1 << logn
    ck = nova_amd.CommitmentKey.generate(0, n, k0=1)
    ds = [torch.from_numpy(util.random_scalars(0, n, seed=i)).cuda() for i in range(4)]
    for d in ds: g.vartime_multiscalar_mul(d, ck)
    reps = 20
    for nthreads in (1, 2, 4):
        def work(i):
            for _ in range(reps): g.vartime_multiscalar_mul(ds[i], ck)
        th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; dt = time.perf_counter() - t
        print(f"2^{logn}: {nthreads} caller thread(s): {dt/(reps*nthreads)*1e3:.3f} ms per MSM  ({n*reps*nthreads/dt/1e6:.0f} M pairs/s aggregate)", flush=True)
    ck.close()
