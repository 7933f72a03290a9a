"""PCIe-inclusive MSM rate: scalars start in (pageable / pinned) host memory, the H2D copy is inside the call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from tests import util
L = _lib.lib(); assert L.nmx_init(0) == 0
g = nova_amd.DlogGroup(0)
for lg in (20, 24):
    n = 1 << lg
    ck = nova_amd.CommitmentKey.generate(0, n, k0=1)
    s = util.random_scalars(0, n, seed=lg)
    pinned = torch.from_numpy(s.copy()).pin_memory().numpy()
    d = torch.from_numpy(s).cuda()
    for name, arg in (("HBM-resident", d), ("pageable host", s), ("pinned host", pinned)):
        for _ in range(3): r = g.vartime_multiscalar_mul(arg, ck)
        t = time.perf_counter()
        for _ in range(10): r = g.vartime_multiscalar_mul(arg, ck)
        dt = (time.perf_counter() - t) / 10
        print(f"BN254 2^{lg}, scalars {name}: {dt*1e3:.3f} ms/MSM, {n/dt/1e6:.1f} M pairs/s", flush=True)
    ck.close()
