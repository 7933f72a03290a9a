"""Large-size checks on the GPU box: BN254 2^22 .. 2^26 on one GPU -- bit-exact vs the oracle up to 2^24, shard
additivity (a size-independent property) beyond -- plus timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nova_amd
from nova_amd import _lib
from oracle import cref
from tests import util
L = _lib.lib(); assert L.nmx_init(0) == 0
g = nova_amd.DlogGroup(0)
cref.set_threads(16)
for logn in (22, 24, 25, 26):
    n = 1 << logn
    t = time.perf_counter(); ck = nova_amd.CommitmentKey.generate(0, n, k0=1); t_gen = time.perf_counter() - t
    s = util.random_scalars(0, n, seed=logn)
    d = torch.from_numpy(s).cuda()
    for _ in range(2): r = g.vartime_multiscalar_mul(d, ck)
    t = time.perf_counter()
    for _ in range(3): r = g.vartime_multiscalar_mul(d, ck)
    dt = (time.perf_counter() - t) / 3
    if logn <= 24:
        t = time.perf_counter(); host = ck.read(0, n); t_rd = time.perf_counter() - t
        t = time.perf_counter(); exp = cref.Prepared(0, host, n).msm(s, n); t_cpu = time.perf_counter() - t
        ok = (r.xy, int(r.is_inf)) == exp
    else:  # beyond the oracle's comfortable size: shard additivity instead (size-independent property)
        t = time.perf_counter()
        parts = [g.vartime_multiscalar_mul(d[o: o + n // 4], ck, offset=o, partial=True).xy for o in range(0, n, n // 4)]
        t_cpu = time.perf_counter() - t
        ok = g.point_sum(np.frombuffer(b"".join(parts), dtype=np.uint8).reshape(4, 128)) == r
    chk = f"CPU oracle {t_cpu:.2f}s ({n/t_cpu/1e6:.2f} M pairs/s, incl. load)" if logn <= 24 else \
        f"check: sum of 4 quarter-key MSMs ({t_cpu*1e3:.1f} ms) == whole"
    print(f"2^{logn}: key gen+tables {t_gen:.2f}s  GPU msm {dt*1e3:.2f} ms ({n/dt/1e6:.0f} M pairs/s)  {chk}  match={ok}", flush=True)
    assert ok
    ck.close()
