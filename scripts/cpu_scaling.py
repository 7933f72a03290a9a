"""CPU-baseline thread scaling probe (oracle/nova_ref.c on the GPU box's host cores)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cref, pyref as R
from tests import util
n = 1 << 18
c = R.BN254_G1
cref.set_threads(os.cpu_count())
b = cref.sequential_bases(c, 1, n)
s = util.random_scalars(0, n)
p = cref.Prepared(0, b, n)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cgroup cpu.max: n/a", e)
for th in (1, 8, 16, 32, 64, 128, 256):
    cref.set_threads(th)
    best = 1e9
    for _ in range(2):
        t = time.perf_counter(); p.msm(s, n); best = min(best, time.perf_counter() - t)
    print(f"threads={th:4d}  2^18 BN254 msm: {best*1e3:8.1f} ms  {n/best/1e6:6.2f} M pairs/s", flush=True)
