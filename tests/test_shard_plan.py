"""Multi-device keys behind the C ABI, the part that needs no GPU: nmx_shard_plan is the rule the library uses to cut a
key into per-device shards and a call into per-shard pieces (capi.hip shard_range / parts_of).  It must agree with the
one-process-per-GPU layout of nova_amd/dist.py (the reference's par_chunks, /root/reference/src/provider/msm.rs:564-574),
cover every pair exactly once, and the sharded sum itself -- per-shard oracle MSMs as 128-byte partials through
nmx_point_sum, the same combine the library runs -- must equal the whole MSM."""
import numpy as np
import pytest

from nova_amd import _lib, shard_plan
from nova_amd.dist import shard_range
from oracle import cref
from oracle import pyref as R
from tests import util


@pytest.mark.parametrize("n_key,k", [(1 << 20, 8), (1000003, 7), (17, 4), (5, 8), (1, 1), (1 << 24, 3)])
def test_plan_matches_dist_shard_range(n_key, k):
    whole = shard_plan(n_key, k, 0, n_key)
    want = [(i, 0, hi - lo) for i, (lo, hi) in enumerate(shard_range(n_key, i, k) for i in range(k)) if hi > lo]
    assert whole == want
    rng = np.random.default_rng(n_key + k)
    for _ in range(50):
        off = int(rng.integers(0, n_key + 1))
        n = int(rng.integers(0, n_key - off + 1))
        plan = shard_plan(n_key, k, off, n)
        covered = []
        for dev, poff, cnt in plan:
            lo, hi = shard_range(n_key, dev, k)
            assert cnt > 0 and lo + poff + cnt <= hi
            covered.append((lo + poff, lo + poff + cnt))
        # the pieces tile [off, off + n) in order, without gaps or overlaps
        assert sum(b - a for a, b in covered) == n
        for (a0, b0), (a1, b1) in zip(covered, covered[1:]):
            assert b0 == a1
        if n:
            assert covered[0][0] == off and covered[-1][1] == off + n
    L = _lib.lib()
    assert L.nmx_shard_plan(10, 0, 0, 1, None, 0) == _lib.E_ARG
    assert L.nmx_shard_plan(10, 2, 8, 3, None, 0) == _lib.E_ARG    # offset + n beyond the key


@pytest.mark.parametrize("c", [R.BN254_G1, R.VESTA], ids=lambda c: c.name)
def test_sharded_sum_of_oracle_partials_is_the_msm(c):
    """Two and three fake devices: the per-shard MSM is injected (the oracle: there is no GPU here), the plan and the
    combine are the library's."""
    L = _lib.lib()
    n_key = 301
    bases = cref.sequential_bases(c, 77, n_key)
    sc = util.random_scalars(c.cid, n_key)
    for k in (2, 3):
        for off, n in ((0, n_key), (5, 250), (150, 1), (100, 0)):
            parts = []
            for dev, poff, cnt in shard_plan(n_key, k, off, n):
                lo, _ = shard_range(n_key, dev, k)
                g = lo + poff
                xy, inf = cref.msm(c.cid, sc[g - off:g - off + cnt], bases[g:g + cnt], cnt)
                parts.append(util.affine_to_partial(c.p, xy, inf))
            buf = np.frombuffer(b"".join(parts) or bytes(128), dtype=np.uint8)
            out = np.zeros(64, np.uint8)
            inf = np.zeros(1, np.uint8)
            assert L.nmx_point_sum(c.cid, buf.ctypes.data, len(parts), out.ctypes.data, inf.ctypes.data) == 0
            exp = cref.msm(c.cid, sc[:n], bases[off:off + n], n) if n else (bytes(64), 1)
            assert (out.tobytes(), int(inf[0])) == exp, (k, off, n)
