"""-m gpu: keys sharded over several devices of ONE process behind the C ABI (nmx_init_devices; the
reference's in-process decomposition, /root/reference/src/provider/msm.rs:564-574,664-676).  The GPU box has one MI355X,
so the logical devices are oversubscribed onto it (NMX_DEVICES_OVERSUBSCRIBE): every shard has its own tables, stream,
workspace and host thread exactly as on k GPUs -- only the physical placement differs.  Everything is compared with the
oracle; k = 1 must be the unsharded path."""
import ctypes
import os
import tempfile

import numpy as np
import pytest

from oracle import cref, keyfiles
from oracle import pyref as R
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture()
def sharded(nmx):
    """k logical devices, every key from 1000 points up sharded; restored afterwards."""
    from nova_amd import _lib
    L = _lib.lib()

    def enter(k):
        assert nmx.init_devices(k, oversubscribe=True) == k
        assert L.nmx_set_option(b"shard_min_n", 1000) == 0
        return L
    yield enter
    assert nmx.init_devices(1) == 1
    assert L.nmx_set_option(b"shard_min_n", 1 << 20) == 0


def pt(c):
    return (c.xy, int(c.is_inf))


@pytest.mark.parametrize("k", [2, 3])
@pytest.mark.parametrize("c", [R.BN254_G1, R.PALLAS], ids=lambda c: c.name)
def test_sharded_key_every_entry_point(nmx, sharded, c, k):
    from nova_amd import _lib
    L = sharded(k)
    n = 6000
    bases = cref.sequential_bases(c, 900 + k, n + 1).copy()
    bases[n // 3] = 0                                    # an identity point inside one shard
    before = _lib.stats()[_lib.STAT_SHARDED_CALLS]
    ck = nmx.CommitmentKey.from_host(c.cid, bases[:n], bases[n].tobytes())
    g, ce = nmx.DlogGroup(c.cid), nmx.CommitmentEngine(c.cid)
    assert np.array_equal(ck.read(0, n), bases[:n]) and np.array_equal(ck.read(n // k - 3, 7), bases[n // k - 3:n // k + 4])
    for kind in ("random", "zero_rm1", "equal"):
        sc = util.scalar_set(c.cid, n, kind)
        for off, m in ((0, n), (0, n // k), (n // k - 5, 11), (17, n - 40), (n - 1, 1), (100, 0)):
            got = g.vartime_multiscalar_mul(sc[:m], ck, offset=off)
            exp = cref.msm(c.cid, sc[:m], bases[off:off + m], m) if m else (bytes(64), 1)
            assert pt(got) == exp, (kind, off, m)
    assert _lib.stats()[_lib.STAT_SHARDED_CALLS] > before
    sc = util.random_scalars(c.cid, n)
    # commit with blinding, partial output, small scalars, sparse forms, batch, HBM-resident scalars
    r = util.random_scalars(c.cid, 1, seed=9)
    assert pt(ce.commit(ck, sc, r)) == cref.commit(c.cid, sc, bases[:n], n, bases[n], r)
    part = g.vartime_multiscalar_mul(sc, ck, partial=True)
    assert pt(g.point_sum([part.xy])) == cref.msm(c.cid, sc, bases[:n], n)
    s64 = util.small_scalars(n, 33)
    assert pt(g.vartime_multiscalar_mul_small(s64, ck)) == cref.msm_u64(c.cid, s64, bases[:n], n, 33)
    idx = np.array([0, 1, n // k - 1, n // k, n // 2, n - 1, 5, 5], dtype=np.uint64)
    ssc = util.random_scalars(c.cid, len(idx), seed=4)
    gathered = bases[idx.astype(np.int64)]
    assert pt(ce.commit_sparse(ck, idx, ssc)) == cref.msm(c.cid, ssc, gathered, len(idx))
    ones = np.zeros((len(idx), 32), np.uint8)
    ones[:, 0] = 1
    assert pt(ce.commit_sparse_binary(ck, idx)) == cref.msm(c.cid, ones, gathered, len(idx))
    lens = [n, n // 2, 333, 2, 0]
    got = [pt(x) for x in g.batch_vartime_multiscalar_mul([sc[:m] for m in lens], ck)]
    assert got == [cref.msm(c.cid, sc[:m], bases[:m], m) if m else (bytes(64), 1) for m in lens]
    import torch
    d = torch.from_numpy(sc.copy()).cuda()
    assert pt(g.vartime_multiscalar_mul(d, ck)) == cref.msm(c.cid, sc, bases[:n], n)
    bad = sc.copy()
    bad[n - 2] = 0xFF                                    # >= r in the last shard: the whole call fails, nothing written
    with pytest.raises(nmx.NmxError) as e:
        g.vartime_multiscalar_mul(bad, ck)
    assert e.value.code == _lib.E_SCALAR_RANGE
    ck.close()


def test_sharded_slice_form_generated_and_file_keys(nmx, sharded):
    from nova_amd import _lib
    L = sharded(3)
    c = R.BN254_G1
    n = 4097
    g = nmx.DlogGroup(c.cid)
    # the trait's slice form: the slice cache makes the array resident -- sharded -- on first sight
    bases = cref.sequential_bases(c, 31337, n).copy()
    sc = util.random_scalars(c.cid, n)
    L.nmx_cache_clear()
    up0, sh0 = _lib.stats()[_lib.STAT_CACHE_UPLOADS], _lib.stats()[_lib.STAT_SHARDED_CALLS]
    for m in (n, 2000, n):
        assert pt(g.vartime_multiscalar_mul(sc[:m], bases[:m])) == cref.msm(c.cid, sc[:m], bases[:m], m)
    assert _lib.stats()[_lib.STAT_CACHE_UPLOADS] == up0 + 1 and _lib.stats()[_lib.STAT_SHARDED_CALLS] == sh0 + 3
    L.nmx_cache_clear()
    # P_i = (k0 + i) G generated shard by shard
    key = nmx.CommitmentKey.generate(c.cid, n, k0=12345)
    assert np.array_equal(key.read(0, n + 1), cref.sequential_bases(c, 12345, n + 1))
    assert pt(g.vartime_multiscalar_mul(sc, key)) == cref.msm(c.cid, sc, key.read(0, n), n)
    key.close()
    # a Pedersen key file streamed into three shards (pedersen.rs:318-340)
    m = 1024
    pts = R.sequential_bases(c, 21, m + 1)
    xy = np.frombuffer(b"".join(R.point_to_xy64(P) for P in pts[1:]), dtype=np.uint8).reshape(m, 64)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ck.key")
        with open(path, "wb") as f:
            f.write(keyfiles.write_pedersen_key(c, pts[0], pts[1:]))
        fk = nmx.CommitmentKey.load_keyfile(c.cid, path, m)
        assert fk.h == R.point_to_xy64(pts[0]) and np.array_equal(fk.read(0, m), xy)
        assert pt(g.vartime_multiscalar_mul(sc[:m], fk)) == cref.msm(c.cid, sc[:m], xy, m)
        fk.close()


def test_one_device_is_the_unsharded_path(nmx):
    from nova_amd import _lib
    L = _lib.lib()
    assert nmx.init_devices(1) == 1 and L.nmx_devices_in_use() == 1
    assert L.nmx_set_option(b"shard_min_n", 1000) == 0
    try:
        c = R.GRUMPKIN
        n = 5000
        bases = cref.sequential_bases(c, 5, n)
        sc = util.random_scalars(c.cid, n)
        before = _lib.stats()[_lib.STAT_SHARDED_CALLS]
        ck = nmx.CommitmentKey.from_host(c.cid, bases)
        assert pt(nmx.DlogGroup(c.cid).vartime_multiscalar_mul(sc, ck)) == cref.msm(c.cid, sc, bases, n)
        assert _lib.stats()[_lib.STAT_SHARDED_CALLS] == before
        ck.close()
        # more devices than the box has, without oversubscription: a loud error, state unchanged
        cnt = L.nmx_device_count()
        assert L.nmx_init_devices(cnt + 1, 0) == _lib.E_NO_DEVICE and L.nmx_devices_in_use() == 1
    finally:
        assert L.nmx_set_option(b"shard_min_n", 1 << 20) == 0
