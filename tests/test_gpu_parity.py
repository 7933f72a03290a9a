"""Parity tests proper (-m gpu): the HIP path, called through the C ABI (nova_amd.provider -> libnova_mi355x.so),
against the oracle on identical seeded inputs -- bit-exact on the affine bytes `to_coordinates()` returns.

Test matrix = the reference's own:
  src/provider/blitzar.rs:48-214          GPU-backend contract: empty, n = 2, n = 100, batches, ragged batches
  src/provider/curve_property_tests.rs:180-218   n in {16, 100, 8104, 8200} x {random, equal, 0/(r-1)}
  src/provider/msm.rs:722-821             msm == naive; msm_small == msm for 9 bit widths; identity bases
plus BASELINE.json sizes (2^16, 2^20) and size-independent properties at full size (linearity, shard additivity).
"""
import threading

import numpy as np
import pytest

from oracle import cref
from oracle import pyref as R
from tests import util

pytestmark = pytest.mark.gpu
ALL = list(R.CURVES.values())


def as_pair(com):
    return (com.xy, int(com.is_inf))


@pytest.mark.parametrize("c", ALL, ids=lambda c: c.name)
def test_blitzar_contract(nmx, c):
    g = nmx.DlogGroup(c.cid)
    # empty -> identity (blitzar.rs:48-66)
    assert as_pair(g.vartime_multiscalar_mul(np.zeros((0, 32), np.uint8), np.zeros((0, 64), np.uint8))) == (bytes(64), 1)
    assert [as_pair(x) for x in g.batch_vartime_multiscalar_mul([np.zeros((0, 32), np.uint8)], np.zeros((0, 64), np.uint8))] == [(bytes(64), 1)]
    # n = 2 and n = 100 vs naive / msm_best role (blitzar.rs:68-166)
    for n in (2, 100):
        bases = cref.sequential_bases(c, 50 + n, n)
        sc = util.random_scalars(c.cid, n)
        assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == cref.msm(c.cid, sc, bases, n)
    # batch of 20 x 100 (blitzar.rs:117-143) and ragged lengths 0..100 (blitzar.rs:185-213)
    bases = cref.sequential_bases(c, 4242, 100)
    for lens in ([100] * 20, [i * 100 // 19 for i in range(20)]):
        vecs = [util.random_scalars(c.cid, L, seed=100 + j) for j, L in enumerate(lens)]
        got = [as_pair(x) for x in g.batch_vartime_multiscalar_mul(vecs, bases)]
        assert got == cref.msm_batch(c.cid, [v.tobytes() for v in vecs], bases, 100)


@pytest.mark.parametrize("c", ALL, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 16, 17, 100, 8104, 8200])
def test_msm_matrix(nmx, c, n):
    g = nmx.DlogGroup(c.cid)
    bases = cref.sequential_bases(c, 1000 + n, n)
    prep = cref.Prepared(c.cid, bases, n)
    ck = nmx.CommitmentKey.from_host(c.cid, bases)
    for kind in ["random", "equal", "zero_rm1", "pm_small", "u1", "u10", "u16", "u32", "u64"]:
        sc = util.scalar_set(c.cid, n, kind)
        assert as_pair(g.vartime_multiscalar_mul(sc, ck)) == prep.msm(sc, n), (c.name, n, kind)
    ck.close()


@pytest.mark.parametrize("c", ALL, ids=lambda c: c.name)
def test_identity_bases_and_duplicates(nmx, c):
    """msm.rs:786-821 + the P == Q / P == -Q bucket branches (msm.rs:106-113,148-155)."""
    g = nmx.DlogGroup(c.cid)
    n = 300
    bases = cref.sequential_bases(c, 5, n).copy()
    for i in (0, 3, n - 1):
        bases[i] = 0
    sc = util.random_scalars(c.cid, n).copy()
    sc[0] = util.int_to_le32(1)
    assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == cref.msm(c.cid, sc, bases, n)
    dup = np.repeat(bases[1:2], 64, axis=0)
    sc = util.scalar_set(c.cid, 64, "equal")
    assert as_pair(g.vartime_multiscalar_mul(sc, dup)) == cref.msm(c.cid, sc, dup, 64)
    sc = util.scalar_set(c.cid, 64, "zero_rm1").copy()
    sc[0::2] = util.int_to_le32(1)
    assert as_pair(g.vartime_multiscalar_mul(sc, dup)) == (bytes(64), 1)


@pytest.mark.parametrize("n", [65, 6001, 100001])
def test_cancelling_pairs_on_repeated_bases(nmx, n):
    """Every bucket branch of msm.rs:106-113,148-155 at once, in the quad accumulate (small n), the table path and
    the plain accumulate: three distinct points repeated, consecutive pairs (k, P), (r - k, P) cancel (P == -Q), equal
    scalars on the same point double (P == Q); one unpaired term keeps the result off the identity."""
    c = R.BN254_G1
    g = nmx.DlogGroup(c.cid)
    pts = cref.sequential_bases(c, 9, 3)
    idx = (np.arange(n) // 2) % 3
    bases = np.ascontiguousarray(pts[idx])
    k = util.random_scalars(c.cid, n, seed=5).copy()
    ki = [int.from_bytes(bytes(row), "little") for row in k]
    for i in range(1, n - 1, 2):
        k[i] = util.int_to_le32((c.r - ki[i - 1]) % c.r)
    exp = cref.msm(c.cid, k, bases, n)
    assert exp[1] == 0
    assert as_pair(g.vartime_multiscalar_mul(k, bases)) == exp
    ck = nmx.CommitmentKey.from_host(c.cid, bases)            # registered: window tables when n >= 4096
    assert as_pair(g.vartime_multiscalar_mul(k, ck)) == exp
    k[n - 1] = util.int_to_le32(0)                             # now everything cancels
    assert as_pair(g.vartime_multiscalar_mul(k, ck)) == (bytes(64), 1)
    same = np.repeat(k[:1], n, axis=0)                         # one scalar on three repeated points: doublings
    assert as_pair(g.vartime_multiscalar_mul(same, ck)) == cref.msm(c.cid, same, bases, n)
    ck.close()


@pytest.mark.parametrize("bits", [0, 1, 4, 8, 10, 16, 20, 32, 40, 64])
def test_msm_small(nmx, bits):
    """msm.rs:751-784 (test_msm_ux) through vartime_multiscalar_mul_small[_with_max_num_bits]."""
    for c in (R.BN254_G1, R.GRUMPKIN, R.PALLAS, R.VESTA):
        g = nmx.DlogGroup(c.cid)
        for n in (8, 3000):
            bases = cref.sequential_bases(c, 31, n)
            s = util.small_scalars(n, bits) if bits else np.zeros(n, np.uint64)
            exp = cref.msm_u64(c.cid, s, bases, n, bits)
            assert as_pair(g.vartime_multiscalar_mul_small_with_max_num_bits(s, bases, bits)) == exp
            assert as_pair(g.vartime_multiscalar_mul_small(s, bases)) == exp
            assert as_pair(g.vartime_multiscalar_mul(util.u64_to_le32(s), bases)) == exp


def test_batch_msm_small(nmx):
    """DlogGroupExt::batch_vartime_multiscalar_mul_small (src/provider/traits.rs:109-117; batch_commit_small,
    src/traits/commitment.rs:139-150): ragged vectors of u64 scalars over prefixes of one base array -- slice form and
    registered key, explicit and automatic bit widths, an empty vector, the max_num_bits = 0 rule (msm.rs:489), a value beyond
    the declared width in one vector (the whole call fails)."""
    from nova_amd import _lib
    for c in (R.BN254_G1, R.VESTA):
        g, ce = nmx.DlogGroup(c.cid), nmx.CommitmentEngine(c.cid)
        n = 5000
        bases = cref.sequential_bases(c, 77, n)
        ck = nmx.CommitmentKey.from_host(c.cid, bases)
        for bits in (None, 1, 10, 33, 64):
            lens = [n, 2500, 17, 1, 0, 4097]
            vs = [util.small_scalars(m, bits or 23, seed=m + 3) if m else np.zeros(0, np.uint64) for m in lens]
            exp = [cref.msm_u64(c.cid, v, bases[:len(v)], len(v), bits or 23) if len(v) else (bytes(64), 1) for v in vs]
            assert [as_pair(x) for x in g.batch_vartime_multiscalar_mul_small(vs, ck, bits)] == exp
            assert [as_pair(x) for x in g.batch_vartime_multiscalar_mul_small(vs, bases, bits)] == exp
        assert [as_pair(x) for x in ce.batch_commit_small(ck, vs)] == exp
        assert [as_pair(x) for x in g.batch_vartime_multiscalar_mul_small(vs, ck, 0)] == [(bytes(64), 1)] * len(vs)
        bad = [v.copy() for v in vs]
        bad[1][7] = 1 << 12
        with pytest.raises(nmx.NmxError) as e:
            g.batch_vartime_multiscalar_mul_small(bad, ck, 10)
        assert e.value.code == _lib.E_SMALL_RANGE
        ck.close()


def test_error_paths(nmx):
    from nova_amd import _lib
    c = R.BN254_G1
    g = nmx.DlogGroup(c.cid)
    n = 64
    bases = cref.sequential_bases(c, 9, n)
    sc = util.random_scalars(c.cid, n).copy()
    sc[3] = util.int_to_le32(c.r)
    with pytest.raises(nmx.NmxError) as e:
        g.vartime_multiscalar_mul(sc, bases)
    assert e.value.code == _lib.E_SCALAR_RANGE
    s = util.small_scalars(n, 10).copy()
    s[5] = 1 << 10
    with pytest.raises(nmx.NmxError) as e:
        g.vartime_multiscalar_mul_small_with_max_num_bits(s, bases, 10)
    assert e.value.code == _lib.E_SMALL_RANGE
    ck = nmx.CommitmentKey.from_host(c.cid, bases)
    with pytest.raises(AssertionError):  # assert!(ck.ck.len() >= v.len()), pedersen.rs:264
        nmx.CommitmentEngine(c.cid).commit(ck, util.random_scalars(c.cid, n + 1))
    ck.close()
    with pytest.raises(nmx.NmxError) as e:
        g.vartime_multiscalar_mul(sc, ck)
    assert e.value.code == _lib.E_HANDLE


@pytest.mark.parametrize("c", ALL, ids=lambda c: c.name)
def test_generated_key_and_montgomery_layout(nmx, c):
    """nmx_bases_generate == the oracle's (k0+i)*G; raw-Montgomery inputs == canonical inputs."""
    n = 1500
    ck = nmx.CommitmentKey.generate(c.cid, n, k0=7)
    host = cref.sequential_bases(c, 7, n + 1)
    assert ck.read(0, n).tobytes() == host[:n].tobytes()
    assert ck.h == host[n].tobytes()
    sc = util.random_scalars(c.cid, n)
    g = nmx.DlogGroup(c.cid)
    exp = cref.msm(c.cid, sc, host[:n], n)
    assert as_pair(g.vartime_multiscalar_mul(sc, ck)) == exp
    # Montgomery-form scalars and bases (the in-memory layout of halo2curves): x*R mod p as LE limbs
    Rm = 1 << 256
    m = 200
    sc_m = np.frombuffer(b"".join(R.fe_to_le32(int.from_bytes(bytes(row), "little") * Rm % c.r) for row in sc[:m]), np.uint8)
    b_m = np.frombuffer(b"".join(R.fe_to_le32(int.from_bytes(bytes(host[i, j:j + 32]), "little") * Rm % c.p)
                                 for i in range(m) for j in (0, 32)), np.uint8)
    assert as_pair(g.vartime_multiscalar_mul(sc_m, b_m, mont=True)) == cref.msm(c.cid, sc[:m], host[:m], m)
    ck.close()


@pytest.mark.parametrize("c", [R.BN254_G1, R.GRUMPKIN, R.PALLAS], ids=lambda c: c.name)
def test_commit_engine(nmx, c):
    """CommitmentEngineTrait::commit / batch_commit / commit_small (pedersen.rs:263-283, hyperkzg.rs:584-612)."""
    ce = nmx.CommitmentEngine(c.cid)
    n = 2000
    host = cref.sequential_bases(c, 3, n + 1)
    ck = nmx.CommitmentKey.from_host(c.cid, host[:n], host[n].tobytes())
    v = util.random_scalars(c.cid, n - 7)
    r = util.random_scalars(c.cid, 1, seed=9)
    assert as_pair(ce.commit(ck, v, r)) == cref.commit(c.cid, v, host[:n - 7], n - 7, host[n], r)
    assert as_pair(ce.commit(ck, v)) == cref.msm(c.cid, v, host[:n - 7], n - 7)
    # HyperKZG prove shape: polynomials of length n/2, n/4, ..., 2 over one key (hyperkzg.rs:1085-1100)
    lens = [1024 >> i for i in range(10)]
    vs = [util.random_scalars(c.cid, L, seed=20 + i) for i, L in enumerate(lens)]
    got = [as_pair(x) for x in ce.batch_commit(ck, vs)]
    assert got == cref.msm_batch(c.cid, [x.tobytes() for x in vs], host[:n], n)
    s = util.small_scalars(n, 16)
    exp_small = cref.msm_u64(c.cid, s, host[:n], n)
    assert as_pair(ce.commit_small(ck, s)) == exp_small
    # commit_small with blinding: msm_small + h*r == commit(field(s), r)
    assert as_pair(ce.commit_small(ck, s, r)) == cref.commit(c.cid, util.u64_to_le32(s), host[:n], n, host[n], r)
    ck.close()


@pytest.mark.parametrize("c", [R.BN254_G1, R.VESTA], ids=lambda c: c.name)
@pytest.mark.parametrize("precompute", [True, False])
def test_sparse_commitments(nmx, c, precompute):
    """commit_sparse_binary / commit_sparse / commit_small_range / batch_commit_small
    (src/provider/pedersen.rs:285-305,395-427) against the oracle on gathered bases."""
    ce = nmx.CommitmentEngine(c.cid)
    n_key = 6000
    host = cref.sequential_bases(c, 41, n_key + 1).copy()
    host[17] = 0
    ck = nmx.CommitmentKey.from_host(c.cid, host[:n_key], host[n_key].tobytes(), precompute=precompute)
    rng = np.random.Generator(np.random.PCG64(5))
    r = util.random_scalars(c.cid, 1, seed=9)
    for k in (0, 1, 300, 5000):
        idx = rng.integers(0, n_key, size=k).astype(np.uint64)  # with repeats: a base may appear several times
        if k >= 300:
            idx[3] = 17  # the identity point
        gathered = host[idx.astype(np.int64)] if k else np.zeros((0, 64), np.uint8)
        ones = util.u64_to_le32(np.ones(k, np.uint64))
        assert as_pair(ce.commit_sparse_binary(ck, idx)) == cref.msm(c.cid, ones, gathered, k)
        assert as_pair(ce.commit_sparse_binary(ck, idx, r)) == cref.commit(c.cid, ones, gathered, k, host[n_key], r)
        sc = util.scalar_set(c.cid, k, "pm_small") if k else np.zeros((0, 32), np.uint8)
        assert as_pair(ce.commit_sparse(ck, idx, sc)) == cref.msm(c.cid, sc, gathered, k)
        assert as_pair(ce.commit_sparse(ck, idx, sc, r)) == cref.commit(c.cid, sc, gathered, k, host[n_key], r)
    from nova_amd import _lib
    with pytest.raises(nmx.NmxError) as e:
        ce.commit_sparse_binary(ck, np.array([n_key], np.uint64))
    assert e.value.code == _lib.E_HANDLE
    v = util.small_scalars(n_key, 12)
    exp = cref.commit(c.cid, util.u64_to_le32(v[100:5100]), host[100:5100], 5000, host[n_key], r)
    assert as_pair(ce.commit_small_range(ck, v, r, 100, 5100, 12)) == exp
    vs = [util.small_scalars(L, 9, seed=L) for L in (10, 4500)]
    got = [as_pair(x) for x in ce.batch_commit_small(ck, vs, [r, None])]
    assert got[0] == cref.commit(c.cid, util.u64_to_le32(vs[0]), host[:10], 10, host[n_key], r)
    assert got[1] == cref.msm(c.cid, util.u64_to_le32(vs[1]), host[:4500], 4500)
    ck.close()


@pytest.mark.parametrize("c", ALL, ids=lambda c: c.name)
def test_precomputed_tables_vs_plain(nmx, c):
    """A key registered with window tables (NMX_BASES_PRECOMPUTE) and the same key without them give the oracle's
    result on every scalar set, for prefixes and interior slices, field and small scalars."""
    import ctypes
    from nova_amd import _lib
    L = _lib.lib()
    n_key = 9000
    host = cref.sequential_bases(c, 77, n_key).copy()
    host[4500] = 0  # identity point inside the key
    g = nmx.DlogGroup(c.cid)
    prep = cref.Prepared(c.cid, host, n_key)
    keys = [nmx.CommitmentKey.from_host(c.cid, host, precompute=p) for p in (True, False)]
    for kind in ["random", "equal", "zero_rm1", "pm_small", "u1", "u16"]:
        sc = util.scalar_set(c.cid, n_key, kind)
        exp = prep.msm(sc, n_key)
        for ck in keys:
            assert as_pair(g.vartime_multiscalar_mul(sc, ck)) == exp, kind
    out = np.zeros(64, np.uint8)
    inf = np.zeros(1, np.uint8)
    for off, n in ((0, 5000), (1234, 7000), (4000, 4096), (8000, 1000)):
        sc = np.ascontiguousarray(util.random_scalars(c.cid, n, seed=off))
        exp = cref.msm(c.cid, sc, host[off:off + n], n)
        for ck in keys:
            assert L.nmx_msm_handle(ck.handle, off, sc.ctypes.data, n, 0, out.ctypes.data, inf.ctypes.data) == 0
            assert (out.tobytes(), int(inf[0])) == exp, (off, n)
    for bits in (1, 10, 33, 64):
        s = util.small_scalars(n_key, bits)
        exp = prep.msm_u64(s, n_key, bits)
        for ck in keys:
            assert as_pair(g.vartime_multiscalar_mul_small_with_max_num_bits(s, ck, bits)) == exp
    [ck.close() for ck in keys]


@pytest.mark.parametrize("logn,precompute", [(16, False), (16, True), (20, True), (20, False)])
def test_bn254_baseline_sizes(nmx, logn, precompute):
    """BASELINE.json configs: BN254 MSM at 2^16 and 2^20, random scalars, bit-exact vs the oracle; plus
    size-independent properties at full size: shard additivity and linearity."""
    c = R.BN254_G1
    n = 1 << logn
    g = nmx.DlogGroup(c.cid)
    ck = nmx.CommitmentKey.generate(c.cid, n, k0=1, precompute=precompute)
    host = ck.read(0, n)
    s = util.random_scalars(c.cid, n, seed=1)
    t = util.random_scalars(c.cid, n, seed=2)
    prep = cref.Prepared(c.cid, host, n)
    full = g.vartime_multiscalar_mul(s, ck)
    assert as_pair(full) == prep.msm(s, n)
    # 8-way contiguous shards -> 128-byte partials -> nmx_point_sum == full result
    from nova_amd import _lib
    import ctypes
    L = _lib.lib()
    parts = np.zeros((8, 128), np.uint8)
    inf = np.zeros(1, np.uint8)
    for k in range(8):
        lo, hi = k * n // 8, (k + 1) * n // 8
        sk = np.ascontiguousarray(s[lo:hi])
        rc = L.nmx_msm_handle(ck.handle, lo, sk.ctypes.data, hi - lo, _lib.OUT_PARTIAL, parts[k].ctypes.data, inf.ctypes.data)
        assert rc == 0, L.nmx_last_error()
    assert as_pair(g.point_sum(parts)) == as_pair(full)
    # linearity: msm(s) + msm(t) == msm(s + t mod r)
    pt = g.vartime_multiscalar_mul(t, ck, partial=True)
    ps = g.vartime_multiscalar_mul(s, ck, partial=True)
    si = s.view(np.uint64).reshape(n, 4)
    ti = t.view(np.uint64).reshape(n, 4)
    st = np.zeros((n, 4), np.uint64)
    carry = np.zeros(n, np.uint64)
    for j in range(4):
        a = si[:, j] + ti[:, j]
        c1 = a < si[:, j]
        b = a + carry
        c2 = b < a
        st[:, j] = b
        carry = (c1 | c2).astype(np.uint64)
    m = util._limbs(c.r)
    ge = ~util._lt(st, m)
    borrow = np.zeros(n, np.uint64)
    for j in range(4):
        sub = np.where(ge, m[j], np.uint64(0))
        d = st[:, j] - sub
        b1 = st[:, j] < sub
        d2 = d - borrow
        b2 = d < borrow
        st[:, j] = d2
        borrow = (b1 | b2).astype(np.uint64)
    sum_sc = st.view(np.uint8).reshape(n, 32)
    assert as_pair(g.point_sum([ps.xy, pt.xy])) == as_pair(g.vartime_multiscalar_mul(sum_sc, ck))
    ck.close()


def test_device_resident_scalars_and_threads(nmx):
    """HBM-resident scalars (torch CUDA tensor) and concurrent callers (the trait fns are static and are called
    from several rayon threads at once: r1cs/mod.rs:509-512, hyperkzg.rs:1062-1065)."""
    import torch
    c = R.BN254_G1
    n = 20000
    g = nmx.DlogGroup(c.cid)
    ck = nmx.CommitmentKey.generate(c.cid, n, k0=3)
    host = ck.read(0, n)
    prep = cref.Prepared(c.cid, host, n)
    sets = [util.random_scalars(c.cid, n, seed=50 + i) for i in range(6)]
    exp = [prep.msm(s, n) for s in sets]
    d = torch.from_numpy(sets[0].copy()).cuda()
    assert as_pair(g.vartime_multiscalar_mul(d, ck)) == exp[0]
    s64 = util.small_scalars(n, 32)
    d64 = torch.from_numpy(s64.view(np.int64).copy()).cuda()
    assert as_pair(g.vartime_multiscalar_mul_small(d64, ck)) == prep.msm_u64(s64, n)
    got = [None] * 6

    def work(i):
        for _ in range(3):
            got[i] = as_pair(g.vartime_multiscalar_mul(sets[i], ck))

    th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert got == exp
    ck.close()
