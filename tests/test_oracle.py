"""Pins the oracle (CPU, no GPU).

Tier 1 (oracle/pyref.py, big-int definition) is pinned against: the reference's modulus / order strings, the
group-order identity r*G == O, and public known-answer points.  Tier 2 (oracle/nova_ref.c, the C restatement of
/root/reference/src/provider/msm.rs) is then pinned against tier 1 on the reference's own test matrix:
  msm.rs:722-821 (n = 8: msm == naive; msm_small == msm for 9 bit-widths; identity bases),
  curve_property_tests.rs:180-218 (n in {16, 100, 8104, 8200} x {random, equal, 0/(r-1)}),
  blitzar.rs:48-214 (empty, n = 2, n = 100, ragged batches).
The reference stores no MSM output vectors; these are the identities its tests evaluate.
"""
import numpy as np
import pytest

from oracle import cref
from oracle import pyref as R
from tests import util

ALL = list(R.CURVES.values())


def ints(sc):
    return [int.from_bytes(bytes(row), "little") for row in sc]


def pts(b):
    return [R.xy64_to_point(bytes(row)) for row in b]


def test_reference_constant_strings():
    # hex strings exactly as they appear in the reference sources
    assert R.BN254_R == int("30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001", 16)  # bn256_grumpkin.rs:39
    assert R.BN254_Q == int("30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47", 16)  # bn256_grumpkin.rs:40
    assert R.PALLAS_Q == int("40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001", 16)  # pasta.rs:37
    assert R.PALLAS_P == int("40000000000000000000000000000000224698fc094cf91b992d30ed00000001", 16)  # pasta.rs:38
    assert R.GRUMPKIN.p == R.BN254_G1.r and R.GRUMPKIN.r == R.BN254_G1.p  # bn256_grumpkin.rs:84-85 (cycle)
    assert R.VESTA.p == R.PALLAS.r and R.VESTA.r == R.PALLAS.p            # pasta.rs:45-46
    for c in ALL:
        assert util.MODULI[c.cid] == c.r


@pytest.mark.parametrize("c", ALL, ids=lambda c: c.name)
def test_pyref_group_order_and_law(c):
    G = (c.gx, c.gy)
    assert R.on_curve(c, G)
    assert R.mul(c, c.r, G) is R.INF                      # order * G == identity
    assert R.mul(c, c.r - 1, G) == R.neg(c, G)
    P, Q = R.mul(c, 5, G), R.mul(c, 7, G)
    assert R.add(c, P, Q) == R.mul(c, 12, G)              # group_law: curve_property_tests.rs:93-116
    assert R.add(c, P, R.neg(c, P)) is R.INF
    assert R.add(c, P, P) == R.mul(c, 10, G)
    assert R.add(c, P, R.INF) == P


def test_pyref_bn254_known_answers():
    # public BN254 (alt_bn128) G1 vectors, e.g. the EIP-196 ecadd/ecmul tests: 2G and 3G
    c = R.BN254_G1
    G = (1, 2)
    two_g = (0x030644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD3,
             0x15ED738C0E0A7C92E7845F96B2AE9C0A68A6A449E3538FC7FF3EBF7A5A18A2C4)
    three_g = (0x0769BF9AC56BEA3FF40232BCB1B6BD159315D84715B8E679F2D355961915ABF0,
               0x2AB799BEE0489429554FDB7C8D086475319E63B40B9C5B57CDF1FF3DD9FE2261)
    assert R.add(c, G, G) == two_g
    assert R.mul(c, 3, G) == three_g


@pytest.mark.parametrize("c", ALL, ids=lambda c: c.name)
def test_cref_sequential_bases(c):
    b = cref.sequential_bases(c, 12345, 19)
    assert pts(b) == R.sequential_bases(c, 12345, 19)
    assert all(R.on_curve(c, P) for P in pts(b))


@pytest.mark.parametrize("c", ALL, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [0, 1, 2, 8, 16, 17, 100])
def test_cref_msm_matches_definition(c, n):
    """msm.rs:722-739 / blitzar.rs:48-115 / curve_property_tests.rs:180-218 at small n, every scalar set."""
    bases = cref.sequential_bases(c, 777 + n, n)
    for kind in ["random", "equal", "zero_rm1", "pm_small"]:
        sc = util.scalar_set(c.cid, n, kind)
        got, inf = cref.msm(c.cid, sc, bases, n)
        exp = R.msm_naive(c, ints(sc), pts(bases))
        assert got == R.point_to_xy64(exp), (c.name, n, kind)
        assert inf == (exp is R.INF)


@pytest.mark.parametrize("n", [8104, 8200])
def test_cref_msm_straddles_msm_best_threshold(n):
    """curve_property_tests.rs:168-170: n around msm_best's internal strategy switch; BN254, random scalars."""
    c = R.BN254_G1
    bases = cref.sequential_bases(c, 99, n)
    sc = util.scalar_set(c.cid, n, "random")
    got, _ = cref.msm(c.cid, sc, bases, n)
    assert got == R.point_to_xy64(R.msm_naive(c, ints(sc), pts(bases)))
    # the other two scalar sets of the reference test are checked C-vs-C (msm() vs the msm_best role)
    prep = cref.Prepared(c.cid, bases, n)
    for kind in ["equal", "zero_rm1"]:
        sc = util.scalar_set(c.cid, n, kind)
        assert prep.msm(sc, n) == prep.msm(sc, n, best_only=True)


@pytest.mark.parametrize("c", ALL, ids=lambda c: c.name)
def test_cref_identity_bases(c):
    """msm.rs:786-821: identity bases with non-zero scalars contribute nothing."""
    for n in (8, 40):
        bases = cref.sequential_bases(c, 5, n).copy()
        sc = util.scalar_set(c.cid, n, "random").copy()
        for i in (0, 3, n - 1):
            bases[i] = 0
        sc[0] = util.int_to_le32(1)
        got, _ = cref.msm(c.cid, sc, bases, n)
        assert got == R.point_to_xy64(R.msm_naive(c, ints(sc), pts(bases)))


@pytest.mark.parametrize("c", ALL, ids=lambda c: c.name)
@pytest.mark.parametrize("bits", [1, 4, 8, 10, 16, 20, 32, 40, 64])
def test_cref_msm_small_matches_msm(c, bits):
    """msm.rs:751-784 (test_msm_ux): msm_small(u64) == msm(field) for every bit-width, n = 8 and n = 100."""
    for n in (8, 100):
        bases = cref.sequential_bases(c, 31, n)
        s = util.small_scalars(n, bits)
        s[0] = (1 << bits) - 1  # make num_bits(max) == bits
        small, _ = cref.msm_u64(c.cid, s, bases, n)                      # msm_small: bits from the maximum
        small_b, _ = cref.msm_u64(c.cid, s, bases, n, max_num_bits=bits)  # ..._with_max_num_bits
        general, _ = cref.msm(c.cid, util.u64_to_le32(s), bases, n)
        exp = R.point_to_xy64(R.msm_naive(c, [int(x) for x in s], pts(bases)))
        assert small == small_b == general == exp


def test_cref_batch_ragged():
    """blitzar.rs:185-213: 20 vectors of lengths 0..100 over one base array, each == msm(bases[..len])."""
    c = R.BN254_G1
    lens = [i * 100 // 19 for i in range(20)]
    bases = cref.sequential_bases(c, 4242, 100)
    vecs = [util.random_scalars(c.cid, L, seed=100 + j).tobytes() for j, L in enumerate(lens)]
    res = cref.msm_batch(c.cid, vecs, bases, 100)
    for j, L in enumerate(lens):
        sc = np.frombuffer(vecs[j], dtype=np.uint8).reshape(L, 32)
        assert res[j][0] == R.point_to_xy64(R.msm_naive(c, ints(sc), pts(bases[:L])))
    assert res[0] == (bytes(64), 1)  # empty vector -> identity (blitzar.rs:48-66)


@pytest.mark.parametrize("c", [R.BN254_G1, R.PALLAS], ids=lambda c: c.name)
def test_cref_commit(c):
    """pedersen.rs:263-270: commit = msm(v, ck[..n]) + h*r."""
    n = 33
    ck = cref.sequential_bases(c, 10, 64)
    h = cref.sequential_bases(c, 999, 1)
    v = util.random_scalars(c.cid, n)
    r = util.random_scalars(c.cid, 1, seed=7)
    got, _ = cref.commit(c.cid, v, ck[:n], n, h, r)
    exp = R.commit(c, pts(ck), pts(h)[0], ints(v), ints(r)[0])
    assert got == R.point_to_xy64(exp)


def test_cref_field_axpy():
    """r1cs/mod.rs:1058-1067: W = W1 + r*W2."""
    for fid, p in enumerate([R.BN254_Q, R.BN254_R, R.PALLAS_P, R.PALLAS_Q]):
        cid = {0: 1, 1: 0, 2: 3, 3: 2}[fid]  # curve whose scalar field is this field
        a = util.random_scalars(cid, 50, seed=1)
        b = util.random_scalars(cid, 50, seed=2)
        r = util.random_scalars(cid, 1, seed=3)
        got = cref.field_axpy(fid, a, b, r, 50)
        exp = R.axpy(p, ints(a), ints(b), ints(r)[0])
        assert got == b"".join(R.fe_to_le32(x) for x in exp)


def test_public_eip196_vectors_pin_both_oracle_tiers():
    """tests/golden/public_kats.json (EIP-196 ecMul vectors, not produced by this repository): tier 1 (big-int) and
    tier 2 (C restatement) must reproduce every published product, and an MSM over all of them must equal the sum of
    the published outputs."""
    from tests import kats
    c = R.BN254_G1
    cases = kats.load()
    assert len(cases) >= 5
    bases, sc = kats.as_arrays(cases)
    total = R.INF
    for i, (name, P, k, Q) in enumerate(cases):
        assert R.on_curve(c, P) and R.on_curve(c, Q), name
        assert R.mul(c, k, P) == Q, name
        assert cref.msm(c.cid, sc[i:i + 1], bases[i:i + 1], 1) == (R.point_to_xy64(Q), 0), name
        total = R.add(c, total, Q)
    assert cref.msm(c.cid, sc, bases, len(cases)) == (R.point_to_xy64(total), 0)


def test_sympy_vectors_pin_both_oracle_tiers_on_all_four_curves():
    """tests/golden/sympy_kats.json: scalar multiples of the generators (2, 3, 5, 9, 2^128 + 12345, r - 1, r - 2, four
    full-width scalars) and a six-term MSM (one zero scalar, one r - 1) on BN254 G1, Grumpkin, Pallas and Vesta, computed by
    SymPy's elliptic-curve arithmetic -- third-party code: the only external anchor there is for the three curves that have no
    published vector we could restate offline.  Its BN254 entries agree with the EIP-196 vectors (2 G, 9 G).
    Both oracle tiers must reproduce every entry; the constants of the file must be the oracle's."""
    from tests import kats
    data = kats.load_sympy()
    assert set(data) == set(R.CURVES)
    for name, d in data.items():
        c = R.CURVES[name]
        assert (d["p"], d["r"], d["gen"]) == (c.p, c.r, (c.gx, c.gy)), name
        for k, Q in d["mul"]:
            assert R.on_curve(c, Q), (name, k)
            assert R.mul(c, k, (c.gx, c.gy)) == Q, (name, k)
            bases, sc = kats.points_scalars([(c.gx, c.gy)], [k])
            assert cref.msm(c.cid, sc, bases, 1) == (R.point_to_xy64(Q), 0), (name, k)
        P, s, total = d["msm"]
        acc = R.INF
        for Pi, si in zip(P, s):
            acc = R.add(c, acc, R.mul(c, si, Pi))
        assert acc == total, name
        bases, sc = kats.points_scalars(P, s)
        assert cref.msm(c.cid, sc, bases, len(P)) == (R.point_to_xy64(total), 0), name
    # cross-check of the two external sources with each other
    eip = {k: Q for (_n, Pt, k, Q) in kats.load() if Pt == (1, 2)}
    sym = dict(data["bn254_g1"]["mul"])
    assert eip[2] == sym[2] and eip[9] == sym[9]
