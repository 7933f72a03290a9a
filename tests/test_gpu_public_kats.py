"""The HIP path against vectors that no code in this repository produced (-m gpu): the EIP-196 BN254 ecMul
known-answer vectors in tests/golden/public_kats.json, through every MSM entry point of the C ABI."""
import numpy as np
import pytest

from oracle import pyref as R
from tests import kats

pytestmark = pytest.mark.gpu


def test_eip196_vectors_through_every_entry_point(nmx):
    c = R.BN254_G1
    g = nmx.DlogGroup(c.cid)
    cases = kats.load()
    bases, sc = kats.as_arrays(cases)
    total = R.INF
    for i, (name, P, k, Q) in enumerate(cases):
        want = (R.point_to_xy64(Q), False)
        got = g.vartime_multiscalar_mul(sc[i:i + 1], bases[i:i + 1])                  # slice form, n = 1
        assert (got.xy, got.is_inf) == want, name
        total = R.add(c, total, Q)
    want = (R.point_to_xy64(total), False)
    got = g.vartime_multiscalar_mul(sc, bases)
    assert (got.xy, got.is_inf) == want
    ck = nmx.CommitmentKey.from_host(c.cid, bases)                                     # registered key with window tables
    got = g.vartime_multiscalar_mul(sc, ck)
    assert (got.xy, got.is_inf) == want
    got = nmx.CommitmentEngine(c.cid).commit(ck, sc)
    assert (got.xy, got.is_inf) == want
    # padded into a larger MSM (zero scalars elsewhere): the table / sort / reduction pipeline at 2^13 pairs
    n = 1 << 13
    from oracle import cref
    big = cref.sequential_bases(c, 900, n).copy()
    bsc = np.zeros((n, 32), np.uint8)
    pos = [17, 1000, 4096, 8000, n - 1]
    for p_, i in zip(pos, range(len(cases))):
        big[p_] = bases[i]
        bsc[p_] = sc[i]
    got = g.vartime_multiscalar_mul(bsc, big)
    assert (got.xy, got.is_inf) == want
    # (1, 2) x 9 through the small-scalar entry point
    s64 = np.array([9], np.uint64)
    got = g.vartime_multiscalar_mul_small(s64, bases[3:4])
    assert got.xy == R.point_to_xy64(cases[3][3])
    ck.close()


@pytest.mark.parametrize("name", ["bn254_g1", "grumpkin", "pallas", "vesta"])
def test_sympy_vectors_through_the_c_abi(nmx, name):
    """tests/golden/sympy_kats.json (SymPy's elliptic-curve arithmetic; the external anchor of Grumpkin / Pallas / Vesta):
    every scalar multiple through the slice form with n = 1, all of them as one MSM over a repeated generator, the six-term
    MSM through the slice form, a registered key with window tables and `commit`, and padded into a 2^13-pair MSM."""
    from oracle import cref
    d = kats.load_sympy()[name]
    c = R.CURVES[name]
    g = nmx.DlogGroup(c.cid)
    G = (c.gx, c.gy)
    total = R.INF
    for k, Q in d["mul"]:
        bases, sc = kats.points_scalars([G], [k])
        got = g.vartime_multiscalar_mul(sc, bases)
        assert (got.xy, got.is_inf) == (R.point_to_xy64(Q), False), (name, k)
        total = R.add(c, total, Q)
    bases, sc = kats.points_scalars([G] * len(d["mul"]), [k for k, _ in d["mul"]])
    got = g.vartime_multiscalar_mul(sc, bases)                       # the same base eleven times: one bucket set, P == Q additions
    assert (got.xy, got.is_inf) == (R.point_to_xy64(total), total is R.INF)
    P, s, want = d["msm"]
    bases, sc = kats.points_scalars(P, s)
    want = (R.point_to_xy64(want), False)
    got = g.vartime_multiscalar_mul(sc, bases)
    assert (got.xy, got.is_inf) == want
    ck = nmx.CommitmentKey.from_host(c.cid, bases)
    got = g.vartime_multiscalar_mul(sc, ck)
    assert (got.xy, got.is_inf) == want
    got = nmx.CommitmentEngine(c.cid).commit(ck, sc)
    assert (got.xy, got.is_inf) == want
    ck.close()
    n = 1 << 13
    big = cref.sequential_bases(c, 500, n).copy()
    bsc = np.zeros((n, 32), np.uint8)
    for pos, i in zip([3, 999, 4095, 4096, 8000, n - 1], range(len(P))):
        big[pos] = bases[i]
        bsc[pos] = sc[i]
    got = g.vartime_multiscalar_mul(bsc, big)
    assert (got.xy, got.is_inf) == want
