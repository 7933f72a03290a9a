"""CPU: the oracle's inner-product argument (oracle/nova_ref.c ref_ipa_prove = ipa_pc.rs:174-281 with the key fold of
pedersen.rs:484-497) against the reference's verifier (tests/ipa_common.py) and against a second, independent statement of the
prover in Python integers at small sizes."""
import numpy as np
import pytest

from oracle import cref
from oracle import pyref as R
from tests import ipa_common as ic

CURVES = [R.BN254_G1, R.GRUMPKIN, R.PALLAS, R.VESTA]


def oracle_prove(curve):
    def prove(ck, ckc, a, b, n, tr):
        return cref.ipa_prove(curve.cid, ck, ckc, a, b, n, cref.make_ipa_transcript(tr))
    return prove


def python_prove(curve, ck, ckc, a, b, n, tr):
    """prove_inner with big integers and affine points, key fold included -- nothing shared with nova_ref.c but the transcript"""
    p = curve.r
    G = [ic.pt(ck[i].tobytes()) for i in range(n)]
    U = ic.pt(ckc.tobytes())
    av, bv = ic.ints(a), ic.ints(b)
    Ls, Rs, infs = [], [], []
    while len(av) > 1:
        h = len(av) // 2
        cL = sum(x * y for x, y in zip(av[:h], bv[h:])) % p
        cR = sum(x * y for x, y in zip(av[h:], bv[:h])) % p
        L = R.msm_naive(curve, av[:h] + [cL], G[h:] + [U])
        Rr = R.msm_naive(curve, av[h:] + [cR], G[:h] + [U])
        Lb, Rb = ic.pt_bytes(L), ic.pt_bytes(Rr)
        r = int.from_bytes(tr(Lb, L is R.INF, Rb, Rr is R.INF), "little")
        ri = pow(r, p - 2, p)
        av = [(x * r + ri * y) % p for x, y in zip(av[:h], av[h:])]
        bv = [(x * ri + r * y) % p for x, y in zip(bv[:h], bv[h:])]
        G = [R.add(curve, R.mul(curve, ri, G[i]), R.mul(curve, r, G[h + i])) for i in range(h)]
        Ls.append(Lb), Rs.append(Rb), infs.append((L is R.INF, Rr is R.INF))
    return Ls, Rs, infs, ic.le(av[0])


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 2, 4, 32, 256])
def test_oracle_proof_passes_the_reference_verifier(curve, n):
    ic.check_ipa(oracle_prove(curve), curve, n, seed=3 + n)


@pytest.mark.parametrize("curve", [R.GRUMPKIN, R.VESTA], ids=lambda c: c.name)
@pytest.mark.parametrize("n", [2, 8, 16])
def test_oracle_equals_the_python_statement(curve, n):
    (Ls, Rs, infs, ah), tr = ic.check_ipa(oracle_prove(curve), curve, n, seed=11)
    ck, ckc, a, b = ic.make_instance(curve, n, 11)
    tr2 = ic.IpaTranscript(curve.r)
    L2, R2, i2, ah2 = python_prove(curve, ck, ckc, a, b, n, tr2)
    assert (Ls, Rs, infs, ah) == (L2, R2, i2, ah2) and tr.rs == tr2.rs


def test_a_tampered_proof_is_rejected():
    curve, n = R.GRUMPKIN, 16
    ck, ckc, a, b = ic.make_instance(curve, n, 5)
    tr = ic.IpaTranscript(curve.r)
    Ls, Rs, infs, ah = cref.ipa_prove(curve.cid, ck, ckc, a, b, n, cref.make_ipa_transcript(tr))
    assert ic.verify(curve, ck, ckc, a, b, n, Ls, Rs, infs, ah, tr.rs)
    bad = ic.le((int.from_bytes(ah, "little") + 1) % curve.r)
    assert not ic.verify(curve, ck, ckc, a, b, n, Ls, Rs, infs, bad, tr.rs)
    assert not ic.verify(curve, ck, ckc, a, b, n, [Rs[0]] + Ls[1:], Rs, infs, ah, tr.rs)
    assert not ic.verify(curve, ck, ckc, a, b, n, Ls, Rs, infs, ah, [tr.rs[0] + 1] + tr.rs[1:])


def test_edge_vectors_and_challenges():
    curve = R.GRUMPKIN
    # a zero witness half: L or R is c * U alone, or the identity when c = 0 as well
    def zero_left(a, b):
        a[: a.shape[0] // 2] = 0
    def zero_all(a, b):
        a[:] = 0
    (Ls, Rs, infs, ah), _ = ic.check_ipa(oracle_prove(curve), curve, 8, seed=2, mutate=zero_left)
    (Ls, Rs, infs, ah), _ = ic.check_ipa(oracle_prove(curve), curve, 8, seed=2, mutate=zero_all)
    assert all(i == (True, True) for i in infs) and ah == bytes(32)
    # challenges 1 and p - 1
    ic.check_ipa(oracle_prove(curve), curve, 16, seed=4, force={0: 1, 2: curve.r - 1})


def test_bad_arguments():
    curve = R.GRUMPKIN
    ck, ckc, a, b = ic.make_instance(curve, 8, 1)
    tr = cref.make_ipa_transcript(ic.IpaTranscript(curve.r))
    with pytest.raises(ValueError):
        cref.ipa_prove(curve.cid, ck, ckc, a[:6], b[:6], 6, tr)             # not a power of two
    with pytest.raises(ValueError):
        cref.ipa_prove(curve.cid, ck, ckc, a, b, 8, cref.make_ipa_transcript(ic.IpaTranscript(curve.r, force={1: 0})))  # r = 0
    big = a.copy()
    big[0] = np.frombuffer(ic.le(curve.r), np.uint8)
    with pytest.raises(ValueError):
        cref.ipa_prove(curve.cid, ck, ckc, big, b, 8, tr)                   # scalar >= modulus
