"""Shared checks for the inner-product argument (EvaluationEngine of the secondary curve, /root/reference/src/provider/ipa_pc.rs):
the same driver runs against the oracle (CPU, tests/test_oracle_ipa.py) and the HIP path through the C ABI (GPU,
tests/test_gpu_ipa.py).  What pins a prover here is the reference's own VERIFIER (`InnerProductArgument::verify`,
ipa_pc.rs:286-390), restated below with Python integers and the oracle's MSM: the reference's tests hold no IPA vectors, its
prover is checked by `verify` inside the SNARK tests (src/spartan/snark.rs, src/nova/mod.rs:1151-1588).
"""
import hashlib

import numpy as np

from oracle import cref
from oracle import pyref as R
from tests import util


class IpaTranscript:
    """Deterministic stand-in for the Keccak transcript around one IPA round: `absorb(b"L", &L); absorb(b"R", &R); squeeze(b"r")`
    (ipa_pc.rs:231-234) becomes r = SHA3-256(state || L || R) mod p, chained; `force` pins a round's challenge (edge cases)."""

    def __init__(self, p, label=b"nova-mi355x ipa stand-in", force=None):
        self.p, self.state, self.L, self.R, self.rs = p, hashlib.sha3_256(label).digest(), [], [], []
        self.force = dict(force or {})

    def __call__(self, L, Li, R_, Ri):
        self.L.append((L, Li))
        self.R.append((R_, Ri))
        h = hashlib.sha3_256(self.state + L + bytes([Li]) + R_ + bytes([Ri])).digest()
        self.state = h
        r = int.from_bytes(h + hashlib.sha3_256(h).digest(), "little") % self.p
        if r == 0:
            r = 1
        r = self.force.get(len(self.rs), r)
        self.rs.append(r)
        return int(r).to_bytes(32, "little")


def le(x):
    return int(x).to_bytes(32, "little")


def ints(v):
    b = v.tobytes() if isinstance(v, np.ndarray) else bytes(v)
    return [int.from_bytes(b[32 * i: 32 * i + 32], "little") for i in range(len(b) // 32)]


def pt(xy64, is_inf=False):
    """xy64 bytes -> pyref point (None = identity)"""
    return R.INF if is_inf or not any(xy64) else R.xy64_to_point(bytes(xy64))


def pt_bytes(P):
    return bytes(64) if P is R.INF else R.point_to_xy64(P)


def msm_pts(curve, scalars, points):
    """sum scalars[i] * points[i] through the oracle's msm (identity points allowed: they are (0, 0) in the ABI)"""
    sc = np.frombuffer(b"".join(le(s % curve.r) for s in scalars), np.uint8).copy()
    bs = np.frombuffer(b"".join(pt_bytes(P) for P in points), np.uint8).copy()
    out, inf = cref.msm(curve.cid, sc, bs, len(scalars))
    return pt(out, inf)


def make_instance(curve, n, seed, k0=77):
    """(ck xy64 array of n points, ck_c' = the scaled one-point key as xy64, a, b canonical scalar arrays, r0)"""
    ck = cref.sequential_bases(curve, k0, n).copy()
    a = util.random_scalars(curve.cid, n, seed=seed)
    b = util.random_scalars(curve.cid, n, seed=seed + 1)
    u = cref.sequential_bases(curve, 900_001 + seed, 1).copy()          # ck_c = CE::setup(b"ipa", 1) (ipa_pc.rs:50)
    r0 = int.from_bytes(hashlib.sha3_256(b"r0" + bytes([seed & 255])).digest(), "little") % curve.r or 1
    ckc = np.frombuffer(pt_bytes(R.mul(curve, r0, pt(u.tobytes()))), np.uint8).copy()   # ck_c.scale(&r) (:190-191)
    return ck, ckc, a, b


def verify(curve, ck, ckc, a, b, n, Ls, Rs, infs, a_hat, rs):
    """InnerProductArgument::verify (ipa_pc.rs:286-390) from the point where the scaled ck_c is known: True iff the proof passes.
    comm_a = commit(ck, a) and c = <a, b> are recomputed here from the witness (the instance the prover was given)."""
    p = curve.r
    ck = np.asarray(ck).reshape(-1, 64)
    ai, bi = ints(a), ints(b)
    rounds = len(Ls)
    if n != 1 << rounds or len(Rs) != rounds or rounds >= 32:
        return False
    comm_a = msm_pts(curve, ai, [pt(ck[i].tobytes()) for i in range(n)]) if n > 16 else \
        R.msm_naive(curve, ai, [pt(ck[i].tobytes()) for i in range(n)])
    c = sum(x * y for x, y in zip(ai, bi)) % p
    U = pt(ckc.tobytes())
    P = R.add(curve, comm_a, R.mul(curve, c, U))                        # :313
    r_sq = [r * r % p for r in rs]                                       # :325-328
    r_inv = [pow(r, p - 2, p) for r in rs]
    r_inv_sq = [x * x % p for x in r_inv]
    s = [0] * n                                                          # :337-351
    v = 1
    for x in r_inv:
        v = v * x % p
    s[0] = v
    for i in range(1, n):
        pos = i.bit_length() - 1
        s[i] = s[i - (1 << pos)] * r_sq[(rounds - 1) - pos] % p
    ck_hat = msm_pts(curve, s, [pt(ck[i].tobytes()) for i in range(n)])    # :353-356
    b_hat = sum(x * y for x, y in zip(bi, s)) % p                        # :358
    pts = [pt(L, i[0]) for L, i in zip(Ls, infs)] + [pt(Rr, i[1]) for Rr, i in zip(Rs, infs)] + [P]
    P_hat = msm_pts(curve, r_sq + r_inv_sq + [1], pts)                   # :360-380
    ah = int.from_bytes(a_hat, "little")
    rhs = msm_pts(curve, [ah, ah * b_hat % p], [ck_hat, U])             # :382-388
    return P_hat == rhs


def check_ipa(prove, curve, n, seed, force=None, mutate=None):
    """prove(ck, ckc, a, b, n, transcript_fn) -> (Ls, Rs, infs, a_hat).  The proof must pass the reference's verifier; returns the
    proof and the transcript for comparisons between provers."""
    ck, ckc, a, b = make_instance(curve, n, seed)
    if mutate:
        mutate(a, b)
    tr = IpaTranscript(curve.r, force=force)
    Ls, Rs, infs, a_hat = prove(ck, ckc, a, b, n, tr)
    assert len(Ls) == len(Rs) == max(n.bit_length() - 1, 0) == len(tr.rs)
    assert verify(curve, ck, ckc, a, b, n, Ls, Rs, infs, a_hat, tr.rs), "the reference's verifier rejects the proof"
    return (Ls, Rs, infs, a_hat), tr
