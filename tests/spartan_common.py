"""Shared checks for Spartan's sum-check provers (BASELINE.json configs[4], the sum-check half): the same drivers run against
the oracle (CPU, tests/test_oracle_spartan.py) and the HIP path through the C ABI (GPU, tests/test_gpu_spartan.py).

What pins a prover here is the reference's own VERIFIER, restated with Python big integers -- the reference has no
prover-level test; its provers are checked by `RelaxedR1CSSNARK::verify` (/root/reference/src/spartan/snark.rs:262-400):
  SumcheckProof::verify            src/spartan/sumcheck.rs:87-129  (p(0) + p(1) == e; e' = p(r))
  outer final claim                snark.rs:281-289   eq(tau, r_x) * (Az(r_x) Bz(r_x) - uCz_E(r_x))
  inner final claim                snark.rs:303-345   claim == ABC(r_y) * Z(r_y)
  verify_batch                     sumcheck.rs:131-161 + spartan/mod.rs:440-470
plus the DEFINITION of every round polynomial by brute force over the hypercube at small sizes.
"""
import hashlib

import numpy as np

from tests import fv_common as fc
from tests import util


class StandInTranscript:
    """Deterministic stand-in for the Keccak transcript (src/provider/keccak.rs is host code of the reference and out of scope):
    `absorb(b"p", poly); squeeze(b"c")` becomes challenge = SHA3-256(state || coefficients) mod p, chained.  Every challenge depends on
    every earlier round polynomial, so two provers produce the same challenge sequence iff their round polynomials agree."""

    def __init__(self, p, label=b"nova-mi355x stand-in", force=None):
        self.p, self.state, self.polys, self.rs = p, hashlib.sha3_256(label).digest(), [], []
        self.force = dict(force or {})      # round index -> forced challenge (edge cases: 0, 1, tau-cancelling values)

    def __call__(self, coeffs):
        self.polys.append([int.from_bytes(c, "little") for c in coeffs])
        h = hashlib.sha3_256(self.state + b"".join(coeffs)).digest()
        self.state = h
        r = int.from_bytes(h + hashlib.sha3_256(h).digest(), "little") % self.p
        r = self.force.get(len(self.rs), r)
        self.rs.append(r)
        return int(r).to_bytes(32, "little")


def le(x):
    return int(x).to_bytes(32, "little")


def ints(v):
    return fc.ints(np.frombuffer(v, np.uint8) if isinstance(v, (bytes, bytearray)) else v)


def eq_eval(p, a, b):
    """EqPolynomial::evaluate (src/spartan/polys/eq.rs:42-52)"""
    acc = 1
    for x, y in zip(a, b):
        acc = acc * ((x * y + (1 - x) * (1 - y)) % p) % p
    return acc


def mle_eval(p, z, r):
    """multilinear extension of z (len 2^len(r)) at r, r[0] binding the top variable (multilinear.rs:65-84 repeated)"""
    cur = list(z)
    for x in r:
        h = len(cur) // 2
        cur = [(cur[i] + x * (cur[i + h] - cur[i])) % p for i in range(h)]
    assert len(cur) == 1
    return cur[0]


def poly_at(p, coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % p
    return acc


def verify_rounds(p, claim, polys, rs, degree_bound):
    """SumcheckProof::verify (sumcheck.rs:87-129): returns the final claim e."""
    e = claim % p
    assert len(polys) == len(rs)
    for coeffs, r in zip(polys, rs):
        assert len(coeffs) - 1 <= degree_bound
        assert (poly_at(p, coeffs, 0) + poly_at(p, coeffs, 1)) % p == e, "p(0) + p(1) != e"
        e = poly_at(p, coeffs, r)
    return e


def brute_round_poly_cubic(p, taus, A, B, C, rs_so_far, x):
    """s_j(x) = sum_{y in {0,1}^(l-j)} eq(tau, (r_1..r_{j-1}, x, y)) (A B - C)(r_1..r_{j-1}, x, y): the definition."""
    l = len(taus)
    j = len(rs_so_far)
    rest = l - j - 1
    tot = 0
    for y in range(1 << rest):
        pt = list(rs_so_far) + [x] + [(y >> (rest - 1 - t)) & 1 for t in range(rest)]
        a, b, c = mle_eval(p, A, pt), mle_eval(p, B, pt), mle_eval(p, C, pt)
        tot = (tot + eq_eval(p, taus, pt) * (a * b - c)) % p
    return tot


def check_cubic3(prove, fid, l, seed, force=None, taus=None, brute=None):
    """prove(fid, claim, taus_vec, A, B, C, transcript_callable) -> (polys, r, claims) as 32-byte strings."""
    p = fc.FIELDS[fid]
    n = 1 << l
    A, B, C = (fc.edge_vectors(fid, n, seed + i) for i in range(3))
    tv = fc.rand_vec(fid, l, seed + 7).copy() if taus is None else fc.vec(taus)
    Ai, Bi, Ci, ti = ints(A), ints(B), ints(C), ints(tv)
    # the claim of a general instance: sum_x eq(tau, x) (A B - C)(x)  (Spartan's outer claim is 0 for a satisfied instance)
    claim = 0
    for x in range(n):
        bits = [(x >> (l - 1 - t)) & 1 for t in range(l)]
        claim = (claim + eq_eval(p, ti, bits) * (Ai[x] * Bi[x] - Ci[x])) % p
    tr = StandInTranscript(p, force=force)
    polys, rs, claims = prove(fid, le(claim), tv, A, B, C, tr)
    polys_i = [[int.from_bytes(c, "little") for c in row] for row in polys]
    rs_i = [int.from_bytes(x, "little") for x in rs]
    assert polys_i == tr.polys and rs_i == tr.rs, "the prover must hand the transcript exactly what it returns"
    e = verify_rounds(p, claim, polys_i, rs_i, 3)
    cl = [int.from_bytes(c, "little") for c in claims]
    assert cl == [mle_eval(p, Ai, rs_i), mle_eval(p, Bi, rs_i), mle_eval(p, Ci, rs_i)], "final claims are A(r), B(r), C(r)"
    assert e == eq_eval(p, ti, rs_i) * (cl[0] * cl[1] - cl[2]) % p, "snark.rs:281-289"
    if brute if brute is not None else l <= 5:
        for j in range(l):
            for x in (0, 1, 2, p - 1):
                assert poly_at(p, polys_i[j], x) == brute_round_poly_cubic(p, ti, Ai, Bi, Ci, rs_i[:j], x), (j, x)
    return polys, rs, claims


def check_quad_prod(prove, fid, l, seed, force=None):
    """prove(fid, claim, num_rounds, A, B, transcript) -> (polys, r, [A(r), B(r)])"""
    p = fc.FIELDS[fid]
    n = 1 << l
    A, B = fc.edge_vectors(fid, n, seed), fc.edge_vectors(fid, n, seed + 1)
    Ai, Bi = ints(A), ints(B)
    claim = sum(a * b for a, b in zip(Ai, Bi)) % p
    tr = StandInTranscript(p, force=force)
    polys, rs, claims = prove(fid, le(claim), l, A, B, tr)
    polys_i = [[int.from_bytes(c, "little") for c in row] for row in polys]
    rs_i = [int.from_bytes(x, "little") for x in rs]
    assert polys_i == tr.polys and rs_i == tr.rs
    e = verify_rounds(p, claim, polys_i, rs_i, 2)
    cl = [int.from_bytes(c, "little") for c in claims]
    assert cl == [mle_eval(p, Ai, rs_i), mle_eval(p, Bi, rs_i)]
    assert e == cl[0] * cl[1] % p, "snark.rs:343-345"
    return polys, rs, claims


def check_batch_eval(prove, fid, num_rounds, seed, force=None, zero_coords=None):
    """prove(fid, claims, num_rounds, polys, eq_points, coeffs, transcript) -> (polys, r, finals).  Verified as
    batch_eval_verify does (src/spartan/mod.rs:440-470 with sumcheck.rs:131-161).  zero_coords = {claim: [coordinates]}: those
    coordinates of that claim's evaluation point are zero (its eq instance then takes the reference's tau = 0 fall-back in those
    rounds, sumcheck.rs:1085-1136, while the other claims do not)."""
    p = fc.FIELDS[fid]
    k = len(num_rounds)
    nmax = max(num_rounds)
    P = [fc.edge_vectors(fid, 1 << nr, seed + 3 * i) for i, nr in enumerate(num_rounds)]
    X = [fc.rand_vec(fid, max(nr, 1), seed + 100 + i)[:nr].copy() for i, nr in enumerate(num_rounds)]
    for i, coords in (zero_coords or {}).items():
        for j in coords:
            X[i][j] = 0
    Pi, Xi = [ints(v) for v in P], [ints(x) if len(x) else [] for x in X]
    claims = [mle_eval(p, Pi[i], Xi[i]) for i in range(k)]
    rho = ints(fc.rand_vec(fid, 1, seed + 999))[0]
    coeffs = [pow(rho, i, p) for i in range(k)]             # powers::<E>(&rho, num_claims)
    tr = StandInTranscript(p, force=force)
    polys, rs, finals = prove(fid, [le(c) for c in claims], list(num_rounds), P, X, [le(c) for c in coeffs], tr)
    polys_i = [[int.from_bytes(c, "little") for c in row] for row in polys]
    rs_i = [int.from_bytes(x, "little") for x in rs]
    assert polys_i == tr.polys and rs_i == tr.rs
    joint = sum(claims[i] * pow(2, nmax - num_rounds[i], p) * coeffs[i] for i in range(k)) % p
    e = verify_rounds(p, joint, polys_i, rs_i, 2)
    fin = [int.from_bytes(c, "little") for c in finals]
    exp = 0
    for i in range(k):
        r_i = rs_i[nmax - num_rounds[i]:]                   # the polynomial joins when `remaining_rounds <= num_rounds[i]`
        assert fin[i] == mle_eval(p, Pi[i], r_i), i
        exp = (exp + coeffs[i] * eq_eval(p, Xi[i], r_i) * fin[i]) % p
    assert e == exp, "spartan/mod.rs:456-466: claim_batch_final == sum coeff_i eq(x_i, r_i) P_i(r_i)"
    return polys, rs, finals


def transposed_kat():
    """The reference's SpMV known answer (src/r1cs/sparse.rs:452-466: [[0,2,7],[0,0,3],[4,0,0]] x [1,2,3] = [25,9,4]) read the
    other way for compute_eval_table_sparse (src/spartan/mod.rs:497-533): M^T x [1,2,3] = [4*3, 2*1, 7*1 + 3*2] = [12, 2, 13]."""
    return dict(indptr=[0, 2, 3, 4], indices=[1, 2, 2, 0], data=[2, 7, 3, 4], rows=3, cols=3, x=[1, 2, 3], out=[12, 2, 13])


def heavy_column_csr(fid, rows, cols, seed, heavy_cols=(0,), per_row=3):
    """CSR with a few columns that appear in EVERY row (the constant-one column of an R1CS instance does) on top of random ones."""
    rng = np.random.Generator(np.random.PCG64(seed))
    counts = rng.integers(0, per_row + 1, size=rows) + len(heavy_cols)
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    nnz = int(indptr[-1])
    indices = rng.integers(0, cols, size=nnz).astype(np.uint64)
    for r in range(rows):
        for t, hc in enumerate(heavy_cols):
            indices[int(indptr[r]) + t] = hc
    data = fc.rand_vec(fid, nnz, seed + 1).copy()
    p = fc.FIELDS[fid]
    small = rng.random(nnz) < 0.6
    vals = rng.integers(0, 4, size=nnz)
    for k in np.nonzero(small)[0]:
        data[k] = util.int_to_le32([1, p - 1, 2, p - 3][vals[k]])
    return indptr, indices, data


def dense_transposed(p, indptr, indices, data, cols, x):
    out = [0] * cols
    d, xi = ints(data), ints(x)
    for r in range(len(indptr) - 1):
        for k in range(int(indptr[r]), int(indptr[r + 1])):
            out[int(indices[k])] = (out[int(indices[k])] + xi[r] * d[k]) % p
    return out
