"""Tier-1 oracle: Python big-int ground truth for the commitment / MSM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under nova_amd/ may import this module; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may (see DESIGN.md "oracle").

What it restates (citations relative to /root/reference):
  * the *definition* the reference's own tests use for MSM parity:  result == sum_i s_i * P_i in the
    group, compared as affine coordinates (src/provider/msm.rs:722-739, src/provider/blitzar.rs:69-214,
    src/provider/curve_property_tests.rs:172-218);
  * output normalisation of `to_coordinates`: identity -> (0, 0, true) (src/provider/traits.rs:303-312);
  * the Pedersen / HyperKZG commit formula  msm(v, ck[..len v]) + h*r  (src/provider/pedersen.rs:263-270,
    src/provider/hyperkzg.rs:584-591);
  * field moduli / group orders as the hex strings in src/provider/bn256_grumpkin.rs:39-40,84-85 and
    src/provider/pasta.rs:37-38,45-46.

The arithmetic itself lives in the third-party crate halo2curves 0.9.0 (Cargo.toml:36-41), which is NOT in
/root/reference and cannot be built here (no Rust toolchain).  Parity pinning: the reference stores no MSM
output vector anywhere -- parity is pinned by definition (group-law result) and this
file is pinned by (i) the reference's modulus/order strings, (ii) order * G == identity for every curve,
(iii) public known-answer points (BN254 2G / 3G as in EIP-196 test vectors), all checked in
tests/test_oracle.py.  By the build rules this counts as "parity unpinned at stored-vector level".
"""

from dataclasses import dataclass

# --- constants: reference file:line given per entry ------------------------------------------------
BN254_R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # bn256_grumpkin.rs:39
BN254_Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # bn256_grumpkin.rs:40
PALLAS_P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001  # pasta.rs:38 (base)
PALLAS_Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001  # pasta.rs:37 (order)


@dataclass(frozen=True)
class Curve:
    """Short-Weierstrass curve y^2 = x^3 + b over F_p (a = 0, relied on at msm.rs:35-36)."""
    name: str
    cid: int      # curve id used by the C ABI (include/nova_mi355x.h)
    p: int        # base-field modulus
    r: int        # group order == scalar-field modulus
    b: int
    gx: int
    gy: int


BN254_G1 = Curve("bn254_g1", 0, BN254_Q, BN254_R, 3, 1, 2)
GRUMPKIN = Curve("grumpkin", 1, BN254_R, BN254_Q, (-17) % BN254_R, 1,
                 0x2CF135E7506A45D632D270D45F1181294833FC48D823F272C)
PALLAS = Curve("pallas", 2, PALLAS_P, PALLAS_Q, 5, PALLAS_P - 1, 2)
VESTA = Curve("vesta", 3, PALLAS_Q, PALLAS_P, 5, PALLAS_Q - 1, 2)
CURVES = {c.name: c for c in (BN254_G1, GRUMPKIN, PALLAS, VESTA)}
CURVES_BY_ID = {c.cid: c for c in CURVES.values()}

INF = None  # affine identity


def on_curve(c: Curve, P) -> bool:
    if P is INF:
        return True
    x, y = P
    return (y * y - (x * x * x + c.b)) % c.p == 0


def neg(c: Curve, P):
    if P is INF:
        return INF
    return (P[0], (-P[1]) % c.p)


def add(c: Curve, P, Q):
    """Affine group law (complete: handles identity, doubling, inverse)."""
    if P is INF:
        return Q
    if Q is INF:
        return P
    p = c.p
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return INF
        lam = (3 * x1 * x1) * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    y3 = (lam * (x1 - x3) - y1) % p
    return (x3, y3)


# Jacobian arithmetic for speed on larger cases (still plain big-int, still definitional)
def _jac_dbl(c, P):
    X, Y, Z = P
    if Z == 0 or Y == 0:
        return (1, 1, 0)
    p = c.p
    A = X * X % p
    B = Y * Y % p
    C = B * B % p
    D = 2 * ((X + B) * (X + B) - A - C) % p
    E = 3 * A % p
    X3 = (E * E - 2 * D) % p
    Y3 = (E * (D - X3) - 8 * C) % p
    Z3 = 2 * Y * Z % p
    return (X3, Y3, Z3)


def _jac_add_affine(c, P, Q):
    """Jacobian P + affine Q (Q != INF)."""
    X1, Y1, Z1 = P
    if Z1 == 0:
        return (Q[0], Q[1], 1)
    p = c.p
    Z1Z1 = Z1 * Z1 % p
    U2 = Q[0] * Z1Z1 % p
    S2 = Q[1] * Z1 * Z1Z1 % p
    if U2 == X1:
        if S2 == Y1:
            return _jac_dbl(c, P)
        return (1, 1, 0)
    H = (U2 - X1) % p
    R = (S2 - Y1) % p
    HH = H * H % p
    HHH = H * HH % p
    V = X1 * HH % p
    X3 = (R * R - HHH - 2 * V) % p
    Y3 = (R * (V - X3) - Y1 * HHH) % p
    Z3 = Z1 * H % p
    return (X3, Y3, Z3)


def _jac_to_affine(c, P):
    X, Y, Z = P
    if Z == 0:
        return INF
    p = c.p
    zi = pow(Z, -1, p)
    zi2 = zi * zi % p
    return (X * zi2 % p, Y * zi2 * zi % p)


def mul(c: Curve, k: int, P):
    """k * P, double-and-add over the integer k (k taken as given, NOT reduced: tests rely on r*G == O)."""
    if P is INF or k == 0:
        return INF
    if k < 0:
        return mul(c, -k, neg(c, P))
    acc = (1, 1, 0)
    for bit in bin(k)[2:]:
        acc = _jac_dbl(c, acc)
        if bit == "1":
            acc = _jac_add_affine(c, acc, P)
    return _jac_to_affine(c, acc)


def msm_naive(c: Curve, scalars, bases):
    """sum_i s_i * P_i  -- the definition the reference tests compare against (msm.rs:730-735)."""
    assert len(scalars) == len(bases)  # msm.rs:226
    acc = INF
    for s, P in zip(scalars, bases):
        acc = add(c, acc, mul(c, s % c.r, P))
    return acc


def commit(c: Curve, ck, h, v, r):
    """Pedersen/HyperKZG commit: msm(v, ck[..len v]) + h*r (pedersen.rs:263-270, hyperkzg.rs:584-591)."""
    assert len(ck) >= len(v)
    return add(c, msm_naive(c, v, ck[: len(v)]), mul(c, r % c.r, h))


def sequential_bases(c: Curve, k0: int, n: int):
    """P_i = (k0 + i) * G, i in [0, n): the construction of curve_property_tests.rs:186-194."""
    G = (c.gx, c.gy)
    out = []
    P = mul(c, k0, G)
    for _ in range(n):
        out.append(P)
        P = add(c, P, G)
    return out


# --- byte marshalling identical to the C ABI (canonical little-endian, identity = all-zero) ----------
def fe_to_le32(x: int) -> bytes:
    return int(x).to_bytes(32, "little")


def point_to_xy64(P) -> bytes:
    """to_coordinates() layout: x||y canonical LE, identity -> 64 zero bytes (traits.rs:303-312)."""
    if P is INF:
        return bytes(64)
    return fe_to_le32(P[0]) + fe_to_le32(P[1])


def xy64_to_point(b: bytes):
    x = int.from_bytes(b[:32], "little")
    y = int.from_bytes(b[32:64], "little")
    if x == 0 and y == 0:
        return INF
    return (x, y)


# --- field-vector definitions for the "next" rows -------------------------------------------
def axpy(p, a, b, r):
    """W = W1 + r*W2 element-wise (src/r1cs/mod.rs:1058-1067)."""
    return [(x + r * y) % p for x, y in zip(a, b)]


def bind_poly_var_top(p, Z, r):
    """Z[i] = Z[i] + r*(Z[i+n] - Z[i]), n = len/2 (src/spartan/polys/multilinear.rs:65-84)."""
    n = len(Z) // 2
    return [(Z[i] + r * (Z[i + n] - Z[i])) % p for i in range(n)]


def axpy2(p, a, b, c, r):
    """E = E1 + r*T + r^2*E2 (src/r1cs/mod.rs:1096-1101)."""
    return [(x + r * y + r * r * z) % p for x, y, z in zip(a, b, c)]


def cross_term(p, az, bz, cz, e, u):
    """T = AZ o BZ - u*CZ - E (src/r1cs/mod.rs:614-620)."""
    return [(a * b - u * c - ee) % p for a, b, c, ee in zip(az, bz, cz, e)]


def cross_term2(p, az, bz, cz, e1, e2, u):
    """T = AZ o BZ - u*CZ - E1 - E2 (commit_T_relaxed, src/r1cs/mod.rs:652-659)."""
    return [(a * b - u * c - x - y) % p for a, b, c, x, y in zip(az, bz, cz, e1, e2)]


def fold_pairs(p, P, x):
    """Pi[j] = P[2j] + x*(P[2j+1] - P[2j]) (src/provider/hyperkzg.rs:1085-1095)."""
    return [(P[2 * j] + x * (P[2 * j + 1] - P[2 * j])) % p for j in range(len(P) // 2)]


def sumcheck_eq_sums(p, mode, A, B, C, eq_right, eq_left=None, shift=0):
    """(t_0, t_inf) of the eq-factored sum-check rounds (src/spartan/sumcheck.rs:900-1075)."""
    h = len(A) // 2
    t0 = tinf = 0
    for i in range(h):
        fac = eq_right[i & ((1 << shift) - 1)] * eq_left[i >> shift] if eq_left is not None else eq_right[i]
        if mode == 1:
            t0 += A[i] * fac
            continue
        c0 = C[i] if mode == 3 else 1
        t0 += (A[i] * B[i] - c0) * fac
        tinf += (A[i + h] - A[i]) * (B[i + h] - B[i]) * fac
    return t0 % p, tinf % p


def eq_evals(p, r):
    """eq(r, x) for x in {0,1}^ell, x read MSB-first against r[0] (src/spartan/polys/eq.rs:29-41,54-73)."""
    ell = len(r)
    out = []
    for x in range(1 << ell):
        v = 1
        for i in range(ell):
            bit = (x >> (ell - 1 - i)) & 1
            v = v * (r[i] if bit else (1 - r[i])) % p
        out.append(v)
    return out


def mle_evaluate(p, Z, r):
    """Z(r) = sum_x Z[x] * eq(r, x) (src/spartan/polys/multilinear.rs:88-129)."""
    return sum(z * e for z, e in zip(Z, eq_evals(p, r))) % p


def spmv(p, indptr, indices, data, z):
    """CSR M*z (src/r1cs/sparse.rs:201-229)."""
    return [sum(data[k] * z[indices[k]] for k in range(indptr[r], indptr[r + 1])) % p for r in range(len(indptr) - 1)]


def poly_eval(p, f, u):
    """Horner (src/provider/hyperkzg.rs:1011-1020): coefficients low to high."""
    acc = 0
    for c in reversed(f):
        acc = (acc * u + c) % p
    return acc


def div_by_monomial(p, f, u):
    """h with f(x) = h(x) (x - u) + f(u): h[i-1] = f[i] + h[i]*u (src/provider/hyperkzg.rs:946-999)."""
    h = [0] * (len(f) - 1)
    nxt = 0
    for i in range(len(f) - 1, 0, -1):
        nxt = (f[i] + nxt * u) % p
        h[i - 1] = nxt
    return h


def sumcheck_plain_sums(p, kind, A, B, C=None):
    """Round sums without an eq factor (src/spartan/sumcheck.rs:163-186, 353-443); kinds as in nova_mi355x.h."""
    h = len(A) // 2
    s0 = s1 = s2 = 0
    for i in range(h):
        a0, a1, b0, b1 = A[i], A[i + h], B[i], B[i + h]
        dA, dB = a1 - a0, b1 - b0
        am, bm = 2 * a0 - a1, 2 * b0 - b1
        if kind == 1:
            s0 += a0 * b0
            s1 += dA * dB
        elif kind == 2:
            s0 += a0 - b0
            s1 += am - bm
        elif kind == 3:
            s0 += a0 * b0
            s1 += am * bm
        else:
            c0, c1 = C[i], C[i + h]
            dC = c1 - c0
            s0 += a0 * b0 * c0
            s1 += dA * dB * dC
            s2 += (a0 - dA) * (b0 - dB) * (c0 - dC)
    return s0 % p, s1 % p, s2 % p


def lincomb_powers(p, vecs, s, n_out):
    """PolyEvalWitness::batch / batch_diff_size: sum_j s^j * P_j, zero-padded (src/spartan/mod.rs:165-277)."""
    out = [0] * n_out
    pw = 1
    for v in vecs:
        for i, x in enumerate(v):
            out[i] = (out[i] + pw * x) % p
        pw = pw * s % p
    return out
