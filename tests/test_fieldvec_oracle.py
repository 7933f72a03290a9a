"""CPU: pins the oracle's field-vector restatements (oracle/nova_ref.c) against the big-int definitions and the
reference's known-answer tests (tests/golden/field_kats.json)."""
import pytest

from oracle import cref
from oracle import pyref as R
from tests import fv_common as C


def test_oracle_kats():
    C.check_kats(lambda fid, p, x: cref.field_bind(fid, p, 0, 1, 2, x, len(p) // 2),
                 lambda fid, z, r: cref.field_bind(fid, z, 0, len(z) // 2, 1, r, len(z) // 2))
    # the definitions themselves
    for case in C.KATS["hyperkzg_fold_eval"]["cases"]:
        cur = case["poly"]
        for x in reversed(case["point"]):
            cur = R.fold_pairs(R.BN254_R, cur, x)
        assert cur == [case["eval"]]
    for case in C.KATS["mle_bind_top_eval"]["cases"]:
        cur = case["evals"]
        for r in case["point"]:
            cur = R.bind_poly_var_top(R.BN254_R, cur, r)
        assert cur == [case["eval"]]


@pytest.mark.parametrize("fid", range(4))
def test_oracle_field_vectors(fid):
    p = C.FIELDS[fid]
    n = 64
    a, b, c, e = (C.edge_vectors(fid, n, s) for s in (1, 2, 3, 4))
    r = C.rand_vec(fid, 1, 9)
    ri = C.ints(r)[0]
    enc = lambda xs: b"".join(R.fe_to_le32(x) for x in xs)
    assert cref.field_axpy(fid, a, b, r, n) == enc(R.axpy(p, C.ints(a), C.ints(b), ri))
    assert cref.field_axpy2(fid, a, b, c, r, n) == enc(R.axpy2(p, C.ints(a), C.ints(b), C.ints(c), ri))
    assert cref.field_cross_term(fid, a, b, c, e, r, n) == enc(R.cross_term(p, C.ints(a), C.ints(b), C.ints(c), C.ints(e), ri))
    assert cref.field_cross_term2(fid, a, b, c, e, b, r, n) == enc(
        R.cross_term2(p, C.ints(a), C.ints(b), C.ints(c), C.ints(e), C.ints(b), ri))
    assert cref.field_bind(fid, a, 0, n // 2, 1, r, n // 2) == enc(R.bind_poly_var_top(p, C.ints(a), ri))
    assert cref.field_bind(fid, a, 0, 1, 2, r, n // 2) == enc(R.fold_pairs(p, C.ints(a), ri))


@pytest.mark.parametrize("fid", range(4))
def test_oracle_sumcheck_eq_sums(fid):
    """sumcheck.rs:900-1075: C restatement == big-int definition, first-half (eqL x eqR) and last-half forms."""
    p = C.FIELDS[fid]
    n, shift = 64, 3
    A, B, Cc = (C.edge_vectors(fid, n, s) for s in (1, 2, 3))
    eqR, eqL, eqF = C.rand_vec(fid, 1 << shift, 5), C.rand_vec(fid, (n // 2) >> shift, 6), C.rand_vec(fid, n // 2, 7)
    enc = lambda t: tuple(R.fe_to_le32(x) for x in t)
    for mode in (1, 2, 3):
        got = cref.sumcheck_eq_sums(fid, mode, A, B, Cc, n, eqR, eqL, shift)
        assert got == enc(R.sumcheck_eq_sums(p, mode, C.ints(A), C.ints(B), C.ints(Cc), C.ints(eqR), C.ints(eqL), shift))
        got = cref.sumcheck_eq_sums(fid, mode, A, B, Cc, n, eqF)
        assert got == enc(R.sumcheck_eq_sums(p, mode, C.ints(A), C.ints(B), C.ints(Cc), C.ints(eqF)))


def test_oracle_kats_eq_mle_spmv():
    C.check_kats2(lambda fid, r: cref.eq_evals(fid, r, len(r)),
                  lambda fid, z, r: cref.mle_evaluate(fid, z, len(r), r),
                  lambda fid, ip, ix, d, cols, z: cref.spmv(fid, ip, ix, d, len(ip) - 1, z))


def test_oracle_kats_multi_evaluate_and_the_mixed_coefficient_matrix():
    """round 6's additions to tests/golden/field_kats.json: multi_evaluate_with's known values (multilinear.rs:456-485) and the reference's
    mixed-coefficient SpMV fixture (sparse.rs:486-544: +1, -1, small, small negative, general, an empty row) against the dense product."""
    C.check_kats3(lambda fid, zs, r: cref.mle_multi_evaluate(fid, [z.tobytes() for z in zs], len(r), r),
                  lambda fid, ip, ix, d, cols, z: cref.spmv(fid, ip, ix, d, len(ip) - 1, z),
                  lambda fid, ip, ix, d, cols, z1, z2: cref.spmv_pair(fid, ip, ix, d, len(ip) - 1, z1, z2))


@pytest.mark.parametrize("fid", range(4))
def test_oracle_eq_mle_spmv_vs_definition(fid):
    p = C.FIELDS[fid]
    enc = lambda xs: b"".join(R.fe_to_le32(x) for x in xs)
    for ell in (0, 1, 4, 7):
        r = C.rand_vec(fid, max(ell, 1), 3)[:ell]
        assert cref.eq_evals(fid, r, ell) == enc(R.eq_evals(p, C.ints(r)))
        z = C.edge_vectors(fid, 1 << ell, 5) if ell else C.rand_vec(fid, 1, 5)
        assert cref.mle_evaluate(fid, z, ell, r) == enc([R.mle_evaluate(p, C.ints(z), C.ints(r))])
    ip, ix, d = C.random_csr(fid, 50, 30, 9)
    z = C.rand_vec(fid, 30, 10)
    assert cref.spmv(fid, ip, ix, d, 50, z) == enc(R.spmv(p, [int(x) for x in ip], [int(x) for x in ix], C.ints(d), C.ints(z)))


@pytest.mark.parametrize("fid", range(4))
def test_oracle_horner_and_div_by_monomial(fid):
    """hyperkzg.rs:946-1020: the quotient really divides (f(x) = h(x)(x-u) + f(u)) and the C pass matches."""
    p = C.FIELDS[fid]
    for n in (1, 2, 7, 200):
        f = C.edge_vectors(fid, n, 3) if n > 5 else C.rand_vec(fid, n, 3)
        u = C.rand_vec(fid, 1, 4)
        fi, ui = C.ints(f), C.ints(u)[0]
        h, ev = R.div_by_monomial(p, fi, ui), R.poly_eval(p, fi, ui)
        # multiply back: h(x)(x-u) + ev == f(x)
        back = [0] * n
        for i, c in enumerate(h):
            back[i + 1] = (back[i + 1] + c) % p
            back[i] = (back[i] - c * ui) % p
        back[0] = (back[0] + ev) % p
        assert back == fi
        got = cref.suffix_horner(fid, f, n, u)
        assert got == b"".join(R.fe_to_le32(x) for x in [ev] + h)


@pytest.mark.parametrize("fid", range(4))
def test_oracle_plain_sums_lincomb_multi_eval_pair(fid):
    """sumcheck.rs:163-186,353-443; spartan/mod.rs:165-277; multilinear.rs:131-180; sparse.rs:215-229:
    C restatements == big-int definitions."""
    p = C.FIELDS[fid]
    enc1 = R.fe_to_le32
    n = 64
    A, B, Cc = (C.edge_vectors(fid, n, s) for s in (1, 2, 3))
    for kind in (1, 2, 3, 4):
        got = cref.sumcheck_plain_sums(fid, kind, A, B, Cc if kind == 4 else None, n)
        exp = R.sumcheck_plain_sums(p, kind, C.ints(A), C.ints(B), C.ints(Cc))
        assert got == tuple(enc1(x) for x in exp), kind
    # quad_prod consistency with the round polynomial: s0 + (s0 + s_quad + linear) = claim is the caller's algebra;
    # here: kind 3's A(-1)B(-1) relates to kind 1 by  S(-1) = S(0) - (S(1) - S(0) - q) + q  with q the dA*dB sum
    s0, q, _ = R.sumcheck_plain_sums(p, 1, C.ints(A), C.ints(B))
    _, sm1, _ = R.sumcheck_plain_sums(p, 3, C.ints(A), C.ints(B))
    s1 = sum(a * b for a, b in zip(C.ints(A)[n // 2:], C.ints(B)[n // 2:])) % p
    assert sm1 == (2 * s0 - s1 + 2 * q) % p
    # random linear combination, ragged lengths
    vecs = [C.edge_vectors(fid, m, 10 + m) for m in (64, 17, 64, 1)] + [C.rand_vec(fid, 1, 3)[:0]]
    s = C.rand_vec(fid, 1, 77)
    exp = b"".join(enc1(x) for x in R.lincomb_powers(p, [C.ints(v) for v in vecs], C.ints(s)[0], 64))
    assert cref.lincomb_powers(fid, [v.tobytes() for v in vecs], s, 64) == exp
    one = C.vec([1])
    assert cref.lincomb_powers(fid, [vecs[0].tobytes(), vecs[2].tobytes()], one, 64) == \
        cref.field_axpy(fid, vecs[0], vecs[2], one, 64)
    # multi_evaluate_with == evaluate_with per polynomial (multilinear.rs:414-439)
    for ell in (0, 1, 5, 6):
        r = C.rand_vec(fid, max(ell, 1), 3)[:ell]
        zs = [C.edge_vectors(fid, 1 << ell, 20 + j) if ell > 2 else C.rand_vec(fid, 1 << ell, 20 + j) for j in range(3)]
        assert cref.mle_multi_evaluate(fid, [z.tobytes() for z in zs], ell, r) == [cref.mle_evaluate(fid, z, ell, r) for z in zs]
    assert cref.mle_multi_evaluate(fid, [], 3, C.rand_vec(fid, 3, 1)) == []
    # multiply_vec_pair == two multiply_vec (sparse.rs:531-544)
    ip, ix, d = C.random_csr(fid, 50, 30, 9)
    z1, z2 = C.rand_vec(fid, 30, 10), C.edge_vectors(fid, 30, 11)
    assert cref.spmv_pair(fid, ip, ix, d, 50, z1, z2) == (cref.spmv(fid, ip, ix, d, 50, z1), cref.spmv(fid, ip, ix, d, 50, z2))


def test_oracle_multi_evaluate_known_values():
    """The reference's stored known-answer case (multilinear.rs:456-485): p(x1,x2,x3) = (x1 + x2) * x3, evaluations
    indexed x1-MSB = [0,0,0,1,0,1,0,2], and the constant polynomial 5, at (1,1,1) -> [2, 5]; plus (2,3,4) -> 20."""
    for fid in range(4):
        z1 = C.vec([0, 0, 0, 1, 0, 1, 0, 2])
        z2 = C.vec([5] * 8)
        got = cref.mle_multi_evaluate(fid, [z1.tobytes(), z2.tobytes()], 3, C.vec([1, 1, 1]))
        assert [int.from_bytes(g, "little") for g in got] == [2, 5]
        got = cref.mle_multi_evaluate(fid, [z1.tobytes(), z2.tobytes()], 3, C.vec([2, 3, 4]))
        assert [int.from_bytes(g, "little") for g in got] == [20, 5]


def test_batch_invert_oracle():
    """batch_invert (src/spartan/mod.rs:54-152) restated in the oracle: x * x^-1 = 1 against Python integers, zero -> the reference's Err."""
    from oracle import cref
    for fid, p in C.FIELDS.items():
        for n in (1, 2, 100, 5000):
            v = C.edge_vectors(fid, n, 40 + n)
            vi = C.ints(v)
            if 0 in vi:
                assert cref.batch_invert(fid, v, n) is None
                v = C.vec([x if x else 7 for x in vi])
                vi = C.ints(v)
            got = C.ints(bytearray(cref.batch_invert(fid, v, n)))
            assert got == [pow(x, -1, p) for x in vi]
        assert cref.batch_invert(fid, C.vec([3, 0, 5]), 3) is None


def test_parallel_paths_of_the_oracle_equal_its_serial_paths():
    """Round 6: the oracle's transposed product, evaluations, eq tables, batch witness and Horner / division pass run OpenMP above a size
    threshold (the reference's rayon code does: spartan/mod.rs:497-533, polys/multilinear.rs:98-180, hyperkzg.rs:961-999).  With one thread
    every one of them takes its serial loop -- the restatement line for line; with several the chunked / column-grouped form: same
    bytes, at sizes on both sides of the thresholds and with ragged chunk ends."""
    from tests import spartan_common as sp
    fid = 1
    keep = cref.get_threads()
    try:
        outs = []
        for th in (1, 5):
            cref.set_threads(th)
            o = []
            for n in (8191, 8192, 20011):
                f = C.rand_vec(fid, n, 50 + n)
                u = C.rand_vec(fid, 1, 51 + n)
                o.append(cref.suffix_horner(fid, f, n, u))
            ell = 14
            z = C.rand_vec(fid, 1 << ell, 60)
            r = C.rand_vec(fid, ell, 61)
            o.append(cref.mle_evaluate(fid, z, ell, r))
            o.append(b"".join(cref.mle_multi_evaluate(fid, [z.tobytes(), C.rand_vec(fid, 1 << ell, 62).tobytes()], ell, r)))
            o.append(cref.eq_evals(fid, r, ell))
            o.append(cref.lincomb_powers(fid, [z.tobytes(), C.rand_vec(fid, 5000, 63).tobytes()], u, 1 << ell))
            ip, ix, dt = sp.heavy_column_csr(fid, 6000, 2500, 7, heavy_cols=(0, 2499))
            x = C.rand_vec(fid, 6000, 64)
            o.append(cref.spmv_transposed(fid, ip, ix, dt, 6000, 2500, x))
            o.append(b"".join(cref.spmv_pair(fid, ip, ix, dt, 6000, C.rand_vec(fid, 2500, 65), C.rand_vec(fid, 2500, 66))))
            outs.append(o)
        assert outs[0] == outs[1]
    finally:
        cref.set_threads(keep)
