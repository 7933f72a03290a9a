"""-m gpu: the HIP field-vector kernels (nova_amd/csrc/fieldvec.hip) through the C ABI against the oracle and the
reference's known-answer tests; host and HBM-resident operands, canonical and Montgomery layouts, and the
commit_T flow (cross term -> MSM) with T never leaving the device."""
import numpy as np
import pytest

from oracle import cref
from oracle import pyref as R
from tests import fv_common as C
from tests import util

pytestmark = pytest.mark.gpu


def test_reference_kats(nmx):
    from nova_amd import fieldvec as fv
    C.check_kats(lambda fid, p, x: fv.fold_pairs(fid, p, x), lambda fid, z, r: fv.bind_poly_var_top(fid, z, r))


@pytest.mark.parametrize("fid", range(4))
@pytest.mark.parametrize("n", [2, 64, 1000, 1 << 16])
def test_field_vectors_vs_oracle(nmx, fid, n):
    from nova_amd import fieldvec as fv
    a, b, c, e = (C.edge_vectors(fid, n, s) for s in (1, 2, 3, 4))
    r = C.rand_vec(fid, 1, 9)
    assert fv.axpy(fid, a, b, r).tobytes() == cref.field_axpy(fid, a, b, r, n)
    assert fv.axpy2(fid, a, b, c, r).tobytes() == cref.field_axpy2(fid, a, b, c, r, n)
    assert fv.cross_term(fid, a, b, c, e, r).tobytes() == cref.field_cross_term(fid, a, b, c, e, r, n)
    e2 = C.edge_vectors(fid, n, 5)
    assert fv.cross_term2(fid, a, b, c, e, e2, r).tobytes() == cref.field_cross_term2(fid, a, b, c, e, e2, r, n)
    assert fv.vec_add(fid, a, b).tobytes() == cref.field_axpy(fid, a, b, util.int_to_le32(1), n)
    assert fv.bind_poly_var_top(fid, a, r).tobytes() == cref.field_bind(fid, a, 0, n // 2, 1, r, n // 2)
    assert fv.fold_pairs(fid, a, r).tobytes() == cref.field_bind(fid, a, 0, 1, 2, r, n // 2)
    for special in (0, 1, C.FIELDS[fid] - 1):  # r = 0, 1, -1
        rs = util.int_to_le32(special)
        assert fv.axpy(fid, a, b, rs).tobytes() == cref.field_axpy(fid, a, b, rs, n)


@pytest.mark.parametrize("fid", [1, 3])
def test_montgomery_layout_and_device_residency(nmx, fid):
    """The reference keeps vectors as R = 2^256 Montgomery limbs: same results after converting in and out."""
    import torch
    from nova_amd import fieldvec as fv
    p = C.FIELDS[fid]
    n = 4096
    Rm = 1 << 256
    to_m = lambda v: C.vec([x * Rm % p for x in C.ints(v)])
    from_m = lambda v: C.vec([x * pow(Rm, -1, p) % p for x in C.ints(v)])
    a, b, c, e = (C.edge_vectors(fid, n, s) for s in (1, 2, 3, 4))
    r = C.rand_vec(fid, 1, 9)
    got = fv.cross_term(fid, to_m(a), to_m(b), to_m(c), to_m(e), to_m(r), mont=True)
    assert from_m(got).tobytes() == cref.field_cross_term(fid, a, b, c, e, r, n)
    got = fv.cross_term2(fid, to_m(a), to_m(b), to_m(c), to_m(e), to_m(a), to_m(r), mont=True)
    assert from_m(got).tobytes() == cref.field_cross_term2(fid, a, b, c, e, a, r, n)
    got = fv.axpy2(fid, to_m(a), to_m(b), to_m(c), to_m(r), mont=True)
    assert from_m(got).tobytes() == cref.field_axpy2(fid, a, b, c, r, n)
    # HBM-resident operands and results; in-place bind like the reference's `*a += r * (*b - *a)`
    da, db = (torch.from_numpy(x.copy()).cuda() for x in (a, b))
    out = fv.axpy(fid, da, db, r)
    assert out.is_cuda and out.cpu().numpy().tobytes() == cref.field_axpy(fid, a, b, r, n)
    z = torch.from_numpy(a.copy()).cuda()
    bound = fv.bind_poly_var_top(fid, z, r, in_place=True)
    assert bound.cpu().numpy().tobytes() == cref.field_bind(fid, a, 0, n // 2, 1, r, n // 2)


def test_commit_T_flow_stays_on_device(nmx):
    """commit_T (src/r1cs/mod.rs:578-625): T = AZ o BZ - u*CZ - E, then CE::commit(ck, T, r_T), then the fold
    E' = E + r*T -- with T produced, committed and folded in HBM."""
    import torch
    from nova_amd import fieldvec as fv
    c = R.BN254_G1
    fid = fv.SCALAR_FIELD_OF_CURVE[c.cid]
    n = 1 << 14
    az, bz, cz, e = (C.rand_vec(fid, n, s) for s in (11, 12, 13, 14))
    u, r, r_T = (C.rand_vec(fid, 1, s) for s in (15, 16, 17))
    ce = nmx.CommitmentEngine(c.cid)
    ck = ce.setup_synthetic(n, k0=5)
    d = [torch.from_numpy(x.copy()).cuda() for x in (az, bz, cz, e)]
    T = fv.cross_term(fid, *d, u)
    com = ce.commit(ck, T, r_T)
    E2 = fv.axpy(fid, d[3], T, r)
    T_ref = cref.field_cross_term(fid, az, bz, cz, e, u, n)
    host_key = ck.read(0, n)
    assert T.cpu().numpy().tobytes() == T_ref
    assert (com.xy, int(com.is_inf)) == cref.commit(c.cid, T_ref, host_key, n, ck.h, r_T)
    assert E2.cpu().numpy().tobytes() == cref.field_axpy(fid, e, T_ref, r, n)
    ck.close()


@pytest.mark.parametrize("fid", range(4))
@pytest.mark.parametrize("logn", [1, 6, 12, 18])
def test_sumcheck_eq_sums(nmx, fid, logn):
    """evaluation_points_* N-scaling sums (sumcheck.rs:900-1075): all three modes, first- and last-half factor forms,
    host / device operands, canonical / Montgomery layouts."""
    import torch
    from nova_amd import fieldvec as fv
    p = C.FIELDS[fid]
    n = 1 << logn
    h = n // 2
    shift = max(0, (logn - 1) // 2)
    A, B, Cc = (C.edge_vectors(fid, n, s) for s in (1, 2, 3))
    eqR, eqL, eqF = C.rand_vec(fid, 1 << shift, 5), C.rand_vec(fid, max(1, h >> shift), 6), C.rand_vec(fid, h, 7)
    for mode in (1, 2, 3):
        exp1 = cref.sumcheck_eq_sums(fid, mode, A, B, Cc, n, eqR, eqL, shift)
        exp2 = cref.sumcheck_eq_sums(fid, mode, A, B, Cc, n, eqF)
        assert fv.sumcheck_eq_sums(fid, mode, A, B, Cc, eqR, eqL, shift) == exp1, (mode, "first half")
        assert fv.sumcheck_eq_sums(fid, mode, A, B, Cc, eqF) == exp2, (mode, "last half")
    if logn == 12:
        d = [torch.from_numpy(x.copy()).cuda() for x in (A, B, Cc, eqR, eqL)]
        assert fv.sumcheck_eq_sums(fid, 3, d[0], d[1], d[2], d[3], d[4], shift) == cref.sumcheck_eq_sums(fid, 3, A, B, Cc, n, eqR, eqL, shift)
        Rm = 1 << 256
        to_m = lambda v: C.vec([x * Rm % p for x in C.ints(v)])
        for mode in (1, 2, 3):
            got = fv.sumcheck_eq_sums(fid, mode, to_m(A), to_m(B), to_m(Cc), to_m(eqR), to_m(eqL), shift, mont=True)
            exp = cref.sumcheck_eq_sums(fid, mode, A, B, Cc, n, eqR, eqL, shift)
            assert tuple(C.ints(C.vec([int.from_bytes(g, "little") * pow(Rm, -1, p) % p]))[0] for g in got) == \
                tuple(int.from_bytes(e, "little") for e in exp), mode
            got = fv.sumcheck_eq_sums(fid, mode, to_m(A), to_m(B), to_m(Cc), to_m(eqF), mont=True)
            exp = cref.sumcheck_eq_sums(fid, mode, A, B, Cc, n, eqF)
            assert tuple(int.from_bytes(g, "little") * pow(Rm, -1, p) % p for g in got) == tuple(int.from_bytes(e, "little") for e in exp)


def test_reference_kats_eq_mle_spmv(nmx):
    from nova_amd import fieldvec as fv

    def spmv(fid, ip, ix, d, cols, z):
        m = fv.SparseMatrix(fid, ip, ix, d, cols)
        out = m.multiply_vec(z)
        m.close()
        return out

    C.check_kats2(lambda fid, r: fv.eq_evals_from_points(fid, r), lambda fid, z, r: fv.mle_evaluate(fid, z, r), spmv)


def test_reference_kats_multi_evaluate_and_the_mixed_coefficient_matrix(nmx):
    """multi_evaluate_with's known values (multilinear.rs:456-485) and the reference's mixed-coefficient SpMV fixture (sparse.rs:486-544)
    through the C ABI: host operands and HBM-resident ones (the mailbox path of the evaluations), multiply_vec, multiply_vec_pair, the
    many-matrix call and -- against the dense product of the transposed fixture -- the transposed product."""
    import torch
    from nova_amd import fieldvec as fv
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).copy()).cuda()

    def spmv(fid, ip, ix, d, cols, z):
        m = fv.SparseMatrix(fid, ip, ix, d, cols)
        out = m.multiply_vec(z)
        assert m.multiply_vec(dev(z)).cpu().numpy().tobytes() == out.tobytes()
        assert fv.multiply_vec_many([m, m], dev(z))[1].cpu().numpy().tobytes() == out.tobytes()
        m.close()
        return out

    def spmv_pair(fid, ip, ix, d, cols, z1, z2):
        m = fv.SparseMatrix(fid, ip, ix, d, cols)
        out = m.multiply_vec_pair(z1, z2)
        m.close()
        return out

    def multi(fid, zs, r):
        host = fv.mle_multi_evaluate(fid, zs, r)
        assert fv.mle_multi_evaluate(fid, [dev(z) for z in zs], r) == host
        assert [fv.mle_evaluate(fid, dev(z), r) for z in zs] == host
        return host
    C.check_kats3(multi, spmv, spmv_pair)
    for fid in C.FIELDS:                                     # M^T x on the same fixture: out[c] = sum_r M[r][c] x[r] (spartan/mod.rs:506-512)
        ip, ix, dt, cols, _z, _z2, _o, _o2 = C.mixed_coefficient_case(fid)
        k, p = C.KATS["spmv_mixed_coefficients"], C.FIELDS[fid]
        x = [3, 1, 4, 1, 5]
        want = [0] * cols
        for r, c, v in k["entries"]:
            want[c] = (want[c] + v * x[r]) % p
        m = fv.SparseMatrix(fid, ip, ix, dt, cols)
        assert C.ints(m.multiply_vec_transposed(C.vec(x))) == want
        assert C.ints(m.multiply_vec_transposed(dev(C.vec(x))).cpu().numpy()) == want
        m.close()


@pytest.mark.parametrize("fid", range(4))
def test_eq_mle_spmv_vs_oracle(nmx, fid):
    import torch
    from nova_amd import fieldvec as fv
    for ell in (0, 1, 5, 12, 17):
        r = C.rand_vec(fid, max(ell, 1), 3)[:ell]
        assert fv.eq_evals_from_points(fid, r).tobytes() == cref.eq_evals(fid, r, ell)
        z = C.edge_vectors(fid, 1 << ell, 5) if ell else C.rand_vec(fid, 1, 5)
        assert fv.mle_evaluate(fid, z, r) == cref.mle_evaluate(fid, z, ell, r)
        if ell == 12:
            assert fv.mle_evaluate(fid, torch.from_numpy(z.copy()).cuda(), r) == cref.mle_evaluate(fid, z, ell, r)
    for rows, cols in ((50, 30), (20000, 15000)):
        ip, ix, d = C.random_csr(fid, rows, cols, 9)
        z = C.rand_vec(fid, cols, 10)
        m = fv.SparseMatrix(fid, ip, ix, d, cols)
        exp = cref.spmv(fid, ip, ix, d, rows, z)
        assert m.multiply_vec(z).tobytes() == exp
        assert m.multiply_vec(torch.from_numpy(z.copy()).cuda()).cpu().numpy().tobytes() == exp
        m.close()


@pytest.mark.parametrize("fid", range(4))
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4096, 4097, 32767, 32768, 32775, 300000, 300003])
def test_horner_and_div_by_monomial(nmx, fid, n):
    """poly_eval + div_by_monomial (hyperkzg.rs:946-1020) as one suffix-Horner pass; chunk boundaries and recursion, both
    sides of the 2^15 threshold of the register-resident top level, lengths that are not multiples of its 8-element chunks."""
    import torch
    from nova_amd import fieldvec as fv
    f = C.edge_vectors(fid, n, 3) if n > 5 else C.rand_vec(fid, n, 3)
    u = C.rand_vec(fid, 1, 4)
    exp = cref.suffix_horner(fid, f, n, u)
    assert fv.suffix_horner(fid, f, u).tobytes() == exp
    assert fv.poly_eval(fid, f, u) == exp[:32]
    if n in (65, 32775, 300000):
        d = torch.from_numpy(f.copy()).cuda()
        assert fv.div_by_monomial(fid, d, u).cpu().numpy().tobytes() == exp[32:]
    for special in (0, 1):
        us = util.int_to_le32(special)
        assert fv.suffix_horner(fid, f, us).tobytes() == cref.suffix_horner(fid, f, n, us)


@pytest.mark.parametrize("fid", range(4))
@pytest.mark.parametrize("knob,value", [("horner_window", 1), ("horner_window", 3), ("horner_window", 7), ("horner_sub", 1),
                                        ("horner_sub", 2), ("horner_sub", 4), ("horner_top", 8), ("horner_top", 4), ("horner_top", 1),
                                        ("horner_order", 0), ("horner_order", 1)])
def test_horner_variants(nmx, fid, knob, value):
    """The single-pass scan (k_horner_scan) with 1 / 3 / 7 groups per look-back round -- every tile then needs several
    rounds, the path a 64-group window takes only when no group behind it has its inclusive value yet --, with 1 / 2 / 4
    sub-tiles per wave at every size, and the two-pass kernels it replaced (horner_top 8 / 4 / 1), all against the oracle:
    tile boundaries (512), partial and empty last sub-tiles, more than one group (64 tiles), edge values."""
    from nova_amd import _lib
    from nova_amd import fieldvec as fv
    L = _lib.lib()
    assert L.nmx_set_option(knob.encode(), value) == 0
    try:
        for n in (1024, 1025, 4097, 33 * 512, 65 * 512 + 9, 300003, 64 * 4 * 512 + 1):
            f = C.edge_vectors(fid, n, 5)
            u = C.rand_vec(fid, 1, 6)
            assert fv.suffix_horner(fid, f, u).tobytes() == cref.suffix_horner(fid, f, n, u), (knob, value, n)
    finally:
        assert L.nmx_set_option(knob.encode(), 64 if knob == "horner_window" else 1 if knob == "horner_order" else 0) == 0


@pytest.mark.parametrize("fid", range(4))
def test_horner_scan_extreme_values(nmx, fid):
    """The lazy-reduction bounds of the scan (sums of 64 weakly reduced terms, products of values up to ~100 p) on the worst
    inputs the field allows: every coefficient p - 1, u in {p - 1, (p - 1) / 2, 2}; and all zero.  1, 2 and 4 sub-tiles per wave."""
    from nova_amd import _lib
    from nova_amd import fieldvec as fv
    L = _lib.lib()
    p = C.FIELDS[fid]
    n = 70000  # 137 / 69 / 35 tiles: more than one group at one sub-tile per wave
    f = C.vec([p - 1] * n)
    z = C.vec([0] * n)
    try:
        for sub in (1, 2, 4):
            assert L.nmx_set_option(b"horner_sub", sub) == 0
            for uv in (p - 1, (p - 1) // 2, 2):
                u = C.vec([uv])
                assert fv.suffix_horner(fid, f, u).tobytes() == cref.suffix_horner(fid, f, n, u), (sub, uv)
            assert fv.suffix_horner(fid, z, C.vec([p - 1])).tobytes() == bytes(32 * n)
    finally:
        assert L.nmx_set_option(b"horner_sub", 0) == 0


def test_horner_scan_watchdog_falls_back_to_two_pass(nmx):
    """A look-back that gives up (forced: one poll allowed) must end in the right answer through the two-pass kernels, counted
    in nmx_stats -- not in a hang and not in a wrong quotient."""
    from nova_amd import _lib
    from nova_amd import fieldvec as fv
    L = _lib.lib()
    fid, n = 1, 300003
    f = C.edge_vectors(fid, n, 21)
    u = C.rand_vec(fid, 1, 22)
    exp = cref.suffix_horner(fid, f, n, u)
    before = _lib.stats()[_lib.STAT_SCAN_TIMEOUTS]
    assert L.nmx_set_option(b"horner_spin_limit", 1) == 0
    try:
        for _ in range(3):
            assert fv.suffix_horner(fid, f, u).tobytes() == exp
    finally:
        assert L.nmx_set_option(b"horner_spin_limit", 0) == 0
    assert fv.suffix_horner(fid, f, u).tobytes() == exp
    assert _lib.stats()[_lib.STAT_SCAN_TIMEOUTS] >= before  # (a lease where no wave ever had to poll twice counts none)


def test_horner_rejects_overlap_and_survives_a_timeout_on_device_vectors(nmx):
    """the time-out fallback re-reads f after the aborted scan has written to out, so an in-place call could
    return a wrong quotient.  In-place and partially overlapping device vectors are an NMX_E_ARG error (the header says so);
    a forced time-out on disjoint DEVICE vectors still ends in the oracle's answer and leaves f untouched."""
    import ctypes
    import torch
    from nova_amd import _lib
    from nova_amd import fieldvec as fv
    L = _lib.lib()
    fid, n = 1, 200000
    f = C.edge_vectors(fid, n, 31)
    u = C.rand_vec(fid, 1, 32)
    exp = cref.suffix_horner(fid, f, n, u)
    buf = torch.zeros((2 * n, 32), dtype=torch.uint8, device="cuda")
    buf[:n] = torch.from_numpy(f)
    torch.cuda.synchronize()
    uu = np.ascontiguousarray(u)
    flags = _lib.SCALARS_DEVICE
    base = buf.data_ptr()
    for off in (0, 32, 32 * (n - 101), -32 * 5):        # in place, shifted by one element, one-element overlap at the end, shifted back
        rc = L.nmx_poly_suffix_horner(fid, base + 32 * 100, n - 100, uu.ctypes.data, flags, base + 32 * 100 + off)
        assert rc == _lib.E_ARG, off
    assert L.nmx_set_option(b"horner_spin_limit", 1) == 0
    try:
        for _ in range(2):
            assert L.nmx_poly_suffix_horner(fid, base, n, uu.ctypes.data, flags, base + 32 * n) == 0   # adjacent, disjoint
            torch.cuda.synchronize()
            assert buf[n:].cpu().numpy().tobytes() == exp
            assert buf[:n].cpu().numpy().tobytes() == f.tobytes()
    finally:
        assert L.nmx_set_option(b"horner_spin_limit", 0) == 0


@pytest.mark.parametrize("fid", range(4))
def test_horner_outputs_are_canonical_for_any_input_words(nmx, fid):
    """the stored copy went through canon4 (valid below 4 p); a coefficient word >= p (any 256-bit value is accepted,
    as by the other field kernels) left a residue that was congruent but not canonical -- a later MSM over it would fail
    with NMX_E_SCALAR_RANGE.  Words >= p are reduced before they enter the chain: every output is the canonical residue."""
    from nova_amd import fieldvec as fv
    p = C.FIELDS[fid]
    n = 5000
    rng = np.random.Generator(np.random.PCG64(77 + fid))
    vals = [int.from_bytes(rng.bytes(32), "little") for _ in range(n)]
    vals[0], vals[1], vals[n - 1], vals[513] = (1 << 256) - 1, p, p + 1, 2 * p + 5
    raw = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(n, 32).copy()
    red = C.vec([v % p for v in vals])
    u = C.rand_vec(fid, 1, 6)
    exp = cref.suffix_horner(fid, red, n, u)
    got = fv.suffix_horner(fid, raw, u)
    assert got.tobytes() == exp
    assert all(int.from_bytes(got[i].tobytes(), "little") < p for i in range(n))


@pytest.mark.parametrize("fid", [0, 2])
def test_horner_scan_many_tiles_and_repeat(nmx, fid):
    """2^20 + 77 coefficients (2049 tiles) device resident, five calls in a row on the same context (the tile states are
    cleared by every call) -- all equal to the oracle; and 5000 coefficients in the reference's Montgomery layout."""
    import torch
    from nova_amd import fieldvec as fv
    p = C.FIELDS[fid]
    Rm = 1 << 256
    to_m = lambda v: C.vec([x * Rm % p for x in C.ints(v)])
    from_m = lambda v: C.vec([x * pow(Rm, -1, p) % p for x in C.ints(v)])
    g = C.edge_vectors(fid, 5000, 13)
    v = C.rand_vec(fid, 1, 14)
    assert from_m(fv.suffix_horner(fid, to_m(g), to_m(v), mont=True)).tobytes() == cref.suffix_horner(fid, g, 5000, v)
    n = (1 << 20) + 77
    f = C.rand_vec(fid, n, 11)
    u = C.rand_vec(fid, 1, 12)
    exp = cref.suffix_horner(fid, f, n, u)
    d = torch.from_numpy(f.copy()).cuda()
    for _ in range(5):
        assert fv.suffix_horner(fid, d, u).cpu().numpy().tobytes() == exp


@pytest.mark.parametrize("fid", range(4))
@pytest.mark.parametrize("logn", [1, 6, 13, 18])
def test_sumcheck_plain_sums(nmx, fid, logn):
    """compute_eval_points_{quad_prod, linear, quadratic, cubic} (sumcheck.rs:163-186, 353-443): host / device
    operands, canonical / Montgomery layouts."""
    import torch
    from nova_amd import fieldvec as fv
    p = C.FIELDS[fid]
    n = 1 << logn
    A, B, Cc = (C.edge_vectors(fid, n, s) for s in (1, 2, 3))
    for kind in (1, 2, 3, 4):
        exp = cref.sumcheck_plain_sums(fid, kind, A, B, Cc if kind == 4 else None, n)
        exp = exp if kind == 4 else exp[:2]
        assert fv.sumcheck_plain_sums(fid, kind, A, B, Cc if kind == 4 else None) == exp, kind
        if logn == 13:
            d = [torch.from_numpy(x.copy()).cuda() for x in (A, B, Cc)]
            assert fv.sumcheck_plain_sums(fid, kind, d[0], d[1], d[2] if kind == 4 else None) == exp, kind
            Rm = 1 << 256
            to_m = lambda v: C.vec([x * Rm % p for x in C.ints(v)])
            got = fv.sumcheck_plain_sums(fid, kind, to_m(A), to_m(B), to_m(Cc) if kind == 4 else None, mont=True)
            assert tuple(int.from_bytes(g, "little") * pow(Rm, -1, p) % p for g in got) == \
                tuple(int.from_bytes(e, "little") for e in exp), kind
    # worst-case operands for the lazy bounds: all p-1 against all 0 / all p-1
    hi, lo = C.vec([p - 1] * n), C.vec([0] * n)
    mix = np.concatenate([hi[: n // 2], lo[: n // 2]]) if n > 1 else hi
    for kind in (1, 2, 3, 4):
        for X, Y, Z in ((hi, hi, hi), (mix, lo, mix), (lo, mix, hi), (mix, mix, mix)):
            exp = cref.sumcheck_plain_sums(fid, kind, X, Y, Z if kind == 4 else None, n)
            assert fv.sumcheck_plain_sums(fid, kind, X, Y, Z if kind == 4 else None) == (exp if kind == 4 else exp[:2])


@pytest.mark.parametrize("fid", range(4))
def test_lincomb_multi_evaluate_spmv_pair(nmx, fid):
    """PolyEvalWitness::batch / batch_diff_size (spartan/mod.rs:165-277), multi_evaluate_with (multilinear.rs:131-180),
    multiply_vec_pair (sparse.rs:215-229)."""
    import torch
    from nova_amd import fieldvec as fv
    p = C.FIELDS[fid]
    s = C.rand_vec(fid, 1, 77)
    for lens in ((64, 17, 64, 1, 0), (1 << 16, 1 << 15, 1 << 16, 1000, 1 << 16, 3, 1 << 14, 1 << 16), (5,)):
        vecs = [C.edge_vectors(fid, m, 10 + j) if m > 8 else C.rand_vec(fid, max(m, 1), 10 + j)[:m] for j, m in enumerate(lens)]
        exp = cref.lincomb_powers(fid, [v.tobytes() for v in vecs], s, max(lens))
        assert fv.lincomb_powers(fid, vecs, s).tobytes() == exp, lens
        if 0 not in lens:
            d = [torch.from_numpy(v.copy()).cuda() for v in vecs]
            out = fv.lincomb_powers(fid, d, s)
            assert out.is_cuda and out.cpu().numpy().tobytes() == exp
    # s = 1 and s = 0 (the reference special-cases coefficient ONE)
    vecs = [C.edge_vectors(fid, 300, 1), C.edge_vectors(fid, 300, 2)]
    for sv in (0, 1, p - 1):
        assert fv.lincomb_powers(fid, vecs, C.vec([sv])).tobytes() == cref.lincomb_powers(fid, [v.tobytes() for v in vecs], C.vec([sv]), 300)
    Rm = 1 << 256
    to_m = lambda v: C.vec([x * Rm % p for x in C.ints(v)])
    got = fv.lincomb_powers(fid, [to_m(v) for v in vecs], to_m(s), mont=True)
    assert C.vec([x * pow(Rm, -1, p) % p for x in C.ints(got)]).tobytes() == cref.lincomb_powers(fid, [v.tobytes() for v in vecs], s, 300)

    # the reference's stored known-answer case (multilinear.rs:456-485)
    z1, z2 = C.vec([0, 0, 0, 1, 0, 1, 0, 2]), C.vec([5] * 8)
    assert [int.from_bytes(g, "little") for g in fv.mle_multi_evaluate(fid, [z1, z2], C.vec([1, 1, 1]))] == [2, 5]
    assert fv.mle_multi_evaluate(fid, [], C.vec([1, 1, 1])) == []
    for ell in (0, 1, 6, 13):
        r = C.rand_vec(fid, max(ell, 1), 3)[:ell]
        zs = [C.edge_vectors(fid, 1 << ell, 20 + j) if ell > 2 else C.rand_vec(fid, 1 << ell, 20 + j) for j in range(3)]
        exp = cref.mle_multi_evaluate(fid, [z.tobytes() for z in zs], ell, r)
        assert fv.mle_multi_evaluate(fid, zs, r) == exp
        assert exp == [fv.mle_evaluate(fid, z, r) for z in zs]  # multilinear.rs:414-439
        if ell == 13:
            assert fv.mle_multi_evaluate(fid, [torch.from_numpy(z.copy()).cuda() for z in zs], r) == exp

    for rows, cols in ((50, 30), (20000, 15000)):
        ip, ix, d = C.random_csr(fid, rows, cols, 9)
        z1, z2 = C.rand_vec(fid, cols, 10), C.edge_vectors(fid, cols, 11)
        m = fv.SparseMatrix(fid, ip, ix, d, cols)
        e1, e2 = cref.spmv_pair(fid, ip, ix, d, rows, z1, z2)
        o1, o2 = m.multiply_vec_pair(z1, z2)
        assert (o1.tobytes(), o2.tobytes()) == (e1, e2)
        o1, o2 = m.multiply_vec_pair(torch.from_numpy(z1.copy()).cuda(), torch.from_numpy(z2.copy()).cuda())
        assert (o1.cpu().numpy().tobytes(), o2.cpu().numpy().tobytes()) == (e1, e2)
        assert m.multiply_vec(z1).tobytes() == e1
        m.close()


def test_concurrent_callers_mixed_entry_points(nmx):
    """The reference calls these loops from rayon workers: every entry point must be re-entrant (one context --
    stream, workspace, aux arena -- per in-flight call).  Six threads interleave MSMs and field-vector calls."""
    import threading
    import nova_amd
    from nova_amd import fieldvec as fv
    fid, cid = 1, 0
    n, ell = 1 << 13, 13
    a, b = C.edge_vectors(fid, n, 1), C.edge_vectors(fid, n, 2)
    r = C.rand_vec(fid, 1, 9)
    pt = C.rand_vec(fid, ell, 3)
    ip, ix, d = C.random_csr(fid, 3000, n, 9)
    mat = fv.SparseMatrix(fid, ip, ix, d, n)
    ck = nova_amd.CommitmentKey.generate(cid, n, k0=3)
    key = ck.read(0, n)
    g = nova_amd.DlogGroup(cid)
    exp = {
        "axpy": cref.field_axpy(fid, a, b, r, n),
        "mle": cref.mle_evaluate(fid, a, ell, pt),
        "horner": cref.suffix_horner(fid, b, n, r),
        "spmv": cref.spmv(fid, ip, ix, d, 3000, a),
        "lin": cref.lincomb_powers(fid, [a.tobytes(), b.tobytes(), a[:100].tobytes()], r, n),
        "sums": cref.sumcheck_plain_sums(fid, 4, a, b, a, n),
        "msm": cref.msm(cid, a, key, n),
    }
    calls = {
        "axpy": lambda: fv.axpy(fid, a, b, r).tobytes(),
        "mle": lambda: fv.mle_evaluate(fid, a, pt),
        "horner": lambda: fv.suffix_horner(fid, b, r).tobytes(),
        "spmv": lambda: mat.multiply_vec(a).tobytes(),
        "lin": lambda: fv.lincomb_powers(fid, [a, b, a[:100]], r).tobytes(),
        "sums": lambda: fv.sumcheck_plain_sums(fid, 4, a, b, a),
        "msm": lambda: (lambda p: (p.xy, int(p.is_inf)))(g.vartime_multiscalar_mul(a, ck)),
    }
    errors = []

    def worker(seed):
        names = list(calls)
        for it in range(12):
            nm = names[(seed * 5 + it * 3) % len(names)]
            try:
                if calls[nm]() != exp[nm]:
                    errors.append((seed, it, nm, "mismatch"))
            except Exception as e:  # noqa: BLE001
                errors.append((seed, it, nm, repr(e)))

    ts = [threading.Thread(target=worker, args=(s,)) for s in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    mat.close()
    ck.close()
    assert not errors, errors[:5]


@pytest.mark.parametrize("fid", range(4))
@pytest.mark.parametrize("logn", [2, 7, 13, 18])
def test_sumcheck_fused_round(nmx, fid, logn):
    """nmx_sumcheck_bind_eq_sums == bind_poly_var_top on A, B, C (multilinear.rs:65-84) followed by the next round's
    evaluation_points_* sums (sumcheck.rs:900-1075): bound tables byte-equal, sums equal, all three modes, first- and
    last-half eq forms, Montgomery layout; then a whole sum-check (every round fused) against the oracle's rounds."""
    import torch
    from nova_amd import fieldvec as fv
    p = C.FIELDS[fid]
    n = 1 << logn
    hq = n // 4
    shift = max(0, (logn - 2) // 2)
    A, B, Cc = (C.edge_vectors(fid, n, s) if n > 8 else C.rand_vec(fid, n, s) for s in (1, 2, 3))
    r = C.rand_vec(fid, 1, 9)
    eqR, eqL, eqF = C.rand_vec(fid, 1 << shift, 5), C.rand_vec(fid, max(1, hq >> shift), 6), C.rand_vec(fid, hq, 7)
    bound = [cref.field_bind(fid, X, 0, n // 2, 1, r, n // 2) for X in (A, B, Cc)]
    for mode in (1, 2, 3):
        for el, er, sh in ((eqL, eqR, shift), (None, eqF, 0)):
            d = [torch.from_numpy(X.copy()).cuda() for X in (A, B, Cc)]
            args = [d[0], d[1] if mode >= 2 else None, d[2] if mode == 3 else None]
            oa, ob, oc, sums = fv.sumcheck_bind_eq_sums(fid, mode, args[0], args[1], args[2], r,
                                                        torch.from_numpy(er).cuda(), None if el is None else torch.from_numpy(el).cuda(), sh)
            outs = [oa, ob, oc]
            for j in range(mode):
                assert outs[j].cpu().numpy().tobytes() == bound[j], (mode, j)
            exp = cref.sumcheck_eq_sums(fid, mode, bound[0], bound[1], bound[2], n // 2, er, el, sh)
            assert sums == exp, (mode, el is None)
    if logn == 13:
        Rm = 1 << 256
        to_m = lambda v: C.vec([x * Rm % p for x in C.ints(v)])
        d = [torch.from_numpy(to_m(X)).cuda() for X in (A, B, Cc)]
        oa, ob, oc, sums = fv.sumcheck_bind_eq_sums(fid, 3, d[0], d[1], d[2], to_m(r), torch.from_numpy(to_m(eqR)).cuda(),
                                                    torch.from_numpy(to_m(eqL)).cuda(), shift, mont=True)
        assert C.vec([x * pow(Rm, -1, p) % p for x in C.ints(oa.cpu().numpy())]).tobytes() == bound[0]
        exp = cref.sumcheck_eq_sums(fid, 3, bound[0], bound[1], bound[2], n // 2, eqR, eqL, shift)
        assert tuple(int.from_bytes(g, "little") * pow(Rm, -1, p) % p for g in sums) == tuple(int.from_bytes(e, "little") for e in exp)
    if logn == 7:
        # every round of a sum-check fused: tables shrink n -> 2, sums of each round equal the oracle's on its own tables
        d = [torch.from_numpy(X.copy()).cuda() for X in (A, B, Cc)]
        host = [A, B, Cc]
        m = n
        rnd = 0
        while m >= 4:
            rr = C.rand_vec(fid, 1, 100 + rnd)
            eq = C.rand_vec(fid, m // 4, 200 + rnd)
            d[0], d[1], d[2], sums = fv.sumcheck_bind_eq_sums(fid, 3, d[0], d[1], d[2], rr, torch.from_numpy(eq).cuda())
            host = [np.frombuffer(cref.field_bind(fid, X, 0, m // 2, 1, rr, m // 2), np.uint8).reshape(-1, 32) for X in host]
            assert sums == cref.sumcheck_eq_sums(fid, 3, host[0], host[1], host[2], m // 2, eq)
            d = [x.contiguous() for x in d]
            m //= 2
            rnd += 1
        assert [x.cpu().numpy().tobytes() for x in d] == [h.tobytes() for h in host]


@pytest.mark.parametrize("fid", range(4))
def test_poly_eval_multi(nmx, fid):
    """nmx_poly_eval_multi == poly_eval per (polynomial, point) (hyperkzg.rs:1011-1020): the HyperKZG shape (lengths
    n, n/2, ..., 2 at three points), ragged / empty / single-coefficient polynomials, lengths around the 16-coefficient
    chunk and the 4096-coefficient block, special points 0, 1, -1, device operands, Montgomery layout."""
    import torch
    from nova_amd import fieldvec as fv
    p = C.FIELDS[fid]
    pts = np.concatenate([C.rand_vec(fid, 2, 5), C.vec([p - 1])])
    polys = [C.edge_vectors(fid, 1 << e, 30 + e) if e > 3 else C.rand_vec(fid, 1 << e, 30 + e) for e in range(14, 0, -1)]
    exp = [[cref.suffix_horner(fid, f, len(f), pts[j])[:32] for j in range(3)] for f in polys]
    assert fv.poly_eval_multi(fid, polys, pts) == exp
    assert fv.poly_eval_multi(fid, [torch.from_numpy(f.copy()).cuda() for f in polys], pts) == exp
    odd = [C.rand_vec(fid, m, 50 + m) for m in (1, 15, 16, 17, 4095, 4096, 4097, 70001)] + [C.rand_vec(fid, 1, 1)[:0]]
    sp = np.concatenate([C.vec([0]), C.vec([1]), C.rand_vec(fid, 1, 8), C.vec([2])])
    exp = [[cref.suffix_horner(fid, f, len(f), sp[j])[:32] if len(f) else bytes(32) for j in range(4)] for f in odd]
    assert fv.poly_eval_multi(fid, odd, sp) == exp
    assert fv.poly_eval_multi(fid, odd[:3], sp[:1]) == [e[:1] for e in exp[:3]]
    Rm = 1 << 256
    to_m = lambda v: C.vec([x * Rm % p for x in C.ints(v)])
    got = fv.poly_eval_multi(fid, [to_m(f) for f in polys[6:]], to_m(pts), mont=True)
    want = [[cref.suffix_horner(fid, f, len(f), pts[j])[:32] for j in range(3)] for f in polys[6:]]
    assert [[(int.from_bytes(g, "little") * pow(Rm, -1, p) % p).to_bytes(32, "little") for g in row] for row in got] == want


@pytest.mark.parametrize("k", [1, 2])
def test_async_nifs_chain_is_ordered_before_the_commitment(nmx, k):
    """NMX_ASYNC: Z = W1 + W2, AZ / BZ / CZ, T = AZ o BZ - u CZ - E (src/r1cs/mod.rs:590-620) enqueued without a host wait
    between them, then commit(T) -- synchronous, ordered behind the chain on whatever context (and, over a sharded key, on
    whatever helper threads) it runs -- and the folds (1044-1107) completed by nmx_sync.  Same bytes as the synchronous calls."""
    import torch
    from nova_amd import _lib
    from nova_amd import fieldvec as fv
    L = _lib.lib()
    cid, fid, n = 0, fv.SCALAR_FIELD_OF_CURVE[0], 60000
    assert nmx.init_devices(k, oversubscribe=True) == k
    assert L.nmx_set_option(b"shard_min_n", 1000) == 0
    try:
        ck = nmx.CommitmentKey.generate(cid, n, k0=5)
        ce = nmx.CommitmentEngine(cid)
        hW1, hW2, hE = (C.rand_vec(fid, n, s) for s in (41, 42, 43))
        u, r = C.rand_vec(fid, 1, 44), C.rand_vec(fid, 1, 45)
        mats = [fv.SparseMatrix(fid, *C.random_csr(fid, n, n, 50 + j), n) for j in range(3)]
        W1, W2, E = (torch.from_numpy(h.copy()).cuda() for h in (hW1, hW2, hE))
        torch.cuda.synchronize()
        # synchronous reference run
        Z0 = fv.vec_add(fid, W1, W2)
        P0 = [m.multiply_vec(Z0) for m in mats]
        T0 = fv.cross_term(fid, P0[0], P0[1], P0[2], E, u)
        c0 = ce.commit(ck, T0, r)
        F0 = fv.axpy(fid, E, T0, r)
        for _ in range(3):
            Z = fv.vec_add(fid, W1, W2, async_=True)
            P = [m.multiply_vec(Z, async_=True) for m in mats]
            T = fv.cross_term(fid, P[0], P[1], P[2], E, u, async_=True)
            c1 = ce.commit(ck, T, r)                       # synchronous: everything before it is complete on return
            assert (c1.xy, c1.is_inf) == (c0.xy, c0.is_inf)
            assert torch.equal(T, T0) and torch.equal(Z, Z0)
            F = fv.axpy(fid, E, T, r, async_=True)
            fv.sync()
            assert torch.equal(F, F0)
        # the compound calls: commit_T's chain and the fold, one call each, synchronous and stream-ordered
        hT = fv.r1cs_cross_term(mats[0], mats[1], mats[2], W1, W2, E, u)
        assert torch.equal(hT, T0)
        assert torch.equal(fv.r1cs_cross_term(mats[0], mats[1], mats[2], Z0, None, E, u), T0)
        for _ in range(2):
            T = fv.r1cs_cross_term(mats[0], mats[1], mats[2], W1, W2, E, u, async_=True)
            c2 = ce.commit(ck, T, r)
            assert (c2.xy, c2.is_inf) == (c0.xy, c0.is_inf) and torch.equal(T, T0)
            Wf, Ef = fv.nifs_fold(fid, W1, W2, E, T, r, async_=True)
            fv.sync()
            assert torch.equal(Ef, F0) and torch.equal(Wf, fv.axpy(fid, W1, W2, r))
        for m in mats:
            m.close()
        ck.close()
    finally:
        assert nmx.init_devices(1) == 1
        assert L.nmx_set_option(b"shard_min_n", 1 << 20) == 0


def _nonzero(fid, n, seed):
    v = C.edge_vectors(fid, n, seed)
    return C.vec([x if x else 5 for x in C.ints(v)]).copy()


@pytest.mark.parametrize("fid", range(4))
@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 128, 129, 1000, 1025, 4096, 4097, (1 << 15) + 5, 131073, 300001, (1 << 21) + 5])
def test_batch_invert_vs_oracle(nmx, fid, n):
    """batch_invert (src/spartan/mod.rs:54-152): host and HBM-resident vectors, every level count (n <= 128 is the host level alone;
    ragged last chunks; 2^15 + 5 is the size of the reference's own test_batch_invert, spartan/mod.rs:547-559; 2^21 + 5 starts with
    16-element chunks and continues with 8-element ones)."""
    import torch
    from nova_amd import fieldvec as fv
    v = _nonzero(fid, n, 70 + n % 13)
    want = cref.batch_invert(fid, v, n)
    assert fv.batch_invert(fid, v).tobytes() == want
    dv = torch.from_numpy(v.copy()).cuda()
    got = fv.batch_invert(fid, dv)
    assert got.is_cuda and got.cpu().numpy().tobytes() == want
    assert dv.cpu().numpy().tobytes() == v.tobytes()  # the input is left alone


@pytest.mark.parametrize("fid", [1, 3])
@pytest.mark.parametrize("n", [100, 5000])
def test_batch_invert_montgomery_layout(nmx, fid, n):
    from nova_amd import fieldvec as fv
    p = C.FIELDS[fid]
    Rm = 1 << 256
    v = _nonzero(fid, n, 3)
    vm = C.vec([x * Rm % p for x in C.ints(v)])
    got = fv.batch_invert(fid, vm, mont=True)
    assert [x * pow(Rm, -1, p) % p for x in C.ints(got)] == C.ints(bytearray(cref.batch_invert(fid, v, n)))


@pytest.mark.parametrize("n,at", [(1, 0), (77, 76), (129, 0), (5000, 4999), (200000, 123457)])
def test_batch_invert_reports_a_zero_element(nmx, n, at):
    """the reference returns Err(NovaError::InternalError) when any element is zero (spartan/mod.rs:103-105)"""
    import torch
    from nova_amd import NmxError
    from nova_amd import _lib as L
    from nova_amd import fieldvec as fv
    v = _nonzero(1, n, 11)
    v[at] = 0
    assert cref.batch_invert(1, v, n) is None
    for operand in (v, torch.from_numpy(v.copy()).cuda()):
        with pytest.raises(NmxError) as e:
            fv.batch_invert(1, operand)
        assert e.value.code == L.E_ZERO
    # and the library is usable afterwards
    v[at] = 9
    assert fv.batch_invert(1, v).tobytes() == cref.batch_invert(1, v, n)


def test_batch_invert_rejects_in_place(nmx):
    import torch
    from nova_amd import _lib as L
    from nova_amd import fieldvec as fv
    v = torch.from_numpy(_nonzero(1, 64, 1).copy()).cuda()
    rc = L.lib().nmx_field_batch_invert(1, v.data_ptr(), 64, L.SCALARS_DEVICE, v.data_ptr())
    assert rc == L.E_ARG


@pytest.mark.parametrize("fid", range(4))
def test_fold_chain_equals_fold_by_fold(nmx, fid):
    """nmx_poly_fold_chain (the loop of hyperkzg.rs:1085-1095 as one call: long folds one launch each, the folds of <= 2048 inputs inside
    one block through LDS) against the oracle's pair fold applied fold by fold -- lengths on both sides of the 2048-input switch, the
    full chain down to 2 elements and a partial one, HBM-resident (synchronous and stream-ordered) and host operands, and the reference's
    known answers (tests/golden/field_kats.json hyperkzg_fold_eval) through the chain."""
    import torch
    from nova_amd import fieldvec as fv
    for ell, k in ((1, 1), (2, 1), (2, 2), (6, 5), (11, 10), (12, 11), (13, 12), (13, 4), (15, 14), (16, 3)):
        n = 1 << ell
        P = C.edge_vectors(fid, n, 70 + ell)
        xs = C.rand_vec(fid, k, 71 + ell)
        want, cur = [], P
        for i in range(k):
            m = len(cur) // 2
            cur = np.frombuffer(cref.field_bind(fid, cur, 0, 1, 2, xs[i:i + 1], m), np.uint8).reshape(m, 32)
            want.append(cur.tobytes())
        d = torch.from_numpy(P.copy()).cuda()
        assert [o.cpu().numpy().tobytes() for o in fv.fold_chain(fid, d, xs)] == want
        outs = fv.fold_chain(fid, d, xs, async_=True)
        fv.sync()
        assert [o.cpu().numpy().tobytes() for o in outs] == want
        assert [o.tobytes() for o in fv.fold_chain(fid, P, xs)] == want                   # host vectors
        assert d.cpu().numpy().tobytes() == P.tobytes()                                    # the input is left alone
    for case in C.KATS["hyperkzg_fold_eval"]["cases"]:                                     # evaluation = the chain's last element
        poly, pt = C.vec(case["poly"]), C.vec(list(reversed(case["point"])))
        out = fv.fold_chain(fid, torch.from_numpy(poly.copy()).cuda(), pt)
        assert C.ints(out[-1].cpu().numpy()) == [case["eval"]]


@pytest.mark.parametrize("fid", [1, 2])
def test_montgomery_layout_through_the_round6_paths(nmx, fid):
    """halo2curves' in-memory layout (R = 2^256 Montgomery limbs, NMX_SCALARS_MONT) through the entry points and code paths round 6 added:
    evaluations of HBM-resident polynomials (the mailbox path), the fold chain, the many-matrix product (forward and transposed), the
    suffix Horner scan with its host-built constants, the evaluation matrix with its host-built tables -- each equal to the canonical
    computation converted in and out."""
    import torch
    from nova_amd import fieldvec as fv
    p = C.FIELDS[fid]
    Rm = 1 << 256
    to_m = lambda v: C.vec([x * Rm % p for x in C.ints(v)])
    un_m = lambda b: b"".join(int(x * pow(Rm, -1, p) % p).to_bytes(32, "little") for x in C.ints(np.frombuffer(bytes(b), np.uint8)))
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v).copy()).cuda()
    ell = 11
    n = 1 << ell
    zs = [C.edge_vectors(fid, n, 400 + i) for i in range(3)]
    r = C.rand_vec(fid, ell, 410)
    want = cref.mle_multi_evaluate(fid, [z.tobytes() for z in zs], ell, r)
    got = fv.mle_multi_evaluate(fid, [dev(to_m(z)) for z in zs], to_m(r), mont=True)
    assert [un_m(g) for g in got] == want
    assert un_m(fv.mle_evaluate(fid, dev(to_m(zs[0])), to_m(r), mont=True)) == want[0]
    # fold chain: 2^13 -> 2 (across the one-block switch)
    P = C.edge_vectors(fid, 1 << 13, 420)
    xs = C.rand_vec(fid, 12, 421)
    cur, want_f = P, []
    for i in range(12):
        m = len(cur) // 2
        cur = np.frombuffer(cref.field_bind(fid, cur, 0, 1, 2, xs[i:i + 1], m), np.uint8).reshape(m, 32)
        want_f.append(cur.tobytes())
    assert [un_m(o.cpu().numpy().tobytes()) for o in fv.fold_chain(fid, dev(to_m(P)), to_m(xs), mont=True)] == want_f
    # the many-matrix product, both directions (matrix registered from Montgomery coefficients)
    ip, ix, dt = C.random_csr(fid, 700, 500, 430)
    mats = [fv.SparseMatrix(fid, ip, ix, to_m(dt), 500, mont=True) for _ in range(2)]
    z, x = C.edge_vectors(fid, 500, 431), C.edge_vectors(fid, 700, 432)
    assert [un_m(o.cpu().numpy().tobytes()) for o in fv.multiply_vec_many(mats, dev(to_m(z)), mont=True)] == [cref.spmv(fid, ip, ix, dt, 700, z)] * 2
    got_t = [un_m(o.cpu().numpy().tobytes()) for o in fv.multiply_vec_many(mats, dev(to_m(x)), transposed=True, mont=True)]
    assert got_t == [cref.spmv_transposed(fid, ip, ix, dt, 700, 500, x)] * 2
    for m_ in mats:
        m_.close()
    # suffix Horner (single-pass scan from 1024 coefficients on) and the evaluation matrix
    f = C.edge_vectors(fid, 40000, 440)
    u = C.rand_vec(fid, 1, 441)
    assert un_m(fv.suffix_horner(fid, dev(to_m(f)), to_m(u), mont=True).cpu().numpy().tobytes()) == cref.suffix_horner(fid, f, 40000, u)
    polys = [C.edge_vectors(fid, m_, 450 + m_) for m_ in (40000, 4097, 2)]
    pts = C.rand_vec(fid, 3, 451)
    got = fv.poly_eval_multi(fid, [dev(to_m(q)) for q in polys], to_m(pts), mont=True)
    assert [[un_m(v) for v in row] for row in got] == [[cref.suffix_horner(fid, q, len(q), pts[j:j + 1])[:32] for j in range(3)] for q in polys]
