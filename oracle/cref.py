"""ctypes loader for oracle/libnova_ref.so (tier-2 oracle, oracle/nova_ref.c).

TEST INFRASTRUCTURE ONLY -- see the header of nova_ref.c.  Builds the library on first use if missing.
"""
import ctypes
import os
import subprocess

import numpy as np

# the transcript's side of a sum-check round: (ctx, round polynomial coefficients, how many, challenge out) -> 0
TRANSCRIPT_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t,
                                 ctypes.POINTER(ctypes.c_uint8))
# the transcript's side of an inner-product-argument round: (ctx, L xy64, L is the identity, R xy64, R is the identity, challenge out) -> 0
IPA_TRANSCRIPT_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_uint8), ctypes.c_int, ctypes.POINTER(ctypes.c_uint8))

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnova_ref.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "nova_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnova_ref.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, sz, u8p, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint64
        L.ref_set_threads.argtypes = [ctypes.c_int]
        L.ref_get_threads.restype = ctypes.c_int
        L.ref_bases_load.restype = vp
        L.ref_bases_load.argtypes = [ctypes.c_int, vp, sz]
        L.ref_bases_free.argtypes = [vp]
        L.ref_msm_prepared.argtypes = [ctypes.c_int, vp, vp, sz, vp, vp]
        L.ref_msm_best_prepared.argtypes = [ctypes.c_int, vp, vp, sz, vp, vp]
        L.ref_msm.argtypes = [ctypes.c_int, vp, vp, sz, vp, vp]
        L.ref_msm_u64_prepared.argtypes = [ctypes.c_int, vp, vp, sz, sz, vp, vp]
        L.ref_msm_u64.argtypes = [ctypes.c_int, vp, vp, sz, sz, vp, vp]
        L.ref_msm_batch.argtypes = [ctypes.c_int, vp, vp, sz, vp, sz, vp, vp]
        L.ref_commit.argtypes = [ctypes.c_int, vp, vp, sz, vp, vp, vp, vp]
        L.ref_batch_add.argtypes = [ctypes.c_int, vp, sz, vp, sz, vp, vp]
        L.ref_sequential_bases.argtypes = [ctypes.c_int, vp, u64, sz, vp]
        L.ref_field_axpy.argtypes = [ctypes.c_int, vp, vp, vp, sz, vp]
        L.ref_field_axpy2.argtypes = [ctypes.c_int, vp, vp, vp, vp, sz, vp]
        L.ref_field_cross_term.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, sz, vp]
        L.ref_field_cross_term2.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, vp, sz, vp]
        L.ref_field_bind.argtypes = [ctypes.c_int, vp, sz, sz, sz, vp, sz, vp]
        L.ref_poly_suffix_horner.argtypes = [ctypes.c_int, vp, sz, vp, vp]
        L.ref_eq_evals.argtypes = [ctypes.c_int, vp, sz, vp]
        L.ref_mle_evaluate.argtypes = [ctypes.c_int, vp, sz, vp, vp]
        L.ref_spmv.argtypes = [ctypes.c_int, vp, vp, vp, sz, vp, vp]
        L.ref_sumcheck_eq_sums.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, sz, vp, vp, ctypes.c_uint, vp]
        L.ref_spmv_pair.argtypes = [ctypes.c_int, vp, vp, vp, sz, vp, vp, vp, vp]
        L.ref_sumcheck_plain_sums.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, sz, vp]
        L.ref_lincomb_powers.argtypes = [ctypes.c_int, vp, vp, sz, vp, sz, vp]
        L.ref_mle_multi_evaluate.argtypes = [ctypes.c_int, vp, sz, sz, vp, vp]
        L.ref_spmv_transposed.argtypes = [ctypes.c_int, vp, vp, vp, sz, sz, vp, vp]
        L.ref_batch_invert.argtypes = [ctypes.c_int, vp, sz, vp]
        L.ref_unipoly_from_evals.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        L.ref_sumcheck_prove_cubic3.argtypes = [ctypes.c_int, vp, vp, sz, vp, vp, vp, TRANSCRIPT_FN, vp, vp, vp, vp]
        L.ref_sumcheck_prove_quad_prod.argtypes = [ctypes.c_int, vp, sz, vp, vp, TRANSCRIPT_FN, vp, vp, vp, vp]
        L.ref_sumcheck_prove_batch_eval.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, sz, TRANSCRIPT_FN, vp, vp, vp, vp]
        L.ref_ipa_prove.argtypes = [ctypes.c_int, vp, vp, vp, vp, sz, IPA_TRANSCRIPT_FN, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def _buf(a):
    """numpy uint8/uint64 array or bytes -> (pointer, keepalive)"""
    if isinstance(a, (bytes, bytearray)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    a = np.ascontiguousarray(a)
    return a.ctypes.data, a


def _out():
    out = np.zeros(64, dtype=np.uint8)
    inf = np.zeros(1, dtype=np.uint8)
    return out, inf


def set_threads(t):
    lib().ref_set_threads(int(t))


def get_threads():
    return lib().ref_get_threads()


def msm(cid, scalars, bases, n):
    """scalars: n x 32 canonical LE bytes; bases: n x 64 canonical x||y.  Returns (64 bytes, is_inf)."""
    sp, _s = _buf(scalars)
    bp, _b = _buf(bases)
    out, inf = _out()
    rc = lib().ref_msm(cid, sp, bp, n, out.ctypes.data, inf.ctypes.data)
    if rc:
        raise ValueError(f"ref_msm rc={rc}")
    return out.tobytes(), int(inf[0])


def msm_u64(cid, scalars_u64, bases, n, max_num_bits=None):
    s = np.ascontiguousarray(scalars_u64, dtype=np.uint64)
    bp, _b = _buf(bases)
    out, inf = _out()
    mb = (1 << 64) - 1 if max_num_bits is None else int(max_num_bits)
    rc = lib().ref_msm_u64(cid, s.ctypes.data, bp, n, mb, out.ctypes.data, inf.ctypes.data)
    if rc:
        raise ValueError(f"ref_msm_u64 rc={rc}")
    return out.tobytes(), int(inf[0])


class Prepared:
    """A host key converted once (what a `Vec<Affine>` already is on the reference side)."""

    def __init__(self, cid, bases, n):
        bp, _b = _buf(bases)
        self.cid, self.n = cid, n
        self.h = lib().ref_bases_load(cid, bp, n)

    def msm(self, scalars, n, best_only=False):
        sp, _s = _buf(scalars)
        out, inf = _out()
        fn = lib().ref_msm_best_prepared if best_only else lib().ref_msm_prepared
        rc = fn(self.cid, sp, self.h, n, out.ctypes.data, inf.ctypes.data)
        if rc:
            raise ValueError(f"ref_msm_prepared rc={rc}")
        return out.tobytes(), int(inf[0])

    def msm_u64(self, scalars_u64, n, max_num_bits=None):
        s = np.ascontiguousarray(scalars_u64, dtype=np.uint64)
        out, inf = _out()
        mb = (1 << 64) - 1 if max_num_bits is None else int(max_num_bits)
        rc = lib().ref_msm_u64_prepared(self.cid, s.ctypes.data, self.h, n, mb, out.ctypes.data, inf.ctypes.data)
        if rc:
            raise ValueError(f"ref_msm_u64_prepared rc={rc}")
        return out.tobytes(), int(inf[0])

    def close(self):
        if self.h:
            lib().ref_bases_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def msm_batch(cid, vecs, bases, n_bases):
    k = len(vecs)
    keep = [np.ascontiguousarray(np.frombuffer(bytes(v), dtype=np.uint8)) if len(v) else np.zeros(1, np.uint8) for v in vecs]
    ptrs = (ctypes.c_void_p * k)(*[a.ctypes.data for a in keep])
    lens = (ctypes.c_size_t * k)(*[len(v) // 32 for v in vecs])
    bp, _b = _buf(bases)
    out = np.zeros(64 * max(k, 1), dtype=np.uint8)
    inf = np.zeros(max(k, 1), dtype=np.uint8)
    rc = lib().ref_msm_batch(cid, ptrs, lens, k, bp, n_bases, out.ctypes.data, inf.ctypes.data)
    if rc:
        raise ValueError(f"ref_msm_batch rc={rc}")
    return [(out[64 * j: 64 * j + 64].tobytes(), int(inf[j])) for j in range(k)]


def commit(cid, v, ck, n, h, r):
    vp_, _v = _buf(v)
    cp, _c = _buf(ck)
    hp, _h = _buf(h)
    rp, _r = _buf(r)
    out, inf = _out()
    rc = lib().ref_commit(cid, vp_, cp, n, hp, rp, out.ctypes.data, inf.ctypes.data)
    if rc:
        raise ValueError(f"ref_commit rc={rc}")
    return out.tobytes(), int(inf[0])


def sequential_bases(curve, k0, n):
    """P_i = (k0+i)*G as an (n, 64) uint8 array; `curve` is an oracle.pyref.Curve."""
    from . import pyref
    g = np.frombuffer(pyref.point_to_xy64((curve.gx, curve.gy)), dtype=np.uint8)
    out = np.zeros((n, 64), dtype=np.uint8)
    rc = lib().ref_sequential_bases(curve.cid, g.ctypes.data, k0, n, out.ctypes.data)
    if rc:
        raise ValueError(f"ref_sequential_bases rc={rc}")
    return out


def field_axpy(fid, a, b, r, n):
    ap, _a = _buf(a)
    bp, _b = _buf(b)
    rp, _r = _buf(r)
    out = np.zeros(32 * n, dtype=np.uint8)
    lib().ref_field_axpy(fid, ap, bp, rp, n, out.ctypes.data)
    return out.tobytes()


def field_axpy2(fid, a, b, c, r, n):
    ps = [_buf(x) for x in (a, b, c, r)]
    out = np.zeros(32 * n, dtype=np.uint8)
    lib().ref_field_axpy2(fid, ps[0][0], ps[1][0], ps[2][0], ps[3][0], n, out.ctypes.data)
    return out.tobytes()


def field_cross_term(fid, az, bz, cz, e, u, n):
    ps = [_buf(x) for x in (az, bz, cz, e, u)]
    out = np.zeros(32 * n, dtype=np.uint8)
    lib().ref_field_cross_term(fid, ps[0][0], ps[1][0], ps[2][0], ps[3][0], ps[4][0], n, out.ctypes.data)
    return out.tobytes()


def field_cross_term2(fid, az, bz, cz, e1, e2, u, n):
    ps = [_buf(x) for x in (az, bz, cz, e1, e2, u)]
    out = np.zeros(32 * n, dtype=np.uint8)
    lib().ref_field_cross_term2(fid, *[q[0] for q in ps], n, out.ctypes.data)
    return out.tobytes()


def field_bind(fid, z, lo, hi, stride, r, n_out):
    pz, _z = _buf(z)
    pr, _r = _buf(r)
    out = np.zeros(32 * n_out, dtype=np.uint8)
    lib().ref_field_bind(fid, pz, lo, hi, stride, pr, n_out, out.ctypes.data)
    return out.tobytes()


def sumcheck_eq_sums(fid, mode, A, B, C, n, eq_right, eq_left=None, shift=0):
    ps = [_buf(x) if x is not None else (None, None) for x in (A, B, C, eq_left, eq_right)]
    out = np.zeros(64, dtype=np.uint8)
    rc = lib().ref_sumcheck_eq_sums(fid, mode, ps[0][0], ps[1][0], ps[2][0], n, ps[3][0], ps[4][0], shift, out.ctypes.data)
    assert rc == 0
    return out[:32].tobytes(), out[32:].tobytes()


def eq_evals(fid, r, ell):
    pr, _r = _buf(r)
    out = np.zeros(32 << ell, dtype=np.uint8)
    lib().ref_eq_evals(fid, pr, ell, out.ctypes.data)
    return out.tobytes()


def mle_evaluate(fid, z, ell, r):
    pz, _z = _buf(z)
    pr, _r = _buf(r)
    out = np.zeros(32, dtype=np.uint8)
    lib().ref_mle_evaluate(fid, pz, ell, pr, out.ctypes.data)
    return out.tobytes()


def spmv(fid, indptr, indices, data, rows, z):
    ip = np.ascontiguousarray(indptr, dtype=np.uint64)
    ix = np.ascontiguousarray(indices, dtype=np.uint64)
    pd, _d = _buf(data)
    pz, _z = _buf(z)
    out = np.zeros(32 * rows, dtype=np.uint8)
    lib().ref_spmv(fid, ip.ctypes.data, ix.ctypes.data, pd, rows, pz, out.ctypes.data)
    return out.tobytes()


def suffix_horner(fid, f, n, u):
    pf, _f = _buf(f)
    pu, _u = _buf(u)
    out = np.zeros(32 * n, dtype=np.uint8)
    lib().ref_poly_suffix_horner(fid, pf, n, pu, out.ctypes.data)
    return out.tobytes()


def spmv_pair(fid, indptr, indices, data, rows, z1, z2):
    ip = np.ascontiguousarray(indptr, dtype=np.uint64)
    ix = np.ascontiguousarray(indices, dtype=np.uint64)
    pd, _d = _buf(data)
    p1, _1 = _buf(z1)
    p2, _2 = _buf(z2)
    o1 = np.zeros(32 * rows, dtype=np.uint8)
    o2 = np.zeros(32 * rows, dtype=np.uint8)
    lib().ref_spmv_pair(fid, ip.ctypes.data, ix.ctypes.data, pd, rows, p1, p2, o1.ctypes.data, o2.ctypes.data)
    return o1.tobytes(), o2.tobytes()


def sumcheck_plain_sums(fid, kind, A, B, C, n):
    ps = [_buf(x) if x is not None else (None, None) for x in (A, B, C)]
    out = np.zeros(96, dtype=np.uint8)
    rc = lib().ref_sumcheck_plain_sums(fid, kind, ps[0][0], ps[1][0], ps[2][0], n, out.ctypes.data)
    assert rc == 0
    return out[:32].tobytes(), out[32:64].tobytes(), out[64:].tobytes()


def _ptr_table(vecs):
    keep = [np.ascontiguousarray(np.frombuffer(bytes(v), dtype=np.uint8)) if len(v) else np.zeros(1, np.uint8) for v in vecs]
    ptrs = (ctypes.c_void_p * max(len(vecs), 1))(*[a.ctypes.data for a in keep])
    return ptrs, keep


def lincomb_powers(fid, vecs, s, n_out):
    k = len(vecs)
    ptrs, _keep = _ptr_table(vecs)
    lens = (ctypes.c_size_t * max(k, 1))(*[len(v) // 32 for v in vecs])
    ps, _s = _buf(s)
    out = np.zeros(32 * max(n_out, 1), dtype=np.uint8)
    lib().ref_lincomb_powers(fid, ptrs, lens, k, ps, n_out, out.ctypes.data)
    return out[: 32 * n_out].tobytes()


def mle_multi_evaluate(fid, zs, ell, r):
    k = len(zs)
    ptrs, _keep = _ptr_table(zs)
    pr, _r = _buf(r) if len(r) else (None, None)
    out = np.zeros(32 * max(k, 1), dtype=np.uint8)
    lib().ref_mle_multi_evaluate(fid, ptrs, k, ell, pr, out.ctypes.data)
    return [out[32 * j: 32 * j + 32].tobytes() for j in range(k)]


def spmv_transposed(fid, indptr, indices, data, rows, cols, rx):
    """compute_eval_table_sparse's inner (src/spartan/mod.rs:506-512): out[col] = sum_row rx[row] * M[row, col]."""
    ip = np.ascontiguousarray(indptr, dtype=np.uint64)
    ix = np.ascontiguousarray(indices, dtype=np.uint64)
    pd, _d = _buf(data)
    px, _x = _buf(rx)
    out = np.zeros(32 * max(cols, 1), dtype=np.uint8)
    rc = lib().ref_spmv_transposed(fid, ip.ctypes.data, ix.ctypes.data, pd, rows, cols, px, out.ctypes.data)
    assert rc == 0
    return out[: 32 * cols].tobytes()


def make_ipa_transcript(fn):
    """Wrap fn(L_xy64: bytes, L_is_inf: bool, R_xy64: bytes, R_is_inf: bool) -> 32-byte challenge as the C callback of the
    inner-product argument (the same callback type serves the product's nmx_ipa_prove)."""
    def cb(_ctx, L, Li, R, Ri, out):
        try:
            ch = fn(bytes(L[:64]), bool(Li), bytes(R[:64]), bool(Ri))
            ctypes.memmove(out, ch, 32)
            return 0
        except Exception:          # never let an exception cross the C frame
            import traceback
            traceback.print_exc()
            return 1
    return IPA_TRANSCRIPT_FN(cb)


def ipa_prove(cid, ck, ck_c, a, b, n, transcript, ctx=None):
    """InnerProductArgument::prove (src/provider/ipa_pc.rs:174-281) with the key fold of pedersen.rs:484-497: ck = n points
    (xy64), ck_c = the scaled one-point key, a / b = n canonical scalars.  Returns (L [rounds] xy64, R [rounds] xy64,
    infs [rounds][2], a_hat) or raises ValueError(code)."""
    rounds = max(n.bit_length() - 1, 0)
    ps = [_buf(x) for x in (ck, ck_c, a, b)]
    oL = np.zeros(64 * max(rounds, 1), np.uint8)
    oR = np.zeros(64 * max(rounds, 1), np.uint8)
    oi = np.zeros(2 * max(rounds, 1), np.uint8)
    ah = np.zeros(32, np.uint8)
    rc = lib().ref_ipa_prove(cid, ps[0][0], ps[1][0], ps[2][0], ps[3][0], n, transcript, ctx, oL.ctypes.data, oR.ctypes.data,
                             oi.ctypes.data, ah.ctypes.data)
    if rc != 0:
        raise ValueError(rc)
    Lb, Rb = oL.tobytes(), oR.tobytes()
    return ([Lb[64 * j: 64 * j + 64] for j in range(rounds)], [Rb[64 * j: 64 * j + 64] for j in range(rounds)],
            [(bool(oi[2 * j]), bool(oi[2 * j + 1])) for j in range(rounds)], ah.tobytes())


def make_transcript(fn):
    """Wrap fn(list of 32-byte coefficient strings) -> 32-byte challenge as the C callback of the sum-check provers (the same
    callback type serves the product's nmx_sumcheck_prove_* entry points)."""
    def cb(_ctx, coeffs, n, out):
        try:
            cs = [bytes(coeffs[32 * i: 32 * i + 32]) for i in range(n)]
            ch = fn(cs)
            ctypes.memmove(out, ch, 32)
            return 0
        except Exception:          # never let an exception cross the C frame
            import traceback
            traceback.print_exc()
            return 1
    return TRANSCRIPT_FN(cb)


def sumcheck_prove_cubic3(fid, claim, taus, A, B, C, transcript, ctx=None):
    """SumcheckProof::prove_cubic_with_three_inputs (src/spartan/sumcheck.rs:446-507).  Returns (polys [rounds][4], r [rounds],
    claims [3]) as 32-byte strings; `transcript` = make_transcript(...)."""
    tp, _t = _buf(taus)
    nr = len(_t.reshape(-1)) // 32
    ps = [_buf(x) for x in (claim, A, B, C)]
    polys = np.zeros(128 * max(nr, 1), np.uint8)
    r = np.zeros(32 * max(nr, 1), np.uint8)
    cl = np.zeros(96, np.uint8)
    rc = lib().ref_sumcheck_prove_cubic3(fid, ps[0][0], tp, nr, ps[1][0], ps[2][0], ps[3][0], transcript, ctx, polys.ctypes.data,
                                         r.ctypes.data, cl.ctypes.data)
    assert rc == 0
    pb, rb, cb = polys.tobytes(), r.tobytes(), cl.tobytes()
    return ([[pb[128 * j + 32 * i: 128 * j + 32 * i + 32] for i in range(4)] for j in range(nr)],
            [rb[32 * j: 32 * j + 32] for j in range(nr)], [cb[32 * i: 32 * i + 32] for i in range(3)])


def sumcheck_prove_quad_prod(fid, claim, num_rounds, A, B, transcript, ctx=None):
    """SumcheckProof::prove_quad_prod (src/spartan/sumcheck.rs:199-249): (polys [rounds][3], r, [A(r), B(r)])."""
    ps = [_buf(x) for x in (claim, A, B)]
    nr = num_rounds
    polys = np.zeros(96 * max(nr, 1), np.uint8)
    r = np.zeros(32 * max(nr, 1), np.uint8)
    cl = np.zeros(64, np.uint8)
    rc = lib().ref_sumcheck_prove_quad_prod(fid, ps[0][0], nr, ps[1][0], ps[2][0], transcript, ctx, polys.ctypes.data,
                                            r.ctypes.data, cl.ctypes.data)
    assert rc == 0
    pb, rb, cb = polys.tobytes(), r.tobytes(), cl.tobytes()
    return ([[pb[96 * j + 32 * i: 96 * j + 32 * i + 32] for i in range(3)] for j in range(nr)],
            [rb[32 * j: 32 * j + 32] for j in range(nr)], [cb[:32], cb[32:]])


def sumcheck_prove_batch_eval(fid, claims, num_rounds, polys, eq_points, coeffs, transcript, ctx=None):
    """SumcheckProof::prove_batch_eval (src/spartan/sumcheck.rs:251-353): (polys [max rounds][3], r, [P_i final])."""
    k = len(polys)
    nmax = max(num_rounds)
    pp, _kp = _ptr_table(polys)
    qp, _kq = _ptr_table(eq_points)
    nr = (ctypes.c_size_t * k)(*num_rounds)
    pc, _c = _buf(b"".join(claims) if isinstance(claims, (list, tuple)) else claims)
    pw, _w = _buf(b"".join(coeffs) if isinstance(coeffs, (list, tuple)) else coeffs)
    out_p = np.zeros(96 * max(nmax, 1), np.uint8)
    r = np.zeros(32 * max(nmax, 1), np.uint8)
    fin = np.zeros(32 * k, np.uint8)
    rc = lib().ref_sumcheck_prove_batch_eval(fid, pc, nr, pp, qp, pw, k, transcript, ctx, out_p.ctypes.data, r.ctypes.data,
                                             fin.ctypes.data)
    assert rc == 0
    pb, rb, fb = out_p.tobytes(), r.tobytes(), fin.tobytes()
    return ([[pb[96 * j + 32 * i: 96 * j + 32 * i + 32] for i in range(3)] for j in range(nmax)],
            [rb[32 * j: 32 * j + 32] for j in range(nmax)], [fb[32 * i: 32 * i + 32] for i in range(k)])


def batch_invert(fid, v, n):
    """batch_invert (src/spartan/mod.rs:54-152): bytes of the inverses, or None when an element is zero (the reference's Err)."""
    pv, _v = _buf(v)
    out = np.zeros(32 * max(n, 1), dtype=np.uint8)
    rc = lib().ref_batch_invert(fid, pv, n, out.ctypes.data)
    assert rc in (0, 1)
    return None if rc else out[: 32 * n].tobytes()


def unipoly_from_evals(fid, evals, at):
    """UniPoly::from_evals_deg2 / _deg3 + evaluate (src/spartan/polys/univariate.rs:90-113, 140-149) on integers: -> (coefficients, value at `at`)."""
    deg = len(evals) - 1
    ev = b"".join(int(x).to_bytes(32, "little") for x in evals)
    pe, _e = _buf(ev)
    pa, _a = _buf(int(at).to_bytes(32, "little"))
    co, val = np.zeros(32 * (deg + 1), dtype=np.uint8), np.zeros(32, dtype=np.uint8)
    assert lib().ref_unipoly_from_evals(fid, deg, pe, pa, co.ctypes.data, val.ctypes.data) == 0
    cb = co.tobytes()
    return [int.from_bytes(cb[32 * i: 32 * i + 32], "little") for i in range(deg + 1)], int.from_bytes(val.tobytes(), "little")
