"""Field-vector kernels either side of the MSM: thin marshalling over the C ABI.

Reference sites (relative to /root/reference):
  axpy        W = W1 + r*W2, E = E1 + r*T           src/r1cs/mod.rs:1058-1067   (RelaxedR1CSWitness::fold)
  axpy2       E = E1 + r*T + r^2*E2                 src/r1cs/mod.rs:1096-1101   (fold_relaxed)
  cross_term  T = AZ o BZ - u*CZ - E                src/r1cs/mod.rs:614-620     (commit_T)
  cross_term2 T = AZ o BZ - u*CZ - E1 - E2          src/r1cs/mod.rs:652-659     (commit_T_relaxed)
  vec_add     Z = Z1 + Z2                           src/r1cs/mod.rs:590-609
  bind_poly_var_top                                 src/spartan/polys/multilinear.rs:65-84
  fold_pairs  Pi[j] = P[2j] + x*(P[2j+1] - P[2j])   src/provider/hyperkzg.rs:1085-1095
Vectors: (n, 32) uint8 numpy arrays (host) or torch CUDA uint8 tensors (HBM-resident; the result is then a CUDA
tensor too and can be passed straight to DlogGroup.vartime_multiscalar_mul).  Field ids as include/nova_mi355x.h.
"""
import numpy as np

from . import _lib as L
from .provider import _check, _host_u8, _is_device_tensor

BN254_FQ, BN254_FR, PASTA_FP, PASTA_FQ = 0, 1, 2, 3
SCALAR_FIELD_OF_CURVE = {0: BN254_FR, 1: BN254_FQ, 2: PASTA_FQ, 3: PASTA_FP}


def _vec(x):
    if _is_device_tensor(x):
        assert x.is_contiguous() and x.numel() * x.element_size() % 32 == 0
        return x.data_ptr(), x.numel() * x.element_size() // 32, True, x
    a = _host_u8(x, 32)
    return a.ctypes.data, a.size // 32, False, a


def _out_like(dev, n, ref):
    if dev:
        import torch
        o = torch.empty((n, 32), dtype=torch.uint8, device=ref.device)
        return o.data_ptr(), o
    o = np.zeros((n, 32), dtype=np.uint8)
    return o.ctypes.data, o


def _chal(r):
    a = _host_u8(r, 32)
    assert a.size == 32
    return a


def _flags(dev, mont, async_=False):
    """async_ (NMX_ASYNC): device-resident operands only -- the call returns once its kernel is enqueued; the calling thread's
    next synchronous call (any MSM / commitment / reduction) or sync() completes it.  Keep the operand tensors alive until then:
    torch's allocator knows nothing about the library's stream."""
    return (L.SCALARS_DEVICE if dev else 0) | (L.SCALARS_MONT if mont else 0) | (L.ASYNC if (async_ and dev) else 0)


def sync():
    """nmx_sync: wait for this thread's asynchronous calls."""
    _check(L.lib().nmx_sync())


def axpy(field, a, b, r, mont=False, async_=False):
    pa, n, dev, _ka = _vec(a)
    pb, nb, devb, _kb = _vec(b)
    assert n == nb and dev == devb, "InvalidWitnessLength (r1cs/mod.rs:1054-1056)"
    po, out = _out_like(dev, n, a)
    rr = _chal(r)
    _check(L.lib().nmx_field_axpy(field, pa, pb, rr.ctypes.data, n, _flags(dev, mont, async_), po))
    return out


def axpy2(field, a, b, c, r, mont=False, async_=False):
    pa, n, dev, _ka = _vec(a)
    pb, nb, _d1, _kb = _vec(b)
    pc, nc, _d2, _kc = _vec(c)
    assert n == nb == nc
    po, out = _out_like(dev, n, a)
    rr = _chal(r)
    _check(L.lib().nmx_field_axpy2(field, pa, pb, pc, rr.ctypes.data, n, _flags(dev, mont, async_), po))
    return out


def cross_term(field, az, bz, cz, e, u, mont=False, async_=False):
    p1, n, dev, _k1 = _vec(az)
    p2, n2, _d2, _k2 = _vec(bz)
    p3, n3, _d3, _k3 = _vec(cz)
    p4, n4, _d4, _k4 = _vec(e)
    assert n == n2 == n3 == n4
    po, out = _out_like(dev, n, az)
    uu = _chal(u)
    _check(L.lib().nmx_field_cross_term(field, p1, p2, p3, p4, uu.ctypes.data, n, _flags(dev, mont, async_), po))
    return out


def cross_term2(field, az, bz, cz, e1, e2, u, mont=False, async_=False):
    """T = AZ o BZ - u*CZ - E1 - E2, u = U1.u + U2.u (commit_T_relaxed, src/r1cs/mod.rs:652-659)."""
    ps = [_vec(x) for x in (az, bz, cz, e1, e2)]
    n, dev = ps[0][1], ps[0][2]
    assert all(q[1] == n for q in ps)
    po, out = _out_like(dev, n, az)
    uu = _chal(u)
    _check(L.lib().nmx_field_cross_term2(field, *[q[0] for q in ps], uu.ctypes.data, n, _flags(dev, mont, async_), po))
    return out


def vec_add(field, a, b, mont=False, async_=False):
    pa, n, dev, _ka = _vec(a)
    pb, nb, _d, _kb = _vec(b)
    assert n == nb
    po, out = _out_like(dev, n, a)
    _check(L.lib().nmx_field_vec_add(field, pa, pb, n, _flags(dev, mont, async_), po))
    return out


def batch_invert(field, v, mont=False):
    """batch_invert (src/spartan/mod.rs:54-118): element-wise inverses; NmxError(E_ZERO) when an element is zero."""
    pv, n, dev, _kv = _vec(v)
    po, out = _out_like(dev, n, v)
    _check(L.lib().nmx_field_batch_invert(field, pv, n, _flags(dev, mont), po))
    return out


def concat(field, parts, n_out=None, async_=False):
    """nmx_field_concat: an HBM-resident vector = the parts (CUDA tensors or host arrays) one after the other, zero-padded to n_out.
    Spartan's z = [W, u, X] (src/spartan/snark.rs:133, 193-196); with one part, a clone ordered on the library's stream."""
    import ctypes
    import torch
    k = len(parts)
    ps = [_vec(p) for p in parts]
    total = sum(q[1] for q in ps)
    n = total if n_out is None else n_out
    mask = sum(1 << i for i, q in enumerate(ps) if q[2])
    ptrs = (ctypes.c_void_p * max(k, 1))(*[q[0] for q in ps])
    lens = (ctypes.c_size_t * max(k, 1))(*[q[1] for q in ps])
    out = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    _check(L.lib().nmx_field_concat(field, ptrs, lens, mask, k, n, _flags(True, False, async_), out.data_ptr()))
    return out


def bind_poly_var_top(field, z, r, mont=False, in_place=False, async_=False):
    """Returns the bound polynomial (len/2 evaluations).  in_place (device tensors only) overwrites z[:len/2]."""
    pz, n, dev, _kz = _vec(z)
    assert n >= 2 and n % 2 == 0, "assert!(self.num_vars > 0)"
    rr = _chal(r)
    if in_place:
        assert dev
        _check(L.lib().nmx_mle_bind_top(field, pz, n, rr.ctypes.data, _flags(dev, mont, async_), pz))
        return z.view(-1)[: (n // 2) * 32].view(n // 2, 32)
    po, out = _out_like(dev, n // 2, z)
    _check(L.lib().nmx_mle_bind_top(field, pz, n, rr.ctypes.data, _flags(dev, mont, async_), po))
    return out


def fold_pairs(field, p, x, mont=False, async_=False):
    pp, n, dev, _kp = _vec(p)
    assert n >= 2 and n % 2 == 0
    xx = _chal(x)
    po, out = _out_like(dev, n // 2, p)
    _check(L.lib().nmx_poly_fold_pairs(field, pp, n, xx.ctypes.data, _flags(dev, mont, async_), po))
    return out


def fold_chain(field, p, xs, mont=False, async_=False):
    """nmx_poly_fold_chain: HyperKZG's fold loop (hyperkzg.rs:1085-1095) in one call -- [fold(p, xs[0]), fold(that, xs[1]), ...]."""
    import ctypes
    pp, n, dev, _kp = _vec(p)
    xx = _host_u8(xs, 32)
    k = xx.size // 32
    assert n >= 2 and n & (n - 1) == 0 and (n >> k) >= 1
    outs = [_out_like(dev, n >> (i + 1), p) for i in range(k)]
    ptrs = (ctypes.c_void_p * k)(*[o[0] for o in outs])
    _check(L.lib().nmx_poly_fold_chain(field, pp, n, xx.ctypes.data, k, _flags(dev, mont, async_), ptrs))
    return [o[1] for o in outs]


def sumcheck_eq_sums(field, mode, A, B, C, eq_right, eq_left=None, shift=0, mont=False):
    """(t_0, t_inf) of EqSumCheckInstance::evaluation_points_{quadratic_with_one_input (mode 1),
    cubic_with_two_inputs (2), cubic_with_three_inputs (3)} (src/spartan/sumcheck.rs:900-1075) as two 32-byte field
    elements.  eq_left given: first-half rounds, factor = eq_left[id >> shift] * eq_right[id & (2^shift - 1)]."""
    pa, n, dev, _ka = _vec(A)
    pb = pc = None
    keep = []
    if mode >= 2:
        pb, nb, _d, kb = _vec(B)
        assert nb == n
        keep.append(kb)
    if mode >= 3:
        pc, nc, _d, kc = _vec(C)
        assert nc == n
        keep.append(kc)
    pr, nr, _d, kr = _vec(eq_right)
    pl, nl = None, 0
    if eq_left is not None:
        pl, nl, _d, kl = _vec(eq_left)
        keep.append(kl)
    out = np.zeros(64, dtype=np.uint8)
    _check(L.lib().nmx_sumcheck_eq_sums(field, mode, pa, pb, pc, n, pl, nl, pr, nr, shift, _flags(dev, mont), out.ctypes.data))
    return out[:32].tobytes(), out[32:].tobytes()


def sumcheck_bind_eq_sums(field, mode, A, B, C, r, eq_right, eq_left=None, shift=0, mont=False):
    """One prover round fused (nmx_sumcheck_bind_eq_sums): binds the HBM-resident tables A, B, C IN PLACE with the
    challenge r and returns (A', B', C', (t_0, t_inf)) where the sums are the next round's, over the bound tables."""
    parts = [_vec(x) if x is not None else (None, 0, True, None) for x in (A, B, C)]
    n = parts[0][1]
    assert all(pt[2] for pt in parts), "HBM-resident tables only"
    assert n >= 4 and n % 4 == 0
    pr, nr, _d, _kr = _vec(eq_right)
    pl, nl = None, 0
    if eq_left is not None:
        pl, nl, _d, _kl = _vec(eq_left)
    rr = _chal(r)
    out = np.zeros(64, dtype=np.uint8)
    _check(L.lib().nmx_sumcheck_bind_eq_sums(field, mode, parts[0][0], parts[1][0], parts[2][0], n, rr.ctypes.data, pl, nl,
                                               pr, nr, shift, _flags(True, mont), parts[0][0], parts[1][0], parts[2][0],
                                               out.ctypes.data))
    half = lambda x: None if x is None else x.view(-1)[: (n // 2) * 32].view(n // 2, 32)
    return half(A), half(B), half(C), (out[:32].tobytes(), out[32:].tobytes())


def sumcheck_plain_sums(field, kind, A, B, C=None, mont=False):
    """Round sums of the sum-checks without an eq factor (src/spartan/sumcheck.rs): kind 1 compute_eval_points_quad_prod
    (:163-186), 2 ..._linear (:353-378), 3 ..._quadratic (:380-405), 4 ..._cubic (:407-443).  Returns a tuple of two
    (kinds 1-3) or three (kind 4) 32-byte field elements."""
    pa, n, dev, _ka = _vec(A)
    pb, nb, _d, _kb = _vec(B)
    assert nb == n
    pc = None
    if kind == 4:
        pc, nc, _d, _kc = _vec(C)
        assert nc == n
    out = np.zeros(96, dtype=np.uint8)
    _check(L.lib().nmx_sumcheck_plain_sums(field, kind, pa, pb, pc, n, _flags(dev, mont), out.ctypes.data))
    t = (out[:32].tobytes(), out[32:64].tobytes(), out[64:].tobytes())
    return t if kind == 4 else t[:2]


def lincomb_powers(field, vecs, s, n_out=None, mont=False):
    """PolyEvalWitness::batch / batch_diff_size (src/spartan/mod.rs:165-277): sum_j s^j * vecs[j], shorter vectors
    zero-padded to the longest (or to n_out)."""
    import ctypes
    k = len(vecs)
    parts = [_vec(v) for v in vecs]
    dev = bool(parts) and parts[0][2]
    assert all(pt[2] == dev for pt in parts)
    lens = [pt[1] for pt in parts]
    n = max(lens, default=0) if n_out is None else n_out
    ptrs = (ctypes.c_void_p * max(k, 1))(*[pt[0] for pt in parts])
    ln = (ctypes.c_size_t * max(k, 1))(*lens)
    po, out = _out_like(dev, n, vecs[0] if k else None)
    ss = _chal(s)
    _check(L.lib().nmx_field_lincomb_powers(field, ptrs, ln, k, ss.ctypes.data, n, _flags(dev, mont), po))
    return out


def eq_evals_from_points(field, r, mont=False, device=False):
    """EqPolynomial::evals_from_points (src/spartan/polys/eq.rs:54-73): (2^ell, 32) table, r[0] most significant.  device=True: the
    table is built and left in HBM (a CUDA tensor) -- compute_eval_table_sparse's operand, src/spartan/snark.rs:182-184."""
    rr = _host_u8(r, 32)
    ell = rr.size // 32
    if device:
        import torch
        out = torch.empty((1 << ell, 32), dtype=torch.uint8, device="cuda")
        _check(L.lib().nmx_eq_evals_from_points(field, rr.ctypes.data, ell, _flags(True, mont), out.data_ptr()))
        return out
    out = np.zeros((1 << ell, 32), dtype=np.uint8)
    _check(L.lib().nmx_eq_evals_from_points(field, rr.ctypes.data, ell, L.SCALARS_MONT if mont else 0, out.ctypes.data))
    return out


def mle_evaluate(field, z, r, mont=False):
    """MultilinearPolynomial::evaluate (src/spartan/polys/multilinear.rs:88-129)."""
    pz, n, dev, _kz = _vec(z)
    rr = _host_u8(r, 32)
    out = np.zeros(32, dtype=np.uint8)
    _check(L.lib().nmx_mle_evaluate(field, pz, n, rr.ctypes.data, rr.size // 32, _flags(dev, mont), out.ctypes.data))
    return out.tobytes()


def mle_multi_evaluate(field, zs, r, mont=False):
    """MultilinearPolynomial::multi_evaluate_with (src/spartan/polys/multilinear.rs:131-180): list of 32-byte values."""
    import ctypes
    k = len(zs)
    if k == 0:
        return []
    parts = [_vec(z) for z in zs]
    dev = parts[0][2]
    n = parts[0][1]
    assert all(pt[1] == n and pt[2] == dev for pt in parts), "assert!(Zs.iter().all(|z| z.len() == n))"
    rr = _host_u8(r, 32)
    ptrs = (ctypes.c_void_p * k)(*[pt[0] for pt in parts])
    out = np.zeros(32 * k, dtype=np.uint8)
    _check(L.lib().nmx_mle_multi_evaluate(field, ptrs, k, n, rr.ctypes.data, rr.size // 32, _flags(dev, mont), out.ctypes.data))
    return [out[32 * j: 32 * j + 32].tobytes() for j in range(k)]


def r1cs_cross_term(A, B, C, z1, z2, e, u, mont=False, async_=False):
    """commit_T's chain as one call (nmx_r1cs_cross_term; src/r1cs/mod.rs:590-620): T = (A Z)(B Z) - u (C Z) - E with Z = z1 + z2
    (z2 None: Z = z1).  A, B, C: SparseMatrix of one shape; CUDA tensors only."""
    p1, n, dev, _k1 = _vec(z1)
    p2 = _vec(z2)[0] if z2 is not None else None
    pe, ne, _de, _ke = _vec(e)
    assert dev and n == A.cols and ne == A.rows
    po, out = _out_like(dev, A.rows, z1)
    uu = _chal(u)
    _check(L.lib().nmx_r1cs_cross_term(A.handle, B.handle, C.handle, p1, p2, n, pe, uu.ctypes.data, _flags(dev, mont, async_), po))
    return out


def nifs_fold(field, w1, w2, e1, t, r, mont=False, async_=False):
    """R1CSWitness::fold as one launch (nmx_nifs_fold; src/r1cs/mod.rs:1044-1067): (w1 + r w2, e1 + r t).  CUDA tensors only."""
    pw1, nw, dev, _a = _vec(w1)
    pw2, nw2, _d, _b = _vec(w2)
    pe1, ne, _d2, _c = _vec(e1)
    pt, nt, _d3, _e = _vec(t)
    assert dev and nw == nw2 and ne == nt
    pw, w = _out_like(dev, nw, w1)
    pe, e = _out_like(dev, ne, e1)
    rr = _chal(r)
    _check(L.lib().nmx_nifs_fold(field, pw1, pw2, nw, pe1, pt, ne, rr.ctypes.data, _flags(dev, mont, async_), pw, pe))
    return w, e


class SparseMatrix:
    """CSR matrix resident in HBM (src/r1cs/sparse.rs:232-260); multiply_vec = sparse.rs:201-229."""

    def __init__(self, field, indptr, indices, data, cols, mont=False):
        import ctypes
        ip = np.ascontiguousarray(indptr, dtype=np.uint64)
        ix = np.ascontiguousarray(indices, dtype=np.uint64)
        d = _host_u8(data, 32)
        self.rows, self.cols, self.field = len(ip) - 1, cols, field
        h = ctypes.c_uint64(0)
        _check(L.lib().nmx_spmv_register(field, ip.ctypes.data, ix.ctypes.data, d.ctypes.data, self.rows, cols,
                                         L.SCALARS_MONT if mont else 0, ctypes.byref(h)))
        self.handle = h.value

    def multiply_vec(self, z, mont=False, async_=False):
        pz, n, dev, _kz = _vec(z)
        assert n == self.cols, "invalid shape"
        po, out = _out_like(dev, self.rows, z)
        _check(L.lib().nmx_spmv_apply(self.handle, pz, n, _flags(dev, mont, async_), po))
        return out

    def multiply_vec_pair(self, z1, z2, mont=False, async_=False):
        """(M*z1, M*z2) in one pass (sparse.rs:215-229)."""
        p1, n1, dev, _k1 = _vec(z1)
        p2, n2, dev2, _k2 = _vec(z2)
        assert n1 == self.cols and n2 == self.cols and dev == dev2, "invalid shape"
        po1, out1 = _out_like(dev, self.rows, z1)
        po2, out2 = _out_like(dev, self.rows, z1)
        _check(L.lib().nmx_spmv_apply_pair(self.handle, p1, p2, n1, _flags(dev, mont, async_), po1, po2))
        return out1, out2

    def multiply_vec_transposed(self, x, mont=False, async_=False):
        """M^T x: compute_eval_table_sparse's product (src/spartan/mod.rs:497-533), x over the rows, result over the columns."""
        px, n, dev, _kx = _vec(x)
        assert n == self.rows, "assert_eq!(rx.len(), S.num_cons())"
        po, out = _out_like(dev, self.cols, x)
        _check(L.lib().nmx_spmv_apply_transposed(self.handle, px, n, _flags(dev, mont, async_), po))
        return out

    def close(self):
        if self.handle:
            _check(L.lib().nmx_spmv_unregister(self.handle))
            self.handle = 0


def multiply_vec_many(mats, x, transposed=False, mont=False, async_=False):
    """nmx_spmv_apply_many: [M x for M in mats] (R1CSShape::multiply_vec, src/r1cs/mod.rs:407-471) or, transposed, [M^T x ...]
    (compute_eval_table_sparse, src/spartan/mod.rs:497-533) in one call; HBM-resident x: the products run side by side."""
    import ctypes
    px, n, dev, _kx = _vec(x)
    k = len(mats)
    hs = (ctypes.c_uint64 * k)(*[m.handle for m in mats])
    outs = [_out_like(dev, m.cols if transposed else m.rows, x) for m in mats]
    ptrs = (ctypes.c_void_p * k)(*[o[0] for o in outs])
    _check(L.lib().nmx_spmv_apply_many(hs, k, 1 if transposed else 0, px, n, _flags(dev, mont, async_), ptrs))
    return [o[1] for o in outs]


def suffix_horner(field, f, u, mont=False):
    """out[i] = sum_{k>=i} f[k] u^(k-i): out[0] = poly_eval(f, u) (hyperkzg.rs:1011-1020), out[1:] = the quotient of
    div_by_monomial(f, u) (hyperkzg.rs:961-999)."""
    pf, n, dev, _kf = _vec(f)
    uu = _chal(u)
    po, out = _out_like(dev, n, f)
    _check(L.lib().nmx_poly_suffix_horner(field, pf, n, uu.ctypes.data, _flags(dev, mont), po))
    return out


def poly_eval_multi(field, polys, points, mont=False):
    """[[f_i(u_j) for j] for i] in one launch (nmx_poly_eval_multi): HyperKZG's v matrix, hyperkzg.rs:1049-1056."""
    import ctypes
    k = len(polys)
    pts = _host_u8(points, 32)
    m = pts.size // 32
    if k == 0 or m == 0:
        return [[] for _ in range(k)]
    parts = [_vec(f) if len(f) else (None, 0, None, None) for f in polys]
    devs = {pt[2] for pt in parts if pt[2] is not None}
    assert len(devs) <= 1
    dev = bool(devs) and devs.pop()
    ptrs = (ctypes.c_void_p * k)(*[pt[0] for pt in parts])
    lens = (ctypes.c_size_t * k)(*[pt[1] for pt in parts])
    out = np.zeros(32 * k * m, dtype=np.uint8)
    _check(L.lib().nmx_poly_eval_multi(field, ptrs, lens, k, pts.ctypes.data, m, _flags(dev, mont), out.ctypes.data))
    return [[out[32 * (i * m + j): 32 * (i * m + j) + 32].tobytes() for j in range(m)] for i in range(k)]


def poly_eval(field, f, u, mont=False):
    out = suffix_horner(field, f, u, mont)
    return (out[0].cpu().numpy() if _is_device_tensor(out) else out[0]).tobytes()


def div_by_monomial(field, f, u, mont=False):
    return suffix_horner(field, f, u, mont)[1:]


# ---- Spartan's sum-check provers, one call each (nmx_sumcheck_prove_*; src/spartan/sumcheck.rs:199-507) ---------------------
def as_transcript(fn):
    """fn(list of 32-byte coefficient strings) -> 32-byte challenge, wrapped as nmx_transcript_fn.  The transcript (Keccak on
    the reference's side, src/provider/keccak.rs) is the caller's: `absorb(b"p", &poly); squeeze(b"c")`."""
    import ctypes
    if isinstance(fn, L.TRANSCRIPT_FN):
        return fn

    def cb(_ctx, coeffs, n, out):
        try:
            ch = fn([bytes(coeffs[32 * i: 32 * i + 32]) for i in range(n)])
            ctypes.memmove(out, ch, 32)
            return 0
        except Exception:                   # an exception must not cross the C frame: the call fails with NMX_E_ARG
            import traceback
            traceback.print_exc()
            return 1
    return L.TRANSCRIPT_FN(cb)


def _rows(buf, n, w):
    b = buf.tobytes()
    return [[b[32 * (w * j + i): 32 * (w * j + i) + 32] for i in range(w)] for j in range(n)]


def sumcheck_prove_cubic_with_three_inputs(field, claim, taus, A, B, C, transcript, mont=False, ctx=None):
    """SumcheckProof::prove_cubic_with_three_inputs (src/spartan/sumcheck.rs:446-507) over HBM-resident tables, bound IN PLACE.
    Returns (round polynomials [rounds][4], challenges [rounds], [A(r), B(r), C(r)]) as 32-byte strings."""
    pa, n, dev, _a = _vec(A)
    pb, nb, _d1, _b = _vec(B)
    pc, nc, _d2, _c = _vec(C)
    t = _host_u8(taus, 32)
    nr = t.size // 32
    assert n == nb == nc == (1 << nr) and dev == _d1 == _d2
    cl = _chal(claim)
    polys, r, out = np.zeros(128 * max(nr, 1), np.uint8), np.zeros(32 * max(nr, 1), np.uint8), np.zeros(96, np.uint8)
    cb = as_transcript(transcript)
    _check(L.lib().nmx_sumcheck_prove_cubic_with_three_inputs(field, cl.ctypes.data, t.ctypes.data, nr, pa, pb, pc, _flags(dev, mont),
                                                             cb, ctx, polys.ctypes.data, r.ctypes.data, out.ctypes.data))
    return _rows(polys, nr, 4), [x[0] for x in _rows(r, nr, 1)], _rows(out, 1, 3)[0]


def sumcheck_prove_quad_prod(field, claim, num_rounds, A, B, transcript, mont=False, ctx=None):
    """SumcheckProof::prove_quad_prod (src/spartan/sumcheck.rs:199-249): (polys [rounds][3], challenges, [A(r), B(r)])."""
    pa, n, dev, _a = _vec(A)
    pb, nb, _d1, _b = _vec(B)
    nr = num_rounds
    assert n == nb == (1 << nr) and dev == _d1
    cl = _chal(claim)
    polys, r, out = np.zeros(96 * max(nr, 1), np.uint8), np.zeros(32 * max(nr, 1), np.uint8), np.zeros(64, np.uint8)
    cb = as_transcript(transcript)
    _check(L.lib().nmx_sumcheck_prove_quad_prod(field, cl.ctypes.data, nr, pa, pb, _flags(dev, mont), cb, ctx, polys.ctypes.data,
                                               r.ctypes.data, out.ctypes.data))
    return _rows(polys, nr, 3), [x[0] for x in _rows(r, nr, 1)], _rows(out, 1, 2)[0]


def sumcheck_prove_batch_eval(field, claims, num_rounds, polys, eq_points, coeffs, transcript, mont=False, ctx=None):
    """SumcheckProof::prove_batch_eval (src/spartan/sumcheck.rs:251-353).  polys: CUDA tensors, bound IN PLACE (the reference binds
    clones: pass copies to keep the originals).  Returns (polys [max rounds][3], challenges, [P_i final])."""
    import ctypes
    k = len(polys)
    parts = [_vec(p) for p in polys]
    dev = parts[0][2]
    assert all(pt[2] == dev for pt in parts) and all(pt[1] == (1 << nr) for pt, nr in zip(parts, num_rounds))
    nmax = max(num_rounds)
    pts = [_host_u8(x, 32) for x in eq_points]
    assert all(x.size // 32 == nr for x, nr in zip(pts, num_rounds))
    pp = (ctypes.c_void_p * k)(*[pt[0] for pt in parts])
    qp = (ctypes.c_void_p * k)(*[x.ctypes.data for x in pts])
    nrs = (ctypes.c_size_t * k)(*num_rounds)
    cl = _host_u8(b"".join(claims) if isinstance(claims, (list, tuple)) else claims, 32)
    co = _host_u8(b"".join(coeffs) if isinstance(coeffs, (list, tuple)) else coeffs, 32)
    out_p, r, fin = np.zeros(96 * nmax, np.uint8), np.zeros(32 * nmax, np.uint8), np.zeros(32 * k, np.uint8)
    cb = as_transcript(transcript)
    _check(L.lib().nmx_sumcheck_prove_batch_eval(field, cl.ctypes.data, nrs, pp, qp, co.ctypes.data, k, _flags(dev, mont), cb, ctx,
                                                out_p.ctypes.data, r.ctypes.data, fin.ctypes.data))
    return _rows(out_p, nmax, 3), [x[0] for x in _rows(r, nmax, 1)], [x[0] for x in _rows(fin, k, 1)]
