"""CPU: the HOST side of the product's sum-check provers (nova_amd/csrc/sc_host.hpp: the per-round algebra and the tail rounds that
finish a proof on the host) compiled with g++ as complete provers over host tables (tests/cpp/sc_host_test.cpp) and put through the
checks the oracle passes -- the reference's verifier, the definition of every round polynomial -- and compared with the oracle
output for output.  On the GPU the same code runs behind nmx_sumcheck_prove_* for the rounds whose tables fit the tail."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import cref
from tests import fv_common as fc
from tests import spartan_common as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def hsc():
    global _lib
    if _lib is None:
        so = os.path.join(ROOT, "tests", "cpp", "libsc_host_test.so")
        src = os.path.join(ROOT, "tests", "cpp", "sc_host_test.cpp")
        deps = [src] + [os.path.join(ROOT, "nova_amd", "csrc", f) for f in ("sc_host.hpp", "host_fp4.hpp", "fp.hpp")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
        _lib = ctypes.CDLL(so)
    return _lib


def _buf(x):
    a = np.ascontiguousarray(np.frombuffer(bytes(x), np.uint8) if isinstance(x, (bytes, bytearray)) else x, dtype=np.uint8)
    return a.ctypes.data, a


def h_cubic3(fid, claim, taus, A, B, C, tr, mont=0):
    ps = [_buf(x) for x in (claim, taus, A, B, C)]
    nr = ps[1][1].size // 32
    polys, r, cl = np.zeros(128 * max(nr, 1), np.uint8), np.zeros(32 * max(nr, 1), np.uint8), np.zeros(96, np.uint8)
    cb = cref.make_transcript(tr)
    rc = hsc().hsc_prove_cubic3(fid, mont, *[ctypes.c_void_p(p[0]) for p in ps[:2]], ctypes.c_size_t(nr), *[ctypes.c_void_p(p[0]) for p in ps[2:]],
                                cb, None, ctypes.c_void_p(polys.ctypes.data), ctypes.c_void_p(r.ctypes.data), ctypes.c_void_p(cl.ctypes.data))
    assert rc == 0
    pb, rb, cb_ = polys.tobytes(), r.tobytes(), cl.tobytes()
    return ([[pb[128 * j + 32 * i: 128 * j + 32 * i + 32] for i in range(4)] for j in range(nr)],
            [rb[32 * j: 32 * j + 32] for j in range(nr)], [cb_[32 * i: 32 * i + 32] for i in range(3)])


def h_quad(fid, claim, nr, A, B, tr):
    ps = [_buf(x) for x in (claim, A, B)]
    polys, r, cl = np.zeros(96 * max(nr, 1), np.uint8), np.zeros(32 * max(nr, 1), np.uint8), np.zeros(64, np.uint8)
    cb = cref.make_transcript(tr)
    rc = hsc().hsc_prove_quad_prod(fid, 0, ctypes.c_void_p(ps[0][0]), ctypes.c_size_t(nr), ctypes.c_void_p(ps[1][0]), ctypes.c_void_p(ps[2][0]), cb,
                                   None, ctypes.c_void_p(polys.ctypes.data), ctypes.c_void_p(r.ctypes.data), ctypes.c_void_p(cl.ctypes.data))
    assert rc == 0
    pb, rb, cb_ = polys.tobytes(), r.tobytes(), cl.tobytes()
    return ([[pb[96 * j + 32 * i: 96 * j + 32 * i + 32] for i in range(3)] for j in range(nr)],
            [rb[32 * j: 32 * j + 32] for j in range(nr)], [cb_[:32], cb_[32:]])


def h_batch(fid, claims, nrs, polys, pts, coeffs, tr):
    k, nmax = len(polys), max(nrs)
    keep = [np.ascontiguousarray(p) for p in polys] + [np.ascontiguousarray(x) for x in pts]
    pp = (ctypes.c_void_p * k)(*[a.ctypes.data for a in keep[:k]])
    qp = (ctypes.c_void_p * k)(*[a.ctypes.data for a in keep[k:]])
    nr = (ctypes.c_size_t * k)(*nrs)
    pc, _c = _buf(b"".join(claims))
    pw, _w = _buf(b"".join(coeffs))
    out_p, r, fin = np.zeros(96 * nmax, np.uint8), np.zeros(32 * nmax, np.uint8), np.zeros(32 * k, np.uint8)
    cb = cref.make_transcript(tr)
    rc = hsc().hsc_prove_batch_eval(fid, 0, ctypes.c_void_p(pc), nr, pp, qp, ctypes.c_void_p(pw), ctypes.c_size_t(k), cb, None,
                                    ctypes.c_void_p(out_p.ctypes.data), ctypes.c_void_p(r.ctypes.data), ctypes.c_void_p(fin.ctypes.data))
    assert rc == 0
    pb, rb, fb = out_p.tobytes(), r.tobytes(), fin.tobytes()
    return ([[pb[96 * j + 32 * i: 96 * j + 32 * i + 32] for i in range(3)] for j in range(nmax)],
            [rb[32 * j: 32 * j + 32] for j in range(nmax)], [fb[32 * i: 32 * i + 32] for i in range(k)])


def o_cubic3(fid, claim, taus, A, B, C, tr):
    return cref.sumcheck_prove_cubic3(fid, claim, taus, A, B, C, cref.make_transcript(tr))


def o_quad(fid, claim, nr, A, B, tr):
    return cref.sumcheck_prove_quad_prod(fid, claim, nr, A, B, cref.make_transcript(tr))


def o_batch(fid, claims, nrs, polys, pts, coeffs, tr):
    return cref.sumcheck_prove_batch_eval(fid, claims, nrs, [p.tobytes() for p in polys], [x.tobytes() for x in pts], coeffs,
                                          cref.make_transcript(tr))


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
@pytest.mark.parametrize("l", [1, 2, 3, 5, 8])
def test_cubic(fid, l):
    assert sp.check_cubic3(h_cubic3, fid, l, seed=300 + l) == sp.check_cubic3(o_cubic3, fid, l, seed=300 + l)


@pytest.mark.parametrize("l", [2, 3, 5, 6])
def test_cubic_fallback(l):
    base = fc.ints(fc.rand_vec(1, l, 55))
    for zero_at in range(l):
        taus = list(base)
        taus[zero_at] = 0
        for force in (None, {zero_at: 1}):
            kw = dict(seed=400 + zero_at, taus=taus, force=force)
            assert sp.check_cubic3(h_cubic3, 1, l, **kw) == sp.check_cubic3(o_cubic3, 1, l, **kw)
    assert sp.check_cubic3(h_cubic3, 1, l, seed=77, taus=[0] * l) == sp.check_cubic3(o_cubic3, 1, l, seed=77, taus=[0] * l)


@pytest.mark.parametrize("fid", [1, 3])
def test_cubic_montgomery_words(fid):
    p = fc.FIELDS[fid]
    Rm = 1 << 256
    to_m = lambda v: fc.vec([x * Rm % p for x in fc.ints(v)])
    un_m = lambda b: int(int.from_bytes(b, "little") * pow(Rm, -1, p) % p).to_bytes(32, "little")

    def prove_m(fid_, claim, taus, A, B, C, tr):
        def tr_m(coeffs):
            ch = tr([un_m(c) for c in coeffs])
            return int(int.from_bytes(ch, "little") * Rm % p).to_bytes(32, "little")
        polys, rs, claims = h_cubic3(fid_, to_m(np.frombuffer(claim, np.uint8)), to_m(taus), to_m(A), to_m(B), to_m(C), tr_m, mont=1)
        return [[un_m(c) for c in row] for row in polys], [un_m(r) for r in rs], [un_m(c) for c in claims]
    assert sp.check_cubic3(prove_m, fid, 6, seed=9) == sp.check_cubic3(o_cubic3, fid, 6, seed=9)


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
@pytest.mark.parametrize("l", [1, 2, 5, 9])
def test_quad(fid, l):
    assert sp.check_quad_prod(h_quad, fid, l, seed=600 + l) == sp.check_quad_prod(o_quad, fid, l, seed=600 + l)


@pytest.mark.parametrize("fid", [1, 3])
@pytest.mark.parametrize("nrs", [[4], [5, 5], [3, 6], [6, 3], [7, 2, 5], [1, 4]])
def test_batch(fid, nrs):
    assert sp.check_batch_eval(h_batch, fid, nrs, seed=800 + sum(nrs)) == sp.check_batch_eval(o_batch, fid, nrs, seed=800 + sum(nrs))
    force = {0: 0, max(nrs) - 1: 1}
    assert sp.check_batch_eval(h_batch, fid, nrs, seed=5, force=force) == sp.check_batch_eval(o_batch, fid, nrs, seed=5, force=force)


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_unipoly_known_answers_of_the_reference(fid):
    """src/spartan/polys/univariate.rs:284-355: from_evals_deg2([1, 6, 2]) = 2x^2 + 3x + 1 (value 28 at 3); from_evals_deg3([1, 7, 1, -1]) =
    x^3 + 2x^2 + 3x + 1 (value 109 at 4) -- through the oracle's constructors and through the product's host algebra (ScAlg)."""
    p = fc.FIELDS[fid]

    def host(evals, at):
        deg = len(evals) - 1
        pe, _e = _buf(b"".join(int(x).to_bytes(32, "little") for x in evals))
        pa, _a = _buf(int(at).to_bytes(32, "little"))
        co, val = np.zeros(32 * (deg + 1), np.uint8), np.zeros(32, np.uint8)
        assert hsc().hsc_unipoly_from_evals(fid, deg, ctypes.c_void_p(pe), ctypes.c_void_p(pa), ctypes.c_void_p(co.ctypes.data),
                                            ctypes.c_void_p(val.ctypes.data)) == 0
        cb = co.tobytes()
        return [int.from_bytes(cb[32 * i: 32 * i + 32], "little") for i in range(deg + 1)], int.from_bytes(val.tobytes(), "little")
    for f in (cref.unipoly_from_evals, lambda _fid, ev, at: host(ev, at)):
        assert f(fid, [1, 6, 2], 3) == ([1, 3, 2], 28)
        assert f(fid, [1, 7, 1, p - 1], 4) == ([1, 3, 2, 1], 109)
        assert f(fid, [1, 6, 2], 0)[1] == 1 and f(fid, [1, 6, 2], 1)[1] == 6            # eval_at_zero / eval_at_one
        assert f(fid, [1, 7, 1, p - 1], p - 1)[1] == p - 1                                # f(-1) = -1


def test_quad_prod_round_polynomial_is_the_references_known_answer():
    """The same quadratic through the PROVERS: A = [1, 2], B = [1, 3] has e0 = 1, e1 = 6, quadratic coefficient (2 - 1)(3 - 1) = 2 and claim 7 --
    the one round polynomial must be [1, 3, 2] (univariate.rs:284-300)."""
    for prove in (o_quad, h_quad):
        tr = sp.StandInTranscript(fc.FIELDS[1])
        polys, _r, _c = prove(1, sp.le(7), 1, fc.vec([1, 2]), fc.vec([1, 3]), tr)
        assert [int.from_bytes(c, "little") for c in polys[0]] == [1, 3, 2]
