"""-m gpu: nmx_commit_begin / nmx_commit_finish -- a commitment that runs beside the caller's next calls (commit(W) beside the
cross term + commit(T) of a folding step: src/r1cs/mod.rs:590-622 never reads comm_W, src/nova/nifs.rs:53-63).  Same results as
nmx_commit, errors reported by finish, tickets retire, ordered behind the thread's stream-ordered field calls."""
import ctypes

import numpy as np
import pytest

from oracle import cref
from oracle import pyref as R
from tests import fv_common as C
from tests import util

pytestmark = pytest.mark.gpu


def _pt(c):
    return (c.xy, int(c.is_inf))


@pytest.mark.parametrize("c", [R.BN254_G1, R.GRUMPKIN, R.PALLAS], ids=lambda c: c.name)
def test_begun_commitments_equal_synchronous_ones_and_the_oracle(nmx, c):
    import torch
    n = 9000
    bases = cref.sequential_bases(c, 99, n).copy()
    ck = nmx.CommitmentKey.from_host(c.cid, bases, h_xy64=cref.sequential_bases(c, 7, 1).tobytes())
    ce = nmx.CommitmentEngine(c.cid)
    vs = [util.random_scalars(c.cid, m, seed=5 + m) for m in (n, 4097, 1, 0)]
    rs = [util.random_scalars(c.cid, 1, seed=50 + i) for i in range(len(vs))]
    want = [_pt(ce.commit(ck, v, r)) for v, r in zip(vs, rs)]
    assert want[0] == cref.commit(c.cid, vs[0], bases, n, ck.h, rs[0])
    # several tickets in flight, host and HBM operands, finished out of order
    tickets = [ce.commit_begin(ck, v, r) for v, r in zip(vs, rs)]
    dv = [torch.from_numpy(v.copy()).cuda() for v in vs[:2]]
    tickets += [ce.commit_begin(ck, v, r) for v, r in zip(dv, rs)]
    got = [None] * len(tickets)
    for i in reversed(range(len(tickets))):
        got[i] = _pt(tickets[i].finish())
    assert got == want + want[:2]
    part = ce.commit_begin(ck, vs[1], rs[1], partial=True).finish()
    assert _pt(nmx.DlogGroup(c.cid).point_sum([part.xy])) == want[1]
    ck.close()


def test_finish_reports_the_commitments_error_and_tickets_retire(nmx):
    from nova_amd import _lib as L
    c = R.BN254_G1
    n = 5000
    ck = nmx.CommitmentKey.from_host(c.cid, cref.sequential_bases(c, 3, n))
    ce = nmx.CommitmentEngine(c.cid)
    v = util.random_scalars(c.cid, n, seed=1)
    bad = v.copy()
    bad[n - 2] = 0xFF                                           # >= r
    t = ce.commit_begin(ck, bad)
    with pytest.raises(nmx.NmxError) as e:
        t.finish()
    assert e.value.code == L.E_SCALAR_RANGE
    out, inf = np.zeros(64, np.uint8), np.zeros(1, np.uint8)
    assert L.lib().nmx_commit_finish(12345678, out.ctypes.data, inf.ctypes.data) == L.E_HANDLE
    t = ce.commit_begin(ck, v)
    tk = t.ticket
    want = _pt(ce.commit(ck, v))
    assert L.lib().nmx_commit_finish(tk, None, None) == L.E_ARG          # a bad call leaves the ticket alone
    assert _pt(t.finish()) == want
    assert L.lib().nmx_commit_finish(tk, out.ctypes.data, inf.ctypes.data) == L.E_HANDLE      # retired
    tk2 = ctypes.c_uint64(0)
    z = np.zeros(64, np.uint8)
    assert L.lib().nmx_commit_begin(ck.handle, v.ctypes.data, n + 1, z.ctypes.data, z.ctypes.data, 0, ctypes.byref(tk2)) == L.E_HANDLE
    assert L.lib().nmx_commit_begin(ck.handle, None, 5, z.ctypes.data, z.ctypes.data, 0, ctypes.byref(tk2)) == L.E_ARG
    assert _pt(ce.commit(ck, v)) == want                        # and the library is fine afterwards
    ck.close()


def test_a_begun_commitment_is_ordered_behind_stream_ordered_field_calls(nmx):
    """the folding step's shape: W' = W1 + r W2 (stream-ordered), commit(W') begun at once, another fold and a synchronous
    commitment beside it"""
    import torch
    from nova_amd import fieldvec as fv
    c = R.BN254_G1
    fid = fv.SCALAR_FIELD_OF_CURVE[c.cid]
    n = 1 << 16
    bases = cref.sequential_bases(c, 11, n).copy()
    ck = nmx.CommitmentKey.from_host(c.cid, bases)
    ce = nmx.CommitmentEngine(c.cid)
    prep = cref.Prepared(c.cid, bases, n)
    a, b = C.rand_vec(fid, n, 1), C.rand_vec(fid, n, 2)
    r = C.rand_vec(fid, 1, 3)
    da, db = torch.from_numpy(a.copy()).cuda(), torch.from_numpy(b.copy()).cuda()
    want_w = prep.msm(np.frombuffer(cref.field_axpy(fid, a, b, r, n), np.uint8).reshape(n, 32), n)
    want_a = prep.msm(a, n)
    for _ in range(5):
        w = fv.axpy(fid, da, db, r, async_=True)                # enqueued, not waited for
        t = ce.commit_begin(ck, w)
        w2 = fv.axpy(fid, db, da, r, async_=True)               # the caller goes on
        got_a = _pt(ce.commit(ck, da))
        assert _pt(t.finish()) == want_w
        assert got_a == want_a
        fv.sync()
        del w2
    ck.close()
