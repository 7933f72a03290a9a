"""C-ABI robustness under rayon-style concurrency (-m gpu).  The trait methods are static and are called from many
worker threads at once (/root/reference/src/r1cs/mod.rs:509-512, hyperkzg.rs:1062-1065, nova/mod.rs:862-881); a handle
may be unregistered, or a cached array evicted, on one thread while MSMs over it run on others.  Every call must
either return the right point or a clean NMX_E_HANDLE -- never a wrong point, a crash or a use-after-free."""
import ctypes
import threading

import numpy as np
import pytest

from oracle import cref
from oracle import pyref as R
from tests import util

pytestmark = pytest.mark.gpu


def test_register_msm_unregister_from_ten_threads(nmx):
    from nova_amd import _lib
    L = _lib.lib()
    c = R.BN254_G1
    n = 6000
    bases = cref.sequential_bases(c, 2024, n)
    lens = [n, 4097, 300, 5999]
    scs = [util.random_scalars(c.cid, m, seed=m) for m in lens]
    exp = [cref.msm(c.cid, s, bases[:m], m) for s, m in zip(scs, lens)]
    shared = {"h": 0}
    lock = threading.Lock()
    errors = []
    stop = threading.Event()
    counts = {"ok": 0, "stale": 0, "cycles": 0}

    def registrar():
        """register -> publish -> (others use it) -> unregister while they may still be running -> repeat"""
        for _ in range(40):
            h = ctypes.c_uint64(0)
            rc = L.nmx_bases_register(c.cid, bases.ctypes.data, n, _lib.BASES_PRECOMPUTE, ctypes.byref(h))
            if rc != 0:
                errors.append(("register", rc, L.nmx_last_error().decode()))
                break
            with lock:
                old, shared["h"] = shared["h"], h.value
            if old:
                rc = L.nmx_bases_unregister(old)     # callers may be in the middle of an MSM over `old`
                if rc != 0:
                    errors.append(("unregister", rc))
            counts["cycles"] += 1
        stop.set()

    def caller(tid):
        out = np.zeros(64, np.uint8)
        inf = np.zeros(1, np.uint8)
        k = 0
        while not stop.is_set():
            j = (tid + k) % len(lens)
            k += 1
            with lock:
                h = shared["h"]
            if not h:
                continue
            out[:] = 0xAB
            rc = L.nmx_msm_handle(h, 0, scs[j].ctypes.data, lens[j], 0, out.ctypes.data, inf.ctypes.data)
            if rc == 0:
                if (out.tobytes(), int(inf[0])) != exp[j]:
                    errors.append(("wrong point", tid, j))
                counts["ok"] += 1
            elif rc == _lib.E_HANDLE:                # unregistered between the read of `h` and the call: clean error,
                if not (out == 0xAB).all():          # and a failed call never writes a point
                    errors.append(("failed call wrote output", tid))
                counts["stale"] += 1
            else:
                errors.append(("msm", rc, L.nmx_last_error().decode()))

    def slice_caller(tid):
        """the slice form over the same array, with the cache being cleared under it"""
        g = nmx.DlogGroup(c.cid)
        k = 0
        while not stop.is_set():
            j = (tid + k) % len(lens)
            k += 1
            got = g.vartime_multiscalar_mul(scs[j], bases[: lens[j]])
            if (got.xy, int(got.is_inf)) != exp[j]:
                errors.append(("wrong point (slice form)", tid, j))
            if k % 5 == 0:
                L.nmx_cache_clear()

    def batch_caller(tid):
        """whole batches (fused runs + worker threads inside the library) over a handle that may vanish under them"""
        k = len(lens)
        ptrs = (ctypes.c_void_p * k)(*[v.ctypes.data for v in scs])
        ls = (ctypes.c_size_t * k)(*lens)
        out = np.zeros((k, 64), np.uint8)
        inf = np.zeros(k, np.uint8)
        while not stop.is_set():
            with lock:
                h = shared["h"]
            if not h:
                continue
            out[:] = 0xAB
            rc = L.nmx_msm_batch_handle(h, ptrs, ls, k, 0, out.ctypes.data, inf.ctypes.data)
            if rc == 0:
                if [(out[j].tobytes(), int(inf[j])) for j in range(k)] != exp:
                    errors.append(("wrong point (batch)", tid))
                counts["ok"] += 1
            elif rc == _lib.E_HANDLE:
                if not (out == 0xAB).all():
                    errors.append(("failed batch wrote output", tid))
                counts["stale"] += 1
            else:
                errors.append(("batch", rc, L.nmx_last_error().decode()))

    ths = [threading.Thread(target=registrar)]
    ths += [threading.Thread(target=caller, args=(t,)) for t in range(5)]
    ths += [threading.Thread(target=batch_caller, args=(t,)) for t in range(2)]
    ths += [threading.Thread(target=slice_caller, args=(t,)) for t in range(2)]
    [t.start() for t in ths]
    [t.join(timeout=300) for t in ths]
    assert not any(t.is_alive() for t in ths), "a thread hung"
    assert not errors, errors[:5]
    assert counts["cycles"] == 40 and counts["ok"] > 40
    with lock:
        if shared["h"]:
            assert L.nmx_bases_unregister(shared["h"]) == 0
    assert L.nmx_cache_clear() == 0


def test_slice_bounds_cannot_wrap(nmx):
    """offset + n is checked without overflow: offset = SIZE_MAX, n = 1 must be NMX_E_HANDLE, not an HBM read."""
    from nova_amd import _lib
    L = _lib.lib()
    c = R.BN254_G1
    ck = nmx.CommitmentKey.from_host(c.cid, cref.sequential_bases(c, 1, 64))
    out = np.zeros(64, np.uint8)
    inf = np.zeros(1, np.uint8)
    sc = util.random_scalars(c.cid, 4)
    big = ctypes.c_size_t(-1).value
    for off, n in ((big, 1), (big - 2, 4), (64, 1), (1, big)):
        assert L.nmx_msm_handle(ck.handle, off, sc.ctypes.data, n, 0, out.ctypes.data, inf.ctypes.data) == _lib.E_HANDLE
        assert L.nmx_bases_read(ck.handle, off, n, out.ctypes.data) == _lib.E_HANDLE
    s64 = np.ones(4, np.uint64)
    assert L.nmx_msm_u64_handle(ck.handle, big, s64.ctypes.data, 1, 8, 0, out.ctypes.data, inf.ctypes.data) == _lib.E_HANDLE
    assert not out.any()
    ck.close()


def test_sharded_keys_and_long_cached_arrays_from_many_threads(nmx):
    """Round 3's concurrent paths: MSMs over a key sharded 3-way (a host thread per shard inside every call), slice-form
    calls over a LONG array (the rolling content check runs on a second host thread, the window tables arrive on the third
    use while other threads are mid-call, the cache is cleared under them) -- from six threads at once."""
    from nova_amd import _lib
    L = _lib.lib()
    c = R.BN254_G1
    n = 1 << 17
    bases = cref.sequential_bases(c, 90210, n)
    prep = cref.Prepared(c.cid, bases, n)
    lens = [n, 70001, 66000, 4500]
    scs = [util.random_scalars(c.cid, m, seed=7 * m) for m in lens]
    exp = [prep.msm(s, m) for s, m in zip(scs, lens)]
    assert nmx.init_devices(3, oversubscribe=True) == 3
    assert L.nmx_set_option(b"shard_min_n", 4096) == 0
    errors = []
    try:
        ck = nmx.CommitmentKey.from_host(c.cid, bases)       # three shards
        g = nmx.DlogGroup(c.cid)

        def handle_caller(tid):
            for k in range(12):
                j = (tid + k) % len(lens)
                got = g.vartime_multiscalar_mul(scs[j], ck)
                if (got.xy, int(got.is_inf)) != exp[j]:
                    errors.append(("sharded handle", tid, j))

        def slice_caller(tid):
            for k in range(12):
                j = (tid + k) % len(lens)
                got = g.vartime_multiscalar_mul(scs[j], bases[: lens[j]])
                if (got.xy, int(got.is_inf)) != exp[j]:
                    errors.append(("slice form", tid, j))
                if tid == 0 and k % 5 == 4:
                    L.nmx_cache_clear()

        ths = [threading.Thread(target=handle_caller, args=(t,)) for t in range(3)]
        ths += [threading.Thread(target=slice_caller, args=(t,)) for t in range(3)]
        [t.start() for t in ths]
        [t.join(timeout=300) for t in ths]
        assert not any(t.is_alive() for t in ths), "a thread hung"
        assert not errors, errors[:5]
        ck.close()
    finally:
        assert nmx.init_devices(1) == 1
        assert L.nmx_set_option(b"shard_min_n", 1 << 20) == 0
        assert L.nmx_cache_clear() == 0


def test_async_nifs_chains_from_several_threads(nmx):
    """NMX_ASYNC under concurrency: four host threads (rayon workers: the primary and the secondary NIFS of src/nova/mod.rs:862-881
    run side by side) each enqueue commit_T's chain and the fold without waiting, commit in between and compare with the
    synchronous results -- contexts are leased and returned with work still pending on their streams all the time, the arena
    that holds Z is re-carved by whoever leases the context next, and every thread's ordering is its own."""
    import torch
    from nova_amd import fieldvec as fv
    from tests import fv_common as C
    cid = 0
    fid = fv.SCALAR_FIELD_OF_CURVE[cid]
    n = 30000
    ck = nmx.CommitmentKey.generate(cid, n, k0=11)
    ce = nmx.CommitmentEngine(cid)
    mats = [fv.SparseMatrix(fid, *C.random_csr(fid, n, n, 70 + j), n) for j in range(3)]
    data = []
    for t in range(4):
        hW1, hW2, hE = (C.rand_vec(fid, n, 100 * t + s) for s in (1, 2, 3))
        u, r = C.rand_vec(fid, 1, 100 * t + 4), C.rand_vec(fid, 1, 100 * t + 5)
        W1, W2, E = (torch.from_numpy(h.copy()).cuda() for h in (hW1, hW2, hE))
        torch.cuda.synchronize()
        T0 = fv.r1cs_cross_term(mats[0], mats[1], mats[2], W1, W2, E, u)
        c0 = ce.commit(ck, T0, r)
        Wf0, Ef0 = fv.nifs_fold(fid, W1, W2, E, T0, r)
        data.append((W1, W2, E, u, r, T0, (c0.xy, c0.is_inf), Wf0, Ef0))
    errors = []

    def worker(t):
        W1, W2, E, u, r, T0, c0, Wf0, Ef0 = data[t]
        try:
            for it in range(25):
                T = fv.r1cs_cross_term(mats[0], mats[1], mats[2], W1, W2, E, u, async_=True)
                cm = ce.commit(ck, T, r)
                if (cm.xy, cm.is_inf) != c0 or not torch.equal(T, T0):
                    errors.append((t, it, "commit_T"))
                Wf, Ef = fv.nifs_fold(fid, W1, W2, E, T, r, async_=True)
                if it % 3 == 0:
                    small = ce.commit(ck, T[:100].contiguous(), r)      # a synchronous call behind the fold: orders it as well
                    assert small is not None
                else:
                    fv.sync()
                if not (torch.equal(Wf, Wf0) and torch.equal(Ef, Ef0)):
                    errors.append((t, it, "fold"))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:5]
    for m in mats:
        m.close()
    ck.close()
