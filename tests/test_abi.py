"""C-ABI surface (no GPU): libnova_mi355x.so loads, exports every symbol include/nova_mi355x.h declares, and
refuses to compute without a device instead of falling back to a CPU path."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nova_mi355x.h")


@pytest.fixture(scope="module")
def L():
    from nova_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nmx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(L):
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/nova_mi355x.h but not exported"


def test_python_binding_covers_header(L):
    from nova_amd import _lib
    bound = {s for s in declared_symbols() if getattr(getattr(L, s), "argtypes", None) is not None or s in
             ("nmx_last_error", "nmx_version", "nmx_shutdown", "nmx_device_count", "nmx_cache_clear")}
    assert bound == set(declared_symbols())
    assert L.nmx_version().decode().startswith("nova-mi355x")
    assert _lib.E_NO_DEVICE == -2


def test_no_cpu_fallback_without_device(L):
    """On a machine without a GPU every compute entry point must fail with NMX_E_NO_DEVICE."""
    if L.nmx_device_count() > 0:
        pytest.skip("a HIP device is visible")
    from nova_amd import _lib
    out = np.zeros(64, np.uint8)
    inf = np.zeros(1, np.uint8)
    sc = np.zeros((4, 32), np.uint8)
    sc[:, 0] = 1
    b = np.zeros((4, 64), np.uint8)
    b[:, 0] = 1
    b[:, 32] = 2
    rc = L.nmx_msm(0, sc.ctypes.data, b.ctypes.data, 4, 0, out.ctypes.data, inf.ctypes.data)
    assert rc == _lib.E_NO_DEVICE
    assert b"no HIP device" in L.nmx_last_error()
    assert not out.any()  # a failed call never writes a point
    h = ctypes.c_uint64(0)
    assert L.nmx_bases_generate(0, 1, 8, 0, ctypes.byref(h)) == _lib.E_NO_DEVICE
    assert L.nmx_init(0) == _lib.E_NO_DEVICE


def test_point_sum_is_host_side_and_matches_oracle(L):
    """nmx_point_sum (the G-term combine of a sharded MSM) needs no device: check it against the oracle."""
    from oracle import cref
    from oracle import pyref as R
    from tests import util
    for c in R.CURVES.values():
        n = 24
        bases = cref.sequential_bases(c, 11, n)
        sc = util.random_scalars(c.cid, n)
        parts = []
        for lo, hi in ((0, 7), (7, 7), (7, 24)):  # includes an empty shard -> identity partial
            xy, inf = cref.msm(c.cid, sc[lo:hi], bases[lo:hi], hi - lo)
            parts.append(util.affine_to_partial(c.p, xy, inf))
        buf = np.frombuffer(b"".join(parts), dtype=np.uint8)
        out = np.zeros(64, np.uint8)
        inf = np.zeros(1, np.uint8)
        assert L.nmx_point_sum(c.cid, buf.ctypes.data, 3, out.ctypes.data, inf.ctypes.data) == 0
        assert (out.tobytes(), int(inf[0])) == cref.msm(c.cid, sc, bases, n)


def test_layout_selfcheck_min_n_and_stats_are_host_side(L):
    """nmx_check_layout (the shim's one-time zero-copy self-check), nmx_min_gpu_n and nmx_stats need no device."""
    from nova_amd import _lib
    from oracle import pyref as R
    Rm = 1 << 256
    for c in R.CURVES.values():
        gen = b"".join(((v * Rm) % c.p).to_bytes(32, "little") for v in (c.gx, c.gy))
        s = ((123456789 * Rm) % c.r).to_bytes(32, "little")
        assert L.nmx_check_layout(c.cid, gen, s, 123456789) == 0
        # canonical bytes, a wrong value, swapped coordinates, a non-reduced limb pattern: all rejected
        canon = b"".join(v.to_bytes(32, "little") for v in (c.gx, c.gy))
        assert L.nmx_check_layout(c.cid, canon, s, 123456789) == _lib.E_FORMAT
        assert L.nmx_check_layout(c.cid, gen, s, 123456788) == _lib.E_FORMAT
        assert L.nmx_check_layout(c.cid, gen[32:] + gen[:32], s, 123456789) == _lib.E_FORMAT
        assert L.nmx_check_layout(c.cid, b"\xff" * 64, s, 123456789) == _lib.E_FORMAT
        assert b"Montgomery" in L.nmx_last_error()
        assert L.nmx_min_gpu_n(c.cid) == int(os.environ.get("NMX_MIN_N", 128))
    st = _lib.stats()
    assert len(st) == _lib.STAT_COUNT and all(v >= 0 for v in st)
    assert L.nmx_check_layout(9, gen, s, 1) == _lib.E_ARG


def test_ipa_prove_argument_checks_are_host_side(L):
    """nmx_ipa_prove refuses bad arguments before it touches a device or a key: null callback / null vectors / unknown flags / n not a
    power of two -> NMX_E_ARG; an unknown key handle -> NMX_E_HANDLE (InnerProductArgument::prove, ipa_pc.rs:174-188)."""
    from nova_amd import _lib
    z = np.zeros(64, np.uint8)
    cb = _lib.IPA_TRANSCRIPT_FN(lambda *a: 0)
    null_cb = ctypes.cast(None, _lib.IPA_TRANSCRIPT_FN)
    p = z.ctypes.data
    assert L.nmx_ipa_prove(1, p, p, p, 2, 0, null_cb, None, p, p, None, p) == _lib.E_ARG
    assert L.nmx_ipa_prove(1, None, p, p, 2, 0, cb, None, p, p, None, p) == _lib.E_ARG
    assert L.nmx_ipa_prove(1, p, None, p, 2, 0, cb, None, p, p, None, p) == _lib.E_ARG
    assert L.nmx_ipa_prove(1, p, p, p, 2, 1 << 9, cb, None, p, p, None, p) == _lib.E_ARG          # NMX_ASYNC is not a flag of this call
    assert L.nmx_ipa_prove(1, p, p, p, 6, 0, cb, None, p, p, None, p) == _lib.E_ARG
    assert L.nmx_ipa_prove(1, p, p, p, 0, 0, cb, None, p, p, None, p) == _lib.E_ARG
    assert L.nmx_ipa_prove(1, p, p, p, 4, 0, cb, None, None, p, None, p) == _lib.E_ARG            # rounds > 0 need out_L / out_R
    assert L.nmx_ipa_prove(0xdeadbeef, p, p, p, 4, 0, cb, None, p, p, None, p) == _lib.E_HANDLE

