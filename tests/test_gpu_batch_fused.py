"""-m gpu: DlogGroupExt::batch_vartime_multiscalar_mul (src/provider/traits.rs:82-90) / HyperKZG batch_commit
(src/provider/hyperkzg.rs:593-612, lengths n/2 ... 2: hyperkzg.rs:1085-1100) with the short vectors FUSED into one pipeline
run over the key's window tables (capi.hip batch_impl -> run_msm_batch, DigitSrc::batch_*): every vector against the
oracle's MSM over its own prefix, and against the unfused path (nmx_set_option("no_batch_fuse", 1)), on keys of every
table width (c = 8, 15, 16), handle and slice form, host / device / Montgomery scalars, empty vectors, identity points,
more vectors than one run takes, and the error path."""
import ctypes

import numpy as np
import pytest

from nova_amd import _lib
from oracle import cref
from oracle import pyref as R
from tests import util

pytestmark = pytest.mark.gpu


def as_pair(com):
    return (com.xy, int(com.is_inf))


def expected(c, vecs, host):
    return [cref.msm(c.cid, v, host[:len(v)], len(v)) if len(v) else (bytes(64), 1) for v in vecs]


def fused_delta(f):
    before = _lib.stats()
    r = f()
    now = _lib.stats()
    return r, now[_lib.STAT_FUSED_RUNS] - before[_lib.STAT_FUSED_RUNS], now[_lib.STAT_MSM_CALLS] - before[_lib.STAT_MSM_CALLS]


@pytest.mark.parametrize("lg", [10, 15, 17], ids=["c8", "c15", "c16"])
def test_hyperkzg_batch_shape(nmx, lg):
    """batch_commit of the folded polynomials of a HyperKZG prove over a 2^lg key: lengths n/2 ... 2 (+ the full length and
    an empty vector), random full-width scalars."""
    c = R.BN254_G1
    n = 1 << lg
    host = cref.sequential_bases(c, 31 + lg, n)
    ck = nmx.CommitmentKey.from_host(c.cid, host)
    g = nmx.DlogGroup(c.cid)
    lens = [n] + [n >> i for i in range(1, lg)] + [0]
    vecs = [util.random_scalars(c.cid, m, seed=500 + j) for j, m in enumerate(lens)]
    exp = expected(c, vecs, host)
    got, fused, calls = fused_delta(lambda: [as_pair(x) for x in g.batch_vartime_multiscalar_mul(vecs, ck)])
    assert got == exp
    # 11, 17 and 18 vectors: one run on a key below 2^14 points; above, the vectors of <= 2^13 pairs run fused on the key's
    # c = 8 prefix tables (2^7 buckets per set instead of 2^14 / 2^15) and the longer ones fused on the key itself: two runs
    assert fused == (1 if lg < 14 else 2)
    assert calls == len(vecs)
    assert _lib.lib().nmx_set_option(b"no_batch_fuse", 1) == 0
    try:
        got2, fused2, _ = fused_delta(lambda: [as_pair(x) for x in g.batch_vartime_multiscalar_mul(vecs, ck)])
    finally:
        _lib.lib().nmx_set_option(b"no_batch_fuse", 0)
    assert got2 == exp and fused2 == 0
    ck.close()


@pytest.mark.parametrize("c", list(R.CURVES.values()), ids=lambda c: c.name)
def test_ragged_sets_identity_points_and_layouts(nmx, c):
    """Every scalar set of the reference's matrix in one fused batch, an identity point inside the key, the slice form
    (bases pointer, served by the slice cache), Montgomery-form scalars + bases, and HBM-resident scalars."""
    import torch
    n = 3000
    host = cref.sequential_bases(c, 77, n).copy()
    host[11] = 0
    kinds = ["random", "equal", "zero_rm1", "pm_small", "u1", "u10", "u16", "u32", "u64"]
    lens = [3000, 1500, 700, 0, 1, 2, 17, 333, 2999]
    vecs = [util.scalar_set(c.cid, m, kinds[j], seed=40 + j) if m else np.zeros((0, 32), np.uint8) for j, m in enumerate(lens)]
    exp = expected(c, vecs, host)
    g = nmx.DlogGroup(c.cid)
    ck = nmx.CommitmentKey.from_host(c.cid, host)
    got, fused, _ = fused_delta(lambda: [as_pair(x) for x in g.batch_vartime_multiscalar_mul(vecs, ck)])
    assert got == exp and fused == 1
    # HBM-resident vectors
    dv = [torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in vecs]
    got, fused, _ = fused_delta(lambda: [as_pair(x) for x in g.batch_vartime_multiscalar_mul(dv, ck)])
    assert got == exp and fused == 1
    ck.close()
    # slice form: pageable host bases.  First and second sight: the array is resident without window tables (no fused
    # run: one MSM per vector); from the third use on it has them and the batch is fused
    for use in range(4):
        got, fused, _ = fused_delta(lambda: [as_pair(x) for x in g.batch_vartime_multiscalar_mul(vecs, host)])
        assert got == exp and fused == (1 if use >= 2 else 0), use
    # Montgomery layouts (what the Rust shim passes: INTEGRATION.md)
    hm = util.to_mont_bases(c.cid, host)
    vm = [util.to_mont_scalars(c.cid, v) for v in vecs]
    got = [as_pair(x) for x in g.batch_vartime_multiscalar_mul(vm, hm, mont=True)]
    assert got == exp
    _lib.lib().nmx_cache_clear()


def test_more_vectors_than_one_run_takes(nmx):
    """A c = 16 key leaves 5 key bits for vector ids: 40 vectors would be fused runs of 32 and 8 -- but vectors of <= 2^13 pairs
    run on the key's c = 8 prefix tables, which take 256 per run: ONE run (with option prefix_tables = 0: 32 + 8); a c = 8 key
    takes 256: 300 tiny vectors -> 256 + 44."""
    c = R.BN254_G1
    g = nmx.DlogGroup(c.cid)
    n = 1 << 17
    host = cref.sequential_bases(c, 5, n)
    ck = nmx.CommitmentKey.from_host(c.cid, host)
    lens = [(j * 7919) % 5000 + 1 for j in range(40)]
    vecs = [util.random_scalars(c.cid, m, seed=900 + j) for j, m in enumerate(lens)]
    got, fused, calls = fused_delta(lambda: [as_pair(x) for x in g.batch_vartime_multiscalar_mul(vecs, ck)])
    assert got == expected(c, vecs, host)
    assert fused == 1 and calls == 40
    ck.close()
    L = _lib.lib()
    assert L.nmx_set_option(b"prefix_tables", 0) == 0
    try:
        ck = nmx.CommitmentKey.from_host(c.cid, host)
        got, fused, calls = fused_delta(lambda: [as_pair(x) for x in g.batch_vartime_multiscalar_mul(vecs, ck)])
        assert got == expected(c, vecs, host)
        assert fused == 2 and calls == 40
        ck.close()
    finally:
        assert L.nmx_set_option(b"prefix_tables", 2) == 0
    n = 512
    host = cref.sequential_bases(c, 6, n)
    ck = nmx.CommitmentKey.from_host(c.cid, host)
    lens = [(j * 31) % 9 for j in range(300)]
    vecs = [util.random_scalars(c.cid, m, seed=1300 + j) for j, m in enumerate(lens)]
    got, fused, calls = fused_delta(lambda: [as_pair(x) for x in g.batch_vartime_multiscalar_mul(vecs, ck)])
    assert got == expected(c, vecs, host) and fused == 2 and calls == 300
    ck.close()


@pytest.mark.parametrize("width", [17, 20])
def test_fused_runs_on_wide_tables(nmx, width):
    """Keys of 2^20+ points carry c = 17 (16 vectors per run, 20 key bits: 1024 x 1024 bins) or c = 20 tables (2 per run);
    forced here on a 2^13-point key so the oracle stays cheap.  7 vectors -> one run at c = 17; 2 + 2 + 2 + 1 at c = 20."""
    L = _lib.lib()
    c = R.BN254_G1
    n = 1 << 13
    host = cref.sequential_bases(c, 99, n).copy()
    host[4000] = 0
    assert L.nmx_set_window_bits(width) == 0
    try:
        ck = nmx.CommitmentKey.from_host(c.cid, host)
        g = nmx.DlogGroup(c.cid)
        lens = [n, n - 1, 5000, 4097, 300, 1, 0]
        kinds = ["random", "zero_rm1", "equal", "pm_small", "random", "random", "random"]
        vecs = [util.scalar_set(c.cid, m, kinds[j], seed=70 + j) if m else np.zeros((0, 32), np.uint8) for j, m in enumerate(lens)]
        got, fused, calls = fused_delta(lambda: [as_pair(x) for x in g.batch_vartime_multiscalar_mul(vecs, ck)])
        assert got == expected(c, vecs, host)
        assert fused == (1 if width == 17 else 3) and calls == 7
        ck.close()
    finally:
        assert L.nmx_set_window_bits(0) == 0


def test_fused_run_on_the_segment_path(nmx):
    """Long vectors fused: 7.8 M sorted entries take the segment-balanced accumulate (msm_seg.hpp) over 8 bucket sets
    (2^18 buckets: FinalSegFn one lane per bucket), witness-like and skewed scalars included."""
    c = R.BN254_G1
    n = 1 << 18
    host = cref.sequential_bases(c, 3, n)
    ck = nmx.CommitmentKey.from_host(c.cid, host)
    g = nmx.DlogGroup(c.cid)
    lens = [n, n >> 1, n >> 2, n >> 3, 5, 0]
    vecs = [util.random_scalars(c.cid, lens[0], seed=1), util.witness_like(c.cid, lens[1], seed=2),
            util.scalar_set(c.cid, lens[2], "equal", seed=3), util.scalar_set(c.cid, lens[3], "zero_rm1", seed=4),
            util.random_scalars(c.cid, 5, seed=5), np.zeros((0, 32), np.uint8)]
    prep = cref.Prepared(c.cid, host, n)
    exp = [prep.msm(np.ascontiguousarray(v), len(v)) if len(v) else (bytes(64), 1) for v in vecs]
    got, fused, calls = fused_delta(lambda: [as_pair(x) for x in g.batch_vartime_multiscalar_mul(vecs, ck)])
    assert got == exp
    assert fused == 2 and calls == 6   # the four long vectors fused on the key, the 5-pair one (and the empty one) on its c = 8 prefix tables
    ck.close()


def test_out_of_range_scalar_fails_the_batch_and_leaves_out_untouched(nmx):
    c = R.BN254_G1
    n = 2048
    host = cref.sequential_bases(c, 8, n)
    ck = nmx.CommitmentKey.from_host(c.cid, host)
    vecs = [np.ascontiguousarray(util.random_scalars(c.cid, m, seed=60 + m)) for m in (2048, 100, 10)]
    vecs[2][3] = 0xff  # >= r
    k = len(vecs)
    ptrs = (ctypes.c_void_p * k)(*[v.ctypes.data for v in vecs])
    lens = (ctypes.c_size_t * k)(*[len(v) for v in vecs])
    out = np.full((k, 64), 0xAB, np.uint8)
    inf = np.full(k, 0xCD, np.uint8)
    rc = _lib.lib().nmx_msm_batch_handle(ck.handle, ptrs, lens, k, 0, out.ctypes.data, inf.ctypes.data)
    assert rc == _lib.E_SCALAR_RANGE
    assert (out == 0xAB).all() and (inf == 0xCD).all()
    ck.close()
