"""The drop-in boundary as the reference calls it (-m gpu): `vartime_multiscalar_mul(&scalars, &ck.ck[..n])` passes a
SLICE and no handle (/root/reference/src/provider/pedersen.rs:263-270, hyperkzg.rs:584-591, traits.rs:79,
blitzar.rs:7-20).  The library's slice cache must make that signature reach the resident-key path: one upload per
array whatever the prefix length, window tables included, results bit-exact against the oracle; and it must never
serve stale points (content fingerprints), leak under eviction, or upload twice under concurrent first use.
"""
import ctypes
import threading

import numpy as np
import pytest

from oracle import cref
from oracle import pyref as R
from tests import util

pytestmark = pytest.mark.gpu


def as_pair(com):
    return (com.xy, int(com.is_inf))


@pytest.fixture()
def fresh(nmx):
    from nova_amd import _lib
    L = _lib.lib()
    assert L.nmx_cache_clear() == 0
    yield _lib
    assert L.nmx_cache_clear() == 0


def delta(_lib, before):
    now = _lib.stats()
    return [a - b for a, b in zip(now, before)]


def test_two_prefixes_one_upload(nmx, fresh):
    """Done-criterion of the boundary row: nmx_msm twice on the same host array with two prefix lengths -> ONE upload,
    both results == oracle (the second call moves no base bytes)."""
    S = fresh
    c = R.BN254_G1
    n = 1 << 16
    bases = cref.sequential_bases(c, 77, n)
    g = nmx.DlogGroup(c.cid)
    before = S.stats()
    for m in (n, 13058, n - 1, 4096, 129):          # longest first, then the shapes callers use (prove_step N, n-1 openings)
        sc = util.random_scalars(c.cid, m, seed=m)
        assert as_pair(g.vartime_multiscalar_mul(sc, bases[:m])) == cref.msm(c.cid, sc, bases[:m], m), m
    d = delta(S, before)
    assert d[S.STAT_CACHE_UPLOADS] == 1 and d[S.STAT_CACHE_REGROWS] == 0
    assert d[S.STAT_CACHE_HITS] == 4 and d[S.STAT_UNCACHED_CALLS] == 0
    assert d[S.STAT_BASE_BYTES_H2D] == 64 * n
    now = S.stats()
    assert now[S.STAT_CACHE_ENTRIES] == 1 and now[S.STAT_CACHE_BYTES] >= 64 * n * 2   # key + window tables


def test_growing_prefix_regrows(nmx, fresh):
    S = fresh
    c = R.GRUMPKIN
    n = 20000
    bases = cref.sequential_bases(c, 5, n)
    g = nmx.DlogGroup(c.cid)
    before = S.stats()
    for m in (5000, 5000, 10538, n):
        sc = util.random_scalars(c.cid, m, seed=m)
        assert as_pair(g.vartime_multiscalar_mul(sc, bases[:m])) == cref.msm(c.cid, sc, bases[:m], m)
    d = delta(S, before)
    assert d[S.STAT_CACHE_UPLOADS] == 3 and d[S.STAT_CACHE_REGROWS] == 2 and d[S.STAT_CACHE_HITS] == 1
    assert S.stats()[S.STAT_CACHE_ENTRIES] == 1


def test_interior_slice_small_and_batch_forms_hit(nmx, fresh):
    """`&ck.ck[a..b]` (commit_small_range, pedersen.rs:285-305), msm_small and batch forms over one resident array."""
    S = fresh
    c = R.PALLAS
    n = 9000
    bases = cref.sequential_bases(c, 21, n)
    g = nmx.DlogGroup(c.cid)
    sc = util.random_scalars(c.cid, n, seed=3)
    before = S.stats()
    assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == cref.msm(c.cid, sc, bases, n)
    a, b = 1234, 8000
    assert as_pair(g.vartime_multiscalar_mul(sc[: b - a], bases[a:b])) == cref.msm(c.cid, sc[: b - a], bases[a:b], b - a)
    s64 = util.small_scalars(3000, 20)
    assert as_pair(g.vartime_multiscalar_mul_small(s64, bases[:3000])) == cref.msm_u64(c.cid, s64, bases[:3000], 3000, 20)
    lens = [n, n // 2, 17, 0, 4096]
    vecs = [util.random_scalars(c.cid, m, seed=50 + j) for j, m in enumerate(lens)]
    got = [as_pair(x) for x in g.batch_vartime_multiscalar_mul(vecs, bases)]
    assert got == cref.msm_batch(c.cid, [v.tobytes() for v in vecs], bases, n)
    d = delta(S, before)
    assert d[S.STAT_CACHE_UPLOADS] == 1 and d[S.STAT_CACHE_HITS] == 3 and d[S.STAT_UNCACHED_CALLS] == 0


@pytest.mark.parametrize("n", [1500, 50000])
def test_reused_address_with_new_content_is_detected(nmx, fresh, n):
    """A Vec freed and reallocated at the same address must not be served from the old resident copy: overwrite the
    array in place with a different key (every point changes), same address, same length."""
    S = fresh
    c = R.BN254_G1
    g = nmx.DlogGroup(c.cid)
    bases = cref.sequential_bases(c, 1000, n).copy()
    sc = util.random_scalars(c.cid, n, seed=9)
    assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == cref.msm(c.cid, sc, bases, n)
    addr = bases.ctypes.data
    bases[:] = cref.sequential_bases(c, 500000, n)
    assert bases.ctypes.data == addr
    before = S.stats()
    assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == cref.msm(c.cid, sc, bases, n)
    d = delta(S, before)
    assert d[S.STAT_CACHE_UPLOADS] == 1 and d[S.STAT_CACHE_HITS] == 0
    if n <= 2048:   # short arrays are verified in full: a single changed point is caught at once
        bases[n // 3] = cref.sequential_bases(c, 42, 1)[0]
        assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == cref.msm(c.cid, sc, bases, n)
    else:           # long arrays: nmx_cache_invalidate is the contract for in-place edits (immediate) ...
        bases[n // 3] = cref.sequential_bases(c, 42, 1)[0]
        assert S.lib().nmx_cache_invalidate(bases.ctypes.data) == 0
        assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == cref.msm(c.cid, sc, bases, n)


@pytest.mark.parametrize("n,where", [(50000, 0.37), (1 << 17, 0.999), (50000, 0.0001)])
def test_single_point_edit_without_invalidate_first_call_is_right(nmx, fresh, n, where):
    """The trait is a pure function of the slice's CONTENTS (/root/reference/src/provider/traits.rs:79): a caller that rewrites
    ONE point of a long cached array in place -- without nmx_cache_invalidate -- must get the new answer from the VERY NEXT
    call (the rolling window let up to 15 calls return the old key's commitment).  Default verification
    re-hashes the caller's whole slice on pool workers while the GPU runs the MSM; the stale result is discarded, the entry
    dropped, the call repeated on a fresh upload."""
    S = fresh
    c = R.BN254_G1
    g = nmx.DlogGroup(c.cid)
    bases = cref.sequential_bases(c, 4242, n).copy()
    sc = util.random_scalars(c.cid, n, seed=19)
    old = cref.msm(c.cid, sc, bases, n)
    assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == old
    assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == old
    j = min(n - 2, max(1, int(n * where)))          # not the first / last point (those are in the quick look of every call)
    bases[j] = cref.sequential_bases(c, 987654, 1)[0]
    new = cref.msm(c.cid, sc, bases, n)
    assert new != old
    before = S.stats()
    assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == new, "the first call after the edit returned the old key's commitment"
    d = delta(S, before)
    assert d[S.STAT_CACHE_STALE] == 1 and d[S.STAT_CACHE_UPLOADS] == 1
    # an interior slice that contains the edited point, and one that does not, both right at once
    lo, hi = max(0, j - 3000), min(n, j + 3000)
    assert as_pair(g.vartime_multiscalar_mul(sc[:hi - lo], bases[lo:hi])) == cref.msm(c.cid, sc[:hi - lo], bases[lo:hi], hi - lo)
    for _ in range(3):                              # and it stays right, served from the cache again
        assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == new
    assert delta(S, before)[S.STAT_CACHE_UPLOADS] == 1


@pytest.mark.parametrize("n,where", [(50000, 0.37), (1 << 17, 0.999)])
def test_rolling_verification_option_catches_an_edit_within_bounded_calls(nmx, fresh, n, where):
    """nmx_set_option("cache_verify", 1) -- for callers that register immutable keys: a rolling window of max(4096, n/16)
    consecutive points per call instead of the whole slice; one edited point anywhere in a long array is then noticed within
    16 calls (the call that notices drops the entry, re-uploads and returns the RIGHT point); until then the answers are those
    of the resident (old) key, never anything else."""
    S = fresh
    L = S.lib()
    assert L.nmx_set_option(b"cache_verify", 1) == 0
    try:
        c = R.BN254_G1
        g = nmx.DlogGroup(c.cid)
        bases = cref.sequential_bases(c, 4242, n).copy()
        sc = util.random_scalars(c.cid, n, seed=19)
        old = cref.msm(c.cid, sc, bases, n)
        assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == old
        j = min(n - 2, max(1, int(n * where)))
        bases[j] = cref.sequential_bases(c, 987654, 1)[0]
        new = cref.msm(c.cid, sc, bases, n)
        before = S.stats()
        seen_new = None
        for call in range(1, 18):
            got = as_pair(g.vartime_multiscalar_mul(sc, bases))
            assert got in (old, new), call
            if got == new:
                seen_new = call
                break
        assert seen_new is not None and seen_new <= 17, "the rolling check never reached the edited point"
        d = delta(S, before)
        assert d[S.STAT_CACHE_STALE] == 1 and d[S.STAT_CACHE_UPLOADS] == 1
    finally:
        assert L.nmx_set_option(b"cache_verify", 0) == 0


def test_interior_slice_after_address_reuse(nmx, fresh):
    """a short interior slice (no sampled grid point inside it in round 2) of an array whose memory was
    reused for another key: its first and last points are hashed on every call, so the stale copy is never used."""
    S = fresh
    c = R.GRUMPKIN
    g = nmx.DlogGroup(c.cid)
    n = 40000
    bases = cref.sequential_bases(c, 31, n).copy()
    sc = util.random_scalars(c.cid, n, seed=2)
    assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == cref.msm(c.cid, sc, bases, n)
    bases[:] = cref.sequential_bases(c, 999999, n)    # same address, another key
    for a, m in ((1234, 5), (20001, 130), (39990, 10)):
        got = g.vartime_multiscalar_mul(sc[:m], bases[a:a + m])
        assert as_pair(got) == cref.msm(c.cid, sc[:m], bases[a:a + m], m), (a, m)


def test_ipa_shaped_churn_builds_no_tables(nmx, fresh):
    """IPA's prover loop (/root/reference/src/provider/ipa_pc.rs:212-230 with pedersen.rs:461-497): every round clones the key
    into two FRESH Vecs -- `ck_R.combine(&ck_c)` and `ck_L.combine(&ck_c)`, n/2 + 1 points each -- commits once to each, then
    folds the key into yet another fresh Vec of half the length.  Every MSM therefore sees an array for the first time: the
    slice cache uploads it WITHOUT window tables (a table build per array would cost more than the one MSM that uses it),
    never hits, and evicts by LRU; results == oracle."""
    import time
    S = fresh
    c = R.PALLAS
    g = nmx.DlogGroup(c.cid)
    n = 1 << 14
    before = S.stats()
    t0 = time.perf_counter()
    rounds = 0
    key = cref.sequential_bases(c, 777, n)
    ck_c = cref.sequential_bases(c, 5, 1)
    while n >= 256:
        h = n // 2
        a = util.random_scalars(c.cid, h + 1, seed=n)
        for half in (key[h:n], key[:h]):                       # c_L = <a_L, ck_R> + <a_L, b_R> c ; c_R likewise
            fresh_vec = np.concatenate([half, ck_c])           # combine(): a new allocation every time
            assert as_pair(g.vartime_multiscalar_mul(a, fresh_vec)) == cref.msm(c.cid, a, fresh_vec, h + 1)
        key = cref.sequential_bases(c, 1000 + n, h)            # ck.fold(): new points, new allocation
        n = h
        rounds += 1
    dt = time.perf_counter() - t0
    d = delta(S, before)
    assert d[S.STAT_CACHE_UPLOADS] == 2 * rounds and d[S.STAT_CACHE_HITS] == 0 and d[S.STAT_TABLE_FALLBACKS] == 0
    now = S.stats()
    # nothing resident carries tables: the cache holds plain keys only (64 B per point; at most 32 entries)
    assert now[S.STAT_CACHE_BYTES] <= 64 * 2 * ((1 << 14) + 64)
    print(f"IPA-shaped churn: {rounds} rounds, 2 first-sight MSMs each ({(1 << 13) + 1} .. 129 points), {dt * 1e3:.1f} ms incl. the oracle compares")


def test_tables_arrive_on_third_use_and_respect_the_budget(nmx, fresh):
    S = fresh
    L = S.lib()
    c = R.BN254_G1
    g = nmx.DlogGroup(c.cid)
    n = 1 << 15
    bases = cref.sequential_bases(c, 5150, n).copy()
    sc = util.random_scalars(c.cid, n, seed=1)
    exp = cref.msm(c.cid, sc, bases, n)
    sizes = []
    for _ in range(4):
        assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == exp
        sizes.append(S.stats()[S.STAT_CACHE_BYTES])
    assert sizes[0] == sizes[1] == 64 * n and sizes[2] == sizes[3] and sizes[2] >= 64 * n * 8   # plain, plain, tables, tables
    # a budget the tables do not fit: the array stays resident WITHOUT them and every call still runs on the GPU
    assert L.nmx_cache_clear() == 0
    assert L.nmx_cache_configure(4 * 64 * n, 0, 0) == 0
    try:
        fb = S.stats()[S.STAT_TABLE_FALLBACKS]
        for _ in range(4):
            assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == exp
        now = S.stats()
        assert now[S.STAT_CACHE_BYTES] == 64 * n and now[S.STAT_TABLE_FALLBACKS] == fb + 1 and now[S.STAT_CACHE_ENTRIES] == 1
    finally:
        import torch
        assert L.nmx_cache_configure(torch.cuda.get_device_properties(0).total_memory // 4, 0, 0) == 0


def test_short_arrays_and_nocache_bypass(nmx, fresh):
    S = fresh
    c = R.VESTA
    g = nmx.DlogGroup(c.cid)
    assert g.min_gpu_n() == 128
    before = S.stats()
    for n in (1, 2, 16, 127):                       # below the cache's min_n: one-shot upload, plain path
        bases = cref.sequential_bases(c, 3, n)
        sc = util.random_scalars(c.cid, n, seed=n)
        assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == cref.msm(c.cid, sc, bases, n)
    n = 5000
    bases = cref.sequential_bases(c, 3, n)
    sc = util.random_scalars(c.cid, n, seed=n)
    assert as_pair(g.vartime_multiscalar_mul(sc, bases, nocache=True)) == cref.msm(c.cid, sc, bases, n)
    d = delta(S, before)
    assert d[S.STAT_UNCACHED_CALLS] == 5 and d[S.STAT_CACHE_UPLOADS] == 0
    assert S.stats()[S.STAT_CACHE_ENTRIES] == 0


def test_montgomery_zero_copy_layout_at_2p16(nmx, fresh):
    """The documented default of the shim: NMX_BASES_MONT | NMX_SCALARS_MONT straight from halo2curves' in-memory
    limbs (x * 2^256 mod p, little-endian 4 x u64), through the slice cache, at 2^16."""
    S = fresh
    c = R.BN254_G1
    n = 1 << 16
    bases = cref.sequential_bases(c, 31337, n)
    sc = util.random_scalars(c.cid, n, seed=4)
    Rm = 1 << 256

    def mont(rows, mod):
        out = np.zeros_like(rows)
        for i, row in enumerate(rows):
            out[i] = np.frombuffer(((int.from_bytes(bytes(row), "little") * Rm) % mod).to_bytes(32, "little"), np.uint8)
        return out

    bm = mont(bases.reshape(-1, 32), c.p).reshape(n, 64)
    sm = mont(sc, c.r)
    g = nmx.DlogGroup(c.cid)
    exp = cref.msm(c.cid, sc, bases, n)
    before = S.stats()
    assert as_pair(g.vartime_multiscalar_mul(sm, bm, mont=True)) == exp
    assert as_pair(g.vartime_multiscalar_mul(sm[:40000], bm[:40000], mont=True)) == cref.msm(c.cid, sc[:40000], bases[:40000], 40000)
    d = delta(S, before)
    assert d[S.STAT_CACHE_UPLOADS] == 1 and d[S.STAT_CACHE_HITS] == 1
    # the same address in the OTHER layout is a different key (the flag is part of the identity)
    assert as_pair(g.vartime_multiscalar_mul(sc, bases)) == exp
    # layout self-check the shim runs once per curve: raw generator + Scalar::from(7)
    for cc in R.CURVES.values():
        gen = b"".join(((v * Rm) % cc.p).to_bytes(32, "little") for v in (cc.gx, cc.gy))
        seven = ((7 * Rm) % cc.r).to_bytes(32, "little")
        assert S.lib().nmx_check_layout(cc.cid, gen, seven, 7) == 0
        assert S.lib().nmx_check_layout(cc.cid, gen, (7).to_bytes(32, "little"), 7) == S.E_FORMAT


def test_eviction_under_budget_and_clear(nmx, fresh):
    S = fresh
    L = S.lib()
    c = R.BN254_G1
    g = nmx.DlogGroup(c.cid)
    n = 4096
    arrays = [cref.sequential_bases(c, 10000 * (j + 1), n) for j in range(4)]
    sc = util.random_scalars(c.cid, n, seed=1)
    exp = [cref.msm(c.cid, sc, a, n) for a in arrays]
    assert L.nmx_cache_configure(0, 0, 2) == 0      # at most two resident arrays
    try:
        before = S.stats()
        for rnd in range(2):
            for a, e in zip(arrays, exp):
                assert as_pair(g.vartime_multiscalar_mul(sc, a)) == e
        d = delta(S, before)
        assert d[S.STAT_CACHE_UPLOADS] == 8 and d[S.STAT_CACHE_EVICTIONS] == 6
        assert S.stats()[S.STAT_CACHE_ENTRIES] == 2
        assert as_pair(g.vartime_multiscalar_mul(sc, arrays[3])) == exp[3]      # most recent: still resident
        assert delta(S, before)[S.STAT_CACHE_UPLOADS] == 8
    finally:
        assert L.nmx_cache_configure(0, 0, 32) == 0
    assert L.nmx_cache_clear() == 0
    assert S.stats()[S.STAT_CACHE_ENTRIES] == 0 and S.stats()[S.STAT_CACHE_BYTES] == 0


def test_concurrent_first_use_uploads_once(nmx, fresh):
    """rayon::join of two commits over the same ck (r1cs/mod.rs:509-512): eight threads hit a cold array at once."""
    S = fresh
    c = R.BN254_G1
    n = 30000
    bases = cref.sequential_bases(c, 123, n)
    g = nmx.DlogGroup(c.cid)
    lens = [n, 13058, 10538, n - 1, 20000, 4097, n, 999]
    scs = [util.random_scalars(c.cid, m, seed=70 + j) for j, m in enumerate(lens)]
    exp = [cref.msm(c.cid, s, bases[:m], m) for s, m in zip(scs, lens)]
    got = [None] * len(lens)
    start = threading.Barrier(len(lens))

    def run(j):
        start.wait()
        got[j] = as_pair(g.vartime_multiscalar_mul(scs[j], bases[: lens[j]]))

    before = S.stats()
    ths = [threading.Thread(target=run, args=(j,)) for j in range(len(lens))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert got == exp
    d = delta(S, before)
    # shorter prefixes may arrive first and be regrown: at most one upload per distinct length in growing order
    assert 1 <= d[S.STAT_CACHE_UPLOADS] <= 7 and S.stats()[S.STAT_CACHE_ENTRIES] == 1
    assert d[S.STAT_CACHE_UPLOADS] + d[S.STAT_CACHE_HITS] == len(lens)
