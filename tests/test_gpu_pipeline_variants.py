"""-m gpu: every pipeline variant computes the same, oracle-exact result.  The hand-written LDS partition
(msm_partition.hpp), the generic radix-sort path, the task accumulate and the segment-balanced accumulate (msm_seg.hpp,
forced on at small sizes, with lane counts that make buckets span 0 .. thousands of segments) are selected with
nmx_set_option and run on all nine scalar sets of the reference's test matrix
(/root/reference/src/provider/curve_property_tests.rs:196-212, benches/commit.rs:33-110)."""
import numpy as np
import pytest

from oracle import cref
from oracle import pyref as R
from tests import util

pytestmark = pytest.mark.gpu
KINDS = ["random", "equal", "zero_rm1", "pm_small", "u1", "u10", "u16", "u32", "u64"]
VARIANTS = [
    {},                                                            # defaults
    {"no_partition": 1},                                           # round-1 pipeline
    {"seg_min_total": 0xFFFFFFFF},                                 # partition + task accumulate
    {"seg_min_total": 1 << 22},                                    # round 2's threshold
    {"no_tree_fuse": 1},                                           # one launch per reduction level
    {"no_tree_fuse": 2, "seg_min_total": 0},                       # fused reduction tree, segments
    {"seg_min_total": 0, "big_slice": 64},                         # big-bucket pass in short slices: groups of tickets, then the bucket's
    {"seg_min_total": 0, "big_slice": 100, "big_threads": 512},    # a slice that is no multiple of the block's quads; 512-thread blocks
    {"seg_min_total": 0, "big_slice": 4096, "big_threads": 256},   # one slice per bucket at these sizes
    {"no_tree_fuse": 2, "tree_threads": 256},                      # fused tree in 256-thread blocks (six levels per launch)
    {"seg_min_total": 0, "seg_lanes": 4096, "seg_min_len": 3},     # segments everywhere, few long ones
    {"seg_min_total": 0, "seg_lanes": 1 << 19, "seg_min_len": 1},  # more lanes than entries per bucket: one-entry pieces
    {"seg_min_total": 0, "seg_lanes": 70001, "no_quad_final": 1},  # odd lane count, single-lane final pass
    {"seg_min_total": 0, "accum_prefetch": 2},
    {"seg_min_total": 0, "seg_lanes": 1 << 18, "seg_heavy_above": 3},   # nearly every bucket through the pre-fold passes
    {"seg_min_total": 0, "seg_lanes": 1 << 18, "seg_heavy_above": 40},  # long serial sums in the final pass instead
]
DEFAULTS = {"no_partition": 0, "seg_min_total": 0xFFFFFFFE,   # 0xfffffffe = the automatic rule
             "seg_lanes": 0, "seg_min_len": 8, "no_quad_final": 0, "accum_prefetch": 0,
            "seg_heavy_above": 0, "no_tree_fuse": 0, "tree_threads": 0, "big_slice": 0, "big_threads": 0}


@pytest.mark.parametrize("c,n", [(R.BN254_G1, 20000), (R.BN254_G1, 1 << 16), (R.PALLAS, 1 << 17)], ids=lambda v: getattr(v, "name", v))
def test_variants_agree_with_oracle(nmx, c, n):
    from nova_amd import _lib
    L = _lib.lib()
    bases = cref.sequential_bases(c, 606 + n, n).copy()
    bases[n // 7] = 0                                             # an identity point: the digit stage must then read bases
    prep = cref.Prepared(c.cid, bases, n)
    ck = nmx.CommitmentKey.from_host(c.cid, bases)
    clean = nmx.CommitmentKey.from_host(c.cid, cref.sequential_bases(c, 5, n))   # and a key without one (bases never read)
    prep_clean = cref.Prepared(c.cid, clean.read(0, n), n)
    g = nmx.DlogGroup(c.cid)
    scs = {k: util.scalar_set(c.cid, n, k) for k in KINDS}
    exp = {k: prep.msm(s, n) for k, s in scs.items()}
    exp_clean = prep_clean.msm(scs["random"], n)
    try:
        for var in VARIANTS:
            for k, v in {**DEFAULTS, **var}.items():
                assert L.nmx_set_option(k.encode(), v) == 0
            for kind in KINDS:
                got = g.vartime_multiscalar_mul(scs[kind], ck)
                assert (got.xy, int(got.is_inf)) == exp[kind], (var, kind)
            got = g.vartime_multiscalar_mul(scs["random"], clean)
            assert (got.xy, int(got.is_inf)) == exp_clean, var
            s64 = util.small_scalars(n, 33)
            got = g.vartime_multiscalar_mul_small(s64, clean)
            assert (got.xy, int(got.is_inf)) == cref.msm_u64(c.cid, s64, clean.read(0, n), n, 33), var
    finally:
        for k, v in DEFAULTS.items():
            L.nmx_set_option(k.encode(), v)
        assert L.nmx_set_option(b"no_such_knob", 1) == _lib.E_ARG
    ck.close()
    clean.close()


@pytest.mark.parametrize("c", [R.BN254_G1, R.GRUMPKIN], ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 17, 200, 1000, 5000, 10538, 13058])
def test_small_msm_block_path_agrees_with_the_task_path_and_the_oracle(nmx, c, n):
    """Round 5: MSMs with at most 1024 buckets (keys below 2^14 points: c = 8 tables; plain keys of a few hundred pairs) sum their
    buckets in two block-level launches (curve_quad.hpp k_small_accum / k_small_combine; option small_blocks = entries per
    four-lane group, 0 = the task path).  Every scalar set of the reference's matrix -- including the ones that put everything into
    one bucket (u1, equal) -- at one entry per group (many partials per bucket), the default and 64 (one block per bucket), on a
    key with tables, a plain key, and a key with identity points."""
    from nova_amd import _lib
    L = _lib.lib()
    bases = cref.sequential_bases(c, 77 + n, n).copy()
    if n > 20:
        bases[n // 3] = 0
    prep = cref.Prepared(c.cid, bases, n)
    keys = [nmx.CommitmentKey.from_host(c.cid, bases), nmx.CommitmentKey.from_host(c.cid, bases, precompute=False)]
    g = nmx.DlogGroup(c.cid)
    scs = {k: util.scalar_set(c.cid, n, k) for k in KINDS}
    exp = {k: prep.msm(s, n) for k, s in scs.items()}
    s64 = util.small_scalars(n, 33)
    exp64 = cref.msm_u64(c.cid, s64, bases, n, 33)
    try:
        for sb in (0, 1, 8, 64):
            assert L.nmx_set_option(b"small_blocks", sb) == 0
            for ck in keys:
                for kind in KINDS:
                    got = g.vartime_multiscalar_mul(scs[kind], ck)
                    assert (got.xy, int(got.is_inf)) == exp[kind], (sb, kind)
                got = g.vartime_multiscalar_mul_small(s64, ck)
                assert (got.xy, int(got.is_inf)) == exp64, sb
    finally:
        assert L.nmx_set_option(b"small_blocks", 8) == 0
    for ck in keys:
        ck.close()


def test_host_scalar_calls_cut_into_overlapping_pieces(nmx):
    """Round 5 (option host_split): an MSM with HOST scalars over a long range is cut into contiguous pieces whose uploads overlap
    the previous piece's MSM (the reference's chunk + reduce decomposition, src/provider/msm.rs:564-574); the result is the
    unsplit one, for prefixes and interior slices, canonical and Montgomery scalars, small scalars; a scalar >= r in any piece
    fails the whole call."""
    from nova_amd import _lib
    L = _lib.lib()
    c = R.BN254_G1
    n = 30000
    bases = cref.sequential_bases(c, 4242, n).copy()
    bases[n // 2 + 3] = 0
    prep = cref.Prepared(c.cid, bases, n)
    ck = nmx.CommitmentKey.from_host(c.cid, bases)
    g = nmx.DlogGroup(c.cid)
    sc = util.random_scalars(c.cid, n, seed=77)
    exp = prep.msm(sc, n)
    s64 = util.small_scalars(n, 40)
    exp64 = cref.msm_u64(c.cid, s64, bases, n, 40)
    off, m = 1234, 20001
    exp_mid = cref.msm(c.cid, sc[:m], bases[off:off + m], m)
    try:
        assert L.nmx_set_option(b"host_split_min_n", 8192) == 0
        assert L.nmx_set_option(b"host_split", 17) == _lib.E_ARG
        for k in (2, 0, 3, 4, 7, 1, 255):
            assert L.nmx_set_option(b"host_split", k) == 0
            got = g.vartime_multiscalar_mul(sc, ck)
            assert (got.xy, int(got.is_inf)) == exp, k
            got = g.vartime_multiscalar_mul(sc[:m], ck, offset=off)
            assert (got.xy, int(got.is_inf)) == exp_mid, k
            got = g.vartime_multiscalar_mul(util.to_mont_scalars(c.cid, sc), ck, mont=True)
            assert (got.xy, int(got.is_inf)) == exp, k
            got = g.vartime_multiscalar_mul_small(s64, ck)
            assert (got.xy, int(got.is_inf)) == exp64, k
            part = g.vartime_multiscalar_mul(sc, ck, partial=True)
            assert (lambda r: (r.xy, int(r.is_inf)))(g.point_sum([part.xy])) == exp, k
        assert L.nmx_set_option(b"host_split", 3) == 0
        bad = sc.copy()
        bad[n - 5] = 0xFF                                    # >= r, in the last piece
        with pytest.raises(nmx.NmxError) as e:
            g.vartime_multiscalar_mul(bad, ck)
        assert e.value.code == _lib.E_SCALAR_RANGE
        got = g.vartime_multiscalar_mul(sc, ck)                # and the library is fine afterwards
        assert (got.xy, int(got.is_inf)) == exp
    finally:
        assert L.nmx_set_option(b"host_split", 255) == 0
        assert L.nmx_set_option(b"host_split_min_n", 1 << 19) == 0
    ck.close()
