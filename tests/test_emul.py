"""CPU-side check of the HIP kernel bodies (no GPU): tests/host_emul runs the same functors and the same
msm_pipeline() as libnova_mi355x.so with a loop backend (see tests/host_emul/emul.cpp) and is compared against
the oracle.  This exercises digit recoding, sorting/bounds, over-long bucket splitting + folds, the reduction tree
and the host tail for every curve, window width and scalar distribution -- it is a debugging aid that saves GPU
minutes, not a shipped path: the real parity gate is tests/test_gpu_parity.py through the C ABI.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import cref
from oracle import pyref as R
from tests import util

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_emul", "emul.cpp")
SO = os.path.join(HERE, "host_emul", "libnmx_emul.so")
CSRC = os.path.join(os.path.dirname(HERE), "nova_amd", "csrc")


@pytest.fixture(scope="module")
def emul():
    deps = [SRC, os.path.join(HERE, "host_emul", "simt.hpp")] + [
        os.path.join(CSRC, f) for f in ("fp.hpp", "curve.hpp", "curves.hpp", "msm_kernels.hpp", "msm_pipeline.hpp",
                                        "msm_partition.hpp", "msm_seg.hpp", "curve_quad.hpp")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-DNMX_DEBUG_BOUNDS", "-shared", "-fPIC", "-o", SO, SRC])
    L = ctypes.CDLL(SO)
    L.emul_msm.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                           ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    L.emul_msm_precomp.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                   ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                   ctypes.c_void_p]
    L.emul_fp_op.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    L.emul_partition_check.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                       ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    L.emul_msm_batch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                 ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                 ctypes.c_void_p]
    return L


def run(L, cid, scalars, bases, n, force_c=0, u64_bits=None):
    s = np.ascontiguousarray(scalars)
    b = np.ascontiguousarray(bases)
    out = np.zeros(64, np.uint8)
    inf = np.zeros(1, np.uint8)
    mode = 0 if u64_bits is None else 1
    rc = L.emul_msm(cid, s.ctypes.data, b.ctypes.data, n, u64_bits or 0, mode, 0, force_c, out.ctypes.data, inf.ctypes.data)
    return rc, out.tobytes(), int(inf[0])


FIELDS = [R.BN254_Q, R.BN254_R, R.PALLAS_P, R.PALLAS_Q]


@pytest.mark.parametrize("fid", range(4))
def test_field_constants(fid):
    """Every generated constant in fp.hpp (9 x 29-bit limbs) re-derived from the modulus."""
    import re
    p = FIELDS[fid]
    src = open(os.path.join(CSRC, "fp.hpp")).read()
    name = ["F_BN254_FQ", "F_BN254_FR", "F_PASTA_FP", "F_PASTA_FQ"][fid]
    blk = src[src.index(f"struct FpParams<{name}>"):]
    blk = blk[:blk.index("\n};") + 3]

    def arr(nm):
        m = re.search(r"uint32_t %s\[\d\] = \{([^}]*)\}" % nm, blk)
        return [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]

    def val29(l):
        return sum(v << (29 * i) for i, v in enumerate(l))

    assert val29(arr("P")) == p and all(v < 1 << 29 for v in arr("P")[:8])
    assert sum(v << (32 * i) for i, v in enumerate(arr("PW"))) == p
    assert val29(arr("ONE")) == (1 << 261) % p
    assert val29(arr("R2")) == (1 << 522) % p
    assert val29(arr("C266")) == (1 << 266) % p
    ninv = int(re.search(r"NINV = (0x[0-9a-f]+)u", blk).group(1), 16)
    assert (ninv * p + 1) % (1 << 29) == 0
    for k, nm in ((2, "K2P"), (4, "K4P"), (8, "K8P")):
        K = arr(nm)
        assert val29(K) == k * p
        assert all(v >= (1 << 29) - 1 for v in K[:8])          # dominates every normalized limb
        assert K[8] >= ((k * p - (1 << 233)) >> 232)             # and the top limb of anything < kp - 2^233
        assert all(v + (1 << 30) < 1 << 32 for v in K)           # a + K - b cannot wrap
    assert (1 << 261) // p >= 127


@pytest.mark.parametrize("fid", range(4))
def test_field_ops(emul, fid):
    """fp.hpp (9 x 29-bit limbs, R' = 2^261, lazy reduction) against big-int arithmetic, incl. operands at the
    edges of the stated bounds (multiples of p up to 8p - 1, all-ones limbs)."""
    p = FIELDS[fid]
    Rm = 1 << 261
    Ri = pow(Rm, -1, p)
    rng = np.random.Generator(np.random.PCG64(fid))

    def op(o, a, b=0):
        out = ctypes.create_string_buffer(32)
        emul.emul_fp_op(fid, o, a.to_bytes(32, "little"), b.to_bytes(32, "little"), out)
        return int.from_bytes(out.raw, "little")

    rnd = [int.from_bytes(rng.bytes(32), "little") % p for _ in range(30)]
    edge = [0, 1, 2, p - 1, p - 2, (1 << 253) - 1, Rm % p, (1 << 232) - 1, ((1 << 29) - 1) * sum(1 << (29 * i) for i in range(8))]
    wide = [p, p + 1, 2 * p - 1, 2 * p, 3 * p + 5, 4 * p - 1]  # weakly reduced operands (still < 2^256)
    wide = [v for v in wide if v < 1 << 256]
    for a in edge + rnd:
        for b in edge + rnd[:6] + wide:
            if (a // p + 1) * (b // p + 1) < 127:
                assert op(0, a, b) == a * b * Ri % p
                assert op(13, a, b) == a * b * Ri % p                     # sos_mul: limb-identical to the chained product
            if 2 * (a // p + 1) * (b // p + 1) < 127:
                assert op(15, a, b) == 2 * a * b * Ri % p                 # a*b + b*a with one reduction
            if (a // p + 1 + b // p + 1) ** 2 < 127 and max(a, b) < 1 << 255:  # Fp::dot: normalised limbs, sum of products < 127 p^2
                assert op(16, a, b) == (a + b) ** 2 * Ri % p              # four products, one reduction
                assert op(17, a, b) == (a * b + a * a + b * b) * Ri % p   # three
                assert op(18, a, b) == a * b * Ri % p                     # one: limb-identical to the product
            assert op(1, a, b % (1 << 255)) == (a + b % (1 << 255)) % p
            if b < 2 * p - (1 << 233):
                assert op(2, a, b) == (a - b) % p
            if b < 4 * p - (1 << 233):
                assert op(10, a, b) == (a - b) % p
            assert op(6, a, b) == (a - b) % p          # 8p spread covers everything < 2^256 ... < 8p - 2^233
        assert op(3, a % p) == a * Rm % p
        assert op(4, a) == a * Ri % p
        assert op(7, a) == a * a * Ri % p
        assert op(14, a) == a * a * Ri % p
        assert op(8, a % p) == (a % p) * 32 % p
        assert op(9, a % p) == (a % p) * pow(1 << 256, -1, p) % p
        assert op(11, a) == a % p
    for a in wide + [5 * p + 7, 7 * p + 123] if 8 * p < 1 << 256 else wide:
        if a < 1 << 256:
            assert op(11, a) == a % p
            assert op(7, a) == a * a * Ri % p
    # inversion (host: binary extended Euclid on 4 x u64; also on weakly reduced inputs r + k*p)
    more = [int.from_bytes(rng.bytes(32), "little") % p for _ in range(200)]
    for a in rnd + more + [1, 2, p - 1, p - 2, (p + 1) // 2, 1 << 200, (1 << 253) % p]:
        if a % p == 0:
            continue
        r = a * Rm % p
        assert op(5, r) == pow(a, -1, p) * Rm % p
        for k in (1, 3):
            if r + k * p < 1 << 256:
                assert op(5, r + k * p) == pow(a, -1, p) * Rm % p
    assert op(5, 0) == 0 and op(5, p) == 0
    # the cheap divisibility filter never misses a true multiple of p
    for k in range(0, 4):
        if k * p < 1 << 256:
            assert op(12, k * p) == 1
    assert sum(op(12, v) for v in rnd) == 0  # 2^-29 false-positive rate: none expected on 30 samples


@pytest.mark.parametrize("c", list(R.CURVES.values()), ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 2, 16, 17, 100, 1000])
def test_emul_msm_all_sets(emul, c, n):
    bases = cref.sequential_bases(c, 1000 + n, n)
    for kind in ["random", "equal", "zero_rm1", "pm_small", "u1", "u10"]:
        sc = util.scalar_set(c.cid, n, kind)
        rc, got, inf = run(emul, c.cid, sc, bases, n)
        assert rc == 0
        assert (got, inf) == cref.msm(c.cid, sc, bases, n), (c.name, n, kind)


@pytest.mark.parametrize("force_c", [1, 2, 5, 9, 13, 16])
def test_emul_window_widths(emul, force_c):
    c = R.BN254_G1
    n = 300
    bases = cref.sequential_bases(c, 3, n)
    for kind in ["random", "zero_rm1"]:
        sc = util.scalar_set(c.cid, n, kind)
        rc, got, inf = run(emul, c.cid, sc, bases, n, force_c=force_c)
        assert rc == 0 and (got, inf) == cref.msm(c.cid, sc, bases, n), (force_c, kind)
    c = R.PALLAS  # 255-bit scalars: top window carries
    bases = cref.sequential_bases(c, 3, n)
    sc = util.scalar_set(c.cid, n, "zero_rm1")
    rc, got, inf = run(emul, c.cid, sc, bases, n, force_c=force_c)
    assert rc == 0 and (got, inf) == cref.msm(c.cid, sc, bases, n)


def test_emul_heavy_buckets_and_identity_bases(emul):
    """All-equal scalars put every point of a window in one bucket: over-long buckets are split into extra tasks
    and folded (PlanFn / FoldFn).  8200 > 256 * lmax exercises all three fold passes."""
    c = R.BN254_G1
    n = 8200
    bases = cref.sequential_bases(c, 77, n).copy()
    bases[5] = 0
    bases[4000] = 0
    for kind in ["equal", "zero_rm1", "random"]:
        sc = util.scalar_set(c.cid, n, kind)
        rc, got, inf = run(emul, c.cid, sc, bases, n)
        assert rc == 0 and (got, inf) == cref.msm(c.cid, sc, bases, n), kind
        if kind == "equal":  # c = 13 -> lmax = 32 -> 257 extra tasks per bucket: the T = 256 fold pass loops
            rc, got, inf = run(emul, c.cid, sc, bases, n, force_c=13)
            assert rc == 0 and (got, inf) == cref.msm(c.cid, sc, bases, n)
    # duplicate bases: P + P inside a bucket takes the doubling branch, P + (-P) the cancellation branch
    dup = np.repeat(bases[:1], 64, axis=0)
    sc = util.scalar_set(c.cid, 64, "equal")
    rc, got, inf = run(emul, c.cid, sc, dup, 64)
    assert rc == 0 and (got, inf) == cref.msm(c.cid, sc, dup, 64)
    sc = util.scalar_set(c.cid, 64, "zero_rm1").copy()
    sc[0::2] = util.int_to_le32(1)  # 1*P + (r-1)*P + ... == identity
    rc, got, inf = run(emul, c.cid, sc, dup, 64)
    assert rc == 0 and (got, inf) == (bytes(64), 1)


@pytest.mark.parametrize("bits", [1, 4, 8, 10, 16, 20, 32, 40, 64])
def test_emul_small_scalars(emul, bits):
    """msm_small_with_max_num_bits semantics (msm.rs:478-503) incl. the out-of-contract error."""
    for c in (R.BN254_G1, R.VESTA):
        n = 200
        bases = cref.sequential_bases(c, 9, n)
        s = util.small_scalars(n, bits)
        rc, got, inf = run(emul, c.cid, s, bases, n, u64_bits=bits)
        assert rc == 0 and (got, inf) == cref.msm_u64(c.cid, s, bases, n, bits)
    if bits < 64:
        s = s.copy()
        s[7] = np.uint64(1) << np.uint64(bits)
        rc, _, _ = run(emul, c.cid, s, bases, n, u64_bits=bits)
        assert rc != 0  # ERR_SMALL_RANGE


def test_emul_scalar_out_of_range(emul):
    c = R.BN254_G1
    n = 20
    bases = cref.sequential_bases(c, 9, n)
    sc = util.random_scalars(c.cid, n).copy()
    sc[3] = util.int_to_le32(c.r)  # == modulus: from_repr rejects
    rc, _, _ = run(emul, c.cid, sc, bases, n)
    assert rc != 0


def run_pre(L, cid, scalars, key, n_key, offset, n, pre_c, u64_bits=None):
    s = np.ascontiguousarray(scalars)
    k = np.ascontiguousarray(key)
    out = np.zeros(64, np.uint8)
    inf = np.zeros(1, np.uint8)
    mode = 0 if u64_bits is None else 1
    rc = L.emul_msm_precomp(cid, s.ctypes.data, k.ctypes.data, n_key, offset, n, u64_bits or 0, mode, pre_c,
                            out.ctypes.data, inf.ctypes.data)
    return rc, out.tobytes(), int(inf[0])


@pytest.mark.parametrize("c", list(R.CURVES.values()), ids=lambda c: c.name)
def test_emul_precomputed_tables(emul, c):
    """Registered-key mode: window w reads table 2^(cw)*P, all windows share one bucket set (PrecompFn +
    pre_stride digits); prefixes and interior slices of the key; every scalar set; small-scalar mode."""
    n_key = 160
    key = cref.sequential_bases(c, 21, n_key).copy()
    key[13] = 0  # an identity point inside the key
    cases = [(8, 0, n_key, "random"), (8, 0, 77, "equal"), (8, 30, 100, "zero_rm1"), (8, 13, 1, "random"),
             (11, 0, n_key, "pm_small"), (16, 5, 150, "random")]
    for pre_c, off, n, kind in cases:
        sc = util.scalar_set(c.cid, n, kind)
        rc, got, inf = run_pre(emul, c.cid, sc, key, n_key, off, n, pre_c)
        assert rc == 0 and (got, inf) == cref.msm(c.cid, sc, key[off:off + n], n), (pre_c, off, n, kind)
    for bits in (1, 33, 64):
        s = util.small_scalars(n_key, bits)
        rc, got, inf = run_pre(emul, c.cid, s, key, n_key, 0, n_key, 9, u64_bits=bits)
        assert rc == 0 and (got, inf) == cref.msm_u64(c.cid, s, key, n_key, bits)


@pytest.mark.parametrize("c,n,grid", [(16, 40000, 0), (16, 40000, 3), (15, 20011, 2), (8, 30000, 5), (8, 700, 0), (12, 5000, 1),
                                      (6, 300, 0), (16, 1, 0), (16, 1025, 0), (20, 4000, 2), (18, 5000, 0),
                                      (17, 2500, 1)])
def test_emul_partition_kernels(emul, c, n, grid):
    """The hand-written LDS partition (msm_partition.hpp) run thread by thread on the CPU (tests/host_emul/simt.hpp:
    one fiber per GPU thread, real barriers) against DigitsFn + a sort: per-bucket multisets, start / end, totals.
    Scalar sets include the adversarial ones for a counting sort: every entry in one bucket per window (equal), almost
    everything dropped as a zero digit (u1), and 0 / r-1."""
    # (the wide geometry launches ~1000 mostly idle 1024-thread blocks per level: two scalar sets keep its emulation short)
    for kind in (["random", "equal", "zero_rm1", "u1", "pm_small"] if c <= 17 else ["random", "zero_rm1"]):
        sc = np.ascontiguousarray(util.scalar_set(0, n, kind))
        for ct in (1, 0):     # the compile-time-width instantiation (c = 8 / 15 / 16) and the run-time-width one
            rc = emul.emul_partition_check(sc.ctypes.data, n, c, 0, grid, n + 7, 3, ct)
            assert rc == 0, (c, n, grid, kind, ct, rc)
    s64 = np.ascontiguousarray(util.small_scalars(n, 33))
    assert emul.emul_partition_check(s64.ctypes.data, n, c, 33, grid, n, 0, 1) == 0


@pytest.mark.parametrize("c,n,grid", [(12, 5000, 0), (16, 3000, 2), (9, 1000, 1), (5, 300, 0), (3, 100, 0), (13, 1, 0)])
def test_emul_partition_kernels_plain_keys(emul, c, n, grid):
    """Round 4: the same two-level counting scatter for keys WITHOUT window tables -- W bucket sets, key =
    w * M + |d| - 1 over ceil(log2(W * M)) bits (19 for the c = 16 / W = 16 of a 2^20-pair plain MSM: the wide geometry; W * M
    need not be a power of two, the bins past the last bucket stay empty) -- against DigitsFn + a sort.  stride = 0 selects
    the plain layout (the value word is the base index itself)."""
    # (the wide geometry launches ~1000 mostly idle 1024-thread blocks per level: two scalar sets keep its emulation short)
    for kind in (["random", "equal", "zero_rm1", "u1"] if c < 16 else ["random", "zero_rm1"]):
        sc = np.ascontiguousarray(util.scalar_set(0, n, kind))
        for ct in (1, 0):
            rc = emul.emul_partition_check(sc.ctypes.data, n, c, 0, grid, 0, 0, ct)
            assert rc == 0, (c, n, grid, kind, ct, rc)
    s64 = np.ascontiguousarray(util.small_scalars(n, 33))
    assert emul.emul_partition_check(s64.ctypes.data, n, c, 33, grid, 0, 0, 1) == 0


def test_emul_plain_msm_runs_the_partition_not_the_sort(emul):
    """A plain-key MSM through msm_pipeline takes the hand-written partition (block-level kernels) for every default window
    width, down to the 85-window shapes of tiny inputs."""
    emul.emul_block_kernel_launches.restype = ctypes.c_ulonglong
    c = R.GRUMPKIN
    for n in (2, 40, 300, 5000):
        bases = cref.sequential_bases(c, 7 + n, n)
        sc = util.scalar_set(c.cid, n, "random")
        before = emul.emul_block_kernel_launches()
        rc, got, inf = run(emul, c.cid, sc, bases, n)
        assert rc == 0 and (got, inf) == cref.msm(c.cid, sc, bases, n)
        assert emul.emul_block_kernel_launches() - before == 5, n    # hist_hi, tiles, part_hi, hist_lo, part_lo


@pytest.mark.parametrize("lanes", [37, 700, 5000])
def test_emul_segment_balanced_accumulate(emul, lanes):
    """msm_seg.hpp (AccumSegFn / PlanSegFn / FoldRawFn / FinalSegFn) on the table path: segments that straddle bucket
    boundaries, buckets spanning 0, a few, > 8 (heavy list) and > 64 (big list, strided pre-folds) lanes, empty buckets,
    identity points inside the key, prefixes / interior slices, small-scalar mode -- all against the oracle."""
    emul.emul_set_seg_min_total(0)
    emul.emul_set_seg_lanes(lanes)
    try:
        c = R.BN254_G1
        n_key = 1000
        key = cref.sequential_bases(c, 21, n_key).copy()
        key[13] = 0
        for pre_c, off, n, kind in [(8, 0, n_key, "random"), (8, 0, n_key, "equal"), (8, 100, 777, "zero_rm1"), (11, 0, 999, "pm_small"),
                                    (16, 5, 150, "random"), (8, 0, 1, "random"), (8, 0, n_key, "u1"), (12, 3, 900, "equal")]:
            sc = util.scalar_set(c.cid, n, kind)
            rc, got, inf = run_pre(emul, c.cid, sc, key, n_key, off, n, pre_c)
            assert rc == 0 and (got, inf) == cref.msm(c.cid, sc, key[off:off + n], n), (lanes, pre_c, off, n, kind)
        s = util.small_scalars(n_key, 33)
        rc, got, inf = run_pre(emul, c.cid, s, key, n_key, 0, n_key, 9, u64_bits=33)
        assert rc == 0 and (got, inf) == cref.msm_u64(c.cid, s, key, n_key, 33)
        cc = R.VESTA
        key = cref.sequential_bases(cc, 5, 300)
        sc = util.scalar_set(cc.cid, 300, "zero_rm1")
        rc, got, inf = run_pre(emul, cc.cid, sc, key, 300, 0, 300, 8)
        assert rc == 0 and (got, inf) == cref.msm(cc.cid, sc, key, 300)
    finally:
        emul.emul_set_seg_min_total(1 << 21)
        emul.emul_set_seg_lanes(37)


@pytest.mark.parametrize("c", list(R.CURVES.values()), ids=lambda c: c.name)
def test_emul_two_pass_tables_match_one_pass(emul, c):
    """PrecompDblFn + PrecompNormFn (one shared inversion per key point) build byte-identical window tables to PrecompFn,
    identity points inside the key included."""
    n = 37
    key = cref.sequential_bases(c, 1234, n).copy()
    key[0] = 0
    key[20] = 0
    for width in (8, 16, 20):
        assert emul.emul_precomp_check(c.cid, key.ctypes.data_as(ctypes.c_void_p), n, width) == 0, (c.name, width)


def run_batch(L, cid, vecs, key, n_key, offset, pre_c, mont=0):
    k = len(vecs)
    arrs = [np.ascontiguousarray(v).reshape(-1, 32) for v in vecs]
    ptrs = (ctypes.c_void_p * k)(*[a.ctypes.data if len(a) else None for a in arrs])
    lens = (ctypes.c_size_t * k)(*[len(a) for a in arrs])
    kk = np.ascontiguousarray(key)
    out = np.zeros((k, 64), np.uint8)
    inf = np.zeros(k, np.uint8)
    rc = L.emul_msm_batch(cid, ptrs, lens, k, kk.ctypes.data, n_key, offset, pre_c, mont, out.ctypes.data, inf.ctypes.data)
    return rc, [(out[j].tobytes(), int(inf[j])) for j in range(k)]


@pytest.mark.parametrize("seg", [False, True], ids=["tasks", "segments"])
def test_emul_fused_batch(emul, seg):
    """a7 as ONE pipeline run (MsmArgs::batch_*): k ragged vectors over prefixes / interior slices of one key's tables, a
    bucket set per vector -- the partition's keys carry the vector id above the bucket bits (narrow geometry up to 15 key
    bits, wide above; compile-time widths 8 / 15 / 16 and a run-time one), empty vectors, single pairs, an identity point
    in the key, every scalar set, Montgomery-form scalars.  Each vector against the oracle's MSM over its own prefix."""
    if seg:
        emul.emul_set_seg_min_total(0)
        emul.emul_set_seg_lanes(701)
    try:
        c = R.BN254_G1
        n_key = 330
        key = cref.sequential_bases(c, 77, n_key).copy()
        key[5] = 0
        kinds = ["random", "equal", "zero_rm1", "u1", "pm_small"]
        cases = [(8, 0, [330, 37, 0, 1, 300]),            # 5 vectors -> 8 bucket sets, 10 key bits
                 (8, 7, [2, 1, 0, 0, 3] * 40),            # 200 tiny vectors -> 256 sets, 15 key bits
                 (16, 0, [150, 20, 1]),                   # c = 16 tables: 17 key bits, the wide geometry
                 (15, 3, [100, 64, 32, 16, 8]),           # c = 15: 17 windows, 960-thread first-level blocks
                 (11, 0, [90, 45]),                       # run-time window width
                 (16, 1, [40] * 16),                      # 16 sets over c = 16: 19 key bits
                 (17, 2, [25] * 9 + [0, 3])]              # 11 -> 16 sets over c = 17: all 20 key bits (1024 x 1024 bins)
        for pre_c, off, lens in (cases[:1] + cases[2:4] if seg else cases):
            vecs = [util.scalar_set(c.cid, n, kinds[j % len(kinds)], seed=100 + j) if n else np.zeros((0, 32), np.uint8)
                    for j, n in enumerate(lens)]
            rc, got = run_batch(emul, c.cid, vecs, key, n_key, off, pre_c)
            assert rc == 0, (pre_c, off, lens, rc)
            for j, n in enumerate(lens):
                exp = cref.msm(c.cid, vecs[j], key[off:off + n], n) if n else (bytes(64), 1)
                assert got[j] == exp, (seg, pre_c, off, lens, j)
        # Montgomery-form scalars on another curve
        cc = R.PALLAS
        key = cref.sequential_bases(cc, 9, 120)
        vecs = [util.scalar_set(cc.cid, n, "random", seed=7 + n) for n in (120, 60, 30)]
        mont = [util.to_mont_scalars(cc.cid, v) for v in vecs]
        rc, got = run_batch(emul, cc.cid, mont, key, 120, 0, 8, mont=1)
        assert rc == 0
        for j, v in enumerate(vecs):
            assert got[j] == cref.msm(cc.cid, v, key[:len(v)], len(v))
    finally:
        emul.emul_set_seg_min_total(1 << 21)
        emul.emul_set_seg_lanes(37)
