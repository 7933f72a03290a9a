"""Seeded synthetic inputs shared by tests and bench.py (SURVEY.md 8(d) "synthetic inputs").

Scalars: uniform in [0, r) by rejection sampling of masked 256-bit PRNG words (numpy PCG64, seed echoing
/root/reference/src/provider/curve_property_tests.rs:26); distributions mirror benches/commit.rs:33-110.
No group arithmetic here: bases come from the oracle (tests) or from nmx_bases_generate (bench).
"""
import numpy as np

SEED = 0x5EEDC0DE12345678

MODULI = {  # scalar-field modulus per curve id (bn256_grumpkin.rs:39-40,84-85; pasta.rs:37-38,45-46)
    0: 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
    1: 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,
    2: 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,
    3: 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,
}


def _limbs(x):
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def _lt(a, m):
    """a: (k,4) uint64 little-endian limbs; m: (4,) -> bool mask a < m"""
    lt = np.zeros(a.shape[0], dtype=bool)
    eq = np.ones(a.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (a[:, i] < m[i])
        eq &= a[:, i] == m[i]
    return lt


def random_scalars(cid, n, seed=SEED):
    """(n, 32) uint8, canonical little-endian, uniform in [0, r)."""
    r = MODULI[cid]
    bits = r.bit_length()
    m = _limbs(r)
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.zeros((n, 4), dtype=np.uint64)
    filled = 0
    top_mask = np.uint64((1 << (bits - 192)) - 1)
    while filled < n:
        k = max(1024, int((n - filled) * 2.2))
        cand = rng.integers(0, 1 << 64, size=(k, 4), dtype=np.uint64, endpoint=False)
        cand[:, 3] &= top_mask
        cand = cand[_lt(cand, m)]
        take = min(len(cand), n - filled)
        out[filled:filled + take] = cand[:take]
        filled += take
    return out.view(np.uint8).reshape(n, 32)


def small_scalars(n, bits, seed=SEED):
    """(n,) uint64 uniform in [0, 2^bits)."""
    rng = np.random.Generator(np.random.PCG64(seed + bits))
    v = rng.integers(0, 1 << 64, size=n, dtype=np.uint64, endpoint=False)
    if bits < 64:
        v &= np.uint64((1 << bits) - 1)
    return v


def u64_to_le32(v):
    """uint64 array -> (n,32) uint8 canonical field scalars"""
    out = np.zeros((len(v), 4), dtype=np.uint64)
    out[:, 0] = v
    return out.view(np.uint8).reshape(len(v), 32)


def int_to_le32(x):
    return np.frombuffer(int(x).to_bytes(32, "little"), dtype=np.uint8)


def scalar_set(cid, n, kind, seed=SEED):
    """The scalar sets of curve_property_tests.rs:196-212 and benches/commit.rs:33-110."""
    r = MODULI[cid]
    if kind == "random":
        return random_scalars(cid, n, seed)
    if kind == "equal":
        one = random_scalars(cid, 1, seed + 1)
        return np.repeat(one, n, axis=0)
    if kind == "zero_rm1":  # alternating 0 / r-1
        out = np.zeros((n, 32), dtype=np.uint8)
        out[1::2] = int_to_le32(r - 1)
        return out
    if kind == "pm_small":  # small positive and small negative scalars (msm.rs signed classification)
        out = random_scalars(cid, n, seed + 2).copy()
        sm = small_scalars(n, 40, seed)
        for i in range(n):
            m = i % 6
            if m == 0:
                out[i] = int_to_le32(int(sm[i]) & 1)
            elif m == 1:
                out[i] = int_to_le32(r - 1 - (int(sm[i]) & 0xFF))
            elif m == 2:
                out[i] = int_to_le32(int(sm[i]) & 0xFFFF)
            elif m == 3:
                out[i] = int_to_le32(r - (int(sm[i]) | 1))
            elif m == 4:
                out[i] = int_to_le32(int(sm[i]))
        return out
    if kind.startswith("u"):
        return u64_to_le32(small_scalars(n, int(kind[1:]), seed))
    raise ValueError(kind)


def witness_like(cid, n, seed):
    """Scalars shaped like an R1CS witness: half zeros, a quarter small (< 2^16), a quarter full-width -- why msm()
    partitions by bit width (/root/reference/src/provider/msm.rs:237-279)."""
    v = random_scalars(cid, n, seed=seed).copy()
    rng = np.random.Generator(np.random.PCG64(seed))
    kind = rng.integers(0, 4, size=n)
    v[kind < 2] = 0
    small = u64_to_le32(small_scalars(n, 16, seed=seed))
    v[kind == 2] = small[kind == 2]
    return v


R_INTERNAL = 1 << 261  # Montgomery radix of the library's internal residue form (nova_amd/csrc/fp.hpp)


def affine_to_partial(p_mod, xy64, is_inf):
    """Canonical affine bytes -> the 128-byte partial format of NMX_OUT_PARTIAL / nmx_point_sum:
    (X, Y, ZZ, ZZZ) = (x, y, 1, 1) * 2^261 mod p, each a 32-byte LE integer; identity <=> ZZ == 0."""
    one = R_INTERNAL % p_mod
    if is_inf:
        return one.to_bytes(32, "little") * 2 + bytes(64)
    x = int.from_bytes(xy64[:32], "little")
    y = int.from_bytes(xy64[32:64], "little")
    return b"".join((v * R_INTERNAL % p_mod).to_bytes(32, "little") for v in (x, y, 1, 1))


def to_mont_scalars(cid, sc):
    """Scalars (n x 32 canonical LE) in the in-memory layout of halo2curves: x * 2^256 mod r as LE limbs."""
    from oracle import pyref as R
    r = [c for c in R.CURVES.values() if c.cid == cid][0].r
    rows = np.ascontiguousarray(sc).reshape(-1, 32)
    out = b"".join(R.fe_to_le32((int.from_bytes(bytes(row), "little") << 256) % r) for row in rows)
    return np.frombuffer(out, np.uint8).reshape(-1, 32).copy()


def to_mont_bases(cid, xy64):
    """Affine points (n x 64 canonical x || y) with both coordinates as x * 2^256 mod p; the identity (0, 0) stays (0, 0)."""
    from oracle import pyref as R
    p = [c for c in R.CURVES.values() if c.cid == cid][0].p
    rows = np.ascontiguousarray(xy64).reshape(-1, 64)
    out = b"".join(R.fe_to_le32((int.from_bytes(bytes(row[j:j + 32]), "little") << 256) % p) for row in rows for j in (0, 32))
    return np.frombuffer(out, np.uint8).reshape(-1, 64).copy()

