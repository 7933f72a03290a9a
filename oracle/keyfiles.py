"""Oracle-side restatement of the reference's on-disk key formats (writers and readers), plain Python big-ints.

TEST INFRASTRUCTURE ONLY (see oracle/pyref.py).  Follows, relative to /root/reference:
  write_header / write_ptau / read_meta_data / read_header / read_points / read_ptau   src/provider/ptau.rs:153-436
  Pedersen save_setup / load_setup ("PEDERSEN_KEY" | h | ck)                           src/provider/pedersen.rs:28,318-340,383-393
A point record is halo2curves `write_raw`: x || y, each the 4 x u64 little-endian limbs of the Montgomery residue
(value * 2^256 mod p).  The identity is (0, 0).  halo2curves itself is not in /root/reference (Cargo.toml:36-41), so the
record layout is pinned only by the reference's own use of it (ptau.rs:192-203, 372-391) -- "parity unpinned" at the
byte level for lack of a stored file; the tests round-trip writer -> product reader and reader -> reader.
"""
import io
import struct

from . import pyref as R

PTAU_VERSION = 1           # ptau.rs:160
NUM_SECTIONS_FULL = 11     # ptau.rs:162
NUM_SECTIONS_PRUNED = 3    # ptau.rs:164
KEY_FILE_HEAD = b"PEDERSEN_KEY"  # pedersen.rs:28
MONT_R = 1 << 256


class PtauFileError(Exception):
    pass


def raw_point(c: R.Curve, P) -> bytes:
    if P is R.INF:
        return bytes(64)
    return (P[0] * MONT_R % c.p).to_bytes(32, "little") + (P[1] * MONT_R % c.p).to_bytes(32, "little")


def parse_raw_point(c: R.Curve, b: bytes):
    """read_raw + is_on_curve (ptau.rs:372-391)"""
    xm, ym = int.from_bytes(b[:32], "little"), int.from_bytes(b[32:64], "little")
    if xm >= c.p or ym >= c.p:
        raise PtauFileError("PointNotOnCurve (non-canonical coordinate)")
    rinv = pow(MONT_R, -1, c.p)
    P = (xm * rinv % c.p, ym * rinv % c.p)
    if P == (0, 0):
        return R.INF
    if not R.on_curve(c, P):
        raise PtauFileError("PointNotOnCurve")
    return P


def write_ptau(c: R.Curve, g1_points, g2_raw: bytes, power: int, num_sections=NUM_SECTIONS_FULL, version=PTAU_VERSION,
               prime=None, magic=b"ptau") -> bytes:
    """write_ptau (ptau.rs:205-268).  g2_raw: the G2 section's bytes, opaque here (pairings stay on the host).
    The keyword overrides exist to build malformed files for the error-path tests."""
    w = io.BytesIO()
    w.write(magic)
    w.write(struct.pack("<II", version, num_sections))
    # header section (write_header, ptau.rs:170-190)
    w.write(struct.pack("<Iq", 1, 4 + 32 + 4))
    w.write(struct.pack("<I", 32))
    w.write((c.p if prime is None else prime).to_bytes(32, "little"))
    w.write(struct.pack("<I", power))
    if num_sections == NUM_SECTIONS_FULL:
        w.write(struct.pack("<Iq", 0, 0))
        for sid in range(4, NUM_SECTIONS_FULL):
            w.write(struct.pack("<Iq", sid, 0))
    g1 = b"".join(raw_point(c, P) for P in g1_points)
    w.write(struct.pack("<Iq", 2, len(g1)))
    w.write(g1)
    w.write(struct.pack("<Iq", 3, len(g2_raw)))
    w.write(g2_raw)
    return w.getvalue()


def read_ptau_g1(c: R.Curve, data: bytes, num_g1: int, num_g2: int):
    """read_ptau restricted to the G1 side (ptau.rs:270-436)."""
    r = io.BytesIO(data)

    def rd(n):
        b = r.read(n)
        if len(b) != n:
            raise PtauFileError("IoError")
        return b
    if rd(4) != b"ptau":
        raise PtauFileError("InvalidHead")
    version, num_sections = struct.unpack("<II", rd(8))
    if version != PTAU_VERSION:
        raise PtauFileError("UnsupportedVersion")
    if num_sections not in (NUM_SECTIONS_FULL, NUM_SECTIONS_PRUNED):
        raise PtauFileError("InvalidNumSections")
    pos = {}
    for _ in range(num_sections):
        sid, size = struct.unpack("<Iq", rd(12))
        if sid in (1, 2, 3):
            pos[sid] = r.tell()
        r.seek(size, io.SEEK_CUR)
    assert all(pos.get(k) for k in (1, 2, 3))
    r.seek(pos[1])
    (n8,) = struct.unpack("<I", rd(4))
    if int.from_bytes(rd(n8), "little") != c.p:
        raise PtauFileError("InvalidPrime")
    (power,) = struct.unpack("<I", rd(4))
    if num_g1 > (1 << power) * 2 - 1:
        raise PtauFileError("InsufficientPowerForG1")
    if num_g2 > (1 << power):
        raise PtauFileError("InsufficientPowerForG2")
    r.seek(pos[2])
    return [parse_raw_point(c, rd(64)) for _ in range(num_g1)]


def write_pedersen_key(c: R.Curve, h, ck) -> bytes:
    """save_setup (pedersen.rs:383-393)"""
    return KEY_FILE_HEAD + raw_point(c, h) + b"".join(raw_point(c, P) for P in ck)


def read_pedersen_key(c: R.Curve, data: bytes, n: int):
    """load_setup (pedersen.rs:318-340): returns (h, ck) with len(ck) = n.next_power_of_two()"""
    num = 1 if n <= 1 else 1 << (n - 1).bit_length()
    if data[:12] != KEY_FILE_HEAD:
        raise PtauFileError("InvalidHead")
    need = 12 + 64 * (num + 1)
    if len(data) < need:
        raise PtauFileError("IoError")
    pts = [parse_raw_point(c, data[12 + 64 * i: 76 + 64 * i]) for i in range(num + 1)]
    return pts[0], pts[1:]
