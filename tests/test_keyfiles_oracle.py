"""CPU: the oracle's restatement of the reference's key-file formats (oracle/keyfiles.py): writer -> reader round trips
and every error branch of read_meta_data / read_header / read_points (ptau.rs:270-391, pedersen.rs:318-340)."""
import pytest

from oracle import keyfiles as K
from oracle import pyref as R


def _pts(c, n, k0=5):
    return R.sequential_bases(c, k0, n)


@pytest.mark.parametrize("c", [R.BN254_G1, R.GRUMPKIN, R.PALLAS, R.VESTA], ids=lambda c: c.name)
def test_ptau_roundtrip_full_and_pruned(c):
    pts = _pts(c, 9) + [R.INF]
    g2 = bytes(range(256))  # opaque here
    for ns in (K.NUM_SECTIONS_FULL, K.NUM_SECTIONS_PRUNED):
        data = K.write_ptau(c, pts, g2, power=4, num_sections=ns)
        assert K.read_ptau_g1(c, data, 10, 2) == pts
        assert K.read_ptau_g1(c, data, 4, 2) == pts[:4]   # a prefix of the section
    # layout facts of write_ptau (ptau.rs:170-268): magic, version, 11 sections, header section first
    data = K.write_ptau(c, pts, g2, power=4)
    assert data[:4] == b"ptau" and data[4:12] == (1).to_bytes(4, "little") + (11).to_bytes(4, "little")
    assert data[12:24] == (1).to_bytes(4, "little") + (40).to_bytes(8, "little")
    assert int.from_bytes(data[28:60], "little") == c.p and data[60:64] == (4).to_bytes(4, "little")


def test_ptau_error_branches():
    c = R.BN254_G1
    pts = _pts(c, 4)
    ok = K.write_ptau(c, pts, b"", power=2)
    cases = [
        (K.write_ptau(c, pts, b"", 2, magic=b"ptax"), "InvalidHead"),
        (K.write_ptau(c, pts, b"", 2, version=2), "UnsupportedVersion"),
        (K.write_ptau(c, pts, b"", 2, prime=R.BN254_R), "InvalidPrime"),
    ]
    for data, msg in cases:
        with pytest.raises(K.PtauFileError, match=msg):
            K.read_ptau_g1(c, data, 4, 2)
    bad = bytearray(ok)
    bad[8:12] = (5).to_bytes(4, "little")
    with pytest.raises(K.PtauFileError, match="InvalidNumSections"):
        K.read_ptau_g1(c, bytes(bad), 4, 2)
    with pytest.raises(K.PtauFileError, match="InsufficientPowerForG1"):
        K.read_ptau_g1(c, ok, 8, 2)       # 2^(2+1) - 1 = 7 < 8
    with pytest.raises(K.PtauFileError, match="InsufficientPowerForG2"):
        K.read_ptau_g1(c, ok, 4, 5)
    # off-curve and non-canonical points (read_points, ptau.rs:372-391)
    off = K.write_ptau(c, pts[:3] + [(1, 3)], b"", 2)
    with pytest.raises(K.PtauFileError, match="PointNotOnCurve"):
        K.read_ptau_g1(c, off, 4, 2)
    assert K.read_ptau_g1(c, off, 3, 2) == pts[:3]     # only the points actually read are checked
    g1_start = ok.index(K.raw_point(c, pts[0]))
    nc = bytearray(ok)
    nc[g1_start: g1_start + 32] = (c.p).to_bytes(32, "little")  # x = p: not canonical
    with pytest.raises(K.PtauFileError, match="PointNotOnCurve"):
        K.read_ptau_g1(c, bytes(nc), 4, 2)
    with pytest.raises(K.PtauFileError, match="IoError"):
        K.read_ptau_g1(c, ok[: g1_start + 100], 4, 2)


@pytest.mark.parametrize("c", [R.GRUMPKIN, R.PALLAS], ids=lambda c: c.name)
def test_pedersen_key_roundtrip(c):
    pts = _pts(c, 9)
    data = K.write_pedersen_key(c, pts[0], pts[1:])
    h, ck = K.read_pedersen_key(c, data, 8)
    assert h == pts[0] and ck == pts[1:]
    h, ck = K.read_pedersen_key(c, data, 3)   # n.next_power_of_two() = 4
    assert len(ck) == 4 and ck == pts[1:5]
    with pytest.raises(K.PtauFileError, match="InvalidHead"):
        K.read_pedersen_key(c, b"PEDERSEN_KEX" + data[12:], 8)
    with pytest.raises(K.PtauFileError, match="IoError"):
        K.read_pedersen_key(c, data, 9)       # needs 16 + 1 points


def test_committed_key_files():
    """tests/golden/keys/* (made by tests/golden/make_key_files.py) read back to the points they were written from."""
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keys")
    data = open(os.path.join(d, "tiny_bn254.ptau"), "rb").read()
    assert len(data) == 4 + 8 + 12 + 40 + 8 * 12 + 12 + 4 * 64 + 12 + 128
    assert K.read_ptau_g1(R.BN254_G1, data, 4, 2) == R.sequential_bases(R.BN254_G1, 7, 4)
    data = open(os.path.join(d, "tiny_pallas.key"), "rb").read()
    pts = R.sequential_bases(R.PALLAS, 5, 5)
    assert K.read_pedersen_key(R.PALLAS, data, 4) == (pts[0], pts[1:])
