"""-m gpu: on-disk keys -> HBM through the C ABI (nmx_bases_register_ptau / _keyfile, NMX_BASES_VALIDATE): the key read
back equals the points the oracle's writer put in the file, MSMs over it match the oracle, and the point checks of
read_points (ptau.rs:372-391) run on the device."""
import ctypes

import numpy as np
import pytest

from oracle import cref
from oracle import keyfiles as K
from oracle import pyref as R
from tests import util

pytestmark = pytest.mark.gpu


def _xy(points):
    return np.frombuffer(b"".join(R.point_to_xy64(P) for P in points), dtype=np.uint8).reshape(-1, 64)


@pytest.mark.parametrize("c", [R.BN254_G1, R.VESTA], ids=lambda c: c.name)
def test_ptau_to_hbm_and_commit(nmx, tmp_path, c):
    import nova_amd
    n = 5000   # the loaded key gets window tables (every registered key with NMX_BASES_PRECOMPUTE does)
    key = cref.sequential_bases(c, 11, 8192)
    pts = [R.xy64_to_point(bytes(row)) for row in key]
    pts[7] = R.INF                                   # an identity base survives the file
    key = _xy(pts)
    for ns in (K.NUM_SECTIONS_FULL, K.NUM_SECTIONS_PRUNED):
        path = tmp_path / f"k{ns}.ptau"
        path.write_bytes(K.write_ptau(c, pts, bytes(256), power=13, num_sections=ns))
        ck = nova_amd.CommitmentKey.load_ptau(c.cid, path, n)
        assert len(ck) == 8192                       # n.next_power_of_two()
        assert ck.read(0, 8192).tobytes() == key.tobytes()
        sc = util.scalar_set(c.cid, n, "random")
        got = nova_amd.DlogGroup(c.cid).vartime_multiscalar_mul(sc, ck)
        assert (got.xy, int(got.is_inf)) == cref.msm(c.cid, sc, key[:n], n)
        ck.close()


def test_ptau_point_checks_on_device(nmx, tmp_path):
    import nova_amd
    from nova_amd import _lib as L
    c = R.BN254_G1
    pts = R.sequential_bases(c, 3, 300)
    good = K.write_ptau(c, pts, b"", power=9)
    start = good.index(K.raw_point(c, pts[0]))
    # point 200 off the curve; coordinate of point 299 not canonical
    off = bytearray(good)
    off[start + 64 * 200 + 32: start + 64 * 201] = K.raw_point(c, (1, 3))[32:]
    nc = bytearray(good)
    nc[start + 64 * 255: start + 64 * 255 + 32] = (c.p).to_bytes(32, "little")
    for name, data in (("off", off), ("nc", nc)):
        path = tmp_path / f"{name}.ptau"
        path.write_bytes(bytes(data))
        with pytest.raises(nova_amd.NmxError) as e:
            nova_amd.CommitmentKey.load_ptau(c.cid, path, 256)
        assert e.value.code == L.E_POINT
        ck = nova_amd.CommitmentKey.load_ptau(c.cid, path, 128)   # only the points actually loaded are checked
        assert ck.read(0, 128).tobytes() == _xy(pts[:128]).tobytes()
        ck.close()
    # the same checks on a host key
    xy = _xy(pts).copy()
    ck = nova_amd.CommitmentKey.from_host_validated(c.cid, xy)
    ck.close()
    xy[17, 32] ^= 1
    with pytest.raises(nova_amd.NmxError) as e:
        nova_amd.CommitmentKey.from_host_validated(c.cid, xy)
    assert e.value.code == L.E_POINT
    # truncated point section
    path = tmp_path / "short.ptau"
    path.write_bytes(good[: start + 64 * 100 + 5])
    with pytest.raises(nova_amd.NmxError) as e:
        nova_amd.CommitmentKey.load_ptau(c.cid, path, 256)
    assert e.value.code == L.E_IO


@pytest.mark.parametrize("c", [R.GRUMPKIN, R.PALLAS], ids=lambda c: c.name)
def test_pedersen_keyfile_to_hbm(nmx, tmp_path, c):
    import nova_amd
    pts = R.sequential_bases(c, 21, 65)
    path = tmp_path / "ck.key"
    path.write_bytes(K.write_pedersen_key(c, pts[0], pts[1:]))
    ck = nova_amd.CommitmentKey.load_keyfile(c.cid, path, 60)     # -> 64 points
    assert len(ck) == 64 and ck.h == R.point_to_xy64(pts[0])
    assert ck.read(0, 64).tobytes() == _xy(pts[1:]).tobytes()
    v = util.scalar_set(c.cid, 64, "random")
    r = util.scalar_set(c.cid, 1, "random")
    got = nova_amd.CommitmentEngine(c.cid).commit(ck, v, r)
    assert (got.xy, int(got.is_inf)) == cref.commit(c.cid, v, _xy(pts[1:]), 64, R.point_to_xy64(pts[0]), r)
    ck.close()


def test_large_file_streams_through_both_staging_buffers(nmx, tmp_path):
    """> 2 x 16 MiB of points: both pinned buffers are reused; content equals the generator's."""
    import nova_amd
    c = R.BN254_G1
    n = (1 << 19) + 12345
    gen = nova_amd.CommitmentKey.generate(c.cid, n, k0=1, precompute=False)
    xy = gen.read(0, n)
    gen.close()
    p, Rm = c.p, 1 << 256
    # canonical -> raw Montgomery records with numpy object ints would be slow; do it via Python ints in bulk
    raw = bytearray(64 * n)
    mv = memoryview(xy.tobytes())
    for i in range(2 * n):
        raw[32 * i: 32 * i + 32] = (int.from_bytes(mv[32 * i: 32 * i + 32], "little") * Rm % p).to_bytes(32, "little")
    import struct
    head = b"ptau" + struct.pack("<II", 1, 3) + struct.pack("<Iq", 1, 40) + struct.pack("<I", 32) + p.to_bytes(32, "little") + \
        struct.pack("<I", 20) + struct.pack("<Iq", 2, len(raw))
    path = tmp_path / "big.ptau"
    path.write_bytes(head + bytes(raw) + struct.pack("<Iq", 3, 0))
    h = ctypes.c_uint64(0)
    from nova_amd import _lib as L
    from nova_amd.provider import _check
    _check(L.lib().nmx_bases_register_ptau(c.cid, str(path).encode(), n, 2, 0, ctypes.byref(h)))
    ck = nova_amd.CommitmentKey(c.cid, h.value, n, bytes(64))
    assert ck.read(0, n).tobytes() == xy.tobytes()
    ck.close()


def test_committed_key_files_to_hbm(nmx):
    """The committed fixtures (tests/golden/keys) through the product loaders."""
    import os
    import nova_amd
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keys")
    ck = nova_amd.CommitmentKey.load_ptau(R.BN254_G1.cid, os.path.join(d, "tiny_bn254.ptau"), 4)
    assert ck.read(0, 4).tobytes() == _xy(R.sequential_bases(R.BN254_G1, 7, 4)).tobytes()
    ck.close()
    pts = R.sequential_bases(R.PALLAS, 5, 5)
    ck = nova_amd.CommitmentKey.load_keyfile(R.PALLAS.cid, os.path.join(d, "tiny_pallas.key"), 4)
    assert ck.h == R.point_to_xy64(pts[0]) and ck.read(0, 4).tobytes() == _xy(pts[1:]).tobytes()
    ck.close()
