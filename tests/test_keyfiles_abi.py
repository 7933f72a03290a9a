"""CPU (no GPU): the key-file loaders of the C ABI parse and reject headers before any device work, so every header
error branch of read_meta_data / read_header (ptau.rs:270-370) and the Pedersen head check (pedersen.rs:324-330) is
checked here; a well-formed file then fails loudly with NMX_E_NO_DEVICE (there is no CPU fallback)."""
import ctypes

import pytest

from nova_amd import _lib as L
from oracle import keyfiles as K
from oracle import pyref as R


def _reg_ptau(path, curve, n1, n2):
    h = ctypes.c_uint64(0)
    rc = L.lib().nmx_bases_register_ptau(curve, str(path).encode(), n1, n2, 0, ctypes.byref(h))
    return rc, L.lib().nmx_last_error().decode()


def test_ptau_header_errors(tmp_path):
    import torch
    c = R.BN254_G1
    pts = R.sequential_bases(c, 3, 4)
    ok = K.write_ptau(c, pts, bytes(128), power=2)
    bad_sections = bytearray(ok)
    bad_sections[8:12] = (5).to_bytes(4, "little")
    cases = {
        "magic": (K.write_ptau(c, pts, b"", 2, magic=b"ptax"), 4, 2, L.E_FORMAT, "InvalidHead"),
        "version": (K.write_ptau(c, pts, b"", 2, version=3), 4, 2, L.E_FORMAT, "UnsupportedVersion"),
        "sections": (bytes(bad_sections), 4, 2, L.E_FORMAT, "InvalidNumSections"),
        "prime": (K.write_ptau(c, pts, b"", 2, prime=R.BN254_R), 4, 2, L.E_FORMAT, "InvalidPrime"),
        "power_g1": (ok, 8, 2, L.E_FORMAT, "InsufficientPowerForG1"),
        "power_g2": (ok, 4, 5, L.E_FORMAT, "InsufficientPowerForG2"),
        "truncated": (ok[:30], 4, 2, L.E_IO, "IoError"),
    }
    for name, (data, n1, n2, code, msg) in cases.items():
        p = tmp_path / f"{name}.ptau"
        p.write_bytes(data)
        rc, err = _reg_ptau(p, c.cid, n1, n2)
        assert rc == code and msg in err, (name, rc, err)
    rc, err = _reg_ptau(tmp_path / "missing.ptau", c.cid, 4, 2)
    assert rc == L.E_IO
    # the file's prime selects the curve: a BN254 file is not a Grumpkin key
    p = tmp_path / "ok.ptau"
    p.write_bytes(ok)
    assert _reg_ptau(p, R.GRUMPKIN.cid, 4, 2)[0] == L.E_FORMAT
    if not torch.cuda.is_available():
        rc, err = _reg_ptau(p, c.cid, 4, 2)
        assert rc == L.E_NO_DEVICE, (rc, err)   # header accepted; the product path needs the GPU


def test_pedersen_keyfile_head(tmp_path):
    import torch
    c = R.PALLAS
    pts = R.sequential_bases(c, 3, 5)
    data = K.write_pedersen_key(c, pts[0], pts[1:])
    h = ctypes.c_uint64(0)
    hxy = (ctypes.c_uint8 * 64)()

    def reg(path, n):
        return L.lib().nmx_bases_register_keyfile(c.cid, str(path).encode(), n, 0, ctypes.byref(h), hxy)
    (tmp_path / "bad.key").write_bytes(b"PEDERSEN_KEX" + data[12:])
    assert reg(tmp_path / "bad.key", 4) == L.E_FORMAT
    (tmp_path / "short.key").write_bytes(data[:40])
    assert reg(tmp_path / "short.key", 4) == L.E_IO
    # h off the curve / non-canonical is rejected on the host, before any device work
    off = bytearray(data)
    off[12 + 32: 12 + 64] = K.raw_point(c, (0, 7))[32:]
    (tmp_path / "off.key").write_bytes(bytes(off))
    assert reg(tmp_path / "off.key", 4) == L.E_POINT
    nc = bytearray(data)
    nc[12: 12 + 32] = (c.p + 1).to_bytes(32, "little")
    (tmp_path / "nc.key").write_bytes(bytes(nc))
    assert reg(tmp_path / "nc.key", 4) == L.E_POINT
    if not torch.cuda.is_available():
        (tmp_path / "ok.key").write_bytes(data)
        assert reg(tmp_path / "ok.key", 4) == L.E_NO_DEVICE
