"""CPU: nova_amd/csrc/host_fp4.hpp (the sum-check provers' host-side field arithmetic, 4 x 64-bit Montgomery limbs) compiled with
g++ and checked against Python integers: products, sums, differences, inverses, the Montgomery form, the plain-integer-times-
constant entry the device sums come through, and the conversion to the device's 9 x 29-bit internal residue (against fp.hpp's own)."""
import os
import subprocess

from tests import fv_common as fc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_fp4_against_python_integers(tmp_path):
    exe = os.path.join(ROOT, "tests", "cpp", "host_fp4_test.bin")
    src = os.path.join(ROOT, "tests", "cpp", "host_fp4_test.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src])
    out = subprocess.check_output([exe, "120"], text=True)
    n = ninv = nfast = 0
    for line in out.splitlines():
        kv = dict(tok.split("=") for tok in line.split())
        if "invfid" in kv:      # the fast (31 steps at a time) and the bit-at-a-time inversion on stress values
            p = fc.FIELDS[int(kv["invfid"])]
            a = int(kv["a"], 16)
            want = pow(a, -1, p) if a else 0
            assert int(kv["inv"], 16) == want == int(kv["inv_slow"], 16), line
            ninv += 1
            nfast += int(kv["fast"])
            continue
        p = fc.FIELDS[int(kv["fid"])]
        a, b = int(kv["a"], 16), int(kv["b"], 16)
        assert int(kv["mul"], 16) == a * b % p
        assert int(kv["add"], 16) == (a + b) % p
        assert int(kv["sub"], 16) == (a - b) % p
        assert int(kv["inv"], 16) == (pow(a, -1, p) if a else 0)
        assert int(kv["mont"], 16) == a * (1 << 256) % p
        assert int(kv["back"], 16) == a
        assert int(kv["pt"], 16) == a * pow(2, int(kv["e"]), p) * pow(1 << 256, -1, p) % p   # (plain X) x (element 2^e) / 2^256
        assert int(kv["dev"], 16) == a * (1 << 261) % p == int(kv["dev_ref"], 16)
        n += 1
    assert n == 480
    assert ninv == 4 * (1 + 80 + 3 * 174 + 2400) and nfast == ninv - 8      # the fast path converged on everything but the two zeros per field
