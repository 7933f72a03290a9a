"""include/nova_mi355x.hpp (the C++ host mirror of DlogGroupExt / CommitmentEngine) compiled with g++ and run against
the oracle.  Without a GPU the binary must stop with NMX_E_NO_DEVICE (exit 3); on the GPU box it must pass."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.bin")


def build():
    import __graft_entry__
    __graft_entry__.build()
    deps = [SRC, os.path.join(ROOT, "include", "nova_mi355x.hpp"), os.path.join(ROOT, "include", "nova_mi355x.h")]
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", BIN, SRC,
                               "-L" + os.path.join(ROOT, "nova_amd"), "-lnova_mi355x",
                               "-L" + os.path.join(ROOT, "oracle"), "-lnova_ref",
                               "-Wl,-rpath," + os.path.join(ROOT, "nova_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                               "-Wl,-rpath,/opt/rocm/lib"])
    return BIN


def test_cpp_mirror_compiles_and_refuses_without_gpu():
    from nova_amd import _lib
    if _lib.lib().nmx_device_count() > 0:
        pytest.skip("a HIP device is visible")
    r = subprocess.run([build()], capture_output=True, text=True)
    assert r.returncode == 3, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_cpp_mirror_on_gpu(tmp_path):
    from oracle import keyfiles as K
    from oracle import pyref as R
    b = BIN if os.path.exists(BIN) else build()
    # key files for the load_setup part of the test: ck = P_{12345 + i}, i < 128, h = P_{12345 + 100}
    for c in (R.BN254_G1, R.PALLAS):
        pts = R.sequential_bases(c, 12345, 128)
        (tmp_path / f"curve{c.cid}.key").write_bytes(K.write_pedersen_key(c, pts[100], pts))
    r = subprocess.run([b], capture_output=True, text=True, env=dict(os.environ, NMX_TEST_KEYDIR=str(tmp_path)))
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)


def test_cpp_chained_replay_driver_builds_and_refuses_without_gpu(tmp_path):
    """bench/csnark_replay.cpp (the chained CompressedSNARK::prove replay through include/nova_mi355x.hpp's `resident` functions) compiles
    with g++ against the product library and, like everything in the product path, stops with NMX_E_NO_DEVICE when there is no GPU
    (exit 3) -- no CPU fall-back.  On the GPU box it is exercised by tests/test_gpu_large.py::test_compressed_snark_replay_matches_oracle."""
    import bench
    from nova_amd import _lib
    b = bench.build_cpp_driver()
    assert os.path.exists(b)
    if _lib.lib().nmx_device_count() > 0:
        pytest.skip("a HIP device is visible")
    r = subprocess.run([b, os.devnull, str(tmp_path / "out.bin"), "1", "0"], capture_output=True, text=True)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "no HIP device" in r.stderr
