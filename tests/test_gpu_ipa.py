"""-m gpu: nmx_ipa_prove (InnerProductArgument::prove, /root/reference/src/provider/ipa_pc.rs:174-281, without the key fold:
nova_amd/csrc/ipa.hpp) through the C ABI against (1) the reference's verifier (tests/ipa_common.py) and (2) the oracle's
restatement WITH the key fold (oracle/nova_ref.c ref_ipa_prove) under the same stand-in transcript: every L, R, challenge and
a_hat identical."""
import numpy as np
import pytest

from oracle import cref
from oracle import pyref as R
from tests import ipa_common as ic
from tests import util

pytestmark = pytest.mark.gpu
CURVES = [R.BN254_G1, R.GRUMPKIN, R.PALLAS, R.VESTA]


def dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x).copy()).cuda()


def gpu_prove(nmx, curve, device=True, mont=False, precompute=True, key_extra=0):
    def prove(ck, ckc, a, b, n, tr):
        key = ck
        if key_extra:   # a registered key longer than the vectors: `ck.split_at(U.b_vec.len())` (ipa_pc.rs:183)
            key = cref.sequential_bases(curve, 77, n + key_extra).copy()
            assert (key[:n] == ck).all()
        if mont:
            K = nmx.CommitmentKey.from_host(curve.cid, util.to_mont_bases(curve.cid, key), mont=True, precompute=precompute)
            u = util.to_mont_bases(curve.cid, ckc)
            aa, bb = util.to_mont_scalars(curve.cid, a), util.to_mont_scalars(curve.cid, b)
            p = curve.r
            rinv256 = pow(1 << 256, p - 2, p)

            def tr_m(L, Li, Rr, Ri):   # points come canonical (as every result of the library); the challenge goes back in Montgomery form
                return ic.le(int.from_bytes(tr(L, Li, Rr, Ri), "little") * (1 << 256) % p)
            Ls, Rs, infs, ah = nmx.ipa_prove(K, u, dev(aa) if device else aa, dev(bb) if device else bb, tr_m, mont=True)
            ah = ic.le(int.from_bytes(ah, "little") * rinv256 % p)
        else:
            K = nmx.CommitmentKey.from_host(curve.cid, key, precompute=precompute)
            Ls, Rs, infs, ah = nmx.ipa_prove(K, ckc, dev(a) if device else a, dev(b) if device else b, tr)
        K.close()
        return Ls, Rs, infs, ah
    return prove


def oracle_prove(curve):
    def prove(ck, ckc, a, b, n, tr):
        return cref.ipa_prove(curve.cid, ck, ckc, a, b, n, cref.make_ipa_transcript(tr))
    return prove


def both(nmx, curve, n, seed, **kw):
    force, mutate = kw.pop("force", None), kw.pop("mutate", None)
    got, tg = ic.check_ipa(gpu_prove(nmx, curve, **kw), curve, n, seed, force=force, mutate=mutate)
    want, tw = ic.check_ipa(oracle_prove(curve), curve, n, seed, force=force, mutate=mutate)
    assert tg.rs == tw.rs, "challenge sequences differ: some L or R does"
    assert got == want
    return got


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 2, 4, 64, 1024])
def test_proof_equals_the_oracle_and_verifies(nmx, curve, n):
    both(nmx, curve, n, seed=20 + n)


@pytest.mark.parametrize("curve", [R.GRUMPKIN, R.PALLAS], ids=lambda c: c.name)
def test_secondary_size_2p14(nmx, curve):
    """the size S2's evaluation argument runs at (the secondary circuit pads to 2^14)"""
    both(nmx, curve, 1 << 14, seed=9)


@pytest.mark.parametrize("kw", [dict(device=False), dict(mont=True), dict(mont=True, device=False), dict(precompute=False),
                                dict(key_extra=1000)], ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
def test_forms(nmx, kw):
    """host vectors, Montgomery layout (scalars, points, challenges, a_hat), a key without window tables (one MSM per vector
    instead of the fused pair), a key longer than the vectors"""
    both(nmx, R.GRUMPKIN, 256, seed=31, **kw)


def test_edge_vectors_and_challenges(nmx):
    curve = R.GRUMPKIN

    def zero_left(a, b):
        a[: a.shape[0] // 2] = 0

    def zero_all(a, b):
        a[:] = 0
    both(nmx, curve, 64, seed=2, mutate=zero_left)
    Ls, Rs, infs, ah = both(nmx, curve, 64, seed=2, mutate=zero_all)
    assert all(i == (True, True) for i in infs) and ah == bytes(32)
    both(nmx, curve, 64, seed=4, force={0: 1, 2: curve.r - 1, 5: 2})


def test_the_vectors_are_left_as_they_were(nmx):
    curve, n = R.GRUMPKIN, 128
    ck, ckc, a, b = ic.make_instance(curve, n, 8)
    K = nmx.CommitmentKey.from_host(curve.cid, ck)
    da, db = dev(a), dev(b)
    nmx.ipa_prove(K, ckc, da, db, ic.IpaTranscript(curve.r))
    assert (da.cpu().numpy() == a).all() and (db.cpu().numpy() == b).all()
    K.close()


def test_errors(nmx):
    from nova_amd import _lib
    curve, n = R.GRUMPKIN, 16
    ck, ckc, a, b = ic.make_instance(curve, n, 1)
    K = nmx.CommitmentKey.from_host(curve.cid, ck)
    for bad, code in ((dict(n=12), _lib.E_ARG), (dict(force={1: 0}), _lib.E_ZERO), (dict(force={0: curve.r}), _lib.E_SCALAR_RANGE)):
        m = bad.get("n", n)
        with pytest.raises(nmx.NmxError) as e:
            nmx.ipa_prove(K, ckc, dev(a[:m]), dev(b[:m]), ic.IpaTranscript(curve.r, force=bad.get("force")))
        assert e.value.code == code
    short = nmx.CommitmentKey.from_host(curve.cid, ck[:8])
    with pytest.raises(nmx.NmxError) as e:
        nmx.ipa_prove(short, ckc, dev(a), dev(b), ic.IpaTranscript(curve.r))
    assert e.value.code == _lib.E_HANDLE

    def boom(*_):
        raise RuntimeError("transcript failure")
    with pytest.raises(nmx.NmxError) as e:
        nmx.ipa_prove(K, ckc, dev(a), dev(b), boom)
    assert e.value.code == _lib.E_ARG
    # the library is still usable and correct afterwards
    Ls, Rs, infs, ah = nmx.ipa_prove(K, ckc, dev(a), dev(b), ic.IpaTranscript(curve.r))
    want = cref.ipa_prove(curve.cid, ck, ckc, a, b, n, cref.make_ipa_transcript(ic.IpaTranscript(curve.r)))
    assert (Ls, Rs, infs, ah) == want
    K.close(), short.close()


@pytest.mark.parametrize("cycle", ["bn254", "pasta"])
def test_bench_workload(nmx, cycle):
    """bench.py --workload ipa_replay (also the `ipa_prove_ms` block of the default line): equal to the oracle, accepted by the verifier"""
    import argparse
    import torch
    import bench
    out = bench.ipa_replay(argparse.Namespace(log2n=10, steps=2, warmup=1, no_cpu_baseline=False, cycle=cycle), torch)
    assert out["cpu_baseline"]["gpu_matches_cpu"] is True and out["cpu_baseline"]["checks"] == {"proof": True, "reference_verifier": True}


def test_raw_abi_without_the_identity_flags(nmx):
    """out_is_inf may be null (the shim that only wants the points); the call is otherwise the same"""
    import ctypes
    from nova_amd import _lib
    from tests import standin
    curve, n = R.GRUMPKIN, 32
    ck, ckc, a, b = ic.make_instance(curve, n, 4)
    K = nmx.CommitmentKey.from_host(curve.cid, ck)
    t1, t2 = standin.Transcript(seed=2), standin.Transcript(seed=2)
    oL, oR, ah = np.zeros(64 * 5, np.uint8), np.zeros(64 * 5, np.uint8), np.zeros(32, np.uint8)
    rc = _lib.lib().nmx_ipa_prove(K.handle, ckc.ctypes.data, a.ctypes.data, b.ctypes.data, n, 0, t1.fn_ipa(_lib.IPA_TRANSCRIPT_FN), t1.ctx,
                                  oL.ctypes.data, oR.ctypes.data, None, ah.ctypes.data)
    assert rc == 0
    Ls, Rs, infs, want_ah = cref.ipa_prove(curve.cid, ck, ckc, a, b, n, t2.fn_ipa(cref.IPA_TRANSCRIPT_FN), ctx=t2.ctx)
    assert oL.tobytes() == b"".join(Ls) and oR.tobytes() == b"".join(Rs) and ah.tobytes() == want_ah
    K.close()

