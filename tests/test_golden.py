"""Committed golden fixtures (tests/golden/): the C oracle on CPU, and the HIP path on GPU, against the frozen
tier-1 answers.  See tests/golden/make_msm_vectors.py for what these vectors are (oracle-generated regression vectors;
the reference stores none for this path)."""
import json
import os

import numpy as np
import pytest

from oracle import cref

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msm_vectors.json")))


def unpack(case):
    sc = np.frombuffer(b"".join(int(x, 16).to_bytes(32, "little") for x in case["scalars_hex"]), np.uint8).reshape(-1, 32)
    bases = np.frombuffer(bytes.fromhex("".join(case["bases_xy64_hex"])), np.uint8).reshape(-1, 64)
    h = np.frombuffer(bytes.fromhex(case["h_xy64_hex"]), np.uint8)
    r = np.frombuffer(int(case["r_hex"], 16).to_bytes(32, "little"), np.uint8)
    return sc, bases, h, r


@pytest.mark.parametrize("case", G["cases"], ids=lambda c: f"{c['curve']}-{c['n']}-{c['kind']}")
def test_c_oracle_matches_golden(case):
    sc, bases, h, r = unpack(case)
    assert cref.msm(case["cid"], sc, bases, case["n"]) == (bytes.fromhex(case["msm_xy64_hex"]), int(case["msm_is_inf"]))
    assert cref.commit(case["cid"], sc, bases, case["n"], h, r)[0] == bytes.fromhex(case["commit_xy64_hex"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", G["cases"], ids=lambda c: f"{c['curve']}-{c['n']}-{c['kind']}")
def test_hip_matches_golden(nmx, case):
    sc, bases, h, r = unpack(case)
    g = nmx.DlogGroup(case["cid"])
    got = g.vartime_multiscalar_mul(sc, bases)
    assert (got.xy, got.is_inf) == (bytes.fromhex(case["msm_xy64_hex"]), case["msm_is_inf"])
    ck = nmx.CommitmentKey.from_host(case["cid"], bases, h.tobytes())
    assert nmx.CommitmentEngine(case["cid"]).commit(ck, sc, r).xy == bytes.fromhex(case["commit_xy64_hex"])
    ck.close()
