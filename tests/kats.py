"""Loader for tests/golden/public_kats.json: the EIP-196 ecMul vectors as (bases, scalars, expected) byte arrays.
The scalars of these vectors are integers < 2^256 that the precompile does NOT reduce first; all of the committed ones
are < r except chfast2's (= q - 1 > r), which is reduced mod r here as `Fr::from_repr` callers must (k*P = (k mod r)*P)."""
import json
import os

import numpy as np

from oracle import pyref as R

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "public_kats.json")


def load():
    rows = json.load(open(PATH))["bn254_g1_ecmul"]
    c = R.BN254_G1
    out = []
    for v in rows:
        P = (int(v["x"], 16), int(v["y"], 16))
        k = int(v["k"], 16) % c.r
        Q = (int(v["rx"], 16), int(v["ry"], 16))
        out.append((v["name"], P, k, Q))
    return out


def as_arrays(cases):
    bases = np.zeros((len(cases), 64), np.uint8)
    sc = np.zeros((len(cases), 32), np.uint8)
    for i, (_n, P, k, _Q) in enumerate(cases):
        bases[i] = np.frombuffer(R.point_to_xy64(P), np.uint8)
        sc[i] = np.frombuffer(k.to_bytes(32, "little"), np.uint8)
    return bases, sc
