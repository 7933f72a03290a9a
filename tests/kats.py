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


SYMPY_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sympy_kats.json")


def load_sympy():
    """tests/golden/sympy_kats.json (SymPy's elliptic-curve arithmetic, gen_sympy_kats.py) ->
    {curve name: {"mul": [(k, Q)], "msm": (bases [P], scalars [k], sum Q), "gen": G}} with integer coordinates."""
    raw = json.load(open(SYMPY_PATH))
    pt = lambda v: (int(v["x"], 16), int(v["y"], 16))
    out = {}
    for name, c in raw.items():
        if name.startswith("_"):
            continue
        out[name] = {"p": int(c["p"], 16), "r": int(c["r"], 16), "gen": pt(c["gen"]),
                     "mul": [(int(v["k"], 16), pt(v)) for v in c["mul"]],
                     "msm": ([pt(v) for v in c["msm"]["bases"]], [int(k, 16) for k in c["msm"]["scalars"]], pt(c["msm"]["sum"]))}
    return out


def points_scalars(points, scalars):
    bases = np.zeros((len(points), 64), np.uint8)
    sc = np.zeros((len(points), 32), np.uint8)
    for i, (P, k) in enumerate(zip(points, scalars)):
        bases[i] = np.frombuffer(R.point_to_xy64(P), np.uint8)
        sc[i] = np.frombuffer(int(k).to_bytes(32, "little"), np.uint8)
    return bases, sc
