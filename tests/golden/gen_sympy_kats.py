"""Generates tests/golden/sympy_kats.json: scalar multiples and small MSMs on all four curves of the path, computed with
SymPy's elliptic-curve arithmetic (sympy.ntheory.elliptic_curve.EllipticCurve over GF(p): third-party code, not this
repository's oracle and not its kernels).  The curve constants typed here are the reference's own strings
(/root/reference/src/provider/bn256_grumpkin.rs:39-40,84-85, pasta.rs:37-38,45-46; curve equations SURVEY.md 8(a)); nothing is
imported from oracle/ or nova_amd/.  Run once in the build container (sympy 1.14): python tests/golden/gen_sympy_kats.py
tests/test_oracle.py checks both oracle tiers against the file, tests/test_gpu_public_kats.py the HIP path."""
import hashlib
import json
import os

from sympy.ntheory.elliptic_curve import EllipticCurve

BN_Q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
BN_R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
PA_P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
PA_Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
CURVES = {  # name: (base modulus, group order, b of y^2 = x^3 + b, generator)
    "bn254_g1": (BN_Q, BN_R, 3, (1, 2)),
    "grumpkin": (BN_R, BN_Q, -17, (1, 0x2cf135e7506a45d632d270d45f1181294833fc48d823f272c)),
    "pallas": (PA_P, PA_Q, 5, (PA_P - 1, 2)),
    "vesta": (PA_Q, PA_P, 5, (PA_Q - 1, 2)),
}


def scalar(label, r):
    return int.from_bytes(hashlib.sha256(label.encode()).digest() + hashlib.sha256((label + "'").encode()).digest(), "big") % r


def xy(P, p):
    return {"x": "%064x" % (int(P.x) % p), "y": "%064x" % (int(P.y) % p)}


out = {"_provenance": __doc__}
for name, (p, r, b, g) in CURVES.items():
    E = EllipticCurve(0, b % p, modulus=p)
    G = E(*g)
    rows = []
    ks = [2, 3, 5, 9, (1 << 128) + 12345, r - 1, r - 2] + [scalar(f"{name}/{i}", r) for i in range(4)]
    for k in ks:
        rows.append({"k": "%064x" % k, **xy(k * G, p)})
    # an MSM: bases P_i = (i + 7) G, scalars from the same labelled stream, expected sum by SymPy additions
    bases = [(i + 7) * G for i in range(6)]
    sc = [scalar(f"{name}/msm/{i}", r) for i in range(6)]
    sc[2] = 0            # a zero scalar
    sc[4] = r - 1        # and -1
    acc = None
    for s, P in zip(sc, bases):
        if s == 0:
            continue
        t = s * P
        acc = t if acc is None else acc + t
    out[name] = {"p": "%064x" % p, "r": "%064x" % r, "gen": xy(G, p), "mul": rows,
                 "msm": {"bases": [xy(P, p) for P in bases], "scalars": ["%064x" % s for s in sc], "sum": xy(acc, p)}}
    # the order annihilates the generator (SymPy's own arithmetic): r G = O
    assert (r * G).z == 0 if hasattr(r * G, "z") else True
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sympy_kats.json"), "w"), indent=1)
print("written", {k: len(v["mul"]) for k, v in out.items() if k[0] != "_"})
