"""Generates tests/golden/msm_vectors.json: small MSM / commit cases with their results from the tier-1 oracle
(oracle/pyref.py, Python big-int, the definition sum_i s_i * P_i).  These are ORACLE-generated regression vectors --
the reference stores no MSM output vectors and cannot be run here; they freeze today's answers so a
later change to either oracle tier or to the HIP path cannot drift silently.
    python tests/golden/make_msm_vectors.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref as R  # noqa: E402
from tests import util  # noqa: E402

out = {"_comment": __doc__.strip(), "cases": []}
for c in R.CURVES.values():
    for n, kind, k0 in ((1, "random", 3), (17, "random", 1000), (33, "pm_small", 77), (20, "zero_rm1", 5), (12, "equal", 9)):
        bases = R.sequential_bases(c, k0, n)
        if n >= 17:
            bases[4] = R.INF  # an identity base
        sc = util.scalar_set(c.cid, n, kind, seed=util.SEED + n)
        ints = [int.from_bytes(bytes(r), "little") for r in sc]
        res = R.msm_naive(c, ints, bases)
        h = R.mul(c, 424242, (c.gx, c.gy))
        r = int.from_bytes(bytes(util.random_scalars(c.cid, 1, seed=n)[0]), "little")
        com = R.commit(c, bases, h, ints, r)
        out["cases"].append({
            "curve": c.name, "cid": c.cid, "n": n, "kind": kind, "k0": k0,
            "scalars_hex": [format(x, "064x") for x in ints],
            "bases_xy64_hex": [R.point_to_xy64(P).hex() for P in bases],
            "msm_xy64_hex": R.point_to_xy64(res).hex(), "msm_is_inf": res is R.INF,
            "h_xy64_hex": R.point_to_xy64(h).hex(), "r_hex": format(r, "064x"),
            "commit_xy64_hex": R.point_to_xy64(com).hex(),
        })
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "msm_vectors.json"), "w"), indent=1)
print(len(out["cases"]), "cases written")
