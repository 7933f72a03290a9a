"""Shared drivers for the field-vector tests: the same checks run against the oracle (CPU) and the HIP path (GPU)."""
import json
import os

import numpy as np

from oracle import pyref as R
from tests import util

FIELDS = {0: R.BN254_Q, 1: R.BN254_R, 2: R.PALLAS_P, 3: R.PALLAS_Q}
CURVE_WITH_SCALAR_FIELD = {0: 1, 1: 0, 2: 3, 3: 2}  # field id -> curve id whose scalar field it is
KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "field_kats.json")))


def rand_vec(fid, n, seed):
    return util.random_scalars(CURVE_WITH_SCALAR_FIELD[fid], n, seed=seed)


def ints(v):
    return [int.from_bytes(bytes(row), "little") for row in np.asarray(v).reshape(-1, 32)]


def vec(vals):
    return np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in vals), dtype=np.uint8).reshape(-1, 32)


def to_bytes(x):
    return bytes(x) if isinstance(x, (bytes, bytearray)) else np.asarray(x).tobytes()


def check_kats2(eq_evals, mle_evaluate, spmv):
    """eq_evals(fid, r_vec) -> bytes; mle_evaluate(fid, z_vec, r_vec) -> bytes; spmv(fid, indptr, indices, data_vec, cols, z_vec) -> bytes"""
    for fid in FIELDS:
        for case in KATS["eq_evals"]["cases"]:
            assert ints(np.frombuffer(to_bytes(eq_evals(fid, vec(case["r"]))), np.uint8)) == case["evals"]
        for case in KATS["mle_bind_top_eval"]["cases"]:
            assert ints(np.frombuffer(to_bytes(mle_evaluate(fid, vec(case["evals"]), vec(case["point"]))), np.uint8)) == [case["eval"]]
        for case in KATS["spmv"]["cases"]:
            got = spmv(fid, case["indptr"], case["indices"], vec(case["data"]), case["cols"], vec(case["z"]))
            assert ints(np.frombuffer(to_bytes(got), np.uint8)) == case["out"]


def random_csr(fid, rows, cols, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    counts = rng.integers(0, 9, size=rows)
    counts[rows // 2] = 40  # one long row
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    nnz = int(indptr[-1])
    indices = rng.integers(0, cols, size=nnz).astype(np.uint64)
    data = rand_vec(fid, nnz, seed + 1).copy()
    p = FIELDS[fid]
    for i, x in enumerate([1, p - 1, 2, p - 7, 0]):  # the reference special-cases +-1 and small coefficients
        data[(5 * i) % nnz] = util.int_to_le32(x)
    return indptr, indices, data


def check_kats(fold_pairs, bind_top):
    """fold_pairs(fid, poly_vec, x_vec) / bind_top(fid, evals_vec, r_vec) -> bytes; all four fields."""
    for fid in FIELDS:
        for case in KATS["hyperkzg_fold_eval"]["cases"]:
            cur = vec(case["poly"])
            for x in reversed(case["point"]):
                cur = np.frombuffer(to_bytes(fold_pairs(fid, cur, vec([x]))), np.uint8).reshape(-1, 32)
            assert ints(cur) == [case["eval"]], (fid, case)
        for case in KATS["mle_bind_top_eval"]["cases"]:
            cur = vec(case["evals"])
            for r in case["point"]:
                cur = np.frombuffer(to_bytes(bind_top(fid, cur, vec([r]))), np.uint8).reshape(-1, 32)
            assert ints(cur) == [case["eval"]], (fid, case)


def edge_vectors(fid, n, seed):
    """random vectors with 0, 1, p-1 sprinkled in"""
    p = FIELDS[fid]
    v = rand_vec(fid, n, seed).copy()
    for i, x in enumerate([0, 1, p - 1, p - 2, 2]):
        if i < n:
            v[(7 * i + seed) % n] = util.int_to_le32(x)
    return v


def mixed_coefficient_case(fid):
    """the reference's mixed-coefficient SpMV fixture (src/r1cs/sparse.rs:486-520) as CSR over field `fid`, with the dense integer products
    of tests/golden/field_kats.json reduced mod p: (indptr, indices, data, cols, z, z2, out, out2)"""
    k = KATS["spmv_mixed_coefficients"]
    p = FIELDS[fid]
    rows, cols = k["rows"], k["cols"]
    ent = sorted(tuple(e) for e in k["entries"])
    indptr = [0] * (rows + 1)
    for r, _c, _v in ent:
        indptr[r + 1] += 1
    for r in range(rows):
        indptr[r + 1] += indptr[r]
    return (np.array(indptr, np.uint64), np.array([c for _r, c, _v in ent], np.uint64), vec([v % p for _r, _c, v in ent]), cols, vec(k["z"]), vec(k["z2"]),
            [x % p for x in k["out_int"]], [x % p for x in k["out2_int"]])


def check_kats3(mle_multi_evaluate, spmv, spmv_pair):
    """round 6's additions to the golden file: mle_multi_evaluate(fid, [z_vec...], r_vec) -> list of 32-byte values;
    spmv(fid, indptr, indices, data_vec, cols, z_vec) -> bytes; spmv_pair(fid, indptr, indices, data_vec, cols, z1, z2) -> (bytes, bytes)"""
    for fid in FIELDS:
        for case in KATS["mle_multi_evaluate_known_values"]["cases"]:
            got = mle_multi_evaluate(fid, [vec(z) for z in case["zs"]], vec(case["point"]))
            assert [int.from_bytes(bytes(g), "little") for g in got] == case["evals"]
        ip, ix, dt, cols, z, z2, out, out2 = mixed_coefficient_case(fid)
        assert ints(np.frombuffer(to_bytes(spmv(fid, ip, ix, dt, cols, z)), np.uint8)) == out
        assert ints(np.frombuffer(to_bytes(spmv(fid, ip, ix, dt, cols, z2)), np.uint8)) == out2
        a, b = spmv_pair(fid, ip, ix, dt, cols, z, z2)
        assert ints(np.frombuffer(to_bytes(a), np.uint8)) == out and ints(np.frombuffer(to_bytes(b), np.uint8)) == out2
