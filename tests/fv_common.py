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
