"""-m gpu: the field-vector kernels at the sizes their roofline numbers are quoted on (2^22 full compare; 2^24 full
compare for the streaming kernels, sums for the reductions) -- first round tested them to 2^18 only.  BN254 Fr, the
scalar field of the headline curve; HBM-resident operands (the configuration bench.py times)."""
import numpy as np
import pytest

from oracle import cref
from tests import fv_common as C
from tests import util

pytestmark = pytest.mark.gpu
FID = 1


def big_vec(log2n, seed):
    """2^log2n elements: a 2^20 random block tiled, each tile's low byte XOR-ed with the tile index so no two tiles are
    equal (values stay below the modulus: only the lowest byte changes)."""
    n = 1 << log2n
    blk = C.rand_vec(FID, min(n, 1 << 20), seed)
    if n <= len(blk):
        return blk
    tiles = n // len(blk)
    out = np.tile(blk, (tiles, 1))
    out[:, 0] ^= np.repeat(np.arange(tiles, dtype=np.uint8), len(blk))
    return out


@pytest.mark.parametrize("log2n", [22, 24])
def test_streaming_kernels_full_compare(nmx, log2n):
    import torch
    from nova_amd import fieldvec as fv
    n = 1 << log2n
    a, b = big_vec(log2n, 1), big_vec(log2n, 2)
    r = C.rand_vec(FID, 1, 9)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert fv.axpy(FID, da, db, r).cpu().numpy().tobytes() == cref.field_axpy(FID, a, b, r, n)
    assert fv.bind_poly_var_top(FID, da, r).cpu().numpy().tobytes() == cref.field_bind(FID, a, 0, n // 2, 1, r, n // 2)
    assert fv.fold_pairs(FID, da, r).cpu().numpy().tobytes() == cref.field_bind(FID, a, 0, 1, 2, r, n // 2)
    got = fv.suffix_horner(FID, da, r)
    exp = cref.suffix_horner(FID, a, n, r)
    assert got.cpu().numpy().tobytes() == exp
    if log2n == 22:
        c, e = big_vec(log2n, 3), big_vec(log2n, 4)
        dc, de = torch.from_numpy(c).cuda(), torch.from_numpy(e).cuda()
        assert fv.cross_term(FID, da, db, dc, de, r).cpu().numpy().tobytes() == cref.field_cross_term(FID, a, b, c, e, r, n)
        assert fv.axpy2(FID, da, db, dc, r).cpu().numpy().tobytes() == cref.field_axpy2(FID, a, b, c, r, n)
        polys = [da, db, dc, de, da[: n // 2], db[: n // 4]]
        hp = [a, b, c, e, a[: n // 2], b[: n // 4]]
        got = fv.lincomb_powers(FID, polys, r)
        assert got.cpu().numpy().tobytes() == cref.lincomb_powers(FID, [h.tobytes() for h in hp], r, n)


@pytest.mark.parametrize("log2n", [22, 24])
def test_reduction_kernels(nmx, log2n):
    import torch
    from nova_amd import fieldvec as fv
    n = 1 << log2n
    A, B, Cc = (big_vec(log2n, s) for s in (11, 12, 13))
    dA, dB, dC = (torch.from_numpy(x).cuda() for x in (A, B, Cc))
    shift = (log2n - 1) // 2
    eqR = C.rand_vec(FID, 1 << shift, 5)
    eqL = C.rand_vec(FID, (n // 2) >> shift, 6)
    dR, dL = torch.from_numpy(eqR).cuda(), torch.from_numpy(eqL).cuda()
    for mode in (1, 2, 3):
        got = fv.sumcheck_eq_sums(FID, mode, dA, dB if mode >= 2 else None, dC if mode >= 3 else None, dR, dL, shift)
        exp = cref.sumcheck_eq_sums(FID, mode, A, B if mode >= 2 else None, Cc if mode >= 3 else None, n, eqR, eqL, shift)
        assert tuple(got) == tuple(exp), mode
    for kind in (1, 2, 3, 4):
        got = fv.sumcheck_plain_sums(FID, kind, dA, dB, dC if kind == 4 else None)
        exp = cref.sumcheck_plain_sums(FID, kind, A, B, Cc if kind == 4 else None, n)
        assert tuple(got) == tuple(exp if kind == 4 else exp[:2]), kind
    point = C.rand_vec(FID, log2n, 8)
    assert fv.mle_evaluate(FID, dA, point) == cref.mle_evaluate(FID, A, log2n, point)
    # one fused round (bind + next round's sums) against bind-then-sum through the oracle
    r = C.rand_vec(FID, 1, 9)
    shift2 = (log2n - 2) // 2
    eqR2 = C.rand_vec(FID, 1 << shift2, 15)
    eqL2 = C.rand_vec(FID, (n // 4) >> shift2, 16)
    wA, wB, wC = dA.clone(), dB.clone(), dC.clone()
    res = fv.sumcheck_bind_eq_sums(FID, 3, wA, wB, wC, r, torch.from_numpy(eqR2).cuda(), torch.from_numpy(eqL2).cuda(), shift2)
    bound = [cref.field_bind(FID, h, 0, n // 2, 1, r, n // 2) for h in (A, B, Cc)]
    exp = cref.sumcheck_eq_sums(FID, 3, bound[0], bound[1], bound[2], n // 2, eqR2, eqL2, shift2)
    assert tuple(res[3]) == tuple(exp)
    assert res[0].cpu().numpy().tobytes()[: 32 * (n // 2)] == bound[0]


def test_spmv_2p22_rows(nmx):
    import torch
    from nova_amd import fieldvec as fv
    n = 1 << 22
    rng = np.random.Generator(np.random.PCG64(5))
    indptr = np.arange(0, 3 * n + 1, 3, dtype=np.uint64)
    indices = rng.integers(0, n, size=3 * n).astype(np.uint64)
    data = big_vec(22, 4)
    data = np.concatenate([data, data, data]).copy()        # 3 n coefficients
    assert len(data) == 3 * n
    data[::5] = util.int_to_le32(1)                        # +1, -1 and small coefficients as R1CS matrices have
    data[1::7] = util.int_to_le32(C.FIELDS[FID] - 1)
    data[2::11] = util.int_to_le32(3)
    z = big_vec(22, 6)
    z2 = big_vec(22, 7)
    mat = fv.SparseMatrix(FID, indptr, indices, data, n)
    dz, dz2 = torch.from_numpy(z).cuda(), torch.from_numpy(z2).cuda()
    assert mat.multiply_vec(dz).cpu().numpy().tobytes() == cref.spmv(FID, indptr, indices, data, n, z)
    o1, o2 = mat.multiply_vec_pair(dz, dz2)
    e1, e2 = cref.spmv_pair(FID, indptr, indices, data, n, z, z2)
    assert o1.cpu().numpy().tobytes() == e1 and o2.cpu().numpy().tobytes() == e2
    mat.close()


@pytest.mark.parametrize("log2n", [22, 24])
def test_batch_invert_full_size(nmx, log2n):
    """batch_invert (src/spartan/mod.rs:54-152) at 2^22 (full compare with the oracle: 16- and 8-element chunk levels) and 2^24 (32-element
    first level): the involution inv(inv(v)) == v bit for bit, and every product v[i] * inv[i] == 1 through the cross-term kernel
    (a o b - u c with b = the inverses, u = 1, c = the all-ones vector, e = 0 must be the zero vector)."""
    import torch
    from nova_amd import fieldvec as fv
    n = (1 << log2n) - (3 if log2n == 22 else 0)            # a ragged last chunk at every level for 2^22 - 3
    v = big_vec(log2n, 5)[:n]
    v[v.reshape(n, 32).any(axis=1) == 0] = util.int_to_le32(7)   # (no zero element)
    dv = torch.from_numpy(np.ascontiguousarray(v)).cuda()
    inv = fv.batch_invert(FID, dv)
    back = fv.batch_invert(FID, inv)
    assert torch.equal(back, dv)
    ones = torch.from_numpy(np.tile(np.frombuffer(util.int_to_le32(1), np.uint8), (n, 1))).cuda()
    zero = torch.zeros_like(dv)
    t = fv.cross_term(FID, dv, inv, ones, zero, util.int_to_le32(1))
    assert not bool(t.any())
    if log2n == 22:
        assert inv.cpu().numpy().tobytes() == cref.batch_invert(FID, np.ascontiguousarray(v), n)
