"""CPU: the oracle's restatement of Spartan's sum-check provers (oracle/nova_ref.c, round 5) against the reference's verifier
and the definition of every round polynomial (tests/spartan_common.py), and its compute_eval_table_sparse inner against the
reference's sparse-matrix known answer read transposed."""
import numpy as np
import pytest

from oracle import cref
from tests import fv_common as fc
from tests import spartan_common as sp


def o_cubic3(fid, claim, taus, A, B, C, tr):
    return cref.sumcheck_prove_cubic3(fid, claim, taus, A, B, C, cref.make_transcript(tr))


def o_quad(fid, claim, nr, A, B, tr):
    return cref.sumcheck_prove_quad_prod(fid, claim, nr, A, B, cref.make_transcript(tr))


def o_batch(fid, claims, nrs, polys, pts, coeffs, tr):
    return cref.sumcheck_prove_batch_eval(fid, claims, nrs, [p.tobytes() for p in polys], [x.tobytes() for x in pts], coeffs,
                                          cref.make_transcript(tr))


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
@pytest.mark.parametrize("l", [1, 2, 3, 4, 5, 8])
def test_cubic_with_three_inputs_against_the_verifier_and_the_definition(fid, l):
    sp.check_cubic3(o_cubic3, fid, l, seed=300 + l)


@pytest.mark.parametrize("l", [2, 3, 4, 5, 6])
def test_cubic_fallback_when_tau_is_zero(l):
    """tau_i = 0 makes l(1) = 0: derive_from_claim_deg2 returns None and the third N-scaling sum runs
    (sumcheck.rs:695-697, 1085-1136); also a challenge that zeroes eval_eq_left for every later round (r = 1 after tau = 0:
    1 - tau - r + 2 r tau = 0)."""
    p = fc.FIELDS[1]
    base = fc.ints(fc.rand_vec(1, l, 55))
    for zero_at in range(l):
        taus = list(base)
        taus[zero_at] = 0
        sp.check_cubic3(o_cubic3, 1, l, seed=400 + zero_at, taus=taus)
        sp.check_cubic3(o_cubic3, 1, l, seed=500 + zero_at, taus=taus, force={zero_at: 1})
    sp.check_cubic3(o_cubic3, 1, l, seed=77, taus=[0] * l)
    sp.check_cubic3(o_cubic3, 1, l, seed=78, taus=[1] * l, force={0: 0, l - 1: p - 1})


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
@pytest.mark.parametrize("l", [1, 2, 5, 9])
def test_quad_prod(fid, l):
    sp.check_quad_prod(o_quad, fid, l, seed=600 + l)
    sp.check_quad_prod(o_quad, fid, l, seed=700 + l, force={0: 0, l - 1: 1})


@pytest.mark.parametrize("fid", [1, 3])
@pytest.mark.parametrize("nrs", [[4], [5, 5], [3, 6], [6, 3], [7, 2, 5], [1, 4]])
def test_batch_eval_with_polynomials_of_different_sizes(fid, nrs):
    sp.check_batch_eval(o_batch, fid, nrs, seed=800 + sum(nrs))


def test_batch_eval_fallback_when_an_evaluation_point_has_a_zero():
    p = fc.FIELDS[1]
    sp.check_batch_eval(o_batch, 1, [4, 6], seed=900, force={0: 0, 3: 1, 5: p - 1})
    # evaluation points that really have zero coordinates (l(1) = 0 in those rounds: sumcheck.rs:1085-1136), in one claim only / in all
    sp.check_batch_eval(o_batch, 1, [6, 5, 7], seed=901, zero_coords={1: [0]})
    sp.check_batch_eval(o_batch, 1, [6, 5], seed=902, zero_coords={0: [2, 3], 1: [4]})
    sp.check_batch_eval(o_batch, 3, [5, 5], seed=903, zero_coords={1: list(range(5))})


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_transposed_product(fid):
    k = sp.transposed_kat()
    got = cref.spmv_transposed(fid, k["indptr"], k["indices"], fc.vec(k["data"]), k["rows"], k["cols"], fc.vec(k["x"]))
    assert sp.ints(got) == k["out"]
    p = fc.FIELDS[fid]
    for rows, cols, seed in ((50, 30, 1), (300, 64, 2), (64, 300, 3)):
        ip, ix, dt = sp.heavy_column_csr(fid, rows, cols, seed, heavy_cols=(0, cols - 1))
        x = fc.edge_vectors(fid, rows, seed + 10)
        got = cref.spmv_transposed(fid, ip, ix, dt, rows, cols, x)
        assert sp.ints(got) == sp.dense_transposed(p, ip, ix, dt, cols, x)
    # eval tables as snark.rs:181-190 uses them: M^T eq(r_x) == sum_row eq(r_x)[row] * M[row, :] and the identity the verifier
    # relies on (snark.rs:325-338): sum_col (M^T eq(r_x))[col] * eq(r_y)[col] == sum_{row, col} eq_x[row] eq_y[col] M[row, col]
    rows, cols = 32, 16
    ip, ix, dt = fc.random_csr(fid, rows, cols, 9)
    rx, ry = fc.rand_vec(fid, 5, 20), fc.rand_vec(fid, 4, 21)
    ex = np.frombuffer(cref.eq_evals(fid, rx, 5), np.uint8).reshape(-1, 32)
    ey = sp.ints(cref.eq_evals(fid, ry, 4))
    t = sp.ints(cref.spmv_transposed(fid, ip, ix, dt, rows, cols, ex))
    d, exi = sp.ints(dt), sp.ints(ex)
    exp = sum(exi[r] * ey[int(ix[k])] * d[k] for r in range(rows) for k in range(int(ip[r]), int(ip[r + 1]))) % p
    assert sum(a * b for a, b in zip(t, ey)) % p == exp


@pytest.mark.parametrize("ell", [3, 6, 10])
def test_the_replayed_prove_sequence_verifies_on_the_oracle(ell):
    """bench.py's spartan_sequence (snark.rs:133-233 as provider calls) through the oracle alone: on a satisfied relaxed instance
    (E = Az o Bz - u Cz) the three sum-check proofs pass the reference's verifier equations (bench.spartan_verify)."""
    import bench
    fid = 1
    p = fc.FIELDS[fid]
    n = 1 << ell
    csr, hW, u, hz = bench.spartan_instance(fid, ell)
    az, bz, cz = (cref.spmv(fid, *m, n, hz) for m in csr)
    hE = np.frombuffer(cref.field_cross_term(fid, az, bz, cz, np.zeros((n, 32), np.uint8), u, n), np.uint8).reshape(n, 32)
    res = bench.spartan_sequence(bench.SpartanCpu(fid, csr, n, hW, hE, hz), ell, p, u)
    assert all(bench.spartan_verify(p, u, res).values())
    # an unsatisfied instance must NOT verify (the check has teeth)
    bad = hE.copy()
    bad[0, 0] ^= 1
    res = bench.spartan_sequence(bench.SpartanCpu(fid, csr, n, hW, bad, hz), ell, p, u)
    assert not bench.spartan_verify(p, u, res)["proof_verifies_outer"]


def test_compressed_snark_sequence_through_the_oracle_alone():
    """bench.py's compressed_snark_sequence (CompressedSNARK::prove, src/nova/mod.rs:793-881, as provider calls) with the oracle on
    both sides: the relaxed fold of a satisfied running instance with a sampled random instance (src/r1cs/mod.rs:786-830, 629-661,
    1070-1107) is satisfied again -- both Spartan proofs over the FOLDED instances pass the reference's verifier equations -- and the
    evaluation argument's openings are consistent: B(u_j) recomputed from the v matrix equals the quotient identity's value."""
    import bench
    from oracle import pyref as R
    sides = {"P": bench.CsnarkSide(0, 6, seed=906), "S": bench.CsnarkSide(1, 5, seed=955)}
    keys = {k: cref.sequential_bases([R.BN254_G1, R.GRUMPKIN][sd.cid], 7, sd.n + 1) for k, sd in sides.items()}
    cpu = {k: bench.CpuProvider(sides[k], keys[k][:sides[k].n], keys[k][sides[k].n].tobytes()) for k in sides}
    res = bench.compressed_snark_sequence(cpu["P"], cpu["S"], sides["P"], sides["S"])
    for tag in ("P", "S"):
        assert all(bench.spartan_verify(sides[tag].p, res[f"fold_{tag}"]["u"], res[f"spartan_{tag}"]).values()), tag
    # every commitment of the sequence was logged for the trait-only form: per side 3 in the fold; the primary adds batch_commit + openings
    assert [k for k, _v, _e in cpu["S"].msm_log] == ["commit"] * 3
    assert [k for k, _v, _e in cpu["P"].msm_log] == ["commit"] * 3 + ["batch", "batch"]
    # HyperKZG's v matrix obeys the verifier's recurrence between consecutive folded polynomials (hyperkzg.rs:1180-1215): with
    # u = [r, -r, r^2] and x = point[ell - i - 1],  r P_{i+1}(r^2) = r (1 - x) P_even(r^2) + x r P_odd(r^2),  where
    # P_even(r^2) = (v[i][0] + v[i][1]) / 2 and r P_odd(r^2) = (v[i][0] - v[i][1]) / 2.  r itself is not part of the output; the
    # last folded polynomial is linear, f(X) = c0 + c1 X, so r = (f(r^2) - c0) / (c1 r).
    p = sides["P"].p
    num = lambda b: int.from_bytes(bytes(b), "little")
    v = [[num(c) for c in row] for row in res["ee_P"]["v"]]
    x = [num(c) for c in res["spartan_P"]["batch"][1]]
    ell = sides["P"].ell
    assert len(v) == ell and len(res["ee_P"]["com"]) == ell - 1 and len(res["ee_P"]["w"]) == 3
    two_inv = pow(2, -1, p)
    c0 = (v[ell - 1][0] + v[ell - 1][1]) * two_inv % p
    c1r = (v[ell - 1][0] - v[ell - 1][1]) * two_inv % p
    r = (v[ell - 1][2] - c0) * pow(c1r, -1, p) % p
    for i in range(ell - 1):
        xi = x[ell - i - 1]
        even = (v[i][0] + v[i][1]) * two_inv % p
        odd_r = (v[i][0] - v[i][1]) * two_inv % p
        assert r * v[i + 1][2] % p == (r * (1 - xi) % p * even + xi * odd_r) % p, i
    # the secondary's evaluation argument: the IPA proof over the batched witness passes the reference's verifier (ipa_pc.rs:286-390)
    # with b = eq(point) and the challenges replayed from the same stand-in transcript state
    from tests import ipa_common as ic
    from tests import standin
    S, ee = sides["S"], res["ee_S"]
    curve = R.GRUMPKIN
    a = cpu["S"].host(res["spartan_S"]["batch_witness"])
    point = b"".join(bytes(c) for c in res["spartan_S"]["batch"][1])
    b = cref.eq_evals(S.fid, point, S.ell)
    assert len(ee["L"]) == len(ee["R"]) == S.ell
    # the round challenges, as the verifier gets them: its transcript, in the state the prover's was in, absorbs every (L, R)
    import ctypes
    tr = standin.Transcript(seed=1)
    ctypes.memmove(tr.state, ee["tr_state"], 48)
    rs = []
    for L, Rr, (Li, Ri) in zip(ee["L"], ee["R"], ee["inf"]):
        out = (ctypes.c_uint8 * 32)()
        standin.lib().standin_ipa_transcript(tr.ctx, (ctypes.c_uint8 * 64)(*L), int(Li), (ctypes.c_uint8 * 64)(*Rr), int(Ri), out)
        rs.append(int.from_bytes(bytes(out), "little"))
    ckc = np.frombuffer(ee["ck_c"], np.uint8).copy()
    key = np.ascontiguousarray(keys["S"][:S.n])
    av, bv = np.frombuffer(a, np.uint8).reshape(-1, 32).copy(), np.frombuffer(b, np.uint8).reshape(-1, 32).copy()
    assert ic.verify(curve, key, ckc, av, bv, S.n, ee["L"], ee["R"], ee["inf"], ee["a_hat"], rs)
    bad = ic.le((int.from_bytes(ee["a_hat"], "little") + 1) % curve.r)
    assert not ic.verify(curve, key, ckc, av, bv, S.n, ee["L"], ee["R"], ee["inf"], bad, rs)
