"""-m gpu: Spartan's sum-check provers as single C calls (nmx_sumcheck_prove_*, nova_amd/csrc/sumcheck_prove.hpp) and
compute_eval_table_sparse's transposed product (nmx_spmv_apply_transposed) against (1) the reference's verifier and the definition
of every round polynomial (tests/spartan_common.py), (2) the oracle's restatement round by round -- same stand-in transcript, so the
challenge sequences coincide iff every round polynomial does -- and (3) the replay of `RelaxedR1CSSNARK::prove`
(/root/reference/src/spartan/snark.rs:113-260) as bench.py runs it, at a size the oracle finishes in seconds."""
import numpy as np
import pytest

from oracle import cref
from tests import fv_common as fc
from tests import spartan_common as sp

pytestmark = pytest.mark.gpu


def dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x).copy()).cuda()


def g_cubic3(mont=False):
    def prove(fid, claim, taus, A, B, C, tr):
        from nova_amd import fieldvec as fv
        return fv.sumcheck_prove_cubic_with_three_inputs(fid, claim, taus, dev(A), dev(B), dev(C), tr, mont=mont)
    return prove


def g_quad(fid, claim, nr, A, B, tr):
    from nova_amd import fieldvec as fv
    return fv.sumcheck_prove_quad_prod(fid, claim, nr, dev(A), dev(B), tr)


def g_batch(fid, claims, nrs, polys, pts, coeffs, tr):
    from nova_amd import fieldvec as fv
    return fv.sumcheck_prove_batch_eval(fid, claims, nrs, [dev(p) for p in polys], pts, coeffs, tr)


def both(check, g_prove, o_prove, *args, **kw):
    """the same instance through the HIP path and the oracle: identical round polynomials, challenges and final claims"""
    got = check(g_prove, *args, **kw)
    exp = check(o_prove, *args, **kw)
    assert got == exp
    return got


def o_cubic3(fid, claim, taus, A, B, C, tr):
    return cref.sumcheck_prove_cubic3(fid, claim, taus, A, B, C, cref.make_transcript(tr))


def o_quad(fid, claim, nr, A, B, tr):
    return cref.sumcheck_prove_quad_prod(fid, claim, nr, A, B, cref.make_transcript(tr))


def o_batch(fid, claims, nrs, polys, pts, coeffs, tr):
    return cref.sumcheck_prove_batch_eval(fid, claims, nrs, [p.tobytes() for p in polys], [x.tobytes() for x in pts], coeffs,
                                          cref.make_transcript(tr))


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
@pytest.mark.parametrize("l", [1, 2, 3, 5, 8])
def test_cubic_with_three_inputs_small(nmx, fid, l):
    both(sp.check_cubic3, g_cubic3(), o_cubic3, fid, l, seed=300 + l)


def test_host_tables_are_uploaded_for_the_call(nmx):
    """Without NMX_SCALARS_DEVICE the tables are host arrays, as the reference's `&mut MultilinearPolynomial` are: same proofs."""
    from nova_amd import fieldvec as fv
    both(sp.check_cubic3, lambda f, c, t, A, B, C, tr: fv.sumcheck_prove_cubic_with_three_inputs(f, c, t, A.copy(), B.copy(), C.copy(), tr), o_cubic3,
         1, 11, seed=17, brute=False)
    both(sp.check_batch_eval, lambda f, cl, nrs, P, X, co, tr: fv.sumcheck_prove_batch_eval(f, cl, nrs, [p.copy() for p in P], X, co, tr), o_batch,
         1, [9, 12], seed=18)


@pytest.mark.parametrize("l", [10, 11, 12, 13, 16])
def test_cubic_with_three_inputs_across_the_kernel_forms(nmx, l):
    """l = 10: every round in one block; 11-13: the first rounds as pass + final sum (k_eq_rows with both eq tables, then the
    last-half form), the block form below 512 indices; 16: several blocks per pass."""
    both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, l, seed=40 + l, brute=False)


@pytest.mark.parametrize("l", [2, 3, 5, 6, 11])
def test_cubic_fallback_when_tau_is_zero(nmx, l):
    """tau_i = 0 (l(1) = 0) and a challenge that zeroes eq's running product: the reference's third N-scaling sum
    (sumcheck.rs:1085-1136); here t(1) from a pass over the swapped halves."""
    base = fc.ints(fc.rand_vec(1, l, 55))
    for zero_at in sorted({0, l // 2, l - 1}):
        taus = list(base)
        taus[zero_at] = 0
        both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, l, seed=400 + zero_at, taus=taus, brute=l <= 5)
        both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, l, seed=500 + zero_at, taus=taus, force={zero_at: 1}, brute=l <= 5)
    both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, l, seed=77, taus=[0] * l, brute=l <= 5)


@pytest.mark.parametrize("fid", [1, 3])
def test_cubic_montgomery_layout(nmx, fid):
    """NMX_SCALARS_MONT: tables, taus, claim, round polynomials and challenges all as R = 2^256 Montgomery limbs (what the Rust shim
    passes zero-copy); converted in and out here, the proof must be the canonical one."""
    p = fc.FIELDS[fid]
    Rm = 1 << 256
    to_m = lambda v: fc.vec([x * Rm % p for x in fc.ints(v)])
    un_m = lambda b: int(int.from_bytes(b, "little") * pow(Rm, -1, p) % p).to_bytes(32, "little")

    def prove_m(fid_, claim, taus, A, B, C, tr):
        from nova_amd import fieldvec as fv

        def tr_m(coeffs):                               # the stand-in transcript sees canonical coefficients
            ch = tr([un_m(c) for c in coeffs])
            return int(int.from_bytes(ch, "little") * Rm % p).to_bytes(32, "little")
        polys, rs, claims = fv.sumcheck_prove_cubic_with_three_inputs(fid_, to_m(np.frombuffer(claim, np.uint8)), to_m(taus), dev(to_m(A)),
                                                                      dev(to_m(B)), dev(to_m(C)), tr_m, mont=True)
        return [[un_m(c) for c in row] for row in polys], [un_m(r) for r in rs], [un_m(c) for c in claims]
    for l in (4, 12):
        both(sp.check_cubic3, prove_m, o_cubic3, fid, l, seed=900 + l, brute=False)


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
@pytest.mark.parametrize("l", [1, 2, 5, 9, 12, 15])
def test_quad_prod(nmx, fid, l):
    both(sp.check_quad_prod, g_quad, o_quad, fid, l, seed=600 + l)
    if l <= 9:
        both(sp.check_quad_prod, g_quad, o_quad, fid, l, seed=700 + l, force={0: 0, l - 1: 1})


@pytest.mark.parametrize("fid", [1, 3])
@pytest.mark.parametrize("nrs", [[4], [5, 5], [3, 6], [6, 3], [7, 2, 5], [1, 4], [13, 12], [12, 14]])
def test_batch_eval_with_polynomials_of_different_sizes(nmx, fid, nrs):
    both(sp.check_batch_eval, g_batch, o_batch, fid, nrs, seed=800 + sum(nrs))


def test_batch_eval_fallback(nmx):
    p = fc.FIELDS[1]
    both(sp.check_batch_eval, g_batch, o_batch, 1, [4, 6], seed=900, force={0: 0, 3: 1, 5: p - 1})


def test_a_failing_transcript_aborts_the_proof(nmx):
    from nova_amd import fieldvec as fv
    import nova_amd
    A, B = fc.rand_vec(1, 16, 1), fc.rand_vec(1, 16, 2)

    def bad(_coeffs):
        raise RuntimeError("transcript refused")
    with pytest.raises(nova_amd.NmxError):
        fv.sumcheck_prove_quad_prod(1, sp.le(5), 4, dev(A), dev(B), bad)
    with pytest.raises(nova_amd.NmxError):               # a challenge >= p: from_repr would reject it
        fv.sumcheck_prove_quad_prod(1, sp.le(5), 4, dev(A), dev(B), lambda c: b"\xff" * 32)
    # host tables are accepted too (uploaded for the call, the host copies left as they were)
    keepA = A.copy()
    got = fv.sumcheck_prove_quad_prod(1, sp.le(5), 4, A, B, lambda c: sp.le(7))
    assert got == fv.sumcheck_prove_quad_prod(1, sp.le(5), 4, dev(A), dev(B), lambda c: sp.le(7)) and np.array_equal(A, keepA)
    # and the library still works afterwards
    both(sp.check_quad_prod, g_quad, o_quad, 1, 4, seed=1)


def test_polling_and_synchronising_agree(nmx):
    from nova_amd import _lib
    L = _lib.lib()
    try:
        for us in (0, 1, 2000):
            assert L.nmx_set_option(b"sc_poll_us", us) == 0
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 12, seed=5, brute=False)
    finally:
        assert L.nmx_set_option(b"sc_poll_us", 2000) == 0


def test_one_launch_rounds_and_two_launch_rounds_agree(nmx):
    """option sc_fused_sum: the pass whose last block sums the partials (agent-scope tickets) against pass + final-sum launch, at
    sizes where a round runs on 4 .. 4096 blocks, all three provers, repeated (the ticket word must come back to zero)."""
    from nova_amd import _lib
    L = _lib.lib()
    try:
        for fused in (1, 0, 1):
            assert L.nmx_set_option(b"sc_fused_sum", fused) == 0
            for l in (11, 14, 17):
                both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, l, seed=60 + l, brute=False)
            both(sp.check_quad_prod, g_quad, o_quad, 1, 16, seed=61)
            both(sp.check_batch_eval, g_batch, o_batch, 1, [15, 13, 16], seed=62)
    finally:
        assert L.nmx_set_option(b"sc_fused_sum", 1) == 0


def test_four_lanes_per_index_and_one_lane_per_index_agree(nmx):
    """option sc_quad: passes of <= 2^13 indices of the cubic and quad_prod provers spread an index over four lanes (binds side by side,
    values exchanged in shuffles).  Sizes on both sides of every switch (one block up to 64 indices, 64 indices per block up to 2^13,
    the one-lane form beyond), the no-bind first round of small instances (host tail lowered), the fallback rounds, both layouts via the
    Montgomery test below; every proof equal to the oracle's with the option on and off."""
    from nova_amd import _lib
    L = _lib.lib()
    p = fc.FIELDS[1]
    try:
        for quad in (1, 0, 1):
            assert L.nmx_set_option(b"sc_quad", quad) == 0
            for l in (8, 9, 10, 14, 15, 16):
                both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, l, seed=40 + l, brute=False)
                both(sp.check_quad_prod, g_quad, o_quad, 1, l, seed=50 + l)
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 3, 11, seed=41, brute=False)
            both(sp.check_quad_prod, g_quad, o_quad, 0, 11, seed=42)
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 12, seed=43, force={0: 0, 3: 1, 7: p - 1, 9: 0}, brute=False)
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 10, seed=44, taus=[0] * 10, brute=False)
            assert L.nmx_set_option(b"sc_host_tail", 2) == 0
            for l in (3, 5, 6, 7, 8):
                both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, l, seed=45 + l, brute=False)
                both(sp.check_quad_prod, g_quad, o_quad, 1, l, seed=55 + l)
            assert L.nmx_set_option(b"sc_host_tail", 7) == 0
    finally:
        assert L.nmx_set_option(b"sc_quad", 1) == 0
        assert L.nmx_set_option(b"sc_host_tail", 7) == 0


def test_prelaunched_passes_and_launched_passes_agree(nmx):
    """option sc_prelaunch: a small pass of the cubic / quad_prod provers is enqueued a round early and takes its challenge from a line
    of pinned memory (default), or is launched once the challenge is known.  Sizes around the limits (pre-launch from 2^14 indices per
    pass down, the four-lane form from 2^12, one block from 64, the host tail at 128 elements), the fallback rounds (never pre-launched
    over), host-added and ticket partials -- and a transcript that fails while a pre-launched pass is waiting: the pass is cancelled,
    the call reports the failure, the next proofs are right."""
    import nova_amd
    from nova_amd import _lib, fieldvec as fv
    L = _lib.lib()
    p = fc.FIELDS[1]
    try:
        for n_pass, pre in enumerate((1, 0, 1)):
            assert L.nmx_set_option(b"sc_prelaunch", pre) == 0
            for l in ((8, 9, 10, 13, 16) if n_pass < 2 else (13,)):
                both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, l, seed=80 + l, brute=False)
                both(sp.check_quad_prod, g_quad, o_quad, 1, l + 1, seed=90 + l)
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 12, seed=82, force={0: 0, 3: 1, 7: p - 1, 9: 0}, brute=False)
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 11, seed=83, taus=[0] * 11, brute=False)
            both(sp.check_batch_eval, g_batch, o_batch, 1, [14, 11, 16, 9], seed=87)
            both(sp.check_batch_eval, g_batch, o_batch, 1, [10, 12, 8], seed=88, force={0: 0, 3: 1, 7: p - 1, 11: 0})
            assert L.nmx_set_option(b"sc_side_streams", 0) == 0
            both(sp.check_batch_eval, g_batch, o_batch, 1, [13, 15], seed=89)
            assert L.nmx_set_option(b"sc_side_streams", 1) == 0
            assert L.nmx_set_option(b"sc_host_parts", 0) == 0
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 14, seed=84, brute=False)
            assert L.nmx_set_option(b"sc_host_parts", 1) == 0
        # a transcript that gives up in round 4 of a 2^13 instance: the next pass is waiting for its challenge by then
        A, B = fc.rand_vec(1, 1 << 13, 1), fc.rand_vec(1, 1 << 13, 2)
        calls = []

        def gives_up(_coeffs):
            calls.append(1)
            if len(calls) == 4:
                raise RuntimeError("transcript refused")
            return sp.le(7 + len(calls))
        with pytest.raises(nova_amd.NmxError):
            fv.sumcheck_prove_quad_prod(1, sp.le(5), 13, dev(A), dev(B), gives_up)
        assert len(calls) == 4
        both(sp.check_quad_prod, g_quad, o_quad, 1, 13, seed=85)
        both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 13, seed=86, brute=False)
    finally:
        assert L.nmx_set_option(b"sc_prelaunch", 1) == 0
        assert L.nmx_set_option(b"sc_host_parts", 1) == 0
        assert L.nmx_set_option(b"sc_side_streams", 1) == 0


def test_resident_rounds_and_a_pass_per_round_agree(nmx):
    """option sc_resident: from tables of <= 2^14 elements every remaining device round of a prover runs inside ONE resident kernel
    (k_sc_resident: wait for the challenge on the device, bind, next sums, partials to the host, wait again; the hand-over at the end)
    instead of one pass per round.  Sizes on both sides of the entry (2^14 elements), instances that are resident from their first
    bind, every host-tail threshold (the kernel ends with the hand-over: 1 .. 256 elements), all four fields, batch claims of mixed
    sizes (each its own resident kernel; more claims than the budget of waiting blocks allows), forced challenges that send a round
    into the tau = 0 fall-back while a resident kernel is waiting (it is cancelled, the proof goes on on launched passes), a
    transcript that fails mid-way, the torn-line injection -- every proof equal to the oracle's, with the option on and off."""
    import nova_amd
    from nova_amd import _lib, fieldvec as fv
    L = _lib.lib()
    p = fc.FIELDS[1]
    try:
        for res in (1, 0, 1):
            assert L.nmx_set_option(b"sc_resident", res) == 0
            for l in (9, 10, 13, 14, 15, 17):
                both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, l, seed=240 + l, brute=False)
                both(sp.check_quad_prod, g_quad, o_quad, 1, l + 1, seed=250 + l)
            for fid in (0, 2, 3):
                both(sp.check_cubic3, g_cubic3(), o_cubic3, fid, 12, seed=260 + fid, brute=False)
                both(sp.check_quad_prod, g_quad, o_quad, fid, 13, seed=265 + fid)
            both(sp.check_batch_eval, g_batch, o_batch, 1, [14, 11, 16, 9], seed=270)
            both(sp.check_batch_eval, g_batch, o_batch, 3, [13, 13], seed=271)
            both(sp.check_batch_eval, g_batch, o_batch, 1, [12, 9, 14, 10, 13, 11, 12, 10, 9, 14, 13, 12], seed=272)   # 12 claims: the budget runs out
            # fall-back rounds while resident kernels wait: forced challenges (cubic), zero coordinates (one claim of a batch)
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 13, seed=273, force={0: 0, 3: 1, 7: p - 1, 9: 0}, brute=False)
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 12, seed=274, taus=[0] * 12, brute=False)
            base = fc.ints(fc.rand_vec(1, 14, 56))
            base[9] = 0
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 14, seed=275, taus=base, brute=False)
            both(sp.check_batch_eval, g_batch, o_batch, 1, [14, 12, 13], seed=276, zero_coords={1: [6], 2: [10]})
            both(sp.check_batch_eval, g_batch, o_batch, 1, [10, 12, 8], seed=277, force={0: 0, 3: 1, 7: p - 1, 11: 0})
            for tail in (0, 1, 4, 8):
                assert L.nmx_set_option(b"sc_host_tail", tail) == 0
                both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 11, seed=280 + tail, brute=False)
                both(sp.check_quad_prod, g_quad, o_quad, 1, 10, seed=285 + tail)
                both(sp.check_batch_eval, g_batch, o_batch, 1, [11, 6, 9], seed=290 + tail)
            assert L.nmx_set_option(b"sc_host_tail", 7) == 0
            assert L.nmx_set_option(b"sc_torn_test", 25) == 0
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 14, seed=295, brute=False)
            both(sp.check_batch_eval, g_batch, o_batch, 1, [13, 12], seed=296)
            assert L.nmx_set_option(b"sc_torn_test", 0) == 0
            assert L.nmx_set_option(b"sc_side_streams", 0) == 0
            both(sp.check_batch_eval, g_batch, o_batch, 1, [13, 14], seed=297)
            assert L.nmx_set_option(b"sc_side_streams", 1) == 0
        # a transcript that gives up while the resident kernel waits for its next challenge
        A, B = fc.rand_vec(1, 1 << 13, 1), fc.rand_vec(1, 1 << 13, 2)
        calls = []

        def gives_up(_coeffs):
            calls.append(1)
            if len(calls) == 4:
                raise RuntimeError("transcript refused")
            return sp.le(7 + len(calls))
        with pytest.raises(nova_amd.NmxError):
            fv.sumcheck_prove_quad_prod(1, sp.le(5), 13, dev(A), dev(B), gives_up)
        assert len(calls) == 4
        both(sp.check_quad_prod, g_quad, o_quad, 1, 13, seed=298)
    finally:
        assert L.nmx_set_option(b"sc_resident", 1) == 0
        assert L.nmx_set_option(b"sc_host_tail", 7) == 0
        assert L.nmx_set_option(b"sc_torn_test", 0) == 0
        assert L.nmx_set_option(b"sc_side_streams", 1) == 0


def test_a_torn_challenge_line_is_polled_past(nmx):
    """The challenge of a pre-launched pass reaches the device as four write-combined 16-byte stores; nothing guarantees that such a
    store arrives whole.  Option sc_torn_test makes the host write, ahead of every challenge, what a buffer evicted in 8-byte chunks
    could leave there: new sequence words and new checksum in all four pieces, the second half of each limb piece still stale.  The
    waiting pass binds its tables IN PLACE with whatever it accepts, so it must reject that line (checksum) and wait for the whole one:
    every proof equals the oracle's, and the device reports the rejected lines (VERDICT r5 weak #2 / next #1a)."""
    from nova_amd import _lib
    L = _lib.lib()
    p = fc.FIELDS[1]
    st0 = _lib.stats()
    try:
        assert L.nmx_set_option(b"sc_torn_test", 40) == 0          # the torn line stays for 40 us: a polling pass reads it many times
        for l in (10, 13, 15):
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, l, seed=180 + l, brute=False)
            both(sp.check_quad_prod, g_quad, o_quad, 1, l + 1, seed=190 + l)
        both(sp.check_cubic3, g_cubic3(), o_cubic3, 3, 12, seed=181, brute=False)
        both(sp.check_quad_prod, g_quad, o_quad, 0, 12, seed=182)
        both(sp.check_batch_eval, g_batch, o_batch, 1, [14, 11, 13], seed=183)
        both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 12, seed=184, force={0: 0, 3: 1, 7: p - 1, 9: 0}, brute=False)
    finally:
        assert L.nmx_set_option(b"sc_torn_test", 0) == 0
    st1 = _lib.stats()
    injected = st1[_lib.STAT_SC_TORN_INJECTED] - st0[_lib.STAT_SC_TORN_INJECTED]
    rejects = st1[_lib.STAT_SC_TORN_REJECTS] - st0[_lib.STAT_SC_TORN_REJECTS]
    if injected == 0:
        pytest.skip("no large BAR on this box: nothing is pre-launched, so no challenge line is written")
    # every torn line was written while a pass waited for it (the pass polls continuously): almost all are seen and rejected
    assert rejects >= injected // 2, (injected, rejects)


def test_a_fallback_round_of_one_claim_beside_prelaunched_passes_of_the_others(nmx):
    """claim j of a batch takes the tau = 0 fall-back (an eq point with a zero coordinate) in a round in which claim i's
    pass would have been enqueued ahead of its challenge.  The fall-back allocates and waits for the device; a pass waiting for the
    host would have blocked it until its 2 s time-out and the call would have failed.  Nothing is pre-launched into such a round."""
    import time
    t0 = time.perf_counter()
    for zc in ({1: [0]}, {1: [3, 4]}, {0: [5], 1: [2]}, {2: [0, 1, 2]}):
        both(sp.check_batch_eval, g_batch, o_batch, 1, [14, 12, 13], seed=210 + len(zc), zero_coords=zc)
    both(sp.check_batch_eval, g_batch, o_batch, 1, [13, 13], seed=215, zero_coords={1: list(range(13))})
    assert time.perf_counter() - t0 < 60, "a pre-launched pass sat out its time-out"


def test_host_added_partials_and_ticket_passes_agree(nmx):
    """option sc_host_parts: a pass of <= 64 blocks sends every block's partial sums to the host, which adds them up (default); 0: the
    block that draws the last ticket does.  Sizes around the 64-block limits of the one-lane (2^14 indices) and four-lane (2^12) forms,
    all three provers, polling off (sc_poll_us = 0: the host synchronises, then reads), repeated."""
    from nova_amd import _lib
    L = _lib.lib()
    try:
        for hp, poll in ((1, 2000), (0, 2000), (1, 0), (1, 2000)):
            assert L.nmx_set_option(b"sc_host_parts", hp) == 0
            assert L.nmx_set_option(b"sc_poll_us", poll) == 0
            for l in (9, 12, 14, 15, 16):
                both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, l, seed=140 + l, brute=False)
                both(sp.check_quad_prod, g_quad, o_quad, 1, l + 1, seed=150 + l)
            both(sp.check_batch_eval, g_batch, o_batch, 1, [16, 13, 15, 9], seed=160)
            assert L.nmx_set_option(b"sc_quad", 0) == 0
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 15, seed=161, brute=False)
            both(sp.check_quad_prod, g_quad, o_quad, 1, 16, seed=162)
            assert L.nmx_set_option(b"sc_quad", 1) == 0
    finally:
        assert L.nmx_set_option(b"sc_host_parts", 1) == 0
        assert L.nmx_set_option(b"sc_poll_us", 2000) == 0
        assert L.nmx_set_option(b"sc_quad", 1) == 0


def test_batch_claims_on_side_streams_and_on_one_stream_agree(nmx):
    """option sc_side_streams: the claims of a batch round are independent passes -- claim i > 0 runs on its own stream (default) or
    all of them queue on the context's.  Up to 16 claims of mixed sizes, the fallback rounds (which allocate and copy on the claim's
    stream), repeated calls (the streams are kept), and a cubic proof right behind on the context's stream."""
    from nova_amd import _lib
    L = _lib.lib()
    p = fc.FIELDS[1]
    try:
        for side in (1, 0, 1):
            assert L.nmx_set_option(b"sc_side_streams", side) == 0
            both(sp.check_batch_eval, g_batch, o_batch, 1, [15, 13, 16], seed=71)
            both(sp.check_batch_eval, g_batch, o_batch, 1, [9, 3, 12, 1, 10, 10, 7, 2, 11, 5, 8, 4, 6, 12, 9, 13], seed=72)
            both(sp.check_batch_eval, g_batch, o_batch, 1, [10, 12, 8], seed=73, force={0: 0, 3: 1, 7: p - 1, 11: 0})
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 12, seed=74, brute=False)
    finally:
        assert L.nmx_set_option(b"sc_side_streams", 1) == 0


@pytest.mark.parametrize("tail", [0, 1, 3, 6, 7, 8])
def test_every_host_tail_threshold_gives_the_same_proof(nmx, tail):
    """option sc_host_tail: tables of <= 2^tail elements finish on the host (sc_host.hpp; 0 = only the final values come over).  The
    proof is the same wherever the hand-over happens -- including instances that fit the tail from the start and batch claims that
    reach it in different rounds."""
    from nova_amd import _lib
    L = _lib.lib()
    assert L.nmx_set_option(b"sc_host_tail", 9) == _lib.E_ARG
    try:
        assert L.nmx_set_option(b"sc_host_tail", tail) == 0
        for l in (1, 3, 7, 9, 12):
            both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, l, seed=20 + l, brute=False)
            both(sp.check_quad_prod, g_quad, o_quad, 1, l, seed=30 + l)
        both(sp.check_cubic3, g_cubic3(), o_cubic3, 1, 9, seed=4, taus=[0] * 9, brute=False)
        for nrs in ([2, 11], [12, 5, 9], [7, 7]):
            both(sp.check_batch_eval, g_batch, o_batch, 1, nrs, seed=3 + sum(nrs))
        both(sp.check_batch_eval, g_batch, o_batch, 1, [5, 10], seed=901, force={0: 0, 4: 1, 9: 0})
    finally:
        assert L.nmx_set_option(b"sc_host_tail", 7) == 0


def test_concat_builds_z_on_the_device(nmx):
    import torch
    from nova_amd import fieldvec as fv
    W, X = fc.rand_vec(1, 1000, 1), fc.rand_vec(1, 3, 2)
    u = fc.rand_vec(1, 1, 3)
    z = fv.concat(1, [dev(W), u, X], n_out=2048)
    exp = np.zeros((2048, 32), np.uint8)
    exp[:1000], exp[1000], exp[1001:1004] = W, u[0], X
    assert np.array_equal(z.cpu().numpy(), exp)
    c = fv.concat(1, [z], async_=True)                     # a clone on the library's stream
    fv.sync()
    assert torch.equal(c.cpu(), z.cpu()) and c.data_ptr() != z.data_ptr()


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_transposed_product(nmx, fid):
    from nova_amd import fieldvec as fv
    k = sp.transposed_kat()
    m = fv.SparseMatrix(fid, k["indptr"], k["indices"], fc.vec(k["data"]), k["cols"])
    assert sp.ints(m.multiply_vec_transposed(fc.vec(k["x"])).tobytes()) == k["out"]
    m.close()
    for rows, cols, seed, heavy in ((50, 30, 1, (0, 29)), (3000, 640, 2, (0, 639)), (640, 3000, 3, (7,)), (20000, 4096, 4, (0, 1, 4095))):
        ip, ix, dt = sp.heavy_column_csr(fid, rows, cols, seed, heavy_cols=heavy)
        x = fc.edge_vectors(fid, rows, seed + 10)
        m = fv.SparseMatrix(fid, ip, ix, dt, cols)
        exp = cref.spmv_transposed(fid, ip, ix, dt, rows, cols, x)
        assert m.multiply_vec_transposed(x).tobytes() == exp                         # host operands
        got = m.multiply_vec_transposed(dev(x))                                      # HBM-resident, twice (the form is cached)
        assert got.cpu().numpy().tobytes() == exp
        dx = dev(x)
        ga = m.multiply_vec_transposed(dx, async_=True)                              # stream-ordered: complete after fv.sync()
        fv.sync()
        assert ga.cpu().numpy().tobytes() == exp
        # the forward product of the same registered matrix is untouched
        z = fc.edge_vectors(fid, cols, seed + 20)
        assert m.multiply_vec(z).tobytes() == cref.spmv(fid, ip, ix, dt, rows, z)
        m.close()


def test_spartan_prove_replay_matches_the_oracle(nmx):
    """bench.py's `spartan_replay` (snark.rs:113-260 as provider calls) at num_cons = 2^12: every commitment-free step of `prove`
    on HBM-resident vectors against the oracle run in the same order, round polynomials included."""
    import argparse
    import torch
    import bench
    args = argparse.Namespace(log2n=12, steps=1, warmup=1, no_cpu_baseline=False)
    out = bench.spartan_replay(args, torch)
    assert out["cpu_baseline"]["gpu_matches_cpu"] is True, out["cpu_baseline"]["checks"]
    assert all(out["proof_verifies"].values())
    assert out["config"]["rounds"] == [12, 13, 12]
    assert set(out["provers"]) == {"sumcheck_outer", "sumcheck_inner", "sumcheck_batch"}


@pytest.mark.parametrize("curve,ell", [(1, 14), (2, 11), (3, 11)])
def test_spartan_prove_replay_on_the_other_scalar_fields(nmx, curve, ell):
    """S2 of CompressedSNARK::prove is the same Spartan prover over GRUMPKIN's scalar field (= BN254 Fq) on the secondary circuit,
    ~2^14 constraints (src/nova/mod.rs:862-881); the Pasta cycle runs it over Pallas's / Vesta's.  The whole sequence (three
    sum-checks, evaluations, transposed products, batch witness) against the oracle and the reference's verifier equations."""
    import argparse
    import torch
    import bench
    args = argparse.Namespace(log2n=ell, steps=1, warmup=1, no_cpu_baseline=False, curve=curve)
    out = bench.spartan_replay(args, torch)
    assert out["cpu_baseline"]["gpu_matches_cpu"] is True, out["cpu_baseline"]["checks"]
    assert all(out["proof_verifies"].values())
    assert out["config"]["rounds"] == [ell, ell + 1, ell]


def test_provers_and_commitments_from_several_threads(nmx):
    """Six host threads at once -- cubic, quad_prod and batch provers (each leases its own context: mailbox, side streams, challenge
    lines of its pre-launched passes), begun commitments and synchronous ones on the same GPU.  Every result equals what the same call
    gives alone (expected values computed first, single-threaded, through the oracle)."""
    import threading
    import nova_amd
    from oracle import pyref as R
    from tests import util
    c = R.BN254_G1
    n = 6000
    bases = cref.sequential_bases(c, 21, n).copy()
    ck = nmx.CommitmentKey.from_host(c.cid, bases)
    ce = nmx.CommitmentEngine(c.cid)
    v = util.random_scalars(c.cid, n, seed=9)
    want_com = (lambda r: (r.xy, int(r.is_inf)))(ce.commit(ck, v))
    jobs = [
        (sp.check_cubic3, g_cubic3(), o_cubic3, (1, 12), dict(seed=201, brute=False)),
        (sp.check_cubic3, g_cubic3(), o_cubic3, (3, 10), dict(seed=202, brute=False)),
        (sp.check_quad_prod, g_quad, o_quad, (1, 14), dict(seed=203)),
        (sp.check_quad_prod, g_quad, o_quad, (0, 11), dict(seed=204)),
        (sp.check_batch_eval, g_batch, o_batch, (1, [12, 9, 13]), dict(seed=205)),
    ]
    want = [chk(o, *a, **kw) for chk, _g, o, a, kw in jobs]
    errs = []

    def prover(i):
        try:
            chk, g, _o, a, kw = jobs[i]
            for _ in range(3):
                assert chk(g, *a, **kw) == want[i], i
        except Exception as e:   # noqa: BLE001 -- reported by the main thread
            errs.append((i, repr(e)))

    def committer():
        try:
            for _ in range(12):
                t = ce.commit_begin(ck, v)
                got = ce.commit(ck, v)
                assert (got.xy, int(got.is_inf)) == want_com
                got = t.finish()
                assert (got.xy, int(got.is_inf)) == want_com
        except Exception as e:   # noqa: BLE001
            errs.append(("commit", repr(e)))
    ths = [threading.Thread(target=prover, args=(i,)) for i in range(len(jobs))] + [threading.Thread(target=committer)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    ck.close()


def test_quad_prod_round_polynomial_is_the_references_known_answer(nmx):
    """src/spartan/polys/univariate.rs:284-300 through the HIP prover: A = [1, 2], B = [1, 3], claim 7 -> the round polynomial 2x^2 + 3x + 1;
    the same instance embedded in the top variable of a 2^10 table (zeros elsewhere) so that the DEVICE computes that first round."""
    for l in (1, 10):
        n = 1 << l
        A, B = [0] * n, [0] * n
        A[0], A[n // 2], B[0], B[n // 2] = 1, 2, 1, 3
        tr = sp.StandInTranscript(fc.FIELDS[1])
        polys, _r, _c = g_quad(1, sp.le(7), l, fc.vec(A), fc.vec(B), tr)
        assert [int.from_bytes(c, "little") for c in polys[0]] == [1, 3, 2]


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_three_products_in_one_call(nmx, fid):
    """nmx_spmv_apply_many: R1CSShape::multiply_vec (A z, B z, C z; src/r1cs/mod.rs:407-471) and compute_eval_table_sparse (the three
    transposed products; src/spartan/mod.rs:497-533) as one call each, the matrices side by side on side streams -- against the oracle
    per matrix, HBM-resident (synchronous and stream-ordered), host operands, repeated (the side streams are reused), with a heavy
    column in one matrix only, and followed at once by a stream-ordered consumer of the results."""
    from nova_amd import fieldvec as fv
    rows, cols = 9000, 5000
    csr = [sp.heavy_column_csr(fid, rows, cols, 11 + j, heavy_cols=(0, cols - 1) if j == 1 else ()) for j in range(3)]
    mats = [fv.SparseMatrix(fid, ip, ix, dt, cols) for ip, ix, dt in csr]
    z, x = fc.edge_vectors(fid, cols, 31), fc.edge_vectors(fid, rows, 32)
    want_f = [cref.spmv(fid, ip, ix, dt, rows, z) for ip, ix, dt in csr]
    want_t = [cref.spmv_transposed(fid, ip, ix, dt, rows, cols, x) for ip, ix, dt in csr]
    dz, dx = dev(z), dev(x)
    for _ in range(3):
        assert [o.cpu().numpy().tobytes() for o in fv.multiply_vec_many(mats, dz)] == want_f
        assert [o.cpu().numpy().tobytes() for o in fv.multiply_vec_many(mats, dx, transposed=True)] == want_t
    outs = fv.multiply_vec_many(mats, dz, async_=True)
    s = fv.axpy(fid, outs[0], outs[2], fc.rand_vec(fid, 1, 5), async_=True)          # a consumer on the call's stream
    outs_t = fv.multiply_vec_many(mats[:2], dx, transposed=True, async_=True)
    fv.sync()
    assert [o.cpu().numpy().tobytes() for o in outs] == want_f
    assert [o.cpu().numpy().tobytes() for o in outs_t] == want_t[:2]
    assert s.cpu().numpy().tobytes() == cref.field_axpy(fid, np.frombuffer(want_f[0], np.uint8).reshape(rows, 32),
                                                         np.frombuffer(want_f[2], np.uint8).reshape(rows, 32), fc.rand_vec(fid, 1, 5), rows)
    assert [o.tobytes() for o in fv.multiply_vec_many(mats, z)] == want_f              # host operands
    assert [o.tobytes() for o in fv.multiply_vec_many(mats, x, transposed=True)] == want_t
    for m in mats:
        m.close()


def test_evaluations_and_many_products_from_several_threads(nmx):
    """Round 6's mailbox evaluations (more polynomials than mailbox slots in one call) and the many-matrix product from four host threads
    at once, beside a prover: every call leases its own context (mailbox, side streams); results equal the single-threaded ones."""
    import threading
    import torch
    from nova_amd import fieldvec as fv
    fid, ell = 1, 12
    n = 1 << ell
    zs = [fc.edge_vectors(fid, n, 300 + i) for i in range(19)]           # 19 > 16 mailbox slots: two passes inside one call
    r = fc.rand_vec(fid, ell, 320)
    want_ev = cref.mle_multi_evaluate(fid, [z.tobytes() for z in zs], ell, r)
    dz = [dev(z) for z in zs]
    assert fv.mle_multi_evaluate(fid, dz, r) == want_ev
    csr = [sp.heavy_column_csr(fid, 3000, 2000, 330 + j, heavy_cols=(0,)) for j in range(3)]
    mats = [fv.SparseMatrix(fid, ip, ix, dt, 2000) for ip, ix, dt in csr]
    x = fc.edge_vectors(fid, 3000, 340)
    want_t = [cref.spmv_transposed(fid, ip, ix, dt, 3000, 2000, x) for ip, ix, dt in csr]
    dx = dev(x)
    want_q = sp.check_quad_prod(o_quad, fid, 13, seed=350)
    errs = []

    def work(i):
        try:
            for _ in range(6):
                if i == 0:
                    assert fv.mle_multi_evaluate(fid, dz, r) == want_ev
                elif i == 1:
                    assert [o.cpu().numpy().tobytes() for o in fv.multiply_vec_many(mats, dx, transposed=True)] == want_t
                elif i == 2:
                    assert [fv.mle_evaluate(fid, z, r) for z in dz[:5]] == want_ev[:5]
                else:
                    assert sp.check_quad_prod(g_quad, fid, 13, seed=350) == want_q
        except Exception as e:   # noqa: BLE001 -- reported by the main thread
            errs.append((i, repr(e)))
    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for m in mats:
        m.close()
