#!/usr/bin/env python3
"""bench.py -- BN254 MSM throughput (BASELINE.json metric: scalar-point pairs/s) on N MI355X GPUs of one node.

A "step" is one pass of the hot path over one batch of synthetic input: one `CommitmentEngine::commit(ck, v, 0)`
(= DlogGroupExt::vartime_multiscalar_mul, /root/reference benches/commit.rs:120-124) over 2^LOG2N uniformly random
BN254 scalars per GPU, commitment key resident in HBM, scalars resident in HBM when the timed region starts.
N > 1: every rank runs the full single-GPU MSM on its own contiguous shard of the (scalar, base) array (weak
scaling: 2^LOG2N pairs per GPU), then the 128-byte partial sums are all-gathered over RCCL and combined
(SURVEY.md 8(e)); value = pairs of all ranks / max-over-ranks time.

  python bench.py                      # 1 GPU, 2^20 pairs, prints ONE JSON line
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus 8 --steps K --warmup W
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_PAIR = 96            # 32 B scalar + 64 B affine base, each read once
STAGES = ["digits", "sort", "bounds_plan", "accum", "fold", "reduce", "tail"]
MAD_PEAK_T = 28.8              # measured v_mad_u64_u32 rate, T/s (bench/ubench.hip)
# multiply-adds per XYZZ mixed addition (curve.hpp add_affine: 5 products, 2 squarings, 1 two-product sum) by base
# field: BN254 Fq / Fr 5 x 162 + 2 x 126 + 243; the Pasta moduli have three zero limbs of nine, whose reduction
# terms are dropped at compile time: 5 x 135 + 2 x 99 + 216
MADS_PER_MADD_BY_CURVE = {0: 1305, 1: 1305, 2: 1089, 3: 1089}


def madds_per_launch(n, args):
    """Mixed additions of one accumulate launch on uniformly random scalars: one per (pair, window)."""
    bits = {0: 254, 1: 254, 2: 255, 3: 255}[args.curve]
    c = args.window_bits or (20 if args.log2n >= 22 else 16)
    return n * (-(-(bits + 1) // c))


def pmc_traffic(args, world):
    """HBM bytes per accumulate launch from the committed PMC summary -- only for the exact configuration it was
    collected on (BN254, 2^20, random scalars, one GPU); everything else reports null."""
    if not (world == 1 and args.curve == 0 and args.log2n == 20 and args.dist == "random" and not args.window_bits):
        return None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_msm_2p20", "pmc_traffic.json")))["accum"]
        return d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=20, help="pairs per GPU = 2^log2n (BASELINE configs[1]: 20)")
    ap.add_argument("--curve", type=int, default=0, help="0 bn254_g1 (headline), 1 grumpkin, 2 pallas, 3 vesta")
    ap.add_argument("--dist", default="random", help="scalar distribution: random | u1 | u10 | u16 | u32 | u64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--iters", type=int, default=65536, help="prove_step_replay: MinRoot iterations per step")
    ap.add_argument("--workload", default="msm", choices=["msm", "axpy", "cross_term", "bind", "sumcheck3", "round3", "quad_prod", "lincomb8", "horner", "mle_eval", "spmv", "prove_step_replay", "hyperkzg_replay"],
                    help="msm = the headline (default); the others time one HBM-bound field-vector kernel of "
                         "SURVEY.md 8(f) at 2^log2n elements per GPU")
    args = ap.parse_args()

    # RCCL writes its version banner / warnings to stdout; stdout must end with ONE JSON line: send RCCL's log to
    # stderr and (see emit()) print the JSON only after the process group is gone and C stdio is flushed.
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    # NMX_BENCH_FORCE_DIST=1 exercises the RCCL exchange path with a single rank (1-GPU boxes)
    force_dist = os.environ.get("NMX_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        if force_dist and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import nova_amd
    from nova_amd import _lib
    from tests import util
    L = _lib.lib()
    rc = L.nmx_init(local_rank)
    assert rc == 0, L.nmx_last_error().decode()
    if args.window_bits:
        L.nmx_set_window_bits(args.window_bits)

    if args.workload == "prove_step_replay":
        return prove_step_replay(args, world, rank, L, torch, dist)
    if args.workload == "hyperkzg_replay":
        return hyperkzg_replay(args, world, rank, L, torch, dist)
    if args.workload != "msm":
        return field_workload(args, world, rank, L, torch, dist)

    n = 1 << args.log2n
    cid = args.curve
    ce = nova_amd.CommitmentEngine(cid)
    group = ce.group
    # shard `rank` of the key: bases P_i = (1 + rank*n + i) * G, generated in HBM
    ck = nova_amd.CommitmentKey.generate(cid, n, k0=1 + rank * n)
    # two scalar vectors per rank, alternated between steps, resident in HBM before timing starts
    host_sc = [util.scalar_set(cid, n, args.dist, seed=util.SEED + 1000 * rank + j) for j in range(2)]
    dev_sc = [torch.from_numpy(s.copy()).cuda() for s in host_sc]
    from nova_amd.dist import sharded_msm

    def step(j):
        if world == 1 and not force_dist:
            return group.vartime_multiscalar_mul(dev_sc[j & 1], ck)
        return sharded_msm(group, ck, dev_sc[j & 1])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for j in range(args.warmup):
        step(j)
    L.nmx_set_profiling(1)
    import ctypes
    prof = (ctypes.c_float * 16)()
    stage_sum = np.zeros(len(STAGES))
    fence()
    t0 = time.perf_counter()
    result = None
    per_call = []
    for j in range(args.steps):
        tj = time.perf_counter()
        result = step(j)  # returns the point: every call ends with its own device->host copy, so this is a full latency
        per_call.append(time.perf_counter() - tj)
        k = L.nmx_profile_last(prof, 16)  # hipEvent times of this step's kernels, on the library's stream
        stage_sum[:k] += np.array(prof[:k])
    fence()
    dt = time.perf_counter() - t0
    L.nmx_set_profiling(0)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    out = None
    MADS_PER_MADD = MADS_PER_MADD_BY_CURVE[cid]
    if rank == 0:
        stage_ms = stage_sum / max(args.steps, 1)
        accum_ms = float(stage_ms[STAGES.index("accum")])
        pairs = n * world * args.steps
        achieved = BYTES_PER_PAIR * n / (accum_ms * 1e-3) / 1e9 if accum_ms > 0 else 0.0
        out = {
            "metric": "BN254 MSM scalar-point pairs/sec" if cid == 0 else f"{nova_amd.CURVE_NAMES[cid]} MSM scalar-point pairs/sec",
            "value": pairs / dt,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32x8 (256-bit modular integer, Montgomery)",
            "data": "synthetic",
            "config": {
                "workload": f"{nova_amd.CURVE_NAMES[cid]} Pippenger MSM 2^{args.log2n} pairs per GPU, {args.dist} scalars, "
                            "bases+scalars resident in HBM (BASELINE.json configs[1])",
                "pairs_per_gpu": n,
                "parallelism": f"shard{world}" if world > 1 else "single",
                "combine": "rccl all_gather of 128-byte partials + host point sum" if world > 1 else "none",
            },
            "stages_ms": {s: round(float(v), 4) for s, v in zip(STAGES, stage_ms)},
            # median and min of the timed calls (rank 0's own calls; `value` uses the whole region)
            "per_call_ms": {"median": round(float(np.median(per_call)) * 1e3, 4), "min": round(min(per_call) * 1e3, 4)},
            "roofline": {
                "bound": "hbm",
                "kernel": "k_launch<AccumFn> (bucket accumulation)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args, world),
                "note": "algorithmic bytes = 96 B/pair x pairs per launch / accum-kernel time (hipEvents on the "
                        "library stream); the MSM is integer-VALU-bound, not HBM-bound (DESIGN.md). traffic = HBM bytes "
                        "per launch from the separate rocprofv3 --pmc passes in profiles/ (null for other configs).",
                # the roofline that actually binds this kernel: 32x32+64 multiply-adds against the measured
                # v_mad_u64_u32 rate (profiles/r01_ubench_instruction_rates.jsonl)
                "valu_mad": {"achieved_T_per_s": MADS_PER_MADD * madds_per_launch(n, args) / (accum_ms * 1e-3) / 1e12 if accum_ms > 0 else 0.0,
                             "peak_T_per_s": MAD_PEAK_T, "frac": (MADS_PER_MADD * madds_per_launch(n, args) / (accum_ms * 1e-3) / 1e12) / MAD_PEAK_T if accum_ms > 0 else 0.0,
                             "mads_per_mixed_add": MADS_PER_MADD},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cid, ck, host_sc[(args.steps - 1) & 1], n, result)
    ck.close()
    emit(out if rank == 0 else None, world > 1 or force_dist, dist)


def emit(result, world_or_dist, dist):
    """Tear the process group down first, flush every C stdio buffer (library banners), then print the one JSON
    line as the very last thing on rank 0's stdout."""
    import ctypes
    if world_or_dist:
        dist.barrier()
        dist.destroy_process_group()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if result is not None:
        print(json.dumps(result), flush=True)


def witness_like(cid, n, seed):
    """Scalars shaped like an R1CS witness: half zeros, a quarter small (< 2^16), a quarter full-width
    (why msm() partitions by bit width, src/provider/msm.rs:237-279)."""
    from tests import util
    v = util.random_scalars(cid, n, seed=seed).copy()
    rng = np.random.Generator(np.random.PCG64(seed))
    kind = rng.integers(0, 4, size=n)
    v[kind < 2] = 0
    small = util.u64_to_le32(util.small_scalars(n, 16, seed=seed))
    v[kind == 2] = small[kind == 2]
    return v


def prove_step_replay(args, world, rank, L, torch, dist):
    """REPLAY of the provider calls of one RecursiveSNARK::prove_step (src/nova/mod.rs:456-541) on the
    MinRoot step circuit (examples/minroot.rs: 3 constraints per iteration + the 9 986-constraint augmented circuit
    on BN254, 10 538 on Grumpkin): 4 MSMs + 2 cross terms + 4 AXPY folds, vectors resident in HBM, commitments
    returned to the host after each MSM (they feed the Poseidon RO challenge on the reference side).  NOT replayed:
    witness synthesis, Poseidon hashing and the sparse matrix-vector products (CPU side of the reference; SpMV is
    SURVEY 8(f) row 3, not built yet).  This is a replay, not prove_step: the Rust reference cannot be built here."""
    import nova_amd
    from nova_amd import fieldvec as fv
    from tests import util
    assert world == 1, "prove_step is sequential: replicas only"
    N = 3 * args.iters + 9986          # primary (BN254) witness / constraint count
    n2 = 10538                         # secondary (Grumpkin)
    cur = {"P": (0, fv.SCALAR_FIELD_OF_CURVE[0], N), "S": (1, fv.SCALAR_FIELD_OF_CURVE[1], n2)}
    ce = {k: nova_amd.CommitmentEngine(c[0]) for k, c in cur.items()}
    ck = {k: ce[k].setup_synthetic(c[2], k0=3) for k, c in cur.items()}
    host, dev = {}, {}
    for k, (cid, fid, n) in cur.items():
        host[k] = {"W": witness_like(cid, n, 11), "W1": util.random_scalars(cid, n, seed=12),
                   "E1": util.random_scalars(cid, n, seed=13)}
        for j, nm in enumerate(("AZ", "BZ", "CZ")):
            host[k][nm] = util.random_scalars(cid, n, seed=20 + j)
        dev[k] = {nm: torch.from_numpy(v).cuda() for nm, v in host[k].items()}
    u = util.random_scalars(0, 1, seed=31)
    r = {k: util.random_scalars(c[0], 1, seed=32) for k, c in cur.items()}
    rT = {k: util.random_scalars(c[0], 1, seed=33) for k, c in cur.items()}
    uS = util.random_scalars(1, 1, seed=34)

    def nifs(k, uu):
        cid, fid, n = cur[k]
        d = dev[k]
        T = fv.cross_term(fid, d["AZ"], d["BZ"], d["CZ"], d["E1"], uu)     # r1cs/mod.rs:614-620
        comT = ce[k].commit(ck[k], T, rT[k])                                  # r1cs/mod.rs:622
        W = fv.axpy(fid, d["W1"], d["W"], r[k])                               # r1cs/mod.rs:1058-1062
        E = fv.axpy(fid, d["E1"], T, r[k])                                    # r1cs/mod.rs:1063-1067
        return comT, W, E

    def step():
        out = []
        out.append(nifs("S", uS)[0])                                          # nova/mod.rs:464  NIFS on the secondary
        out.append(ce["P"].commit(ck["P"], dev["P"]["W"]))                    # nova/mod.rs:477-496 primary witness commit
        out.append(nifs("P", u)[0])                                           # nova/mod.rs:502  NIFS on the primary
        out.append(ce["S"].commit(ck["S"], dev["S"]["W"]))                    # nova/mod.rs:515-541 secondary witness commit
        return out

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    outj = {
        "metric": "RecursiveSNARK prove_step provider-call REPLAY ms (minroot, BN254/Grumpkin)", "value": dt * 1e3, "unit": "ms",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": False,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 (256-bit modular integer)", "data": "synthetic",
        "config": {"workload": f"prove_step replay: minroot {args.iters} iterations/step -> primary N={N} (BN254), secondary n={n2} "
                               "(Grumpkin); 4 MSMs + 2 cross terms + 4 folds; no SpMV / synthesis / Poseidon (BASELINE.json configs[3])"},
        "roofline": None,
    }
    if not args.no_cpu_baseline:
        from oracle import cref
        threads = effective_cpus()
        cref.set_threads(threads)
        keys = {k: ck[k].read(0, cur[k][2]) for k in cur}
        prep = {k: cref.Prepared(cur[k][0], keys[k], cur[k][2]) for k in cur}
        t1 = time.perf_counter()
        exp = []
        for k, uu in (("S", uS), ("P", u)):
            cid, fid, n = cur[k]
            h = host[k]
            T = cref.field_cross_term(fid, h["AZ"], h["BZ"], h["CZ"], h["E1"], uu, n)
            comT = cref.commit(cid, T, keys[k], n, ck[k].h, rT[k])
            cref.field_axpy(fid, h["W1"], h["W"], r[k], n)
            cref.field_axpy(fid, h["E1"], T, r[k], n)
            comW = prep[k].msm(h["W"], n)
            exp.append((comT, comW))
        t_cpu = time.perf_counter() - t1
        ok = [(res[0].xy, int(res[0].is_inf)) == exp[0][0], (res[1].xy, int(res[1].is_inf)) == exp[1][1],
              (res[2].xy, int(res[2].is_inf)) == exp[1][0], (res[3].xy, int(res[3].is_inf)) == exp[0][1]]
        outj["cpu_baseline"] = {"value": t_cpu * 1e3, "unit": "ms", "cores": threads, "kind": "port",
                                "sample": "the same call sequence once through oracle/nova_ref.c (commit re-loads the key "
                                          "each call, as a fresh Vec<Affine> would not)", "gpu_matches_cpu": all(ok)}
    print(json.dumps(outj), flush=True)
    for k in ck:
        ck[k].close()


def hyperkzg_replay(args, world, rank, L, torch, dist):
    """REPLAY of the provider-side work of one HyperKZG `prove` (src/provider/hyperkzg.rs:926-1110) for
    n = 2^log2n on BN254, everything resident in HBM, commitments / evaluations returned to the host:
      ell-1 pair folds Pi[j] = P[2j] + x*(P[2j+1]-P[2j])                      (hyperkzg.rs:1085-1095)
      batch_commit of the folded polynomials, lengths n/2 ... 2               (hyperkzg.rs:1100)
      3*ell Horner evaluations f_i(u_j)                                       (hyperkzg.rs:1011-1020,1049-1056)
      B = sum q^i f_i                                                         (hyperkzg.rs:1028-1040)
      3 x kzg_open: h = div_by_monomial(B, u_j), commit(h)                    (hyperkzg.rs:961-1004,1062-1065)
    NOT replayed: the Keccak transcript (challenges are fixed random scalars).  A replay, not `prove`."""
    import nova_amd
    from nova_amd import fieldvec as fv
    from tests import util
    assert world == 1
    ell = args.log2n
    n = 1 << ell
    cid = 0
    fid = fv.SCALAR_FIELD_OF_CURVE[cid]
    ce = nova_amd.CommitmentEngine(cid)
    ck = ce.setup_synthetic(n, k0=5)
    hP = util.random_scalars(cid, n, seed=41)
    xs = util.random_scalars(cid, ell, seed=42)
    us = util.random_scalars(cid, 3, seed=43)
    qs = util.random_scalars(cid, ell, seed=44)
    dP = torch.from_numpy(hP).cuda()

    def step():
        polys, cur = [dP], dP
        for i in range(ell - 1):
            cur = fv.fold_pairs(fid, cur, xs[ell - i - 1])
            polys.append(cur)
        coms = ce.batch_commit(ck, polys[1:])
        evals = fv.poly_eval_multi(fid, polys, us)   # the whole v matrix (hyperkzg.rs:1049-1056) in one launch
        B = fv.lincomb_powers(fid, polys, qs[0])     # B = sum_i q^i f_i (kzg_compute_batch_polynomial, hyperkzg.rs:1028-1040)
        # the three openings run in parallel in the reference too (`u.into_par_iter()`, hyperkzg.rs:1062-1065)
        opens = [None] * 3

        def open_at(j):
            opens[j] = ce.commit(ck, fv.div_by_monomial(fid, B, us[j]).contiguous())
        if os.environ.get("NMX_REPLAY_SERIAL_OPENS"):
            for j in range(3):
                open_at(j)
        else:
            ths = [threading.Thread(target=open_at, args=(j,)) for j in range(3)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        return coms, evals, opens

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        coms, evals, opens = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    outj = {
        "metric": "HyperKZG prove provider-call REPLAY ms (BN254)", "value": dt * 1e3, "unit": "ms", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": False, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32x8 (256-bit modular integer)", "data": "synthetic",
        "config": {"workload": f"HyperKZG prove replay, n = 2^{ell}: {ell - 1} pair folds, batch_commit of lengths n/2..2, {3 * ell} Horner "
                               "evaluations (one launch), batch polynomial (one launch), 3 x (div_by_monomial + MSM of n-1) (BASELINE.json configs[4]); no transcript"},
        "roofline": None,
    }
    if not args.no_cpu_baseline:
        from oracle import cref
        threads = effective_cpus()
        cref.set_threads(threads)
        key = ck.read(0, n)
        prep = cref.Prepared(cid, key, n)
        t1 = time.perf_counter()
        cur, hp = hP, [hP]
        for i in range(ell - 1):
            m = len(cur) // 2
            cur = np.frombuffer(cref.field_bind(fid, cur, 0, 1, 2, xs[ell - i - 1], m), np.uint8).reshape(m, 32)
            hp.append(cur)
        ecoms = [prep.msm(p_, len(p_)) for p_ in hp[1:]]
        eevals = [[cref.suffix_horner(fid, f, len(f), us[j])[:32] for j in range(3)] for f in hp]
        Bh = np.frombuffer(cref.lincomb_powers(fid, [h.tobytes() for h in hp], qs[0], n), np.uint8).reshape(n, 32)
        eopens = []
        for j in range(3):
            h = np.frombuffer(cref.suffix_horner(fid, Bh, n, us[j]), np.uint8).reshape(n, 32)[1:]
            eopens.append(prep.msm(np.ascontiguousarray(h), n - 1))
        t_cpu = time.perf_counter() - t1
        ok = ([(c.xy, int(c.is_inf)) for c in coms] == ecoms and evals == eevals
              and [(c.xy, int(c.is_inf)) for c in opens] == eopens)
        outj["cpu_baseline"] = {"value": t_cpu * 1e3, "unit": "ms", "cores": threads, "kind": "port",
                                "sample": "the same call sequence once through oracle/nova_ref.c (Horner / division passes are "
                                          "single-threaded there)", "gpu_matches_cpu": ok}
    print(json.dumps(outj), flush=True)
    ck.close()


def effective_cpus():
    """Host cores this process may actually use: min(affinity, cgroup CPU quota).  The GPU boxes expose 256
    hardware threads but a 16-CPU cgroup quota; oversubscribing it makes the OpenMP baseline 4x slower
    (profiles/r01_msm_2p20/cpu_baseline_thread_scaling.txt)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def field_workload(args, world, rank, L, torch, dist):
    """One field-vector kernel per step on HBM-resident vectors (BN254 scalar field unless --curve); ranks are
    independent replicas.  Workloads and the reference loops they replace:
      axpy        NIFS witness fold                      r1cs/mod.rs:1058-1067
      cross_term  commit_T's T                           r1cs/mod.rs:614-620
      bind        MLE bind_poly_var_top                  spartan/polys/multilinear.rs:65-84
      sumcheck3   eq-factored cubic round sums           spartan/sumcheck.rs:900-958
      round3      one whole cubic round fused: bind A, B, C with the challenge + the next round's sums
                  (nmx_sumcheck_bind_eq_sums; the reference: sumcheck.rs:535-545 then :900-958)
      quad_prod   plain quadratic round sums             spartan/sumcheck.rs:163-186
      lincomb8    PolyEvalWitness::batch of 8 polys      spartan/mod.rs:223-277
      horner      poly_eval + div_by_monomial            provider/hyperkzg.rs:946-1020
      mle_eval    MultilinearPolynomial::evaluate_with   spartan/polys/multilinear.rs:98-129
      spmv        R1CS A*z, 3 non-zeros per row average  r1cs/sparse.rs:201-229"""
    import ctypes
    from nova_amd import fieldvec as fv
    from oracle import cref
    from tests import util
    n = 1 << args.log2n
    cid = args.curve
    fid = fv.SCALAR_FIELD_OF_CURVE[cid]
    wl = args.workload
    nvec = {"axpy": 2, "cross_term": 4, "bind": 1, "sumcheck3": 3, "round3": 3, "quad_prod": 2, "lincomb8": 8, "horner": 1,
            "mle_eval": 1, "spmv": 1}[wl]
    host = [util.random_scalars(cid, n, seed=util.SEED + 7 * j + rank) for j in range(nvec)]
    dev = [torch.from_numpy(h).cuda() for h in host]
    r = util.random_scalars(cid, 1, seed=99)
    # algorithmic reads + writes per element of the input vectors (sumcheck3: a0,a1,b0,b1,c0 per index = 80 B per
    # element; quad_prod: a0,a1,b0,b1 per index = 64 B per element; horner: read f, write out = 64 B;
    # lincomb8: 8 reads + 1 write; mle_eval: one read; spmv: per row 3 x (32 B value + 4 B index + 32 B gathered z)
    # + 8 B indptr + 32 B result)
    # round3: three tables read once (96 B per element) and their bound halves written (48 B)
    bytes_per_elem = {"axpy": 96, "cross_term": 160, "bind": 48, "sumcheck3": 80, "round3": 144, "quad_prod": 64, "lincomb8": 288,
                      "horner": 64, "mle_eval": 32, "spmv": 3 * 68 + 40}[wl]
    shift = (args.log2n - 1) // 2
    eqR = eqL = mat = csr = None
    if wl == "sumcheck3":
        eqR = torch.from_numpy(util.random_scalars(cid, 1 << shift, seed=5)).cuda()
        eqL = torch.from_numpy(util.random_scalars(cid, (n // 2) >> shift, seed=6)).cuda()
    if wl == "round3":  # the eq tables of the NEXT round (half length n / 4)
        shift = (args.log2n - 2) // 2
        eqR = torch.from_numpy(util.random_scalars(cid, 1 << shift, seed=5)).cuda()
        eqL = torch.from_numpy(util.random_scalars(cid, (n // 4) >> shift, seed=6)).cuda()
        work = [torch.empty_like(t) for t in dev]
    if wl == "mle_eval":
        point = util.random_scalars(cid, args.log2n, seed=8)
    if wl == "spmv":  # n rows x n columns, 3 random non-zeros per row (the minroot shape: 3 constraints per iteration)
        rng = np.random.Generator(np.random.PCG64(5))
        indptr = np.arange(0, 3 * n + 1, 3, dtype=np.uint64)
        indices = rng.integers(0, n, size=3 * n).astype(np.uint64)
        data = util.random_scalars(cid, 3 * n, seed=4)
        csr = (indptr, indices, data)
        mat = fv.SparseMatrix(fid, indptr, indices, data, n)

    def step():
        if wl == "axpy":
            return fv.axpy(fid, dev[0], dev[1], r)
        if wl == "cross_term":
            return fv.cross_term(fid, dev[0], dev[1], dev[2], dev[3], r)
        if wl == "sumcheck3":
            return fv.sumcheck_eq_sums(fid, 3, dev[0], dev[1], dev[2], eqR, eqL, shift)
        if wl == "round3":  # binds in place: work on copies refreshed outside the kernel timing (hipEvents bracket the kernel)
            for w, t in zip(work, dev):
                w.copy_(t)
            torch.cuda.synchronize()
            return fv.sumcheck_bind_eq_sums(fid, 3, work[0], work[1], work[2], r, eqR, eqL, shift)[3]
        if wl == "quad_prod":
            return fv.sumcheck_plain_sums(fid, 1, dev[0], dev[1])
        if wl == "lincomb8":
            return fv.lincomb_powers(fid, dev, r)
        if wl == "horner":
            return fv.suffix_horner(fid, dev[0], r)
        if wl == "mle_eval":
            return fv.mle_evaluate(fid, dev[0], point)
        if wl == "spmv":
            return mat.multiply_vec(dev[0])
        return fv.bind_poly_var_top(fid, dev[0], r)

    def cpu(m):
        """the oracle on the first m elements -> (bytes to compare, number of elements the time covers)"""
        if wl == "axpy":
            return cref.field_axpy(fid, host[0][:m], host[1][:m], r, m), m
        if wl == "cross_term":
            return cref.field_cross_term(fid, host[0][:m], host[1][:m], host[2][:m], host[3][:m], r, m), m
        if wl == "sumcheck3":
            return b"".join(cref.sumcheck_eq_sums(fid, 3, host[0], host[1], host[2], n, eqR.cpu().numpy(), eqL.cpu().numpy(), shift)), n
        if wl == "round3":
            bound = [cref.field_bind(fid, h, 0, n // 2, 1, r, n // 2) for h in host]
            return b"".join(cref.sumcheck_eq_sums(fid, 3, bound[0], bound[1], bound[2], n // 2, eqR.cpu().numpy(),
                                                  eqL.cpu().numpy(), shift)), n
        if wl == "quad_prod":
            return b"".join(cref.sumcheck_plain_sums(fid, 1, host[0], host[1], None, n)[:2]), n
        if wl == "lincomb8":
            return cref.lincomb_powers(fid, [h[:m].tobytes() for h in host], r, m), m
        if wl == "horner":
            return cref.suffix_horner(fid, host[0], n, r), n
        if wl == "mle_eval":
            return cref.mle_evaluate(fid, host[0], args.log2n, point), n
        if wl == "spmv":
            return cref.spmv(fid, csr[0][: m + 1], csr[1], csr[2], m, host[0]), m
        return cref.field_bind(fid, host[0], 0, n // 2, 1, r, min(m, n // 2)), min(m, n // 2)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    L.nmx_set_profiling(1)
    prof = (ctypes.c_float * 4)()
    ksum = 0.0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        L.nmx_profile_last(prof, 4)
        ksum += prof[0]
    fence()
    dt = time.perf_counter() - t0
    L.nmx_set_profiling(0)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    res = None
    if rank == 0:
        kms = ksum / args.steps
        achieved = bytes_per_elem * n / (kms * 1e-3) / 1e9
        res = {
            "metric": f"field elements/sec ({wl})", "value": n * world * args.steps / dt, "unit": "elements/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32x8 (256-bit modular integer)", "data": "synthetic",
            "config": {"workload": f"{wl} over 2^{args.log2n} {['bn254_fq','bn254_fr','pasta_fp','pasta_fq'][fid]} "
                                   "elements per GPU, HBM-resident (SURVEY.md 8(f))", "parallelism": f"replicas{world}"},
            "kernel_ms": kms,
            "roofline": {"bound": "hbm", "kernel": f"k_launch<{wl}>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "note": f"{bytes_per_elem} algorithmic bytes per element / kernel time from hipEvents on the library stream"},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = effective_cpus()
            cref.set_threads(threads)
            t1 = time.perf_counter()
            exp, cnt = cpu(min(n, 1 << 22))
            t = time.perf_counter() - t1
            if isinstance(out, tuple):
                got = b"".join(out)
            elif isinstance(out, bytes):
                got = out
            else:
                got = out.cpu().numpy().reshape(-1).tobytes()
            res["cpu_baseline"] = {"value": cnt / t, "unit": "elements/s", "cores": threads, "kind": "port",
                                   "sample": f"{cnt} elements, one pass, oracle/nova_ref.c",
                                   "gpu_matches_cpu": got[: len(exp)] == exp}
    if mat is not None:
        mat.close()
    emit(res if rank == 0 else None, world > 1, dist)


def cpu_baseline(cid, ck, scalars, n, gpu_result):
    """The oracle (C restatement of the reference's msm(), OpenMP over all host cores) timed on the same inputs,
    bounded to ~10-30 s of CPU work; also cross-checks the GPU result when the sample is the whole workload."""
    from oracle import cref
    threads = effective_cpus()
    cref.set_threads(threads)
    sample = min(n, 1 << 20)
    host = ck.read(0, sample)
    prep = cref.Prepared(cid, host, sample)
    sc = np.ascontiguousarray(scalars[:sample])
    best = None
    res = None
    for _ in range(2):
        t0 = time.perf_counter()
        res = prep.msm(sc, sample)
        t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    out = {
        "value": sample / best,
        "unit": "pairs/s",
        "cores": threads,
        "kind": "port",
        "sample": f"2^{sample.bit_length() - 1} pairs of the same workload, best of 2, oracle/nova_ref.c "
                  "(C restatement of msm.rs + Pippenger in the msm_best role; the Rust reference cannot be built here)",
        "seconds": best,
    }
    if sample == n and gpu_result is not None and len(gpu_result.xy) == 64:
        out["gpu_matches_cpu"] = (gpu_result.xy, int(gpu_result.is_inf)) == res
    return out


if __name__ == "__main__":
    main()
