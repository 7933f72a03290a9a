#!/usr/bin/env python3
"""bench.py -- BN254 MSM throughput (BASELINE.json metric: scalar-point pairs/s) on N MI355X GPUs of one node.

A "step" is one pass of the hot path over one batch of synthetic input: one `CommitmentEngine::commit(ck, v, 0)`
(= DlogGroupExt::vartime_multiscalar_mul, /root/reference benches/commit.rs:120-124) over uniformly random BN254
scalars, commitment key resident in HBM, scalars resident in HBM when the timed region starts.

  N = 1 (default)  BASELINE.json configs[1]: 2^20 pairs on one GPU.  The same JSON line also carries the other
                   headline-adjacent numbers, each cross-checked against the CPU oracle: `incl_h2d` (scalars in pageable
                   host memory), `trait_form` (the reference's slice-form signature over pageable
                   Montgomery-layout host arrays: the slice cache path), `anchor_2p24_single_gpu` (the N = 1 point of the
                   configs[2] curve), `prove_step_replay_ms` (configs[3] shapes incl. the six SpMVs) and
                   `hyperkzg_replay_ms` (configs[4]).  --no-extras skips them.
  N > 1 (default)  BASELINE.json configs[2]: a FIXED TOTAL of 2^24 pairs sharded contiguously over the N ranks
                   (2^24 / N per GPU, "scaling": "strong"): every rank runs the full single-GPU MSM on its shard of
                   the (scalar, base) array, then the 128-byte partial sums are all-gathered over RCCL and combined
                   (the reference's par_chunks + reduce, src/provider/msm.rs:564-574);
                   value = 2^24 x steps / max-over-ranks time; `combine_ms` is the exchange + point sum alone.
                   (--log2n X with N > 1 gives the old weak-scaling run, 2^X pairs per GPU.)

  python bench.py                      # 1 GPU, 2^20 pairs, prints ONE JSON line
  python bench.py --log2n 24 --no-extras   # the single-GPU anchor of the 2^24 curve on its own
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
DTYPE = "9x29-bit limbs (256-bit modular integer, Montgomery R = 2^261)"
VOP3_RATE_T = 30.0             # measured full-rate VOP3 issue of this chip, T lane-ops/s (same file: 29.4-36 by instruction)
MAD_RATE_T = 29.4              # measured v_mad_u64_u32 issue rate of this chip, T/s (profiles/r01_ubench_instruction_rates.jsonl)
# multiply-adds per XYZZ mixed addition (curve.hpp add_affine: 6 products, 2 squarings, 1 two-product sum with one
# reduction) by base field: BN254 Fq / Fr 6 x 162 + 2 x 126 + 243 = 1467 (the static count of the kernel agrees:
# profiles/r03_msm_2p20/accum_isa_hist.txt); the Pasta moduli have three zero limbs of nine, whose reduction terms are
# dropped at compile time: 6 x 135 + 2 x 99 + 216 = 1224
MADS_PER_MADD_BY_CURVE = {0: 1467, 1: 1467, 2: 1224, 3: 1224}
VALU_PEAK_T = 256 * 64 * 2.4e9 / 1e12   # 39.3 T lane-ops/s: 256 CUs x 64 lanes per clock x 2.4 GHz max clock (MI355X_MICROARCH.md)


def table_window_bits(n, args):
    """Window width of the tables of an n-point key: choose_c_precomp (nova_amd/csrc/msm_pipeline.hpp), restated for the
    operation counts below (the library owns the rule)."""
    if args.window_bits:
        return args.window_bits
    lg = max(n, 2).bit_length() - 1
    return 20 if lg >= 22 else 17 if lg >= 20 else 16 if lg >= 17 else 15 if lg >= 14 else 8


def madds_per_launch(n, args):
    """Mixed additions of one accumulate launch on uniformly random scalars: one per (pair, window)."""
    bits = {0: 254, 1: 254, 2: 255, 3: 255}[args.curve]
    return n * (-(-(bits + 1) // table_window_bits(n, args)))


PMC_R6 = os.path.join("profiles", "r06_msm", "pmc.json")   # scripts/gpu_r6_pmc.sh on the round-6 build
PMC_R6_KERNEL = "k_launch<AccumSegFn<0, 1> >"


def _pmc_accum():
    """(entry, source) of the accumulate kernel in the newest committed PMC summary."""
    try:
        return json.load(open(os.path.join(ROOT, PMC_R6)))[PMC_R6_KERNEL], PMC_R6
    except (OSError, KeyError, ValueError):
        pass
    for rnd in ("r04_msm_2p20", "r03_msm_2p20", "r02_msm_2p20", "r01_msm_2p20"):
        try:
            path = os.path.join("profiles", rnd, "pmc_traffic.json")
            return json.load(open(os.path.join(ROOT, path)))["accum"], path
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def pmc_valu(args, world):
    """VALU wave-instructions per accumulate launch from the committed PMC summary (same configuration rule as
    pmc_traffic): the basis of roofline.valu_issue.  Returns (instructions, effective clock GHz of the profiled pass, source)."""
    if not (world == 1 and args.curve == 0 and args.log2n == 20 and args.dist == "random" and not args.window_bits):
        return None
    d, src = _pmc_accum()
    if not d or "SQ_INSTS_VALU" not in d:
        return None
    clock = d.get("effective_clock_GHz")
    if clock is None and d.get("GRBM_GUI_ACTIVE") and d.get("us"):
        clock = d["GRBM_GUI_ACTIVE"] / 8 / (d["us"] * 1e3)      # cycles per XCD / ns
    return d["SQ_INSTS_VALU"], clock, src


def pmc_traffic(args, world):
    """HBM bytes per accumulate launch from the committed PMC summary -- only for the exact configuration it was
    collected on (BN254, 2^20, random scalars, one GPU); everything else reports null."""
    if not (world == 1 and args.curve == 0 and args.log2n == 20 and args.dist == "random" and not args.window_bits):
        return None
    d, _src = _pmc_accum()
    try:
        return d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]
    except (KeyError, TypeError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=None,
                    help="pairs per GPU = 2^log2n.  Default: 20 on one GPU (BASELINE configs[1]); with --gpus N > 1 the "
                         "default is the strong-scaling run of configs[2] instead (see --total-log2n)")
    ap.add_argument("--total-log2n", type=int, default=24,
                    help="N > 1: total pairs = 2^total_log2n split contiguously over the ranks (BASELINE configs[2]: 24)")
    ap.add_argument("--strong", action="store_true",
                    help="force the fixed-total (strong-scaling) sharding even at N = 1 (exercises the N > 1 code path on one GPU "
                         "together with NMX_BENCH_FORCE_DIST=1)")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1 headline run: skip incl_h2d / trait_form / 2^24 anchor / prove_step / HyperKZG replays")
    ap.add_argument("--curve", type=int, default=0, help="0 bn254_g1 (headline), 1 grumpkin, 2 pallas, 3 vesta")
    ap.add_argument("--dist", default="random", help="scalar distribution: random | u1 | u10 | u16 | u32 | u64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--no-tables", action="store_true",
                    help="register the key WITHOUT window tables: the plain path (W bucket sets) that first / second sight of a "
                         "cached array, IPA's per-round keys and keys whose tables do not fit take")
    ap.add_argument("--iters", type=int, default=65536, help="prove_step_replay: MinRoot iterations per step")
    ap.add_argument("--log2n-secondary", type=int, default=None, help="compressed_snark_replay: log2 of the secondary circuit's size (default min(14, log2n))")
    ap.add_argument("--cycle", default="bn254", choices=["bn254", "pasta"], help="prove_step replay: the curve cycle (BN254/Grumpkin or Pallas/Vesta)")
    ap.add_argument("--opens", default="batch", choices=["batch", "threads", "serial"],
                    help="hyperkzg replay: the three kzg_open commitments as one batch_commit (default), three host threads (the reference's par_iter), or one after the other")
    ap.add_argument("--separate-field-ops", action="store_true", help="prove_step replay: vec_add, 3 x SpMV, cross term and the two folds as separate (stream-ordered) calls")
    ap.add_argument("--overlap-commits", type=int, default=0, choices=[0, 1, 2],
                    help="prove_step replay: commit(W) begun (nmx_commit_begin) beside the cross term + commit(T) it does not depend on "
                         "(1: the primary pair inside one prove_step; 2: also the secondary pair, across the step boundary)")
    ap.add_argument("--serial-snarks", action="store_true", help="compressed_snark_replay: S1::prove and S2::prove one after the other instead of side by side (rayon::join)")
    ap.add_argument("--separate-folds", action="store_true", help="hyperkzg replay: the ell - 1 pair folds as ell - 1 calls instead of nmx_poly_fold_chain")
    ap.add_argument("--separate-spmv", action="store_true", help="spartan replay: the three (transposed) products as three calls instead of nmx_spmv_apply_many")
    ap.add_argument("--sync-field-ops", action="store_true", help="prove_step replay: every field-vector call waits for its kernel (round 3's form)")
    ap.add_argument("--workload", default="msm", choices=["msm", "axpy", "cross_term", "bind", "sumcheck3", "round3", "quad_prod", "lincomb8", "horner", "mle_eval", "spmv", "prove_step_replay", "hyperkzg_replay", "spartan_replay", "compressed_snark_replay", "ipa_replay"],
                    help="msm = the headline (default); the others time one HBM-bound field-vector kernel of "
                         "SURVEY.md 8(f) at 2^log2n elements per GPU")
    args = ap.parse_args()
    strong = (args.gpus > 1 and args.log2n is None) or args.strong      # configs[2] as written: fixed total, contiguous shards
    args.inproc_log2n = args.total_log2n if strong else args.log2n     # in-process mode (below): pairs in total
    if args.log2n is None:
        args.log2n = 20

    # RCCL writes its version banner / warnings to stdout; stdout must end with ONE JSON line: send RCCL's log to
    # stderr and (see emit()) print the JSON only after the process group is gone and C stdio is flushed.
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: ONE process drives the N GPUs through the C ABI's multi-device
        # keys (nmx_init_devices) -- the way a Rust host process would (one address space)
        return emit(inprocess_multi(args, torch), False, dist)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run, or " \
                               f"without a launcher for the in-process mode)"
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
        assert world == 1, "prove_step is sequential: replicas only"
        return emit(prove_step_replay(args, torch), False, dist)
    if args.workload == "hyperkzg_replay":
        assert world == 1
        return emit(hyperkzg_replay(args, torch), False, dist)
    if args.workload == "spartan_replay":
        assert world == 1
        return emit(spartan_replay(args, torch), False, dist)
    if args.workload == "compressed_snark_replay":
        assert world == 1
        return emit(compressed_snark_replay(args, torch), False, dist)
    if args.workload == "ipa_replay":
        assert world == 1
        return emit(ipa_replay(args, torch), False, dist)
    if args.workload != "msm":
        return field_workload(args, world, rank, L, torch, dist)

    from nova_amd.dist import shard_range, sharded_msm
    cid = args.curve
    if strong:
        total = 1 << args.total_log2n
        lo, hi = shard_range(total, rank, world)
        n, k0 = hi - lo, 1 + lo            # this rank's contiguous shard of ONE key: P_i = (1 + i) * G, i in [lo, hi)
    else:
        n = 1 << args.log2n
        total = n * world
        k0 = 1 + rank * n
    ce = nova_amd.CommitmentEngine(cid)
    group = ce.group
    ck = nova_amd.CommitmentKey.generate(cid, n, k0=k0, precompute=not args.no_tables)
    # two scalar vectors per rank, alternated between steps, resident in HBM before timing starts
    host_sc = [util.scalar_set(cid, n, args.dist, seed=util.SEED + 1000 * rank + j) for j in range(2)]
    dev_sc = [torch.from_numpy(s.copy()).cuda() for s in host_sc]
    multi = world > 1 or force_dist
    combine_s = [0.0]

    def step(j):
        if not multi:
            return group.vartime_multiscalar_mul(dev_sc[j & 1], ck)
        return sharded_msm(group, ck, dev_sc[j & 1], timing=combine_s)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for j in range(args.warmup):
        step(j)
    combine_s[0] = 0.0
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
        pairs = total * args.steps
        achieved = BYTES_PER_PAIR * n / (accum_ms * 1e-3) / 1e9 if accum_ms > 0 else 0.0
        name = nova_amd.CURVE_NAMES[cid]
        if strong:
            workload = (f"{name} Pippenger MSM, 2^{args.total_log2n} pairs in total sharded contiguously over {world} GPUs "
                        f"({n} pairs on rank 0), {args.dist} scalars, bases+scalars resident in HBM, 128-byte partials "
                        "all-gathered over RCCL and summed (BASELINE.json configs[2])")
        else:
            workload = (f"{name} Pippenger MSM 2^{args.log2n} pairs per GPU, {args.dist} scalars, bases+scalars resident in "
                        f"HBM (BASELINE.json configs[{1 if args.log2n == 20 and world == 1 else 2}])")
        out = {
            "metric": "BN254 MSM scalar-point pairs/sec" if cid == 0 else f"{name} MSM scalar-point pairs/sec",
            "value": pairs / dt,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": DTYPE,
            "data": "synthetic",
            "config": {
                "workload": workload,
                "pairs_total": total,
                "pairs_per_gpu": n,
                "parallelism": f"shard{world}" if world > 1 else "single",
                "combine": "rccl all_gather of 128-byte partials + host point sum" if multi else "none",
            },
            "stages_ms": {s: round(float(v), 4) for s, v in zip(STAGES, stage_ms)},
            # median and min of the timed calls (rank 0's own calls; `value` uses the whole region)
            "per_call_ms": {"median": round(float(np.median(per_call)) * 1e3, 4), "min": round(min(per_call) * 1e3, 4)},
            "roofline": {
                "bound": "hbm",
                # msm_seg.hpp from 2^22 sorted entries on (table path, c <= 20)
                "kernel": "k_launch<AccumSegFn> (bucket accumulation, segment-balanced)"
                          if madds_per_launch(n, args) >= (1 << 22) and table_window_bits(n, args) <= 20
                          else "k_launch<AccumFn> (bucket accumulation)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args, world),
                "note": "algorithmic bytes = 96 B/pair x pairs per launch / accum-kernel time (hipEvents on the "
                        "library stream); the MSM is integer-VALU-bound, not HBM-bound (DESIGN.md). traffic = HBM bytes "
                        "per launch from the separate rocprofv3 --pmc passes in profiles/ (null for other configs).",
                # the multiply-adds alone (1 467 of the ~2 150 VALU instructions of a mixed addition) against the rate the chip
                # issues v_mad_u64_u32 at when it issues nothing else (measured, profiles/r01_ubench_instruction_rates.jsonl):
                # the share of that ceiling the kernel's USEFUL arithmetic reaches.  The ceiling of the whole instruction stream
                # (lanes x clock) is `valu_issue` below.
                "valu_mad": {"achieved_T_per_s": MADS_PER_MADD * madds_per_launch(n, args) / (accum_ms * 1e-3) / 1e12 if accum_ms > 0 else 0.0,
                             "peak_T_per_s": MAD_RATE_T,
                             "peak_is": "measured v_mad_u64_u32-only issue rate (micro-benchmark), not lanes x clock",
                             "frac": (MADS_PER_MADD * madds_per_launch(n, args) / (accum_ms * 1e-3) / 1e12) / MAD_RATE_T if accum_ms > 0 else 0.0,
                             "mads_per_mixed_add": MADS_PER_MADD},
            },
        }
        pv = pmc_valu(args, world)
        if pv and accum_ms > 0:
            # What binds the kernel: VALU instruction issue.  A mixed addition is ~2 230 VALU instructions
            # (SQ_INSTS_VALU / wave-additions), 1 467 of them multiply-adds.  Ceiling = lanes x clock from the hardware guide
            # (every VOP3 instruction at full rate, 2.4 GHz): no instruction mix can exceed it, so frac <= 1 by construction.
            # Under this kernel the chip clocks to its power budget (effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time of
            # the profiled pass, ~2.06 GHz): at THAT clock the issue ports are ~99 % busy.
            insts, gui, pmc_src = pv
            lane_ops = insts * 64
            ach = lane_ops / (accum_ms * 1e-3) / 1e12
            out["roofline"]["valu_issue"] = {
                "valu_insts_per_mixed_add": round(insts / (madds_per_launch(n, args) / 64), 1),
                "achieved_T_lane_ops_per_s": ach,
                "peak_T_lane_ops_per_s": VALU_PEAK_T,
                "frac": ach / VALU_PEAK_T,
                "source": f"{pmc_src} (rocprofv3 --pmc SQ_INSTS_VALU, separate pass) / accum-kernel time of this run",
            }
            if gui:   # GRBM_GUI_ACTIVE / 8 XCDs / the kernel's duration in the same counter pass
                out["roofline"]["valu_issue"]["effective_clock_GHz_profiled_pass"] = round(gui, 2)
        tr = out["roofline"]["traffic"]
        if tr:
            out["roofline"]["traffic_ratio"] = round(tr / (BYTES_PER_PAIR * n), 1)   # HBM bytes moved / algorithmic bytes
        st = _lib.stats()
        out["launch_gap_us"] = round(st[_lib.STAT_LAUNCH_GAP_NS] / 1e3, 2)   # one dependent one-wave launch on this box
        out["reduction_tree"] = "one launch per level (NMX_TUNE_NO_TREE_FUSE=1)" if os.environ.get("NMX_TUNE_NO_TREE_FUSE") == "1" else "fused levels"
        if multi:
            out["combine_ms"] = round(combine_s[0] / max(args.steps, 1) * 1e3, 4)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cid, ck, host_sc[(args.steps - 1) & 1], n, result)
        if world == 1 and not force_dist and not args.no_extras and args.log2n == 20 and args.dist == "random" and cid == 0:
            extras(out, args, torch, L, ck, host_sc[0], dev_sc[0])
    ck.close()
    emit(out if rank == 0 else None, multi, dist)


def inprocess_multi(args, torch):
    """BASELINE.json configs[2] from ONE process (what `python bench.py --gpus N` without a launcher runs): a key of
    2^total_log2n points sharded over the GPUs by the library itself (nmx_init_devices: shard i and its window tables resident
    on GPU i) and the scalars SHARD-RESIDENT -- piece i drawn on GPU i, handed over as NMX_SCALARS_SHARDED, the way the reference
    chunks coefficients and bases together (src/provider/msm.rs:564-574).  Every step is one synchronous nmx_msm_handle call
    that fans out a host thread + stream per GPU; the 128-byte partials are all-gathered over RCCL (one rank per GPU) and
    summed.  One result is compared with the CPU oracle at FULL size.  Fewer GPUs visible than asked for: says so on stderr
    and in the JSON, and runs on what there is."""
    import nova_amd
    from nova_amd import _lib
    L = _lib.lib()
    assert L.nmx_init(0) == 0, L.nmx_last_error().decode()
    visible = L.nmx_device_count()
    oversub = os.environ.get("NMX_BENCH_OVERSUB") == "1"   # builder's 1-GPU boxes: N logical devices on the GPUs there are
    k = args.gpus if oversub else min(args.gpus, visible)
    if k < args.gpus:
        print(f"bench.py: WARNING --gpus {args.gpus} requested but only {visible} HIP device(s) visible: running the "
              f"in-process sharded MSM on {k} device(s)", file=sys.stderr, flush=True)
    assert nova_amd.init_devices(k, oversubscribe=oversub) == k, L.nmx_last_error().decode()
    if oversub:
        assert L.nmx_set_option(b"shard_min_n", 1024) == 0
    if os.environ.get("NMX_BENCH_COMBINE"):           # 1 = host sum, 2 = RCCL required (also with one GPU)
        assert L.nmx_set_option(b"combine", int(os.environ["NMX_BENCH_COMBINE"])) == 0
    cid = args.curve
    total = 1 << args.inproc_log2n
    torch.cuda.set_device(0)
    t0 = time.perf_counter()
    ck = nova_amd.CommitmentKey.generate(cid, total, k0=1)   # P_i = (1 + i) G, cut into k contiguous shards
    t_key = time.perf_counter() - t0
    group = nova_amd.DlogGroup(cid)
    # [(logical device, offset in the shard, count)] from the layout the REGISTERED key has: the generated key carries its
    # blinding point behind ck (total + 1 points), so the cut is not at total / k
    plan = ck.shard_plan(0, total)
    assert args.dist == "random", "the in-process mode draws its scalars on the devices: --dist random only"

    def draw(dev, cnt, seed):
        """cnt uniformly random 253-bit scalars (all below the 254 / 255-bit moduli), drawn ON GPU `dev`."""
        dev = dev % visible
        g = torch.Generator(device=f"cuda:{dev}")
        g.manual_seed(seed)
        w = torch.randint(0, 1 << 31, (cnt, 8), dtype=torch.int64, device=f"cuda:{dev}", generator=g)
        w = (w * 2 + torch.randint(0, 2, (cnt, 8), dtype=torch.int64, device=f"cuda:{dev}", generator=g)).to(torch.int32)
        w[:, 7] &= 0x1FFFFFFF
        return w.view(torch.uint8).reshape(cnt, 32).contiguous()

    # two scalar vectors, alternated between steps; piece i of each lives in the HBM of GPU i (logical device i = HIP device i)
    sets = [[draw(dev, cnt, 1000 * j + dev + 1) for dev, _off, cnt in plan] for j in range(2)]
    ndev_sync = min(k, visible)
    for d in range(ndev_sync):
        torch.cuda.synchronize(d)
    for j in range(args.warmup):
        group.vartime_multiscalar_mul(sets[j & 1], ck)
    before = _lib.stats()[_lib.STAT_SHARDED_CALLS]
    L.nmx_set_profiling(1)
    import ctypes
    prof = (ctypes.c_float * 16)()
    shard_sum, combine_sum, ranks, branches = None, 0.0, 0, None
    stage_sum = np.zeros(len(STAGES))
    for d in range(ndev_sync):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    res = None
    per_call = []
    for j in range(args.steps):
        tj = time.perf_counter()
        res = group.vartime_multiscalar_mul(sets[j & 1], ck)   # synchronous: returns the affine point
        per_call.append(time.perf_counter() - tj)
        nst = L.nmx_profile_last(prof, 16)
        stage_sum[:min(nst, len(STAGES))] += np.array(prof[:min(nst, len(STAGES))])
        if k > 1:
            rec = _lib.profile_last_sharded()
            m = np.array([s["stages_ms"] for s in rec["shards"]])
            shard_sum = m if shard_sum is None else shard_sum + m
            combine_sum += rec["combine_ms"]
            ranks, branches = rec["rccl_ranks"], [s["branch"] for s in rec["shards"]]
    for d in range(ndev_sync):
        torch.cuda.synchronize(d)
    dt = time.perf_counter() - t0
    L.nmx_set_profiling(0)
    sharded_calls = _lib.stats()[_lib.STAT_SHARDED_CALLS] - before
    last = sets[(args.steps - 1) & 1]
    name = nova_amd.CURVE_NAMES[cid]
    steps = max(args.steps, 1)
    out = {
        "metric": "BN254 MSM scalar-point pairs/sec" if cid == 0 else f"{name} MSM scalar-point pairs/sec",
        "value": total * args.steps / dt,
        "unit": "pairs/s",
        "n_gpus": k,
        "requested_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": DTYPE,
        "data": "synthetic",
        "config": {
            "workload": f"{name} Pippenger MSM, 2^{total.bit_length() - 1} pairs of one key sharded contiguously over {k} "
                        f"GPU(s) by ONE host process (nmx_init_devices; BASELINE.json configs[2]), 2^{total.bit_length() - 1} "
                        "distinct uniformly random 253-bit scalars SHARD-RESIDENT (piece i drawn in the HBM of GPU i, "
                        "NMX_SCALARS_SHARDED: nothing crosses xGMI or PCIe inside the call), one 128-byte partial per shard, "
                        "all-gathered over RCCL and summed",
            "pairs_total": total,
            "pairs_per_gpu": total // k,
            "parallelism": f"in-process shard{k}" if k > 1 else "single (in-process mode, one device visible)",
            "combine": ("rccl all_gather of 128-byte partials (one rank per GPU) + host point sum" if ranks
                        else "host sum of 128-byte partials (nmx_point_sum)") if k > 1 else "none",
        },
        "per_call_ms": {"median": round(float(np.median(per_call)) * 1e3, 4), "min": round(min(per_call) * 1e3, 4)},
        "stages_ms": {s: round(float(v) / steps, 4) for s, v in zip(STAGES, stage_sum)},
        "rccl_ranks": ranks,
        "combine_ms": round(combine_sum / steps, 4),
        "sharded_calls": sharded_calls,
        "key_generation_s": round(t_key, 3),
        "last_result_is_inf": bool(res.is_inf),
    }
    if k > 1 and shard_sum is not None:
        out["stages_ms"]["_what"] = "per stage, the slowest shard (the call waits for all of them)"
        out["shards"] = [{"gpu": plan[i][0], "pairs": plan[i][2], "scalars": branches[i],
                          "stages_ms": {s: round(float(v) / steps, 4) for s, v in zip(STAGES, shard_sum[i])}}
                         for i in range(len(plan))]
    accum_ms = out["stages_ms"]["accum"]
    per_gpu = max(c for _d, _o, c in plan)
    if accum_ms > 0:
        ach = BYTES_PER_PAIR * per_gpu / (accum_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_launch<AccumSegFn> (bucket accumulation, segment-balanced), slowest shard",
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                           "note": "96 B/pair x pairs of one shard / that shard's accumulate-kernel time (hipEvents on the library "
                                   "stream); the MSM is integer-VALU-bound, not HBM-bound (DESIGN.md section 5)"}
    # contrast: the same call with ONE HBM array on GPU 0 (NMX_SCALARS_DEVICE), every other shard pulling its slice over xGMI
    # inside the call -- what round 3 timed
    if k > 1 and not args.no_extras:
        try:
            one = torch.cat([p.to("cuda:0") for p in last]).contiguous()
            torch.cuda.synchronize(0)
            group.vartime_multiscalar_mul(one, ck)
            ts = []
            for _ in range(3):
                tj = time.perf_counter()
                r1 = group.vartime_multiscalar_mul(one, ck)
                ts.append(time.perf_counter() - tj)
            out["scalars_on_gpu0"] = {"ms": round(float(np.median(ts)) * 1e3, 4), "matches": r1 == res,
                                      "branches": [s["branch"] for s in _lib.profile_last_sharded()["shards"]],
                                      "what": "same pairs, scalars in ONE array on GPU 0: every other shard pulls its slice peer-to-peer inside the call"}
            del one
        except (RuntimeError, nova_amd.NmxError) as e:
            out["scalars_on_gpu0"] = {"error": str(e)}
    # the CPU oracle at FULL size on the last timed input: the baseline and the bit-exact check in one pass
    if not args.no_cpu_baseline:
        from oracle import cref
        threads = effective_cpus()
        cref.set_threads(threads)
        host_sc = np.concatenate([p.cpu().numpy() for p in last])
        host_b = ck.read(0, total)
        t1 = time.perf_counter()
        exp = cref.msm(cid, host_sc, host_b, total)
        t_cpu = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": total / t_cpu, "unit": "pairs/s", "cores": threads, "kind": "port",
                               "sample": f"the whole workload once (2^{total.bit_length() - 1} pairs), oracle/nova_ref.c (C restatement of "
                                         "msm.rs + Pippenger in the msm_best role; the Rust reference cannot be built here)",
                               "seconds": round(t_cpu, 3), "gpu_matches_cpu": (res.xy, int(res.is_inf)) == exp}
        del host_sc, host_b
    if k < args.gpus:
        out["fallback"] = f"{args.gpus} GPUs requested, {visible} visible"
    if oversub:
        out["oversubscribed"] = f"{k} logical devices on {visible} GPU(s) (NMX_BENCH_OVERSUB=1): a functional run, not a scaling point"
    ck.close()
    return out


def fieldvec_block(args, torch, L):
    """The HBM-bound field-vector kernels either side of the MSM, one roofline entry each for the default
    N = 1 line: BN254 scalar field, vectors resident in HBM, kernel time from hipEvents on the library's stream
    (nmx_set_profiling), achieved = algorithmic bytes / kernel time, frac of the 8 TB/s HBM peak, and a cross-check of the GPU
    output against the CPU oracle (element-wise kernels: the first 2^20 elements; reductions: the whole input).  Inputs are
    drawn on the device (uniform 253-bit integers, all below the modulus)."""
    import ctypes
    from nova_amd import fieldvec as fv
    from oracle import cref
    cid, fid = 0, fv.BN254_FR
    cref.set_threads(effective_cpus())
    g = torch.Generator(device="cuda")
    g.manual_seed(20260924)

    def rand_vec(n):
        w = torch.randint(0, 1 << 31, (n, 8), dtype=torch.int64, device="cuda", generator=g)
        w = (w * 2 + torch.randint(0, 2, (n, 8), dtype=torch.int64, device="cuda", generator=g)).to(torch.int32)
        w[:, 7] &= 0x1FFFFFFF                                  # < 2^253 < r
        return w.view(torch.uint8).reshape(n, 32).contiguous()

    N24, N22, N20 = 1 << 24, 1 << 22, 1 << 20
    pool = [rand_vec(N24) for _ in range(5)]
    torch.cuda.synchronize()                                   # the library runs on its own (non-blocking) streams
    r = np.frombuffer((0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF12345678 % util_modulus(fid)).to_bytes(32, "little"),
                      dtype=np.uint8).reshape(1, 32).copy()
    host_cache = {}

    def host(j, n):                                            # host copy of the first n elements of pool[j]
        if (j, n) not in host_cache:
            host_cache[(j, n)] = pool[j][:n].cpu().numpy()
        return host_cache[(j, n)]

    prof = (ctypes.c_float * 4)()

    def kernel_ms(fn, steps=5, warm=2, pre=None):
        for _ in range(warm):
            if pre:
                pre()
            fn()
        L.nmx_set_profiling(1)
        tot, out = 0.0, None
        for _ in range(steps):
            if pre:
                pre()
            out = fn()
            L.nmx_profile_last(prof, 4)
            tot += prof[0]
        L.nmx_set_profiling(0)
        return tot / steps, out

    def as_bytes(o, m=None):
        if isinstance(o, tuple):
            return b"".join(o)
        if isinstance(o, bytes):
            return o
        return (o if m is None else o[:m]).cpu().numpy().reshape(-1).tobytes()

    res = {}

    def entry(name, n, bytes_per_elem, ms, ok, note=None):
        ach = bytes_per_elem * n / (ms * 1e-3) / 1e9
        res[name] = {"log2n": n.bit_length() - 1, "kernel_ms": round(ms, 4), "bytes_per_elem": bytes_per_elem,
                     "achieved_GBs": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "gpu_matches_cpu": bool(ok)}
        if note:
            res[name]["note"] = note

    m = N20                                                    # element-wise kernels: oracle on the first 2^20 elements
    A, B, C, D, E = pool
    ms, o = kernel_ms(lambda: fv.axpy(fid, A, B, r))
    entry("axpy", N24, 96, ms, as_bytes(o, m) == cref.field_axpy(fid, host(0, m), host(1, m), r, m))
    ms, o = kernel_ms(lambda: fv.axpy2(fid, A, B, C, r))
    entry("axpy2", N24, 128, ms, as_bytes(o, m) == cref.field_axpy2(fid, host(0, m), host(1, m), host(2, m), r, m))
    ms, o = kernel_ms(lambda: fv.cross_term(fid, A, B, C, D, r))
    entry("cross_term", N24, 160, ms, as_bytes(o, m) == cref.field_cross_term(fid, host(0, m), host(1, m), host(2, m), host(3, m), r, m))
    ms, o = kernel_ms(lambda: fv.cross_term2(fid, A, B, C, D, E, r))
    entry("cross_term2", N24, 192, ms,
          as_bytes(o, m) == cref.field_cross_term2(fid, host(0, m), host(1, m), host(2, m), host(3, m), host(4, m), r, m))
    ms, o = kernel_ms(lambda: fv.bind_poly_var_top(fid, A, r))
    hA = A.cpu().numpy()
    entry("bind", N24, 48, ms, as_bytes(o, m) == cref.field_bind(fid, hA, 0, N24 // 2, 1, r, m))
    # reductions: whole input through the oracle
    hB, hC = B.cpu().numpy(), C.cpu().numpy()
    shift = (24 - 1) // 2
    eqR, eqL = rand_vec(1 << shift), rand_vec((N24 // 2) >> shift)
    torch.cuda.synchronize()
    ms, o = kernel_ms(lambda: fv.sumcheck_eq_sums(fid, 3, A, B, C, eqR, eqL, shift))
    entry("sumcheck3", N24, 80, ms,
          as_bytes(o) == b"".join(cref.sumcheck_eq_sums(fid, 3, hA, hB, hC, N24, eqR.cpu().numpy(), eqL.cpu().numpy(), shift)),
          "multiplier-bound: 4 + 2/K field reductions per index")
    ms, o = kernel_ms(lambda: fv.sumcheck_plain_sums(fid, 1, A, B))
    entry("quad_prod", N24, 64, ms, as_bytes(o)[:64] == b"".join(cref.sumcheck_plain_sums(fid, 1, hA, hB, None, N24)[:2]))
    shift3 = (24 - 2) // 2
    eqR3, eqL3 = rand_vec(1 << shift3), rand_vec((N24 // 4) >> shift3)
    torch.cuda.synchronize()
    work = [torch.empty_like(t) for t in (A, B, C)]

    def refresh():
        for w, t in zip(work, (A, B, C)):
            w.copy_(t)
        torch.cuda.synchronize()
    ms, o = kernel_ms(lambda: fv.sumcheck_bind_eq_sums(fid, 3, work[0], work[1], work[2], r, eqR3, eqL3, shift3)[3], pre=refresh)
    bound = [cref.field_bind(fid, h, 0, N24 // 2, 1, r, N24 // 2) for h in (hA, hB, hC)]
    entry("round3", N24, 144, ms,
          as_bytes(o) == b"".join(cref.sumcheck_eq_sums(fid, 3, bound[0], bound[1], bound[2], N24 // 2, eqR3.cpu().numpy(),
                                                        eqL3.cpu().numpy(), shift3)),
          "one whole cubic round: bind A, B, C + the next round's sums")
    del work, bound
    point = rand_vec(24).cpu().numpy()
    ms, o = kernel_ms(lambda: fv.mle_evaluate(fid, A, point))
    entry("mle_eval", N24, 32, ms, as_bytes(o) == cref.mle_evaluate(fid, hA, 24, point))
    ms, o = kernel_ms(lambda: fv.mle_evaluate(fid, A[:N20], point[:20]))
    entry("mle_eval_2p20", N20, 32, ms, as_bytes(o) == cref.mle_evaluate(fid, hA[:N20], 20, point[:20]), "launch floor: one launch for both eq tables, then the pass and its one-block final sum")
    # 2^22: lincomb of 8, suffix Horner, SpMV
    vecs = [p[j * N22:(j + 1) * N22] for p in (A, B) for j in range(4)]
    ms, o = kernel_ms(lambda: fv.lincomb_powers(fid, vecs, r))
    entry("lincomb8", N22, 288, ms, as_bytes(o, m) == cref.lincomb_powers(fid, [v[:m].cpu().numpy().tobytes() for v in vecs], r, m))
    ms, o = kernel_ms(lambda: fv.suffix_horner(fid, A[:N22], r))
    entry("horner", N22, 64, ms, as_bytes(o) == cref.suffix_horner(fid, hA[:N22], N22, r),
          "single-pass scan with decoupled look-back: 670 VALU instructions per coefficient (valu_floor_ms) and 40 % of wave cycles waiting on its look-back round trips (profiles/r03_fieldvec/horner_scan.txt, profiles/r04_fieldvec/horner_22_pmc.json)")
    rng = np.random.Generator(np.random.PCG64(5))
    indptr = np.arange(0, 3 * N22 + 1, 3, dtype=np.uint64)
    indices = rng.integers(0, N22, size=3 * N22).astype(np.uint64)
    data = hB[:3 * N22].copy()
    kind = rng.random(3 * N22)
    one = np.zeros(32, np.uint8)
    one[0] = 1
    data[kind < 0.6] = one
    data[(kind >= 0.6) & (kind < 0.9)] = np.frombuffer((util_modulus(fid) - 1).to_bytes(32, "little"), dtype=np.uint8)
    mat = fv.SparseMatrix(fid, indptr, indices, data, N22)
    ms, o = kernel_ms(lambda: mat.multiply_vec(A[:N22]))
    entry("spmv", N22, 3 * 68 + 40, ms, as_bytes(o, m) == cref.spmv(fid, indptr[: m + 1], indices, data, m, hA[:N22]),
          "CSR, 3 non-zeros per row, 9 of 10 coefficients +-1 (R1CS-like)")
    mat.close()
    # The same kernels at 2^20 elements -- the size BASELINE.json configs[4] runs them at: a few microseconds of
    # HBM time each, so what these entries show is the launch floor (kernel_ms is the hipEvent bracket of ONE call's launches)
    a20, b20, c20, d20, e20 = (p_[:N20] for p_ in pool)
    h20 = [host(j, N20) for j in range(5)]
    main = res
    res = {}
    ms, o = kernel_ms(lambda: fv.axpy(fid, a20, b20, r))
    entry("axpy", N20, 96, ms, as_bytes(o) == cref.field_axpy(fid, h20[0], h20[1], r, N20))
    ms, o = kernel_ms(lambda: fv.axpy2(fid, a20, b20, c20, r))
    entry("axpy2", N20, 128, ms, as_bytes(o) == cref.field_axpy2(fid, h20[0], h20[1], h20[2], r, N20))
    ms, o = kernel_ms(lambda: fv.cross_term(fid, a20, b20, c20, d20, r))
    entry("cross_term", N20, 160, ms, as_bytes(o) == cref.field_cross_term(fid, h20[0], h20[1], h20[2], h20[3], r, N20))
    ms, o = kernel_ms(lambda: fv.cross_term2(fid, a20, b20, c20, d20, e20, r))
    entry("cross_term2", N20, 192, ms, as_bytes(o) == cref.field_cross_term2(fid, h20[0], h20[1], h20[2], h20[3], h20[4], r, N20))
    ms, o = kernel_ms(lambda: fv.bind_poly_var_top(fid, a20, r))
    entry("bind", N20, 48, ms, as_bytes(o) == cref.field_bind(fid, h20[0], 0, N20 // 2, 1, r, N20 // 2))
    sh20 = (20 - 1) // 2
    eqR20, eqL20 = rand_vec(1 << sh20), rand_vec((N20 // 2) >> sh20)
    torch.cuda.synchronize()
    ms, o = kernel_ms(lambda: fv.sumcheck_eq_sums(fid, 3, a20, b20, c20, eqR20, eqL20, sh20))
    entry("sumcheck3", N20, 80, ms, as_bytes(o) == b"".join(cref.sumcheck_eq_sums(fid, 3, h20[0], h20[1], h20[2], N20, eqR20.cpu().numpy(),
                                                                                    eqL20.cpu().numpy(), sh20)))
    ms, o = kernel_ms(lambda: fv.sumcheck_plain_sums(fid, 1, a20, b20))
    entry("quad_prod", N20, 64, ms, as_bytes(o)[:64] == b"".join(cref.sumcheck_plain_sums(fid, 1, h20[0], h20[1], None, N20)[:2]))
    sh203 = (20 - 2) // 2
    eqR203, eqL203 = rand_vec(1 << sh203), rand_vec((N20 // 4) >> sh203)
    work = [torch.empty_like(t) for t in (a20, b20, c20)]

    def refresh20():
        for w, t in zip(work, (a20, b20, c20)):
            w.copy_(t)
        torch.cuda.synchronize()
    ms, o = kernel_ms(lambda: fv.sumcheck_bind_eq_sums(fid, 3, work[0], work[1], work[2], r, eqR203, eqL203, sh203)[3], pre=refresh20)
    bound = [cref.field_bind(fid, h, 0, N20 // 2, 1, r, N20 // 2) for h in h20[:3]]
    entry("round3", N20, 144, ms, as_bytes(o) == b"".join(cref.sumcheck_eq_sums(fid, 3, bound[0], bound[1], bound[2], N20 // 2,
                                                                                 eqR203.cpu().numpy(), eqL203.cpu().numpy(), sh203)))
    del work, bound
    vecs20 = [p_[j * N20:(j + 1) * N20] for p_ in (A, B) for j in range(4)]
    ms, o = kernel_ms(lambda: fv.lincomb_powers(fid, vecs20, r))
    entry("lincomb8", N20, 288, ms, as_bytes(o) == cref.lincomb_powers(fid, [v.cpu().numpy().tobytes() for v in vecs20], r, N20))
    ms, o = kernel_ms(lambda: fv.suffix_horner(fid, a20, r))
    entry("horner", N20, 64, ms, as_bytes(o) == cref.suffix_horner(fid, h20[0], N20, r))
    ms, o = kernel_ms(lambda: fv.mle_evaluate(fid, a20, point[:20]))
    entry("mle_eval", N20, 32, ms, as_bytes(o) == cref.mle_evaluate(fid, h20[0], 20, point[:20]))
    ip20 = np.arange(0, 3 * N20 + 1, 3, dtype=np.uint64)
    ix20 = rng.integers(0, N20, size=3 * N20).astype(np.uint64)
    mat = fv.SparseMatrix(fid, ip20, ix20, data[:3 * N20], N20)
    ms, o = kernel_ms(lambda: mat.multiply_vec(a20))
    entry("spmv", N20, 3 * 68 + 40, ms, as_bytes(o) == cref.spmv(fid, ip20, ix20, data[:3 * N20], N20, h20[0]))
    ms, o = kernel_ms(lambda: mat.multiply_vec_transposed(a20))
    entry("spmv_transposed", N20, 3 * 68 + 40, ms, as_bytes(o) == cref.spmv_transposed(fid, ip20, ix20, data[:3 * N20], N20, N20, h20[0]),
          "compute_eval_table_sparse's product (src/spartan/mod.rs:497-533) over the same matrix")
    mat.close()
    at20 = res
    at20["_min_frac"] = min(v["frac"] for v in at20.values() if isinstance(v, dict))
    res = main
    # The multiplier ceiling of each kernel, from the counters: VALU wave-instructions per launch (rocprofv3 --pmc SQ_INSTS_VALU,
    # a separate pass, committed under profiles/) x 64 lanes / the chip's measured VOP3 issue rate = the time the arithmetic
    # alone takes.  A kernel whose valu_floor_ms is close to kernel_ms is multiplier-bound: on a slower-clocked lease its HBM
    # fraction drops with the clock and no memory-side change can lift it.
    try:
        floors = json.load(open(os.path.join(ROOT, "profiles", "r04_fieldvec", "valu_insts.json")))
    except (OSError, ValueError):
        floors = {}
    for name, e in res.items():
        f = floors.get(name)
        if isinstance(e, dict) and f and f.get("log2n") == e["log2n"]:
            e["valu_floor_ms"] = round(f["SQ_INSTS_VALU"] * 64 / VOP3_RATE_T / 1e12 * 1e3, 4)
            e["valu_floor_share"] = round(e["valu_floor_ms"] / e["kernel_ms"], 3) if e["kernel_ms"] > 0 else None
    res["_valu_floor"] = ("valu_floor_ms = SQ_INSTS_VALU per launch (profiles/r04_fieldvec/valu_insts.json, rocprofv3 --pmc, separate pass) "
                          f"x 64 / {VOP3_RATE_T} T lane-ops/s (measured VOP3 issue rate, profiles/r01_ubench_instruction_rates.jsonl)")
    res["_what"] = ("bn254_fr vectors resident in HBM; kernel_ms = hipEvents on the library stream, mean of 5 launches; frac = "
                    "algorithmic bytes / kernel time / 8000 GB/s; gpu_matches_cpu = oracle/nova_ref.c on the same inputs")
    res["_min_frac"] = min(v["frac"] for k, v in res.items() if isinstance(v, dict))
    res["at_2p20"] = at20
    return res


def extras(out, args, torch, L, ck, host_scalars, dev_scalars):
    """The other headline-adjacent numbers of the N = 1 run, in the same JSON line (each with its own cross-check
    against the CPU oracle).  BN254, 2^20 unless stated."""
    import nova_amd
    from nova_amd import _lib, fieldvec as fv
    from oracle import cref
    from tests import util
    cid, n = 0, len(host_scalars)
    g = nova_amd.DlogGroup(cid)
    K = 5

    def med_ms(fn, reps=K, warm=1):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3, r

    ref = g.vartime_multiscalar_mul(dev_scalars, ck)       # HBM-resident result of the same inputs (checked by cpu_baseline)
    # (1) SURVEY.md 8(d) "incl. H2D of scalars": canonical scalars in pageable host memory, key resident (handle form)
    ms, r = med_ms(lambda: g.vartime_multiscalar_mul(host_scalars, ck))
    out["incl_h2d"] = {"ms": round(ms, 4), "value": n / (ms * 1e-3), "unit": "pairs/s", "scalars": "pageable host memory, canonical",
                       "matches_hbm_resident_result": r == ref}
    # (2) the reference's own signature: vartime_multiscalar_mul(&[Scalar], &ck.ck[..n]) = nmx_msm(slice) over host arrays
    # in halo2curves' in-memory layout (Montgomery limbs, NMX_SCALARS_MONT | NMX_BASES_MONT), bases through the slice cache
    host_bases = ck.read(0, n)
    one256 = {fid: ((1 << 256) % util_modulus(fid)).to_bytes(32, "little") for fid in (fv.BN254_FQ, fv.BN254_FR)}

    def to_mont(fid, canon_rows):                           # x -> x * 2^256 mod p on the GPU: 0 + R * x
        d = torch.from_numpy(np.ascontiguousarray(canon_rows).reshape(-1, 32)).cuda()
        return fv.axpy(fid, torch.zeros_like(d), d, one256[fid]).cpu().numpy()

    bm = to_mont(fv.BN254_FQ, host_bases).reshape(n, 64)
    sm = to_mont(fv.BN254_FR, host_scalars)
    L.nmx_cache_clear()
    s0 = _lib.stats()
    t0 = time.perf_counter()
    cold = g.vartime_multiscalar_mul(sm, bm, mont=True)    # first sight: upload + convert + window tables
    cold_ms = (time.perf_counter() - t0) * 1e3
    ms, r = med_ms(lambda: g.vartime_multiscalar_mul(sm, bm, mont=True), warm=0)
    short = n - 1 - 12345                                   # a shorter prefix of the same array: still no upload
    r_short = g.vartime_multiscalar_mul(sm[:short], bm[:short], mont=True)
    s1 = _lib.stats()
    out["trait_form"] = {
        "ms": round(ms, 4), "value": n / (ms * 1e-3), "unit": "pairs/s", "first_call_ms": round(cold_ms, 3),
        "what": "nmx_msm(curve, scalars, bases, n, NMX_SCALARS_MONT|NMX_BASES_MONT): slice form, pageable host arrays, "
                "bases resident via the slice cache (tables included), scalars cross PCIe inside the call",
        "uploads": s1[_lib.STAT_CACHE_UPLOADS] - s0[_lib.STAT_CACHE_UPLOADS],
        "hits": s1[_lib.STAT_CACHE_HITS] - s0[_lib.STAT_CACHE_HITS],
        "vs_incl_h2d": round(ms / out["incl_h2d"]["ms"], 3),
        "matches_hbm_resident_result": r == ref and cold == ref,
        "prefix_matches_handle_form": r_short == g.vartime_multiscalar_mul(host_scalars[:short], ck),
    }
    L.nmx_cache_clear()
    del bm, sm
    # (2b) the scalar sets of the reference's own bench (benches/commit.rs:33-110: u1, u10, u16, u32, u64 beside `random`; SURVEY.md 8(d)
    # config 2) through the same call, HBM-resident field scalars, each checked against the oracle at full size
    try:
        sets = {}
        ok_all = True
        keyb = ck.read(0, n)
        prep = cref.Prepared(cid, keyb, n)
        cref.set_threads(effective_cpus())
        for kind in ("u1", "u10", "u16", "u32", "u64"):
            hs = util.scalar_set(cid, n, kind, seed=util.SEED + 77)
            ds = torch.from_numpy(hs).cuda()
            ms, r = med_ms(lambda: g.vartime_multiscalar_mul(ds, ck))
            okk = (r.xy, int(r.is_inf)) == prep.msm(hs, n)
            ok_all = ok_all and okk
            sets[kind] = {"ms": round(ms, 4), "value": n / (ms * 1e-3), "gpu_matches_cpu": okk}
            del ds
        sets["random"] = {"ms": round(out["ms_per_step"], 4), "value": out["value"]}
        sets["what"] = ("commit(ck, v) with r = 0 at 2^20 for the scalar distributions of benches/commit.rs (values of 1 / 10 / 16 / 32 / 64 bits stored as "
                        "field scalars, as the reference's bench does), key and scalars resident in HBM, median of 5")
        sets["all_match_cpu"] = ok_all
        out["commit_rs_sets"] = sets
        del prep, keyb
    except Exception as e:                                 # never lose the headline to an auxiliary block
        out["commit_rs_sets"] = {"error": repr(e)}
    # (3) N = 1 anchor of the configs[2] curve: 2^24 pairs on one GPU (c = 20 tables, 13 GiB)
    try:
        n24 = 1 << 24
        ck24 = nova_amd.CommitmentKey.generate(cid, n24, k0=1)
        # 2^22 random scalars tiled four times over 2^24 distinct bases: the same bucket statistics, a quarter of the
        # host-side generation time
        sc24 = torch.from_numpy(util.random_scalars(cid, n24 // 4, seed=util.SEED + 24)).cuda().repeat(4, 1).contiguous()
        ms, r24 = med_ms(lambda: g.vartime_multiscalar_mul(sc24, ck24), reps=3)
        # size-independent check: the two halves as partials sum to the whole (shard additivity)
        h = n24 // 2
        parts = [g.vartime_multiscalar_mul(sc24[:h], ck24, partial=True).xy,
                 g.vartime_multiscalar_mul(sc24[h:], ck24, partial=True, offset=h).xy]
        out["anchor_2p24_single_gpu"] = {"ms": round(ms, 3), "value": n24 / (ms * 1e-3), "unit": "pairs/s",
                                         "halves_sum_to_whole": g.point_sum(parts) == r24,
                                         "what": "2^24 distinct bases (c = 20 tables, 13 GiB), 2^22 random scalars tiled x4 (host-side "
                                                 "generation time); the full 2^24 oracle compare is tests/test_gpu_large.py::"
                                                 "test_2p24_against_the_oracle"}
        ck24.close()
        del sc24
    except nova_amd.NmxError as e:                          # e.g. a box with less free HBM: report, do not fail the headline
        out["anchor_2p24_single_gpu"] = {"error": str(e)}
    # (4) configs[3]: prove_step provider-call replay, 65 536 MinRoot iterations per step, incl. the six SpMVs
    a2 = argparse.Namespace(**vars(args))
    a2.iters, a2.steps, a2.warmup, a2.overlap_commits, a2.also_overlap = 65536, 5, 2, 0, True
    ps = prove_step_replay(a2, torch)
    out["prove_step_replay_ms"] = {"ms": round(ps["value"], 4), "iters_per_step": 65536, "cpu_ms": round(ps["cpu_baseline"]["value"], 2),
                                   "cpu_cores": ps["cpu_baseline"]["cores"], "gpu_matches_cpu": ps["cpu_baseline"]["gpu_matches_cpu"],
                                   "breakdown_ms": ps.get("breakdown_ms"), "overlap": ps.get("overlap"),
                                   "trait_only": ps.get("trait_only"), "what": ps["config"]["workload"]}
    # (5) configs[4]: HyperKZG prove replay at n = 2^20
    a3 = argparse.Namespace(**vars(args))
    a3.log2n, a3.steps, a3.warmup = 20, 3, 1
    hk = hyperkzg_replay(a3, torch, ck=ck)
    # (5b) configs[4], the sum-check half: Spartan prove replay at num_cons = 2^20 (the three provers as one call each)
    try:
        a4 = argparse.Namespace(**vars(args))
        a4.log2n, a4.steps, a4.warmup = 20, 3, 1
        sr = spartan_replay(a4, torch)
        out["spartan_replay_ms"] = {"ms": round(sr["value"], 3), "log2n": 20, "cpu_ms": round(sr["cpu_baseline"]["value"], 1),
                                    "cpu_cores": sr["cpu_baseline"]["cores"], "gpu_matches_cpu": sr["cpu_baseline"]["gpu_matches_cpu"],
                                    "breakdown_ms": sr["breakdown_ms"], "provers": sr["provers"], "what": sr["config"]["workload"]}
    except Exception as e:                                 # never lose the headline to an auxiliary block
        out["spartan_replay_ms"] = {"error": repr(e)}
    # (5c) configs[4] as ONE chained sequence: CompressedSNARK::prove (folds -> Spartan -> HyperKZG on the batched witness in HBM)
    try:
        a5 = argparse.Namespace(**vars(args))
        a5.log2n, a5.steps, a5.warmup, a5.log2n_secondary = 20, 5, 3, 14
        cs = compressed_snark_replay(a5, torch)
        out["compressed_snark_replay_ms"] = {"ms": round(cs["value"], 3), "log2n": 20, "log2n_secondary": 14,
                                             "cpu_ms": round(cs["cpu_baseline"]["value"], 1), "cpu_cores": cs["cpu_baseline"]["cores"],
                                             "gpu_matches_cpu": cs["cpu_baseline"]["gpu_matches_cpu"], "groups_ms": cs["groups_ms"],
                                             "breakdown_ms": cs["breakdown_ms"], "proof_verifies": all(cs["proof_verifies"].values()),
                                             "trait_only": {k: v for k, v in cs["trait_only"].items() if k != "per_call_ms"},
                                             "cpp_driver": {k: v for k, v in (cs.get("cpp_driver") or {}).items() if k not in ("what", "per_step_ms")},
                                             "what": cs["config"]["workload"]}
    except Exception as e:                                 # never lose the headline to an auxiliary block
        out["compressed_snark_replay_ms"] = {"error": repr(e)}
    # (5d) the secondary's evaluation argument on its own: the inner-product argument at 2^14
    try:
        a6 = argparse.Namespace(**vars(args))
        a6.log2n, a6.steps, a6.warmup = 14, 5, 2
        ir = ipa_replay(a6, torch)
        out["ipa_prove_ms"] = {"ms": round(ir["value"], 3), "log2n": 14, "ms_per_round": ir["ms_per_round"], "cpu_ms": round(ir["cpu_baseline"]["value"], 1),
                               "cpu_cores": ir["cpu_baseline"]["cores"], "gpu_matches_cpu": ir["cpu_baseline"]["gpu_matches_cpu"],
                               "checks": ir["cpu_baseline"]["checks"], "what": ir["config"]["workload"]}
    except Exception as e:                                 # never lose the headline to an auxiliary block
        out["ipa_prove_ms"] = {"error": repr(e)}
    # (6) the field-vector kernels' rooflines (north_star: >= 40 % of HBM is about THESE kernels)
    try:
        out["fieldvec"] = fieldvec_block(args, torch, L)
    except Exception as e:                                 # never lose the headline to an auxiliary block
        out["fieldvec"] = {"error": repr(e)}
    out["hyperkzg_replay_ms"] = {"ms": round(hk["value"], 3), "log2n": 20, "cpu_ms": round(hk["cpu_baseline"]["value"], 1),
                                 "cpu_cores": hk["cpu_baseline"]["cores"], "gpu_matches_cpu": hk["cpu_baseline"]["gpu_matches_cpu"],
                                 "what": hk["config"]["workload"]}


def util_modulus(fid):
    from oracle import pyref as R
    return [R.BN254_Q, R.BN254_R, R.PALLAS_P, R.PALLAS_Q][fid]


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
    from tests import util
    return util.witness_like(cid, n, seed)


def minroot_like_matrices(fid, rows, cols, seed):
    """Three CSR matrices shaped like the MinRoot step circuit's A, B, C (examples/minroot.rs:83-135: x_{i+1}^5 = x_i + y_i as
    three multiplication gates per iteration): A rows carry two terms, B and C one; nine coefficients in ten are 1, the
    rest full-width (the augmented circuit's constants)."""
    from tests import util
    rng = np.random.Generator(np.random.PCG64(seed))
    mats = []
    for j, per_row in enumerate((2, 1, 1)):
        nnz = per_row * rows
        indptr = np.arange(0, nnz + 1, per_row, dtype=np.uint64)
        indices = rng.integers(0, cols, size=nnz).astype(np.uint64)
        data = np.zeros((nnz, 32), np.uint8)
        data[:, 0] = 1
        big = rng.random(nnz) < 0.1
        cidx = {0: 1, 1: 0, 2: 3, 3: 2}[fid]          # a curve whose scalar field is `fid`
        data[big] = util.random_scalars(cidx, int(big.sum()), seed=seed + j)
        mats.append((indptr, indices, data))
    return mats


def prove_step_replay(args, torch):
    """REPLAY of the provider calls of one RecursiveSNARK::prove_step (src/nova/mod.rs:456-541) on the
    MinRoot step circuit (examples/minroot.rs: 3 constraints per iteration + the 9 986-constraint augmented circuit
    on BN254, 10 538 on Grumpkin).  Per step, as the reference orders them: for each of the two NIFS folds Z = Z1 + Z2,
    the three sparse products A*Z, B*Z, C*Z (`multiply_vec`, src/r1cs/mod.rs:612 -> sparse.rs:201-229; matrices resident in
    HBM), the cross term T, commit(T), the two AXPY folds; plus the two witness commitments: 4 MSMs + 6 SpMVs + 2 vector
    adds + 2 cross terms + 4 AXPYs, every vector resident in HBM, every commitment returned to the host (they feed the
    Poseidon RO challenge on the reference side).  NOT replayed: witness synthesis and Poseidon hashing (host side of
    the reference).  This is a replay, not prove_step: the Rust reference cannot be built here."""
    import nova_amd
    from nova_amd import fieldvec as fv
    from tests import util
    N = 3 * args.iters + 9986          # primary (BN254 / Pallas) witness / constraint count
    n2 = 10538                         # secondary (Grumpkin / Vesta)
    # the curve cycle: BN254 / Grumpkin (configs[3]) or Pallas / Vesta (north_star's other cycle; src/provider/pasta.rs)
    cP, cS = {"bn254": (0, 1), "pasta": (2, 3)}[getattr(args, "cycle", "bn254")]
    cur = {"P": (cP, fv.SCALAR_FIELD_OF_CURVE[cP], N), "S": (cS, fv.SCALAR_FIELD_OF_CURVE[cS], n2)}
    ce = {k: nova_amd.CommitmentEngine(c[0]) for k, c in cur.items()}
    ck = {k: ce[k].setup_synthetic(c[2], k0=3) for k, c in cur.items()}
    host, dev, csr, mats = {}, {}, {}, {}
    for k, (cid, fid, n) in cur.items():
        host[k] = {"W": witness_like(cid, n, 11), "W1": util.random_scalars(cid, n, seed=12),
                   "E1": util.random_scalars(cid, n, seed=13)}
        dev[k] = {nm: torch.from_numpy(v).cuda() for nm, v in host[k].items()}
        csr[k] = minroot_like_matrices(fid, n, n, seed=100 + cid)     # Z has one entry per variable here (io folded in)
        mats[k] = [fv.SparseMatrix(fid, ip, ix, dt, n) for ip, ix, dt in csr[k]]
    u = util.random_scalars(cP, 1, seed=31)
    r = {k: util.random_scalars(c[0], 1, seed=32) for k, c in cur.items()}
    rT = {k: util.random_scalars(c[0], 1, seed=33) for k, c in cur.items()}
    uS = util.random_scalars(cS, 1, seed=34)

    spans = None                     # per-call wall times of the instrumented passes (every call is synchronous)

    def call(name, fn):
        if spans is None:
            return fn()
        t = time.perf_counter()
        v = fn()
        spans.setdefault(name, []).append(time.perf_counter() - t)
        return v

    def nifs(k, uu):
        cid, fid, n = cur[k]
        d = dev[k]
        # the field kernels between two commitments are stream-ordered calls (NMX_ASYNC): nothing on the host looks at Z, AZ,
        # BZ, CZ, T, W, E before the next commitment, which is synchronous and ordered behind them
        a = not getattr(args, "sync_field_ops", False)
        if a and not getattr(args, "separate_field_ops", False):
            # one call per reference function: commit_T's chain (nmx_r1cs_cross_term) and the fold (nmx_nifs_fold)
            T = call(f"{k}.cross_term", lambda: fv.r1cs_cross_term(mats[k][0], mats[k][1], mats[k][2], d["W1"], d["W"], d["E1"], uu, async_=True))
            comT = call(f"{k}.commit_T", lambda: ce[k].commit(ck[k], T, rT[k]))
            W, E = call(f"{k}.fold_x2", lambda: fv.nifs_fold(fid, d["W1"], d["W"], d["E1"], T, r[k], async_=True))
            keep.extend((T, W, E))
            return comT, W, E
        Z = call(f"{k}.vec_add", lambda: fv.vec_add(fid, d["W1"], d["W"], async_=a))                       # r1cs/mod.rs:590-609
        AZ, BZ, CZ = (call(f"{k}.spmv_x3", lambda m=m: m.multiply_vec(Z, async_=a)) for m in mats[k])      # r1cs/mod.rs:612
        T = call(f"{k}.cross_term", lambda: fv.cross_term(fid, AZ, BZ, CZ, d["E1"], uu, async_=a))         # r1cs/mod.rs:614-620
        comT = call(f"{k}.commit_T", lambda: ce[k].commit(ck[k], T, rT[k]))                                # r1cs/mod.rs:622
        W = call(f"{k}.fold_x2", lambda: fv.axpy(fid, d["W1"], d["W"], r[k], async_=a))                    # r1cs/mod.rs:1058-1062
        E = call(f"{k}.fold_x2", lambda: fv.axpy(fid, d["E1"], T, r[k], async_=a))                         # r1cs/mod.rs:1063-1067
        keep.extend((Z, AZ, BZ, CZ, T, W, E))      # alive until the step's last (synchronous) call has returned
        return comT, W, E

    keep = []

    def step():
        out = []
        keep.clear()
        out.append(nifs("S", uS)[0])                                          # nova/mod.rs:464  NIFS on the secondary
        out.append(call("P.commit_W", lambda: ce["P"].commit(ck["P"], dev["P"]["W"])))   # nova/mod.rs:477-496 primary witness commit
        out.append(nifs("P", u)[0])                                           # nova/mod.rs:502  NIFS on the primary
        out.append(call("S.commit_W", lambda: ce["S"].commit(ck["S"], dev["S"]["W"])))   # nova/mod.rs:515-541 secondary witness commit
        return out

    # commit(W) does not feed commit_T (r1cs/mod.rs:590-622 reads W2 and X, never comm_W; the RO absorbs comm_W, nifs.rs:53, but
    # is squeezed only after comm_T, :60-63): a GPU provider can run the two MSMs side by side -- rayon::join on the Rust side,
    # nmx_commit_begin / nmx_commit_finish here -- and one's latency-bound tail hides under the other's accumulation.
    def make_step(ov):
        if not ov:
            return step

        def step_ov():
            out = [None] * 4
            keep.clear()
            if ov == 2:   # the secondary witness commitment of the PREVIOUS step runs beside this step's secondary fold
                t = ce["S"].commit_begin(ck["S"], dev["S"]["W"])
            out[0] = nifs("S", uS)[0]
            if ov == 2:
                out[3] = t.finish()
            t = ce["P"].commit_begin(ck["P"], dev["P"]["W"])                     # nmx_commit_begin: W.commit(ck), r1cs.rs:47
            out[2] = nifs("P", u)[0]                                             # cross term, commit(T), folds
            out[1] = t.finish()                                                  # before U2.absorb_in_ro, nifs.rs:53
            if ov != 2:
                out[3] = ce["S"].commit(ck["S"], dev["S"]["W"])
            return out
        return step_ov

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r_ = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps, r_
    ov = getattr(args, "overlap_commits", 0)
    dt, res = timed(make_step(ov))
    overlap = None
    if getattr(args, "also_overlap", False):   # the default line: the serial replay is `ms`, the two overlapped orders beside it
        overlap = {}
        same = True
        for o_, name in ((1, "within_step_ms"), (2, "across_steps_ms")):
            d_, r_ = timed(make_step(o_))
            overlap[name] = round(d_ * 1e3, 4)
            same = same and [(c_.xy, c_.is_inf) for c_ in r_] == [(c_.xy, c_.is_inf) for c_ in res]
        overlap["same_commitments_as_serial"] = same
        overlap["what"] = ("commit(W) begun with nmx_commit_begin and collected behind the cross term + commit(T) it does not feed "
                           "(r1cs/mod.rs:590-622 never reads comm_W; nifs.rs:53-63): within_step = the primary pair of one prove_step; "
                           "across_steps = also the secondary pair, whose commit(W) ends step i and whose fold opens step i + 1")
    # the same step again with every provider call timed on its own (wall clock around the C call; the field-vector calls are
    # stream-ordered unless --sync-field-ops, so their spans are enqueue times and their kernels run under the next commitment's
    # span; P = primary BN254, S = secondary Grumpkin; x2 / x3 = sum over that many calls)
    spans = {}
    passes = 5
    for _ in range(passes):
        step()
    breakdown = {k: round(sum(v) / passes * 1e3, 4) for k, v in sorted(spans.items())}
    breakdown["_sum"] = round(sum(v for v in breakdown.values()), 4)
    spans = None
    outj = {
        "metric": f"RecursiveSNARK prove_step provider-call REPLAY ms (minroot, {nova_amd.CURVE_NAMES[cP]}/{nova_amd.CURVE_NAMES[cS]})", "value": dt * 1e3, "unit": "ms",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": False,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": f"prove_step replay: minroot {args.iters} iterations/step -> primary N={N} ({nova_amd.CURVE_NAMES[cP]}), secondary n={n2} "
                               f"({nova_amd.CURVE_NAMES[cS]}); 4 MSMs + 6 SpMVs + 2 vector adds + 2 cross terms + 4 folds (stream-ordered: commit_T's chain and the fold one call each); no synthesis / Poseidon "
                               "(BASELINE.json configs[3])"},
        "roofline": None,
        "breakdown_ms": breakdown,
    }
    if ov:
        outj["overlap_commits"] = ov
    if overlap:
        outj["overlap"] = overlap
    if not args.no_cpu_baseline:
        from oracle import cref
        threads = effective_cpus()
        cref.set_threads(threads)
        keys = {k: ck[k].read(0, cur[k][2]) for k in cur}
        prep = {k: cref.Prepared(cur[k][0], keys[k], cur[k][2]) for k in cur}
        t1 = time.perf_counter()
        exp, tlog = [], {}
        for k, uu in (("S", uS), ("P", u)):
            cid, fid, n = cur[k]
            h = host[k]
            Z = np.frombuffer(cref.field_axpy(fid, h["W1"], h["W"], util.int_to_le32(1), n), np.uint8).reshape(n, 32)  # W1 + 1*W
            AZ, BZ, CZ = (cref.spmv(fid, ip, ix, dt, n, Z) for ip, ix, dt in csr[k])
            T = cref.field_cross_term(fid, AZ, BZ, CZ, h["E1"], uu, n)
            comT = cref.commit(cid, T, keys[k], n, ck[k].h, rT[k])
            cref.field_axpy(fid, h["W1"], h["W"], r[k], n)
            cref.field_axpy(fid, h["E1"], T, r[k], n)
            comW = prep[k].msm(h["W"], n)
            exp.append((comT, comW))
            tlog[k] = (cid, keys[k], [("commit", [np.frombuffer(T, np.uint8).reshape(n, 32)], None), ("commit", [h["W"]], [comW])], prep[k])
        t_cpu = time.perf_counter() - t1
        ok = [(res[0].xy, int(res[0].is_inf)) == exp[0][0], (res[1].xy, int(res[1].is_inf)) == exp[1][1],
              (res[2].xy, int(res[2].is_inf)) == exp[1][0], (res[3].xy, int(res[3].is_inf)) == exp[0][1]]
        outj["cpu_baseline"] = {"value": t_cpu * 1e3, "unit": "ms", "cores": threads, "kind": "port",
                                "sample": "the same call sequence once through oracle/nova_ref.c (commit re-loads the key "
                                          "each call, as a fresh Vec<Affine> would not)", "serial_parts": CPU_SERIAL_PARTS,
                                "gpu_matches_cpu": all(ok)}
        # what the step costs the provider when Nova calls it UNCHANGED (only the DlogGroupExt override wired in): the four MSMs
        # in slice form over host scalars -- 206 594 / 10 538 of them at the default size --, everything else on the reference's CPU
        if getattr(args, "trait_only", True):
            outj["trait_only"] = trait_only_msms(tlog)
    for k in ck:
        ck[k].close()
        for m in mats[k]:
            m.close()
    return outj


def hyperkzg_replay(args, torch, ck=None):
    """REPLAY of the provider-side work of one HyperKZG `prove` (src/provider/hyperkzg.rs:926-1110) for
    n = 2^log2n on BN254, everything resident in HBM, commitments / evaluations returned to the host:
      ell-1 pair folds Pi[j] = P[2j] + x*(P[2j+1]-P[2j])                      (hyperkzg.rs:1085-1095)
      batch_commit of the folded polynomials, lengths n/2 ... 2               (hyperkzg.rs:1100)
      3*ell Horner evaluations f_i(u_j)                                       (hyperkzg.rs:1011-1020,1049-1056)
      B = sum q^i f_i                                                         (hyperkzg.rs:1028-1040)
      3 x kzg_open: h = div_by_monomial(B, u_j), commit(h)                    (hyperkzg.rs:961-1004,1062-1065)
    NOT replayed: the Keccak transcript (challenges are fixed random scalars).  A replay, not `prove`.
    ck: an already resident key of >= n points to use (the headline run passes its own)."""
    import nova_amd
    from nova_amd import fieldvec as fv
    from tests import util
    ell = args.log2n
    n = 1 << ell
    cid = 0
    fid = fv.SCALAR_FIELD_OF_CURVE[cid]
    ce = nova_amd.CommitmentEngine(cid)
    own_ck = ck is None
    if own_ck:
        ck = ce.setup_synthetic(n, k0=5)
    assert len(ck) >= n
    hP = util.random_scalars(cid, n, seed=41)
    xs = util.random_scalars(cid, ell, seed=42)
    us = util.random_scalars(cid, 3, seed=43)
    qs = util.random_scalars(cid, ell, seed=44)
    dP = torch.from_numpy(hP).cuda()

    def step():
        if getattr(args, "separate_folds", False) or getattr(args, "sync_field_ops", False):
            polys, cur = [dP], dP
            for i in range(ell - 1):   # stream-ordered: the host looks at none of the folded polynomials before they are committed
                cur = fv.fold_pairs(fid, cur, xs[ell - i - 1], async_=not getattr(args, "sync_field_ops", False))
                polys.append(cur)
        else:                          # the fold loop as ONE call (nmx_poly_fold_chain: the short folds inside one block)
            polys = [dP] + fv.fold_chain(fid, dP, np.ascontiguousarray(xs[::-1][:ell - 1]), async_=True)
        coms = ce.batch_commit(ck, polys[1:])
        evals = fv.poly_eval_multi(fid, polys, us)   # the whole v matrix (hyperkzg.rs:1049-1056) in one launch
        B = fv.lincomb_powers(fid, polys, qs[0])     # B = sum_i q^i f_i (kzg_compute_batch_polynomial, hyperkzg.rs:1028-1040)
        # the three openings run in parallel in the reference too (`u.into_par_iter()`, hyperkzg.rs:1062-1065)
        opens = [None] * 3

        def open_at(j):
            opens[j] = ce.commit(ck, fv.div_by_monomial(fid, B, us[j]).contiguous())
        opens_mode = getattr(args, "opens", "batch")
        if opens_mode == "batch":
            # the three quotients committed by ONE batch_commit over the key (a fused run: one partition / accumulate / reduction
            # pass, a bucket set per quotient) instead of three concurrent commits -- on the reference side kzg_open's
            # `u.into_par_iter()` of three commits becomes three divisions + one CE::batch_commit
            hs = [fv.div_by_monomial(fid, B, us[j]).contiguous() for j in range(3)]
            opens = ce.batch_commit(ck, hs)
        elif opens_mode == "serial":
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
        "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
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
                                "sample": "the same call sequence once through oracle/nova_ref.c (OpenMP wherever the reference uses rayon; "
                                          "the division passes chunked as hyperkzg.rs:961-999)", "serial_parts": CPU_SERIAL_PARTS,
                                "gpu_matches_cpu": ok}
    if own_ck:
        ck.close()
    return outj


def spartan_like_matrices(fid, n, seed):
    """A, B, C of a padded Spartan shape (num_cons = num_vars = n, z = [W | u | X | 0...] of 2 n entries): minroot-like rows over
    the live columns [0, n + 2), and -- what a real R1CS matrix has and uniformly random columns lack -- the constant column
    (index n, the `u` entry of z) in one row in eight of A: a column of ~n/8 entries that the transposed product must split."""
    mats = minroot_like_matrices(fid, n, n + 2, seed)
    ip, ix, dt = mats[0]
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    heavy = rng.random(len(ix)) < 0.125
    ix = ix.copy()
    ix[heavy] = n
    mats[0] = (ip, ix, dt)
    return mats


SPARTAN_SEED = 2025
FIELD_NAMES = {0: "BN254 Fq = Grumpkin's scalar field", 1: "BN254 Fr", 2: "Pallas Fp = Vesta's scalar field", 3: "Pallas Fq = Pallas's scalar field"}


def spartan_instance(fid, ell, seed=None):
    """Host data of the replayed instance: the three matrices, W (witness-like), u, one public input, z = [W | u | X | 0...]."""
    from tests import util
    n = 1 << ell
    cid = {1: 0, 0: 1, 3: 2, 2: 3}[fid]
    csr = spartan_like_matrices(fid, n, seed=(700 + ell) if seed is None else seed)
    hW = witness_like(cid, n, 71)
    u, x0 = util.random_scalars(cid, 1, seed=73), util.random_scalars(cid, 1, seed=74)
    hz = np.zeros((2 * n, 32), np.uint8)
    hz[:n], hz[n], hz[n + 1] = hW, u[0], x0[0]
    return csr, hW, u, hz


def spartan_sequence(be, ell, p, u, call=lambda name, fn: fn(), inst=None, tr=None):
    """The provider calls of `RelaxedR1CSSNARK::prove` up to the evaluation argument, in the reference's order
    (src/spartan/snark.rs:133-233), against a provider `be` (the HIP path or the oracle behind one interface).
    inst = (W, E, X): the instance's vectors where they are (the chained replay hands over a folded witness that never left HBM);
    default: the provider's own be.W / be.E / be.concat_z().  tr: a transcript already under way."""
    from tests import standin
    le = lambda v: int(v).to_bytes(32, "little")
    num = lambda b: int.from_bytes(bytes(b), "little")
    tr = tr or standin.Transcript(seed=SPARTAN_SEED)
    tau = [tr.squeeze() for _ in range(ell)]                                  # snark.rs:141-143
    if inst is not None:
        class _View:                                                          # the same provider, looking at the instance handed in
            W, E = inst[0], inst[1]

            def __getattr__(self, name):
                return getattr(be_, name)
        be_, be = be, _View()
        zc = call("z_concat", lambda: be_.concat_z(inst[0], u, inst[2]))      # :133, :193-196
    else:
        zc = call("z_concat", lambda: be.concat_z())                          # :133, :193-196
    if hasattr(be, "spmv_all"):                                               # S.multiply_vec(&z) is ONE reference function (r1cs/mod.rs:407-471)
        Az, Bz, Cz = call("spmv_x3", lambda: be.spmv_all(zc))                  # :146
    else:
        Az, Bz, Cz = (call("spmv_x3", lambda j=j: be.spmv(j, zc)) for j in range(3))
    uCzE = call("uCz_E", lambda: be.axpy(be.E, Cz, u))                        # :147-149
    outer = call("sumcheck_outer", lambda: be.cubic3(le(0), b"".join(tau), Az, Bz, uCzE, tr))     # :158-165
    r_x = outer[1]
    claim_Az, claim_Bz = outer[2][0], outer[2][1]
    claim_Cz, eval_E = call("evaluate_Cz_E", lambda: be.multi_evaluate([Cz, be.E], b"".join(r_x)))  # :169-170
    tr.absorb(claim_Az + claim_Bz + claim_Cz + eval_E)                         # :171-174
    r = tr.squeeze()                                                           # :177
    rr = num(r)
    claim_inner = (num(claim_Az) + rr * num(claim_Bz) + rr * rr * num(claim_Cz)) % p
    evals_rx = call("eq_evals", lambda: be.eq_evals(b"".join(r_x)))            # :182
    if hasattr(be, "spmv_t_all"):                                              # compute_eval_table_sparse (spartan/mod.rs:497-533) likewise
        eA, eB, eC = call("spmv_T_x3", lambda: be.spmv_t_all(evals_rx))        # :184
    else:
        eA, eB, eC = (call("spmv_T_x3", lambda j=j: be.spmv_t(j, evals_rx)) for j in range(3))
    ABC = call("poly_ABC", lambda: be.axpy2(eA, eB, eC, r))                    # :188-190
    inner = call("sumcheck_inner", lambda: be.quad(le(claim_inner), ell + 1, ABC, zc, tr))   # :199-205
    r_y = inner[1]
    eval_W = call("evaluate_W", lambda: be.evaluate(be.W, b"".join(r_y[1:])))  # :215
    tr.absorb(eval_W)
    rho = tr.squeeze()                                                         # spartan/mod.rs:395
    Wc, Ec = call("clone_W_E", lambda: (be.clone(be.W), be.clone(be.E)))       # spartan/mod.rs:407-410
    batch = call("sumcheck_batch", lambda: be.batch([eval_W, eval_E], [ell, ell], [Wc, Ec], [b"".join(r_y[1:]), b"".join(r_x)],
                                                    [le(1), rho], tr))         # spartan/mod.rs:414-421
    tr.absorb(b"".join(batch[2]))
    c = tr.squeeze()                                                           # spartan/mod.rs:425
    w_joint = call("batch_witness", lambda: be.lincomb([be.W, be.E], c))       # spartan/mod.rs:429
    # (the batched witness stays where it is -- it is EE::prove's input; the caller brings it to the host outside the timed region)
    return {"outer": outer, "inner": inner, "batch": batch, "evaluations": (claim_Cz, eval_E, eval_W), "batch_witness": w_joint,
            "tau": tau, "r": r, "rho": rho, "c": c}


def spartan_verify(p, u, res):
    """The reference's verifier equations on a replayed proof (snark.rs:276-289 outer, :303-345 inner with the prover's own
    evaluations of ABC and z standing in for the verifier's matrix evaluations, spartan/mod.rs:440-470 batch)."""
    from tests import spartan_common as spc
    num = lambda b: int.from_bytes(bytes(b), "little")
    ival = lambda rows: [[num(c) for c in row] for row in rows]
    tau_i = [num(t) for t in res["tau"]]
    rx_i, ry_i, rb_i = ([num(r) for r in res[k][1]] for k in ("outer", "inner", "batch"))
    cAz, cBz, cT = (num(c) for c in res["outer"][2])
    claim_Cz, eval_E, eval_W = (num(c) for c in res["evaluations"])
    e_out = spc.verify_rounds(p, 0, ival(res["outer"][0]), rx_i, 3)
    ok_outer = e_out == spc.eq_eval(p, tau_i, rx_i) * (cAz * cBz - cT) % p and cT == (num(u.tobytes()) * claim_Cz + eval_E) % p
    rr = num(res["r"])
    e_in = spc.verify_rounds(p, (cAz + rr * cBz + rr * rr * claim_Cz) % p, ival(res["inner"][0]), ry_i, 2)
    ok_inner = e_in == num(res["inner"][2][0]) * num(res["inner"][2][1]) % p
    rho = num(res["rho"])
    e_b = spc.verify_rounds(p, (eval_W + rho * eval_E) % p, ival(res["batch"][0]), rb_i, 2)
    fW, fE = (num(c) for c in res["batch"][2])
    ok_batch = e_b == (spc.eq_eval(p, ry_i[1:], rb_i) * fW + rho * spc.eq_eval(p, rx_i, rb_i) * fE) % p
    return {"proof_verifies_outer": ok_outer, "proof_verifies_inner": ok_inner, "proof_verifies_batch": ok_batch}


class SpartanCpu:
    """the oracle behind spartan_sequence's provider interface (the checker / cpu_baseline leg)"""

    def __init__(self, fid, csr, n, hW, hE, hz):
        from oracle import cref
        self.cref, self.fid, self.csr, self.n, self.W, self.E, self.z = cref, fid, csr, n, hW, hE, hz

    def _np(self, b, m):
        return np.frombuffer(b, np.uint8).reshape(m, 32)

    def clone(self, v):
        return v.copy()

    def concat_z(self):
        return self.z.copy()

    def spmv(self, j, v):
        return self._np(self.cref.spmv(self.fid, *self.csr[j], self.n, v), self.n)

    def spmv_t(self, j, v):
        return self._np(self.cref.spmv_transposed(self.fid, *self.csr[j], self.n, 2 * self.n, v), 2 * self.n)

    def axpy(self, a, b, r):
        return self._np(self.cref.field_axpy(self.fid, a, b, r, len(a)), len(a))

    def axpy2(self, a, b, c, r):
        return self._np(self.cref.field_axpy2(self.fid, a, b, c, r, len(a)), len(a))

    def multi_evaluate(self, zs, r):
        return self.cref.mle_multi_evaluate(self.fid, [z.tobytes() for z in zs], len(r) // 32, r)

    def evaluate(self, z, r):
        return self.cref.mle_evaluate(self.fid, z, len(r) // 32, r)

    def eq_evals(self, r):
        return self._np(self.cref.eq_evals(self.fid, r, len(r) // 32), 1 << (len(r) // 32))

    def lincomb(self, vs, s):
        return self.cref.lincomb_powers(self.fid, [v.tobytes() for v in vs], s, max(len(v) for v in vs))

    def host(self, v):
        return bytes(v)

    def cubic3(self, claim, taus, A, B, C, tr):
        return self.cref.sumcheck_prove_cubic3(self.fid, claim, taus, A, B, C, tr.fn(self.cref.TRANSCRIPT_FN), ctx=tr.ctx)

    def quad(self, claim, nr, A, B, tr):
        return self.cref.sumcheck_prove_quad_prod(self.fid, claim, nr, A, B, tr.fn(self.cref.TRANSCRIPT_FN), ctx=tr.ctx)

    def batch(self, claims, nrs, polys, pts, coeffs, tr):
        return self.cref.sumcheck_prove_batch_eval(self.fid, claims, nrs, [q.tobytes() for q in polys], pts, coeffs,
                                                   tr.fn(self.cref.TRANSCRIPT_FN), ctx=tr.ctx)


def spartan_replay(args, torch):
    """REPLAY of the provider-side work of `RelaxedR1CSSNARK::prove` up to the evaluation argument (src/spartan/snark.rs:113-260;
    BASELINE.json configs[4], the sum-check half -- the HyperKZG half is `hyperkzg_replay`) for num_cons = num_vars = 2^log2n on
    BN254's scalar field, every vector resident in HBM, in the reference's order (spartan_sequence):
      z = [W, u, X] (:133)                                              one device copy (the reference's concat)
      Az, Bz, Cz = S.multiply_vec(z) (:146)                             3 x nmx_spmv_apply
      uCz_E = u Cz + E (:147-149)                                       nmx_field_axpy
      outer sum-check, prove_cubic_with_three_inputs (:158-165)         ONE call: nmx_sumcheck_prove_cubic_with_three_inputs
      claim_Cz = Cz(r_x), eval_E = E(r_x) (:169-170)                    nmx_mle_multi_evaluate
      evals_rx = eq(r_x, .) (:182); compute_eval_table_sparse (:184)    nmx_eq_evals_from_points + 3 x nmx_spmv_apply_transposed
      poly_ABC = A + r B + r^2 C (:188-190)                             nmx_field_axpy2
      inner sum-check, prove_quad_prod over (poly_ABC, z) (:199-205)    ONE call: nmx_sumcheck_prove_quad_prod
      eval_W = W(r_y[1..]) (:215)                                       nmx_mle_evaluate
      batch_eval_reduce (:232-233; src/spartan/mod.rs:377-437): prove_batch_eval over clones of W and E, then W + c E
                                                                        ONE call: nmx_sumcheck_prove_batch_eval, nmx_field_lincomb_powers
    The instance is SATISFIED (E := Az o Bz - u Cz, built with the product's own kernels), so the replayed proof verifies: the
    reference's verifier equations are checked on it (spartan_verify).  NOT replayed: the Keccak transcript -- a native stand-in
    (tests/standin) answers the provers' per-round callback and the squeezes between them, on both sides alike --, the
    commitment side of batch_eval_reduce (a two-term group combination) and EE::prove.  A replay, not `prove`: the Rust
    reference cannot be built here."""
    import ctypes
    from nova_amd import _lib, fieldvec as fv
    ell = args.log2n
    n = 1 << ell
    # the curve whose scalar field the instance lives in: 0 = BN254 (S1 of CompressedSNARK::prove), 1 = Grumpkin (S2, the secondary
    # circuit's SNARK, src/nova/mod.rs:862-881), 2 / 3 = Pallas / Vesta
    fid = fv.SCALAR_FIELD_OF_CURVE[getattr(args, "curve", 0)]
    p = util_modulus(fid)
    L = _lib.lib()
    csr, hW, u, hz = spartan_instance(fid, ell)
    x0 = hz[n + 1:n + 2].copy()
    mats = [fv.SparseMatrix(fid, ip, ix, dt, 2 * n) for ip, ix, dt in csr]
    dW, dz = (torch.from_numpy(v).cuda() for v in (hW, hz))
    dE = fv.r1cs_cross_term(mats[0], mats[1], mats[2], dz, None, torch.zeros((n, 32), dtype=torch.uint8, device="cuda"), u)
    hE = dE.cpu().numpy()
    spans, prof = None, None

    def call(name, fn):
        if spans is None:
            return fn()
        t = time.perf_counter()
        v = fn()
        spans.setdefault(name, []).append(time.perf_counter() - t)
        if prof is not None and name.startswith("sumcheck"):
            buf = (ctypes.c_float * 8)()
            if L.nmx_profile_last(buf, 8) >= 7:
                prof.setdefault(name, []).append([buf[i] for i in range(7)])
        return v

    class Gpu:
        W, E, z = dW, dE, dz
        clone = staticmethod(lambda v: fv.concat(fid, [v], async_=True))        # on the library's stream (torch's clone is not ordered with it)
        concat_z = staticmethod(lambda: fv.concat(fid, [dW, u, x0], n_out=2 * n, async_=True))   # consumed by stream-ordered calls only
        spmv = staticmethod(lambda j, v: mats[j].multiply_vec(v, async_=True))
        spmv_t = staticmethod(lambda j, v: mats[j].multiply_vec_transposed(v, async_=True))
        if not getattr(args, "separate_spmv", False):
            spmv_all = staticmethod(lambda v: fv.multiply_vec_many(mats, v, async_=True))
            spmv_t_all = staticmethod(lambda v: fv.multiply_vec_many(mats, v, transposed=True, async_=True))
        axpy = staticmethod(lambda a, b, r: fv.axpy(fid, a, b, r, async_=True))
        axpy2 = staticmethod(lambda a, b, c, r: fv.axpy2(fid, a, b, c, r, async_=True))
        multi_evaluate = staticmethod(lambda zs, r: fv.mle_multi_evaluate(fid, zs, r))
        evaluate = staticmethod(lambda z, r: fv.mle_evaluate(fid, z, r))
        eq_evals = staticmethod(lambda r: fv.eq_evals_from_points(fid, r, device=True))
        lincomb = staticmethod(lambda vs, s: fv.lincomb_powers(fid, vs, s))
        host = staticmethod(lambda v: v.cpu().numpy().tobytes())

        @staticmethod
        def cubic3(claim, taus, A, B, C, tr):
            return fv.sumcheck_prove_cubic_with_three_inputs(fid, claim, taus, A, B, C, tr.fn(_lib.TRANSCRIPT_FN), ctx=tr.ctx)

        @staticmethod
        def quad(claim, nr, A, B, tr):
            return fv.sumcheck_prove_quad_prod(fid, claim, nr, A, B, tr.fn(_lib.TRANSCRIPT_FN), ctx=tr.ctx)

        @staticmethod
        def batch(claims, nrs, polys, pts, coeffs, tr):
            return fv.sumcheck_prove_batch_eval(fid, claims, nrs, polys, pts, coeffs, tr.fn(_lib.TRANSCRIPT_FN), ctx=tr.ctx)

    for _ in range(args.warmup):
        spartan_sequence(Gpu, ell, p, u)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = spartan_sequence(Gpu, ell, p, u)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # the same sequence with every provider call timed on its own and the provers' own split (profiling on): per prover
    # [total ms, waiting for a round's result (GPU + latency), host algebra + launches + tail rounds, transcript callback, launches,
    #  rounds run on the device, rounds finished on the host]
    spans, prof = {}, {}
    L.nmx_set_profiling(1)
    passes = 3
    for _ in range(passes):
        spartan_sequence(Gpu, ell, p, u, call)
    L.nmx_set_profiling(0)
    breakdown = {k: round(sum(v) / passes * 1e3, 4) for k, v in sorted(spans.items())}
    breakdown["_sum"] = round(sum(breakdown.values()), 4)
    provers = {k: {"ms": round(float(np.mean([q[0] for q in v])), 4), "wait_ms": round(float(np.mean([q[1] for q in v])), 4),
                   "host_algebra_ms": round(float(np.mean([q[2] for q in v])), 4), "transcript_ms": round(float(np.mean([q[3] for q in v])), 4),
                   "launches": int(v[-1][4]), "rounds": int(v[-1][5]), "host_tail_rounds": int(v[-1][6])} for k, v in prof.items()}
    spans, prof = None, None
    outj = {
        "metric": f"Spartan RelaxedR1CSSNARK prove (sum-check half) provider-call REPLAY ms ({FIELD_NAMES[fid]})", "value": dt * 1e3, "unit": "ms",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": False, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": f"Spartan prove replay, num_cons = num_vars = 2^{ell}: 3 SpMV, outer cubic sum-check ({ell} rounds, one call), "
                               f"2 evaluations, eq table, 3 transposed SpMV, inner quad_prod sum-check ({ell + 1} rounds, one call), 1 evaluation, "
                               f"batch_eval sum-check over W and E ({ell} rounds, one call), batch witness (BASELINE.json configs[4], sum-check half); "
                               "stand-in transcript",
                   "rounds": [ell, ell + 1, ell]},
        "roofline": None, "breakdown_ms": breakdown, "provers": provers,
        "proof_verifies": spartan_verify(p, u, res),
    }
    res["batch_witness"] = Gpu.host(res["batch_witness"])
    if not args.no_cpu_baseline:
        from oracle import cref
        threads = effective_cpus()
        cref.set_threads(threads)
        cpu = SpartanCpu(fid, csr, n, hW, hE, hz)
        t1 = time.perf_counter()
        exp = spartan_sequence(cpu, ell, p, u)
        t_cpu = time.perf_counter() - t1
        exp["batch_witness"] = cpu.host(exp["batch_witness"])
        checks = {k: res[k] == exp[k] for k in ("outer", "inner", "batch", "evaluations", "batch_witness")}
        checks.update(outj["proof_verifies"])
        outj["cpu_baseline"] = {"value": t_cpu * 1e3, "unit": "ms", "cores": threads, "kind": "port",
                                "sample": "the same call sequence once through oracle/nova_ref.c (OpenMP wherever the reference uses rayon)",
                                "serial_parts": CPU_SERIAL_PARTS,
                                "gpu_matches_cpu": all(checks.values()), "checks": checks}
    for m in mats:
        m.close()
    return outj


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4] as ONE sequence: CompressedSNARK::prove (src/nova/mod.rs:793-881)
# ---------------------------------------------------------------------------------------------------------------------------
class CsnarkSide:
    """Host data of one side (one curve of the cycle) of the chained replay: the padded shape (num_cons = num_vars = 2^ell,
    z = [W | u | X | 0...] of 2^(ell+1) entries), the running relaxed instance (W1, E1, u1, X1) -- SATISFIED: E1 = Az1 o Bz1 - u1 Cz1,
    built by `fill_E1` with the provider's own kernels --, the randomness of sample_random_instance_witness (W2, u2, X2;
    src/r1cs/mod.rs:786-800 draws it from OsRng on the host: data, not provider work) and the blinds."""

    def __init__(self, cid, ell, seed):
        from nova_amd import fieldvec as fv
        from tests import util
        self.cid, self.ell, self.n = cid, ell, 1 << ell
        self.fid = fv.SCALAR_FIELD_OF_CURVE[cid]
        self.p = util_modulus(self.fid)
        self.csr = spartan_like_matrices(self.fid, self.n, seed=seed)
        rs = lambda k, sd: util.random_scalars(cid, k, seed=seed + sd)
        self.W1, self.W2 = rs(self.n, 1), rs(self.n, 2)          # a running witness after many folds is full-width
        self.u1, self.X1, self.u2, self.X2 = rs(1, 3), rs(1, 4), rs(1, 5), rs(1, 6)
        self.r_W, self.r_E, self.r_T = rs(1, 7), rs(1, 8), rs(1, 9)
        self.E1 = None

    def z(self, W, u, X):
        hz = np.zeros((2 * self.n, 32), np.uint8)
        hz[:self.n], hz[self.n], hz[self.n + 1] = W, u[0], X[0]
        return hz


class GpuProvider:
    """The HIP path behind the replay sequences' provider interface: every vector resident in HBM, the field calls stream-ordered
    (NMX_ASYNC) where nothing on the host looks at their result before the next synchronous call."""

    def __init__(self, torch, side, ck):
        import nova_amd
        from nova_amd import _lib, fieldvec as fv
        self.t, self.fv, self._lib, self.s, self.ck = torch, fv, _lib, side, ck
        self.fid, self.n = side.fid, side.n
        self.ce = nova_amd.CommitmentEngine(side.cid)
        self.mats = [fv.SparseMatrix(side.fid, ip, ix, dt, 2 * side.n) for ip, ix, dt in side.csr]
        up = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
        self.W1, self.W2 = up(side.W1), up(side.W2)
        self.zeros = torch.zeros((side.n, 32), dtype=torch.uint8, device="cuda")
        if side.E1 is None:                                      # the running instance is satisfied: E1 = Az1 o Bz1 - u1 Cz1
            dE1 = fv.r1cs_cross_term(self.mats[0], self.mats[1], self.mats[2], up(side.z(side.W1, side.u1, side.X1)), None, self.zeros, side.u1)
            side.E1 = dE1.cpu().numpy()
        self.E1 = up(side.E1)
        self.ipa_u = self.ipa_generator()

    def close(self):
        for m in self.mats:
            m.close()

    pt = staticmethod(lambda c: (c.xy, int(c.is_inf)))

    def concat_z(self, W, u, X):
        return self.fv.concat(self.fid, [W, u, X], n_out=2 * self.n, async_=True)

    def cross_term0(self, z, u):                                 # AZ o BZ - u CZ (sample_random_instance_witness, r1cs/mod.rs:803-812)
        return self.fv.r1cs_cross_term(self.mats[0], self.mats[1], self.mats[2], z, None, self.zeros, u, async_=True)

    def commit(self, v, r=None):
        return self.pt(self.ce.commit(self.ck, v, r))

    def batch_commit(self, vs):
        return [self.pt(c) for c in self.ce.batch_commit(self.ck, vs)]

    def commit_pair(self, a, b):
        """two independent commitments (rayon::join on the reference side): the first begun with nmx_commit_begin, the second
        synchronous beside it -- one's latency-bound tail under the other's accumulation"""
        t = self.ce.commit_begin(self.ck, a[0], a[1])
        second = self.pt(self.ce.commit(self.ck, b[0], b[1]))
        return self.pt(t.finish()), second

    def vec_add(self, a, b):
        return self.fv.vec_add(self.fid, a, b, async_=True)

    def spmv(self, j, v):
        return self.mats[j].multiply_vec(v, async_=True)

    def spmv_t(self, j, v):
        return self.mats[j].multiply_vec_transposed(v, async_=True)

    def spmv_all(self, v):
        return self.fv.multiply_vec_many(self.mats, v, async_=True)

    def spmv_t_all(self, v):
        return self.fv.multiply_vec_many(self.mats, v, transposed=True, async_=True)

    def cross_term2(self, az, bz, cz, e1, e2, u):
        return self.fv.cross_term2(self.fid, az, bz, cz, e1, e2, u, async_=True)

    def axpy(self, a, b, r):
        return self.fv.axpy(self.fid, a, b, r, async_=True)

    def axpy2(self, a, b, c, r):
        return self.fv.axpy2(self.fid, a, b, c, r, async_=True)

    def clone(self, v):
        return self.fv.concat(self.fid, [v], async_=True)

    def multi_evaluate(self, zs, r):
        return self.fv.mle_multi_evaluate(self.fid, zs, r)

    def evaluate(self, z, r):
        return self.fv.mle_evaluate(self.fid, z, r)

    def eq_evals(self, r):
        return self.fv.eq_evals_from_points(self.fid, r, device=True)

    def lincomb(self, vs, s):
        return self.fv.lincomb_powers(self.fid, vs, s)

    def fold_pairs(self, v, x):
        return self.fv.fold_pairs(self.fid, v, x, async_=True)

    def fold_chain(self, v, xs):
        return self.fv.fold_chain(self.fid, v, xs, async_=True)

    def ipa_generator(self):
        """ck_c = CE::setup(b"ipa", 1) (src/provider/ipa_pc.rs:50): one more generator, made where the keys are made"""
        import nova_amd
        k = nova_amd.CommitmentKey.generate(self.s.cid, 1, k0=IPA_GENERATOR_K0)
        u = k.read(0, 1).tobytes()
        k.close()
        return u

    def scale_point(self, U, r):
        """ck_c.scale(&r) (ipa_pc.rs:190-191; pedersen.rs:499-506): one point times one scalar = a commitment to nothing with h = U"""
        import nova_amd
        view = nova_amd.CommitmentKey(self.s.cid, self.ck.handle, self.ck.n, U)
        return self.ce.commit(view, np.zeros((0, 32), np.uint8), r).xy

    def ipa(self, ckc, a, b, tr):
        import nova_amd
        return nova_amd.ipa_prove(self.ck, ckc, a, b, tr.fn_ipa(self._lib.IPA_TRANSCRIPT_FN), ctx=tr.ctx)

    def poly_eval_multi(self, polys, us):
        return self.fv.poly_eval_multi(self.fid, polys, us)

    def div_by_monomial(self, B, u):
        return self.fv.div_by_monomial(self.fid, B, u).contiguous()

    def host(self, v):
        return v.cpu().numpy().tobytes()

    def cubic3(self, claim, taus, A, B, C, tr):
        return self.fv.sumcheck_prove_cubic_with_three_inputs(self.fid, claim, taus, A, B, C, tr.fn(self._lib.TRANSCRIPT_FN), ctx=tr.ctx)

    def quad(self, claim, nr, A, B, tr):
        return self.fv.sumcheck_prove_quad_prod(self.fid, claim, nr, A, B, tr.fn(self._lib.TRANSCRIPT_FN), ctx=tr.ctx)

    def batch(self, claims, nrs, polys, pts, coeffs, tr):
        return self.fv.sumcheck_prove_batch_eval(self.fid, claims, nrs, polys, pts, coeffs, tr.fn(self._lib.TRANSCRIPT_FN), ctx=tr.ctx)


class CpuProvider(SpartanCpu):
    """the oracle behind the same interface (the checker / cpu_baseline leg).  `msm_log`: every commitment's (kind, host vectors) in
    call order -- what Nova hands to DlogGroupExt when the provider is wired in through the trait alone (trait_only)."""

    def __init__(self, side, key, h):
        super().__init__(side.fid, side.csr, side.n, None, None, None)
        from oracle import cref
        self.s, self.key, self.h = side, key, h
        self.prep = cref.Prepared(side.cid, key, len(key))
        self.zeros = np.zeros((side.n, 32), np.uint8)
        if side.E1 is None:                                      # (CPU-only runs: the satisfied running instance through the oracle)
            side.E1 = self.cross_term0(side.z(side.W1, side.u1, side.X1), side.u1)
        self.W1, self.W2, self.E1 = side.W1, side.W2, side.E1
        self.msm_log = []
        self.ipa_u = self.ipa_generator()

    def concat_z(self, W, u, X):
        return self.s.z(W, u, X)

    def cross_term0(self, z, u):
        az, bz, cz = (self.spmv(j, z) for j in range(3))
        return self._np(self.cref.field_cross_term(self.fid, az, bz, cz, self.zeros, u, self.n), self.n)

    def commit(self, v, r=None):
        v = np.ascontiguousarray(v)
        if r is None or bytes(r) == bytes(32):
            out = self.prep.msm(v, len(v))
            self.msm_log.append(("commit", [v], [out]))
            return out
        self.msm_log.append(("commit", [v], None))             # (the unblinded MSM the trait call returns: computed by trait_only_msms, untimed)
        return self.cref.commit(self.s.cid, v, self.key, len(v), self.h, r)

    def batch_commit(self, vs):
        vs = [np.ascontiguousarray(v) for v in vs]
        out = [self.prep.msm(v, len(v)) for v in vs]
        self.msm_log.append(("batch", vs, out))
        return out

    def vec_add(self, a, b):
        from tests import util
        return self._np(self.cref.field_axpy(self.fid, a, b, util.int_to_le32(1), len(a)), len(a))

    def cross_term2(self, az, bz, cz, e1, e2, u):
        return self._np(self.cref.field_cross_term2(self.fid, az, bz, cz, e1, e2, u, self.n), self.n)

    def lincomb(self, vs, s):
        m = max(len(v) for v in vs)
        return self._np(self.cref.lincomb_powers(self.fid, [np.ascontiguousarray(v).tobytes() for v in vs], s, m), m)

    def fold_pairs(self, v, x):
        m = len(v) // 2
        return self._np(self.cref.field_bind(self.fid, v, 0, 1, 2, x, m), m)

    def poly_eval_multi(self, polys, us):
        return [[self.cref.suffix_horner(self.fid, f, len(f), u)[:32] for u in us] for f in polys]

    def div_by_monomial(self, B, u):
        return np.ascontiguousarray(self._np(self.cref.suffix_horner(self.fid, B, len(B), u), len(B))[1:])

    def ipa_generator(self):
        from oracle import pyref
        return self.cref.sequential_bases(pyref.CURVES_BY_ID[self.s.cid], IPA_GENERATOR_K0, 1).tobytes()

    def scale_point(self, U, r):
        out, _inf = self.cref.commit(self.s.cid, np.zeros((0, 32), np.uint8), self.key, 0, U, r)
        return bytes(out)

    def ipa(self, ckc, a, b, tr):
        """InnerProductArgument::prove with the key fold of pedersen.rs:484-497 (oracle/nova_ref.c ref_ipa_prove)"""
        n = len(a)
        return self.cref.ipa_prove(self.s.cid, self.key[:n], ckc, np.ascontiguousarray(a), np.ascontiguousarray(b), n,
                                   tr.fn_ipa(self.cref.IPA_TRANSCRIPT_FN), ctx=tr.ctx)


def relaxed_fold_sequence(be, side, tr, call):
    """sample_random_instance_witness (src/r1cs/mod.rs:786-830) + NIFSRelaxed::prove (src/nova/nifs.rs:118-175: commit_T_relaxed,
    src/r1cs/mod.rs:629-661; fold_relaxed, :1070-1107) on the running instance of `side`: what CompressedSNARK::prove runs once per
    curve before the SNARKs (src/nova/mod.rs:812-845).  Returns the folded (W, E, u, X) -- W and E stay where the provider keeps
    them -- and the three commitments."""
    le = lambda v: int(v).to_bytes(32, "little")
    num = lambda b: int.from_bytes(bytes(b), "little")
    p = side.p
    z2 = call("rand.z", lambda: be.concat_z(be.W2, side.u2, side.X2))
    E2 = call("rand.E", lambda: be.cross_term0(z2, side.u2))                              # r1cs/mod.rs:803-812
    # :815-818 `rayon::join(|| commit(W), || commit(E))`: two threads on the reference side; a provider that can takes them side by side
    if hasattr(be, "commit_pair"):
        cW2, cE2 = call("rand.commit_W_E", lambda: be.commit_pair((be.W2, side.r_W), (E2, side.r_E)))
    else:
        cW2 = call("rand.commit_W", lambda: be.commit(be.W2, side.r_W))
        cE2 = call("rand.commit_E", lambda: be.commit(E2, side.r_E))
    z1 = call("fold.z", lambda: be.concat_z(be.W1, side.u1, side.X1))                     # commit_T_relaxed :638-639
    Z = call("fold.vec_add", lambda: be.vec_add(z1, z2))                                  # :643-647
    u12 = np.frombuffer(le((num(side.u1) + num(side.u2)) % p), np.uint8).reshape(1, 32)   # :648
    if hasattr(be, "spmv_all"):
        AZ, BZ, CZ = call("fold.spmv_x3", lambda: be.spmv_all(Z))                         # :650 (one reference function: multiply_vec)
    else:
        AZ, BZ, CZ = (call("fold.spmv_x3", lambda j=j: be.spmv(j, Z)) for j in range(3))
    T = call("fold.cross_term2", lambda: be.cross_term2(AZ, BZ, CZ, be.E1, E2, u12))      # :652-659
    cT = call("fold.commit_T", lambda: be.commit(T, side.r_T))                            # :661
    for c in (cW2, cE2, cT):                                                              # nifs.rs:147-160 (the RO; here the stand-in)
        tr.absorb(c[0])
    r = tr.squeeze()
    W = call("fold.W", lambda: be.axpy(be.W1, be.W2, r))                                  # fold_relaxed :1082-1086
    E = call("fold.E", lambda: be.axpy2(be.E1, T, E2, r))                                 # :1087-1092  E1 + r T + r^2 E2
    rr = num(r)
    u = np.frombuffer(le((num(side.u1) + rr * num(side.u2)) % p), np.uint8).reshape(1, 32)
    X = np.frombuffer(le((num(side.X1) + rr * num(side.X2)) % p), np.uint8).reshape(1, 32)
    return {"W": W, "E": E, "u": u, "X": X, "r": r, "commitments": [cW2, cE2, cT], "keep": (z1, z2, Z, AZ, BZ, CZ, T, E2)}


def hyperkzg_sequence(be, ell, p, hat_P, point, tr, call):
    """EE::prove of HyperKZG (src/provider/hyperkzg.rs:926-1116) on a polynomial that is already where the provider keeps its
    vectors: ell - 1 pair folds (:1085-1095), batch_commit (:1100), r from the commitments (:1105), u = [r, -r, r^2] (:1106),
    the v matrix (:1049-1056), q from it (:1058), B (:1059), three openings (:1062-1065; here one batch_commit of the quotients)."""
    le = lambda v: int(v).to_bytes(32, "little")
    num = lambda b: int.from_bytes(bytes(b), "little")
    if hasattr(be, "fold_chain") and ell > 1:                                   # the whole loop :1085-1095 as one call
        xs = b"".join(bytes(point[ell - i - 1]) for i in range(ell - 1))
        polys = [hat_P] + list(call("ee.fold_pairs", lambda: be.fold_chain(hat_P, xs)))
    else:
        polys, cur = [hat_P], hat_P
        for i in range(ell - 1):
            cur = call("ee.fold_pairs", lambda c=cur, i=i: be.fold_pairs(c, point[ell - i - 1]))
            polys.append(cur)
    coms = call("ee.batch_commit", lambda: be.batch_commit(polys[1:]))
    for c in coms:
        tr.absorb(c[0])
    r = num(tr.squeeze())
    us = np.frombuffer(le(r) + le((p - r) % p) + le(r * r % p), np.uint8).reshape(3, 32)
    evals = call("ee.evals", lambda: be.poly_eval_multi(polys, us))
    tr.absorb(b"".join(b"".join(row) for row in evals))
    q = tr.squeeze()
    B = call("ee.batch_poly", lambda: be.lincomb(polys, q))
    hs = [call("ee.div_x3", lambda j=j: be.div_by_monomial(B, us[j:j + 1])) for j in range(3)]
    opens = call("ee.commit_opens", lambda: be.batch_commit(hs))
    return {"com": coms, "v": evals, "w": opens}


IPA_GENERATOR_K0 = 424243


def ipa_replay(args, torch):
    """The inner-product argument alone (EvaluationEngine::prove of the Pedersen / IPA engines, src/provider/ipa_pc.rs:69-82, 174-281) at
    n = 2^log2n on the secondary curve of the cycle (Grumpkin; --cycle pasta: Vesta): a and b resident in HBM, the registered key with its
    window tables, the native stand-in transcript.  The HIP path never folds the key (nova_amd/csrc/ipa.hpp); the oracle's leg folds it as
    the reference does (pedersen.rs:484-497).  Checked: every L, R and a_hat equal, and the proof against the reference's verifier
    equation (tests/ipa_common.py) when n <= 2^14."""
    import nova_amd
    from nova_amd import _lib
    from tests import standin, util
    ell = args.log2n if args.log2n and args.log2n <= 20 else 14
    n = 1 << ell
    cid = {"bn254": 1, "pasta": 3}[getattr(args, "cycle", "bn254")]
    ce = nova_amd.CommitmentEngine(cid)
    ck = ce.setup_synthetic(n, k0=7)
    gk = nova_amd.CommitmentKey.generate(cid, 1, k0=IPA_GENERATOR_K0)
    U = gk.read(0, 1).tobytes()
    gk.close()
    ha, hb, r0 = util.random_scalars(cid, n, seed=61), util.random_scalars(cid, n, seed=62), util.random_scalars(cid, 1, seed=63)
    da, db = torch.from_numpy(ha).cuda(), torch.from_numpy(hb).cuda()
    ckc = ce.commit(nova_amd.CommitmentKey(cid, ck.handle, ck.n, U), np.zeros((0, 32), np.uint8), r0).xy

    def run():
        tr = standin.Transcript(seed=SPARTAN_SEED)
        return nova_amd.ipa_prove(ck, ckc, da, db, tr.fn_ipa(_lib.IPA_TRANSCRIPT_FN), ctx=tr.ctx)
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    outj = {"metric": "InnerProductArgument::prove ms (one evaluation argument)", "value": dt * 1e3, "unit": "ms", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE,
            "data": "synthetic", "ms_per_round": round(dt * 1e3 / max(ell, 1), 4),
            "config": {"workload": f"InnerProductArgument::prove ({nova_amd.CURVE_NAMES[cid]}, n = 2^{ell}, {ell} rounds): per round one launch for the folds of "
                                   "the previous round, the expanded scalar vectors and the inner products, then ONE fused two-vector MSM over the "
                                   "registered key (the key is never folded); stand-in transcript"},
            "roofline": None}
    if not args.no_cpu_baseline:
        from oracle import cref, pyref
        threads = effective_cpus()
        cref.set_threads(threads)
        key = ck.read(0, n)
        t1 = time.perf_counter()
        tr = standin.Transcript(seed=SPARTAN_SEED)
        exp = cref.ipa_prove(cid, key, np.frombuffer(bytes(ckc), np.uint8).copy(), ha, hb, n, tr.fn_ipa(cref.IPA_TRANSCRIPT_FN), ctx=tr.ctx)
        t_cpu = time.perf_counter() - t1
        checks = {"proof": tuple(res) == tuple(exp)}
        if ell <= 14:
            import ctypes
            from tests import ipa_common as ic
            tv, rs = standin.Transcript(seed=SPARTAN_SEED), []
            for Lp, Rp, (Li, Ri) in zip(res[0], res[1], res[2]):
                o = (ctypes.c_uint8 * 32)()
                standin.lib().standin_ipa_transcript(tv.ctx, (ctypes.c_uint8 * 64)(*Lp), int(Li), (ctypes.c_uint8 * 64)(*Rp), int(Ri), o)
                rs.append(int.from_bytes(bytes(o), "little"))
            checks["reference_verifier"] = bool(ic.verify(pyref.CURVES_BY_ID[cid], key, np.frombuffer(bytes(ckc), np.uint8).copy(), ha, hb, n,
                                                          res[0], res[1], res[2], res[3], rs))
        outj["cpu_baseline"] = {"value": t_cpu * 1e3, "unit": "ms", "cores": threads, "kind": "port",
                                "sample": "the same argument once through oracle/nova_ref.c, key fold included (2 n scalar multiplications)",
                                "gpu_matches_cpu": all(checks.values()), "checks": checks}
    ck.close()
    return outj


def ipa_sequence(be, ell, poly, point, tr, call):
    """EE::prove of the inner-product engine (src/provider/ipa_pc.rs:69-82 -> InnerProductArgument::prove :174-281) on a polynomial that
    is already where the provider keeps its vectors: b = EqPolynomial::new(point).evals() (:78), r from the transcript (:186-190),
    ck_c.scale(&r) (:191), then the log n rounds -- c_L, c_R, the commitments L and R, the folds of a and b; the reference also folds the
    key (pedersen.rs:484-497: 2 n scalar multiplications per proof), the HIP path commits against the registered key instead
    (nova_amd/csrc/ipa.hpp) and the oracle does what the reference does."""
    b = call("ee.eq_evals", lambda: be.eq_evals(b"".join(bytes(x) for x in point)))
    r0 = tr.squeeze()
    state = bytes(tr.state.raw)                                    # (what a verifier's transcript holds here: tests replay the rounds from it)
    ckc = call("ee.scale_ck_c", lambda: be.scale_point(be.ipa_u, r0))
    Ls, Rs, infs, a_hat = call("ee.ipa", lambda: be.ipa(ckc, poly, b, tr))
    return {"ck_c": bytes(ckc), "L": Ls, "R": Rs, "inf": infs, "a_hat": a_hat, "tr_state": state}


def compressed_snark_sequence(beP, beS, sideP, sideS, call=lambda name, fn: fn(), parallel_snarks=False):
    """The provider-side work of CompressedSNARK::prove (src/nova/mod.rs:793-881) in the reference's order:
      secondary: sample_random_instance_witness + NIFSRelaxed::prove          (:812-826)
      primary:   sample_random_instance_witness + NIFSRelaxed::prove          (:829-843)
      S1::prove on the primary = RelaxedR1CSSNARK::prove (src/spartan/snark.rs:113-260): the sum-check sequence, whose batched
                 witness W + c E (spartan/mod.rs:429) is the polynomial EE::prove folds and opens (snark.rs:236-244)
      S2::prove on the secondary: the sum-check sequence, then ITS evaluation argument on its batched witness -- the inner-product
                 argument (src/provider/ipa_pc.rs:69-82, 174-281; ipa_sequence)
    NOT replayed: NIFS::prove of (:797-809) -- the prove_step replay's secondary fold --, the RO / Keccak transcripts (one stand-in
    per SNARK and one per fold), derandomize (:846-861: two scalar products of h per side).  parallel_snarks: S1::prove and S2::prove side
    by side on two host threads, as the reference's `rayon::join` at :862-881 runs them (every provider call leases its own context and
    stream); False: one after the other (the oracle's leg: its OpenMP loops already use every core)."""
    from tests import standin
    cp = lambda tag: (lambda nm, fn: call(f"{tag}.{nm}", fn))      # spans are kept per side: P = primary, S = secondary
    out = {}
    for tag, be, side in (("S", beS, sideS), ("P", beP, sideP)):
        out[f"fold_{tag}"] = relaxed_fold_sequence(be, side, standin.Transcript(seed=SPARTAN_SEED + 1), cp(tag))
    def snark(tag, be, side):
        f = out[f"fold_{tag}"]
        tr = standin.Transcript(seed=SPARTAN_SEED)
        sp = spartan_sequence(be, side.ell, side.p, f["u"], cp(tag), inst=(f["W"], f["E"], f["X"]), tr=tr)
        out[f"spartan_{tag}"] = sp
        if tag == "P":                                            # snark.rs:236-244: batched_w.p, batched_u.x = the batch sum-check's r
            out["ee_P"] = hyperkzg_sequence(be, side.ell, side.p, sp["batch_witness"], sp["batch"][1], tr, cp(tag))
        else:
            out["ee_S"] = ipa_sequence(be, side.ell, sp["batch_witness"], sp["batch"][1], tr, cp(tag))
    if not parallel_snarks:
        snark("P", beP, sideP)
        snark("S", beS, sideS)
        return out
    failed = []

    def second():
        try:
            snark("S", beS, sideS)
        except Exception as e:   # noqa: BLE001 -- re-raised on the calling thread
            failed.append(e)
    t = threading.Thread(target=second)
    t.start()
    try:
        snark("P", beP, sideP)
    finally:
        t.join()
    if failed:
        raise failed[0]
    return out


def csnark_digest(res, host):
    """what the two legs must agree on, byte for byte"""
    d = {}
    for tag in ("S", "P"):
        f, sp = res[f"fold_{tag}"], res[f"spartan_{tag}"]
        d[f"fold_{tag}.commitments"] = f["commitments"]
        d[f"fold_{tag}.r"] = f["r"]
        for k in ("outer", "inner", "batch", "evaluations"):
            d[f"spartan_{tag}.{k}"] = sp[k]
        d[f"spartan_{tag}.batch_witness"] = host[tag](sp["batch_witness"])
    d["ee_P"] = res["ee_P"]
    d["ee_S"] = res["ee_S"]
    return d


def compressed_snark_replay(args, torch):
    """REPLAY of CompressedSNARK::prove as ONE chained sequence (BASELINE.json configs[4]; see compressed_snark_sequence): primary
    BN254 at num_cons = 2^log2n, secondary Grumpkin at 2^log2n_secondary (default 14, the augmented circuit's size), every vector
    resident in HBM from the folds to the openings -- the folded witness is the Spartan instance, Spartan's batched witness is
    the polynomial HyperKZG folds and opens.  Checked against oracle/nova_ref.c run in the same order (every commitment, round
    polynomial, evaluation, the batched witnesses, the evaluation argument) and the reference's verifier equations on both Spartan
    proofs.  `trait_only`: the MSM calls of the same sequence in the form Nova makes them when ONLY the DlogGroupExt override of
    INTEGRATION.md section 2 is applied (slice-form nmx_msm / nmx_msm_batch, host scalars, bases through the slice cache; the field
    work stays on the reference's CPU and is not timed here).  A replay, not `prove`: the Rust reference cannot be built here."""
    import nova_amd
    from nova_amd import _lib
    L = _lib.lib()
    ellP, ellS = args.log2n, getattr(args, "log2n_secondary", None) or min(14, args.log2n)
    cP, cS = {"bn254": (0, 1), "pasta": (2, 3)}[getattr(args, "cycle", "bn254")]
    sides = {"P": CsnarkSide(cP, ellP, seed=900 + ellP), "S": CsnarkSide(cS, ellS, seed=950 + ellS)}
    cks = {k: nova_amd.CommitmentEngine(sd.cid).setup_synthetic(sd.n, k0=7) for k, sd in sides.items()}
    gpu = {k: GpuProvider(torch, sd, cks[k]) for k, sd in sides.items()}
    spans = None

    def call(name, fn):
        if spans is None:
            return fn()
        t = time.perf_counter()
        v = fn()
        spans.setdefault(name, []).append(time.perf_counter() - t)
        return v

    par = not getattr(args, "serial_snarks", False)
    run = lambda: compressed_snark_sequence(gpu["P"], gpu["S"], sides["P"], sides["S"], call, parallel_snarks=par)
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    spans = {}
    passes = 3
    for _ in range(passes):
        run()
    breakdown = {k: round(sum(v) / passes * 1e3, 4) for k, v in sorted(spans.items())}
    breakdown["_sum"] = round(sum(breakdown.values()), 4)
    groups = {}
    for k, v in breakdown.items():
        if k == "_sum":
            continue
        side, nm = k.split(".", 1)
        g = "fold" if nm.startswith(("rand.", "fold.")) else "ee" if nm.startswith("ee.") else "spartan"
        groups[f"{side}.{g}"] = round(groups.get(f"{side}.{g}", 0.0) + v, 4)
    spans = None
    verifies = {}
    for tag in ("P", "S"):
        for k, v in spartan_verify(sides[tag].p, res[f"fold_{tag}"]["u"], res[f"spartan_{tag}"]).items():
            verifies[f"{tag}.{k}"] = v
    outj = {
        "metric": "CompressedSNARK prove provider-call REPLAY ms (one chained sequence)", "value": dt * 1e3, "unit": "ms", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": False, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": f"CompressedSNARK::prove replay ({nova_amd.CURVE_NAMES[cP]} primary 2^{ellP}, {nova_amd.CURVE_NAMES[cS]} secondary 2^{ellS}): per side "
                               "random instance (3 SpMV + 2 commitments) + relaxed fold (3 SpMV, five-input cross term, commit T, two folds); Spartan "
                               "sum-check sequence on both folded instances; HyperKZG EE::prove on the primary's batched witness where it lies in HBM "
                               "(BASELINE.json configs[4]) and the inner-product argument on the secondary's; stand-in transcripts; S1 and S2 "
                               + ("side by side on two host threads (rayon::join, nova/mod.rs:862-881)" if par else "one after the other"),
                   "parallel_snarks": par},
        "roofline": None, "breakdown_ms": breakdown, "groups_ms": groups, "proof_verifies": verifies,
    }
    if not args.no_cpu_baseline:
        from oracle import cref
        threads = effective_cpus()
        cref.set_threads(threads)
        keys = {k: cks[k].read(0, sides[k].n) for k in sides}
        cpu = {k: CpuProvider(sides[k], keys[k], cks[k].h) for k in sides}
        t1 = time.perf_counter()
        exp = compressed_snark_sequence(cpu["P"], cpu["S"], sides["P"], sides["S"])
        t_cpu = time.perf_counter() - t1
        dg = csnark_digest(res, {k: gpu[k].host for k in gpu})
        de = csnark_digest(exp, {k: cpu[k].host for k in cpu})
        checks = {k: dg[k] == de[k] for k in dg}
        checks.update(verifies)
        outj["cpu_baseline"] = {"value": t_cpu * 1e3, "unit": "ms", "cores": threads, "kind": "port",
                                "sample": "the same chained sequence once through oracle/nova_ref.c",
                                "serial_parts": CPU_SERIAL_PARTS, "gpu_matches_cpu": all(checks.values()), "checks": checks}
        # the unchanged-caller form: every commitment of the sequence as the trait's slice-form call, host scalars and host bases
        outj["trait_only"] = trait_only_msms({k: (sides[k].cid, keys[k], cpu[k].msm_log, cpu[k].prep) for k in sides})
        if not getattr(args, "no_cpp_driver", False):
            for k in gpu:                                  # (the driver registers its own keys and matrices: give the HBM back first)
                gpu[k].close()
                cks[k].close()
            gpu = {}
            outj["cpp_driver"] = cpp_chained_replay(sides, exp, {k: cpu[k].host for k in cpu}, args.steps, args.warmup, serial_snarks=not par)
    for k in gpu:
        gpu[k].close()
        cks[k].close()
    return outj


CPP_DRIVER_SRC = os.path.join(ROOT, "bench", "csnark_replay.cpp")
CPP_DRIVER_BIN = os.path.join(ROOT, "bench", "csnark_replay.bin")


def build_cpp_driver(force=False):
    """g++ bench/csnark_replay.cpp (the chained replay driven from C++ through include/nova_mi355x.hpp): host code only, links the
    product library, the HIP runtime (device buffers) and the stand-in transcript.  Built by __graft_entry__.build() so that the
    binary travels to the GPU box."""
    import subprocess
    from tests import standin
    standin.build()
    deps = [CPP_DRIVER_SRC, os.path.join(ROOT, "include", "nova_mi355x.hpp"), os.path.join(ROOT, "include", "nova_mi355x.h")]
    if force or not os.path.exists(CPP_DRIVER_BIN) or os.path.getmtime(CPP_DRIVER_BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", CPP_DRIVER_BIN, CPP_DRIVER_SRC,
                               "-L" + os.path.join(ROOT, "nova_amd"), "-lnova_mi355x", "-L" + os.path.join(ROOT, "tests", "standin"),
                               "-lstandin_transcript", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-rpath,$ORIGIN/../nova_amd",
                               "-Wl,-rpath,$ORIGIN/../tests/standin", "-Wl,-rpath,/opt/rocm/lib"])
    return CPP_DRIVER_BIN


def _write_records(path, recs):
    import struct
    with open(path, "wb") as f:
        for name, data in recs.items():
            b = bytes(data)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<Q", len(b)) + b)


def _read_records(path):
    import struct
    out = {}
    with open(path, "rb") as f:
        buf = f.read()
    i = 0
    while i < len(buf):
        (nl,) = struct.unpack_from("<I", buf, i)
        name = buf[i + 4:i + 4 + nl].decode()
        (nb,) = struct.unpack_from("<Q", buf, i + 4 + nl)
        i += 12 + nl
        out[name] = buf[i:i + nb]
        i += nb
    return out


def cpp_chained_replay(sides, exp, cpu_host, steps, warmup, k0=7, serial_snarks=False):
    """The same chained sequence driven from C++ (bench/csnark_replay.cpp through include/nova_mi355x.hpp) on the same instance: its
    wall time per sequence, and every output compared with the oracle's run `exp` (csnark_digest)."""
    import struct
    import subprocess
    import tempfile
    recs = {}
    for tag, sd in sides.items():
        recs[f"{tag}.meta"] = np.array([sd.cid, sd.fid, sd.ell, k0], np.uint64).tobytes()
        for j, (ip, ix, dt) in enumerate(sd.csr):
            recs[f"{tag}.ip{j}"] = np.ascontiguousarray(ip, np.uint64).tobytes()
            recs[f"{tag}.ix{j}"] = np.ascontiguousarray(ix, np.uint64).tobytes()
            recs[f"{tag}.dt{j}"] = np.ascontiguousarray(dt).tobytes()
        for nm in ("W1", "W2", "E1", "u1", "X1", "u2", "X2", "r_W", "r_E", "r_T"):
            recs[f"{tag}.{nm}"] = np.ascontiguousarray(getattr(sd, nm)).tobytes()
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "instance.bin"), os.path.join(td, "out.bin")
        _write_records(fin, recs)
        # (at least three untimed sequences: contexts the worker threads lease for the first time still grow their workspaces then)
        r = subprocess.run([build_cpp_driver(), fin, fout, str(steps), str(max(warmup, 3)), "1" if serial_snarks else "0"], capture_output=True, text=True)
        if r.returncode != 0:
            return {"error": f"rc {r.returncode}: {r.stderr[-300:]}"}
        got = _read_records(fout)
    pts = lambda b: [(bytes(b[65 * i:65 * i + 64]), int(b[65 * i + 64])) for i in range(len(b) // 65)]
    flat = lambda b: [bytes(b[32 * i:32 * i + 32]) for i in range(len(b) // 32)]
    rows = lambda b, w: [flat(b[32 * w * j:32 * w * (j + 1)]) for j in range(len(b) // (32 * w))]
    dg = {}
    for tag in ("S", "P"):
        dg[f"fold_{tag}.commitments"] = [pts(got[f"{tag}.fold.{k}"])[0] for k in ("cW2", "cE2", "cT")]
        dg[f"fold_{tag}.r"] = bytes(got[f"{tag}.fold.r"])
        for k, w in (("outer", 4), ("inner", 3), ("batch", 3)):
            dg[f"spartan_{tag}.{k}"] = (rows(got[f"{tag}.spartan.{k}.polys"], w), flat(got[f"{tag}.spartan.{k}.r"]), flat(got[f"{tag}.spartan.{k}.claims"]))
        dg[f"spartan_{tag}.evaluations"] = flat(got[f"{tag}.spartan.evaluations"])
        dg[f"spartan_{tag}.batch_witness"] = bytes(got[f"{tag}.spartan.batch_witness"])
    v = flat(got["P.ee.v"])
    dg["ee_P"] = {"com": pts(got["P.ee.com"]), "v": [v[3 * i:3 * i + 3] for i in range(len(v) // 3)], "w": pts(got["P.ee.w"])}
    eS = {k: bytes(got[f"S.ee.{k}"]) for k in ("ck_c", "L", "R", "inf", "a_hat", "tr_state")}
    dg["ee_S"] = {"ck_c": eS["ck_c"], "L": [eS["L"][64 * i:64 * i + 64] for i in range(len(eS["L"]) // 64)],
                  "R": [eS["R"][64 * i:64 * i + 64] for i in range(len(eS["R"]) // 64)],
                  "inf": [[eS["inf"][2 * i], eS["inf"][2 * i + 1]] for i in range(len(eS["inf"]) // 2)], "a_hat": eS["a_hat"],
                  "tr_state": eS["tr_state"]}
    de = csnark_digest(exp, cpu_host)

    def plain(x):                                       # one shape on both sides: nested lists of bytes / ints
        if isinstance(x, (bytes, bytearray, memoryview, np.ndarray)):
            return bytes(x)
        if isinstance(x, dict):
            return {k: plain(val) for k, val in x.items()}
        if isinstance(x, (list, tuple)):
            return [plain(val) for val in x]
        return int(x) if isinstance(x, (int, np.integer)) else x
    checks = {k: plain(dg[k]) == plain(de[k]) for k in de}
    grp = struct.unpack("<6d", got["ms_per_group"])
    per_step = struct.unpack(f"<{len(got['ms_per_step']) // 8}d", got["ms_per_step"])
    return {"ms": round(struct.unpack("<d", got["ms_per_sequence"])[0], 4), "steps": steps, "gpu_matches_cpu": all(checks.values()),
            "median_ms": round(float(np.median(per_step)), 4), "per_step_ms": [round(x, 3) for x in per_step],
            "groups_ms": dict(zip(("S.fold", "P.fold", "P.spartan", "P.ee", "S.spartan" if serial_snarks else "S.after_P", "S.ee"), (round(g, 4) for g in grp))),
            "failed": [k for k, ok in checks.items() if not ok],
            "what": "bench/csnark_replay.cpp: the same provider calls in the same order through include/nova_mi355x.hpp (namespace resident), "
                    "device buffers allocated once, no Python between the calls"}


# what of the oracle's side of a replay still runs on ONE host thread (round 6: the transposed product, evaluations, eq tables, batch
# witness, Horner / division passes and the pair product went OpenMP, as the reference's rayon code is: spartan/mod.rs:497-533,
# polys/multilinear.rs:98-180, polys/eq.rs:54-73, hyperkzg.rs:961-999)
CPU_SERIAL_PARTS = ["the transcript callback and the O(1) algebra of a sum-check round", "the integer counting sort inside the transposed product",
                    "vectors shorter than 4 096 elements (sum-check tables of the late rounds, eq tables of the short side)"]


def trait_only_msms(logs, reps=5):
    """The commitments of a replayed sequence as Nova issues them through `DlogGroupExt` ALONE (only the
    trait override applied, nothing in r1cs/mod.rs / nifs.rs / snark.rs patched): `vartime_multiscalar_mul(&[Scalar], &ck[..n])` =
    nmx_msm and `batch_vartime_multiscalar_mul` = nmx_msm_batch over pageable HOST scalars and the caller's HOST base array (resident
    through the slice cache after first sight, window tables from the third use on, content re-verified on every call).  The field
    work between the commitments stays on the reference's CPU; what is timed is the sum of the provider calls.
    logs: {side: (curve, host base array, [(kind, [host vectors], expected results or None)] in call order, oracle key)}."""
    import nova_amd
    from nova_amd import _lib
    _lib.lib().nmx_cache_clear()
    groups = {k: nova_amd.DlogGroup(v[0]) for k, v in logs.items()}

    def one_pass(record, results):
        t_all = 0.0
        for k, (cid, bases, log, _prep) in logs.items():
            for i, (kind, vs, _exp) in enumerate(log):
                t = time.perf_counter()
                if kind == "commit":
                    got = [groups[k].vartime_multiscalar_mul(vs[0], bases[:len(vs[0])])]
                else:
                    got = groups[k].batch_vartime_multiscalar_mul(vs, bases[:max(len(v) for v in vs)])
                d = time.perf_counter() - t
                t_all += d
                if record is not None:
                    lens = ",".join(str(len(v)) for v in vs[:3]) + (",.." if len(vs) > 3 else "")
                    record.setdefault(f"{k}.{i}.{kind}[{lens}]", []).append(d)
                    results[(k, i)] = [(g.xy, int(g.is_inf)) for g in got]
        return t_all
    for _ in range(4):                                        # first sight, second use, tables at the third, one warm pass
        one_pass(None, None)
    rec, results, total = {}, {}, []
    for _ in range(reps):
        total.append(one_pass(rec, results))
    ok = True
    for k, (cid, bases, log, prep) in logs.items():           # against the oracle (blinded commitments: their unblinded MSM, computed here)
        for i, (kind, vs, exp) in enumerate(log):
            exp = exp if exp is not None else [prep.msm(v, len(v)) for v in vs]
            ok = ok and results[(k, i)] == exp
    st = _lib.stats()
    return {"ms": round(float(np.median(total)) * 1e3, 4), "calls": sum(len(v[2]) for v in logs.values()),
            "gpu_matches_cpu": ok,
            "per_call_ms": {k: round(float(np.median(v)) * 1e3, 4) for k, v in rec.items()},
            "cache": {"uploads": st[_lib.STAT_CACHE_UPLOADS], "hits": st[_lib.STAT_CACHE_HITS], "stale": st[_lib.STAT_CACHE_STALE]},
            "what": "sum of the sequence's MSM provider calls in slice form (nmx_msm / nmx_msm_batch): pageable host scalars, host bases via the "
                    "slice cache (content verified every call); the field work between them is the reference's own CPU code and is not timed"}


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
    if wl == "spmv":  # n rows x n columns, 3 random non-zeros per row (the minroot shape: 3 constraints per iteration);
        # coefficients as R1CS matrices have them: +1 / -1 in nine entries of ten, full-width otherwise (--dist random:
        # every coefficient full-width, the worst case for the class encoding of nova_amd/csrc/fieldvec.hip)
        rng = np.random.Generator(np.random.PCG64(5))
        indptr = np.arange(0, 3 * n + 1, 3, dtype=np.uint64)
        indices = rng.integers(0, n, size=3 * n).astype(np.uint64)
        data = util.random_scalars(cid, 3 * n, seed=4).copy()
        if args.dist != "random_coeffs":
            kind = rng.random(3 * n)
            data[kind < 0.6] = util.int_to_le32(1)
            data[(kind >= 0.6) & (kind < 0.9)] = util.int_to_le32(util.MODULI[cid] - 1)
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
            "dtype": DTYPE, "data": "synthetic",
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
