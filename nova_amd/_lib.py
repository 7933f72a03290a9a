"""ctypes binding of include/nova_mi355x.h.  Fails loudly when the HIP library has not been built."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("NMX_SO") or os.path.join(_HERE, "libnova_mi355x.so")  # NMX_SO: an A/B build of the same library

# flags / error codes (include/nova_mi355x.h)
SCALARS_MONT, BASES_MONT, SCALARS_DEVICE, BASES_DEVICE, OUT_PARTIAL, BASES_PRECOMPUTE, BASES_VALIDATE = 1, 2, 4, 8, 16, 32, 64
BASES_NOCACHE = 128
SCALARS_SHARDED = 256
ASYNC = 512
OP_AXPY, OP_AXPY2, OP_CROSS_TERM, OP_CROSS_TERM2, OP_VEC_ADD = range(5)
BRANCH_NAMES = {0: "host", 1: "local", 2: "peer_copy", 3: "shard_resident", 4: "none"}
PROF_STAGES = 12
# nmx_stats indices
(STAT_CACHE_HITS, STAT_CACHE_UPLOADS, STAT_CACHE_REGROWS, STAT_CACHE_EVICTIONS, STAT_CACHE_ENTRIES, STAT_CACHE_BYTES,
 STAT_UNCACHED_CALLS, STAT_BASE_BYTES_H2D, STAT_MSM_CALLS, STAT_FUSED_RUNS, STAT_SHARDED_CALLS, STAT_CACHE_STALE,
 STAT_TABLE_FALLBACKS, STAT_LAUNCH_GAP_NS, STAT_SCAN_TIMEOUTS, STAT_SC_TORN_INJECTED, STAT_SC_TORN_REJECTS,
 STAT_COUNT) = range(18)
DEVICES_OVERSUBSCRIBE = 1
E_ARG, E_NO_DEVICE, E_HIP, E_SCALAR_RANGE, E_SMALL_RANGE, E_HANDLE, E_TOO_LARGE = -1, -2, -3, -4, -5, -6, -7
E_IO, E_FORMAT, E_POINT, E_ZERO = -8, -9, -10, -11
BITS_AUTO = 0xFFFFFFFF

# nmx_transcript_fn: (ctx, round polynomial coefficients, how many, challenge out) -> 0
TRANSCRIPT_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t,
                                 ctypes.POINTER(ctypes.c_uint8))
# nmx_ipa_transcript_fn: (ctx, L xy64, L is the identity, R xy64, R is the identity, challenge out) -> 0
IPA_TRANSCRIPT_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_uint8), ctypes.c_int, ctypes.POINTER(ctypes.c_uint8))

_lib = None


def lib():
    """The loaded library.  No fallback: a missing .so is an error (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: build the HIP extension first (__graft_entry__.build()); "
            "nova_amd has no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.  If this library pulled in
    # /opt/rocm's copy first, a later `import torch` would bring up a second runtime that cannot see the GPU
    # ("No HIP GPUs are available").  Loading torch first makes both share torch's copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(SO_PATH)
    i, u32, u64, sz, vp = ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_void_p
    L.nmx_init.argtypes = [i]
    L.nmx_shutdown.argtypes = []
    L.nmx_device_count.argtypes = []
    L.nmx_sync.argtypes = []
    L.nmx_init_devices.argtypes = [i, u32]
    L.nmx_devices_in_use.argtypes = []
    L.nmx_shard_plan.argtypes = [sz, i, sz, sz, vp, i]
    L.nmx_bases_shard_plan.argtypes = [u64, sz, sz, vp, i, ctypes.POINTER(sz)]
    L.nmx_last_error.restype = ctypes.c_char_p
    L.nmx_version.restype = ctypes.c_char_p
    L.nmx_bases_register.argtypes = [i, vp, sz, u32, ctypes.POINTER(u64)]
    L.nmx_bases_unregister.argtypes = [u64]
    L.nmx_bases_read.argtypes = [u64, sz, sz, vp]
    L.nmx_bases_generate.argtypes = [i, u64, sz, u32, ctypes.POINTER(u64)]
    L.nmx_bases_register_ptau.argtypes = [i, ctypes.c_char_p, sz, sz, u32, ctypes.POINTER(u64)]
    L.nmx_bases_register_keyfile.argtypes = [i, ctypes.c_char_p, sz, u32, ctypes.POINTER(u64), vp]
    L.nmx_msm.argtypes = [i, vp, vp, sz, u32, vp, vp]
    L.nmx_msm_handle.argtypes = [u64, sz, vp, sz, u32, vp, vp]
    L.nmx_msm_u64.argtypes = [i, vp, vp, sz, u32, u32, vp, vp]
    L.nmx_msm_u64_handle.argtypes = [u64, sz, vp, sz, u32, u32, vp, vp]
    L.nmx_msm_sparse_handle.argtypes = [u64, vp, vp, sz, u32, vp, vp]
    L.nmx_msm_batch.argtypes = [i, vp, vp, sz, vp, sz, u32, vp, vp]
    L.nmx_msm_batch_handle.argtypes = [u64, vp, vp, sz, u32, vp, vp]
    L.nmx_msm_u64_batch.argtypes = [i, vp, vp, sz, vp, sz, u32, u32, vp, vp]
    L.nmx_msm_u64_batch_handle.argtypes = [u64, vp, vp, sz, u32, u32, vp, vp]
    L.nmx_commit.argtypes = [u64, vp, sz, vp, vp, u32, vp, vp]
    L.nmx_commit_begin.argtypes = [u64, vp, sz, vp, vp, u32, ctypes.POINTER(ctypes.c_uint64)]
    L.nmx_commit_finish.argtypes = [u64, vp, vp]
    L.nmx_point_sum.argtypes = [i, vp, sz, vp, vp]
    L.nmx_svec_alloc.argtypes = [sz, sz, ctypes.POINTER(u64)]
    L.nmx_svec_free.argtypes = [u64]
    L.nmx_svec_write.argtypes = [u64, vp]
    L.nmx_svec_read.argtypes = [u64, vp]
    L.nmx_svec_parts.argtypes = [u64, vp, vp, vp, i]
    L.nmx_svec_map.argtypes = [i, i, vp, i, vp, u32, u64]
    L.nmx_msm_svec.argtypes = [u64, u64, sz, u32, vp, vp]
    L.nmx_commit_svec.argtypes = [u64, u64, sz, vp, vp, u32, vp, vp]
    L.nmx_profile_last_sharded.argtypes = [vp, vp, vp, i, vp, vp]
    L.nmx_field_axpy.argtypes = [i, vp, vp, vp, sz, u32, vp]
    L.nmx_field_axpy2.argtypes = [i, vp, vp, vp, vp, sz, u32, vp]
    L.nmx_field_cross_term.argtypes = [i, vp, vp, vp, vp, vp, sz, u32, vp]
    L.nmx_field_cross_term2.argtypes = [i, vp, vp, vp, vp, vp, vp, sz, u32, vp]
    L.nmx_field_vec_add.argtypes = [i, vp, vp, sz, u32, vp]
    L.nmx_mle_bind_top.argtypes = [i, vp, sz, vp, u32, vp]
    L.nmx_poly_fold_pairs.argtypes = [i, vp, sz, vp, u32, vp]
    L.nmx_sumcheck_eq_sums.argtypes = [i, i, vp, vp, vp, sz, vp, sz, vp, sz, u32, u32, vp]
    L.nmx_poly_suffix_horner.argtypes = [i, vp, sz, vp, u32, vp]
    L.nmx_eq_evals_from_points.argtypes = [i, vp, sz, u32, vp]
    L.nmx_mle_evaluate.argtypes = [i, vp, sz, vp, sz, u32, vp]
    L.nmx_spmv_register.argtypes = [i, vp, vp, vp, sz, sz, u32, ctypes.POINTER(u64)]
    L.nmx_spmv_unregister.argtypes = [u64]
    L.nmx_spmv_apply.argtypes = [u64, vp, sz, u32, vp]
    L.nmx_r1cs_cross_term.argtypes = [u64, u64, u64, vp, vp, sz, vp, vp, u32, vp]
    L.nmx_nifs_fold.argtypes = [i, vp, vp, sz, vp, vp, sz, vp, u32, vp, vp]
    L.nmx_spmv_apply_pair.argtypes = [u64, vp, vp, sz, u32, vp, vp]
    L.nmx_poly_fold_chain.argtypes = [ctypes.c_int, vp, sz, vp, sz, u32, vp]
    L.nmx_spmv_apply_many.argtypes = [vp, sz, ctypes.c_int, vp, sz, u32, vp]
    L.nmx_sumcheck_plain_sums.argtypes = [i, i, vp, vp, vp, sz, u32, vp]
    L.nmx_poly_eval_multi.argtypes = [i, vp, vp, sz, vp, sz, u32, vp]
    L.nmx_sumcheck_bind_eq_sums.argtypes = [i, i, vp, vp, vp, sz, vp, vp, sz, vp, sz, u32, u32, vp, vp, vp, vp]
    L.nmx_field_lincomb_powers.argtypes = [i, vp, vp, sz, vp, sz, u32, vp]
    L.nmx_mle_multi_evaluate.argtypes = [i, vp, sz, sz, vp, sz, u32, vp]
    L.nmx_spmv_apply_transposed.argtypes = [u64, vp, sz, u32, vp]
    L.nmx_field_concat.argtypes = [i, vp, vp, u64, sz, sz, u32, vp]
    L.nmx_field_batch_invert.argtypes = [i, vp, sz, u32, vp]
    L.nmx_sumcheck_prove_cubic_with_three_inputs.argtypes = [i, vp, vp, sz, vp, vp, vp, u32, TRANSCRIPT_FN, vp, vp, vp, vp]
    L.nmx_sumcheck_prove_quad_prod.argtypes = [i, vp, sz, vp, vp, u32, TRANSCRIPT_FN, vp, vp, vp, vp]
    L.nmx_sumcheck_prove_batch_eval.argtypes = [i, vp, vp, vp, vp, vp, sz, u32, TRANSCRIPT_FN, vp, vp, vp, vp]
    L.nmx_ipa_prove.argtypes = [u64, vp, vp, vp, sz, u32, IPA_TRANSCRIPT_FN, vp, vp, vp, vp, vp]
    L.nmx_set_profiling.argtypes = [i]
    L.nmx_profile_last.argtypes = [ctypes.POINTER(ctypes.c_float), i]
    L.nmx_set_window_bits.argtypes = [u32]
    L.nmx_set_option.argtypes = [ctypes.c_char_p, u32]
    L.nmx_cache_clear.argtypes = []
    L.nmx_cache_invalidate.argtypes = [vp]
    L.nmx_cache_configure.argtypes = [sz, sz, sz]
    L.nmx_min_gpu_n.argtypes = [i]
    L.nmx_min_gpu_n.restype = sz
    L.nmx_check_layout.argtypes = [i, vp, vp, u64]
    L.nmx_stats.argtypes = [ctypes.POINTER(u64), i]
    _lib = L
    return L


def profile_last_sharded(cap=64):
    """nmx_profile_last_sharded of the calling thread -> {"shards": [{"dev", "branch", "stages_ms"}], "combine_ms", "rccl_ranks"}."""
    ms = (ctypes.c_float * (PROF_STAGES * cap))()
    dev = (ctypes.c_int * cap)()
    br = (ctypes.c_int * cap)()
    cms = ctypes.c_float(0)
    ranks = ctypes.c_int(0)
    n = lib().nmx_profile_last_sharded(ms, dev, br, cap, ctypes.byref(cms), ctypes.byref(ranks))
    return {"shards": [{"dev": dev[j], "branch": BRANCH_NAMES.get(br[j], str(br[j])),
                        "stages_ms": [round(ms[j * PROF_STAGES + q], 4) for q in range(7)]} for j in range(min(n, cap))],
            "combine_ms": round(cms.value, 4), "rccl_ranks": ranks.value}


def stats():
    """nmx_stats as a list indexed by the STAT_* constants."""
    buf = (ctypes.c_uint64 * STAT_COUNT)()
    lib().nmx_stats(buf, STAT_COUNT)
    return list(buf)
