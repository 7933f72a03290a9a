/* nova_mi355x.h -- C ABI of libnova_mi355x.so: the MI355X (gfx950) commitment / MSM provider for Nova.
 *
 * Drop-in boundary.  Each entry point names the reference interface it replaces; paths are
 * relative to the reference tree (microsoft/Nova, nova-snark 0.75.0).  The Rust-side binding a maintainer
 * would add (an external `nova-mi355x-sys` shim crate, wired exactly where the optional CUDA `blitzar`
 * backend is wired: src/provider/bn256_grumpkin.rs:43-78, src/provider/blitzar.rs:7-40) is in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns 0 on success, a negative NMX_E_* otherwise, and
 *     nmx_last_error() gives a thread-local message.  A failed call never writes a (possibly wrong) point.
 *   - field elements: 32 bytes.  Default = canonical little-endian integer < modulus, i.e. the bytes of
 *     `to_repr()` / `to_bytes()` (what blitzar.rs:10 sends).  With the *_MONT flags = the raw in-memory
 *     4 x u64 Montgomery limbs (R = 2^256) of halo2curves (what `SerdeObject::write_raw` moves, ptau.rs:205).
 *   - affine point: x || y, 64 bytes; the identity is the all-zero encoding in both forms
 *     (src/provider/traits.rs:303-312 returns (0, 0, true)).
 *   - results: affine canonical x || y plus an `is_inf` byte -- exactly `to_coordinates()`.
 *   - all functions are thread-safe and re-entrant (the trait methods are static and are called from rayon
 *     worker threads concurrently: src/r1cs/mod.rs:509-512, src/provider/hyperkzg.rs:1062-1065).
 *   - inputs are borrowed for the duration of the call only.
 */
#ifndef NOVA_MI355X_H
#define NOVA_MI355X_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* curve ids: src/provider/bn256_grumpkin.rs:26-33 (bn256, grumpkin), src/provider/pasta.rs:24-31 */
enum { NMX_BN254_G1 = 0, NMX_GRUMPKIN = 1, NMX_PALLAS = 2, NMX_VESTA = 3, NMX_NUM_CURVES = 4 };

/* flags */
enum {
  NMX_SCALARS_MONT = 1u << 0,   /* scalars are raw Montgomery limbs instead of canonical LE            */
  NMX_BASES_MONT = 1u << 1,     /* base coordinates are raw Montgomery limbs instead of canonical LE     */
  NMX_SCALARS_DEVICE = 1u << 2, /* `scalars` is a device (HBM) pointer on the library's device           */
  NMX_BASES_DEVICE = 1u << 3,   /* `bases` is a device pointer (nmx_bases_register only)                 */
  NMX_BASES_PRECOMPUTE = 1u << 5, /* nmx_bases_register / nmx_bases_generate: also build the window tables          */
  NMX_BASES_VALIDATE = 1u << 6,   /* nmx_bases_register*: reject coordinates >= p and points off the curve with
                                     NMX_E_POINT (identity (0,0) passes) -- what read_points enforces on loaded
                                     keys, /root/reference/src/provider/ptau.rs:372-391                            */
                                /* 2^(c*w) * P_i in HBM (W x the key size; c = 17, W = 15 for keys of 2^20 points).    */
                                /* MSMs over >= 4096 points of such a key run all windows into one bucket set. */
  NMX_BASES_NOCACHE = 1u << 7,  /* slice-form calls (nmx_msm, nmx_msm_u64, nmx_msm_batch): do not look the base  */
                                /* array up in / insert it into the slice cache (one-shot arrays)               */
  NMX_SCALARS_SHARDED = 1u << 8, /* shard-resident scalars (nmx_msm_handle, nmx_msm_u64_handle with an explicit max_num_bits,
                                    nmx_commit, nmx_msm_batch_handle): `scalars` is a HOST array of device pointers, one per
                                    piece that nmx_bases_shard_plan(handle, offset, n) reports (= nmx_shard_plan(REGISTERED key
                                    length, devices, offset, n)), in its order; every piece must be device memory on its shard's
                                    GPU and at least count_j scalars long (checked: NMX_E_ARG otherwise); piece j = the `count_j` scalars of the pairs whose bases live on `device_j`, in
                                    the HBM of THAT device -- the reference chunks coefficients and bases together
                                    (src/provider/msm.rs:564-574), so nothing crosses xGMI or PCIe inside the call.  A key on
                                    one device has one piece.  nmx_svec_* allocates vectors in this layout.          */
  NMX_ASYNC = 1u << 9,          /* element-wise field kernels and SpMV on HBM-resident vectors (nmx_field_axpy / _axpy2 /
                                   _cross_term / _cross_term2 / _vec_add, nmx_mle_bind_top, nmx_poly_fold_pairs, nmx_spmv_apply[_pair],
                                   nmx_r1cs_cross_term, nmx_nifs_fold; ignored elsewhere and with
                                   host operands): return once the kernel is ENQUEUED.  The calling host thread's later calls
                                   -- any entry point -- are ordered behind it, and every call that is synchronous (all MSMs
                                   and commitments, all reductions, anything with a host operand) still returns with
                                   everything the thread enqueued before it complete; nmx_sync() waits explicitly.  A vector
                                   written by an asynchronous call must not be handed to ANOTHER host thread, freed or
                                   read by the host before one of the two.  The NIFS chain between two commitments --
                                   Z = W1 + W2, AZ / BZ / CZ, T (src/r1cs/mod.rs:590-622), the folds (1044-1107) -- is the
                                   intended use: five dependent launches with no host wake-up between them.          */
  NMX_OUT_PARTIAL = 1u << 4     /* write a 128-byte partial sum instead of an affine point: the per-GPU   */
                                /* result of a sharded MSM, input of nmx_point_sum.  Format: extended     */
                                /* Jacobian (X, Y, ZZ, ZZZ), x = X/ZZ, y = Y/ZZZ, each coordinate the     */
                                /* 32-byte LE integer v * 2^261 mod p (the library's internal residue     */
                                /* form), identity <=> ZZ == 0.                                           */
};

/* error codes */
enum {
  NMX_OK = 0,
  NMX_E_ARG = -1,          /* null pointer / bad curve id / bad flags / length mismatch                  */
  NMX_E_NO_DEVICE = -2,    /* no usable HIP device: the library never falls back to a CPU path            */
  NMX_E_HIP = -3,          /* a HIP runtime call failed (message has the hipError string)                 */
  NMX_E_SCALAR_RANGE = -4, /* a canonical scalar >= modulus (from_repr would reject it)                   */
  NMX_E_SMALL_RANGE = -5,  /* a small scalar >= 2^max_num_bits (msm.rs:543-552 would index out of bounds)  */
  NMX_E_HANDLE = -6,       /* unknown / stale base handle, or offset + n beyond the registered key       */
  NMX_E_TOO_LARGE = -7,    /* n * windows >= 2^32                                                         */
  NMX_E_IO = -8,           /* key file cannot be opened / read (PtauFileError::IoError)                    */
  NMX_E_FORMAT = -9,       /* key file header rejected (InvalidHead, UnsupportedVersion, InvalidNumSections,
                              InvalidPrime, InsufficientPowerForG1/G2; ptau.rs:104-151)                    */
  NMX_E_POINT = -10,       /* a loaded point is not canonical or not on the curve (PointNotOnCurve)        */
  NMX_E_ZERO = -11         /* nmx_field_batch_invert: an element is zero (NovaError::InternalError, spartan/mod.rs:103-105) */
};

/* ---- lifetime ------------------------------------------------------------------------------------- */
/* Selects the HIP device for this process (one process per GPU) and creates the stream/workspace pool.
 * Idempotent.  device < 0 => honour LOCAL_RANK, else device 0.  No reference counterpart (the reference is
 * single-address-space); corresponds to blitzar's implicit backend init. */
int nmx_init(int device);
int nmx_shutdown(void);
int nmx_device_count(void);
int nmx_sync(void); /* waits for the calling thread's NMX_ASYNC calls (no-op when there are none) */
/* Several GPUs behind ONE host process (the reference is one address space: its own MSM decomposition is in-process,
 * `par_chunks` + `reduce(identity, +)`, src/provider/msm.rs:564-574,664-676).  After
 * nmx_init_devices(k, flags) the process owns logical devices 0 .. k-1 (HIP devices 0 .. k-1; k == 0: every visible
 * device; k > nmx_device_count(): NMX_E_NO_DEVICE unless NMX_DEVICES_OVERSUBSCRIBE, which maps logical device i to HIP
 * device i mod count -- several shards per GPU, for tests on a 1-GPU box).  Every key of at least `shard_min_n` points
 * (default 2^20, env NMX_SHARD_MIN_N, nmx_set_option("shard_min_n")) registered AFTERWARDS -- nmx_bases_register*,
 * nmx_bases_generate, and the slice cache behind nmx_msm / nmx_msm_u64 / nmx_msm_batch -- is cut into k contiguous shards
 * (shard i = points [i*n/k ...), exactly nova_amd/dist.py shard_range), shard i resident on device i with its own window
 * tables.  An MSM / commit over such a key runs one host thread + stream per shard touched, each producing a 128-byte
 * partial.  The combine step: inside ONE process every shard worker hands its partial to the calling thread
 * in host memory, so the default is the host sum of the k partials (nmx_point_sum: <= 8 additions of ~0.5 us, nothing on a
 * stream).  nmx_set_option("combine", 2) routes it through RCCL instead -- ONE ncclAllGather of the 128-byte partials over
 * xGMI (bound at run time; one rank per GPU, issued by the calling thread inside a group call), then the G-term sum from rank
 * 0's copy -- which exercises the communicator but adds two copies and k stream syncs; the all-gather is the real exchange
 * step in the one-process-per-GPU deployment (nova_amd/dist.py).  No bucket array crosses devices, and the call is still one
 * synchronous C call.
 * Scalars: shard-resident (NMX_SCALARS_SHARDED, nmx_svec_*: each piece already in the HBM of the GPU that holds its bases --
 * the intended form), or one HBM array on logical device 0 (NMX_SCALARS_DEVICE: a shard on another GPU pulls its slice
 * peer-to-peer over xGMI inside the call), or host memory (each GPU pulls its slice over its own PCIe link).
 * Field-vector kernels on plain pointers, keys below the threshold and keys registered from a device pointer stay on logical
 * device 0; nmx_svec_map runs the NIFS kernels shard by shard.  Without this call (or env NMX_DEVICES=k) the library uses one device, as
 * before.  May be called again to change k (keys keep the layout they were registered with). */
#define NMX_DEVICES_OVERSUBSCRIBE 1u
int nmx_init_devices(int count, uint32_t flags);
int nmx_devices_in_use(void); /* logical devices new keys are sharded over (1 = unsharded) */
/* How a key of n_key points is laid out over k devices and which shards a call over key[offset, offset + n) touches:
 * writes up to cap triples (device, offset inside the shard, count) and returns how many the call needs.  Pure host
 * arithmetic (no device needed): the rule the library itself uses. */
int nmx_shard_plan(size_t n_key, int k, size_t offset, size_t n, size_t* out_triples, int cap);
/* The same triples for a REGISTERED key, from the layout it actually has (a key keeps the layout of the device set it was
 * registered under; an unsharded key answers one triple), plus its registered length in *n_key (may be NULL).  This is the
 * plan to cut NMX_SCALARS_SHARDED pieces by: a plan computed from a length that is not the registered one (e.g. a key
 * registered with its blinding point appended has n + 1 points) puts the cut in the wrong place.  Returns the number of
 * triples or a negative error. */
int nmx_bases_shard_plan(uint64_t handle, size_t offset, size_t n, size_t* out_triples, int cap, size_t* n_key);
const char* nmx_last_error(void);
const char* nmx_version(void);

/* ---- commitment-key residency ------------------------------------------------------------------------
 * The reference passes bases as a prefix slice of a long-lived key (`&ck.ck[..v.len()]`,
 * src/provider/pedersen.rs:267, src/provider/hyperkzg.rs:588).  Register the whole key once; every later
 * call addresses bases[offset .. offset + n) of it in HBM.  `handle` 0 is never valid. */
int nmx_bases_register(int curve, const void* bases_xy64, size_t n, uint32_t flags, uint64_t* handle);
/* On-disk keys.  Points in both formats are halo2curves `write_raw` records: x || y as raw
 * R = 2^256 Montgomery limbs, 64 bytes -- the layout NMX_BASES_MONT takes, so the file is streamed to HBM through
 * pinned staging buffers and converted / validated there; NMX_BASES_VALIDATE is always applied (read_points does).
 *  - nmx_bases_register_ptau: HyperKZG `load_setup` (src/provider/hyperkzg.rs:658-674) -> `read_ptau`
 *    (src/provider/ptau.rs:270-436): "ptau" magic, version 1, 11 or 3 sections, header section (n8, prime == base
 *    modulus of `curve`, power with num_g1 <= 2^(power+1) - 1 and num_g2 <= 2^power), then the first num_g1 points
 *    of section 2 become the key.  Section 3 (G2) is left to the host: pairings are outside this library.
 *  - nmx_bases_register_keyfile: Pedersen `load_setup` (src/provider/pedersen.rs:318-340): 12-byte head
 *    "PEDERSEN_KEY", then h, then ck[0 .. n) (the caller passes n already rounded with next_power_of_two as the
 *    reference does); h comes back as canonical x || y in h_xy64.
 * flags: NMX_BASES_PRECOMPUTE. */
int nmx_bases_register_ptau(int curve, const char* path, size_t num_g1, size_t num_g2, uint32_t flags, uint64_t* handle);
int nmx_bases_register_keyfile(int curve, const char* path, size_t n, uint32_t flags, uint64_t* handle,
                               uint8_t* h_xy64);
int nmx_bases_unregister(uint64_t handle);
/* copies registered bases [offset, offset+n) back to the host as canonical x||y (test / key export helper) */
int nmx_bases_read(uint64_t handle, size_t offset, size_t n, void* out_xy64);

/* Synthetic key: P_i = (k0 + i) * G for i in [0, n), generated in HBM (the construction of
 * src/provider/curve_property_tests.rs:186-194; plays the role of the test-utils `setup`,
 * src/provider/hyperkzg.rs:357-376, whose real keys come from the host). */
int nmx_bases_generate(int curve, uint64_t k0, size_t n, uint32_t flags, uint64_t* handle);

/* ---- MSM ---------------------------------------------------------------------------------------------
 * DlogGroupExt::vartime_multiscalar_mul (src/provider/traits.rs:79; impls src/provider/bn256_grumpkin.rs:45-47,
 * src/provider/traits.rs:371-377) == msm() (src/provider/msm.rs:225-419):  out = sum_i scalars[i] * bases[i].
 * n == 0 -> identity (msm.rs:228).  Identity bases and zero scalars contribute nothing (msm.rs:247-249).
 *
 * Slice form = the trait's own signature: the reference passes `&ck.ck[..n]` and no handle (pedersen.rs:263-270,
 * hyperkzg.rs:584-591, blitzar.rs:7-20).  The library therefore keeps a SLICE CACHE of resident keys: a call whose
 * `bases_xy64` pointer is the first element of (or lies inside) an array it has seen before runs over the resident
 * copy exactly like nmx_msm_handle -- no upload, no conversion; a longer prefix of the same array re-registers it at the
 * new length, so one resident copy serves `&ck[..n]` for every n.  First and second sight of an array: the key alone;
 * from its third use on it also has its window tables, built from the resident copy (arrays seen once or twice -- IPA's
 * per-round Vecs, ipa_pc.rs:212-230 -- never pay for them; nmx_set_option("cache_table_after")).  Tables that do not
 * fit the cache budget or the HBM left: the key stays resident without them (plain GPU path).
 * The trait is a pure function of the slice, and the cache keeps it one: identity of an array = (curve, layout flag, host
 * address), confirmed on EVERY call from the caller's bytes against the 64-bit hashes of ALL points recorded at upload.
 * Default (option "cache_verify" = 0): every point of the slice is re-hashed on every hit -- slices up to 2048 points before
 * anything is launched; longer ones get a quick look first (first point, last point, eight probes that move from call to call:
 * a freed-and-reused address fails here) and the full pass runs on host pool threads WHILE the GPU computes the MSM (2^20
 * points = 64 MiB, ~1 ms on 8 threads, under a ~2 ms call); the result is handed out only after the pass succeeded, otherwise
 * the entry is dropped, the call repeated on a fresh upload and the stale result discarded (NMX_STAT_CACHE_STALE).  So an
 * in-place edit of even ONE point changes the very next result.  Host cost: one read of the slice per call; the library never
 * runs more than 8 verification workers at a time across ALL callers (concurrent rayon callers share them, the surplus
 * verifies on its calling thread's share only) -- INTEGRATION.md section 2 has the numbers.
 * Opt-in "cache_verify" = 1 (callers whose keys are immutable, which is every caller in the reference tree): the quick look
 * plus a rolling window of max(4096, n/16) consecutive points that continues where the previous call stopped; an in-place edit
 * is then caught within 16 calls instead of at once, and nmx_cache_invalidate is the contract for in-place edits.
 * Arrays shorter than the cache's min_n (default 128 points) and calls with NMX_BASES_NOCACHE are uploaded for the call only.
 * LRU eviction under a byte budget (default: a quarter of the device's HBM), also when an upload runs out of HBM; evicted or
 * invalidated keys stay alive until the calls using them return.  Keys that must never be re-hashed: register them
 * (nmx_bases_register) and call the *_handle forms. */
int nmx_msm(int curve, const void* scalars, const void* bases_xy64, size_t n, uint32_t flags,
            uint8_t* out, uint8_t* out_is_inf);
/* Slice-cache control.  nmx_cache_configure: max_bytes / min_n / max_entries, 0 = leave unchanged. */
int nmx_cache_clear(void);
int nmx_cache_invalidate(const void* bases_xy64);
int nmx_cache_configure(size_t max_bytes, size_t min_n, size_t max_entries);
/* Smallest n for which the shim should send an MSM to the GPU at all (below it: stay on the CPU `msm()`).  The
 * reference issues thousands of tiny MSMs through this same trait -- IPA's CommitmentKeyExtTrait::fold is n/2
 * two-point MSMs per round (src/provider/pedersen.rs:484-497), msm() itself switches to msm_simple for n <= 16
 * (src/provider/msm.rs:233-235) -- and one GPU MSM costs 0.2-0.3 ms whatever its size.  Default 128 (measured
 * cross-over against the CPU oracle), env NMX_MIN_N overrides.  The library itself accepts any n. */
size_t nmx_min_gpu_n(int curve);
/* One-time self-check for the zero-copy layouts (NMX_BASES_MONT / NMX_SCALARS_MONT): the shim passes the raw
 * in-memory bytes of the curve's standard generator (`G1Affine::generator()`, 64 bytes) and of the scalar
 * `Scalar::from(value)` (32 bytes); returns NMX_OK iff they are x*2^256 mod p little-endian limbs as the *_MONT
 * flags assume (halo2curves does not promise repr(C)), NMX_E_FORMAT otherwise.  Pure host code. */
int nmx_check_layout(int curve, const void* generator_raw64, const void* scalar_raw32, uint64_t value);
/* Monotonic counters of this process (cap >= NMX_STAT_COUNT entries are written; returns NMX_STAT_COUNT). */
enum {
  NMX_STAT_CACHE_HITS = 0,    /* slice-form calls served by a resident key                                   */
  NMX_STAT_CACHE_UPLOADS = 1, /* keys uploaded into the slice cache (first sight, or content mismatch)        */
  NMX_STAT_CACHE_REGROWS = 2, /* of those: re-registrations because a longer prefix of a known array arrived  */
  NMX_STAT_CACHE_EVICTIONS = 3,
  NMX_STAT_CACHE_ENTRIES = 4, /* current                                                                      */
  NMX_STAT_CACHE_BYTES = 5,   /* current HBM bytes held by the slice cache (keys + tables)                    */
  NMX_STAT_UNCACHED_CALLS = 6,/* slice-form calls that uploaded their bases for the call only                 */
  NMX_STAT_BASE_BYTES_H2D = 7,/* bytes of base points copied host -> device by slice-form calls               */
  NMX_STAT_MSM_CALLS = 8,     /* MSMs run (every entry point; a batch counts each vector)                     */
  NMX_STAT_FUSED_RUNS = 9,    /* batch calls whose short vectors ran as one fused pipeline run                */
  NMX_STAT_SHARDED_CALLS = 10,/* MSMs that fanned out over the shards of a multi-device key                   */
  NMX_STAT_CACHE_STALE = 11,  /* slice-form calls whose rolling content check caught an in-place edit of a cached
                                 array: the entry was dropped and the call repeated on a fresh upload          */
  NMX_STAT_TABLE_FALLBACKS = 12, /* keys left without window tables because the tables did not fit (budget / HBM) */
  NMX_STAT_LAUNCH_GAP_NS = 13, /* gauge: cost of one dependent one-wave launch on this box, measured once: a box
                                  diagnostic (slow-launch boxes of a pool show here); 0 until the first MSM       */
  NMX_STAT_SCAN_TIMEOUTS = 14, /* suffix-Horner calls whose single-pass scan gave up a look-back wait and were repeated on
                                  the two-pass kernels (never observed; the guard that turns a hang into a slower call) */
  NMX_STAT_SC_TORN_INJECTED = 15, /* option "sc_torn_test" only: deliberately torn challenge lines the host wrote ahead of the whole one */
  NMX_STAT_SC_TORN_REJECTS = 16,  /* ... and waits in which a pre-launched sum-check pass saw such a line and polled past it
                                     (sequence words new, checksum wrong); counted while "sc_torn_test" is on            */
  NMX_STAT_COUNT = 17
};
int nmx_stats(uint64_t* out, int cap);
/* same, bases taken from a registered key */
int nmx_msm_handle(uint64_t handle, size_t offset, const void* scalars, size_t n, uint32_t flags,
                   uint8_t* out, uint8_t* out_is_inf);

/* DlogGroupExt::vartime_multiscalar_mul_small_with_max_num_bits (src/provider/traits.rs:99-106) ==
 * msm_small_with_max_num_bits (src/provider/msm.rs:478-503): scalars are integers < 2^max_num_bits
 * (u8..u64 widened by the caller, `Into<u64>`); max_num_bits == 0 -> identity (msm.rs:489).
 * vartime_multiscalar_mul_small (traits.rs:93-96 / msm.rs:469-475) = this with max_num_bits = bit length of
 * the largest scalar, which nmx_msm_u64 computes itself when max_num_bits == NMX_BITS_AUTO. */
#define NMX_BITS_AUTO 0xffffffffu
int nmx_msm_u64(int curve, const uint64_t* scalars, const void* bases_xy64, size_t n, uint32_t max_num_bits,
                uint32_t flags, uint8_t* out, uint8_t* out_is_inf);
int nmx_msm_u64_handle(uint64_t handle, size_t offset, const uint64_t* scalars, size_t n,
                       uint32_t max_num_bits, uint32_t flags, uint8_t* out, uint8_t* out_is_inf);

/* Sparse commitments over a registered key (src/provider/pedersen.rs:395-427, src/provider/hyperkzg.rs:751-790):
 *   scalars != NULL: commit_sparse's MSM   out = sum_j scalars[j] * ck[indices[j]]
 *   scalars == NULL: commit_sparse_binary / batch_add (src/provider/msm.rs:689-708)   out = sum_j ck[indices[j]]
 * indices are `usize` on the reference side (host pointer); an index >= the key length is NMX_E_HANDLE. */
int nmx_msm_sparse_handle(uint64_t handle, const uint64_t* indices, const void* scalars, size_t k, uint32_t flags,
                          uint8_t* out, uint8_t* out_is_inf);

/* DlogGroupExt::batch_vartime_multiscalar_mul (src/provider/traits.rs:82-90; blitzar override
 * src/provider/blitzar.rs:22-40): k MSMs over one base array, the j-th using bases[..lens[j]]
 * (HyperKZG batch_commit, src/provider/hyperkzg.rs:593-612).  out = k x 64 bytes, out_is_inf = k bytes.
 * The shortest vectors of a batch (up to 32 on a 2^14 .. 2^19-point key, 16 at 2^20 .. 2^21, 256 below 2^14) run as ONE fused pass over the
 * key's window tables with a bucket set per vector; the others run as independent MSMs on concurrent streams.  Keys of >= 2^22
 * points (2^19 buckets per set: nothing beyond pairs fuses) carry a second, narrow table set over their first 2^18 points: the
 * vectors -- and single MSMs / commitments -- that stay inside it run there (option "prefix_tables" = 0: off). */
int nmx_msm_batch(int curve, const void* const* scalar_vecs, const size_t* lens, size_t k,
                  const void* bases_xy64, size_t n_bases, uint32_t flags, uint8_t* out, uint8_t* out_is_inf);
int nmx_msm_batch_handle(uint64_t handle, const void* const* scalar_vecs, const size_t* lens, size_t k,
                         uint32_t flags, uint8_t* out, uint8_t* out_is_inf);
/* DlogGroupExt::batch_vartime_multiscalar_mul_small (src/provider/traits.rs:109-117; CommitmentEngineTrait::batch_commit_small,
 * src/traits/commitment.rs:139-150): k vectors of u64 scalars over prefixes of one base array, the j-th using
 * bases[..lens[j]]; max_num_bits as nmx_msm_u64 (NMX_BITS_AUTO: per vector, from the data; 0: every result is the identity,
 * msm.rs:489).  Vector by vector over the resident key, several in flight. */
int nmx_msm_u64_batch(int curve, const uint64_t* const* scalar_vecs, const size_t* lens, size_t k, const void* bases_xy64,
                      size_t n_bases, uint32_t max_num_bits, uint32_t flags, uint8_t* out_xy64, uint8_t* out_is_inf);
int nmx_msm_u64_batch_handle(uint64_t handle, const uint64_t* const* scalar_vecs, const size_t* lens, size_t k,
                             uint32_t max_num_bits, uint32_t flags, uint8_t* out_xy64, uint8_t* out_is_inf);

/* CommitmentEngineTrait::commit (src/traits/commitment.rs:52-195; Pedersen src/provider/pedersen.rs:263-270,
 * HyperKZG src/provider/hyperkzg.rs:584-591):  out = msm(v, ck[..n]) + h * r.  `h_xy64` / `r` follow the same
 * flags as bases / scalars and are host pointers. */
int nmx_commit(uint64_t ck_handle, const void* v, size_t n, const void* h_xy64, const void* r,
               uint32_t flags, uint8_t* out, uint8_t* out_is_inf);
/* A commitment that runs beside the caller's next calls.  The two MSMs of a folding step do not depend on each other:
 * commit_T reads W2 and X, never comm_W (src/r1cs/mod.rs:590-622), and the RO that absorbed comm_W (src/nova/nifs.rs:53) is
 * squeezed only behind comm_T (:60-63).  So `W.commit(ck)` (src/frontend/r1cs.rs:47) can be BEGUN, `S.commit_T(..)` computed,
 * and comm_W collected before `U2.absorb_in_ro` -- on the reference side a `rayon::join`, here two calls.  Each MSM ends in a
 * latency-bound tail (fold passes, reduce tree) that leaves most of the chip idle; side by side one's tail hides under the
 * other's accumulation (prove_step replay, bench.py --overlap-commits: 1.41 -> 1.1-1.2 ms).
 *   nmx_commit_begin   same arguments as nmx_commit; h_xy64 and r are copied, `v` must stay valid and unchanged until the
 *                      ticket is finished.  The commitment is ordered behind the calling thread's NMX_ASYNC calls like a
 *                      synchronous one would be, and runs on its own context and stream.
 *   nmx_commit_finish  waits, writes out[64] (128 with NMX_OUT_PARTIAL) / out_is_inf and retires the ticket.  Whatever the
 *                      commitment failed with is reported HERE (code and nmx_last_error); an unknown or already finished ticket
 *                      is NMX_E_HANDLE.  Every ticket must be finished (nmx_shutdown waits for the ones that were not and drops
 *                      their results).  nmx_profile_last does not describe these calls. */
int nmx_commit_begin(uint64_t ck_handle, const void* v, size_t n, const void* h_xy64, const void* r, uint32_t flags,
                     uint64_t* ticket);
int nmx_commit_finish(uint64_t ticket, uint8_t* out, uint8_t* out_is_inf);

/* ---- shard-resident vectors (multi-device keys) ----------------------------
 * A field vector laid out like the key it will be committed against: element i lives in the HBM of the device that holds
 * point i of a key of `n_key` points (nmx_shard_plan(n_key, nmx_devices_in_use(), 0, n)), so W, E, T are BORN on the shard
 * that commits them -- the reference chunks coefficients and bases together (src/provider/msm.rs:564-574).  Elements are raw
 * 32-byte words (canonical or Montgomery: the caller's choice, stated per call with NMX_SCALARS_MONT as everywhere else).
 *   nmx_svec_alloc / free / write (host -> shards) / read (shards -> host)
 *   nmx_svec_parts: the pieces (device pointer, element count, HIP device ordinal) for hosts that fill them with their own
 *     kernels; returns the number of pieces
 *   nmx_svec_map: the element-wise NIFS kernels shard by shard, every GPU on its own piece at the same time --
 *     NMX_OP_AXPY  out = in0 + r*in1                       RelaxedR1CSWitness::fold   src/r1cs/mod.rs:1058-1067
 *     NMX_OP_AXPY2 out = in0 + r*in1 + r^2*in2             fold_relaxed               src/r1cs/mod.rs:1096-1101
 *     NMX_OP_CROSS_TERM  out = in0*in1 - u*in2 - in3       commit_T                   src/r1cs/mod.rs:614-620
 *     NMX_OP_CROSS_TERM2 out = in0*in1 - u*in2 - in3 - in4 commit_T_relaxed           src/r1cs/mod.rs:652-659
 *     NMX_OP_VEC_ADD out = in0 + in1                       Z = Z1 + Z2                src/r1cs/mod.rs:590-609
 *     (`challenge` = r or u, host pointer; operands and `out` must share one layout; out may be an operand)
 *   nmx_msm_svec / nmx_commit_svec: nmx_msm_handle(key, 0, ..) / nmx_commit over the first n elements, the scalars taken
 *     shard by shard from the vector (it must have been allocated with n_key = the key's length). */
enum { NMX_OP_AXPY = 0, NMX_OP_AXPY2 = 1, NMX_OP_CROSS_TERM = 2, NMX_OP_CROSS_TERM2 = 3, NMX_OP_VEC_ADD = 4 };
int nmx_svec_alloc(size_t n_key, size_t n, uint64_t* svec);
int nmx_svec_free(uint64_t svec);
int nmx_svec_write(uint64_t svec, const void* host_elems32);
int nmx_svec_read(uint64_t svec, void* host_elems32);
int nmx_svec_parts(uint64_t svec, void** dev_ptrs, size_t* counts, int* hip_devices, int cap);
int nmx_svec_map(int field, int op, const uint64_t* in, int n_in, const void* challenge, uint32_t flags, uint64_t out);
int nmx_msm_svec(uint64_t key_handle, uint64_t svec, size_t n, uint32_t flags, uint8_t* out, uint8_t* out_is_inf);
int nmx_commit_svec(uint64_t ck_handle, uint64_t svec, size_t n, const void* h_xy64, const void* r, uint32_t flags,
                    uint8_t* out, uint8_t* out_is_inf);

/* Sum of `count` 128-byte partials (NMX_OUT_PARTIAL results gathered from the ranks of a sharded MSM) into one
 * affine point: the G-term combine of SURVEY.md 8(e) (the reference's rayon `reduce(identity, +)`,
 * src/provider/msm.rs:566-571,667-673). */
int nmx_point_sum(int curve, const uint8_t* partials128, size_t count, uint8_t* out, uint8_t* out_is_inf);

/* ---- field-vector kernels either side of the MSM -------------------------------
 * field ids: 0 = BN254 Fq (Grumpkin scalars), 1 = BN254 Fr (BN254 scalars), 2 = Pasta Fp (Vesta scalars),
 * 3 = Pasta Fq (Pallas scalars).  Vectors are n x 32 bytes; flags: NMX_SCALARS_MONT (vectors and the challenge
 * are raw Montgomery limbs), NMX_SCALARS_DEVICE (every vector pointer incl. `out` is an HBM pointer -- results
 * then stay resident for the next nmx_msm_handle(..., NMX_SCALARS_DEVICE)); the challenge is always a host pointer. */
enum { NMX_F_BN254_FQ = 0, NMX_F_BN254_FR = 1, NMX_F_PASTA_FP = 2, NMX_F_PASTA_FQ = 3 };
/* out = a + r*b : RelaxedR1CSWitness::fold, W = W1 + r*W2 and E = E1 + r*T (src/r1cs/mod.rs:1058-1067) */
int nmx_field_axpy(int field, const void* a, const void* b, const void* r, size_t n, uint32_t flags, void* out);
/* out = a + r*b + r^2*c : fold_relaxed's E (src/r1cs/mod.rs:1096-1101) */
int nmx_field_axpy2(int field, const void* a, const void* b, const void* c, const void* r, size_t n, uint32_t flags,
                    void* out);
/* out = az*bz - u*cz - e : the cross term T of commit_T (src/r1cs/mod.rs:614-620) */
int nmx_field_cross_term(int field, const void* az, const void* bz, const void* cz, const void* e, const void* u,
                         size_t n, uint32_t flags, void* out);
/* out = az*bz - u*cz - e1 - e2 : the cross term of commit_T_relaxed, two relaxed instances (src/r1cs/mod.rs:652-659;
 * NIFSRelaxed::prove, src/nova/mod.rs:817,833), u = U1.u + U2.u formed by the caller */
int nmx_field_cross_term2(int field, const void* az, const void* bz, const void* cz, const void* e1, const void* e2,
                          const void* u, size_t n, uint32_t flags, void* out);
/* out = a + b : Z = Z1 + Z2 (src/r1cs/mod.rs:590-609) */
int nmx_field_vec_add(int field, const void* a, const void* b, size_t n, uint32_t flags, void* out);
/* batch_invert (src/spartan/mod.rs:54-118; callers: ppsnark's lookup argument src/spartan/ppsnark.rs:430, IPA src/provider/
 * ipa_pc.rs:328): out[i] = 1 / v[i].  NMX_E_ZERO when an element is zero (the reference returns Err(NovaError::InternalError)) --
 * nothing meaningful is written then.  Montgomery's trick level by level on the device (strided 8..32-element chunks per lane), the
 * top <= 128 products inverted on the host.  `out` must not overlap `v`. */
int nmx_field_batch_invert(int field, const void* v, size_t n, uint32_t flags, void* out);
/* out[n_out] = parts[0] || parts[1] || ... || 0 ... 0 for vectors that live in HBM: Spartan's `z = [W.W, vec![U.u], U.X].concat()`
 * then `z.resize(2 * num_vars, 0)` (src/spartan/snark.rs:133, 193-196), and -- with one part -- the clones batch_eval_reduce
 * takes of W and E before it binds them (src/spartan/mod.rs:407-410).  Bit i of device_mask: parts[i] is an HBM pointer (else a
 * host pointer: u and X are host scalars); `out` is HBM (NMX_SCALARS_DEVICE required).  Copies on the library's stream, ordered
 * with the calling thread's other calls; NMX_ASYNC applies when every part is in HBM.  k <= 64. */
int nmx_field_concat(int field, const void* const* parts, const size_t* lens, uint64_t device_mask, size_t k, size_t n_out,
                     uint32_t flags, void* out);
/* MultilinearPolynomial::bind_poly_var_top (src/spartan/polys/multilinear.rs:65-84):
 * out[i] = z[i] + r*(z[i + len/2] - z[i]), i < len/2.  `out` may equal `z` (in place, as the reference does). */
int nmx_mle_bind_top(int field, const void* z, size_t len, const void* r, uint32_t flags, void* out);
/* HyperKZG fold step (src/provider/hyperkzg.rs:1085-1095): out[j] = p[2j] + x*(p[2j+1] - p[2j]), j < len/2 */
int nmx_poly_fold_pairs(int field, const void* p, size_t len, const void* x, uint32_t flags, void* out);
/* The whole fold loop of HyperKZG's prove (src/provider/hyperkzg.rs:1085-1095) as one call (round 6): outs[0] = fold(p, xs[0]) of len / 2
 * elements, outs[i] = fold(outs[i - 1], xs[i]) of len >> (i + 1), i < k <= log2(len); the reference runs it with k = ell - 1 and
 * xs[i] = point[ell - i - 1].  len a power of two; xs: k x 32 bytes on the host; outs: k distinct vectors, none of them p.  With
 * NMX_SCALARS_DEVICE the folds of more than 2048 inputs are one launch each and all the shorter ones run in ONE block (each reading
 * its predecessor's output from LDS); NMX_ASYNC as for nmx_poly_fold_pairs.  Same results, fold by fold, as k single calls. */
int nmx_poly_fold_chain(int field, const void* p, size_t len, const void* xs, size_t k, uint32_t flags, void* const* outs);

/* The N-scaling sums of one eq-factored sum-check round (src/spartan/sumcheck.rs:900-1075), over id in [0, len/2)
 * with x0 = X[id], x1 = X[id + len/2] and factor = eqL[id >> shift] * eqR[id & (2^shift - 1)] (first-half rounds,
 * sumcheck.rs:1233-1247) or factor = eqR[id] when eqL == NULL (last half, sumcheck.rs:1249-1251):
 *   mode 3 (A, B, C): out = (sum (a0*b0 - c0)*factor, sum (a1-a0)*(b1-b0)*factor)    cubic_with_three_inputs
 *   mode 2 (A, B):    out = (sum (a0*b0 - 1)*factor,  sum (a1-a0)*(b1-b0)*factor)    cubic_with_two_inputs
 *   mode 1 (A):       out = (sum a0*factor, 0)                                        quadratic_with_one_input
 * out64 = two field elements in the vectors' own form (host pointer).  The O(1) derivation of the round polynomial
 * from these sums and the claim stays on the caller's side (sumcheck.rs:686-753). */
int nmx_sumcheck_eq_sums(int field, int mode, const void* A, const void* B, const void* C, size_t len, const void* eqL,
                         size_t n_eqL, const void* eqR, size_t n_eqR, uint32_t shift, uint32_t flags, uint8_t* out64);
/* One whole prover round in a single pass over HBM-resident tables (NMX_SCALARS_DEVICE required): bind A, B, C with
 * the round challenge r (bind_poly_var_top, src/spartan/polys/multilinear.rs:65-84, called at
 * src/spartan/sumcheck.rs:535-545) AND return the NEXT round's sums (t_0, t_inf) over the bound tables, exactly what
 * nmx_mle_bind_top x3 followed by nmx_sumcheck_eq_sums(mode, outA, outB, outC, len/2, ...) would give; the eq tables
 * are the next round's.  outX may equal X (in place, as the reference binds).  len % 4 == 0. */
int nmx_sumcheck_bind_eq_sums(int field, int mode, const void* A, const void* B, const void* C, size_t len, const void* r,
                              const void* eqL, size_t n_eqL, const void* eqR, size_t n_eqR, uint32_t shift,
                              uint32_t flags, void* outA, void* outB, void* outC, uint8_t* out64);
/* The sums of one round of the sum-checks WITHOUT an eq factor, over id in [0, len/2), dX = x1 - x0, X(-1) = 2*x0 - x1:
 *   kind 1 (A, B)    quad_prod  out = (sum a0*b0,    sum dA*dB)                  src/spartan/sumcheck.rs:163-186
 *   kind 2 (A, B)    linear     out = (sum a0 - b0,  sum A(-1) - B(-1))          src/spartan/sumcheck.rs:353-378
 *   kind 3 (A, B)    quadratic  out = (sum a0*b0,    sum A(-1)*B(-1))            src/spartan/sumcheck.rs:380-405
 *   kind 4 (A, B, C) cubic      out = (sum a0*b0*c0, sum dA*dB*dC, sum A(-1)*B(-1)*C(-1))   sumcheck.rs:407-443
 * out96 = three field elements (the third is zero for kinds 1-3), host pointer, in the vectors' own form. */
int nmx_sumcheck_plain_sums(int field, int kind, const void* A, const void* B, const void* C, size_t len, uint32_t flags,
                            uint8_t* out96);
/* ---- Spartan's sum-check provers, ONE call each (BASELINE.json configs[4]; src/spartan/snark.rs:113-260) -------------------
 * A sum-check round is a streaming pass whose size halves every round, followed by a challenge only the host's transcript can
 * produce; at 2^20 the passes are ~0.1 ms and the rest is per-round latency.  So the round LOOP lives behind the boundary: the
 * tables (HBM-resident with NMX_SCALARS_DEVICE -- the intended form; host arrays otherwise: uploaded for the call, the host
 * copies left untouched) are bound in place, the bind of a round is fused with the next round's
 * sums, results reach the host through a polled mailbox in pinned memory, rounds that fit one block are one launch, and the
 * O(1) algebra of a round (derive_from_claim_deg2/1, UniPoly::from_evals_deg3/2, evaluate, EqSumCheckInstance::bound --
 * src/spartan/sumcheck.rs:680-753, 1226-1231, polys/univariate.rs:90-149) runs in the library.  The one thing that stays on the
 * caller's side is the transcript step `transcript.absorb(b"p", &poly); transcript.squeeze(b"c")` (sumcheck.rs:224-227,
 * 481-484, 315-318): the callback receives the round polynomial as UniPoly coefficients (constant term first; the shim builds
 * `UniPoly { coeffs }`, whose to_transcript_bytes drops the linear term itself) and returns the challenge; elements in the
 * vectors' own form (NMX_SCALARS_MONT as everywhere).  A non-zero return aborts the proof (NMX_E_ARG); a challenge >= p is
 * NMX_E_SCALAR_RANGE.  Outputs (any may be NULL): out_polys = rounds x n_coeffs x 32 bytes (what SumcheckProof compresses),
 * out_r = rounds x 32 (the challenges), final evaluations as listed.  The tables' contents after a call are unspecified (partly
 * bound: the reference's are consumed too).  Once the tables hold <= 2^"sc_host_tail" elements (option, default 7 = 128
 * elements, 0..8) the remaining rounds run on the HOST -- a few hundred field products take the host 1-8 us, any kernel round
 * trip 20-25 us; the last device bind lands the tables in pinned memory.  Option "sc_fused_sum" (default 1): a round is one
 * launch, the block that finishes last adds the per-block partials up; 0: pass + one-block sum.  Option "sc_poll_us": how long a round's mailbox is
 * polled before the stream is synchronised instead (default 2000; 0: always synchronise).  Option "sc_side_streams" (default 1): the
 * claims of a prove_batch_eval round are independent passes and run on one stream each; 0: all on the call's stream.  Options
 * "sc_host_parts" (default 1: a pass of <= 64 blocks hands every block's partial sums to the host, which adds them; 0: the last block
 * does) and "sc_quad" (default 1: passes of <= 2^12 indices of the cubic / quad_prod provers spread an index over four lanes) shorten
 * the dependent chain of the small rounds; "sc_prelaunch" (default 1; needs a large-BAR device and sc_poll_us != 0) enqueues the
 * provers' small passes a round early -- the pass waits on the device for its challenge, which the host
 * writes through the BAR -- taking the launch latency off those rounds; results are identical either way.  The 64-byte challenge
 * line carries a sequence word per 16-byte piece and a 64-bit checksum of its payload: the waiting pass accepts it only whole
 * (no store granularity of the write-combining path is assumed; option "sc_torn_test" = microseconds a deliberately torn line is
 * left on the device first, tests only, with NMX_STAT_SC_TORN_INJECTED / _REJECTS).  At most 256 blocks per device wait this way
 * at any time; passes beyond that budget are launched late.  THE TRANSCRIPT CALLBACK MUST NOT SYNCHRONISE THE DEVICE (hipFree,
 * hipDeviceSynchronize, a blocking copy on the NULL stream, a torch operator): a waiting pass is waiting for the challenge the
 * callback is about to return, and a device-wide wait behind it ends only with the pass's 2 s time-out (the call then fails with
 * NMX_E_HIP; nothing wrong is returned).
 *  - nmx_sumcheck_prove_cubic_with_three_inputs == SumcheckProof::prove_cubic_with_three_inputs (sumcheck.rs:446-507) with its
 *    EqSumCheckInstance (sumcheck.rs:593-1253; all sqrt-size eq tables built by one launch): A, B, C of 2^num_rounds elements,
 *    taus = num_rounds elements (host), 4 coefficients per round, out_claims = [A(r), B(r), C(r)].  A tau of zero (or a
 *    challenge that zeroes eq's running product) takes the reference's third-sum fallback (sumcheck.rs:1085-1136).
 *  - nmx_sumcheck_prove_quad_prod == prove_quad_prod (sumcheck.rs:199-249): A, B of 2^num_rounds elements, 3 coefficients per
 *    round, out_claims = [A(r), B(r)].
 *  - nmx_sumcheck_prove_batch_eval == prove_batch_eval (sumcheck.rs:251-353; batch_eval_reduce, src/spartan/mod.rs:377-437):
 *    k <= 16 claims e_i = P_i(x_i), polys[i] of 2^num_rounds[i] elements (BOUND IN PLACE: the reference binds clones,
 *    spartan/mod.rs:407-410 -- pass copies to keep the originals), eq_points[i] = num_rounds[i] elements (host), claims and
 *    coeffs = k elements each (host), 3 coefficients per round over max(num_rounds) rounds, out_finals = [P_i(r_i)]. */
typedef int (*nmx_transcript_fn)(void* ctx, const uint8_t* coeffs32, size_t n_coeffs, uint8_t* challenge32_out);
int nmx_sumcheck_prove_cubic_with_three_inputs(int field, const void* claim, const void* taus, size_t num_rounds, void* A, void* B,
                                               void* C, uint32_t flags, nmx_transcript_fn transcript, void* ctx, uint8_t* out_polys,
                                               uint8_t* out_r, uint8_t* out_claims);
int nmx_sumcheck_prove_quad_prod(int field, const void* claim, size_t num_rounds, void* A, void* B, uint32_t flags,
                                 nmx_transcript_fn transcript, void* ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims);
int nmx_sumcheck_prove_batch_eval(int field, const void* claims, const size_t* num_rounds, void* const* polys,
                                  const void* const* eq_points, const void* coeffs, size_t k, uint32_t flags,
                                  nmx_transcript_fn transcript, void* ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_finals);
/* ---- inner-product argument (the evaluation engine of the secondary curve) -----------------------------------------------------
 * InnerProductArgument::prove (src/provider/ipa_pc.rs:174-281), reached through EvaluationEngine::prove (:69-82) -- the evaluation
 * argument of S2 in CompressedSNARK::prove (src/nova/mod.rs:862-881; Grumpkin / Pallas / Vesta engines, src/provider/mod.rs:38-148).
 * One call runs every round: c_L, c_R (inner_product, :84-90, :207-208), L = commit(ck_R.combine(ck_c), a_L || c_L, 0) and
 * R = commit(ck_L.combine(ck_c), a_R || c_R, 0) (:213-232), the folds of a and b (:237-247).  The reference also folds the key each
 * round -- ck.fold(r^-1, r), n/2 two-point MSMs (src/provider/pedersen.rs:484-497) -- and commits against the folded halves; this
 * library never folds it: round k's L and R are MSMs over the REGISTERED key with scalars a_k[i] * S_k[m] (S_k = the 2^k products of
 * the earlier r_t^(+-1)), one fused two-vector run over the key's window tables.  Same group elements, so the same L, R, challenges
 * and a_hat as the reference's loop (checked against the oracle's key-folding restatement and the reference's verifier,
 * ipa_pc.rs:286-390, in tests/).
 *   ck_handle   the Pedersen key (CommitmentKey::ck), registered; the first n points are used (`ck.split_at(U.b_vec.len())`, :183);
 *               n > its length: NMX_E_HANDLE
 *   ck_c_xy64   host, 64 bytes: the ALREADY SCALED one-point key `ck_c.scale(&r)` (:190-191; the caller's transcript squeezed that r
 *               after absorbing the instance), in the form NMX_BASES_MONT names
 *   a, b        W.a_vec and U.b_vec: n elements each, n a power of two (2^ell evaluations); host, or HBM with NMX_SCALARS_DEVICE;
 *               canonical, or Montgomery limbs with NMX_SCALARS_MONT (then r and a_hat are Montgomery too).  Not modified.
 *   transcript  called once per round with L and R (affine canonical x || y like every result of this library, and whether each is
 *               the identity): absorb(b"L", &L); absorb(b"R", &R); squeeze(b"r") (:231-234); writes r and returns 0.  Non-zero:
 *               NMX_E_ARG; r >= modulus: NMX_E_SCALAR_RANGE; r = 0: NMX_E_ZERO (`r.invert().unwrap()` panics, :235).
 *   out_L, out_R  log2(n) points of 64 bytes each (L_vec, R_vec; canonical x || y); out_is_inf (may be null): 2 log2(n) bytes, [2k] = L_k is the identity,
 *               [2k + 1] = R_k;  out_a_hat: 32 bytes (a_vec[0] after the last fold, :271).  n = 1: no round, a_hat = a[0].
 * Flags: NMX_SCALARS_MONT, NMX_SCALARS_DEVICE, NMX_BASES_MONT; anything else NMX_E_ARG.  On an error the outputs written so far are
 * unspecified and nothing of the call still runs. */
typedef int (*nmx_ipa_transcript_fn)(void* ctx, const uint8_t* L_xy64, int L_is_inf, const uint8_t* R_xy64, int R_is_inf,
                                     uint8_t* r32_out);
int nmx_ipa_prove(uint64_t ck_handle, const void* ck_c_xy64, const void* a, const void* b, size_t n, uint32_t flags,
                  nmx_ipa_transcript_fn transcript, void* ctx, uint8_t* out_L, uint8_t* out_R, uint8_t* out_is_inf, uint8_t* out_a_hat);
/* PolyEvalWitness::batch / batch_diff_size (src/spartan/mod.rs:165-277): out[i] = sum_j s^j * vecs[j][i], i < n_out,
 * vectors shorter than n_out read as zero-padded; every lens[j] <= n_out.  `vecs`, `lens`, `s` are host arrays; the
 * vectors themselves and `out` follow NMX_SCALARS_DEVICE. */
int nmx_field_lincomb_powers(int field, const void* const* vecs, const size_t* lens, size_t k, const void* s, size_t n_out,
                             uint32_t flags, void* out);

/* out[i] = sum_{k >= i} f[k] * u^(k-i), i < n (coefficient form).  out[0] is `poly_eval(f, u)` (Horner,
 * src/provider/hyperkzg.rs:1011-1020); out[1..n) is the quotient h of `div_by_monomial(f, u)`
 * (src/provider/hyperkzg.rs:961-999, h[i-1] = f[i] + h[i]*u) that kzg_open commits to.  `f` and `out` must NOT overlap
 * (NMX_E_ARG for device-resident vectors that do: every out[i] depends on all later coefficients while other waves still
 * read them).  Coefficients may be any 256-bit words; the outputs are canonical (< p). */
int nmx_poly_suffix_horner(int field, const void* f, size_t n, const void* u, uint32_t flags, void* out);
/* HyperKZG's evaluation matrix (src/provider/hyperkzg.rs:1011-1020 `poly_eval`, called for every folded polynomial at
 * the three points r, -r, r^2, :1049-1056): out[i * m + j] = polys[i](points[j]) for k polynomials of any lengths
 * (coefficients low to high; an empty polynomial evaluates to 0) at m <= 4 points, all in one launch.  `polys`, `lens`,
 * `points` and `out` are host arrays; the coefficient vectors follow NMX_SCALARS_DEVICE. */
int nmx_poly_eval_multi(int field, const void* const* polys, const size_t* lens, size_t k, const void* points, size_t m,
                        uint32_t flags, uint8_t* out);
/* EqPolynomial::evals_from_points (src/spartan/polys/eq.rs:54-73): out[2^ell] = eq(r, x) for x in {0,1}^ell, r[0] the
 * most significant variable.  r: ell x 32 bytes, host.  out: host, or HBM with NMX_SCALARS_DEVICE. */
int nmx_eq_evals_from_points(int field, const void* r, size_t ell, uint32_t flags, void* out);
/* MultilinearPolynomial::evaluate / evaluate_with (src/spartan/polys/multilinear.rs:88-129): Z(r), len == 2^ell. */
int nmx_mle_evaluate(int field, const void* z, size_t len, const void* r, size_t ell, uint32_t flags, uint8_t* out32);
/* MultilinearPolynomial::multi_evaluate_with (src/spartan/polys/multilinear.rs:131-180): k polynomials of the same
 * length 2^ell at one point; the eq tables are built once.  out = k x 32 bytes (host). */
int nmx_mle_multi_evaluate(int field, const void* const* zs, size_t k, size_t len, const void* r, size_t ell,
                           uint32_t flags, uint8_t* out);
/* SparseMatrix (CSR, scipy naming: data / indices / indptr, src/r1cs/sparse.rs:232-260) resident in HBM, and
 * SparseMatrix::multiply_vec (sparse.rs:201-229): out[rows] = M * z.  indptr / indices are `usize` on the reference
 * side, hence uint64_t here.  R1CS matrices are fixed per circuit: register once, apply every step. */
int nmx_spmv_register(int field, const uint64_t* indptr, const uint64_t* indices, const void* data, size_t rows,
                      size_t cols, uint32_t flags, uint64_t* handle);
int nmx_spmv_unregister(uint64_t handle);
int nmx_spmv_apply(uint64_t handle, const void* z, size_t z_len, uint32_t flags, void* out);
/* NIFS::prove's provider work between two commitments as ONE call each (HBM-resident vectors only; NMX_ASYNC applies):
 *  - nmx_r1cs_cross_term: the body of commit_T (src/r1cs/mod.rs:590-620): Z = z1 + z2 (z2 NULL: Z = z1), then for every row
 *    T = (A Z)(B Z) - u (C Z) - E in one pass -- AZ, BZ, CZ never reach HBM.  A, B, C: matrices of one shape and field
 *    (nmx_spmv_register); z1, z2: z_len = cols elements; e, out: rows elements.  Bit-identical to nmx_field_vec_add +
 *    3 x nmx_spmv_apply + nmx_field_cross_term.
 *  - nmx_nifs_fold: R1CSWitness::fold (src/r1cs/mod.rs:1044-1067): w = w1 + r w2 (n_w elements) and e = e1 + r t (n_e
 *    elements) in one launch.  Outputs must not alias inputs other than element-wise in place (w == w1 is fine). */
int nmx_r1cs_cross_term(uint64_t A, uint64_t B, uint64_t C, const void* z1, const void* z2, size_t z_len, const void* e,
                        const void* u, uint32_t flags, void* out);
int nmx_nifs_fold(int field, const void* w1, const void* w2, size_t n_w, const void* e1, const void* t, size_t n_e, const void* r,
                  uint32_t flags, void* w, void* e);
/* compute_eval_table_sparse's product (src/spartan/mod.rs:497-533: `M_evals[col] += rx[row] * val` over every entry), i.e.
 * out[cols] = M^T * x with x_len == rows -- Spartan's inner sum-check needs it for A, B and C with x = eq(r_x, .)
 * (src/spartan/snark.rs:181-190).  The transposed form (CSC cut into lanes; a column as long as the constant-one column of an
 * R1CS matrix is split so that no lane walks it alone) is built from the resident matrix on the first call and kept with it. */
int nmx_spmv_apply_transposed(uint64_t handle, const void* x, size_t x_len, uint32_t flags, void* out);
/* (M*z1, M*z2) in one pass over the matrix: PrecomputedSparseMatrix::multiply_vec_pair (src/r1cs/sparse.rs:215-229) */
int nmx_spmv_apply_pair(uint64_t handle, const void* z1, const void* z2, size_t z_len, uint32_t flags, void* out1,
                        void* out2);
/* k matrices of one shape family against ONE vector in one call (round 6):
 *   transposed = 0: outs[i] = M_i * x          -- R1CSShape::multiply_vec (src/r1cs/mod.rs:407-471: A z, B z, C z under rayon::join)
 *   transposed = 1: outs[i] = M_i^T * x        -- compute_eval_table_sparse (src/spartan/mod.rs:497-533: the three tables, rayon::join)
 * With NMX_SCALARS_DEVICE the products run side by side (matrix 0 on the call's stream, the others on side streams that start behind
 * it and that it continues behind): they are latency-bound gathers, three in flight take little longer than one; NMX_ASYNC as for the
 * single-matrix calls.  Host operands: one matrix after the other.  1 <= k <= 8, all matrices over the same field. */
int nmx_spmv_apply_many(const uint64_t* handles, size_t k, int transposed, const void* x, size_t x_len, uint32_t flags,
                        void* const* outs);

/* ---- measurement ------------------------------------------------------------------------------------
 * With profiling on, every MSM brackets its stages with hipEvents on the stream the kernels run on;
 * nmx_profile_last returns the last call's stage times in milliseconds (same thread).
 * stage order: digits, sort, bounds+plan, accum, fold, reduce, tail(D2H+host Horner) ; returns #stages. */
int nmx_set_profiling(int on);
int nmx_profile_last(float* ms, int cap);
/* The calling thread's last MSM over a multi-device key, shard by shard: returns the number of shards; for shard j < cap
 * ms[j * NMX_PROF_STAGES + s] = its stage times (profiling on; same stage order), dev[j] = its logical device, branch[j] =
 * where its scalars came from (NMX_BRANCH_*); *combine_ms = the combine step alone (all-gather + point sum), *rccl_ranks =
 * ranks of the RCCL communicator it used (0: host sum).  After such a call nmx_profile_last gives the per-stage maximum
 * over the shards. */
#define NMX_PROF_STAGES 12
enum { NMX_BRANCH_HOST = 0,          /* host scalars: the shard's GPU pulled its slice over its own PCIe link            */
       NMX_BRANCH_LOCAL = 1,         /* one HBM array on device 0, this shard is on device 0: used in place              */
       NMX_BRANCH_PEER_COPY = 2,     /* one HBM array on device 0, this shard elsewhere: hipMemcpyPeerAsync over xGMI    */
       NMX_BRANCH_SHARD_RESIDENT = 3,/* NMX_SCALARS_SHARDED: the piece was already in this GPU's HBM                     */
       NMX_BRANCH_NONE = 4 };        /* no scalar array (all-ones sparse form)                                           */
int nmx_profile_last_sharded(float* ms, int* dev, int* branch, int cap, float* combine_ms, int* rccl_ranks);
/* forces the window width (0 = heuristic) -- tuning / tests only */
int nmx_set_window_bits(uint32_t c);
/* Pipeline selection knobs for A/B measurements and tests (the same ones the NMX_TUNE_* environment variables set at
 * start-up); every combination computes the same result.  Names: "no_partition" (1: generic radix-sort path instead of
 * the hand-written LDS partition), "seg_min_total" (segment-balanced accumulate from this many sorted entries on;
 * 0xffffffff: never), "seg_min_len", "seg_lanes" (0: a multiple of the kernel's resident lanes), "no_quad_accum",
 * "no_quad_final", "accum_prefetch" (0: by table size; 1 or 2), "no_batch_fuse" (1: every vector of a batch call runs
 * as its own MSM), "horner_top" (suffix Horner from 1024 coefficients on: 0 = the single-pass scan with decoupled look-back; the two-pass
 * kernels: 8 = 8-coefficient chunks in registers, 4, 1 = chunk-per-lane recursion only), "horner_window" (groups per look-back
 * round of the single-pass scan, 64; 1..63 force its multi-round path in tests), "horner_sub" (512-coefficient sub-tiles per
 * wave of the scan: 0 = by size, 1, 2, 4), "eq_max_blocks" (grid cap of the eq-factored sum passes: 0 = 768 for evaluate_with, 2048 otherwise), "horner_spin_limit" (polls before a wave of the scan gives up and the call falls back
 * to the two-pass kernels: 0 = 2^22; tests set 1),
 * "host_split" / "host_split_min_n" (an MSM with HOST scalars over at least host_split_min_n = 2^19 pairs of a key on one device is cut
 * into contiguous pieces (host_split = 255, the default: 2 / 3 / 4 pieces from 2^19 / 2^20 / 2^21 pairs; 2..16: that many) -- the
 * reference's own chunk + reduce decomposition, src/provider/msm.rs:564-574 -- so that piece i's scalars cross PCIe while piece
 * i - 1 computes; 0 / 1: one upload, then one MSM),
 * "small_blocks" (MSMs with at most 1024 buckets -- keys below 2^14 points: bucket sums in two block-level launches, this many
 * entries per four-lane group, default 8; 0 = the task path: plan, expand, accumulate, strided folds),
 * "seg_heavy_above" (pieces per bucket summed without a pre-fold pass: 0 = by table width, else 1..63), "no_tree_fuse" (bucket
 * reduction: 0 = fused levels or one launch per level by the box's measured launch gap, 1 = one launch per level, 2 = fused),
 * "hist_grid" (blocks of the partition's counting pass; 0 = as the placing pass: measured flat, profiles/r03_msm_2p20/tail_ab.txt),
 * "shard_min_n", "cache_table_after", "max_table_mib" (see the sections above), "force_peer_copy" (1: the HBM-resident scalars of
 * a sharded call are staged through hipMemcpyPeerAsync even when the shard sits on the source GPU: exercises the cross-device
 * branch on a one-GPU box), "combine" (0 and 1: host sum of the partials, the default; 2: RCCL all-gather required -- also with
 * one GPU, an error if it cannot be loaded), "cache_verify" (0: every slice-cache hit
 * re-hashes the caller's whole slice; 1: rolling window, for callers that register immutable keys).
 * Unknown name: NMX_E_ARG. */
int nmx_set_option(const char* name, uint32_t value);

#ifdef __cplusplus
}
#endif
#endif /* NOVA_MI355X_H */
