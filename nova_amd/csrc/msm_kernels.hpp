// msm_kernels.hpp -- the per-thread bodies of the Pippenger bucket-MSM kernels.
//
// Every kernel is a small functor with `operator()(uint32_t tid)`; msm_pipeline.hpp launches them through
// one generic __global__ trampoline (device backend) -- or through a plain loop (tests/host_emul, g++ only,
// used to debug indexing on machines without a GPU; never linked into libnova_mi355x.so).
//
// Replaces: /root/reference/src/provider/msm.rs:225-419 (`msm`), :478-503 (`msm_small_with_max_num_bits`)
// and the third-party halo2curves::msm::msm_best they delegate to (msm.rs:411,500).
//
// Pipeline (N pairs, window width c, W = ceil((bits+1)/c) windows, M = 2^(c-1) buckets per window):
//   1. DigitsFn      scalar -> W signed c-bit digits; emits (key = w*M + |d|-1, val = idx | sign<<31);
//                    zero digits / zero scalars / identity bases (msm.rs:247-249) get the trash key W*M.
//   2. radix sort    (key, val) pairs by key              [rocPRIM on the device backend]
//   3. BoundsFn      bucket k -> [start, end) in the sorted array
//   4. PlanFn        buckets longer than lmax are split into lmax-sized extra tasks (bounded work per lane
//                    whatever the scalar distribution: all-equal scalars put N points in one bucket)
//   5. AccumFn       one lane per (bucket | extra task): gather affine points, XYZZ mixed adds (msm.rs:129-165)
//   6. FoldFn x<=6   strided folds of a heavy bucket's partial sums -> bucket
//   7. ReducePairFn  per-window sum_k k*B_k as a binary tree, log2(M) levels, two dependent additions per level
//                    (msm.rs:555-561,637-643 do this sum serially per thread)
//   8. host tail     Horner over the W window sums (msm.rs:651-661), one inversion, canonical bytes.
#pragma once
#include "curve.hpp"

namespace nmx {

// atomics: the device instruction on the GPU; a plain read-modify-write in the host passes (hipcc's host pass parses
// the kernels too, and the g++ emulation runs one thread at a time)
NMX_HD uint32_t nmx_atomic_add(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicAdd(p, v);
#else
  uint32_t o = *p;
  *p = o + v;
  return o;
#endif
}
NMX_HD void nmx_atomic_or(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  atomicOr(p, v);
#else
  *p |= v;
#endif
}
NMX_HD void nmx_atomic_max(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  atomicMax(p, v);
#else
  if (v > *p) *p = v;
#endif
}

// error bits accumulated on the device
enum : uint32_t { ERR_SCALAR_RANGE = 1u, ERR_SMALL_RANGE = 2u };

struct MsmShape {
  uint32_t n;         // pairs
  uint32_t c;         // window width (bits)
  uint32_t W;         // digit windows
  uint32_t WB;        // bucket sets: W, or 1 when window w reads the precomputed table 2^(cw) * P (all windows share
                      // one bucket set and no window combination is needed); a fused batch over tables has one set per
                      // vector (rounded up to a power of two)
  uint32_t M;         // buckets per set = 2^(c-1)
  uint32_t nbuckets;  // WB*M ; key nbuckets is the trash bucket
  uint32_t lmax;      // longest run one lane accumulates
  uint32_t total;     // n*W sorted entries
};

// ----------------------------------------------------------------------------------------------------
// 1. digits
// ----------------------------------------------------------------------------------------------------
// Signed-digit recoding: d_w in [-(2^(c-1) - 1), 2^(c-1)], sum_w d_w 2^(cw) = s.  With W*c >= bits + 1 the final
// carry is always zero.
// ----------------------------------------------------------------------------------------------------
// digit source: everything the digit stage knows about a call, shared by DigitsFn (generic sort path) and the
// partition kernels (msm_partition.hpp)
// ----------------------------------------------------------------------------------------------------
template <int SFID> struct DigitSrc {
  const uint32_t* scalars;  // n x 8 u32 (canonical or Montgomery), or n x 2 (u64 mode)
  const uint32_t* bases;    // n x 16 u32, only to test for the identity encoding; null when the key holds none
  uint32_t* err;
  MsmShape sh;
  uint32_t scalars_mont, u64_bits, pre_stride, pre_offset;
  const uint32_t* gather;
  uint32_t all_ones;
  // fused batch (a7: k vectors over prefixes of one key, run as ONE pipeline): pair i of the concatenation is pair
  // i - batch_off[j] of vector j, read from batch_vec[j]; its buckets are set j (kbase = j * M).  batch_k = 0: one vector.
  uint32_t batch_k = 0;
  const uint32_t* batch_off = nullptr;          // [batch_k + 1] prefix sums of the vector lengths
  const uint32_t* const* batch_vec = nullptr;   // [batch_k] scalar arrays (n_j x 8 u32)

  // canonical scalar words of pair i, the row of its base in the key and the first bucket of its bucket set; false: the
  // pair contributes nothing (out-of-range scalar -> error bit; identity base, msm.rs:247-249)
  NMX_HD bool load(uint32_t i, uint32_t (&s)[9], uint32_t& bi, uint32_t& kbase, bool report) const {
    bool ok = true;
    const uint32_t* sc = scalars;
    kbase = 0;
    if (batch_k) {
      uint32_t lo = 0, hi = batch_k;  // batch_off[lo] <= i < batch_off[hi]
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (batch_off[mid] <= i) lo = mid;
        else hi = mid;
      }
      i -= batch_off[lo];
      sc = batch_vec[lo];
      kbase = lo * sh.M;
    }
    bi = (gather ? gather[i] : i) + pre_offset;
    if (u64_bits) {
      s[0] = all_ones ? 1u : sc[2 * (size_t)i];
      s[1] = all_ones ? 0u : sc[2 * (size_t)i + 1];
#pragma unroll
      for (int j = 2; j < 9; j++) s[j] = 0;
      if (u64_bits < 64) {
        const uint64_t v = ((uint64_t)s[1] << 32) | s[0];
        if (v >> u64_bits) {
          if (report) nmx_atomic_or(err, ERR_SMALL_RANGE);
          ok = false;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) s[j] = sc[8 * (size_t)i + j];
      s[8] = 0;
      if (!Fp<SFID>::words_lt_p(s)) {  // from_repr would have rejected it on the reference side
        if (report) nmx_atomic_or(err, ERR_SCALAR_RANGE);
        ok = false;
      }
      if (scalars_mont) Fp<SFID>::from_words(s).mont256_to_canonical().to_words(s);
    }
    if (bases) {
      uint32_t o = 0;
      const uint32_t* b = bases + 16 * (size_t)bi;
#pragma unroll
      for (int j = 0; j < 16; j++) o |= b[j];
      if (o == 0) ok = false;
    }
    return ok;
  }
  // first bucket of window w's digits: with window tables every window shares the bucket set kbase (0, or vector j's set in a
  // fused batch); plain keys keep one bucket set per window
  NMX_HD uint32_t key_base(uint32_t w, uint32_t kbase) const { return pre_stride ? kbase : w * sh.M; }
  // signed digit of window w: |d| in [0, 2^(c-1)], neg = sign; carry threads through the windows low to high
  NMX_HD void digit(const uint32_t (&s)[9], uint32_t w, uint32_t& carry, uint32_t& d, uint32_t& neg) const {
    const uint32_t bit = w * sh.c, word = bit >> 5, off = bit & 31;
    const uint64_t two = (word < 8) ? (((uint64_t)s[word + 1] << 32) | s[word]) : 0;
    d = (uint32_t)((two >> off) & ((1u << sh.c) - 1u)) + carry;
    neg = 0;
    if (d > sh.M) {
      d = (1u << sh.c) - d;
      neg = 1;
      carry = 1;
    } else {
      carry = 0;
    }
  }
};


// Generic path (rocPRIM radix sort): only for shapes the hand-written partition does not cover (msm_pipeline.hpp
// partition_supported) and under the `no_partition` option -- since round 4 plain keys and c = 20 tables take the partition.
// Materialises (key, val) pairs for the sort.
template <int SFID> struct DigitsFn {
  DigitSrc<SFID> src;
  uint32_t* keys;  // W x n
  uint32_t* vals;  // W x n
  NMX_HD void operator()(uint32_t i) const {
    const MsmShape& sh = src.sh;
    uint32_t s[9], bi, kbase;
    const bool skip = !src.load(i, s, bi, kbase, true);
    // the words are walked with constant indices and the windows peeled off a 64-bit bit buffer (c <= 20): indexing s[] by a
    // run-time word number keeps the array in memory -- promoted to LDS, 9 KB per block (msm_partition.hpp, for_each_digit)
    const uint32_t c = sh.c, mask = (1u << c) - 1u;
    uint64_t acc = 0;
    uint32_t have = 0, w = 0, carry = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      acc |= (uint64_t)s[j] << have;
      have += 32;
      while (have >= c && w < sh.W) {
        uint32_t d = ((uint32_t)acc & mask) + carry, neg = 0;
        acc >>= c;
        have -= c;
        if (d > sh.M) {
          d = (1u << c) - d;
          neg = 1;
          carry = 1;
        } else {
          carry = 0;
        }
        const uint32_t key = (d == 0 || skip) ? sh.nbuckets : ((src.pre_stride ? kbase : w * sh.M) + d - 1);
        const size_t o = (size_t)w * sh.n + i;
        keys[o] = key;
        vals[o] = (w * src.pre_stride + bi) | (neg << 31);
        w++;
      }
    }
  }
};

// ----------------------------------------------------------------------------------------------------
// 3. bucket boundaries in the sorted key array
// ----------------------------------------------------------------------------------------------------
struct alignas(16) KeyQuad {
  uint32_t x, y, z, w;
};
struct BoundsFn {
  static constexpr uint32_t kPerLane = 4;  // one 16-byte load per lane
  const uint32_t* keys;  // sorted, 16-byte aligned
  uint32_t* start;       // nbuckets + 1, zero-initialised
  uint32_t* end;         // nbuckets + 1, zero-initialised
  uint32_t total;
  NMX_HD void operator()(uint32_t q) const {
    const uint32_t j0 = q * kPerLane;
    uint32_t k[kPerLane + 2];  // k[0] = the key before the group, k[5] = the key after it
    if (j0 + kPerLane <= total) {
      const KeyQuad v = *reinterpret_cast<const KeyQuad*>(keys + j0);
      k[1] = v.x;
      k[2] = v.y;
      k[3] = v.z;
      k[4] = v.w;
    } else {
#pragma unroll
      for (uint32_t u = 0; u < kPerLane; u++) k[1 + u] = j0 + u < total ? keys[j0 + u] : 0;
    }
    k[0] = j0 ? keys[j0 - 1] : 0;
    k[kPerLane + 1] = j0 + kPerLane < total ? keys[j0 + kPerLane] : 0;
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      const uint32_t j = j0 + u;
      if (j >= total) break;
      if (j == 0 || k[u] != k[u + 1]) start[k[u + 1]] = j;
      if (j + 1 == total || k[u + 2] != k[u + 1]) end[k[u + 1]] = j + 1;
    }
  }
};

// ----------------------------------------------------------------------------------------------------
// 4. plan: split over-long buckets
// ----------------------------------------------------------------------------------------------------
struct HeavyRec {
  uint32_t bucket, off, cnt, pad;
};
struct TaskRec {
  uint32_t start, len;
};
struct PlanFn {
  // Appends to the split-bucket list.  In table mode EVERY bucket is split (n*W/M points against lmax = 24), and
  // appends through same-address atomics are bound by the atomic unit (~8 ns each: 0.28 ms for the 2^19 buckets
  // of a 2^22-pair MSM even with one atomic per wave).  So: every lane of a wave is alive (lanes past n come
  // with valid = false), a lane plans kPerLane consecutive buckets, and the wave reserves its list slots and task
  // slots with ONE 64-bit atomic; the two rarely-needed statistics are only touched by buckets with > 64 tasks.
  static constexpr bool kFullWaves = true;
  static constexpr uint32_t kPerLane = 4;
  const uint32_t* start;
  const uint32_t* end;
  uint32_t* counters;  // [0] = extra tasks used, [1] = split buckets (one u64: [1]:[0]), [3] = most tasks in one big
                       // bucket, [4] = big buckets
  HeavyRec* heavy;
  HeavyRec* big;  // the buckets split into more than 64 tasks (at most total / (64 * lmax) of them)
  MsmShape sh;
  NMX_HD void operator()(uint32_t q) const { (*this)(q, true); }
  NMX_HD void operator()(uint32_t q, bool valid) const {
    uint32_t nt[kPerLane];
    uint64_t mine = 0;  // hi: split buckets, lo: tasks
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      const uint32_t k = q * kPerLane + u;
      const uint32_t s = (valid && k < sh.nbuckets) ? end[k] - start[k] : 0;
      nt[u] = s > sh.lmax ? (s + sh.lmax - 1) / sh.lmax : 0;
      if (nt[u]) mine += (1ull << 32) | nt[u];
    }
    uint64_t base;  // list / task slots of this lane's first split bucket
#if defined(__HIP_DEVICE_COMPILE__)
    if (__ballot(mine != 0) == 0) return;  // wave-uniform
    const uint32_t lane = __lane_id();
    unsigned long long inc = mine;  // inclusive scan over the wave (neither half can carry into the other)
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long t = __shfl_up(inc, d);
      if (lane >= (uint32_t)d) inc += t;
    }
    const unsigned long long total = __shfl(inc, 63);
    unsigned long long wave_base = 0;
    if (lane == 0) wave_base = atomicAdd((unsigned long long*)counters, total);
    wave_base = __shfl(wave_base, 0);
    base = wave_base + inc - mine;
#else
    base = ((uint64_t)counters[1] << 32) | counters[0];
    counters[0] += (uint32_t)mine;
    counters[1] += (uint32_t)(mine >> 32);
#endif
    uint32_t h = (uint32_t)(base >> 32), off = (uint32_t)base;
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      if (!nt[u]) continue;
      const HeavyRec r{q * kPerLane + u, off, nt[u], 0};
      heavy[h++] = r;
      off += nt[u];
      if (nt[u] > 64) {  // see FoldFn: passes with T >= 64
        nmx_atomic_max(&counters[3], nt[u]);
        big[nmx_atomic_add(&counters[4], 1)] = r;
      }
    }
  }
};
// Task records of the over-long buckets, `lanes` lanes per bucket.  (A single lane per bucket would serialise
// N/lmax writes when one bucket holds a whole window -- all-equal or 0/1 scalars, or a nearly empty top window.)
struct ExpandFn {
  const uint32_t* start;
  const uint32_t* end;
  const uint32_t* counters;
  const HeavyRec* heavy;
  TaskRec* extra;
  MsmShape sh;
  uint32_t lanes, groups;
  NMX_HD void operator()(uint32_t tid) const {
    const uint32_t j = tid % lanes, nh = counters[1];
    for (uint32_t h = tid / lanes; h < nh; h += groups) {
      const HeavyRec r = heavy[h];
      const uint32_t b0 = start[r.bucket], s = end[r.bucket] - b0;
      // equal shares, not lmax-sized chunks plus a short remainder: the lanes of a wave finish together
      for (uint32_t t = j; t < r.cnt; t += lanes) {
        const uint32_t lo = (uint32_t)(((uint64_t)t * s) / r.cnt), hi = (uint32_t)(((uint64_t)(t + 1) * s) / r.cnt);
        extra[r.off + t] = TaskRec{b0 + lo, hi - lo};
      }
    }
  }
};

// ----------------------------------------------------------------------------------------------------
// 5. bucket accumulation (the dominant kernel)
// ----------------------------------------------------------------------------------------------------
template <int FID> struct AccumFn {
  const AffineW* bases;      // internal form, canonical, packed
  const uint32_t* vals;      // sorted
  const uint32_t* start;
  const uint32_t* end;
  const uint32_t* counters;
  const TaskRec* extra;
  XYZZW* buckets;   // nbuckets
  XYZZW* partials;  // extra-task results
  MsmShape sh;

  NMX_HD XYZZ<FID> run(uint32_t b, uint32_t len) const {
    XYZZ<FID> acc = XYZZ<FID>::identity();
    if (len == 0) return acc;
    // software pipeline, two stages: the index of point j+2 and the gather of point j+1 (64 B from a random row of a
    // table that can be GiBs) are in flight while point j is being added (~6000 VALU cycles); the gather never
    // waits for its own index load
    uint32_t v = vals[b];
    uint32_t vn = len > 1 ? vals[b + 1] : v;
    AffineW cur = bases[v & 0x7fffffffu];
    for (uint32_t j = 0; j < len; j++) {
      uint32_t vnn = vn;
      AffineW nxt = cur;
      if (j + 1 < len) nxt = bases[vn & 0x7fffffffu];
      if (j + 2 < len) vnn = vals[b + j + 2];
      acc.add_affine(Affine<FID>::load(cur), (v >> 31) != 0);  // identity bases never get here (trash key)
      cur = nxt;
      v = vn;
      vn = vnn;
    }
    return acc;
  }
  NMX_HD void operator()(uint32_t t) const {
    if (t < sh.nbuckets) {
      uint32_t b = start[t], s = end[t] - b;
      if (s <= sh.lmax) run(b, s).store(buckets[t]);  // heavy buckets are written by the folds
    } else {
      uint32_t e = t - sh.nbuckets;
      if (e >= counters[0]) return;
      TaskRec r = extra[e];
      run(r.start, r.len).store(partials[e]);
    }
  }
};

// ----------------------------------------------------------------------------------------------------
// 6. strided fold of heavy buckets' partials.  Lane j of group g folds positions j, j+T, j+2T, ... < cnt
//    into position j.  Applied with T = 32768, 4096, 512, 64, 8, 1; the last pass writes the bucket.
// ----------------------------------------------------------------------------------------------------
template <int FID> struct FoldFn {
  const uint32_t* counters;
  const HeavyRec* heavy;  // the list this pass walks: all split buckets (T < 64) or only the big ones (T >= 64)
  XYZZW* partials;
  XYZZW* buckets;
  uint32_t T;       // lanes per heavy bucket in this pass
  uint32_t cap;     // positions valid on entry = min(cnt, cap); cap = 0xffffffff for the first pass
  uint32_t groups;  // grid = groups * T lanes; groups loop over the heavy list
  NMX_HD void operator()(uint32_t tid) const {
    if (T >= 64 && counters[3] <= T) return;  // no big bucket has more than T partials: nothing to fold here
    uint32_t j = tid % T;
    uint32_t nh = counters[T >= 64 ? 4 : 1];
    for (uint32_t h = tid / T; h < nh; h += groups) {
      HeavyRec r = heavy[h];
      uint32_t cnt = r.cnt < cap ? r.cnt : cap;
      if (j >= cnt) continue;
      if (T != 1 && j + T >= cnt) continue;  // nothing to add: position j already holds its sum
      XYZZ<FID> acc = XYZZ<FID>::load(partials[r.off + j]);
      for (uint32_t q = j + T; q < cnt; q += T) acc.template add<kLatTail>(XYZZ<FID>::load(partials[r.off + q]));
      if (T == 1)
        acc.store(buckets[r.bucket]);
      else
        acc.store(partials[r.off + j]);
    }
  }
};

// ----------------------------------------------------------------------------------------------------
// Key registration: tables T_w[i] = 2^(c*w) * P_i, w = 1..W-1 (T_0 = the key itself).  288 GB of HBM per GPU makes
// W = 16 copies of a commitment key cheap (1 GiB for 2^20 BN254 points); in exchange every window of an MSM drops
// into ONE set of 2^(c-1) buckets: the bucket reduction shrinks W-fold and the host-side window combination
// (msm.rs:651-661) disappears.  Commitment keys are long-lived in Nova (created once by `setup`,
// src/provider/pedersen.rs:249-259), so this is paid once per key.
// ----------------------------------------------------------------------------------------------------
template <int FID> struct PrecompFn {
  AffineW* tables;  // W x n
  uint32_t n, c, W;
  NMX_HD void operator()(uint32_t i) const {
    Affine<FID> a = Affine<FID>::load(tables[i]);
    XYZZ<FID> p = XYZZ<FID>::from_affine(a);
    for (uint32_t w = 1; w < W; w++) {
      for (uint32_t q = 0; q < c; q++) p.dbl_in_place();
      a = p.to_affine();
      a.store(tables[(size_t)w * n + i]);
      p = XYZZ<FID>::from_affine(a);  // back to zz = zzz = 1: keeps the next doublings cheap and bounded
    }
  }
};

// The same tables in two passes, for keys large enough to matter (round 2): PrecompFn pays one Fermat inversion (380
// multiplications) per table point, 77 % of its time.  Pass 1 keeps doubling in extended-Jacobian form and parks every
// window's point as raw limbs; pass 2 inverts the W - 1 denominators zz * zzz of a key point with ONE inversion (Montgomery's
// trick: running products out, back-substitution in) and writes the affine table points.  9 multiplications + 1/(W-1)
// of an inversion per table point instead of 384: 2^20 BN254 points 45 -> 13 ms.  Identical table bytes (canonical affine).
template <int FID> struct PrecompDblFn {
  const AffineW* key;  // the key (table 0)
  XYZZL* raw;          // (W - 1) x m
  uint32_t i0, m, c, W;
  NMX_HD void operator()(uint32_t j) const {
    XYZZ<FID> p = XYZZ<FID>::from_affine(Affine<FID>::load(key[i0 + j]));
    for (uint32_t w = 1; w < W; w++) {
      for (uint32_t q = 0; q < c; q++) p.dbl_in_place();
      p.store_raw(raw[(size_t)(w - 1) * m + j]);
    }
  }
};
template <int FID> struct PrecompNormFn {
  const XYZZL* raw;  // (W - 1) x m
  uint32_t* pref;    // (W - 1) x m x 8 words: running products of the denominators
  AffineW* tables;   // W x n
  uint32_t n, i0, m, W;
  NMX_HD void operator()(uint32_t j) const {
    using F = Fp<FID>;
    F acc = F::one();
    for (uint32_t w = 1; w < W; w++) {
      const XYZZ<FID> p = XYZZ<FID>::load_raw(raw[(size_t)(w - 1) * m + j]);
      acc.canon().to_words(pref + 8 * ((size_t)(w - 1) * m + j));
      if (!p.is_identity()) acc = acc * (p.zz * p.zzz);
    }
    F inv = acc.inv();
    for (uint32_t w = W - 1; w >= 1; w--) {
      const XYZZ<FID> p = XYZZ<FID>::load_raw(raw[(size_t)(w - 1) * m + j]);
      Affine<FID> a;
      if (p.is_identity()) {
        a.x = F::zero();
        a.y = F::zero();
      } else {
        const F iw = inv * F::from_words(pref + 8 * ((size_t)(w - 1) * m + j));  // 1 / (zz * zzz) of this window
        inv = inv * (p.zz * p.zzz);
        a.x = (p.x * (iw * p.zzz)).canon();
        a.y = (p.y * (iw * p.zz)).canon();
      }
      a.store(tables[(size_t)w * n + i0 + j]);
    }
  }
};

// ----------------------------------------------------------------------------------------------------
// 7. bucket reduction:  F = sum_{b=0..n-1} (b+1) * B_b  per window, as a binary tree that is shallow in
//    *dependent point additions* (the lower levels have fewer lanes than the chip has SIMDs, so their cost is
//    latency, not throughput: ~7 us per dependent XYZZ addition).
//    Invariant per window:  F = Sum(Y) + G(D),  G(X) = sum_j j * X_j (0-based).  Start: Y = D = B.
//    One level:   D'_j = 2 * (D_2j + D_2j+1)          (keeps G's weights: 2j*D_2j + (2j+1)*D_2j+1 = j*D'_j + D_2j+1)
//                 Y'_j = Y_2j + Y_2j+1 + D_2j+1
//    After log2(n) levels n = 1, G = 0 and F = Y_0.  The two outputs of a pair are computed by different waves
//    (role 0 / role 1), so a level is two dependent additions deep whatever n is.  The reference does this sum
//    serially per thread (msm.rs:555-561,637-643).
// ----------------------------------------------------------------------------------------------------
template <int FID> struct ReducePairFn {
  const XYZZW* D;
  const XYZZW* Y;  // == D on the first level (Y = D = B)
  XYZZW* D_out;
  XYZZW* Y_out;
  uint32_t n_in;          // elements per window on entry (a power of two >= 2)
  uint32_t pairs;         // W * n_in / 2
  uint32_t pairs_padded;  // pairs rounded up to a multiple of 64: roles never share a wave
  uint32_t first;
  NMX_HD void operator()(uint32_t tid) const {
    const uint32_t role = tid >= pairs_padded ? 1u : 0u;
    const uint32_t j = tid - role * pairs_padded;
    if (j >= pairs) return;
    const uint32_t half = n_in >> 1;
    const uint32_t w = j / half, u = j - w * half;
    const size_t base = (size_t)w * n_in + 2 * (size_t)u;
    const size_t o = (size_t)w * half + u;
    if (role == 0) {
      if (n_in == 2) return;  // last level: only Y_0 is needed
      XYZZ<FID> d = XYZZ<FID>::load(D[base]);
      d.template add<kLatTail>(XYZZ<FID>::load(D[base + 1]));
      d.template dbl_in_place<kLatTail>();
      d.store(D_out[o]);
    } else {
      XYZZ<FID> y;
      if (first) {  // Y = D = B:  B_2j + 2 * B_2j+1
        y = XYZZ<FID>::load(D[base + 1]);
        y.template dbl_in_place<kLatTail>();
        y.template add<kLatTail>(XYZZ<FID>::load(D[base]));
      } else {
        y = XYZZ<FID>::load(Y[base + 1]);
        y.template add<kLatTail>(XYZZ<FID>::load(D[base + 1]));
        y.template add<kLatTail>(XYZZ<FID>::load(Y[base]));
      }
      y.store(Y_out[o]);
    }
  }
};

}  // namespace nmx
