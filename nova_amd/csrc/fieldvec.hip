// fieldvec.hip -- HBM-bound field-vector kernels either side of the MSM in Nova's prover.
//
//   AxpyFn       out = a + r*b              NIFS witness fold  W = W1 + r*W2, E = E1 + r*T
//                                           (/root/reference/src/r1cs/mod.rs:1058-1067)
//   Axpy2Fn      out = a + r*b + r^2*c      relaxed fold       E = E1 + r*T + r^2*E2      (r1cs/mod.rs:1096-1101)
//   CrossTermFn  out = az*bz - u*cz - e     commit_T's T       (r1cs/mod.rs:614-620)
//   VecAddFn     out = a + b                Z = Z1 + Z2        (r1cs/mod.rs:590-609)
//   BindTopFn    out = lo + r*(hi - lo)     MLE bind_poly_var_top (/root/reference/src/spartan/polys/multilinear.rs:65-84)
//                                           and the HyperKZG halving Pi[j] = P[2j] + x*(P[2j+1] - P[2j]) with
//                                           stride 2 (/root/reference/src/provider/hyperkzg.rs:1085-1095)
//
// One field element (32 B) per lane, two global_load_dwordx4 per operand, fully coalesced; 1-3 modmuls per 96-160
// bytes, so these are bound by HBM, not by the multiplier (unlike the MSM).  Linear maps commute with the
// Montgomery factor, so vectors are processed in whatever form they arrive in (canonical, or the reference's
// R = 2^256 Montgomery limbs) with the challenge pre-scaled once on the host; only the product az*bz needs a
// form-dependent constant.  Results stay in HBM so the next MSM (`NMX_SCALARS_DEVICE`) reads them in place.
#include "runtime.hpp"
#include "host_fp4.hpp"

namespace nmx {

template <int FID> NMX_HD Fp<FID> ld(const uint32_t* p, size_t i) { return Fp<FID>::from_words(p + 8 * i); }
template <int FID> NMX_HD void st(uint32_t* p, size_t i, const Fp<FID>& v) { v.canon().to_words(p + 8 * i); }

template <int FID> struct AxpyFn {
  const uint32_t *a, *b;
  uint32_t* out;
  Fp<FID> r;  // r * 2^261, canonical
  NMX_HD void operator()(uint32_t i) const { st<FID>(out, i, (ld<FID>(a, i) + r * ld<FID>(b, i)).norm()); }
};
template <int FID> struct Axpy2Fn {
  const uint32_t *a, *b, *c;
  uint32_t* out;
  Fp<FID> r, r2;  // r * 2^261, r^2 * 2^261
  NMX_HD void operator()(uint32_t i) const {
    st<FID>(out, i, (ld<FID>(a, i) + Fp<FID>::mul_add(r, ld<FID>(b, i), r2, ld<FID>(c, i))).norm());  // one reduction for both products
  }
};
template <int FID> struct CrossTermFn {
  const uint32_t *az, *bz, *cz, *e;
  uint32_t* out;
  Fp<FID> u;  // (p - u) * 2^261: the product u*cz enters negated, so that az*bz*k - u*cz is ONE reduction (mul_add)
  Fp<FID> k;  // 2^522 / F: brings (az*F)(bz*F)/2^261 back to az*bz*F  (F = 1 canonical, 2^256 Montgomery)
  NMX_HD void operator()(uint32_t i) const {
    using F = Fp<FID>;
    F ab = ld<FID>(az, i) * ld<FID>(bz, i);              // < 1.01 p
    F t = F::mul_add(ab, k, ld<FID>(cz, i), u);          // az bz k - u cz   < 1.02 p
    t = F::sub2(t, ld<FID>(e, i)).norm();                // - e + 2p   (e canonical)
    st<FID>(out, i, t);
  }
};
// commit_T_relaxed's term (src/r1cs/mod.rs:652-659): az*bz - u*cz - e1 - e2, both error vectors in the one pass
template <int FID> struct CrossTerm2Fn {
  const uint32_t *az, *bz, *cz, *e1, *e2;
  uint32_t* out;
  Fp<FID> u, k;  // as CrossTermFn (u negated)
  NMX_HD void operator()(uint32_t i) const {
    using F = Fp<FID>;
    F ab = ld<FID>(az, i) * ld<FID>(bz, i);              // < 1.01 p
    F t = F::mul_add(ab, k, ld<FID>(cz, i), u);          // az bz k - u cz          < 1.02 p
    F es = (ld<FID>(e1, i) + ld<FID>(e2, i)).norm();     // e1 + e2 (canonical)     < 2 p
    t = F::sub4(t, es).norm();                           // - (e1 + e2) + 4p        < 5.1 p  (canon() takes < 16 p)
    st<FID>(out, i, t);
  }
};
template <int FID> struct VecAddFn {
  const uint32_t *a, *b;
  uint32_t* out;
  NMX_HD void operator()(uint32_t i) const { st<FID>(out, i, (ld<FID>(a, i) + ld<FID>(b, i)).norm()); }
};
template <int FID> struct BindTopFn {
  const uint32_t *lo, *hi;  // element i of each, `stride` elements apart
  uint32_t* out;
  Fp<FID> r;  // r * 2^261
  uint32_t stride;
  NMX_HD void operator()(uint32_t i) const {
    using F = Fp<FID>;
    F l = ld<FID>(lo, (size_t)i * stride), h = ld<FID>(hi, (size_t)i * stride);
    st<FID>(out, i, (l + r * F::sub2(h, l).norm()).norm());
  }
};

// EqPolynomial::evals_from_points, one doubling step (/root/reference/src/spartan/polys/eq.rs:54-73):
//   y = x[i] * r;  x[i + size] = y;  x[i] -= y        for i < size
template <int FID> struct EqStepFn {
  uint32_t* buf;
  Fp<FID> r;  // r * 2^261
  uint32_t size;
  NMX_HD void operator()(uint32_t i) const {
    using F = Fp<FID>;
    F x = ld<FID>(buf, i);
    F y = r * x;                                   // < 1.01 p
    st<FID>(buf, (size_t)i + size, y);
    st<FID>(buf, i, F::sub2(x, y).norm());         // x - y + 2p
  }
};

// Small eq tables in ONE launch: out[x] = prod_i (x_i ? r_i : 1 - r_i), x_0 the most significant bit, ell <= 12
// (the sqrt-size tables of evaluate_with: ell doubling launches are ~10 us each, this is one).  ell modmuls per entry
// instead of one -- irrelevant at <= 4096 entries.
template <int FID> struct EqDirectFn {
  static constexpr uint32_t kMaxEll = 12;
  uint32_t* out;
  Fp<FID> r[kMaxEll], nr[kMaxEll];  // r_i * 2^261 and (1 - r_i) * 2^261, canonical
  Fp<FID> one;                      // ONE in the vectors' form (1 or 2^256 mod p), as a plain residue
  uint32_t ell;
  NMX_HD void operator()(uint32_t x) const {
    Fp<FID> acc = one;
    for (uint32_t i = 0; i < ell; i++) acc = acc * (((x >> (ell - 1 - i)) & 1u) ? r[i] : nr[i]);
    st<FID>(out, x, acc);
  }
};

// Both sqrt-size tables of an evaluation (multilinear.rs:141-147) in ONE launch: lanes [0, 2^ellL) build the left table, the rest
// the right one (a 2^20 evaluation was four launches -- two tables, the pass, the final sum -- for 28 us; now two).
template <int FID> struct EqDirect2Fn {
  uint32_t *outL, *outR;
  Fp<FID> r[2 * EqDirectFn<FID>::kMaxEll], nr[2 * EqDirectFn<FID>::kMaxEll];  // left challenges, then the right ones
  Fp<FID> one;
  uint32_t ellL, ellR;
  NMX_HD void operator()(uint32_t g) const {
    const bool right = g >= (1u << ellL);
    const uint32_t x = right ? g - (1u << ellL) : g, ell = right ? ellR : ellL, o = right ? ellL : 0u;
    Fp<FID> acc = one;
    for (uint32_t i = 0; i < ell; i++) acc = acc * (((x >> (ell - 1 - i)) & 1u) ? r[o + i] : nr[o + i]);
    st<FID>(right ? outR : outL, x, acc);
  }
};

// batch_invert (src/spartan/mod.rs:54-118: Montgomery's trick, chunked over threads there).  Here a lane owns the STRIDED chunk
// {c, c + T, c + 2T, ...} of K elements (T = number of chunks: consecutive lanes touch consecutive elements), a forward pass
// leaves every element's prefix product in `out` and the chunk products in a vector that is inverted the same way one level up
// (the last level, <= 128 products, on the host: one extended-Euclid inversion), and a backward pass per level turns prefix
// products into inverses.  All products are taken on the stored words as they are (canonical or Montgomery): the form factor is
// a scale the top level applies once, and every lower level inherits it through the chunk inverses.
// Cost: 3 products per element at level 0 (the trick's minimum) + 3/K of that above it -- VALU-bound for long vectors (2^24:
// 52 ps per element = 3.1 products at the multiplier's ~58 G/s; the 160 B per element of traffic would take 20 ps), and a chain
// of 3K DEPENDENT products per level for short ones, which is why K shrinks with the level's size (binv_chunk).
static constexpr uint32_t kBinvHostBelow = 128;
static inline uint32_t binv_chunk(size_t n) { return n >= (1u << 23) ? 32 : n >= (1u << 21) ? 16 : 8; }
template <int FID> struct BatchInvFwdFn {
  const uint32_t* v;
  uint32_t *prefix, *chunk_prod;
  uint32_t n, T, K;
  NMX_HD void operator()(uint32_t c) const {
    Fp<FID> acc = Fp<FID>::one();
    for (uint32_t j = 0; j < K; j++) {
      const uint64_t i = (uint64_t)c + (uint64_t)j * T;
      if (i >= n) break;
      st<FID>(prefix, i, acc);
      acc = acc * ld<FID>(v, i);
    }
    st<FID>(chunk_prod, c, acc);
  }
};
template <int FID> struct BatchInvBwdFn {
  const uint32_t *v, *chunk_inv;
  uint32_t* prefix;  // in: prefix products; out: the inverses
  uint32_t n, T, K;
  NMX_HD void operator()(uint32_t c) const {
    Fp<FID> acc = ld<FID>(chunk_inv, c);
    uint32_t cnt = 0;
    while (cnt < K && (uint64_t)c + (uint64_t)cnt * T < n) cnt++;
    for (uint32_t j = cnt; j-- > 0;) {
      const uint64_t i = (uint64_t)c + (uint64_t)j * T;
      const Fp<FID> p = ld<FID>(prefix, i), w = ld<FID>(v, i);
      st<FID>(prefix, i, acc * p);
      acc = (acc * w).canon();
    }
  }
};

// Large eq tables (13 <= ell <= 24) in TWO launches instead of ell doubling steps (a 2^20 table was 20 dependent launches,
// ~130 us, for 32 MB of output: profiles/r05_spartan): the sqrt-size tables of the two halves of the point -- the left one in the
// INTERNAL form, so that one product per entry gives out[x] = L[x >> ellR] * R[x & mask] in the vectors' own form -- and the
// product pass.  compute_eval_table_sparse's operand (src/spartan/snark.rs:182) is such a table.
template <int FID> struct EqSplit2Fn {
  uint32_t *outL, *outR;
  Fp<FID> r[2 * EqDirectFn<FID>::kMaxEll], nr[2 * EqDirectFn<FID>::kMaxEll];
  Fp<FID> oneL, oneR;
  uint32_t ellL, ellR;
  NMX_HD void operator()(uint32_t g) const {
    const bool right = g >= (1u << ellL);
    const uint32_t x = right ? g - (1u << ellL) : g, ell = right ? ellR : ellL, o = right ? ellL : 0u;
    Fp<FID> acc = right ? oneR : oneL;
    for (uint32_t i = 0; i < ell; i++) acc = acc * (((x >> (ell - 1 - i)) & 1u) ? r[o + i] : nr[o + i]);
    st<FID>(right ? outR : outL, x, acc);
  }
};
template <int FID> struct EqProductFn {
  const uint32_t *L, *R;
  uint32_t* out;
  uint32_t shift, mask;
  NMX_HD void operator()(uint32_t x) const { st<FID>(out, x, ld<FID>(L, x >> shift) * ld<FID>(R, x & mask)); }
};

// Coefficient classes, the GPU form of the reference's PrecomputedSparseMatrix (src/r1cs/sparse.rs:19-199: +-1 entries
// are added / subtracted, |k| <= 7 by repeated doubling, the rest multiplied).  R1CS matrices are almost all +-1: here
// the class rides in the top four bits of the 32-bit column index (columns < 2^28), so a unit or small entry costs
// 4 B of matrix traffic instead of 36 B and its 32-byte coefficient is never read -- SpMV is gather-bound, the saving
// is bytes, not multiplications.  Classes: 0 general, 1 +1, 2 -1, 3..8 +2..+7, 9..14 -2..-7.
static constexpr uint32_t kSpmvColBits = 28;
template <int FID> struct SpmvClassifyFn {  // after the coefficients are in internal form; writes the class into indices
  const uint32_t* data;
  uint32_t* indices;
  NMX_HD void operator()(uint32_t k) const {
    using F = Fp<FID>;
    uint32_t w[8], m[8];
    ld<FID>(data, k).to_canonical().to_words(w);
    uint32_t hi = 0;
    for (int j = 1; j < 8; j++) hi |= w[j];
    uint32_t cls = 0;
    if (hi == 0 && w[0] >= 1 && w[0] <= 7) cls = w[0] == 1 ? 1u : w[0] + 1u;  // +1 -> 1, +2..+7 -> 3..8
    if (!cls) {  // p - value small?
      uint64_t bw = 0;
      uint32_t mh = 0;
      for (int j = 0; j < 8; j++) {
        const uint64_t d = (uint64_t)FpParams<FID>::PW[j] - w[j] - bw;
        m[j] = (uint32_t)d;
        bw = (d >> 32) & 1u;
        if (j) mh |= m[j];
      }
      if (mh == 0 && m[0] >= 1 && m[0] <= 7) cls = m[0] == 1 ? 2u : m[0] + 7u;  // -1 -> 2, -2..-7 -> 9..14
    }
    indices[k] |= cls << kSpmvColBits;
  }
};
// coefficient class `cls` (>= 1) applied to z: a value < p, canonical.  z is any 256-bit value: it is reduced first (a
// z >= p -- the general Montgomery path reduces those correctly too -- would otherwise leave k z beyond canon()'s 16 p)
template <int FID> NMX_HD Fp<FID> spmv_small_term(uint32_t cls, const Fp<FID>& z_any) {
  using F = Fp<FID>;
  const uint32_t k = cls <= 2 ? 1u : (cls <= 8 ? cls - 1u : cls - 7u);  // |coefficient|
  const F zf = z_any.canon();                         // 2^256 < 6 p for all four fields
  F t;
#pragma unroll
  for (int i = 0; i < 9; i++) t.l[i] = zf.l[i] * k;  // z canonical: limbs < 2^29, k <= 7
  t = t.norm();                                       // value < 7 p
  if (cls == 2 || cls >= 9) t = F::sub8(F::zero(), t).norm();  // 8p - k z
  return t.canon();
}

// CSR sparse matrix x vector, one row per lane (src/r1cs/sparse.rs:201-229 multiply_vec).  Matrix values are stored in
// internal form at registration, so data * z comes out in z's own form with no correction.
// one row of M z, normalised (< 16 p: canon() brings it to the stored form)
template <int FID>
NMX_HD Fp<FID> spmv_row(const uint32_t* indptr, const uint32_t* indices, const uint32_t* data, const uint32_t* z, uint32_t colmask,
                        uint32_t row) {
  using F = Fp<FID>;
  F acc = F::zero();
  uint32_t pending = 0;
  for (uint32_t k = indptr[row]; k < indptr[row + 1]; k++) {
    const uint32_t w = indices[k], cls = (w & ~colmask) >> kSpmvColBits;
    const F zf = ld<FID>(z, w & colmask);
    acc = acc + (cls ? spmv_small_term<FID>(cls, zf) : ld<FID>(data, k) * zf);
    if (++pending == 6) {
      acc = acc.norm().canon();
      pending = 0;
    }
  }
  return acc.norm();
}
template <int FID> struct SpmvFn {
  const uint32_t* indptr;   // rows + 1
  const uint32_t* indices;  // nnz: column | class << 28 (class 0 everywhere when the matrix is not tagged)
  const uint32_t* data;     // nnz x 8, internal form
  const uint32_t* z;        // cols x 8
  uint32_t* out;            // rows x 8
  uint32_t colmask;         // 2^28 - 1 (tagged) or all ones
  NMX_HD void operator()(uint32_t row) const { st<FID>(out, row, spmv_row<FID>(indptr, indices, data, z, colmask, row)); }
};

// M^T x over VIRTUAL rows (compute_eval_table_sparse, /root/reference/src/spartan/mod.rs:497-533: M_evals[col] += rx[row] * val
// for every entry -- a scatter-add on the reference's side, a gather over the transposed matrix here).  The CSC arrays are cut
// into virtual rows at registration of the transposed form (Global::SparseSet::Transposed): short columns are one virtual row
// that writes its output element; a long column (the constant-one column of an R1CS matrix has an entry per constraint) is
// several, each writing a partial that k_spmv_heavy adds up (one block per split column) -- no lane walks more than 32 entries.
template <int FID> struct SpmvSegFn {
  const uint32_t* vptr;     // nvirt + 1
  const uint32_t* indices;  // nnz: row of M | class << 28
  const uint32_t* data;     // nnz x 8, internal form
  const uint32_t* x;        // rows of M x 8
  const uint32_t* vout;     // nvirt: output row, or 2^31 | partial index
  uint32_t *out, *partial;
  uint32_t colmask;
  NMX_HD void operator()(uint32_t v) const {
    const Fp<FID> acc = spmv_row<FID>(vptr, indices, data, x, colmask, v);
    const uint32_t o = vout[v];
    if (o & 0x80000000u) st<FID>(partial, o & 0x7fffffffu, acc);
    else st<FID>(out, o, acc);
  }
};
// one split column per BLOCK: 256 lanes add up its partials (the constant-one column of a 2^20-constraint matrix leaves thousands),
// a shuffle / LDS tree joins them.  Additions only.
template <int FID> __global__ __launch_bounds__(256) void k_spmv_heavy(const uint32_t* hrow, const uint32_t* hstart, const uint32_t* partial, uint32_t* out) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[9 * 256];
  const uint32_t b = hstart[blockIdx.x], e = hstart[blockIdx.x + 1], t = threadIdx.x;
  F acc = F::zero();
  uint32_t pending = 0;
  for (uint32_t k = b + t; k < e; k += 256u) {
    acc = acc + ld<FID>(partial, k);
    if (++pending == 6) {
      acc = acc.norm().canon();
      pending = 0;
    }
  }
  acc = acc.norm().canon();
#pragma unroll
  for (int i = 0; i < 9; i++) lds[i * 256 + t] = acc.l[i];
  __syncthreads();
  for (uint32_t s = 128; s >= 1; s >>= 1) {
    if (t < s) {
      F o;
#pragma unroll
      for (int i = 0; i < 9; i++) o.l[i] = lds[i * 256 + t + s];
      acc = (acc + o).norm().canon();
#pragma unroll
      for (int i = 0; i < 9; i++) lds[i * 256 + t] = acc.l[i];
    }
    __syncthreads();
  }
  if (t == 0) st<FID>(out, hrow[blockIdx.x], acc);
}

// commit_T in one pass over the rows (src/r1cs/mod.rs:612-620): T[row] = (A z)[row] (B z)[row] - u (C z)[row] - E[row].  The
// three products and the cross term of CrossTermFn without AZ, BZ, CZ ever reaching HBM, and one launch instead of four;
// bit-identical to the separate calls (each row product is canonicalised exactly as SpmvFn's store does).
template <int FID> struct SpmvCrossFn {
  const uint32_t *ipA, *ixA, *dA, *ipB, *ixB, *dB, *ipC, *ixC, *dC;
  const uint32_t *z, *e;
  uint32_t* out;
  uint32_t colmask;
  Fp<FID> u, k;  // as CrossTermFn: (p - u) * 2^261, and 2^522 / F
  NMX_HD void operator()(uint32_t row) const {
    using F = Fp<FID>;
    const F az = spmv_row<FID>(ipA, ixA, dA, z, colmask, row).canon();
    const F bz = spmv_row<FID>(ipB, ixB, dB, z, colmask, row).canon();
    const F cz = spmv_row<FID>(ipC, ixC, dC, z, colmask, row).canon();
    const F ab = az * bz;
    F t = F::mul_add(ab, k, cz, u);
    t = F::sub2(t, ld<FID>(e, row)).norm();
    st<FID>(out, row, t);
  }
};
// NIFS fold in one launch (src/r1cs/mod.rs:1058-1067): W = W1 + r W2 over n_w elements, E = E1 + r T over n_e
template <int FID> struct FoldPairFn {
  const uint32_t *w1, *w2, *e1, *t;
  uint32_t *w, *e;
  uint32_t n_w;
  Fp<FID> r;
  NMX_HD void operator()(uint32_t i) const {
    if (i < n_w) st<FID>(w, i, (ld<FID>(w1, i) + r * ld<FID>(w2, i)).norm());
    else st<FID>(e, i - n_w, (ld<FID>(e1, i - n_w) + r * ld<FID>(t, i - n_w)).norm());
  }
};

// (M*z1, M*z2) in one pass over the matrix (sparse.rs:215-229, multiply_vec_pair: AZ/BZ/CZ of two instances at once)
template <int FID> struct SpmvPairFn {
  const uint32_t* indptr;
  const uint32_t* indices;
  const uint32_t* data;
  const uint32_t *z1, *z2;
  uint32_t *out1, *out2;
  uint32_t colmask;
  NMX_HD void operator()(uint32_t row) const {
    using F = Fp<FID>;
    F a1 = F::zero(), a2 = F::zero();
    uint32_t pending = 0;
    for (uint32_t k = indptr[row]; k < indptr[row + 1]; k++) {
      const uint32_t w = indices[k], cls = (w & ~colmask) >> kSpmvColBits, col = w & colmask;
      const F y1 = ld<FID>(z1, col), y2 = ld<FID>(z2, col);
      if (cls) {
        a1 = a1 + spmv_small_term<FID>(cls, y1);
        a2 = a2 + spmv_small_term<FID>(cls, y2);
      } else {
        const F m = ld<FID>(data, k);
        a1 = a1 + m * y1;
        a2 = a2 + m * y2;
      }
      if (++pending == 6) {
        a1 = a1.norm().canon();
        a2 = a2.norm().canon();
        pending = 0;
      }
    }
    st<FID>(out1, row, a1.norm());
    st<FID>(out2, row, a2.norm());
  }
};

// out[i] = sum_j w[j] * v_j[i], vectors shorter than the output read as zero-padded: PolyEvalWitness::batch and
// batch_diff_size with w[j] = s^j (/root/reference/src/spartan/mod.rs:165-277), the random linear combination in
// front of the batched PCS opening.  (k + 1) x 32 B per element against k modmuls: HBM-bound for every k.
template <int FID> struct LinCombFn {
  const uint64_t* vecs;  // k device addresses
  const uint64_t* lens;  // k element counts
  const uint32_t* w;     // k x 8, internal form (s^j * 2^261), canonical
  uint32_t* out;
  uint32_t k;
  NMX_HD void operator()(uint32_t i) const {
    using F = Fp<FID>;
    // four products under one reduction (Fp::dot: 4 x 81 + 81 multiply-adds instead of 4 x 162; the weights are wave-uniform):
    // the kernel was multiplier-bound at k = 8 (329 us at 2^22 against a VALU floor of 314 us, profiles/r03_fieldvec/lincomb8_pmc.json)
    F acc = F::zero();
    uint32_t pending = 0;
    for (uint32_t j = 0; j < k; j += 4) {
      F v[4], c[4];
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        const bool on = j + q < k && i < lens[j + q < k ? j + q : j];
        v[q] = on ? ld<FID>((const uint32_t*)(uintptr_t)vecs[j + q], i) : F::zero();
        c[q] = j + q < k ? ld<FID>(w, j + q) : F::zero();
      }
      acc = acc + F::template dot<4, true>(v, c);  // < 1.04 p each
      if (++pending == 6) {
        acc = acc.norm().canon();
        pending = 0;
      }
    }
    st<FID>(out, i, acc.norm());
  }
};

// (A tiled variant -- 2048-coefficient tiles transposed through LDS, block scan of lane heads -- was built and measured in
// round 2: bit-exact but 17 % slower at 2^24, see profiles/r02_fieldvec/horner_tiled_rejected.txt.)
// Suffix Horner  out[i] = sum_{k >= i} f[k] * u^(k-i):  out[0] is `poly_eval(f, u)` (hyperkzg.rs:1011-1020) and out[1..]
// is the quotient of `div_by_monomial` (hyperkzg.rs:961-999: h[i-1] = f[i] + h[i]*u).  Same three phases as the
// reference's chunked version -- chunk-local recurrences, carries between chunks with u^chunk, fix-up -- with
// 16-element chunks (one lane each: the chains are pure latency, so they are short -- 64-element chunks took 0.20 ms
// for 2^20 coefficients, 16-element ones 0.12 ms) and the carry phase applied recursively.
static constexpr uint32_t kHornerChunk = 16;
// a coefficient as the chains want it: below p.  Any 256-bit word is accepted (as by the other field kernels); one >= p -- up
// to 5.3 p -- is reduced here, behind one compare chain, so that f + u t stays below the 4 p of the stored copies' canon4.
template <int FID> NMX_HD Fp<FID> horner_coeff(const uint32_t* w) {
  Fp<FID> v = Fp<FID>::from_words(w);
  if (!Fp<FID>::words_lt_p(w)) v = v.canon();
  return v;
}
template <int FID> struct HornerLocalFn {
  const uint32_t* f;
  uint32_t* out;    // local suffix values
  uint32_t* heads;  // heads[c] = local value at the first element of chunk c
  Fp<FID> u;        // u * 2^261
  uint32_t n;
  NMX_HD void operator()(uint32_t c) const {
    using F = Fp<FID>;
    const uint32_t lo = c * kHornerChunk, hi = lo + kHornerChunk < n ? lo + kHornerChunk : n;
    // t < 2.02 p throughout (f < p, u t < p (1 + 2.02 / 127)): the chain runs on the weakly reduced value and only the
    // stored copy is canonicalised, with the two subtractions a value below 4p needs
    F t = F::zero();
    for (uint32_t i = hi; i-- > lo;) {
      t = (horner_coeff<FID>(f + 8 * (size_t)i) + u * t).norm();
      t.canon4().to_words(out + 8 * (size_t)i);
    }
    t.canon4().to_words(heads + 8 * (size_t)c);
  }
};
template <int FID> struct HornerFixFn {
  uint32_t* out;
  const uint32_t* carries;  // carries[c] = global suffix value at the start of chunk c
  const uint32_t* pw;       // pw[k] = u^k * 2^261, k = 0..16
  uint32_t n;
  NMX_HD void operator()(uint32_t i) const {
    using F = Fp<FID>;
    const uint32_t c = i / kHornerChunk, nc = (n + kHornerChunk - 1) / kHornerChunk;
    if (c + 1 >= nc) return;  // last chunk: local values are already global
    const uint32_t dist = (c + 1) * kHornerChunk - i;  // 1..16
    F v = ld<FID>(out, i) + ld<FID>(pw, dist) * ld<FID>(carries, c + 1);
    st<FID>(out, i, v.norm());
  }
};

// Top level for long inputs: the chunk-per-lane kernels above write every local value 512 B away from its neighbour lane's
// (32 of the 128 bytes of a line per store) and read it back in the fix-up -- 2 GB of traffic and mostly partial lines at
// 2^24.  Here a lane holds its 8 coefficients in registers (all 16 loads issued before the dependent chain: both of its
// 128-byte lines are consumed while they are in flight), the first pass keeps only the chunk head, the carries come from
// the recursion above run over the heads at u^8, and the second pass re-walks the chunk from the carry and stores its
// 256 bytes back to back: 96 B and 2 multiplications per coefficient, no power table.  Applied level after level while a
// level has >= 2^15 elements (K = chunk: 8, or 4 = one 128-byte line per lane); the short levels use the kernels above.
template <int FID, uint32_t K> struct HornerHeadFn {
  const uint32_t* f;
  uint32_t* heads;  // heads[c] = sum_{k < K} f[K c + k] u^k
  Fp<FID> u;
  uint32_t n;
  NMX_HD void operator()(uint32_t c) const {
    using F = Fp<FID>;
    const uint32_t lo = c * K, cnt = n - lo < K ? n - lo : K;
    F t = F::zero();
    if (cnt == K) {
      uint32_t w[K][8];
#pragma unroll
      for (uint32_t k = 0; k < K; k++)
#pragma unroll
        for (int j = 0; j < 8; j++) w[k][j] = f[8 * (size_t)(lo + k) + j];
#pragma unroll
      for (uint32_t k = K; k-- > 0;) t = (horner_coeff<FID>(w[k]) + u * t).norm();
    } else {
      for (uint32_t k = cnt; k-- > 0;) t = (horner_coeff<FID>(f + 8 * (size_t)(lo + k)) + u * t).norm();
    }
    t.canon4().to_words(heads + 8 * (size_t)c);  // t < 2.02 p
  }
};
template <int FID, uint32_t K> struct HornerWalkFn {
  const uint32_t* f;
  const uint32_t* carries;  // carries[c] = global suffix value at the start of chunk c
  uint32_t* out;
  Fp<FID> u;
  uint32_t n, nc;
  NMX_HD void operator()(uint32_t c) const {
    using F = Fp<FID>;
    const uint32_t lo = c * K, cnt = n - lo < K ? n - lo : K;
    F t = c + 1 < nc ? ld<FID>(carries, c + 1) : F::zero();
    if (cnt == K) {
      uint32_t w[K][8];
#pragma unroll
      for (uint32_t k = 0; k < K; k++)
#pragma unroll
        for (int j = 0; j < 8; j++) w[k][j] = f[8 * (size_t)(lo + k) + j];
#pragma unroll
      for (uint32_t k = K; k-- > 0;) {  // t < 2.02 p (carry canonical): the chain stays weakly reduced
        t = (horner_coeff<FID>(w[k]) + u * t).norm();
        t.canon4().to_words(w[k]);
      }
#pragma unroll
      for (uint32_t k = 0; k < K; k++)
#pragma unroll
        for (int j = 0; j < 8; j++) out[8 * (size_t)(lo + k) + j] = w[k][j];
    } else {
      for (uint32_t k = cnt; k-- > 0;) {
        t = (horner_coeff<FID>(f + 8 * (size_t)(lo + k)) + u * t).norm();
        t.canon4().to_words(out + 8 * (size_t)(lo + k));
      }
    }
  }
};

// ----------------------------------------------------------------------------------------------------
// Single-pass suffix Horner (round 3, second half): ONE kernel reads every coefficient once and writes every result once
// (64 B per coefficient, all of it in whole 1 KiB wave transactions), with the carries handed from tile to tile by a
// two-level decoupled look-back instead of a recursion of launches.
//
//   tile    = J sub-tiles of 512 coefficients (J = 1, 2, 4 by input size), owned by ONE wave (no block barrier anywhere); of the
//             sub-tile in hand lane l holds coefficients 8l .. 8l+7 in registers.  Its 16 KiB arrive as 16 coalesced 16-byte
//             loads per lane and are transposed through 8.5 KiB of LDS per wave in two halves (row stride 272 B: conflict-free
//             both ways); results leave the same way.  With J > 1 the local phase runs over all sub-tiles first (the last one
//             stays in registers), the look-backs run once, and the walk re-reads the lower sub-tiles (L2 / Infinity Cache).
//   local   : head_l = sum_k f[8l+k] u^k (two dot products with one reduction each, Fp::dot), H_l = u^(8l) head_l (per-lane constant U_l), then a suffix
//             SUM over the lanes -- additions, not products, because every term already carries its power of u:
//             S_l = sum_{m >= l} H_m, parked in LDS.  A = sum_j u^(512 j) S_0 of sub-tile j = sum_k f[k] u^k (k tile-local) is the
//             tile's aggregate, published at once.
//   carry   : C = out[first element of the next tile], the suffix value at the tile's end.  Tiles form GROUPS of 64
//             (32 768 coefficients).  Inside the group: C1 = sum_m u^(512 m) A_(tile+1+m) over the group's later tiles -- one
//             product per lane with the per-lane constant W_m and a wave sum; those tiles were dispatched earlier and
//             publish A before they wait for anything.  Across groups: the wave that draws a group's last ticket folds the 64
//             aggregates into the group aggregate GA (status 1) and, once it knows the carry GC at the group's end, publishes
//             the group's inclusive value GI = GA + u^32768 GC (status 2); every tile of group g gets GC by looking back over
//             the groups behind it, lane m at group g + 1 + m:  GC = sum_{m < j} X_m GA_(g+1+m) + X_j GI_(g+1+j)  for the first j
//             with an inclusive value (X_m = u^(32768 m)), further rounds of 64 groups if there is none.  Then
//             C = C1 + u^(512 (63 - position in group)) GC.  (A first version looked back over TILES only: with ~3000 tiles
//             in flight the nearest inclusive value was ~25 rounds of 64 away, each round a poll + 64 loads + product + wave sum,
//             and the kernel was 3 x slower than the two-pass form from 2^20 on -- profiles/r03_fieldvec/horner_scan.txt.)
//             Tiles are numbered against the dispatch order (block 0 owns the LAST tile), so a wave only ever waits for
//             waves dispatched before it.
//   walk    : lane l re-walks its chunk from  c_l = u^(-8(l+1)) (S_(l+1) + u^512 C)  (per-lane constant V_l) and stores
//             canonical values.
// Per 512 coefficients: 3.5 product-equivalents for the heads, 8 for the walk, 3 more (H, the sub-tile's carry, V); per tile 4-6
// for the two look-backs -- 721 VALU instructions per coefficient at J = 1 against ~700 + the recursion for the two-pass kernels
// above, but no partial-line traffic and no dependent launches.  u = 0 is left to those kernels; the
// per-lane constants come from a small kernel (k_horner_tables) that also clears the tile / group states.
// Bounds (p(1 + ab/127) for a product of values < ap, < bp; input words may be any 256-bit value, < 5.3p): head < 7.4p,
// H < 1.06p, S < 67.7p (published raw as A), W A < 1.54p, C1 < 96.6p, GA < 98.1p reduced to < 1.78p before it is published,
// X GA < 1.02p, GI < 3.6p, X GI < 1.03p, GC < 66p (one round; reduced to < 3.6p per further round), W GC < 1.53p, C < 98.2p,
// u^512 C < 1.78p, S + u^512 C < 69.5p, c_l < 1.55p; every sum is normalised before it is multiplied (limbs < 2^30 in,
// < 2^29 out).
static constexpr uint32_t kScanSub = 512, kScanMin = 1024, kScanGroup = 64;
static constexpr uint32_t kScanTblU = 0, kScanTblV = 64, kScanTblW = 128, kScanTblX = 193, kScanTblP = 258, kScanTblS = 264, kScanTblN = 268;  // W_0..W_64, X_0..X_64, u^2..u^7, u^(512 j) j < 4
template <int FID> struct HornerScanArgs {
  const uint32_t* f;
  uint32_t* out;
  const uint32_t* tbl;  // kScanTblN entries of 9 raw limbs (canonical internal residues)
  uint32_t* status;     // [ntiles] 0 nothing, 1 aggregate published
  uint32_t* gcnt;       // [ngroups] tickets drawn
  uint32_t* gstatus;    // [ngroups] 0 nothing, 1 aggregate, 2 inclusive
  uint32_t* agg;        // [ntiles][9] raw limbs
  uint32_t* gagg;       // [ngroups][9]
  uint32_t* ginc;       // [ngroups][9]
  Fp<FID> u;
  uint32_t n, ntiles, ngroups;
  uint32_t window;  // groups per look-back round, 1 .. 64 (64 in production; smaller values only to test the multi-round path)
  uint32_t* watchdog;  // two pinned host words: [0] set by a wave that polled kScanSpinLimit times in vain, [1] by a coefficient >= p (see horner_scan_t)
  uint32_t spin_limit;
  uint32_t* ticket;    // start-order counter of the waves (cleared with the flags)
  uint32_t order;      // 1: tiles by start-order ticket; 0: by block id (option horner_order)
};
template <int FID> struct HornerTblArgs {
  uint32_t* tbl;
  uint32_t* flags;  // status | gcnt | gstatus, contiguous
  Fp<FID> u1, u8, v8, uS, uT, uG;  // u, u^8, u^-8, u^512, u^T (T = coefficients per tile), u^(64 T) (internal, canonical)
  uint32_t nflags;
};
template <int FID> __global__ __launch_bounds__(256) void k_horner_tables(HornerTblArgs<FID> a) {
  using F = Fp<FID>;
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g < a.nflags) a.flags[g] = 0;
  if (blockIdx.x != 0) return;
  for (uint32_t t = threadIdx.x; t < kScanTblN; t += 256u) {
    const F base = t < kScanTblV ? a.u8 : (t < kScanTblW ? a.v8 : (t < kScanTblX ? a.uT : (t < kScanTblP ? a.uG : (t < kScanTblS ? a.u1 : a.uS))));
    const uint32_t e = t < kScanTblV ? t : (t < kScanTblW ? t - kScanTblV + 1 : (t < kScanTblX ? t - kScanTblW : (t < kScanTblP ? t - kScanTblX : (t < kScanTblS ? t - kScanTblP + 2 : t - kScanTblS))));  // <= 64
    F acc = F::one();
    for (int b = 6; b >= 0; b--) {
      acc = acc.sqr();
      const F m = acc * base;
      if ((e >> b) & 1u) acc = m;
    }
    acc = acc.canon();
#pragma unroll
    for (int i = 0; i < 9; i++) a.tbl[9 * t + i] = acc.l[i];
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
template <int FID> __device__ __forceinline__ Fp<FID> fp_ld_limbs(const uint32_t* p) {
  Fp<FID> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = p[i];
  return r;
}
template <int FID> __device__ __forceinline__ void fp_st_limbs(uint32_t* p, const Fp<FID>& v) {
#pragma unroll
  for (int i = 0; i < 9; i++) p[i] = v.l[i];
}
// value of lane + d (zero past the wave's end)
template <int FID> __device__ __forceinline__ Fp<FID> fp_from_above(const Fp<FID>& x, uint32_t d, uint32_t lane) {
  Fp<FID> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const uint32_t v = (uint32_t)__shfl_down((int)x.l[i], d, 64);
    r.l[i] = lane + d < 64u ? v : 0u;
  }
  return r;
}
// S_l = sum_{m >= l} x_m, normalised after every step (64 values of < 2^29-limbs stay far below the 32-bit limb)
template <int FID> __device__ __forceinline__ Fp<FID> wave_suffix_sum(Fp<FID> x, uint32_t lane) {
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1) x = (x + fp_from_above<FID>(x, d, lane)).norm();
  return x;
}
// lane 0 <- the sum of all 64 lanes (the other lanes end with partial sums or wrapped garbage: a lane past the wave's end reads
// itself).  Normalised every second step: limbs < 2^29 -> < 2^31 -> norm.
template <int FID> __device__ __forceinline__ Fp<FID> wave_total(Fp<FID> x) {
#pragma unroll
  for (uint32_t d = 32; d >= 1; d >>= 1) {
#pragma unroll
    for (int i = 0; i < 9; i++) x.l[i] += (uint32_t)__shfl_down((int)x.l[i], d, 64);
    if (d == 16 || d == 4 || d == 1) x = x.norm();
  }
  return x;
}
template <int FID> __device__ __forceinline__ Fp<FID> wave_bcast0(const Fp<FID>& x) {
  Fp<FID> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)x.l[i]);
  return r;
}
// LDS hand-over inside one wave: LDS instructions of a wave execute in order, the compiler must not move them across
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Hand-over between waves of different XCDs WITHOUT agent-scope fences: a release fence is an L2 write-back and an acquire
// fence an L2 invalidate, of the whole L2 -- with every wave streaming 16 KiB of results through it, 3-5 of them per tile
// serialised the kernel (0.53 ms at 2^22, profiles/r03_fieldvec/horner_scan.txt).  The few words that cross (aggregates,
// flags, tickets) are agent-scope relaxed atomics instead -- sc1 accesses that bypass the non-coherent L2 -- ordered by
// waiting for the data stores' acknowledgement before the flag store, and by issuing the data loads only after the flag
// has been seen.
template <int FID> __device__ __forceinline__ void desc_store(uint32_t* p, const Fp<FID>& v) {
#pragma unroll
  for (int i = 0; i < 9; i++) __hip_atomic_store(p + i, v.l[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int FID> __device__ __forceinline__ Fp<FID> desc_load(const uint32_t* p) {
  Fp<FID> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return r;
}
__device__ __forceinline__ void publish_flag(uint32_t* flag, uint32_t v) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t poll_flag(const uint32_t* flag) {
  return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void seen_barrier() { asm volatile("" ::: "memory"); }
#endif

template <int FID, int J> __global__ __launch_bounds__(256) void k_horner_scan(HornerScanArgs<FID> a) {
#if defined(__HIP_DEVICE_COMPILE__)
  using F = Fp<FID>;
  constexpr uint32_t T = J * kScanSub;  // coefficients per tile
  __shared__ uint4 lds_all[4][32 * 17];
  __shared__ uint32_t park_all[4][J * 9 * 64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // Position in START order, drawn from a ticket counter when the wave starts (as CUB's decoupled look-back does), not taken
  // from blockIdx: a wave then only ever waits for waves that were already RUNNING when it drew its ticket, whatever order the
  // dispatcher places blocks in -- forward progress no longer rests on "lower block ids become resident first", and the
  // watchdog below is a backstop for a fault, not for a scheduling assumption.
  uint32_t r = blockIdx.x * 4u + wave;  // order == 0: position in dispatch order (see HornerScanArgs::order)
  if (a.order) {  // one ticket per BLOCK (a ticket per wave -- 2048 same-address device atomics at 2^20 -- cost 15-19 us)
    __shared__ uint32_t blk;
    if (threadIdx.x == 0) blk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    r = blk * 4u + wave;
  }
  if (r >= a.ntiles) return;
  const uint32_t tile = a.ntiles - 1u - r;
  const uint32_t grp = tile / kScanGroup, pos = tile % kScanGroup;
  const uint32_t gtiles = a.ntiles - grp * kScanGroup < kScanGroup ? a.ntiles - grp * kScanGroup : kScanGroup;
  const bool last_group = grp + 1u == a.ngroups;
  uint4* my = lds_all[wave];
  uint32_t* park = park_all[wave];
  const size_t e0 = (size_t)tile * T;  // first element of the tile
  uint32_t w[8][8];                    // words of coefficient 8 lane + k of the sub-tile in hand

  // sub-tile j (512 coefficients) -> w: 16 coalesced 16-byte loads per lane, transposed through LDS in two halves
  auto load_sub = [&](uint32_t j) __attribute__((always_inline)) {
    const size_t e = e0 + (size_t)j * kScanSub;
    const uint4* src = (const uint4*)a.f + 2 * e;
    const uint32_t left = e >= a.n ? 0u : (a.n - e < kScanSub ? (uint32_t)(a.n - e) : kScanSub);
#pragma unroll
    for (uint32_t p = 0; p < 2; p++) {
      uint4 v[8];
#pragma unroll
      for (uint32_t qq = 0; qq < 8; qq++) {
        const uint32_t unit = (8u * p + qq) * 64u + lane;  // 16-byte unit of the sub-tile
        v[qq] = (unit >> 1) < left ? src[unit] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (uint32_t qq = 0; qq < 8; qq++) my[(4u * qq + (lane >> 4)) * 17u + (lane & 15u)] = v[qq];
      wave_lds_sync();
      if ((lane >> 5) == p) {
#pragma unroll
        for (uint32_t t = 0; t < 16; t++) {
          const uint4 x = my[(lane & 31u) * 17u + t];
          w[t >> 1][(t & 1u) * 4u + 0] = x.x;
          w[t >> 1][(t & 1u) * 4u + 1] = x.y;
          w[t >> 1][(t & 1u) * 4u + 2] = x.z;
          w[t >> 1][(t & 1u) * 4u + 3] = x.w;
        }
      }
      wave_lds_sync();
    }
  };
  auto store_sub = [&](uint32_t j) __attribute__((always_inline)) {
    const size_t e = e0 + (size_t)j * kScanSub;
    uint4* dst = (uint4*)a.out + 2 * e;
    const uint32_t left = e >= a.n ? 0u : (a.n - e < kScanSub ? (uint32_t)(a.n - e) : kScanSub);
#pragma unroll
    for (uint32_t p = 0; p < 2; p++) {
      if ((lane >> 5) == p) {
#pragma unroll
        for (uint32_t t16 = 0; t16 < 16; t16++)
          my[(lane & 31u) * 17u + t16] = make_uint4(w[t16 >> 1][(t16 & 1u) * 4u + 0], w[t16 >> 1][(t16 & 1u) * 4u + 1],
                                                    w[t16 >> 1][(t16 & 1u) * 4u + 2], w[t16 >> 1][(t16 & 1u) * 4u + 3]);
      }
      wave_lds_sync();
#pragma unroll
      for (uint32_t qq = 0; qq < 8; qq++) {
        const uint32_t unit = (8u * p + qq) * 64u + lane;
        const uint4 x = my[(4u * qq + (lane >> 4)) * 17u + (lane & 15u)];
        if ((unit >> 1) < left) dst[unit] = x;
      }
      wave_lds_sync();
    }
  };

  // with two waves per SIMD (J > 1) there are registers to spare: the per-lane constants are loaded once, not behind an L2
  // round trip in every sub-tile (J = 1 sits at the 168-register limit of three waves per SIMD and re-reads them)
  F Ulane, Vlane;
  if constexpr (J > 1) {
    Ulane = fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblU + lane));
    Vlane = fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblV + lane));
  }
  // ---- phase 1, sub-tile by sub-tile (the last one stays in registers): chunk heads, scaled, suffix sum over the lanes
  F A;  // lane 0: the tile's aggregate sum_k f[k] u^k
#pragma unroll 1
  for (uint32_t j = 0; j < (uint32_t)J; j++) {
    load_sub(j);
    // Coefficients must be below p for the walk's stored copies (canon4: two subtractions, valid below 4 p) to come out
    // canonical.  A word whose top 64 bits reach p's makes the whole call repeat on the two-pass kernels, which reduce such
    // words first (horner_coeff): any 256-bit input ends in canonical outputs, the hot path pays ~4 instructions per
    // coefficient and no cold block (an exact compare in a rarely taken branch cost 18 registers -- a wave per SIMD at J = 1).
    // A canonical element trips the test with probability < 2^-62 (its top two words equal p's).
    {
      constexpr uint32_t P7 = FpParams<FID>::PW[7], P6 = FpParams<FID>::PW[6];
      uint32_t sus = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) sus |= ((w[k][7] > P7) || (w[k][7] == P7 && w[k][6] >= P6)) ? 1u : 0u;
      if (sus) __hip_atomic_store(a.watchdog + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (a plain store: no PCIe atomics needed)
    }
    // head = f0 + (f1 u + f2 u^2 + f3 u^3 + f4 u^4) + (f5 u^5 + f6 u^6 + f7 u^7): seven products, TWO reductions, no dependent
    // chain (Fp::dot; the eight-step Horner form is 8 x 162 multiply-adds and 8 reductions, this one 7 x 81 + 2 x 81)
    F t;
    {
      const F fa[4] = {F::from_words(w[1]), F::from_words(w[2]), F::from_words(w[3]), F::from_words(w[4])};
      const F pa[4] = {a.u, fp_ld_limbs<FID>(a.tbl + 9u * kScanTblP), fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblP + 1u)),
                       fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblP + 2u))};
      t = F::template dot<4, true>(fa, pa);
    }
    if constexpr (J > 1) {  // the words stay live for the walk: unpack the second batch only after the first product
#pragma unroll
      for (int k = 5; k < 8; k++)
#pragma unroll
        for (int q = 0; q < 8; q++) asm volatile("" : "+v"(w[k][q]) : "v"(t.l[0]));
    }
    {
      const F fb[3] = {F::from_words(w[5]), F::from_words(w[6]), F::from_words(w[7])};
      const F pb[3] = {fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblP + 3u)), fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblP + 4u)),
                       fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblP + 5u))};
      t = t + F::template dot<3, true>(fb, pb);
    }
    if constexpr (J > 1) {
#pragma unroll
      for (int q = 0; q < 8; q++) asm volatile("" : "+v"(w[0][q]) : "v"(t.l[0]));
    }
    t = (F::from_words(w[0]) + t).norm();   // < 3.06 p (7.4 p for words >= p)
    const F H = t * (J > 1 ? Ulane : fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblU + lane)));  // u^(8 lane) head   < 1.06 p
    const F S = wave_suffix_sum<FID>(H, lane);                                  // < 67.7 p
    // S waits in LDS (stride-64 words: conflict-free) while the other sub-tiles and the look-backs run
#pragma unroll
    for (int i = 0; i < 9; i++) park[(j * 9u + i) * 64u + lane] = S.l[i];
    if (j == 0) A = S;
    else A = (A + S * fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblS + j))).norm();  // + u^(512 j) S   < 67.7 p + 1.54 p (J - 1)
  }
  wave_lds_sync();
  if (lane == 0) {  // the tile's aggregate, at once
    desc_store<FID>(a.agg + 9 * (size_t)tile, A);
    publish_flag(a.status + tile, 1u);
  }

  // ---- a ticket of the group; the wave that draws the last one folds the group's aggregates
  uint32_t ticket = 0;
  if (lane == 0) ticket = __hip_atomic_fetch_add(a.gcnt + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // after the aggregate's acknowledgement (publish_flag)
  ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
  const bool closer = ticket + 1u == gtiles;
  seen_barrier();
  if (closer) {
    F val = F::zero();
    if (lane < gtiles) val = desc_load<FID>(a.agg + 9 * ((size_t)grp * kScanGroup + lane));
    const F term = val * fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblW + lane));      // < 1.57 p
    const F GA = wave_total<FID>(term) * F::one();                                 // lane 0: the group's aggregate, < 1.8 p
    if (lane == 0) {
      desc_store<FID>((last_group ? a.ginc : a.gagg) + 9 * (size_t)grp, GA);
      publish_flag(a.gstatus + grp, last_group ? 2u : 1u);
    }
  }

  // ---- inside the group: the aggregates of its later tiles
  F C = F::zero();
  const uint32_t nlook = gtiles - 1u - pos;
  // A wave only ever waits for waves dispatched before it, which need nothing from it: the polls below end.  Should that
  // order ever fail to hold (it is the dispatcher's, not the language's), a wave that has polled spin_limit times gives up,
  // says so in a pinned host word and lets everything behind it run on: the host then repeats the call on the two-pass
  // kernels instead of the process hanging.
  uint32_t spins = 0;
  if (nlook) {
    for (;;) {
      const uint32_t st = lane < nlook ? poll_flag(a.status + tile + 1u + lane) : 1u;
      if (__ballot(st == 0u) == 0) break;
      if (++spins > a.spin_limit) {
        if (lane == 0) __hip_atomic_store(a.watchdog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    seen_barrier();
    F val = F::zero();
    if (lane < nlook) val = desc_load<FID>(a.agg + 9 * ((size_t)tile + 1u + lane));
    const F term = val * fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblW + lane));
    C = wave_bcast0<FID>(wave_total<FID>(term));                                    // < 99 p
  }

  // ---- across groups: the carry at the group's end
  if (!last_group) {
    F GC = F::zero();
    F scale = F::one();  // u^(64 T * groups skipped)
    bool first_round = true;
    const uint64_t win_m = a.window >= 64u ? ~0ull : ((1ull << a.window) - 1ull);  // lanes that look
    for (uint32_t base = grp + 1u;; base += a.window) {
      const uint32_t gg = base + lane;
      uint32_t fi;
      for (;;) {
        const uint32_t st = gg < a.ngroups ? poll_flag(a.gstatus + gg) : 2u;
        const uint64_t inc_m = __ballot(st == 2u) & win_m, zero_m = __ballot(st == 0u) & win_m;
        fi = inc_m ? (uint32_t)__ffsll((long long)inc_m) - 1u : 64u;
        const uint64_t need = fi >= 63u ? ~0ull : ((2ull << fi) - 1ull);
        if ((zero_m & need) == 0) break;
        if (++spins > a.spin_limit) {
          if (lane == 0) __hip_atomic_store(a.watchdog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      seen_barrier();
      F val = F::zero();
      if (lane <= fi && lane < a.window && gg < a.ngroups) val = desc_load<FID>((lane < fi ? a.gagg : a.ginc) + 9 * (size_t)gg);
      const F term = val * fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblX + lane));   // u^(64 T lane) * value   < 1.03 p
      const F sum = wave_bcast0<FID>(wave_total<FID>(term));                     // < 66 p
      if (first_round) GC = sum;
      else GC = ((GC * F::one()) + (scale * sum)).norm();                        // < 1.77 p + 1.77 p
      if (fi < 64u) break;
      first_round = false;
      scale = (scale * fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblX + a.window))).canon();
    }
    // (only the last group can be short: the distance from this tile's end to the group's end is 63 - pos tiles)
    C = (C + fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblW + (kScanGroup - 1u - pos))) * GC).norm();   // < 100.6 p
    if (closer && lane == 0) {  // the group's inclusive value for the groups in front of it
      const F GI = (desc_load<FID>(a.gagg + 9 * (size_t)grp) + fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblX + 1u)) * GC).norm();  // < 3.6 p
      desc_store<FID>(a.ginc + 9 * (size_t)grp, GI);
      publish_flag(a.gstatus + grp, 2u);
    }
  }

  // ---- phase 3: walk the sub-tiles from the top, each from the value the one above it ended with
  F t = C;  // the suffix value at the end of the sub-tile in hand (wave-uniform)
#pragma unroll 1
  for (uint32_t jj = 0; jj < (uint32_t)J; jj++) {
    const uint32_t j = (uint32_t)J - 1u - jj;
    if (jj != 0) {
      t = wave_bcast0<FID>(t);  // lane 0 ended on the first coefficient of the sub-tile above
      load_sub(j);
    }
    const F TC = fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblS + 1u)) * t;  // u^512 * carry   < 1.8 p
    F S_next;  // S of lane + 1, zero for the last lane
#pragma unroll
    for (int i = 0; i < 9; i++) S_next.l[i] = lane < 63u ? park[(j * 9u + i) * 64u + lane + 1u] : 0u;
    t = ((S_next + TC).norm()) * (J > 1 ? Vlane : fp_ld_limbs<FID>(a.tbl + 9u * (kScanTblV + lane)));  // < 1.55 p
#pragma unroll
    for (uint32_t k = 8; k-- > 0;) {
      if constexpr (J > 1) {  // unpack coefficient k when its turn comes: unpacked ahead of the chain, all eight cost 72 more registers
#pragma unroll
        for (int q = 0; q < 8; q++) asm volatile("" : "+v"(w[k][q]) : "v"(t.l[0]));
      }
      t = (F::from_words(w[k]) + a.u * t).norm();  // < 2.02 p (coefficients below p: checked in phase 1)
      t.canon4().to_words(w[k]);
    }
    store_sub(j);
  }
#endif
}

// ---- host side ---------------------------------------------------------------------------------------
// launch one functor over n lanes; with profiling on, bracket it with hipEvents on the context's stream
struct VecIO;

// a challenge given in the ABI form -> its internal residue (value * 2^261 mod p)
template <int FID> static Fp<FID> challenge(const void* r, bool mont) {
  uint32_t w[8];
  memcpy(w, r, 32);
  require(Fp<FID>::words_lt_p(w), NMX_E_SCALAR_RANGE, "challenge >= field modulus");
  Fp<FID> f = Fp<FID>::from_words(w);
  return (mont ? f.mont256_to_internal() : f.to_internal()).canon();
}

struct VecIO {  // stages host vectors through the context arena; device vectors are used in place
  Ctx& c;
  bool dev;
  bool async = false;  // NMX_ASYNC on device-resident operands: the call returns once its kernel is enqueued (runtime.hpp async_mark)
  size_t n;
  size_t used = 0;
  std::vector<std::pair<void*, const void*>> outs;  // (device, host)
  VecIO(Ctx& ctx, uint32_t flags, size_t n_, int n_vecs)
      : c(ctx), dev((flags & NMX_SCALARS_DEVICE) != 0), async((flags & NMX_ASYNC) && (flags & NMX_SCALARS_DEVICE)), n(n_) {
    if (!dev) arena_reserve(c, (size_t)n_vecs * ((n * 32 + 255) & ~(size_t)255) + 256);
  }
  const uint32_t* in(const void* p, size_t elems) {
    if (dev) return (const uint32_t*)p;
    char* d = c.arena + used;
    used += (elems * 32 + 255) & ~(size_t)255;
    HIPCHK(hipMemcpyAsync(d, p, elems * 32, hipMemcpyHostToDevice, c.stream));
    return (const uint32_t*)d;
  }
  uint32_t* out(void* p, size_t elems) {
    if (dev) return (uint32_t*)p;
    char* d = c.arena + used;
    used += (elems * 32 + 255) & ~(size_t)255;
    outs.push_back({d, p});
    out_elems = elems;
    return (uint32_t*)d;
  }
  size_t out_elems = 0;
  void finish() {
    for (auto& o : outs)
      HIPCHK(hipMemcpyAsync((void*)o.second, o.first, out_elems * 32, hipMemcpyDeviceToHost, c.stream));
    if (async && outs.empty()) {
      async_mark(c);  // no wait: the next call of this host thread is ordered behind this one
      return;
    }
    stream_wait(c.stream);
  }
};

template <class Fn> static void timed_launch(Ctx& c, const Fn& f, size_t n, VecIO* io) {
  const bool prof = G.profiling;
  DeviceBackend be(c, false, prof);
  be.mark("kernel");
  be.launch(f, (uint32_t)n);
  be.mark("end");
  io->finish();
  if (prof && be.nmarks == 2 && !(io->async && io->outs.empty())) {  // (an asynchronous call has no kernel time to report yet)
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    prof_store(&ms, 1);
  }
}

template <int FID> struct FieldImpl {
  using F = Fp<FID>;
  static void axpy(Ctx& c, const void* a, const void* b, const void* r, size_t n, uint32_t flags, void* out) {
    const bool mont = flags & NMX_SCALARS_MONT;
    VecIO io(c, flags, n, 3);
    AxpyFn<FID> f{io.in(a, n), io.in(b, n), io.out(out, n), challenge<FID>(r, mont)};
    timed_launch(c, f, n, &io);
  }
  static void axpy2(Ctx& c, const void* a, const void* b, const void* cc, const void* r, size_t n, uint32_t flags,
                    void* out) {
    const bool mont = flags & NMX_SCALARS_MONT;
    VecIO io(c, flags, n, 4);
    F ri = challenge<FID>(r, mont);
    Axpy2Fn<FID> f{io.in(a, n), io.in(b, n), io.in(cc, n), io.out(out, n), ri, (ri * ri).canon()};
    timed_launch(c, f, n, &io);
  }
  static void cross_term(Ctx& c, const void* az, const void* bz, const void* cz, const void* e, const void* u, size_t n,
                         uint32_t flags, void* out) {
    const bool mont = flags & NMX_SCALARS_MONT;
    VecIO io(c, flags, n, 5);
    // canonical data: k = 2^522 (R2); Montgomery data (F = 2^256): k = 2^522 / 2^256 = 2^266 (C266)
    F k = mont ? F::from_limbs(FpParams<FID>::C266) : F::from_limbs(FpParams<FID>::R2);
    const F nu = F::sub2(F::zero(), challenge<FID>(u, mont)).norm().canon();  // p - u (0 for u = 0)
    CrossTermFn<FID> f{io.in(az, n), io.in(bz, n), io.in(cz, n), io.in(e, n), io.out(out, n), nu, k};
    timed_launch(c, f, n, &io);
  }
  static void cross_term2(Ctx& c, const void* az, const void* bz, const void* cz, const void* e1, const void* e2,
                          const void* u, size_t n, uint32_t flags, void* out) {
    const bool mont = flags & NMX_SCALARS_MONT;
    VecIO io(c, flags, n, 6);
    F k = mont ? F::from_limbs(FpParams<FID>::C266) : F::from_limbs(FpParams<FID>::R2);
    const F nu = F::sub2(F::zero(), challenge<FID>(u, mont)).norm().canon();
    CrossTerm2Fn<FID> f{io.in(az, n), io.in(bz, n), io.in(cz, n), io.in(e1, n), io.in(e2, n), io.out(out, n), nu, k};
    timed_launch(c, f, n, &io);
  }
  static void vec_add(Ctx& c, const void* a, const void* b, size_t n, uint32_t flags, void* out) {
    VecIO io(c, flags, n, 3);
    VecAddFn<FID> f{io.in(a, n), io.in(b, n), io.out(out, n)};
    timed_launch(c, f, n, &io);
  }
  // out[i] = z[lo_off + i*stride] + r * (z[hi_off + i*stride] - z[lo_off + i*stride]),  i < n_out
  static void bind(Ctx& c, const void* z, size_t z_len, size_t lo_off, size_t hi_off, size_t stride, const void* r,
                   size_t n_out, uint32_t flags, void* out) {
    const bool mont = flags & NMX_SCALARS_MONT;
    const bool dev = flags & NMX_SCALARS_DEVICE;
    VecIO io(c, flags, z_len + n_out, 2);
    const uint32_t* zd = io.in(z, z_len);
    uint32_t* od = (dev && out == z) ? (uint32_t*)out : io.out(out, n_out);
    BindTopFn<FID> f{zd + 8 * lo_off, zd + 8 * hi_off, od, challenge<FID>(r, mont), (uint32_t)stride};
    timed_launch(c, f, n_out, &io);
  }
};

// builds eq(r, .) over {0,1}^ell into the device buffer d_out (2^ell elements, the vectors' form)
template <int FID> static void eq_evals_t(Ctx& c, const void* r_host, uint32_t ell, uint32_t flags, uint32_t* d_out) {
  using F = Fp<FID>;
  const bool mont = flags & NMX_SCALARS_MONT;
  // evals[0] = ONE in the vectors' form: 1, or 2^256 mod p
  F one = F::zero();
  one.l[0] = 1;
  uint32_t w[8];
  if (mont) {
    // 2^256 mod p as a plain integer = the internal form (x * 2^261) of x = 1/32
    F two5 = F::zero();
    two5.l[0] = 32;
    F inv32 = two5.to_internal().canon().inv();  // (1/32) * 2^261 = 2^256 mod p
    inv32.canon().to_words(w);
  } else {
    one.to_words(w);
  }
  DeviceBackend be(c, false, false);
  if (ell <= EqDirectFn<FID>::kMaxEll) {
    EqDirectFn<FID> f;
    f.out = d_out;
    f.ell = ell;
    f.one = F::from_words(w);
    const F one_i = F::one();
    for (uint32_t i = 0; i < ell; i++) {
      f.r[i] = challenge<FID>((const uint8_t*)r_host + 32 * i, mont);
      f.nr[i] = F::sub2(one_i, f.r[i]).norm().canon();
    }
    for (uint32_t i = ell; i < EqDirectFn<FID>::kMaxEll; i++) f.r[i] = f.nr[i] = F::zero();
    be.launch(f, 1u << ell);
    return;
  }
  if (ell <= 2 * EqDirectFn<FID>::kMaxEll) {  // two sqrt-size tables + one product per entry (EqSplit2Fn / EqProductFn)
    const uint32_t ellR = ell / 2, ellL = ell - ellR;
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    arena_reserve(c, pad(((size_t)1 << ellL) * 32) + pad(((size_t)1 << ellR) * 32) + 256);
    EqSplit2Fn<FID> f;
    f.outL = (uint32_t*)c.arena, f.outR = (uint32_t*)(c.arena + pad(((size_t)1 << ellL) * 32));
    f.ellL = ellL, f.ellR = ellR, f.oneL = F::one(), f.oneR = F::from_words(w);
    const F one_i = F::one();
    for (uint32_t i = 0; i < 2 * EqDirectFn<FID>::kMaxEll; i++) f.r[i] = f.nr[i] = F::zero();
    for (uint32_t i = 0; i < ell; i++) {
      f.r[i] = challenge<FID>((const uint8_t*)r_host + 32 * (size_t)i, mont);
      f.nr[i] = F::sub2(one_i, f.r[i]).norm().canon();
    }
    be.launch(f, (1u << ellL) + (1u << ellR));
    EqProductFn<FID> g{f.outL, f.outR, d_out, ellR, (1u << ellR) - 1u};
    be.launch(g, 1u << ell);
    return;
  }
  HIPCHK(hipMemcpyAsync(d_out, w, 32, hipMemcpyHostToDevice, c.stream));
  stream_wait(c.stream);  // w is a stack buffer
  uint32_t size = 1;
  for (int j = (int)ell - 1; j >= 0; j--) {  // for r in r.iter().rev()
    EqStepFn<FID> f{d_out, challenge<FID>((const uint8_t*)r_host + 32 * j, mont), size};
    be.launch(f, size);
    size *= 2;
  }
}

template <int FID> static void eq_evals_pair_t(Ctx& c, const void* r_host, uint32_t ellL, uint32_t ellR, uint32_t flags, uint32_t* d_outL,
                                               uint32_t* d_outR) {
  using F = Fp<FID>;
  if (ellL > EqDirectFn<FID>::kMaxEll || ellR > EqDirectFn<FID>::kMaxEll) {
    eq_evals_t<FID>(c, r_host, ellL, flags, d_outL);
    eq_evals_t<FID>(c, (const uint8_t*)r_host + 32 * (size_t)ellL, ellR, flags, d_outR);
    return;
  }
  const bool mont = flags & NMX_SCALARS_MONT;
  EqDirect2Fn<FID> f;
  f.outL = d_outL, f.outR = d_outR, f.ellL = ellL, f.ellR = ellR;
  uint32_t w[8];
  if (mont) {  // ONE in the vectors' form, as in eq_evals_t
    F two5 = F::zero();
    two5.l[0] = 32;
    two5.to_internal().canon().inv().canon().to_words(w);
  } else {
    F one = F::zero();
    one.l[0] = 1;
    one.to_words(w);
  }
  f.one = F::from_words(w);
  const F one_i = F::one();
  for (uint32_t i = 0; i < 2 * EqDirectFn<FID>::kMaxEll; i++) f.r[i] = f.nr[i] = F::zero();
  for (uint32_t i = 0; i < ellL + ellR; i++) {
    f.r[i] = challenge<FID>((const uint8_t*)r_host + 32 * (size_t)i, mont);
    f.nr[i] = F::sub2(one_i, f.r[i]).norm().canon();
  }
  DeviceBackend be(c, false, false);
  be.launch(f, (1u << ellL) + (1u << ellR));
}

template <int FID> static void spmv_convert_t(Ctx& c, uint32_t* d_data, size_t nnz, uint32_t flags) {
  DeviceBackend be(c, false, false);
  struct Conv {
    uint32_t* v;
    uint32_t from_mont;
    NMX_HD void operator()(uint32_t i) const {
      Fp<FID> f = Fp<FID>::from_words(v + 8 * (size_t)i);
      (from_mont ? f.mont256_to_internal() : f.to_internal()).canon().to_words(v + 8 * (size_t)i);
    }
  };
  Conv f{d_data, (flags & NMX_SCALARS_MONT) ? 1u : 0u};
  be.launch(f, (uint32_t)nnz);
}
template <int FID> static void spmv_classify_t(Ctx& c, const uint32_t* d_data, uint32_t* d_indices, size_t nnz) {
  DeviceBackend be(c, false, false);
  SpmvClassifyFn<FID> f{d_data, d_indices};
  be.launch(f, (uint32_t)nnz);
}
template <int FID>
static void spmv_apply_t(Ctx& c, const uint32_t* indptr, const uint32_t* indices, const uint32_t* data, size_t rows,
                         size_t cols, const void* z, uint32_t flags, void* out) {
  VecIO io(c, flags, rows + cols, 2);
  const uint32_t* dz = io.in(z, cols);
  uint32_t* dout = io.out(out, rows);
  SpmvFn<FID> f{indptr, indices, data, dz, dout, cols <= ((size_t)1 << kSpmvColBits) ? (1u << kSpmvColBits) - 1u : 0xffffffffu};
  timed_launch(c, f, rows, &io);
}

// x: rows of M; out: cols of M.  tr_*: the virtual-row form of M^T (capi.hip builds it).
template <int FID>
static void spmv_apply_transposed_t(Ctx& c, const uint32_t* vptr, const uint32_t* indices, const uint32_t* data, const uint32_t* vout,
                                    const uint32_t* hrow, const uint32_t* hstart, size_t nvirt, size_t nheavy, size_t nparts, size_t rows,
                                    size_t cols, const void* x, uint32_t flags, void* out) {
  const bool dev = (flags & NMX_SCALARS_DEVICE) != 0, async = dev && (flags & NMX_ASYNC);
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t pb = pad((nparts ? nparts : 1) * 32);
  arena_reserve(c, pb + (dev ? 0 : pad(rows * 32) + pad(cols * 32)) + 256);
  uint32_t* partial = (uint32_t*)c.arena;
  const uint32_t* dx = (const uint32_t*)x;
  uint32_t* dout = (uint32_t*)out;
  if (!dev) {
    HIPCHK(hipMemcpyAsync(c.arena + pb, x, rows * 32, hipMemcpyHostToDevice, c.stream));
    dx = (const uint32_t*)(c.arena + pb);
    dout = (uint32_t*)(c.arena + pb + pad(rows * 32));
  }
  const bool prof = G.profiling;
  DeviceBackend be(c, false, prof);
  be.mark("kernel");
  SpmvSegFn<FID> f{vptr, indices, data, dx, vout, dout, partial, rows <= ((size_t)1 << kSpmvColBits) ? (1u << kSpmvColBits) - 1u : 0xffffffffu};
  be.launch(f, (uint32_t)nvirt);
  if (nheavy) {
    hipLaunchKernelGGL((k_spmv_heavy<FID>), dim3((uint32_t)nheavy), dim3(256), 0, c.stream, hrow, hstart, (const uint32_t*)partial, dout);
    HIPCHK(hipGetLastError());
  }
  be.mark("end");
  if (!dev) HIPCHK(hipMemcpyAsync(out, dout, cols * 32, hipMemcpyDeviceToHost, c.stream));
  if (async) {
    async_mark(c);
    return;
  }
  stream_wait(c.stream);
  if (prof && be.nmarks == 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    prof_store(&ms, 1);
  }
}

template <int FID>
static void spmv_apply_pair_t(Ctx& c, const uint32_t* indptr, const uint32_t* indices, const uint32_t* data, size_t rows,
                              size_t cols, const void* z1, const void* z2, uint32_t flags, void* out1, void* out2) {
  VecIO io(c, flags, rows + cols, 4);
  const uint32_t* d1 = io.in(z1, cols);
  const uint32_t* d2 = io.in(z2, cols);
  uint32_t* o1 = io.out(out1, rows);
  uint32_t* o2 = io.out(out2, rows);
  SpmvPairFn<FID> f{indptr, indices, data, d1, d2, o1, o2, cols <= ((size_t)1 << kSpmvColBits) ? (1u << kSpmvColBits) - 1u : 0xffffffffu};
  timed_launch(c, f, rows, &io);
}

// commit_T's chain (src/r1cs/mod.rs:590-620) on HBM-resident vectors: Z = z1 + z2 (z2 may be null: Z = z1), then SpmvCrossFn.
// Z lives in the context arena; with NMX_ASYNC nothing waits (the arena is only re-carved by later calls on the same stream).
template <int FID>
static void r1cs_cross_term_t(Ctx& c, const uint32_t* const* ip, const uint32_t* const* ix, const uint32_t* const* dt, size_t rows,
                              size_t cols, const void* z1, const void* z2, const void* e, const void* u, uint32_t flags, void* out) {
  using F = Fp<FID>;
  const bool mont = flags & NMX_SCALARS_MONT;
  VecIO io(c, flags, rows, 0);
  const uint32_t* dz = (const uint32_t*)z1;
  DeviceBackend be(c, false, false);
  if (z2) {
    arena_reserve(c, cols * 32 + 256);
    uint32_t* zsum = (uint32_t*)c.arena;
    VecAddFn<FID> add{(const uint32_t*)z1, (const uint32_t*)z2, zsum};
    be.launch(add, (uint32_t)cols);
    dz = zsum;
  }
  const F k = mont ? F::from_limbs(FpParams<FID>::C266) : F::from_limbs(FpParams<FID>::R2);
  const F nu = F::sub2(F::zero(), challenge<FID>(u, mont)).norm().canon();
  SpmvCrossFn<FID> f{ip[0], ix[0], dt[0], ip[1], ix[1], dt[1], ip[2], ix[2], dt[2], dz, (const uint32_t*)e, (uint32_t*)out,
                     cols <= ((size_t)1 << kSpmvColBits) ? (1u << kSpmvColBits) - 1u : 0xffffffffu, nu, k};
  timed_launch(c, f, rows, &io);
}
template <int FID>
static void nifs_fold_t(Ctx& c, const void* w1, const void* w2, size_t n_w, const void* e1, const void* t, size_t n_e, const void* r,
                        uint32_t flags, void* w, void* e) {
  VecIO io(c, flags, n_w + n_e, 0);
  FoldPairFn<FID> f{(const uint32_t*)w1, (const uint32_t*)w2, (const uint32_t*)e1, (const uint32_t*)t, (uint32_t*)w, (uint32_t*)e,
                    (uint32_t)n_w, challenge<FID>(r, flags & NMX_SCALARS_MONT)};
  timed_launch(c, f, n_w + n_e, &io);
}

// returns false when an element is zero (the reference: Err(NovaError::InternalError), spartan/mod.rs:103-105)
template <int FID> static bool batch_invert_t(Ctx& c, const void* v, size_t n, uint32_t flags, void* out) {
  using F = Fp<FID>;
  using H = HostFp4<FID>;
  const bool mont = flags & NMX_SCALARS_MONT, dev = flags & NMX_SCALARS_DEVICE;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  // level sizes: n, ceil(n / K(n)), ... down to at most kBinvHostBelow
  std::vector<size_t> sz{n};
  while (sz.back() > kBinvHostBelow) sz.push_back((sz.back() + binv_chunk(sz.back()) - 1) / binv_chunk(sz.back()));
  const size_t L = sz.size() - 1;  // device levels 0 .. L - 1; level L on the host
  size_t need = 256;
  if (!dev) need += 2 * pad(n * 32);
  for (size_t l = 1; l <= L; l++) need += 2 * pad(sz[l] * 32);  // products (= the level's input) and prefix / inverse arrays
  arena_reserve(c, need);
  size_t used = 0;
  auto take = [&](size_t bytes) {
    char* d = c.arena + used;
    used += pad(bytes);
    return (uint32_t*)d;
  };
  std::vector<const uint32_t*> in(L + 1);
  std::vector<uint32_t*> res(L + 1);
  if (dev) {
    in[0] = (const uint32_t*)v, res[0] = (uint32_t*)out;
  } else {
    uint32_t* d = take(n * 32);
    HIPCHK(hipMemcpyAsync(d, v, n * 32, hipMemcpyHostToDevice, c.stream));
    in[0] = d, res[0] = take(n * 32);
  }
  for (size_t l = 1; l <= L; l++) {
    uint32_t* prod = take(sz[l] * 32);
    in[l] = prod, res[l] = take(sz[l] * 32);
  }
  const bool prof = G.profiling;
  DeviceBackend be(c, false, prof);
  be.mark("kernel");
  for (size_t l = 0; l < L; l++) {
    BatchInvFwdFn<FID> f{in[l], res[l], (uint32_t*)in[l + 1], (uint32_t)sz[l], (uint32_t)sz[l + 1], binv_chunk(sz[l])};
    be.launch(f, (uint32_t)sz[l + 1]);
  }
  // top level on the host: out = Fm^2 / Y for every word Y (Fm = 1 or 2^256: the words' form; see the kernels' header)
  const size_t nt = sz[L];
  if (!c.pinned) HIPCHK(hipHostMalloc((void**)&c.pinned, DeviceBackend::kPinnedBytes, hipHostMallocDefault));
  static_assert(2 * kBinvHostBelow * 32 <= DeviceBackend::kPinnedBytes, "the host level lives in the context's pinned buffer");
  uint32_t *top = (uint32_t*)c.pinned, *topinv = top + 8 * kBinvHostBelow;
  HIPCHK(hipMemcpyAsync(top, in[L], nt * 32, hipMemcpyDeviceToHost, c.stream));
  stream_wait(c.stream);
  {
    std::vector<H> e(nt), pre(nt);
    H acc = H::one();
    for (size_t i = 0; i < nt; i++) {
      if (!F::words_lt_p(top + 8 * i)) throw Fail{NMX_E_SCALAR_RANGE, "batch_invert: element >= field modulus"};
      e[i] = H::from_canonical(top + 8 * i);  // the word as a plain value
      pre[i] = acc;
      acc = acc * e[i];
    }
    if (acc.is_zero()) return false;
    H inv = acc.inv() * (mont ? H::pow2(512) : H::one());
    for (size_t i = nt; i-- > 0;) {
      (inv * pre[i]).to_canonical(topinv + 8 * i);
      inv = inv * e[i];
    }
  }
  HIPCHK(hipMemcpyAsync(res[L], topinv, nt * 32, hipMemcpyHostToDevice, c.stream));
  for (size_t l = L; l-- > 0;) {
    BatchInvBwdFn<FID> f{in[l], res[l + 1], res[l], (uint32_t)sz[l], (uint32_t)sz[l + 1], binv_chunk(sz[l])};
    be.launch(f, (uint32_t)sz[l + 1]);
  }
  be.mark("end");
  if (!dev) HIPCHK(hipMemcpyAsync(out, res[0], n * 32, hipMemcpyDeviceToHost, c.stream));
  stream_wait(c.stream);
  if (prof && be.nmarks == 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    prof_store(&ms, 1);
  }
  return true;
}

template <int FID>
static void lincomb_t(Ctx& c, const void* const* vecs, const size_t* lens, size_t k, const void* s, size_t n_out,
                      uint32_t flags, void* out) {
  using F = Fp<FID>;
  const bool mont = flags & NMX_SCALARS_MONT, dev = flags & NMX_SCALARS_DEVICE;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t need = pad(k * 8) * 2 + pad(k * 32) + 256;
  if (!dev) {
    for (size_t j = 0; j < k; j++) need += pad(lens[j] * 32);
    need += pad(n_out * 32);
  }
  arena_reserve(c, need);
  size_t used = 0;
  auto take = [&](size_t bytes) {
    char* d = c.arena + used;
    used += pad(bytes);
    return d;
  };
  std::vector<uint64_t> addr(k), ln(k);
  std::vector<uint32_t> w(8 * k);
  // powers::<E>(&s, k) = [1, s, s^2, ...] (spartan/mod.rs:130-137), in internal form
  const F si = challenge<FID>(s, mont);
  F pw = F::one();
  for (size_t j = 0; j < k; j++) {
    pw.canon().to_words(w.data() + 8 * j);
    pw = (pw * si).canon();
    ln[j] = lens[j];
    if (dev) {
      addr[j] = (uint64_t)(uintptr_t)vecs[j];
    } else {
      char* d = take(lens[j] * 32);
      if (lens[j]) HIPCHK(hipMemcpyAsync(d, vecs[j], lens[j] * 32, hipMemcpyHostToDevice, c.stream));
      addr[j] = (uint64_t)(uintptr_t)d;
    }
  }
  uint64_t* d_addr = (uint64_t*)take(k * 8);
  uint64_t* d_len = (uint64_t*)take(k * 8);
  uint32_t* d_w = (uint32_t*)take(k * 32);
  HIPCHK(hipMemcpyAsync(d_addr, addr.data(), k * 8, hipMemcpyHostToDevice, c.stream));
  HIPCHK(hipMemcpyAsync(d_len, ln.data(), k * 8, hipMemcpyHostToDevice, c.stream));
  HIPCHK(hipMemcpyAsync(d_w, w.data(), k * 32, hipMemcpyHostToDevice, c.stream));
  uint32_t* d_out = dev ? (uint32_t*)out : (uint32_t*)take(n_out * 32);
  LinCombFn<FID> f{d_addr, d_len, d_w, d_out, (uint32_t)k};
  const bool prof = G.profiling;
  DeviceBackend be(c, false, prof);
  be.mark("kernel");
  be.launch(f, (uint32_t)n_out);
  be.mark("end");
  if (!dev) HIPCHK(hipMemcpyAsync(out, d_out, n_out * 32, hipMemcpyDeviceToHost, c.stream));
  stream_wait(c.stream);  // also: addr / ln / w are stack-owned
  if (prof && be.nmarks == 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    prof_store(&ms, 1);
  }
}

// Workspace of one suffix-Horner call, carved from the context arena (no allocation on the call path once the arena
// has grown): per recursion level the chunk heads and carries, plus one 65-entry power table.
static constexpr size_t kHornerTopMin = 1u << 15;  // shorter inputs are launch-latency-bound either way
struct HornerArena {
  char* base;
  size_t used = 0;
  static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
  static size_t need(size_t n, bool top) {
    size_t total = 0;
    while (top && n >= kHornerTopMin) {  // heads + carries of every register-resident level (sized for the smaller chunk)
      const size_t nc = (n + 3) / 4;
      total += 2 * pad(nc * 32);
      n = nc;
    }
    for (size_t m = n; m > 1;) {
      const size_t nc = (m + kHornerChunk - 1) / kHornerChunk;
      total += 2 * pad(nc * 32) + pad((kHornerChunk + 1) * 32);
      m = nc;
    }
    return total + pad(32) + 256;
  }
  void* take(size_t bytes) {
    void* p = base + used;
    used += pad(bytes);
    return p;
  }
};

// out (device, n elements) <- suffix Horner of f (device); level `lvl` of the recursion evaluates at u^(16^lvl), whose
// power table pw_all[lvl] (u_l^0 .. u_l^16, internal form) was uploaded before the first launch: no host round trip
// between the levels.
template <int FID> static void horner_dev(Ctx& c, const uint32_t* f, uint32_t n, const std::vector<Fp<FID>>& u_lvl,
                                          const uint32_t* pw_all, uint32_t lvl, uint32_t* out, HornerArena& ws) {
  DeviceBackend be(c, false, false);
  const uint32_t nc = (n + kHornerChunk - 1) / kHornerChunk;
  uint32_t* heads = (uint32_t*)ws.take((size_t)nc * 32);
  HornerLocalFn<FID> lf{f, out, heads, u_lvl[lvl], n};
  be.launch(lf, nc);
  if (nc == 1) return;
  uint32_t* carries = (uint32_t*)ws.take((size_t)nc * 32);
  horner_dev<FID>(c, heads, nc, u_lvl, pw_all, lvl + 1, carries, ws);
  HornerFixFn<FID> ff{out, carries, pw_all + (size_t)lvl * (kHornerChunk + 1) * 8, n};
  be.launch(ff, n);
}
// the single-pass kernel (k_horner_scan): per-lane constants + tile states from one small launch, then the scan itself
// -> false if a wave of the scan gave up waiting (never observed; the caller then runs the two-pass kernels)
static constexpr uint32_t kScanSpinLimit = 1u << 22;  // polls of >= 0.5 us each: seconds, against waits of microseconds
template <int FID> static bool horner_scan_t(Ctx& c, const void* f, size_t n, const Fp<FID>& u0, const HostFp4<FID>& hu, bool dev, void* out) {
  using F = Fp<FID>;
  using H = HostFp4<FID>;
  // sub-tiles per wave: the look-backs are paid once per tile, so long inputs take 2 (from 2^21 coefficients) or 4 (from 2^22);
  // short ones keep 1: more tiles in flight (profiles/r03_fieldvec/horner_scan.txt)
  uint32_t J = G.horner_sub;
  if (J != 1 && J != 2 && J != 4) J = n >= (1u << 22) ? 4u : (n >= (1u << 21) ? 2u : 1u);
  const size_t tile = (size_t)J * kScanSub;
  const uint32_t nt = (uint32_t)((n + tile - 1) / tile), ng = (nt + kScanGroup - 1) / kScanGroup;
  // the scan's five constants: ~17 squarings and one inversion, in 4 x 64-bit host arithmetic (HostFp4; the portable build of the
  // device form took ~10 us per call here -- three calls per HyperKZG prove)
  H h8 = hu;
  for (int i = 0; i < 3; i++) h8 = h8 * h8;
  H hS = h8;
  for (int i = 0; i < 6; i++) hS = hS * hS;
  H hT = hS;
  for (uint32_t q = J; q > 1; q >>= 1) hT = hT * hT;
  H hG = hT;
  for (int i = 0; i < 6; i++) hG = hG * hG;
  const F u8 = h8.to_device(), uS = hS.to_device(), uT = hT.to_device(), uG = hG.to_device();
  const F v8 = h8.inv().to_device();  // u != 0 (checked by the caller)
  const size_t nflags = (size_t)nt + 2 * (size_t)ng + 1;  // tile states, group tickets, group states, the start-order counter
  arena_reserve(c, HornerArena::pad(kScanTblN * 36) + HornerArena::pad(nflags * 4) + HornerArena::pad((size_t)nt * 36) +
                       2 * HornerArena::pad((size_t)ng * 36) + (dev ? 0 : 2 * HornerArena::pad(n * 32)) + 256);
  HornerArena ws{c.arena};
  uint32_t* tbl = (uint32_t*)ws.take(kScanTblN * 36);
  uint32_t* flags = (uint32_t*)ws.take(nflags * 4);
  uint32_t* agg = (uint32_t*)ws.take((size_t)nt * 36);
  uint32_t* gagg = (uint32_t*)ws.take((size_t)ng * 36);
  uint32_t* ginc = (uint32_t*)ws.take((size_t)ng * 36);
  const uint32_t* df = (const uint32_t*)f;
  uint32_t* dout = (uint32_t*)out;
  if (!dev) {
    void* a = ws.take(n * 32);
    dout = (uint32_t*)ws.take(n * 32);
    HIPCHK(hipMemcpyAsync(a, f, n * 32, hipMemcpyHostToDevice, c.stream));
    df = (const uint32_t*)a;
  }
  const bool prof = G.profiling;
  DeviceBackend be(c, false, prof);
  be.mark("kernel");
  HornerTblArgs<FID> ta{tbl, flags, u0, u8, v8, uS, uT, uG, (uint32_t)nflags};
  be.launch_kernel(k_horner_tables<FID>, (uint32_t)((nflags + 255) / 256), 256, ta);
  const uint32_t win = G.horner_window;
  if (!c.pinned) HIPCHK(hipHostMalloc((void**)&c.pinned, DeviceBackend::kPinnedBytes, hipHostMallocDefault));
  volatile uint32_t* wd = (volatile uint32_t*)(c.pinned + DeviceBackend::kPinnedBytes - 64);  // past every landing zone
  wd[0] = 0, wd[1] = 0;
  const uint32_t limit = G.horner_spin_limit ? (uint32_t)G.horner_spin_limit : kScanSpinLimit;
  HornerScanArgs<FID> sa{df, dout, tbl, flags, flags + nt, flags + nt + ng, agg, gagg, ginc, u0, (uint32_t)n, nt, ng,
                         win >= 1 && win <= 64 ? win : 64u, (uint32_t*)wd, limit, flags + nt + 2 * (size_t)ng,
                         (uint32_t)G.horner_order.load(std::memory_order_relaxed)};
  if (J == 1) be.launch_kernel(k_horner_scan<FID, 1>, (nt + 3) / 4, 256, sa);
  else if (J == 2) be.launch_kernel(k_horner_scan<FID, 2>, (nt + 3) / 4, 256, sa);
  else be.launch_kernel(k_horner_scan<FID, 4>, (nt + 3) / 4, 256, sa);
  be.mark("end");
  if (!dev) HIPCHK(hipMemcpyAsync(out, dout, n * 32, hipMemcpyDeviceToHost, c.stream));
  stream_wait(c.stream);
  if (wd[0] | wd[1]) {  // [0]: a wave gave up waiting; [1]: a coefficient word >= p -- either way the two-pass kernels take the call
    if (wd[0]) note_scan_timeout();
    return false;
  }
  if (prof && be.nmarks == 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    prof_store(&ms, 1);
  }
  return true;
}

template <int FID>
static void horner_t(Ctx& c, const void* f, size_t n, const void* u, uint32_t flags, void* out) {
  using F = Fp<FID>;
  const bool dev = flags & NMX_SCALARS_DEVICE;
  if (n >= kScanMin && n < (1ull << 32) && G.horner_top == 0) {
    const F us = challenge<FID>(u, flags & NMX_SCALARS_MONT);
    if (!us.is_zero_limbs()) {  // u = 0: out = f, left to the chunk kernels below
      const HostFp4<FID> hu = (flags & NMX_SCALARS_MONT) ? HostFp4<FID>::from_mont256(u) : HostFp4<FID>::from_canonical(u);
      if (horner_scan_t<FID>(c, f, n, us, hu, dev, out)) return;
    }
  }

  // per level: u_l = u^(16^l) and its powers 0..16 -- 16 host multiplications per level, at most 8 levels
  std::vector<F> u_lvl;
  std::vector<uint32_t> pwh;
  const F u0 = challenge<FID>(u, flags & NMX_SCALARS_MONT);
  const uint32_t K = G.horner_top == 4 ? 4u : 8u;
  // register-resident levels: level l works on n_top[l] elements at u_top[l] = u^(K^l); the chunk-per-lane recursion takes
  // over at the first level shorter than 2^15
  std::vector<F> u_top;
  std::vector<size_t> n_top;
  size_t n_rec = n;
  {
    F ul = u0;
    while (n_rec >= kHornerTopMin && G.horner_top != 1) {
      u_top.push_back(ul);
      n_top.push_back(n_rec);
      for (uint32_t q = K; q > 1; q >>= 1) ul = ul.sqr().canon();
      n_rec = (n_rec + K - 1) / K;
    }
    for (size_t m = n_rec;; m = (m + kHornerChunk - 1) / kHornerChunk) {
      u_lvl.push_back(ul);
      F p = F::one();
      for (uint32_t k = 0; k <= kHornerChunk; k++) {
        uint32_t w[8];
        p.canon().to_words(w);
        pwh.insert(pwh.end(), w, w + 8);
        if (k < kHornerChunk) p = (p * ul).canon();
      }
      ul = p;  // u_l^16
      if (m <= kHornerChunk) break;
    }
  }
  arena_reserve(c, HornerArena::need(n, G.horner_top != 1) + HornerArena::pad(pwh.size() * 4) + (dev ? 0 : 2 * HornerArena::pad(n * 32)));
  HornerArena ws{c.arena};
  uint32_t* d_pw = (uint32_t*)ws.take(pwh.size() * 4);
  HIPCHK(hipMemcpyAsync(d_pw, pwh.data(), pwh.size() * 4, hipMemcpyHostToDevice, c.stream));  // pwh lives to the sync below
  const uint32_t* df = (const uint32_t*)f;
  uint32_t* dout = (uint32_t*)out;
  if (!dev) {
    void* a = ws.take(n * 32);
    dout = (uint32_t*)ws.take(n * 32);
    HIPCHK(hipMemcpyAsync(a, f, n * 32, hipMemcpyHostToDevice, c.stream));
    df = (const uint32_t*)a;
  }
  const bool prof = G.profiling;
  DeviceBackend be(c, false, prof);
  be.mark("kernel");
  {
    // down: heads of every register-resident level; bottom: the chunk-per-lane recursion; up: re-walk from the carries
    const size_t L = n_top.size();
    std::vector<const uint32_t*> src(L + 1);
    std::vector<uint32_t*> carr(L + 1);
    src[0] = df;
    carr[0] = dout;
    for (size_t l = 0; l < L; l++) {
      const uint32_t nl = (uint32_t)n_top[l], nc = (nl + K - 1) / K;
      uint32_t* heads = (uint32_t*)ws.take((size_t)nc * 32);
      carr[l + 1] = (uint32_t*)ws.take((size_t)nc * 32);
      src[l + 1] = heads;
      if (K == 4) {
        HornerHeadFn<FID, 4> hf{src[l], heads, u_top[l], nl};
        be.launch(hf, nc);
      } else {
        HornerHeadFn<FID, 8> hf{src[l], heads, u_top[l], nl};
        be.launch(hf, nc);
      }
    }
    horner_dev<FID>(c, src[L], (uint32_t)n_rec, u_lvl, d_pw, 0, carr[L], ws);
    for (size_t l = L; l-- > 0;) {
      const uint32_t nl = (uint32_t)n_top[l], nc = (nl + K - 1) / K;
      if (K == 4) {
        HornerWalkFn<FID, 4> wf{src[l], carr[l + 1], carr[l], u_top[l], nl, nc};
        be.launch(wf, nc);
      } else {
        HornerWalkFn<FID, 8> wf{src[l], carr[l + 1], carr[l], u_top[l], nl, nc};
        be.launch(wf, nc);
      }
    }
  }
  be.mark("end");
  if (!dev) HIPCHK(hipMemcpyAsync(out, dout, n * 32, hipMemcpyDeviceToHost, c.stream));
  stream_wait(c.stream);
  if (prof && be.nmarks == 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    prof_store(&ms, 1);
  }
}
void fv_suffix_horner(Ctx& c, int field, const void* f, size_t n, const void* u, uint32_t flags, void* out) {
  switch (field) {
    case 0: horner_t<0>(c, f, n, u, flags, out); break;
    case 1: horner_t<1>(c, f, n, u, flags, out); break;
    case 2: horner_t<2>(c, f, n, u, flags, out); break;
    case 3: horner_t<3>(c, f, n, u, flags, out); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}

void fv_eq_evals(Ctx& c, int field, const void* r_host, uint32_t ell, uint32_t flags, uint32_t* d_out) {
  switch (field) {
    case 0: eq_evals_t<0>(c, r_host, ell, flags, d_out); break;
    case 1: eq_evals_t<1>(c, r_host, ell, flags, d_out); break;
    case 2: eq_evals_t<2>(c, r_host, ell, flags, d_out); break;
    case 3: eq_evals_t<3>(c, r_host, ell, flags, d_out); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}
void fv_eq_evals_pair(Ctx& c, int field, const void* r_host, uint32_t ellL, uint32_t ellR, uint32_t flags, uint32_t* d_outL,
                      uint32_t* d_outR) {
  switch (field) {
    case 0: eq_evals_pair_t<0>(c, r_host, ellL, ellR, flags, d_outL, d_outR); break;
    case 1: eq_evals_pair_t<1>(c, r_host, ellL, ellR, flags, d_outL, d_outR); break;
    case 2: eq_evals_pair_t<2>(c, r_host, ellL, ellR, flags, d_outL, d_outR); break;
    case 3: eq_evals_pair_t<3>(c, r_host, ellL, ellR, flags, d_outL, d_outR); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}
void fv_spmv_convert(Ctx& c, int field, uint32_t* d_data, size_t nnz, uint32_t flags) {
  switch (field) {
    case 0: spmv_convert_t<0>(c, d_data, nnz, flags); break;
    case 1: spmv_convert_t<1>(c, d_data, nnz, flags); break;
    case 2: spmv_convert_t<2>(c, d_data, nnz, flags); break;
    case 3: spmv_convert_t<3>(c, d_data, nnz, flags); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}
void fv_spmv_classify(Ctx& c, int field, const uint32_t* d_data, uint32_t* d_indices, size_t nnz, size_t cols) {
  if (cols > ((size_t)1 << kSpmvColBits)) return;  // no room for the class bits: every entry stays general
  switch (field) {
    case 0: spmv_classify_t<0>(c, d_data, d_indices, nnz); break;
    case 1: spmv_classify_t<1>(c, d_data, d_indices, nnz); break;
    case 2: spmv_classify_t<2>(c, d_data, d_indices, nnz); break;
    case 3: spmv_classify_t<3>(c, d_data, d_indices, nnz); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}
void fv_spmv_apply(Ctx& c, int field, const uint32_t* indptr, const uint32_t* indices, const uint32_t* data, size_t rows,
                   size_t cols, const void* z, uint32_t flags, void* out) {
  switch (field) {
    case 0: spmv_apply_t<0>(c, indptr, indices, data, rows, cols, z, flags, out); break;
    case 1: spmv_apply_t<1>(c, indptr, indices, data, rows, cols, z, flags, out); break;
    case 2: spmv_apply_t<2>(c, indptr, indices, data, rows, cols, z, flags, out); break;
    case 3: spmv_apply_t<3>(c, indptr, indices, data, rows, cols, z, flags, out); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}

void fv_spmv_apply_transposed(Ctx& c, int field, const uint32_t* vptr, const uint32_t* indices, const uint32_t* data, const uint32_t* vout,
                              const uint32_t* hrow, const uint32_t* hstart, size_t nvirt, size_t nheavy, size_t nparts, size_t rows, size_t cols,
                              const void* x, uint32_t flags, void* out) {
  switch (field) {
    case 0: spmv_apply_transposed_t<0>(c, vptr, indices, data, vout, hrow, hstart, nvirt, nheavy, nparts, rows, cols, x, flags, out); break;
    case 1: spmv_apply_transposed_t<1>(c, vptr, indices, data, vout, hrow, hstart, nvirt, nheavy, nparts, rows, cols, x, flags, out); break;
    case 2: spmv_apply_transposed_t<2>(c, vptr, indices, data, vout, hrow, hstart, nvirt, nheavy, nparts, rows, cols, x, flags, out); break;
    case 3: spmv_apply_transposed_t<3>(c, vptr, indices, data, vout, hrow, hstart, nvirt, nheavy, nparts, rows, cols, x, flags, out); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}

// R1CSShape::multiply_vec (src/r1cs/mod.rs:407-471: A z, B z, C z side by side under rayon::join) and compute_eval_table_sparse
// (src/spartan/mod.rs:497-533: the three transposed products, rayon::join again) as ONE call over HBM-resident vectors: matrix 0 on
// the context's stream, the others on its side streams -- these products are gathers of 32-byte elements, bound by latency (a 2^20-row
// transposed product: 80 us for 143 MB), so three of them in flight together take little longer than one.  The side streams start
// behind everything enqueued on the context's stream so far and the context's stream continues behind all of them: to the caller
// the call is stream-ordered like any other (NMX_ASYNC allowed).
template <int FID> static void spmv_many_t(Ctx& c, const SpmvManyItem* it, size_t k, bool transposed, const void* x, uint32_t flags) {
  const bool async = (flags & NMX_ASYNC) != 0;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t need = 256;
  std::vector<size_t> off(k, 0);
  if (transposed)
    for (size_t i = 0; i < k; i++) {
      off[i] = need;
      need += pad((it[i].nparts ? it[i].nparts : 1) * 32);
    }
  arena_reserve(c, need);
  require(k >= 1 && k - 1 <= (size_t)Ctx::kSideStreams, NMX_E_ARG, "too many matrices in one call");
  if (!c.side_ev) HIPCHK(hipEventCreateWithFlags(&c.side_ev, hipEventDisableTiming));
  if (k > 1) HIPCHK(hipEventRecord(c.side_ev, c.stream));
  std::vector<hipEvent_t> done(k, nullptr);
  try {
    for (size_t i = 0; i < k; i++) {
      hipStream_t st = c.stream;
      if (i > 0) {
        if (!c.side[i - 1]) HIPCHK(hipStreamCreateWithFlags(&c.side[i - 1], hipStreamNonBlocking));
        st = c.side[i - 1];
        HIPCHK(hipStreamWaitEvent(st, c.side_ev, 0));
      }
      const SpmvManyItem& m = it[i];
      if (transposed) {
        uint32_t* partial = (uint32_t*)(c.arena + off[i]);
        SpmvSegFn<FID> f{m.vptr, m.tix, m.tdata, (const uint32_t*)x, m.vout, (uint32_t*)m.out, partial,
                         m.rows <= ((size_t)1 << kSpmvColBits) ? (1u << kSpmvColBits) - 1u : 0xffffffffu};
        if (m.nvirt) hipLaunchKernelGGL((k_launch<SpmvSegFn<FID>>), dim3((uint32_t)((m.nvirt + 255) / 256)), dim3(256), 0, st, f, (uint32_t)m.nvirt);
        if (m.nheavy) hipLaunchKernelGGL((k_spmv_heavy<FID>), dim3((uint32_t)m.nheavy), dim3(256), 0, st, m.hrow, m.hstart, (const uint32_t*)partial, (uint32_t*)m.out);
      } else {
        SpmvFn<FID> f{m.indptr, m.indices, m.data, (const uint32_t*)x, (uint32_t*)m.out,
                      m.cols <= ((size_t)1 << kSpmvColBits) ? (1u << kSpmvColBits) - 1u : 0xffffffffu};
        if (m.rows) hipLaunchKernelGGL((k_launch<SpmvFn<FID>>), dim3((uint32_t)((m.rows + 255) / 256)), dim3(256), 0, st, f, (uint32_t)m.rows);
      }
      HIPCHK(hipGetLastError());
      if (i > 0) {
        HIPCHK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        HIPCHK(hipEventRecord(done[i], st));
        HIPCHK(hipStreamWaitEvent(c.stream, done[i], 0));
      }
    }
  } catch (...) {
    for (size_t i = 1; i < k; i++)
      if (c.side[i - 1]) (void)hipStreamSynchronize(c.side[i - 1]);
    (void)hipStreamSynchronize(c.stream);
    for (hipEvent_t e : done)
      if (e) (void)hipEventDestroy(e);
    throw;
  }
  for (hipEvent_t e : done)
    if (e) (void)hipEventDestroy(e);  // (a recorded event may be destroyed: the waits already enqueued keep what they need)
  if (async) {
    async_mark(c);
    return;
  }
  stream_wait(c.stream);
}
void fv_spmv_many(Ctx& c, int field, const SpmvManyItem* items, size_t k, bool transposed, const void* x, uint32_t flags) {
  switch (field) {
    case 0: spmv_many_t<0>(c, items, k, transposed, x, flags); break;
    case 1: spmv_many_t<1>(c, items, k, transposed, x, flags); break;
    case 2: spmv_many_t<2>(c, items, k, transposed, x, flags); break;
    case 3: spmv_many_t<3>(c, items, k, transposed, x, flags); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}

void fv_spmv_apply_pair(Ctx& c, int field, const uint32_t* indptr, const uint32_t* indices, const uint32_t* data,
                        size_t rows, size_t cols, const void* z1, const void* z2, uint32_t flags, void* out1, void* out2) {
  switch (field) {
    case 0: spmv_apply_pair_t<0>(c, indptr, indices, data, rows, cols, z1, z2, flags, out1, out2); break;
    case 1: spmv_apply_pair_t<1>(c, indptr, indices, data, rows, cols, z1, z2, flags, out1, out2); break;
    case 2: spmv_apply_pair_t<2>(c, indptr, indices, data, rows, cols, z1, z2, flags, out1, out2); break;
    case 3: spmv_apply_pair_t<3>(c, indptr, indices, data, rows, cols, z1, z2, flags, out1, out2); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}
void fv_r1cs_cross_term(Ctx& c, int field, const uint32_t* const* ip, const uint32_t* const* ix, const uint32_t* const* dt, size_t rows,
                        size_t cols, const void* z1, const void* z2, const void* e, const void* u, uint32_t flags, void* out) {
  switch (field) {
    case 0: r1cs_cross_term_t<0>(c, ip, ix, dt, rows, cols, z1, z2, e, u, flags, out); break;
    case 1: r1cs_cross_term_t<1>(c, ip, ix, dt, rows, cols, z1, z2, e, u, flags, out); break;
    case 2: r1cs_cross_term_t<2>(c, ip, ix, dt, rows, cols, z1, z2, e, u, flags, out); break;
    case 3: r1cs_cross_term_t<3>(c, ip, ix, dt, rows, cols, z1, z2, e, u, flags, out); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}
void fv_nifs_fold(Ctx& c, int field, const void* w1, const void* w2, size_t n_w, const void* e1, const void* t, size_t n_e,
                  const void* r, uint32_t flags, void* w, void* e) {
  switch (field) {
    case 0: nifs_fold_t<0>(c, w1, w2, n_w, e1, t, n_e, r, flags, w, e); break;
    case 1: nifs_fold_t<1>(c, w1, w2, n_w, e1, t, n_e, r, flags, w, e); break;
    case 2: nifs_fold_t<2>(c, w1, w2, n_w, e1, t, n_e, r, flags, w, e); break;
    case 3: nifs_fold_t<3>(c, w1, w2, n_w, e1, t, n_e, r, flags, w, e); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}
bool fv_batch_invert(Ctx& c, int field, const void* v, size_t n, uint32_t flags, void* out) {
  switch (field) {
    case 0: return batch_invert_t<0>(c, v, n, flags, out);
    case 1: return batch_invert_t<1>(c, v, n, flags, out);
    case 2: return batch_invert_t<2>(c, v, n, flags, out);
    case 3: return batch_invert_t<3>(c, v, n, flags, out);
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}
void fv_lincomb(Ctx& c, int field, const void* const* vecs, const size_t* lens, size_t k, const void* s, size_t n_out,
                uint32_t flags, void* out) {
  switch (field) {
    case 0: lincomb_t<0>(c, vecs, lens, k, s, n_out, flags, out); break;
    case 1: lincomb_t<1>(c, vecs, lens, k, s, n_out, flags, out); break;
    case 2: lincomb_t<2>(c, vecs, lens, k, s, n_out, flags, out); break;
    case 3: lincomb_t<3>(c, vecs, lens, k, s, n_out, flags, out); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}

#define FIELD_SWITCH(field, CALL)                                        \
  switch (field) {                                                       \
    case 0: FieldImpl<0>::CALL; break;                                   \
    case 1: FieldImpl<1>::CALL; break;                                   \
    case 2: FieldImpl<2>::CALL; break;                                   \
    case 3: FieldImpl<3>::CALL; break;                                   \
    default: throw Fail{NMX_E_ARG, "bad field id"};                      \
  }

void fv_axpy(Ctx& c, int field, const void* a, const void* b, const void* r, size_t n, uint32_t flags, void* out) {
  FIELD_SWITCH(field, axpy(c, a, b, r, n, flags, out));
}
void fv_axpy2(Ctx& c, int field, const void* a, const void* b, const void* cc, const void* r, size_t n, uint32_t flags,
              void* out) {
  FIELD_SWITCH(field, axpy2(c, a, b, cc, r, n, flags, out));
}
void fv_cross_term(Ctx& c, int field, const void* az, const void* bz, const void* cz, const void* e, const void* u,
                   size_t n, uint32_t flags, void* out) {
  FIELD_SWITCH(field, cross_term(c, az, bz, cz, e, u, n, flags, out));
}
void fv_cross_term2(Ctx& c, int field, const void* az, const void* bz, const void* cz, const void* e1, const void* e2,
                    const void* u, size_t n, uint32_t flags, void* out) {
  FIELD_SWITCH(field, cross_term2(c, az, bz, cz, e1, e2, u, n, flags, out));
}
void fv_vec_add(Ctx& c, int field, const void* a, const void* b, size_t n, uint32_t flags, void* out) {
  FIELD_SWITCH(field, vec_add(c, a, b, n, flags, out));
}
// ---- HyperKZG's fold loop as one call (round 6) ---------------------------------------------------------------------------
// `for i in 0..ell-1 { Pi[j] = x[ell-i-1] * (P[2j+1] - P[2j]) + P[2j] }` (src/provider/hyperkzg.rs:1085-1095): ell - 1 folds, each
// half as long as the one before -- as separate calls 19 launches at 2^20, of which the last dozen move a few KiB each and cost a
// launch (~10 us of queue time, ~5 us of host time) apiece.  Folds of more than kFoldChainLen inputs stay one launch each
// (BindTopFn); the rest run in ONE block: the first reads its input from HBM, every later one reads the previous output from
// LDS; every output is also written to its own HBM vector (they are what batch_commit commits to).
static constexpr uint32_t kFoldChainLen = 2048, kFoldChainMax = 12;
template <int FID> struct FoldChainArgs {
  const uint32_t* in;
  uint32_t* out[kFoldChainMax];
  Fp<FID> x[kFoldChainMax];  // challenges, internal form
  uint32_t len, k;           // input length (<= kFoldChainLen, even), folds (len >> k >= 1)
};
template <int FID> __global__ __launch_bounds__(1024) void k_fold_chain(FoldChainArgs<FID> a) {
  using F = Fp<FID>;
  constexpr uint32_t N0 = kFoldChainLen / 2, N1 = kFoldChainLen / 4;
  __shared__ uint32_t buf0[8 * N0], buf1[8 * N1];  // [word][element]: outputs of the even / odd folds
  const uint32_t t = threadIdx.x;
  uint32_t n_out = a.len >> 1;
  for (uint32_t i = 0; i < a.k; i++, n_out >>= 1) {
    uint32_t* dst = (i & 1u) ? buf1 : buf0;
    const uint32_t* src = (i & 1u) ? buf0 : buf1;
    const uint32_t dn = (i & 1u) ? N1 : N0, sn = (i & 1u) ? N0 : N1;
    for (uint32_t j = t; j < n_out; j += 1024u) {
      F lo, hi;
      if (i == 0) {
        lo = ld<FID>(a.in, 2 * (size_t)j), hi = ld<FID>(a.in, 2 * (size_t)j + 1);
      } else {
        uint32_t wl[8], wh[8];
#pragma unroll
        for (int w = 0; w < 8; w++) wl[w] = src[w * sn + 2 * j], wh[w] = src[w * sn + 2 * j + 1];
        lo = F::from_words(wl), hi = F::from_words(wh);
      }
      const F y = (lo + a.x[i] * F::sub2(hi, lo).norm()).norm().canon();
      uint32_t w8[8];
      y.to_words(w8);
#pragma unroll
      for (int w = 0; w < 8; w++) {
        a.out[i][8 * (size_t)j + w] = w8[w];
        if (i + 1 < a.k) dst[w * dn + j] = w8[w];
      }
    }
    __syncthreads();
  }
}
template <int FID>
static void fold_chain_t(Ctx& c, const void* p, size_t len, const void* xs, size_t k, uint32_t flags, void* const* outs) {
  using F = Fp<FID>;
  const bool mont = flags & NMX_SCALARS_MONT, async = (flags & NMX_ASYNC) != 0;
  DeviceBackend be(c, false, false);
  const uint32_t* cur = (const uint32_t*)p;
  size_t cur_len = len, i = 0;
  try {
  for (; i < k && cur_len > kFoldChainLen; i++, cur_len /= 2) {
    BindTopFn<FID> f{cur, cur + 8, (uint32_t*)outs[i], challenge<FID>((const uint8_t*)xs + 32 * i, mont), 2u};
    be.launch(f, (uint32_t)(cur_len / 2));
    cur = (const uint32_t*)outs[i];
  }
  if (i < k) {
    FoldChainArgs<FID> a;
    a.in = cur, a.len = (uint32_t)cur_len, a.k = (uint32_t)(k - i);
    require(a.k <= kFoldChainMax, NMX_E_ARG, "fold chain: more folds than the input has halvings");
    for (uint32_t q = 0; q < kFoldChainMax; q++) {
      a.out[q] = q < a.k ? (uint32_t*)outs[i + q] : nullptr;
      a.x[q] = q < a.k ? challenge<FID>((const uint8_t*)xs + 32 * (i + q), mont) : F::zero();
    }
    hipLaunchKernelGGL((k_fold_chain<FID>), dim3(1), dim3(1024), 0, c.stream, a);
    HIPCHK(hipGetLastError());
  }
  } catch (...) {
    (void)hipStreamSynchronize(c.stream);  // (a challenge out of range, a failed launch: nothing of this call still writes into outs)
    throw;
  }
  if (async) {
    async_mark(c);
    return;
  }
  stream_wait(c.stream);
}
void fv_fold_chain(Ctx& c, int field, const void* p, size_t len, const void* xs, size_t k, uint32_t flags, void* const* outs) {
  switch (field) {
    case 0: fold_chain_t<0>(c, p, len, xs, k, flags, outs); break;
    case 1: fold_chain_t<1>(c, p, len, xs, k, flags, outs); break;
    case 2: fold_chain_t<2>(c, p, len, xs, k, flags, outs); break;
    case 3: fold_chain_t<3>(c, p, len, xs, k, flags, outs); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}

void fv_bind(Ctx& c, int field, const void* z, size_t z_len, size_t lo_off, size_t hi_off, size_t stride,
             const void* r, size_t n_out, uint32_t flags, void* out) {
  FIELD_SWITCH(field, bind(c, z, z_len, lo_off, hi_off, stride, r, n_out, flags, out));
}

}  // namespace nmx
