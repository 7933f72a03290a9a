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
    st<FID>(out, i, (ld<FID>(a, i) + r * ld<FID>(b, i) + r2 * ld<FID>(c, i)).norm());
  }
};
template <int FID> struct CrossTermFn {
  const uint32_t *az, *bz, *cz, *e;
  uint32_t* out;
  Fp<FID> u;  // u * 2^261
  Fp<FID> k;  // 2^522 / F: brings (az*F)(bz*F)/2^261 back to az*bz*F  (F = 1 canonical, 2^256 Montgomery)
  NMX_HD void operator()(uint32_t i) const {
    using F = Fp<FID>;
    F ab = (ld<FID>(az, i) * ld<FID>(bz, i)) * k;       // < 1.02 p
    F uc = u * ld<FID>(cz, i);                          // < 1.01 p
    F t = F::sub2(ab, uc).norm();                       // ab - uc + 2p
    t = F::sub2(t, ld<FID>(e, i)).norm();               // - e + 2p   (e canonical)
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
  size_t n;
  size_t used = 0;
  std::vector<std::pair<void*, const void*>> outs;  // (device, host)
  VecIO(Ctx& ctx, bool device, size_t n_, int n_vecs) : c(ctx), dev(device), n(n_) {
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
    HIPCHK(hipStreamSynchronize(c.stream));
  }
};

template <class Fn> static void timed_launch(Ctx& c, const Fn& f, size_t n, VecIO* io) {
  const bool prof = G.profiling;
  DeviceBackend be(c, false, prof);
  be.mark("kernel");
  be.launch(f, (uint32_t)n);
  be.mark("end");
  io->finish();
  if (prof && be.nmarks == 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    prof_store(&ms, 1);
  }
}

template <int FID> struct FieldImpl {
  using F = Fp<FID>;
  static void axpy(Ctx& c, const void* a, const void* b, const void* r, size_t n, uint32_t flags, void* out) {
    const bool mont = flags & NMX_SCALARS_MONT;
    VecIO io(c, flags & NMX_SCALARS_DEVICE, n, 3);
    AxpyFn<FID> f{io.in(a, n), io.in(b, n), io.out(out, n), challenge<FID>(r, mont)};
    timed_launch(c, f, n, &io);
  }
  static void axpy2(Ctx& c, const void* a, const void* b, const void* cc, const void* r, size_t n, uint32_t flags,
                    void* out) {
    const bool mont = flags & NMX_SCALARS_MONT;
    VecIO io(c, flags & NMX_SCALARS_DEVICE, n, 4);
    F ri = challenge<FID>(r, mont);
    Axpy2Fn<FID> f{io.in(a, n), io.in(b, n), io.in(cc, n), io.out(out, n), ri, (ri * ri).canon()};
    timed_launch(c, f, n, &io);
  }
  static void cross_term(Ctx& c, const void* az, const void* bz, const void* cz, const void* e, const void* u, size_t n,
                         uint32_t flags, void* out) {
    const bool mont = flags & NMX_SCALARS_MONT;
    VecIO io(c, flags & NMX_SCALARS_DEVICE, n, 5);
    // canonical data: k = 2^522 (R2); Montgomery data (F = 2^256): k = 2^522 / 2^256 = 2^266 (C266)
    F k = mont ? F::from_limbs(FpParams<FID>::C266) : F::from_limbs(FpParams<FID>::R2);
    CrossTermFn<FID> f{io.in(az, n), io.in(bz, n), io.in(cz, n), io.in(e, n), io.out(out, n), challenge<FID>(u, mont), k};
    timed_launch(c, f, n, &io);
  }
  static void vec_add(Ctx& c, const void* a, const void* b, size_t n, uint32_t flags, void* out) {
    VecIO io(c, flags & NMX_SCALARS_DEVICE, n, 3);
    VecAddFn<FID> f{io.in(a, n), io.in(b, n), io.out(out, n)};
    timed_launch(c, f, n, &io);
  }
  // out[i] = z[lo_off + i*stride] + r * (z[hi_off + i*stride] - z[lo_off + i*stride]),  i < n_out
  static void bind(Ctx& c, const void* z, size_t z_len, size_t lo_off, size_t hi_off, size_t stride, const void* r,
                   size_t n_out, uint32_t flags, void* out) {
    const bool mont = flags & NMX_SCALARS_MONT;
    const bool dev = flags & NMX_SCALARS_DEVICE;
    VecIO io(c, dev, z_len + n_out, 2);
    const uint32_t* zd = io.in(z, z_len);
    uint32_t* od = (dev && out == z) ? (uint32_t*)out : io.out(out, n_out);
    BindTopFn<FID> f{zd + 8 * lo_off, zd + 8 * hi_off, od, challenge<FID>(r, mont), (uint32_t)stride};
    timed_launch(c, f, n_out, &io);
  }
};

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
void fv_vec_add(Ctx& c, int field, const void* a, const void* b, size_t n, uint32_t flags, void* out) {
  FIELD_SWITCH(field, vec_add(c, a, b, n, flags, out));
}
void fv_bind(Ctx& c, int field, const void* z, size_t z_len, size_t lo_off, size_t hi_off, size_t stride,
             const void* r, size_t n_out, uint32_t flags, void* out) {
  FIELD_SWITCH(field, bind(c, z, z_len, lo_off, hi_off, stride, r, n_out, flags, out));
}

}  // namespace nmx
