// tests/cpp/host_mirror_test.cpp -- exercises include/nova_mi355x.hpp (C++ host mirror of DlogGroupExt /
// CommitmentEngine) against the oracle (libnova_ref.so).  Reads like the reference's blitzar tests
// (/root/reference/src/provider/blitzar.rs:48-214).  Exit code 0 = pass, 3 = no GPU (NMX_E_NO_DEVICE raised), else fail.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "../../include/nova_mi355x.hpp"

extern "C" {
int ref_msm(int curve, const uint8_t* s, const uint8_t* b, size_t n, uint8_t* out, uint8_t* inf);
int ref_commit(int curve, const uint8_t* v, const uint8_t* ck, size_t n, const uint8_t* h, const uint8_t* r, uint8_t* out,
               uint8_t* inf);
int ref_sequential_bases(int curve, const uint8_t* gen, uint64_t k0, size_t n, uint8_t* out);
}
using namespace nova::provider;

// ---- Spartan's sum-check provers through the C++ mirror (nova::spartan::Sumcheck) against the oracle's restatement -----------------
extern "C" {
typedef int (*ref_transcript_fn)(void* ctx, const uint8_t* coeffs, size_t n_coeffs, uint8_t* challenge32);
int ref_sumcheck_prove_cubic3(int field, const uint8_t* claim, const uint8_t* taus, size_t nr, const uint8_t* A, const uint8_t* B, const uint8_t* C,
                              ref_transcript_fn cb, void* ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims);
int ref_sumcheck_prove_quad_prod(int field, const uint8_t* claim, size_t nr, const uint8_t* A, const uint8_t* B, ref_transcript_fn cb, void* ctx,
                                 uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims);
int ref_sumcheck_prove_batch_eval(int field, const uint8_t* claims, const size_t* num_rounds, const uint8_t* const* polys,
                                  const uint8_t* const* eq_points, const uint8_t* coeffs, size_t k, ref_transcript_fn cb, void* ctx,
                                  uint8_t* out_polys, uint8_t* out_r, uint8_t* out_finals);
}
// a stand-in transcript: chained 64-bit mixing of everything absorbed, 31 bytes out (< 2^248 < every modulus)
struct Standin {
  uint64_t s[4] = {1, 2, 3, 4};
  static uint64_t mix(uint64_t x) {
    x ^= x >> 30, x *= 0xbf58476d1ce4e5b9ull, x ^= x >> 27, x *= 0x94d049bb133111ebull, x ^= x >> 31;
    return x;
  }
  Scalar round(const std::vector<Scalar>& co) {
    for (const Scalar& c : co)
      for (int i = 0; i < 32; i += 8) {
        uint64_t w;
        memcpy(&w, c.data() + i, 8);
        s[(i / 8) & 3] = mix(s[(i / 8) & 3] ^ w) + s[(i / 8 + 1) & 3];
      }
    Scalar r;
    for (int i = 0; i < 4; i++) {
      s[i] = mix(s[i] + s[(i + 3) & 3] + 0x9e3779b97f4a7c15ull);
      memcpy(r.data() + 8 * i, &s[i], 8);
    }
    r[31] = 0;
    return r;
  }
};
static int standin_cb(void* ctx, const uint8_t* coeffs, size_t n, uint8_t* out) {
  std::vector<Scalar> co(n);
  for (size_t i = 0; i < n; i++) memcpy(co[i].data(), coeffs + 32 * i, 32);
  const Scalar r = static_cast<Standin*>(ctx)->round(co);
  memcpy(out, r.data(), 32);
  return 0;
}
template <int FIELD> static int run_sumcheck(int topmask) {
  using SC = nova::spartan::Sumcheck<FIELD>;
  std::mt19937_64 rng(11);
  auto rand_vec = [&](size_t n) {
    std::vector<Scalar> v(n);
    for (auto& s : v) {
      for (int i = 0; i < 32; i += 8) {
        uint64_t w = rng();
        memcpy(s.data() + i, &w, 8);
      }
      s[31] &= topmask;
    }
    return v;
  };
  auto same = [](const nova::spartan::SumcheckProof& p, size_t nco, const std::vector<uint8_t>& polys, const std::vector<uint8_t>& r,
                 const std::vector<uint8_t>& cl) {
    for (size_t j = 0; j < p.polys.size(); j++) {
      for (size_t i = 0; i < nco; i++)
        if (memcmp(p.polys[j][i].data(), polys.data() + 32 * (nco * j + i), 32)) return false;
      if (memcmp(p.r[j].data(), r.data() + 32 * j, 32)) return false;
    }
    for (size_t i = 0; i < p.claims.size(); i++)
      if (memcmp(p.claims[i].data(), cl.data() + 32 * i, 32)) return false;
    return true;
  };
  for (size_t l : {1ul, 5ul, 11ul}) {
    const size_t n = (size_t)1 << l;
    const auto A = rand_vec(n), B = rand_vec(n), C = rand_vec(n), taus = rand_vec(l);
    const Scalar claim = rand_vec(1)[0];  // any claim: the prover does not check it (the verifier would)
    {
      Standin t1, t2;
      const auto got = SC::prove_cubic_with_three_inputs(claim, taus, A, B, C, t1);
      std::vector<uint8_t> p(128 * l), r(32 * l), c(96);
      if (ref_sumcheck_prove_cubic3(FIELD, claim.data(), taus[0].data(), l, A[0].data(), B[0].data(), C[0].data(), standin_cb, &t2, p.data(), r.data(), c.data())) return 1;
      if (got.polys.size() != l || !same(got, 4, p, r, c)) return 1;
    }
    {
      Standin t1, t2;
      const auto got = SC::prove_quad_prod(claim, l, A, B, t1);
      std::vector<uint8_t> p(96 * l), r(32 * l), c(64);
      if (ref_sumcheck_prove_quad_prod(FIELD, claim.data(), l, A[0].data(), B[0].data(), standin_cb, &t2, p.data(), r.data(), c.data())) return 1;
      if (!same(got, 3, p, r, c)) return 1;
    }
  }
  {  // batch_eval over polynomials of different sizes (spartan/mod.rs:377-437: W and E)
    const std::vector<std::vector<Scalar>> polys = {rand_vec(1 << 9), rand_vec(1 << 12), rand_vec(1 << 4)}, pts = {rand_vec(9), rand_vec(12), rand_vec(4)};
    const auto claims = rand_vec(3), coeffs = rand_vec(3);
    Standin t1, t2;
    const auto got = SC::prove_batch_eval(claims, polys, pts, coeffs, t1);
    const size_t nr[3] = {9, 12, 4};
    const uint8_t* pp[3] = {polys[0][0].data(), polys[1][0].data(), polys[2][0].data()};
    const uint8_t* qp[3] = {pts[0][0].data(), pts[1][0].data(), pts[2][0].data()};
    std::vector<uint8_t> p(96 * 12), r(32 * 12), f(96);
    if (ref_sumcheck_prove_batch_eval(FIELD, claims[0].data(), nr, pp, qp, coeffs[0].data(), 3, standin_cb, &t2, p.data(), r.data(), f.data())) return 1;
    if (got.polys.size() != 12 || !same(got, 3, p, r, f)) return 1;
  }
  // a transcript that throws: the call fails (NMX_E_ARG), nothing crosses the C frame
  struct Throws {
    Scalar round(const std::vector<Scalar>&) { throw std::runtime_error("refused"); }
  } bad;
  try {
    const auto A = rand_vec(16), B = rand_vec(16);
    SC::prove_quad_prod(A[0], 4, A, B, bad);
    return 1;
  } catch (const Error& e) {
    if (e.code != NMX_E_ARG) return 1;
  }
  return 0;
}

// ---- the inner-product argument through the C++ mirror (nova::ipa::prove) against the oracle's key-folding restatement ---------------
extern "C" {
typedef int (*ref_ipa_transcript_fn)(void* ctx, const uint8_t* L_xy64, int L_is_inf, const uint8_t* R_xy64, int R_is_inf, uint8_t* r32);
int ref_ipa_prove(int curve, const uint8_t* ck_xy64, const uint8_t* ck_c_xy64, const uint8_t* a_le32, const uint8_t* b_le32, size_t n,
                  ref_ipa_transcript_fn cb, void* ctx, uint8_t* out_L, uint8_t* out_R, uint8_t* out_inf, uint8_t* out_a_hat);
}
struct IpaStandin {  // absorb L and R, squeeze r: the sum-check stand-in over the points' words
  Standin st;
  Scalar round(const Point& L, const Point& R) {
    std::vector<Scalar> co(4);
    memcpy(co[0].data(), L.xy.data(), 32), memcpy(co[1].data(), L.xy.data() + 32, 32);
    memcpy(co[2].data(), R.xy.data(), 32), memcpy(co[3].data(), R.xy.data() + 32, 32);
    Scalar r = st.round(co);
    r[0] |= 1;  // never zero
    return r;
  }
};
static int ipa_standin_cb(void* ctx, const uint8_t* L, int Li, const uint8_t* R, int Ri, uint8_t* out) {
  Point l, r;
  memcpy(l.xy.data(), L, 64), memcpy(r.xy.data(), R, 64);
  l.is_inf = Li != 0, r.is_inf = Ri != 0;
  const Scalar c = static_cast<IpaStandin*>(ctx)->round(l, r);
  memcpy(out, c.data(), 32);
  return 0;
}
template <int CURVE> static int run_ipa(const uint8_t gen[64], int topmask) {
  std::mt19937_64 rng(11);
  for (size_t n : {1ul, 2ul, 64ul, 512ul}) {
    std::vector<Affine> bases(n + 1);
    ref_sequential_bases(CURVE, gen, 4242, n + 1, bases[0].data());
    std::vector<Scalar> a(n), b(n);
    for (auto* v : {&a, &b})
      for (auto& s : *v) {
        for (int i = 0; i < 32; i += 8) {
          uint64_t w = rng();
          memcpy(s.data() + i, &w, 8);
        }
        s[31] &= topmask;
      }
    CommitmentKey ck(CURVE, std::vector<Affine>(bases.begin(), bases.begin() + n), bases[n]);
    const Affine ck_c = bases[n];  // (any point of the group serves as the scaled one-point key)
    IpaStandin t1, t2;
    const auto got = nova::ipa::prove(ck, ck_c, a, b, t1);
    size_t rounds = 0;
    while (((size_t)1 << rounds) < n) rounds++;
    std::vector<uint8_t> L(64 * rounds + 1), R(64 * rounds + 1), inf(2 * rounds + 1);
    Scalar a_hat;
    if (ref_ipa_prove(CURVE, bases[0].data(), ck_c.data(), a[0].data(), b[0].data(), n, ipa_standin_cb, &t2, L.data(), R.data(), inf.data(), a_hat.data()))
      return 1;
    if (got.L_vec.size() != rounds || got.R_vec.size() != rounds || !(got.a_hat == a_hat)) return 1;
    for (size_t k = 0; k < rounds; k++) {
      if (memcmp(got.L_vec[k].xy.data(), L.data() + 64 * k, 64) || memcmp(got.R_vec[k].xy.data(), R.data() + 64 * k, 64)) return 1;
      if (got.L_vec[k].is_inf != (inf[2 * k] != 0) || got.R_vec[k].is_inf != (inf[2 * k + 1] != 0)) return 1;
    }
  }
  try {  // InvalidInputLength
    CommitmentKey ck(CURVE, std::vector<Affine>(4), Affine{});
    IpaStandin t;
    nova::ipa::prove(ck, Affine{}, std::vector<Scalar>(4), std::vector<Scalar>(3), t);
    return 1;
  } catch (const std::invalid_argument&) {
  }
  return 0;
}

template <int CURVE> static int run(const uint8_t gen[64], int topmask) {
  const size_t n = 100;
  std::vector<Affine> bases(n + 1);
  ref_sequential_bases(CURVE, gen, 12345, n + 1, bases[0].data());
  std::mt19937_64 rng(7);
  std::vector<Scalar> sc(n);
  for (auto& s : sc) {
    for (int i = 0; i < 32; i += 8) {
      uint64_t v = rng();
      memcpy(s.data() + i, &v, 8);
    }
    s[31] &= topmask;  // < modulus
  }
  Point exp;
  uint8_t inf;
  // test_vartime_multiscalar_mul (blitzar.rs:102-115)
  Point got = DlogGroupExt<CURVE>::vartime_multiscalar_mul(sc, std::vector<Affine>(bases.begin(), bases.begin() + n));
  ref_msm(CURVE, sc[0].data(), bases[0].data(), n, exp.xy.data(), &inf);
  exp.is_inf = inf;
  if (!(got == exp)) return 1;
  // the slice form keeps the caller's bases resident: the second call over the same vector is a cache hit, uploads
  // nothing and returns the same point (the reference passes &ck.ck[..n] on every call: pedersen.rs:263-270)
  if (DlogGroupExt<CURVE>::min_gpu_n() == 0) return 1;
  {
    const std::vector<Affine> held(bases.begin(), bases.begin() + n);
    uint64_t st0[NMX_STAT_COUNT], st1[NMX_STAT_COUNT], st2[NMX_STAT_COUNT];
    nmx_cache_clear();
    nmx_cache_configure(0, 64, 0);  // n = 100 is below the default caching threshold (128)
    nmx_stats(st0, NMX_STAT_COUNT);
    if (!(DlogGroupExt<CURVE>::vartime_multiscalar_mul(sc, held) == exp)) return 1;
    nmx_stats(st1, NMX_STAT_COUNT);
    if (!(DlogGroupExt<CURVE>::vartime_multiscalar_mul(sc, held) == exp)) return 1;
    nmx_stats(st2, NMX_STAT_COUNT);
    if (st1[NMX_STAT_CACHE_UPLOADS] != st0[NMX_STAT_CACHE_UPLOADS] + 1) return 1;
    if (st2[NMX_STAT_CACHE_UPLOADS] != st1[NMX_STAT_CACHE_UPLOADS] || st2[NMX_STAT_CACHE_HITS] != st1[NMX_STAT_CACHE_HITS] + 1) return 1;
    if (st2[NMX_STAT_BASE_BYTES_H2D] != st1[NMX_STAT_BASE_BYTES_H2D]) return 1;
    nmx_cache_configure(0, 128, 0);
  }
  // empty -> identity (blitzar.rs:48-66)
  if (!DlogGroupExt<CURVE>::vartime_multiscalar_mul({}, std::vector<Affine>{}).is_inf) return 1;
  // commit with blinding over a registered key (pedersen.rs:263-270)
  CommitmentKey ck(CURVE, std::vector<Affine>(bases.begin(), bases.begin() + n), bases[n]);
  Scalar r = sc[3];
  got = CommitmentEngine<CURVE>::commit(ck, sc, r);
  ref_commit(CURVE, sc[0].data(), bases[0].data(), n, bases[n].data(), r.data(), exp.xy.data(), &inf);
  exp.is_inf = inf;
  if (!(got == exp)) return 1;
  {  // the same commitment begun, another one computed beside it, then collected (nmx_commit_begin / nmx_commit_finish)
    auto pending = CommitmentEngine<CURVE>::commit_begin(ck, sc, r);
    if (!(CommitmentEngine<CURVE>::commit(ck, sc, r) == exp)) return 1;
    if (!(pending.finish() == exp)) return 1;
    auto dropped = CommitmentEngine<CURVE>::commit_begin(ck, sc, r);  // its destructor retires the ticket
  }
  // ragged batch (blitzar.rs:185-213)
  std::vector<std::vector<Scalar>> vs;
  for (size_t L : {0ul, 1ul, 37ul, 100ul}) vs.emplace_back(sc.begin(), sc.begin() + L);
  auto res = CommitmentEngine<CURVE>::batch_commit(ck, vs);
  for (size_t j = 0; j < vs.size(); j++) {
    ref_msm(CURVE, sc[0].data(), bases[0].data(), vs[j].size(), exp.xy.data(), &inf);
    exp.is_inf = inf;
    if (!(res[j] == exp)) return 1;
  }
  // batch of small scalars (traits.rs:109-117): every vector equals the field-scalar MSM of the widened values
  {
    std::vector<std::vector<uint64_t>> us;
    std::vector<std::vector<Scalar>> wide;
    for (size_t L : {0ul, 5ul, 64ul, 100ul}) {
      std::vector<uint64_t> u(L);
      std::vector<Scalar> w(L);
      for (size_t i = 0; i < L; i++) {
        u[i] = rng() & 0xFFFFFFFFFFull;  // 40 bits
        w[i].fill(0);
        memcpy(w[i].data(), &u[i], 8);
      }
      us.push_back(u);
      wide.push_back(w);
    }
    auto small = DlogGroupExt<CURVE>::batch_vartime_multiscalar_mul_small(us, ck, 40);
    auto small_auto = DlogGroupExt<CURVE>::batch_vartime_multiscalar_mul_small(us, ck);
    for (size_t j = 0; j < us.size(); j++) {
      if (wide[j].empty()) {
        if (!small[j].is_inf || !small_auto[j].is_inf) return 1;
        continue;
      }
      ref_msm(CURVE, wide[j][0].data(), bases[0].data(), wide[j].size(), exp.xy.data(), &inf);
      exp.is_inf = inf;
      if (!(small[j] == exp) || !(small_auto[j] == exp)) return 1;
    }
  }
  // load_setup from a PEDERSEN_KEY file written by the harness: h = bases[n], ck = the same sequence (pedersen.rs:318-340)
  if (const char* dir = getenv("NMX_TEST_KEYDIR")) {
    CommitmentKey fk = CommitmentKey::load_keyfile(CURVE, std::string(dir) + "/curve" + std::to_string(CURVE) + ".key", n);
    if (fk.len() != 128 || !(fk.h() == bases[n])) return 1;
    if (!(CommitmentEngine<CURVE>::commit(fk, sc, r) == got)) return 1;
    try {
      CommitmentKey::load_keyfile(CURVE, std::string(dir) + "/missing.key", n);
      return 1;
    } catch (const Error& e) {
      if (e.code != NMX_E_IO) return 1;
    }
  }
  // assert!(ck.ck.len() >= v.len())
  try {
    std::vector<Scalar> too_long(n + 1, sc[0]);
    CommitmentEngine<CURVE>::commit(ck, too_long, r);
    return 1;
  } catch (const std::invalid_argument&) {
  }
  return 0;
}

int main() {
  uint8_t g_bn[64] = {0}, g_pallas[64] = {0};
  g_bn[0] = 1;
  g_bn[32] = 2;  // BN254 G1 generator (1, 2)
  // Pallas generator (-1, 2): p - 1 little-endian
  const uint8_t pm1[32] = {0x00, 0x00, 0x00, 0x00, 0xed, 0x30, 0x2d, 0x99, 0x1b, 0xf9, 0x4c, 0x09, 0xfc, 0x98, 0x46, 0x22,
                           0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x40};
  memcpy(g_pallas, pm1, 32);
  g_pallas[32] = 2;
  try {
    if (run<NMX_BN254_G1>(g_bn, 0x1f)) return 1;
    if (run<NMX_PALLAS>(g_pallas, 0x3f)) return 1;
    if (run_sumcheck<NMX_F_BN254_FR>(0x1f)) return 1;
    if (run_sumcheck<NMX_F_PASTA_FQ>(0x3f)) return 1;
    if (run_ipa<NMX_BN254_G1>(g_bn, 0x1f)) return 1;
    if (run_ipa<NMX_PALLAS>(g_pallas, 0x3f)) return 1;
  } catch (const Error& e) {
    fprintf(stderr, "%s\n", e.what());
    return e.code == NMX_E_NO_DEVICE ? 3 : 2;
  }
  printf("host mirror ok\n");
  return 0;
}
