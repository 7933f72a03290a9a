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
  } catch (const Error& e) {
    fprintf(stderr, "%s\n", e.what());
    return e.code == NMX_E_NO_DEVICE ? 3 : 2;
  }
  printf("host mirror ok\n");
  return 0;
}
