// bench/csnark_replay.cpp -- the chained CompressedSNARK::prove replay (bench.py compressed_snark_sequence; the reference:
// /root/reference/src/nova/mod.rs:793-881) driven from C++ through include/nova_mi355x.hpp's `resident` functions, every vector in
// HBM: the same provider calls in the same order as the Python driver, without its per-call marshalling.  bench.py writes the
// instance (matrices, running instance, the sampled randomness, blinds) to a file, runs this program, reads back every commitment,
// round polynomial, evaluation and the batched witnesses, and compares them with the oracle's run: this program only TIMES and
// REPORTS -- the checker stays where it is.
//   csnark_replay <instance file> <output file> <steps> <warmup> [serial_snarks]
// File format (both ways): records  u32 name_len | name | u64 bytes | data.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <string>
#include <thread>

#include "../include/nova_mi355x.hpp"

extern "C" {  // tests/standin/standin_transcript.c (test / bench scaffolding: a native stand-in for the Keccak transcript)
void standin_init(void* t, uint64_t seed);
void standin_absorb(void* t, const uint8_t* data, size_t n);
void standin_squeeze(void* t, uint8_t out[32]);
int standin_transcript(void* ctx, const uint8_t* coeffs, size_t n_coeffs, uint8_t* challenge32);
int standin_ipa_transcript(void* ctx, const uint8_t* L_xy64, int L_is_inf, const uint8_t* R_xy64, int R_is_inf, uint8_t* r32);
}
using namespace nova;
using provider::Scalar;
using Blob = std::vector<uint8_t>;
namespace R = nova::resident;

#define HIPOK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                        \
      exit(2);                                                                       \
    }                                                                                \
  } while (0)

static std::map<std::string, Blob> read_records(const char* path) {
  std::map<std::string, Blob> m;
  FILE* f = fopen(path, "rb");
  if (!f) {
    perror(path);
    exit(2);
  }
  for (;;) {
    uint32_t nl;
    if (fread(&nl, 4, 1, f) != 1) break;
    std::string name(nl, '\0');
    uint64_t bytes;
    if (fread(&name[0], 1, nl, f) != nl || fread(&bytes, 8, 1, f) != 1) exit(2);
    Blob b(bytes);
    if (bytes && fread(b.data(), 1, bytes, f) != bytes) exit(2);
    m[name] = std::move(b);
  }
  fclose(f);
  return m;
}
static void write_record(FILE* f, const std::string& name, const void* data, size_t bytes) {
  const uint32_t nl = (uint32_t)name.size();
  const uint64_t b = bytes;
  fwrite(&nl, 4, 1, f), fwrite(name.data(), 1, nl, f), fwrite(&b, 8, 1, f);
  if (bytes) fwrite(data, 1, bytes, f);
}

// ---- 256-bit modular arithmetic for the handful of host scalars of the sequence (u1 + r u2, the inner claim, -r, r^2) ----------
struct U256 {
  uint64_t l[4];
};
static U256 from_scalar(const Scalar& s) {
  U256 r;
  memcpy(r.l, s.data(), 32);
  return r;
}
static Scalar to_scalar(const U256& a) {
  Scalar s;
  memcpy(s.data(), a.l, 32);
  return s;
}
static bool geq(const U256& a, const U256& b) {
  for (int i = 3; i >= 0; i--)
    if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
  return true;
}
static U256 sub(const U256& a, const U256& b) {
  U256 r;
  unsigned __int128 br = 0;
  for (int i = 0; i < 4; i++) {
    const unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - br;
    r.l[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  return r;
}
static U256 addmod(const U256& a, const U256& b, const U256& p) {  // a, b < p < 2^255
  U256 r;
  unsigned __int128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (unsigned __int128)a.l[i] + b.l[i];
    r.l[i] = (uint64_t)c;
    c >>= 64;
  }
  return geq(r, p) ? sub(r, p) : r;
}
static U256 mulmod(const U256& a, const U256& b, const U256& p) {  // double-and-add: a few microseconds, used ~10 times per sequence
  U256 acc{{0, 0, 0, 0}};
  for (int i = 255; i >= 0; i--) {
    acc = addmod(acc, acc, p);
    if ((b.l[i / 64] >> (i % 64)) & 1) acc = addmod(acc, a, p);
  }
  return acc;
}
static U256 modulus(int field) {
  static const U256 P[4] = {
      {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}},   // BN254 Fq
      {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}},   // BN254 Fr
      {{0x992d30ed00000001ull, 0x224698fc094cf91bull, 0x0000000000000000ull, 0x4000000000000000ull}},   // Pallas Fp
      {{0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0000000000000000ull, 0x4000000000000000ull}}};  // Pallas Fq
  return P[field];
}

struct Transcript {
  uint8_t state[48];
  explicit Transcript(uint64_t seed) { standin_init(state, seed); }
  void absorb(const void* d, size_t n) { standin_absorb(state, (const uint8_t*)d, n); }
  Scalar squeeze() {
    Scalar s;
    standin_squeeze(state, s.data());
    return s;
  }
};

struct Dev {  // a device buffer of n 32-byte elements
  void* p = nullptr;
  size_t n = 0;
  void alloc(size_t n_) {
    n = n_;
    HIPOK(hipMalloc(&p, (n ? n : 1) * 32));
  }
  void upload(const Blob& b) {
    alloc(b.size() / 32);
    HIPOK(hipMemcpy(p, b.data(), b.size(), hipMemcpyHostToDevice));
  }
  Blob download() const {
    Blob b(n * 32);
    HIPOK(hipMemcpy(b.data(), p, b.size(), hipMemcpyDeviceToHost));
    return b;
  }
};

static const uint64_t kSeed = 2025;  // bench.SPARTAN_SEED

struct Side {
  std::string tag;
  int cid = 0, fid = 0;
  size_t ell = 0, n = 0;
  U256 p{};
  provider::CommitmentKey* ck = nullptr;
  uint64_t mats[3] = {0, 0, 0};
  Dev W1, W2, E1, zeros;
  Scalar u1, X1, u2, X2, rW, rE, rT;
  // scratch, allocated once
  Dev z1, z2, Z, AZ, BZ, CZ, E2, T, W, E, zc, Az, Bz, Cz, uCzE, erx, eA, eB, eC, ABC, Wc, Ec, wj, chain, Bp, hs[3], hq;
  provider::Affine ipa_u{};      // ck_c = CE::setup(b"ipa", 1) (ipa_pc.rs:50)
  std::vector<Scalar> ee_point;  // handed from the sum-check sequence to EE::prove
  uint8_t ee_tr[48];
  // results of the last run
  std::map<std::string, Blob> out;
  void put(const std::string& k, const void* d, size_t b) { out[tag + "." + k] = Blob((const uint8_t*)d, (const uint8_t*)d + b); }
  void put_point(const std::string& k, const provider::Point& pt) {
    uint8_t b[65];
    memcpy(b, pt.xy.data(), 64);
    b[64] = pt.is_inf ? 1 : 0;
    put(k, b, 65);
  }
};
static Scalar scalar_of(const Blob& b) {
  Scalar s;
  memcpy(s.data(), b.data(), 32);
  return s;
}

static void load_side(Side& s, const std::map<std::string, Blob>& in, const std::string& tag) {
  auto get = [&](const std::string& k) -> const Blob& {
    auto it = in.find(tag + "." + k);
    if (it == in.end()) {
      fprintf(stderr, "missing record %s.%s\n", tag.c_str(), k.c_str());
      exit(2);
    }
    return it->second;
  };
  s.tag = tag;
  uint64_t meta[4];
  memcpy(meta, get("meta").data(), 32);  // cid, fid, ell, k0
  s.cid = (int)meta[0], s.fid = (int)meta[1], s.ell = meta[2], s.n = (size_t)1 << s.ell;
  s.p = modulus(s.fid);
  s.ck = new provider::CommitmentKey(provider::CommitmentKey::generate(s.cid, s.n, meta[3]));
  for (int j = 0; j < 3; j++) {
    const Blob &ip = get("ip" + std::to_string(j)), &ix = get("ix" + std::to_string(j)), &dt = get("dt" + std::to_string(j));
    provider::check(nmx_spmv_register(s.fid, (const uint64_t*)ip.data(), (const uint64_t*)ix.data(), dt.data(), ip.size() / 8 - 1, 2 * s.n, 0, &s.mats[j]));
  }
  s.W1.upload(get("W1")), s.W2.upload(get("W2")), s.E1.upload(get("E1"));
  s.zeros.alloc(s.n);
  HIPOK(hipMemset(s.zeros.p, 0, s.n * 32));
  s.u1 = scalar_of(get("u1")), s.X1 = scalar_of(get("X1")), s.u2 = scalar_of(get("u2")), s.X2 = scalar_of(get("X2"));
  s.rW = scalar_of(get("r_W")), s.rE = scalar_of(get("r_E")), s.rT = scalar_of(get("r_T"));
  const size_t n = s.n;
  for (Dev* d : {&s.z1, &s.z2, &s.Z, &s.zc, &s.eA, &s.eB, &s.eC, &s.ABC}) d->alloc(2 * n);
  for (Dev* d : {&s.AZ, &s.BZ, &s.CZ, &s.E2, &s.T, &s.W, &s.E, &s.Az, &s.Bz, &s.Cz, &s.uCzE, &s.erx, &s.Wc, &s.Ec, &s.wj, &s.chain, &s.Bp, &s.hq}) d->alloc(n);
  for (Dev& d : s.hs) d.alloc(n);
  s.ipa_u = provider::CommitmentKey::generate(s.cid, 0, 424243).h();  // bench.IPA_GENERATOR_K0
}

// sample_random_instance_witness + NIFSRelaxed::prove (bench.py relaxed_fold_sequence; r1cs/mod.rs:786-830, 629-661, 1070-1107)
static void relaxed_fold(Side& s, Scalar& u_out, Scalar& X_out) {
  const int f = s.fid;
  const size_t n = s.n;
  Transcript tr(kSeed + 1);
  R::concat_z(f, s.W2.p, n, s.u2, s.X2, 2 * n, s.z2.p);
  R::r1cs_cross_term(s.mats, s.z2.p, 2 * n, s.zeros.p, s.u2, s.E2.p);
  const uint64_t t = R::commit_begin(*s.ck, s.W2.p, n, s.rW);  // rayon::join(commit(W), commit(E)), r1cs/mod.rs:815-818
  const provider::Point cE2 = R::commit(*s.ck, s.E2.p, n, s.rE);
  const provider::Point cW2 = R::commit_finish(t);
  R::concat_z(f, s.W1.p, n, s.u1, s.X1, 2 * n, s.z1.p);
  R::vec_add(f, s.z1.p, s.z2.p, 2 * n, s.Z.p);
  const Scalar u12 = to_scalar(addmod(from_scalar(s.u1), from_scalar(s.u2), s.p));
  void* outs[3] = {s.AZ.p, s.BZ.p, s.CZ.p};
  R::multiply_vec3(s.mats, false, s.Z.p, 2 * n, outs);
  R::cross_term2(f, s.AZ.p, s.BZ.p, s.CZ.p, s.E1.p, s.E2.p, u12, n, s.T.p);
  const provider::Point cT = R::commit(*s.ck, s.T.p, n, s.rT);
  for (const provider::Point* c : {&cW2, &cE2, &cT}) tr.absorb(c->xy.data(), 64);
  const Scalar r = tr.squeeze();
  R::axpy(f, s.W1.p, s.W2.p, r, n, s.W.p);
  R::axpy2(f, s.E1.p, s.T.p, s.E2.p, r, n, s.E.p);
  const U256 rr = from_scalar(r);
  u_out = to_scalar(addmod(from_scalar(s.u1), mulmod(rr, from_scalar(s.u2), s.p), s.p));
  X_out = to_scalar(addmod(from_scalar(s.X1), mulmod(rr, from_scalar(s.X2), s.p), s.p));
  s.put_point("fold.cW2", cW2), s.put_point("fold.cE2", cE2), s.put_point("fold.cT", cT);
  s.put("fold.r", r.data(), 32);
}

static std::vector<Scalar> scalars_of(const std::vector<uint8_t>& b) {
  std::vector<Scalar> v(b.size() / 32);
  for (size_t i = 0; i < v.size(); i++) memcpy(v[i].data(), b.data() + 32 * i, 32);
  return v;
}

// RelaxedR1CSSNARK::prove up to the evaluation argument (bench.py spartan_sequence; spartan/snark.rs:133-233), then, for the primary,
// EE::prove of HyperKZG on the batched witness where it lies (bench.py hyperkzg_sequence; hyperkzg.rs:926-1116)
static void spartan_prove(Side& s, const Scalar& u, const Scalar& X, bool with_ee) {
  const int f = s.fid;
  const size_t n = s.n, ell = s.ell;
  Transcript tr(kSeed);
  std::vector<Scalar> tau(ell);
  for (Scalar& t : tau) t = tr.squeeze();
  R::concat_z(f, s.W.p, n, u, X, 2 * n, s.zc.p);
  void* o3[3] = {s.Az.p, s.Bz.p, s.Cz.p};
  R::multiply_vec3(s.mats, false, s.zc.p, 2 * n, o3);
  R::axpy(f, s.E.p, s.Cz.p, u, n, s.uCzE.p);
  const Scalar zero{};
  const R::Proof outer = R::prove_cubic(f, zero, tau, s.Az.p, s.Bz.p, s.uCzE.p, &standin_transcript, tr.state);
  const std::vector<Scalar> r_x = scalars_of(outer.r);
  const std::vector<Scalar> ev = R::mle_multi_evaluate(f, {s.Cz.p, s.E.p}, n, r_x);  // claim_Cz, eval_E
  tr.absorb(outer.claims.data(), 64);                                                // claim_Az, claim_Bz
  tr.absorb(ev[0].data(), 32), tr.absorb(ev[1].data(), 32);
  const Scalar r = tr.squeeze();
  const U256 rr = from_scalar(r), cAz = from_scalar(scalar_of(Blob(outer.claims.begin(), outer.claims.begin() + 32))),
             cBz = from_scalar(scalar_of(Blob(outer.claims.begin() + 32, outer.claims.begin() + 64)));
  const U256 r2 = mulmod(rr, rr, s.p);
  const Scalar claim_inner = to_scalar(addmod(addmod(cAz, mulmod(rr, cBz, s.p), s.p), mulmod(r2, from_scalar(ev[0]), s.p), s.p));
  R::eq_evals(f, r_x, s.erx.p);
  void* t3[3] = {s.eA.p, s.eB.p, s.eC.p};
  R::multiply_vec3(s.mats, true, s.erx.p, n, t3);
  R::axpy2(f, s.eA.p, s.eB.p, s.eC.p, r, 2 * n, s.ABC.p);
  const R::Proof inner = R::prove_quad(f, claim_inner, ell + 1, s.ABC.p, s.zc.p, &standin_transcript, tr.state);
  std::vector<Scalar> r_y = scalars_of(inner.r);
  const std::vector<Scalar> ry1(r_y.begin() + 1, r_y.end());
  const Scalar eval_W = R::mle_evaluate(f, s.W.p, n, ry1);
  tr.absorb(eval_W.data(), 32);
  const Scalar rho = tr.squeeze();
  R::clone(f, s.W.p, n, s.Wc.p), R::clone(f, s.E.p, n, s.Ec.p);
  Scalar one{};
  one[0] = 1;
  const R::Proof batch = R::prove_batch(f, {eval_W, ev[1]}, {ell, ell}, {s.Wc.p, s.Ec.p}, {ry1.data(), r_x.data()}, {one, rho}, &standin_transcript, tr.state);
  tr.absorb(batch.claims.data(), batch.claims.size());
  const Scalar c = tr.squeeze();
  R::lincomb_powers(f, {s.W.p, s.E.p}, {n, n}, c, n, s.wj.p);
  s.put("spartan.outer.polys", outer.polys.data(), outer.polys.size()), s.put("spartan.outer.r", outer.r.data(), outer.r.size());
  s.put("spartan.outer.claims", outer.claims.data(), outer.claims.size());
  s.put("spartan.inner.polys", inner.polys.data(), inner.polys.size()), s.put("spartan.inner.r", inner.r.data(), inner.r.size());
  s.put("spartan.inner.claims", inner.claims.data(), inner.claims.size());
  s.put("spartan.batch.polys", batch.polys.data(), batch.polys.size()), s.put("spartan.batch.r", batch.r.data(), batch.r.size());
  s.put("spartan.batch.claims", batch.claims.data(), batch.claims.size());
  uint8_t evs[96];
  memcpy(evs, ev[0].data(), 32), memcpy(evs + 32, ev[1].data(), 32), memcpy(evs + 64, eval_W.data(), 32);
  s.put("spartan.evaluations", evs, 96);
  (void)with_ee;
  s.ee_point = scalars_of(batch.r);   // EE::prove: hat_P = the batched witness, point = the batch sum-check's r (snark.rs:236-244)
  memcpy(s.ee_tr, tr.state, sizeof s.ee_tr);
}
static void ee_prove(Side& s) {
  const int f = s.fid;
  const size_t n = s.n, ell = s.ell;
  Transcript tr(0);
  memcpy(tr.state, s.ee_tr, sizeof s.ee_tr);  // the SNARK's transcript goes on (snark.rs:236: `&mut transcript`)
  const std::vector<Scalar>& point = s.ee_point;
  std::vector<const void*> polys{s.wj.p};
  std::vector<size_t> lens{n};
  char* cur = (char*)s.chain.p;
  const void* prev = s.wj.p;
  size_t len = n;
  std::vector<Scalar> xs;
  std::vector<void*> fold_out;
  for (size_t i = 0; i + 1 < ell; i++) {
    len /= 2;
    xs.push_back(point[ell - i - 1]), fold_out.push_back(cur);
    polys.push_back(cur), lens.push_back(len);
    cur += len * 32;
  }
  (void)prev;
  if (!xs.empty()) R::fold_chain(f, s.wj.p, n, xs, fold_out);  // the loop hyperkzg.rs:1085-1095 as one call
  const std::vector<provider::Point> com = R::batch_commit(*s.ck, std::vector<const void*>(polys.begin() + 1, polys.end()), std::vector<size_t>(lens.begin() + 1, lens.end()));
  for (const provider::Point& q : com) tr.absorb(q.xy.data(), 64);
  const U256 er = from_scalar(tr.squeeze());
  const U256 zero256{{0, 0, 0, 0}};
  const std::vector<Scalar> us{to_scalar(er), to_scalar(geq(zero256, er) ? zero256 : sub(s.p, er)), to_scalar(mulmod(er, er, s.p))};
  const std::vector<Scalar> v = R::poly_eval_multi(f, polys, lens, us);
  tr.absorb(v[0].data(), 32 * v.size());
  const Scalar q = tr.squeeze();
  R::lincomb_powers(f, polys, lens, q, n, s.Bp.p);
  std::vector<const void*> hv;
  for (int j = 0; j < 3; j++) {
    R::suffix_horner(f, s.Bp.p, n, us[j], s.hs[j].p);
    hv.push_back((const char*)s.hs[j].p + 32);  // h = out[1..n)
  }
  const std::vector<provider::Point> w = R::batch_commit(*s.ck, hv, {n - 1, n - 1, n - 1});
  Blob cb(65 * com.size()), wb(65 * 3);
  for (size_t i = 0; i < com.size(); i++) memcpy(cb.data() + 65 * i, com[i].xy.data(), 64), cb[65 * i + 64] = com[i].is_inf;
  for (size_t i = 0; i < 3; i++) memcpy(wb.data() + 65 * i, w[i].xy.data(), 64), wb[65 * i + 64] = w[i].is_inf;
  s.put("ee.com", cb.data(), cb.size()), s.put("ee.v", v[0].data(), 32 * v.size()), s.put("ee.w", wb.data(), wb.size());
}

// EE::prove of the inner-product engine on the secondary's batched witness (bench.py ipa_sequence; ipa_pc.rs:69-82, 174-281)
static void ee_ipa(Side& s) {
  const int f = s.fid;
  const size_t n = s.n;
  Transcript tr(0);
  memcpy(tr.state, s.ee_tr, sizeof s.ee_tr);  // the SNARK's transcript goes on (snark.rs:236: `&mut transcript`)
  R::eq_evals(f, s.ee_point, s.erx.p);        // b = EqPolynomial::new(point).evals() (:78)
  const Scalar r0 = tr.squeeze();
  s.put("ee.tr_state", tr.state, sizeof tr.state);
  const provider::Affine ckc = R::scale_point(*s.ck, s.ipa_u, r0);  // ck_c.scale(&r) (:190-191)
  const R::IpaProof pr = R::ipa_prove(*s.ck, ckc, s.wj.p, s.erx.p, n, &standin_ipa_transcript, tr.state);
  s.put("ee.ck_c", ckc.data(), 64);
  s.put("ee.L", pr.L.data(), pr.L.size()), s.put("ee.R", pr.R.data(), pr.R.size()), s.put("ee.inf", pr.inf.data(), pr.inf.size());
  s.put("ee.a_hat", pr.a_hat.data(), 32);
}

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s <instance> <output> <steps> <warmup>\n", argv[0]);
    return 2;
  }
  const int steps = atoi(argv[3]), warmup = atoi(argv[4]);
  const bool serial_snarks = argc > 5 && atoi(argv[5]) != 0;
  try {
    provider::check(nmx_init(0));
    const auto in = read_records(argv[1]);
    Side P, S;
    load_side(P, in, "P"), load_side(S, in, "S");
    // wall time per group, summed over the timed steps: fold S, fold P, Spartan P, EE P, Spartan S (every group ends in a synchronous
    // call -- a commitment, an evaluation, the batched witness -- so its work is complete when its clock stops)
    double grp[6] = {0, 0, 0, 0, 0, 0};
    bool timing = false;
    auto lap = [&](int g, std::chrono::steady_clock::time_point& t) {
      const auto now = std::chrono::steady_clock::now();
      if (timing) grp[g] += std::chrono::duration<double, std::milli>(now - t).count();
      t = now;
    };
    auto run = [&] {
      Scalar uS, XS, uP, XP;
      auto t = std::chrono::steady_clock::now();
      relaxed_fold(S, uS, XS);            // nova/mod.rs:812-826
      provider::check(nmx_sync());
      lap(0, t);
      relaxed_fold(P, uP, XP);            // :829-843
      provider::check(nmx_sync());
      lap(1, t);
      if (serial_snarks) {
        spartan_prove(P, uP, XP, false);  // S1::prove (:863-871): the sum-check sequence ...
        lap(2, t);
        ee_prove(P);                      // ... and EE::prove on its batched witness (snark.rs:236-244)
        lap(3, t);
        spartan_prove(S, uS, XS, false);  // S2::prove (:872-880) ...
        provider::check(nmx_sync());
        lap(4, t);
        ee_ipa(S);                        // ... and its evaluation argument, the inner-product argument
        lap(5, t);
      } else {                            // rayon::join(S1::prove, S2::prove), nova/mod.rs:862-881: S2 on a second host thread
        std::exception_ptr err;
        std::thread th([&] {
          try {
            spartan_prove(S, uS, XS, false);
            provider::check(nmx_sync());
            ee_ipa(S);
          } catch (...) {
            err = std::current_exception();
          }
        });
        try {
          spartan_prove(P, uP, XP, false);
          lap(2, t);
          ee_prove(P);
          lap(3, t);
        } catch (...) {
          th.join();
          throw;
        }
        th.join();
        if (err) std::rethrow_exception(err);
        lap(4, t);                        // (what S2 still needed after S1 was done)
      }
    };
    for (int i = 0; i < warmup; i++) run();
    timing = true;
    HIPOK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<double> step_ms;
    for (int i = 0; i < steps; i++) {
      const auto ts = std::chrono::steady_clock::now();
      run();
      step_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count());
    }
    HIPOK(hipDeviceSynchronize());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (steps > 0 ? steps : 1);
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 2;
    write_record(f, "ms_per_sequence", &ms, 8);
    write_record(f, "ms_per_step", step_ms.data(), step_ms.size() * 8);  // every timed sequence on its own
    for (double& g : grp) g /= (steps > 0 ? steps : 1);
    write_record(f, "ms_per_group", grp, sizeof grp);  // fold S, fold P, Spartan P, EE P, Spartan S (parallel: what S2 still needed), EE S (serial only)
    for (Side* s : {&P, &S}) {
      for (const auto& kv : s->out) write_record(f, kv.first, kv.second.data(), kv.second.size());
      const Blob wj = s->wj.download();
      write_record(f, s->tag + ".spartan.batch_witness", wj.data(), wj.size());
    }
    fclose(f);
    printf("csnark_replay: %.4f ms per sequence over %d steps\n", ms, steps);
    return 0;
  } catch (const provider::Error& e) {
    fprintf(stderr, "%s\n", e.what());
    return e.code == NMX_E_NO_DEVICE ? 3 : 1;
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
  }
}
