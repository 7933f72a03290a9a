// keyfile.hpp -- on-disk commitment keys straight into HBM.  Included by capi.hip only.
//
//   .ptau (snarkjs / Perpetual Powers of Tau)   /root/reference/src/provider/ptau.rs:153-436
//       "ptau" | u32 version = 1 | u32 num_sections in {11, 3} | sections: (u32 id, i64 size, payload)
//       section 1: u32 n8 | n8 bytes prime (LE) | u32 power      section 2: tauG1 points      section 3: tauG2 points
//   PEDERSEN_KEY                                 /root/reference/src/provider/pedersen.rs:28,318-340,383-393
//       12-byte head | h | ck[0..n)
// A point is halo2curves' `write_raw` record: x || y, each 4 x u64 raw R = 2^256 Montgomery limbs, little-endian.
// That is the NMX_BASES_MONT layout, so the payload is never touched on the host: it goes file -> pinned staging
// (two buffers, read and hipMemcpyAsync overlapped) -> HBM, and the canonicity / on-curve checks of `read_points`
// (ptau.rs:372-391) run on the device (ValidateFn).
#pragma once
#include <stdio.h>
#include <string>
#include "runtime.hpp"

namespace nmx {

struct FileCloser {
  FILE* f;
  ~FileCloser() {
    if (f) fclose(f);
  }
};

static void read_exact(FILE* f, void* dst, size_t bytes, const char* what) {
  if (bytes && fread(dst, 1, bytes, f) != bytes) throw Fail{NMX_E_IO, std::string("IoError: short read (") + what + ")"};
}
static uint32_t read_u32(FILE* f, const char* what) {
  uint8_t b[4];
  read_exact(f, b, 4, what);
  return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
}
static int64_t read_i64(FILE* f, const char* what) {
  uint8_t b[8];
  read_exact(f, b, 8, what);
  uint64_t v = 0;
  for (int i = 7; i >= 0; i--) v = (v << 8) | b[i];
  return (int64_t)v;
}
static void seek_to(FILE* f, int64_t pos) {
  if (pos < 0 || fseeko(f, (off_t)pos, SEEK_SET) != 0) throw Fail{NMX_E_IO, "IoError: seek failed"};
}

struct PtauMeta {
  int64_t pos_header = 0, pos_tau_g1 = 0, pos_tau_g2 = 0;
};

// read_meta_data (ptau.rs:270-327)
static PtauMeta ptau_read_meta(FILE* f) {
  char magic[4];
  read_exact(f, magic, 4, "magic");
  require(memcmp(magic, "ptau", 4) == 0, NMX_E_FORMAT, "InvalidHead");
  const uint32_t version = read_u32(f, "version");
  require(version == 1, NMX_E_FORMAT, "UnsupportedVersion");
  const uint32_t num_sections = read_u32(f, "num_sections");
  require(num_sections == 11 || num_sections == 3, NMX_E_FORMAT, "InvalidNumSections");  // full / pruned files
  PtauMeta m;
  for (uint32_t s = 0; s < num_sections; s++) {
    const uint32_t id = read_u32(f, "section id");
    const int64_t size = read_i64(f, "section size");
    const int64_t pos = (int64_t)ftello(f);
    if (id == 1) m.pos_header = pos;
    if (id == 2) m.pos_tau_g1 = pos;
    if (id == 3) m.pos_tau_g2 = pos;
    require(size >= 0, NMX_E_FORMAT, "negative section size");
    seek_to(f, pos + size);
  }
  // assert_ne!(pos_header, 0) etc. (ptau.rs:318-320)
  require(m.pos_header && m.pos_tau_g1 && m.pos_tau_g2, NMX_E_FORMAT, "header / tauG1 / tauG2 section missing");
  return m;
}

// read_header (ptau.rs:329-370)
static void ptau_read_header(FILE* f, const uint32_t* modulus_words, size_t num_g1, size_t num_g2) {
  const uint32_t n8 = read_u32(f, "n8");
  require(n8 <= 4096, NMX_E_FORMAT, "InvalidPrime");
  std::vector<uint8_t> prime(n8 ? n8 : 1);
  read_exact(f, prime.data(), n8, "prime");
  uint8_t expect[32];
  memcpy(expect, modulus_words, 32);
  bool same = n8 >= 32 && memcmp(prime.data(), expect, 32) == 0;
  for (uint32_t i = 32; same && i < n8; i++) same = prime[i] == 0;  // compared as integers (BigUint)
  require(same, NMX_E_FORMAT, "InvalidPrime");
  const uint32_t power = read_u32(f, "power");
  require(power < 40, NMX_E_FORMAT, "power out of range");
  const uint64_t max_num_g2 = 1ull << power, max_num_g1 = max_num_g2 * 2 - 1;
  require(num_g1 <= max_num_g1, NMX_E_FORMAT, "InsufficientPowerForG1");
  require(num_g2 <= max_num_g2, NMX_E_FORMAT, "InsufficientPowerForG2");
}

// file (positioned at the first point) -> device, n points, through two pinned staging buffers
static BaseFill file_fill(FILE* f, size_t n) {
  return [f, n](void* d_dst, hipStream_t stream) {
    const size_t chunk = (size_t)16 << 20;  // 16 MiB = 2^18 points per copy
    struct Pinned {
      void* p[2] = {nullptr, nullptr};
      hipEvent_t done[2];
      bool have_ev = false;
      ~Pinned() {
        for (int i = 0; i < 2; i++)
          if (p[i]) (void)hipHostFree(p[i]);
        if (have_ev)
          for (int i = 0; i < 2; i++) (void)hipEventDestroy(done[i]);
      }
    } st;
    const size_t total = n * 64;
    const size_t sz = total < chunk ? total : chunk;
    for (int i = 0; i < 2; i++) HIPCHK(hipHostMalloc(&st.p[i], sz ? sz : 64, hipHostMallocDefault));
    for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&st.done[i], hipEventDisableTiming));
    st.have_ev = true;
    size_t off = 0;
    bool used[2] = {false, false};
    for (int b = 0; off < total; b ^= 1) {
      const size_t len = total - off < sz ? total - off : sz;
      if (used[b]) HIPCHK(hipEventSynchronize(st.done[b]));  // the copy out of this buffer has finished
      read_exact(f, st.p[b], len, "points");                  // overlaps the other buffer's copy
      HIPCHK(hipMemcpyAsync((char*)d_dst + off, st.p[b], len, hipMemcpyHostToDevice, stream));
      HIPCHK(hipEventRecord(st.done[b], stream));
      used[b] = true;
      off += len;
    }
    HIPCHK(hipStreamSynchronize(stream));  // staging buffers are freed on return
  };
}

}  // namespace nmx
