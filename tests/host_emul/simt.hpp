// tests/host_emul/simt.hpp -- TEST-ONLY: runs block-level GPU kernels (LDS + __syncthreads + atomics) on the CPU.
//
// One ucontext fiber per thread of a block, blocks one after another.  A fiber runs until it calls syncthreads() (it
// parks; when every live fiber of the block has parked, all are released) or returns.  `static` arrays inside a kernel
// play the role of LDS (one block at a time); atomics are plain read-modify-writes.  Enough to debug indexing, barrier
// placement and counting logic of nova_amd/csrc/msm_partition.hpp without a GPU; it says nothing about performance,
// memory ordering between blocks (blocks never overlap here) or wave-level intrinsics (not supported).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <ucontext.h>

#include <functional>
#include <stdexcept>
#include <vector>

namespace simt {

struct Fiber {
  ucontext_t ctx;
  unsigned tid = 0;
  enum { kReady, kParked, kDone } st = kReady;
};
struct Sched {
  ucontext_t main;
  std::vector<Fiber> fibers;
  std::vector<char*> stacks;
  Fiber* cur = nullptr;
  unsigned bid = 0, bdim = 0, gdim = 0;
  const std::function<void()>* body = nullptr;
  ~Sched() {
    for (char* s : stacks) free(s);
  }
};
inline Sched& S() {
  static Sched s;
  return s;
}
inline unsigned tid() { return S().cur->tid; }
inline unsigned bid() { return S().bid; }
inline unsigned bdim() { return S().bdim; }
inline unsigned gdim() { return S().gdim; }
inline void syncthreads() {
  Sched& s = S();
  Fiber* f = s.cur;
  f->st = Fiber::kParked;
  swapcontext(&f->ctx, &s.main);
}
inline void trampoline() {
  Sched& s = S();
  (*s.body)();
  s.cur->st = Fiber::kDone;
  swapcontext(&s.cur->ctx, &s.main);
}

static constexpr size_t kStack = 96 << 10;

// run body() as `grid` blocks of `block` threads
inline void launch(unsigned grid, unsigned block, const std::function<void()>& body) {
  Sched& s = S();
  if (s.cur) throw std::runtime_error("simt: nested launch");
  s.body = &body;
  s.bdim = block;
  s.gdim = grid;
  if (s.fibers.size() < block) s.fibers.resize(block);
  while (s.stacks.size() < block) s.stacks.push_back((char*)malloc(kStack));
  for (unsigned b = 0; b < grid; b++) {
    s.bid = b;
    for (unsigned t = 0; t < block; t++) {
      Fiber& f = s.fibers[t];
      f.tid = t;
      f.st = Fiber::kReady;
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = s.stacks[t];
      f.ctx.uc_stack.ss_size = kStack;
      f.ctx.uc_link = nullptr;
      makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    for (;;) {
      unsigned live = 0, parked = 0;
      for (unsigned t = 0; t < block; t++) {
        Fiber& f = s.fibers[t];
        if (f.st == Fiber::kDone) continue;
        if (f.st == Fiber::kReady) {
          s.cur = &f;
          swapcontext(&s.main, &f.ctx);
          s.cur = nullptr;
        }
        if (f.st != Fiber::kDone) live++;
        if (f.st == Fiber::kParked) parked++;
      }
      if (live == 0) break;
      if (parked != live) throw std::runtime_error("simt: scheduler inconsistency");
      // a barrier some threads skipped by returning early is undefined behaviour on the GPU: make it loud here
      for (unsigned t = 0; t < block; t++)
        if (s.fibers[t].st == Fiber::kDone && live) {
          bool any_parked = false;
          for (unsigned u = 0; u < block; u++) any_parked |= s.fibers[u].st == Fiber::kParked;
          if (any_parked) throw std::runtime_error("simt: __syncthreads() reached by only part of the block");
        }
      for (unsigned t = 0; t < block; t++)
        if (s.fibers[t].st == Fiber::kParked) s.fibers[t].st = Fiber::kReady;
    }
  }
  s.body = nullptr;
}

}  // namespace simt
