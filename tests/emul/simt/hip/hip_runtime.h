// hip_runtime.h (SIMT emulation shim) — TEST INFRASTRUCTURE ONLY.
//
// Lets g++ (with -fsanitize=address,undefined) compile and run the GPU-only code of cave_amd/csrc -- the one-wave
// "lite" Newton solver (cone_core.h), the one-/two-wave band elimination (cone_band.h), the wave and workgroup
// contexts (ctx_wave.h, ctx_block.h) and the DPP / readlane / ballot primitives under them (wave_prims.h) --
// exactly as written, on a machine without a GPU.  A workgroup is NT fibers (ucontext) on one OS thread; every
// cross-lane primitive and every barrier is a rendezvous of the lanes of a wave (or of the workgroup).  Between two
// rendezvous the lanes run one after another in an order that can be shuffled (seeded), so code that relies on an
// ordering the source does not state -- a missing barrier between two waves, an LDS hand-over inside a wave without
// CAVE_WAVE_ORDER() -- computes wrong numbers here instead of "usually working".  Nothing in cave_amd loads this.
//
// Found through -Itests/emul/simt by wave_prims.h's `#include <hip/hip_runtime.h>` when CAVE_SIMT_EMUL is defined.
#pragma once
#ifndef CAVE_SIMT_EMUL
#error "this shim is for the CAVE_SIMT_EMUL test build only"
#endif
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#if defined(__SANITIZE_ADDRESS__)
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
#endif

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)

// Context switch without the signal-mask system calls of swapcontext (two per switch: the emulation makes ~10^5
// switches per Newton iteration).  x86-64 System V: callee-saved registers on the outgoing stack, swap stack pointers.
#if defined(__x86_64__)
#define SIMT_ASM_SWITCH 1
extern "C" void simt_switch_stack(void** save_sp, void* next_sp);
asm(R"(
.text
.p2align 4
.globl simt_switch_stack
.hidden simt_switch_stack
.type simt_switch_stack,@function
simt_switch_stack:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size simt_switch_stack, .-simt_switch_stack
)");
#endif

namespace simt {

struct Dim3 { unsigned x, y, z; };

struct Sim {
  static constexpr int MAXT = 512, MAXW = MAXT / 64;
  static constexpr size_t STACK = 1u << 20;
  int nt = 0;
  int cur = 0;
  ucontext_t main_ctx;
  ucontext_t ctx[MAXT];
  void* sp[MAXT + 1] = {nullptr};  // saved stack pointers (asm switch; [MAXT]: the launcher)
  char* stack[MAXT] = {nullptr};
  bool done[MAXT];
  struct Bar { int count = 0; unsigned gen = 0; };
  Bar wbar[MAXW];
  Bar bbar;
  uint64_t xch[MAXW][2][64];
  unsigned char par[MAXT];
  unsigned long progress = 0;
  Dim3 block_idx{0, 0, 0}, grid_dim{1, 1, 1}, block_dim{64, 1, 1};
  std::function<void()> body;
  uint64_t rng = 0x9e3779b97f4a7c15ull;
  bool shuffle = false;
  int order[MAXT];
  void* fake_stack[MAXT + 1] = {nullptr};
  const void* main_bottom = nullptr;
  size_t main_size = 0;
  unsigned long switches = 0, rendezvous = 0;
};

inline Sim& S() {
  static Sim s;
  return s;
}

inline void switch_to(int from, int to) {  // from / to: fiber index, or -1 for the main context
  Sim& s = S();
  ucontext_t* a = from < 0 ? &s.main_ctx : &s.ctx[from];
  ucontext_t* b = to < 0 ? &s.main_ctx : &s.ctx[to];
  s.switches++;
#if defined(__SANITIZE_ADDRESS__)
  void** save = &s.fake_stack[from < 0 ? Sim::MAXT : from];
  if (to < 0) __sanitizer_start_switch_fiber(from >= 0 && s.done[from] ? nullptr : save, s.main_bottom, s.main_size);
  else __sanitizer_start_switch_fiber(from >= 0 && s.done[from] ? nullptr : save, s.stack[to], Sim::STACK);
#endif
  s.cur = to;
#if defined(SIMT_ASM_SWITCH)
  (void)a; (void)b;
  simt_switch_stack(&s.sp[from < 0 ? Sim::MAXT : from], s.sp[to < 0 ? Sim::MAXT : to]);
#else
  swapcontext(a, b);
#endif
#if defined(__SANITIZE_ADDRESS__)
  const void* ob = nullptr;
  size_t os = 0;
  __sanitizer_finish_switch_fiber(s.fake_stack[from < 0 ? Sim::MAXT : from], &ob, &os);
  if (from < 0 && false) { (void)ob; (void)os; }
#endif
}

// give the processor to the next lane that has not finished (round robin, or a seeded random pick)
inline void yield() {
  Sim& s = S();
  const int me = s.cur;
  int nxt = me;
  if (s.shuffle) {
    s.rng ^= s.rng << 13; s.rng ^= s.rng >> 7; s.rng ^= s.rng << 17;
    int start = (int)(s.rng % (uint64_t)s.nt);
    for (int k = 0; k < s.nt; ++k) {
      int c = (start + k) % s.nt;
      if (!s.done[c] && c != me) { nxt = c; break; }
    }
  } else {
    for (int k = 1; k <= s.nt; ++k) {
      int c = (me + k) % s.nt;
      if (!s.done[c]) { nxt = c; break; }
    }
  }
  if (nxt != me) switch_to(me, nxt);
}

[[noreturn]] inline void deadlock(const char* what) {
  Sim& s = S();
  fprintf(stderr, "simt: DEADLOCK in %s (lane %d of %d; block %u): a rendezvous was not reached by every lane -- a "
                  "cross-lane primitive or barrier inside divergent control flow, or mismatched barrier counts\n",
          what, s.cur, s.nt, s.block_idx.x);
  abort();
}

inline void rendezvous(Sim::Bar& b, int size, const char* what) {
  Sim& s = S();
  const unsigned my = b.gen;
  if (++b.count == size) {
    b.count = 0;
    b.gen++;
    s.progress++;
    s.rendezvous++;
    return;
  }
  unsigned long seen = s.progress;
  long idle = 0;
  while (b.gen == my) {
    yield();
    if (s.progress != seen) { seen = s.progress; idle = 0; }
    else if (++idle > 8L * s.nt + 64) deadlock(what);
  }
}

inline int tid() { return S().cur; }
inline int lane() { return S().cur & 63; }
inline int wave() { return S().cur >> 6; }
inline int wave_size() {  // lanes of this wave that exist (the last wave of a short workgroup may be partial)
  Sim& s = S();
  const int w = s.cur >> 6;
  const int n = s.nt - 64 * w;
  return n > 64 ? 64 : n;
}
inline void wave_sync() { rendezvous(S().wbar[wave()], wave_size(), "wave_sync"); }
inline void block_sync() { rendezvous(S().bbar, S().nt, "__syncthreads / s_barrier"); }

// every lane publishes v; returns the table of the wave (valid until this lane's next-but-one exchange)
inline const uint64_t* exchange(uint64_t v) {
  Sim& s = S();
  const int me = s.cur, w = me >> 6, l = me & 63;
  const int p = s.par[me];
  s.par[me] ^= 1;
  s.xch[w][p][l] = v;
  rendezvous(s.wbar[w], wave_size(), "cross-lane exchange");
  return s.xch[w][p];
}

inline void fiber_main() {
  Sim& s = S();
#if defined(__SANITIZE_ADDRESS__)
  __sanitizer_finish_switch_fiber(nullptr, &s.main_bottom, &s.main_size);
#endif
  s.body();
  const int me = s.cur;
  s.done[me] = true;
  s.progress++;
  // hand over to any unfinished lane, else back to the launcher
  for (int k = 1; k <= s.nt; ++k) {
    int c = (me + k) % s.nt;
    if (!s.done[c]) { switch_to(me, c); }
  }
  switch_to(me, -1);
  abort();  // never resumed
}

// run one workgroup of nt lanes; body() is what every lane executes
inline void run_block(int nt, unsigned block_idx, unsigned grid_dim, std::function<void()> body, uint64_t seed = 0) {
  Sim& s = S();
  if (nt <= 0 || nt > Sim::MAXT) abort();
  s.nt = nt;
  s.block_idx = Dim3{block_idx, 0, 0};
  s.grid_dim = Dim3{grid_dim, 1, 1};
  s.block_dim = Dim3{(unsigned)nt, 1, 1};
  s.body = std::move(body);
  s.shuffle = seed != 0;
  s.rng = seed ? seed * 0x9e3779b97f4a7c15ull + 1 : 1;
  s.bbar = Sim::Bar{};
  for (int w = 0; w < Sim::MAXW; ++w) s.wbar[w] = Sim::Bar{};
  for (int t = 0; t < nt; ++t) {
    if (!s.stack[t]) {
      s.stack[t] = (char*)mmap(nullptr, Sim::STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (s.stack[t] == MAP_FAILED) abort();
    }
    s.done[t] = false;
    s.par[t] = 0;
#if defined(__SANITIZE_ADDRESS__)
    // a lane that finished never returned from its frames (it switched away for good): their redzones are still
    // poisoned, and the next workgroup's frames on this stack would trip over them
    __asan_unpoison_memory_region(s.stack[t], Sim::STACK);
#endif
#if defined(SIMT_ASM_SWITCH)
    {  // a fresh stack that "returns" into fiber_main: [6 callee-saved registers][entry][alignment slot]
      void** top = (void**)(((uintptr_t)s.stack[t] + Sim::STACK) & ~(uintptr_t)15);
      top[-1] = nullptr;
      top[-2] = (void*)&fiber_main;
      for (int i = 3; i <= 8; ++i) top[-i] = nullptr;
      s.sp[t] = (void*)(top - 8);
    }
#else
    getcontext(&s.ctx[t]);
    s.ctx[t].uc_stack.ss_sp = s.stack[t];
    s.ctx[t].uc_stack.ss_size = Sim::STACK;
    s.ctx[t].uc_link = nullptr;
    makecontext(&s.ctx[t], (void (*)())fiber_main, 0);
#endif
  }
  switch_to(-1, 0);
  for (int t = 0; t < nt; ++t)
    if (!s.done[t]) deadlock("kernel exit (a lane never finished)");
}

// ---- DPP source lane (gfx9 dpp_ctrl encodings used by wave_prims.h / cone_core.h); -1 = no source
inline int dpp_src(int l, int ctrl) {
  const int row = l & ~15, r = l & 15;
  if (ctrl >= 0 && ctrl <= 0xff) return (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);           // quad_perm
  if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = ctrl & 15; return r + n <= 15 ? l + n : -1; }  // row_shl
  if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl & 15; return r >= n ? l - n : -1; }       // row_shr
  if (ctrl >= 0x121 && ctrl <= 0x12f) { const int n = ctrl & 15; return row | ((r - n) & 15); }      // row_ror
  if (ctrl == 0x130) return l < 63 ? l + 1 : -1;   // wave_shl:1
  if (ctrl == 0x134) return (l + 1) & 63;          // wave_rol:1
  if (ctrl == 0x138) return l > 0 ? l - 1 : -1;    // wave_shr:1
  if (ctrl == 0x13c) return (l - 1) & 63;          // wave_ror:1
  if (ctrl == 0x140) return row | (15 - r);        // row_mirror
  if (ctrl == 0x141) return row | (r < 8 ? 7 - r : 23 - r);  // row_half_mirror
  if (ctrl == 0x142) return l >= 16 ? row - 1 : -1;          // row_bcast:15 (lane 15 of the previous row)
  if (ctrl == 0x143) return l >= 32 ? 31 : -1;               // row_bcast:31
  fprintf(stderr, "simt: dpp_ctrl 0x%x not emulated\n", ctrl);
  abort();
}

inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int l = lane();
  const uint64_t* t = exchange((uint32_t)src);
  const bool enabled = ((row_mask >> (l >> 4)) & 1) && ((bank_mask >> ((l & 15) >> 2)) & 1);
  const int sl = dpp_src(l, ctrl);
  int out = old;
  if (enabled) {
    if (sl >= 0 && sl < wave_size()) out = (int)(uint32_t)t[sl];
    else if (bound_ctrl) out = 0;
  }
  return out;
}
inline int readlane(int v, int l) { return (int)(uint32_t)exchange((uint32_t)v)[l & 63]; }
inline uint64_t ballot(bool p) {
  const uint64_t* t = exchange(p ? 1u : 0u);
  uint64_t m = 0;
  const int n = wave_size();
  for (int i = 0; i < n; ++i) m |= (t[i] & 1ull) << i;
  return m;
}
inline uint32_t mbcnt_lo(uint32_t mask, uint32_t base) {
  const int l = lane();
  const uint32_t below = l >= 32 ? 0xffffffffu : ((1u << l) - 1u);
  return base + (uint32_t)__builtin_popcount(mask & below);
}
inline uint32_t mbcnt_hi(uint32_t mask, uint32_t base) {
  const int l = lane();
  const uint32_t below = l <= 32 ? 0u : ((l - 32) >= 32 ? 0xffffffffu : ((1u << (l - 32)) - 1u));
  return base + (uint32_t)__builtin_popcount(mask & below);
}

template <class T> inline T atomic_add(T* p, T v) { T o = *p; *p = o + v; return o; }

}  // namespace simt

// ---- HIP surface used by cave_amd/csrc
struct float4 { float x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
#define threadIdx (::simt::Dim3{(unsigned)::simt::tid(), 0u, 0u})
#define blockIdx (::simt::S().block_idx)
#define gridDim (::simt::S().grid_dim)
#define blockDim (::simt::S().block_dim)
static inline void __syncthreads() { ::simt::block_sync(); }
static inline uint64_t __ballot(bool p) { return ::simt::ballot(p); }
static inline int __popcll(uint64_t x) { return __builtin_popcountll(x); }
static inline double __hiloint2double(int hi, int lo) {
  uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
  double d;
  memcpy(&d, &u, 8);
  return d;
}
static inline int __double2hiint(double d) { uint64_t u; memcpy(&u, &d, 8); return (int)(uint32_t)(u >> 32); }
static inline int __double2loint(double d) { uint64_t u; memcpy(&u, &d, 8); return (int)(uint32_t)u; }
static inline double atomicAdd(double* p, double v) { return ::simt::atomic_add(p, v); }
static inline float atomicAdd(float* p, float v) { return ::simt::atomic_add(p, v); }
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { return ::simt::atomic_add(p, v); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return ::simt::atomic_add(p, v); }
static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
template <class P, class T> static inline T __hip_atomic_fetch_add(P p, T v, int, int) { T o = *p; *p = o + v; return o; }

#define __builtin_amdgcn_readlane(v, l) ::simt::readlane((int)(v), (int)(l))
#define __builtin_amdgcn_readfirstlane(v) ::simt::readlane((int)(v), 0)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) ::simt::update_dpp((int)(old), (int)(src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_mbcnt_lo(m, b) ::simt::mbcnt_lo((uint32_t)(m), (uint32_t)(b))
#define __builtin_amdgcn_mbcnt_hi(m, b) ::simt::mbcnt_hi((uint32_t)(m), (uint32_t)(b))
#ifdef SIMT_APPROX_RCP
#define __builtin_amdgcn_rcp(x) ((double)(1.0f / (float)(x)))  /* ~24 bits, like v_rcp_f64 before its Newton steps */
#else
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#endif
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() ::simt::wave_sync()
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_s_memrealtime() 0ull
