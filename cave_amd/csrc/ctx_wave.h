// ctx_wave.h — the SPMD context for gfx950: ONE 64-lane wavefront per instance.
//
// A workgroup is a single wave, so __syncthreads() lowers to an LDS-counter
// wait (the s_barrier is elided by hipcc under __launch_bounds__(64)); LDS
// operations of one wave execute in order, which is what makes the
// "rows in sequence, lanes over entries" loops of cone_core.h race-free.
#pragma once
#include <hip/hip_runtime.h>
#include "cone_common.h"

namespace cave {

__device__ __forceinline__ double readlane_f64(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <int U>
__device__ __forceinline__ uint32_t scan_dense_wave(const float* __restrict__ A, uint32_t n, int lane, uint32_t* eflat,
                                                    float* eval, uint32_t cap);

struct WaveCtx {
  static constexpr int NT = 64;
  static constexpr int TEAM = 4;         // lanes sharing one CSR row in gradient()
  static constexpr int SCAN_UNROLL = 8;  // 8 x 1 KiB dwordx4 loads in flight per wave
  __device__ __forceinline__ uint32_t scan_dense(const float* A, uint32_t n, uint32_t* eflat, float* eval,
                                                 uint32_t cap) const {
    return scan_dense_wave<SCAN_UNROLL>(A, n, lane, eflat, eval, cap);
  }
  static constexpr int PMAX = 64;  // largest reduced system solve_spd handles in registers
  int lane;
#ifdef CAVE_STAMPS
  unsigned long long st[16];
#endif
  __device__ __forceinline__ int tid() const { return lane; }
  __device__ __forceinline__ void sync() const { __syncthreads(); }

  // ---- wave-wide reductions / scans on the DPP crossbar (no LDS traffic, fixed summation tree)
  template <int CTRL, int RM>
  static __device__ __forceinline__ double dpp_f64(double old, double x) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, RM, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, RM, 0xf, false);
    return __hiloint2double(hi, lo);
  }
  template <int CTRL, int RM>
  static __device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, RM, 0xf, false);
  }
  __device__ __forceinline__ double reduce_sum(double v) const {
    v += dpp_f64<0xb1, 0xf>(0.0, v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4e, 0xf>(0.0, v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x114, 0xf>(0.0, v);  // row_shr:4
    v += dpp_f64<0x118, 0xf>(0.0, v);  // row_shr:8
    v += dpp_f64<0x142, 0xa>(0.0, v);  // row_bcast:15 -> rows 1,3
    v += dpp_f64<0x143, 0xc>(0.0, v);  // row_bcast:31 -> rows 2,3
    return readlane_f64(v, 63);
  }
  // sum over the TEAM = 4 lanes of a quad; every lane gets the result
  __device__ __forceinline__ double team_reduce_sum(double v) const {
    v += dpp_f64<0xb1, 0xf>(0.0, v);
    v += dpp_f64<0x4e, 0xf>(0.0, v);
    return v;
  }
  __device__ __forceinline__ double reduce_max(double v) const {
    v = fmax(v, dpp_f64<0xb1, 0xf>(v, v));
    v = fmax(v, dpp_f64<0x4e, 0xf>(v, v));
    v = fmax(v, dpp_f64<0x114, 0xf>(v, v));
    v = fmax(v, dpp_f64<0x118, 0xf>(v, v));
    v = fmax(v, dpp_f64<0x142, 0xa>(v, v));
    v = fmax(v, dpp_f64<0x143, 0xc>(v, v));
    return readlane_f64(v, 63);
  }
  __device__ __forceinline__ uint32_t reduce_add_u32(uint32_t v) const {
    v += dpp_u32<0xb1, 0xf>(0u, v);
    v += dpp_u32<0x4e, 0xf>(0u, v);
    v += dpp_u32<0x114, 0xf>(0u, v);
    v += dpp_u32<0x118, 0xf>(0u, v);
    v += dpp_u32<0x142, 0xa>(0u, v);
    v += dpp_u32<0x143, 0xc>(0u, v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
  }
  __device__ __forceinline__ void atomic_add_u32(uint32_t* p, uint32_t v) const { atomicAdd(p, v); }
  __device__ __forceinline__ void atomic_add_f64(double* p, double v) const { atomicAdd(p, v); }  // ds_add_f64

  // in-place exclusive scan of an LDS array; returns the total
  __device__ __forceinline__ uint32_t exclusive_scan_u32(uint32_t* a, int n) const {
    uint32_t carry = 0;
    for (int base = 0; base < n; base += 64) {
      int i = base + lane;
      uint32_t v = (i < n) ? a[i] : 0u;
      uint32_t inc = v;
      inc += dpp_u32<0x111, 0xf>(0u, inc);  // row_shr:1  (Hillis-Steele inside each 16-lane row)
      inc += dpp_u32<0x112, 0xf>(0u, inc);  // row_shr:2
      inc += dpp_u32<0x114, 0xf>(0u, inc);  // row_shr:4
      inc += dpp_u32<0x118, 0xf>(0u, inc);  // row_shr:8
      inc += dpp_u32<0x142, 0xa>(0u, inc);  // row_bcast:15 -> rows 1,3
      inc += dpp_u32<0x143, 0xc>(0u, inc);  // row_bcast:31 -> rows 2,3
      if (i < n) a[i] = carry + inc - v;
      carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    __syncthreads();
    return carry;
  }
  // ordered list of i in [0,n) with (flags[i] & mask) == val
  __device__ __forceinline__ uint32_t compact_mask_u8(const uint8_t* flags, int n, uint8_t mask, uint8_t val,
                                                      uint32_t* out) const {
    uint32_t cnt = 0;
    for (int base = 0; base < n; base += 64) {
      int i = base + lane;
      bool pr = (i < n) && ((flags[i] & mask) == val);
      uint64_t b = __ballot(pr);
      if (pr) out[cnt + mbcnt64(b)] = (uint32_t)i;
      cnt += (uint32_t)__popcll(b);
    }
    return cnt;
  }
  __device__ __forceinline__ uint32_t compact_nonzero_u8(const uint8_t* flags, int n, uint32_t* out) const {
    uint32_t cnt = 0;
    for (int base = 0; base < n; base += 64) {
      int i = base + lane;
      bool pr = (i < n) && (flags[i] != 0);
      uint64_t b = __ballot(pr);
      if (pr) out[cnt + mbcnt64(b)] = (uint32_t)i;
      cnt += (uint32_t)__popcll(b);
    }
    return cnt;
  }

  // Solve the p x p system whose free rows are rows of (H + delta*I) and whose
  // fixed rows (act) are identity rows:   H_FF x_F + H_FA x_A = rhs_F,  x_A = rhs_A.
  // Lane i keeps row i of [H | rhs] in registers; Gauss-Jordan elimination,
  // pivot row broadcast with v_readlane (no LDS traffic, no barriers).
  template <int PM>
  __device__ __forceinline__ void solve_spd_regs(const double* H, int ldh, const double* rhs, const uint8_t* act, int p,
                                                 double reg_rel, double* dv) const {
    const bool live = lane < p;
    const bool my_act = live && act[lane] != 0;
    double diag0 = (live && !my_act) ? H[lane * ldh + lane] : 0.0;
    const double maxdiag = reduce_max(diag0);
    const double reg = reg_rel * maxdiag;
    double h[PM];
#pragma unroll
    for (int j = 0; j < PM; ++j) {
      double v = 0.0;
      if (j < p) {
        if (live && !my_act) v = H[lane * ldh + j];
        if (j == lane) v = my_act ? 1.0 : v + reg;
      }
      h[j] = v;
    }
    double b = live ? rhs[lane] : 0.0;
    double diag = 1.0;
    bool dead = !live;
    const double thresh = 1e-300;
#pragma unroll
    for (int k = 0; k < PM; ++k) {
      if (k < p) {
        double piv = readlane_f64(h[k], k);
        double bk = readlane_f64(b, k);
        if (piv > thresh) {  // uniform
          if (lane == k) diag = piv;
          double fac = (lane == k) ? 0.0 : h[k] / piv;
#pragma unroll
          for (int j = k + 1; j < PM; ++j) h[j] -= fac * readlane_f64(h[j], k);  // columns >= p hold zeros
          b -= fac * bk;
        } else if (lane == k) {
          dead = true;
        }
      }
    }
    if (live) dv[lane] = dead ? 0.0 : b / diag;
  }

  __device__ __forceinline__ void solve_spd(const double* H, int ldh, const double* g, const uint8_t* act, int p,
                                            double reg_rel, double* dv) const {
    if (p <= 8) solve_spd_regs<8>(H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 16) solve_spd_regs<16>(H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 20) solve_spd_regs<20>(H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 24) solve_spd_regs<24>(H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 28) solve_spd_regs<28>(H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 32) solve_spd_regs<32>(H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 40) solve_spd_regs<40>(H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 48) solve_spd_regs<48>(H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 56) solve_spd_regs<56>(H, ldh, g, act, p, reg_rel, dv);
    else solve_spd_regs<64>(H, ldh, g, act, p, reg_rel, dv);
  }
};

// Stream one dense instance (n floats, row-major) and append its non-zeros, in flat (row-major)
// order, as (flat index, value) pairs to (eflat, eval).  Returns the number of non-zeros seen
// (entries beyond `cap` are counted, not stored; eflat/eval must have cap + 64 slots, the last 64
// are per-lane dump slots).  Row/column are derived afterwards, once per
// entry (cone_instance.h finish_scan), so the per-KiB streaming loop stays short: 4 compares,
// a wave-uniform "anything here?" test, ordered slot computation with mbcnt, two LDS stores.
__device__ __forceinline__ void scan_chunk1(float v, bool valid, uint32_t f, uint32_t& cursor, uint32_t* eflat,
                                            float* eval, uint32_t cap) {
  bool nz = valid && (v != 0.0f);
  uint64_t m = __ballot(nz);
  if (m == 0ull) return;
  uint32_t pos = cursor + mbcnt64(m);
  if (nz && pos < cap) { eflat[pos] = f; eval[pos] = v; }
  cursor += (uint32_t)__popcll(m);
}

__device__ __forceinline__ void scan_chunk4(float4 v, uint32_t f, uint32_t& cursor, uint32_t* eflat, float* eval,
                                            uint32_t cap) {
  bool n0 = v.x != 0.0f, n1 = v.y != 0.0f, n2 = v.z != 0.0f, n3 = v.w != 0.0f;
  uint64_t m0 = __ballot(n0), m1 = __ballot(n1), m2 = __ballot(n2), m3 = __ballot(n3);
  if ((m0 | m1 | m2 | m3) == 0ull) return;
  // lane-major, then component order == flat (row-major) order.  Branch-free: components that are
  // zero (or beyond the capacity) go to this lane's dump slot cap + lane (a shared slot would
  // serialise the 60-odd idle lanes on one LDS bank).
  const uint32_t p0 = cursor + mbcnt64(m0) + mbcnt64(m1) + mbcnt64(m2) + mbcnt64(m3);
  const uint32_t p1 = p0 + (uint32_t)n0, p2 = p1 + (uint32_t)n1, p3 = p2 + (uint32_t)n2;
  const uint32_t dump = cap + (uint32_t)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));  // cap + lane
  const uint32_t i0 = (n0 && p0 < cap) ? p0 : dump, i1 = (n1 && p1 < cap) ? p1 : dump;
  const uint32_t i2 = (n2 && p2 < cap) ? p2 : dump, i3 = (n3 && p3 < cap) ? p3 : dump;
  eflat[i0] = f;      eval[i0] = v.x;
  eflat[i1] = f + 1u; eval[i1] = v.y;
  eflat[i2] = f + 2u; eval[i2] = v.z;
  eflat[i3] = f + 3u; eval[i3] = v.w;
  cursor += (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
}

template <int U>
__device__ __forceinline__ uint32_t scan_dense_wave(const float* __restrict__ A, uint32_t n, int lane, uint32_t* eflat,
                                                    float* eval, uint32_t cap) {
  uint32_t cursor = 0;
  // head: elements before the first 16-byte boundary
  uint32_t head = (uint32_t)(((16u - (uint32_t)((uintptr_t)A & 15u)) & 15u) >> 2);
  if (head > n) head = n;
  if (head) {
    bool valid = (uint32_t)lane < head;
    float v = valid ? A[lane] : 0.0f;
    scan_chunk1(v, valid, (uint32_t)lane, cursor, eflat, eval, cap);
  }
  const float4* __restrict__ A4 = reinterpret_cast<const float4*>(A + head);
  const uint32_t n4 = (n - head) >> 2;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // main: batches of U x 1 KiB per wave, software-pipelined two deep (the next batch's loads are
  // in flight while the current one is scanned), so HBM latency overlaps the ballot/emit work
  const uint32_t step = 64u * U;
  float4 bufA[U], bufB[U];
  auto load_batch = [&](float4* buf, uint32_t t) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // unconditional 16-byte load from a clamped index (a select around the load would be split
      // into four predicated dword loads); out-of-range chunks are zeroed when scanned
      uint32_t i = t + (uint32_t)u * 64u + (uint32_t)lane;
      buf[u] = A4[i < n4 ? i : n4 - 1u];
    }
  };
  auto scan_batch = [&](const float4* buf, uint32_t t) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t i = t + (uint32_t)u * 64u + (uint32_t)lane;
      float4 v = buf[u];
      if (i >= n4) v = z4;
      scan_chunk4(v, head + 4u * i, cursor, eflat, eval, cap);
    }
  };
  if (n4 > 0) load_batch(bufA, 0);
  for (uint32_t t = 0; t < n4; t += 2u * step) {
    if (t + step < n4) load_batch(bufB, t + step);
    scan_batch(bufA, t);
    if (t + step < n4) {
      if (t + 2u * step < n4) load_batch(bufA, t + 2u * step);
      scan_batch(bufB, t + step);
    }
  }
  // tail
  const uint32_t done = head + 4u * n4;
  if (done < n) {
    bool valid = done + (uint32_t)lane < n;
    float v = valid ? A[done + lane] : 0.0f;
    scan_chunk1(v, valid, done + (uint32_t)lane, cursor, eflat, eval, cap);
  }
  return cursor;
}

}  // namespace cave
