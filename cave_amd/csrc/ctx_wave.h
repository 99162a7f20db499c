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
__device__ __forceinline__ uint32_t scan_dense_wave(const float* __restrict__ A, uint32_t n, uint32_t d, int lane,
                                                    uint16_t* ecol, float* eval, uint32_t* rowcnt, uint32_t cap);

struct WaveCtx {
  static constexpr int NT = 64;
  static constexpr int TEAM = 4;         // lanes sharing one CSR row in gradient()
  static constexpr int SCAN_UNROLL = 8;  // 8 x 1 KiB dwordx4 loads in flight per wave
  __device__ __forceinline__ uint32_t scan_dense(const float* A, uint32_t n, uint32_t d, uint16_t* ecol, float* eval,
                                                 uint32_t* rowcnt, uint32_t cap) const {
    return scan_dense_wave<SCAN_UNROLL>(A, n, d, lane, ecol, eval, rowcnt, cap);
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
    else if (p <= 24) solve_spd_regs<24>(H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 32) solve_spd_regs<32>(H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 48) solve_spd_regs<48>(H, ldh, g, act, p, reg_rel, dv);
    else solve_spd_regs<64>(H, ldh, g, act, p, reg_rel, dv);
  }
};

// Stream one dense instance (n = m*d floats, row-major) and append its non-zeros
// in row-major order to (ecol, eval); rowcnt[row] receives the per-row count.
// Returns the number of non-zeros seen (entries beyond `cap` are counted, not stored).
// Non-zeros are rare (a few per KiB for structured cones), so the per-entry work sits behind
// one wave-uniform test per 1 KiB chunk and runs only in the lanes that own a non-zero.
__device__ __forceinline__ void emit_entry(uint32_t pos, uint32_t f, float v, uint32_t d, double inv_d, uint16_t* ecol,
                                           float* eval, uint32_t* rowcnt, uint32_t cap) {
  // f / d without an integer divide: the fp64 estimate is within +-1 of the quotient for f < 2^32
  uint32_t row = (uint32_t)((double)f * inv_d);
  int32_t col = (int32_t)(f - row * d);
  if (col < 0) { row -= 1u; col += (int32_t)d; }
  else if (col >= (int32_t)d) { row += 1u; col -= (int32_t)d; }
  if (pos < cap) {
    ecol[pos] = (uint16_t)col;
    eval[pos] = v;
  }
  atomicAdd(&rowcnt[row], 1u);
}

__device__ __forceinline__ void scan_chunk1(float v, bool valid, uint32_t f, uint32_t d, double inv_d,
                                            uint32_t& cursor, uint16_t* ecol, float* eval, uint32_t* rowcnt,
                                            uint32_t cap) {
  bool nz = valid && (v != 0.0f);
  uint64_t m = __ballot(nz);
  if (m == 0ull) return;
  if (nz) emit_entry(cursor + mbcnt64(m), f, v, d, inv_d, ecol, eval, rowcnt, cap);
  cursor += (uint32_t)__popcll(m);
}

__device__ __forceinline__ void scan_chunk4(float4 v, uint32_t f, uint32_t d, double inv_d, uint32_t& cursor,
                                            uint16_t* ecol, float* eval, uint32_t* rowcnt, uint32_t cap) {
  bool n0 = v.x != 0.0f, n1 = v.y != 0.0f, n2 = v.z != 0.0f, n3 = v.w != 0.0f;
  uint64_t m0 = __ballot(n0), m1 = __ballot(n1), m2 = __ballot(n2), m3 = __ballot(n3);
  if ((m0 | m1 | m2 | m3) == 0ull) return;
  // lane-major, then component order == flat (row-major) order
  uint32_t pos = cursor + mbcnt64(m0) + mbcnt64(m1) + mbcnt64(m2) + mbcnt64(m3);
  uint32_t nzm = (uint32_t)n0 | ((uint32_t)n1 << 1) | ((uint32_t)n2 << 2) | ((uint32_t)n3 << 3);
  while (nzm) {
    int ci = __ffs((int)nzm) - 1;
    nzm &= nzm - 1u;
    float val = ci == 0 ? v.x : (ci == 1 ? v.y : (ci == 2 ? v.z : v.w));
    emit_entry(pos, f + (uint32_t)ci, val, d, inv_d, ecol, eval, rowcnt, cap);
    pos++;
  }
  cursor += (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
}

template <int U>
__device__ __forceinline__ uint32_t scan_dense_wave(const float* __restrict__ A, uint32_t n, uint32_t d, int lane,
                                                    uint16_t* ecol, float* eval, uint32_t* rowcnt, uint32_t cap) {
  uint32_t cursor = 0;
  const double inv_d = 1.0 / (double)d;
  // head: elements before the first 16-byte boundary
  uint32_t head = (uint32_t)(((16u - (uint32_t)((uintptr_t)A & 15u)) & 15u) >> 2);
  if (head > n) head = n;
  if (head) {
    bool valid = (uint32_t)lane < head;
    float v = valid ? A[lane] : 0.0f;
    scan_chunk1(v, valid, (uint32_t)lane, d, inv_d, cursor, ecol, eval, rowcnt, cap);
  }
  const float4* __restrict__ A4 = reinterpret_cast<const float4*>(A + head);
  const uint32_t n4 = (n - head) >> 2;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // main: U x 1 KiB in flight per wave
  for (uint32_t t = 0; t < n4; t += 64u * U) {
    float4 buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t i = t + (uint32_t)u * 64u + (uint32_t)lane;
      buf[u] = (i < n4) ? A4[i] : z4;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t i = t + (uint32_t)u * 64u + (uint32_t)lane;
      scan_chunk4(buf[u], head + 4u * i, d, inv_d, cursor, ecol, eval, rowcnt, cap);
    }
  }
  // tail
  const uint32_t done = head + 4u * n4;
  if (done < n) {
    bool valid = done + (uint32_t)lane < n;
    float v = valid ? A[done + lane] : 0.0f;
    scan_chunk1(v, valid, done + (uint32_t)lane, d, inv_d, cursor, ecol, eval, rowcnt, cap);
  }
  return cursor;
}

}  // namespace cave
