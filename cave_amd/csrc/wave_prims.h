// wave_prims.h — wave64 primitives for gfx950 used by both SPMD contexts
// (ctx_wave.h: one wave per instance; ctx_block.h: NW cooperating waves per instance).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "cone_common.h"

namespace cave {

#if defined(__HIPCC__) && !defined(CAVE_SIMT_EMUL)
// 16-byte non-temporal load (global_load_dwordx4 ... nt): streamed once-read data stays out of the way of re-read data
__device__ __forceinline__ float4 nt_load_f4(const float4* p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
#endif

__device__ __forceinline__ double readlane_f64(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint32_t lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// ---- reductions / scans on the DPP crossbar (no LDS traffic, fixed summation tree)
template <int CTRL, int RM>
__device__ __forceinline__ double dpp_f64(double old, double x) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, RM, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, RM, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int RM>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, RM, 0xf, false);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<0xb1, 0xf>(0.0, v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4e, 0xf>(0.0, v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x114, 0xf>(0.0, v);  // row_shr:4
  v += dpp_f64<0x118, 0xf>(0.0, v);  // row_shr:8
  v += dpp_f64<0x142, 0xa>(0.0, v);  // row_bcast:15 -> rows 1,3
  v += dpp_f64<0x143, 0xc>(0.0, v);  // row_bcast:31 -> rows 2,3
  return readlane_f64(v, 63);
}
// two reductions at once: the DPP steps of one chain fill the wait states of the other (a lone chain pays two
// idle slots per step between the VALU write of a register and its DPP read: 165 cycles per reduction, measured)
template <bool MAX_B>
__device__ __forceinline__ void wave_reduce2_f64(double& a, double& b) {
#define CAVE_STEP(CTRL, RM)                                                      \
  {                                                                              \
    const double ta = dpp_f64<CTRL, RM>(0.0, a);                                 \
    const double tb = MAX_B ? dpp_f64<CTRL, RM>(b, b) : dpp_f64<CTRL, RM>(0.0, b); \
    a += ta;                                                                     \
    b = MAX_B ? fmax(b, tb) : b + tb;                                            \
  }
  CAVE_STEP(0xb1, 0xf) CAVE_STEP(0x4e, 0xf) CAVE_STEP(0x114, 0xf) CAVE_STEP(0x118, 0xf) CAVE_STEP(0x142, 0xa) CAVE_STEP(0x143, 0xc)
#undef CAVE_STEP
  a = readlane_f64(a, 63);
  b = readlane_f64(b, 63);
}
__device__ __forceinline__ double quad_sum_f64(double v) {  // every lane of the quad gets the sum
  v += dpp_f64<0xb1, 0xf>(0.0, v);
  v += dpp_f64<0x4e, 0xf>(0.0, v);
  return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
  v = fmax(v, dpp_f64<0xb1, 0xf>(v, v));
  v = fmax(v, dpp_f64<0x4e, 0xf>(v, v));
  v = fmax(v, dpp_f64<0x114, 0xf>(v, v));
  v = fmax(v, dpp_f64<0x118, 0xf>(v, v));
  v = fmax(v, dpp_f64<0x142, 0xa>(v, v));
  v = fmax(v, dpp_f64<0x143, 0xc>(v, v));
  return readlane_f64(v, 63);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  v += dpp_u32<0xb1, 0xf>(0u, v);
  v += dpp_u32<0x4e, 0xf>(0u, v);
  v += dpp_u32<0x114, 0xf>(0u, v);
  v += dpp_u32<0x118, 0xf>(0u, v);
  v += dpp_u32<0x142, 0xa>(0u, v);
  v += dpp_u32<0x143, 0xc>(0u, v);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// inclusive prefix sum across the 64 lanes (Hillis-Steele inside each 16-lane row, then row carries)
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t inc) {
  inc += dpp_u32<0x111, 0xf>(0u, inc);  // row_shr:1
  inc += dpp_u32<0x112, 0xf>(0u, inc);  // row_shr:2
  inc += dpp_u32<0x114, 0xf>(0u, inc);  // row_shr:4
  inc += dpp_u32<0x118, 0xf>(0u, inc);  // row_shr:8
  inc += dpp_u32<0x142, 0xa>(0u, inc);  // row_bcast:15 -> rows 1,3
  inc += dpp_u32<0x143, 0xc>(0u, inc);  // row_bcast:31 -> rows 2,3
  return inc;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [LO, HI)
template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (LO < HI) {
    f(std::integral_constant<int, LO>{});
    static_for<LO + 1, HI>(f);
  }
}

// Solve the p x p system whose free rows are rows of (H + delta*I) and whose
// fixed rows (act) are identity rows:   H_FF x_F + H_FA x_A = rhs_F,  x_A = rhs_A.
// ONE wave: lane i keeps row i of [H | rhs] in registers; Gauss-Jordan elimination with the pivot row
// normalised (lane k uses the multiplier (piv - 1) / piv, so x = the right-hand-side column at the end),
// pivot row broadcast with v_readlane (no LDS traffic, no barriers).
//
// What a pivot costs on one wave (measured, tools/micro/gj_bench.hip, prim_bench.hip): a dependent chain of
// ~140 cycles (v_readlane pair ~19, v_rcp_f64 ~17, each dependent f64 fma ~5.5) + ~15 cycles per remaining
// column (two v_readlane + one v_fma_f64).  Hence:
//   * the pivot row is read G columns at a time: the SGPR results of a batch are all written before the first
//     v_fma reads one, so the VALU-writes-SGPR hazard costs no s_nop (24.8 -> ~15 cycles per column);
//   * v_rcp_f64 (~24 bits) + ONE Newton step (~48 bits): the Newton direction of the outer iteration needs no
//     more, and an exact line search follows;
//   * nothing is selected between the pivot and its reciprocal: a non-positive / NaN pivot (numerically
//     dependent row) is clamped for the reciprocal and masked out of the multiplier: the row is skipped, x_k = 0.
// LOWER: only the lower triangle of H is valid (H[i][j], j <= i); the upper part is read transposed.
template <int PM, bool LOWER = false, int G = 4, bool TRI = false, class PH = const double*, class PR = const double*,
          class PA = const uint8_t*, class PD = double*>
__device__ __forceinline__ void gj_solve_regs(int lane, PH H, int ldh, PR rhs, PA act, int p, double reg_rel, PD dv) {
  const bool live = lane < p;
  const bool my_act = live && act[lane] != 0;
  double diag0 = (live && !my_act) ? H[TRI ? tri_idx((uint32_t)lane, (uint32_t)lane) : (uint32_t)(lane * ldh + lane)] : 0.0;
  const double maxdiag = wave_max_f64(diag0);
  const double reg = reg_rel * maxdiag;
  double h[PM + 1];  // h[PM]: right-hand side
#pragma unroll
  for (int j = 0; j < PM; ++j) {
    double v = 0.0;
    if (j < p) {
      if (live && !my_act) v = TRI ? H[tri_idx((uint32_t)lane, (uint32_t)j)] : (LOWER && j > lane) ? H[j * ldh + lane] : H[lane * ldh + j];
      if (j == lane) v = my_act ? 1.0 : v + reg;
    } else if (j == lane) v = 1.0;  // rows beyond p: identity
    h[j] = v;
  }
  h[PM] = live ? rhs[lane] : 0.0;
  uint64_t deadmask = 0;  // pivots that were not positive
  static_for<0, PM>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if (k < p) {
      const double piv = readlane_f64(h[k], k);
      const bool ok = piv > 1e-300;
      const double pv = fmax(piv, 1e-300);
      double inv = __builtin_amdgcn_rcp(pv);
      inv = fma(fma(-pv, inv, 1.0), inv, inv);
      deadmask |= ok ? 0ull : (1ull << k);
      const double num = ok ? h[k] - ((lane == k) ? 1.0 : 0.0) : 0.0;
      const double fac = num * inv;
      constexpr int NCOL = PM - k;  // columns k+1 .. PM (PM = right-hand side); columns >= p hold zeros
      static_for<0, (NCOL + G - 1) / G>([&](auto gc) {
        constexpr int j0 = k + 1 + decltype(gc)::value * G;
        double sv[G];
#pragma unroll
        for (int g = 0; g < G; ++g)
          if (j0 + g <= PM) sv[g] = readlane_f64(h[j0 + g], k);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g)
          if (j0 + g <= PM) h[j0 + g] -= fac * sv[g];
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  });
  if (live) dv[lane] = ((deadmask >> lane) & 1ull) ? 0.0 : h[PM];
}

// PARTIAL Gauss-Jordan: only the pivots k < nF are eliminated, from every row, of [H + reg I | rhs] (no fixed rows).
// With the rows ordered [F | I] the lanes then hold, in their columns j >= nF and the right-hand side:
//   rows i <  nF:  X_I = H_FF^-1 H_FI  and  x_g = H_FF^-1 rhs_F        (the pivot rows are normalised)
//   rows i >= nF:  S = H_II - H_IF H_FF^-1 H_FI  (Schur complement)  and  rhs_I - H_IF x_g
// which go to XS[i * nI + (j - nF)] and xg[i] (LDS).  The active-set loop of a Newton step then works on S alone
// and the rows of F follow by x_F = x_g - X_I x_I: one elimination per Newton iteration however many bounds block
// (the slowest TSP-20 instances of a batch spent 100 k of 285 k cycles in repeated full eliminations).
// A non-positive pivot drops its row as in gj_solve_regs (its X_I row and x_g entry are zero).
template <int PM, bool LOWER = false, int G = 4>
__device__ __forceinline__ void gj_partial_regs(int lane, const double* H, int ldh, const double* rhs, int p, int nF,
                                                double reg_rel, double* XS, double* xg) {
  const bool live = lane < p;
  double diag0 = live ? H[lane * ldh + lane] : 0.0;
  const double maxdiag = wave_max_f64(diag0);
  const double reg = reg_rel * maxdiag;
  double h[PM + 1];  // h[PM]: right-hand side
#pragma unroll
  for (int j = 0; j < PM; ++j) {
    double v = 0.0;
    if (j < p) {
      if (live) v = (LOWER && j > lane) ? H[j * ldh + lane] : H[lane * ldh + j];
      if (j == lane) v += reg;
    } else if (j == lane) v = 1.0;  // rows beyond p: identity
    h[j] = v;
  }
  h[PM] = live ? rhs[lane] : 0.0;
  uint64_t deadmask = 0;
  static_for<0, PM>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if (k < nF) {
      const double piv = readlane_f64(h[k], k);
      const bool ok = piv > 1e-300;
      const double pv = fmax(piv, 1e-300);
      double inv = __builtin_amdgcn_rcp(pv);
      inv = fma(fma(-pv, inv, 1.0), inv, inv);
      deadmask |= ok ? 0ull : (1ull << k);
      const double num = ok ? h[k] - ((lane == k) ? 1.0 : 0.0) : 0.0;
      const double fac = num * inv;
      constexpr int NCOL = PM - k;
      static_for<0, (NCOL + G - 1) / G>([&](auto gc) {
        constexpr int j0 = k + 1 + decltype(gc)::value * G;
        double sv[G];
#pragma unroll
        for (int g = 0; g < G; ++g)
          if (j0 + g <= PM) sv[g] = readlane_f64(h[j0 + g], k);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g)
          if (j0 + g <= PM) h[j0 + g] -= fac * sv[g];
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  });
  const bool dead = ((deadmask >> lane) & 1ull) != 0ull;  // (only pivots k < nF can be flagged)
  const int nI = p - nF;
  if (live) xg[lane] = dead ? 0.0 : h[PM];
  static_for<0, PM>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if (live && j >= nF && j < p) XS[lane * nI + (j - nF)] = dead ? 0.0 : h[j];
  });
}

template <int PLIM, bool LOWER = false>
__device__ __forceinline__ void gj_partial(int lane, const double* H, int ldh, const double* rhs, int p, int nF,
                                           double reg_rel, double* XS, double* xg) {
  if (p <= 8) gj_partial_regs<8, LOWER>(lane, H, ldh, rhs, p, nF, reg_rel, XS, xg);
  else if (p <= 16) gj_partial_regs<16, LOWER>(lane, H, ldh, rhs, p, nF, reg_rel, XS, xg);
  else if (p <= 20) gj_partial_regs<20, LOWER>(lane, H, ldh, rhs, p, nF, reg_rel, XS, xg);
  else if (p <= 22) gj_partial_regs<22, LOWER>(lane, H, ldh, rhs, p, nF, reg_rel, XS, xg);
  else if (p <= 24) gj_partial_regs<24, LOWER>(lane, H, ldh, rhs, p, nF, reg_rel, XS, xg);
  else if (p <= 26) gj_partial_regs<26, LOWER>(lane, H, ldh, rhs, p, nF, reg_rel, XS, xg);
  else if (p <= 28) gj_partial_regs<28, LOWER>(lane, H, ldh, rhs, p, nF, reg_rel, XS, xg);
  else if constexpr (PLIM >= 32) gj_partial_regs<32, LOWER>(lane, H, ldh, rhs, p, nF, reg_rel, XS, xg);
}

// ---- active-set loop of a small SPD system in registers (lite_model_step, cone_core.h)
// Tableau of  y = S x + c , one row per lane (lanes 0 .. NM-1), columns 0 .. NM-1 and the constant column NM.  A Jordan
// EXCHANGE of pivot (J, J) swaps the roles of x_J and y_J:
//     T'[J][J] = 1/d,   T'[J][k] = -T[J][k]/d,   T'[i][J] = T[i][J]/d,   T'[i][k] = T[i][k] - T[i][J] T[J][k]/d      (d = T[J][J]).
// With the free variables exchanged (their y = 0) and the others at x = 0, the constant column of an exchanged row IS the
// minimiser's x_J, and of the other rows the gradient there.  Fixing a variable at its bound = exchanging it back (the
// step is an involution), one rank-one update of NM + 1 columns (~250 cycles) where the loop used to run a fresh
// Gauss-Jordan of the whole system per round (~1.6 k cycles + its LDS traffic).  Lane J uses the multiplier (d + 1)/d,
// so its row needs no separate code path; the pivot row is read G columns at a time (see gj_solve_regs).
// Returns false (and changes nothing) when the pivot is not positive.
template <int NM, int J, int G = 4>
__device__ __forceinline__ bool tableau_exchange(double (&T)[NM + 1], int lane) {
  const double d = readlane_f64(T[J], J);
  if (!(d > 1e-300)) return false;  // wave-uniform
  double inv = __builtin_amdgcn_rcp(d);
  inv = fma(fma(-d, inv, 1.0), inv, inv);
  inv = fma(fma(-d, inv, 1.0), inv, inv);
  const double f = (T[J] + ((lane == J) ? 1.0 : 0.0)) * inv;
  constexpr int NC = NM + 1;
  static_for<0, (NC + G - 1) / G>([&](auto gc) {
    constexpr int k0 = decltype(gc)::value * G;
    double sv[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (k0 + g < NC && k0 + g != J) sv[g] = readlane_f64(T[k0 + g], J);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (k0 + g < NC && k0 + g != J) T[k0 + g] -= f * sv[g];
    __builtin_amdgcn_sched_barrier(0);
  });
  T[J] = (lane == J) ? inv : f;
  return true;
}
// minimum over lanes 0 .. 7 (the other lanes must hold a neutral value); every lane gets it
__device__ __forceinline__ double min8_f64(double v) {
  v = fmin(v, dpp_f64<0xb1, 0xf>(v, v));
  v = fmin(v, dpp_f64<0x4e, 0xf>(v, v));
  v = fmin(v, dpp_f64<0x114, 0xf>(v, v));  // row_shr:4: lanes 4 .. 7 now hold the minimum of both quads
  return readlane_f64(v, 7);
}
__device__ __forceinline__ double max8_f64(double v) {
  v = fmax(v, dpp_f64<0xb1, 0xf>(v, v));
  v = fmax(v, dpp_f64<0x4e, 0xf>(v, v));
  v = fmax(v, dpp_f64<0x114, 0xf>(v, v));
  return readlane_f64(v, 7);
}

// The round-1 form of the same elimination: one column at a time (v_readlane pair, hazard nop, v_fma), explicit
// diagonal.  ~20 % slower than gj_solve_regs but it needs fewer registers, which is what counts in the 4-wave
// kernels (128-VGPR budget: with the batched form they spill ~70 registers).
// TRI: H is the packed lower triangle (cone_common.h tri_idx; ldh unused)
template <int PM, bool TRI = false>
__device__ __forceinline__ void gj_solve_regs_small(int lane, const double* H, int ldh, const double* rhs,
                                              const uint8_t* act, int p, double reg_rel, double* dv) {
  const bool live = lane < p;
  const bool my_act = live && act[lane] != 0;
  double diag0 = (live && !my_act) ? H[TRI ? tri_idx((uint32_t)lane, (uint32_t)lane) : (uint32_t)(lane * ldh + lane)] : 0.0;
  const double maxdiag = wave_max_f64(diag0);
  const double reg = reg_rel * maxdiag;
  double h[PM];
#pragma unroll
  for (int j = 0; j < PM; ++j) {
    double v = 0.0;
    if (j < p) {
      if (live && !my_act) v = H[TRI ? tri_idx((uint32_t)lane, (uint32_t)j) : (uint32_t)(lane * ldh + j)];
      if (j == lane) v = my_act ? 1.0 : v + reg;
    }
    h[j] = v;
  }
  double b = live ? rhs[lane] : 0.0;
  double diag = 1.0;
  bool dead = !live;
#pragma unroll
  for (int k = 0; k < PM; ++k) {
    if (k < p) {
      const double piv = readlane_f64(h[k], k);
      const double bk = readlane_f64(b, k);
      // reciprocal by v_rcp_f64 + two Newton steps (full double accuracy, a third of the IEEE divide);
      // a non-positive / NaN pivot (numerically dependent row) gives inv = 0: the row is skipped, x_k = 0
      const bool ok = piv > 1e-300;
      double inv = __builtin_amdgcn_rcp(ok ? piv : 1.0);
      inv = fma(fma(-piv, inv, 1.0), inv, inv);
      inv = fma(fma(-piv, inv, 1.0), inv, inv);
      inv = ok ? inv : 0.0;
      if (lane == k) { diag = ok ? piv : 1.0; dead = dead || !ok; }
      const double fac = (lane == k) ? 0.0 : h[k] * inv;
#pragma unroll
      for (int j = k + 1; j < PM; ++j) h[j] -= fac * readlane_f64(h[j], k);  // columns >= p hold zeros
      b -= fac * bk;
    }
  }
  if (live) dv[lane] = dead ? 0.0 : b / diag;
}

template <int PLIM, bool TRI = false>
__device__ __forceinline__ void gj_solve_small(int lane, const double* H, int ldh, const double* g, const uint8_t* act,
                                               int p, double reg_rel, double* dv) {
  if (p <= 8) gj_solve_regs_small<8, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 16) gj_solve_regs_small<16, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 20) gj_solve_regs_small<20, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 24) gj_solve_regs_small<24, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 28) gj_solve_regs_small<28, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 32) gj_solve_regs_small<32, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if constexpr (PLIM > 32) {
    if (p <= 40) gj_solve_regs_small<40, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 48) gj_solve_regs_small<48, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 56) gj_solve_regs_small<56, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
    else gj_solve_regs_small<64, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  }
}

// The batched form for the multi-wave contexts with the 256-register budget (wave 0 solves systems of up to 64 rows: the
// TSP-50 class), as a REAL call: inlined into those kernels the 57-column row would spill, the one-column-at-a-time form
// above (which they used until round 4) pays the scalar-write hazard per column: 25 cycles against 15.
template <bool TRI>
CAVE_NOINLINE __device__ void gj_solve_wide_call(int lane, const double* H_, int ldh, const double* g_, const uint8_t* act_, int p,
                                                 double reg_rel, double* dv_) {
  // (everything lives in LDS on this path: typed, so that the call reads ds_* rather than flat_*)
  const auto H = space_cast<3>(H_);
  const auto g = space_cast<3>(g_);
  const auto act = space_cast<3>(act_);
  const auto dv = space_cast<3>(dv_);
  if (p <= 16) gj_solve_regs<16, false, 4, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 24) gj_solve_regs<24, false, 4, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 32) gj_solve_regs<32, false, 4, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 40) gj_solve_regs<40, false, 4, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 48) gj_solve_regs<48, false, 4, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 56) gj_solve_regs<56, false, 4, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
  else gj_solve_regs<64, false, 4, TRI>(lane, H, ldh, g, act, p, reg_rel, dv);
}

// size-specialised dispatch; PLIM bounds the register footprint (2*PM + 2 VGPRs for the row)
template <int PLIM, bool LOWER = false>
__device__ __forceinline__ void gj_solve(int lane, const double* H, int ldh, const double* g, const uint8_t* act, int p,
                                         double reg_rel, double* dv) {
  if (p <= 8) gj_solve_regs<8, LOWER>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 16) gj_solve_regs<16, LOWER>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 20) gj_solve_regs<20, LOWER>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 24) gj_solve_regs<24, LOWER>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 28) gj_solve_regs<28, LOWER>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 32) gj_solve_regs<32, LOWER>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if constexpr (PLIM > 32) {
    if (p <= 40) gj_solve_regs<40, LOWER>(lane, H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 48) gj_solve_regs<48, LOWER>(lane, H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 56) gj_solve_regs<56, LOWER>(lane, H, ldh, g, act, p, reg_rel, dv);
    else gj_solve_regs<64, LOWER>(lane, H, ldh, g, act, p, reg_rel, dv);
  }
}

// ---- streaming scan building blocks (see scan_dense_* in the contexts)
// Per 1 KiB chunk (one float4 per lane): which components are non-zero, this lane's ordered slot
// offset inside the chunk, and the chunk's total.  Lane-major, then component order == flat order.
struct ChunkSlots {
  uint32_t rel;    // slot of this lane's first non-zero, relative to the chunk start
  uint32_t nzm;    // 4-bit component mask
  uint32_t total;  // wave-uniform
  uint32_t cm;     // wave-uniform: bit c set when ANY lane's component c is non-zero
};
__device__ __forceinline__ ChunkSlots chunk_slots(float4 v) {
  bool n0 = v.x != 0.0f, n1 = v.y != 0.0f, n2 = v.z != 0.0f, n3 = v.w != 0.0f;
  uint64_t m0 = __ballot(n0), m1 = __ballot(n1), m2 = __ballot(n2), m3 = __ballot(n3);
  ChunkSlots s;
  s.rel = mbcnt64(m0) + mbcnt64(m1) + mbcnt64(m2) + mbcnt64(m3);
  s.nzm = (uint32_t)n0 | ((uint32_t)n1 << 1) | ((uint32_t)n2 << 2) | ((uint32_t)n3 << 3);
  s.total = (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
  s.cm = (m0 ? 1u : 0u) | (m1 ? 2u : 0u) | (m2 ? 4u : 0u) | (m3 ? 8u : 0u);
  return s;
}
// Branch-free store of a chunk's non-zeros at slots base+rel...; zero components (and slots beyond
// `cap`) go to this thread's private dump slot (a shared one would serialise on one LDS bank).
__device__ __forceinline__ void chunk_emit(float4 v, uint32_t f, uint32_t base, uint32_t rel, uint32_t nzm,
                                           uint32_t dump, uint32_t* eflat, float* eval, uint32_t cap) {
  const uint32_t p0 = base + rel;
  const uint32_t p1 = p0 + (nzm & 1u), p2 = p1 + ((nzm >> 1) & 1u), p3 = p2 + ((nzm >> 2) & 1u);
  const uint32_t i0 = ((nzm & 1u) && p0 < cap) ? p0 : dump, i1 = ((nzm & 2u) && p1 < cap) ? p1 : dump;
  const uint32_t i2 = ((nzm & 4u) && p2 < cap) ? p2 : dump, i3 = ((nzm & 8u) && p3 < cap) ? p3 : dump;
  eflat[i0] = f;      eval[i0] = v.x;
  eflat[i1] = f + 1u; eval[i1] = v.y;
  eflat[i2] = f + 2u; eval[i2] = v.z;
  eflat[i3] = f + 3u; eval[i3] = v.w;
}

// Variant for output arrays in global memory (large-cone path): predicated stores, no dump slots
// (a store per zero element would double the HBM traffic of the scan).
__device__ __forceinline__ void chunk_emit_cond(float4 v, uint32_t f, uint32_t base, uint32_t rel, uint32_t nzm,
                                                uint32_t* eflat, float* eval, uint32_t cap) {
  const uint32_t p0 = base + rel;
  const uint32_t p1 = p0 + (nzm & 1u), p2 = p1 + ((nzm >> 1) & 1u), p3 = p2 + ((nzm >> 2) & 1u);
  if ((nzm & 1u) && p0 < cap) { eflat[p0] = f;      eval[p0] = v.x; }
  if ((nzm & 2u) && p1 < cap) { eflat[p1] = f + 1u; eval[p1] = v.y; }
  if ((nzm & 4u) && p2 < cap) { eflat[p2] = f + 2u; eval[p2] = v.z; }
  if ((nzm & 8u) && p3 < cap) { eflat[p3] = f + 3u; eval[p3] = v.w; }
}

}  // namespace cave
