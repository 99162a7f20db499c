// cone_core.h — per-instance cone projection, written once against a Ctx.
//
// Replaces, for one training instance, the work of
//   _project_nnls        (/root/reference  src/cave.py:298-309)
//   _average_ctrs        (src/cave.py:222-228)
//   the loss / target algebra of forward() and _get_projection()
//                        (src/cave.py:55-73,121-129,197-219)
//
// Algorithm (not the reference's Lawson-Hanson):
//   rows of A are split into signed-unit rows (c*e_k), +a/-a pairs (free
//   multiplier) and remaining "general" rows (non-negative multiplier).  With
//   M the reduced rows and Pi the coordinate-wise clip induced by the unit
//   rows, the projection residual is res* = Pi(y - M^T theta*) where theta*
//   minimises the C^1 piecewise-quadratic
//        f(theta) = 1/2 || Pi(y - M^T theta) ||^2 ,  theta_i >= 0 for unpaired rows
//   which is solved by a projected semismooth Newton method in fp64
//   (smoothed generalised Hessian M W M^T, active-set inner loop on the
//   quadratic model, exact line search on the true f).  proj = y - res*, rnorm = ||res*||_2.
#pragma once
#include <type_traits>
// tunables of the smoothed Hessian / inexact line search (defaults = the shipped values; tools/diag/tune_newton.py
// rebuilds the serial test build with other values to measure iteration counts)
#ifndef CAVE_MU_COEF
#define CAVE_MU_COEF 0.2
#endif
#ifndef CAVE_MU_CAP
#define CAVE_MU_CAP 0.7
#endif
#ifndef CAVE_BMU0
#define CAVE_BMU0 0.1      // band form: first smoothing scale (x max|y|), its decay per iteration, and the gradient-tied floor
#endif
#ifndef CAVE_BMU_DECAY
#define CAVE_BMU_DECAY 0.1
#endif
#ifndef CAVE_BMU_COEF
#define CAVE_BMU_COEF 0.03
#endif
#ifndef CAVE_PSITOL
#define CAVE_PSITOL 1e-1
#endif
#if defined(CAVE_STAMPS_FINE)
#define CAVE_ACCF(slot) CAVE_ACC(slot)
#else
#define CAVE_ACCF(slot) do {} while (0)
#endif
#include "cone_common.h"
#if defined(CAVE_GPU_CODE)
#include "wave_prims.h"
#endif
#include "cone_band.h"

namespace cave {

// ------------------------------------------------------------------ cone build

struct ConeBuild {
  int d, m;
  uint32_t nnz_all;  // non-zeros of the whole instance (row-major order)
  // ---- build-phase temporaries (top of the arena; dropped by Arena::release_top)
  uint32_t* erc;     // [cap] (row << 16) | col of every non-zero
  float* eall;       // [cap] its value
  uint32_t* rptr;    // [m+1]
  uint32_t* ucnt;    // [d] low 16 bits: number of +e_k rows, high 16 bits: number of -e_k rows
  float* rs2;        // [m] squared l2 norm of each row
  uint8_t* rowtag;   // [m]
  uint32_t* vraw;    // [p_raw] general rows, in row order (both twins of a pair)
  float* vnorm;      // [p_raw] l2 norm of those rows
  float* vinv;       // [p] 1/norm of the kept UNPAIRED rows that count in _average_ctrs, else 0
  int p_raw;
  // ---- persistent (bottom of the arena): what the solver reads
  uint8_t* usign;    // [d]
  bool pm1;          // every reduced-row entry is +-1: the sign sits in bit 15 of mcol / cvar, no value arrays
  uint32_t* mptr;    // [p+1] CSR row pointers of the reduced rows
  uint16_t* mcol;    // [nnzM]
  float* mval;       // [nnzM]  (null when pm1)
  uint8_t* vkind;    // [p]
  int p;
  uint32_t nnzM;
  uint32_t* cptr;    // [d+1]
  uint16_t* cvar;    // [nnzM]
  float* cvalc;      // [nnzM]  (null when pm1)
  int n_valid_proj;
  int n_valid_avg;
};

static constexpr uint32_t kHashPrefix = 12;
static constexpr uint32_t kLongRow = 64;   // rows longer than this are handled by the whole team, one at a time
static constexpr uint8_t ROW_PM1 = 0x20;   // every entry of the row is +-1
static constexpr uint8_t ROW_PENDING = 0xFF;

CAVE_HD uint8_t classify_row(float s1, float s2, bool allpm1, uint32_t len) {
  uint8_t tag = ROW_DROP;
  if (s1 > kDropRowAbsSum) {  // src/cave.py:303
    tag = (len == 1) ? ROW_UNIT : ROW_GENERAL;
    if (sqrtf(s2) > kAvgRowNorm) tag |= ROW_AVG_VALID;  // src/cave.py:224-225
    if (allpm1) tag |= ROW_PM1;
  }
  return tag;
}

// Classify rows, detect +a/-a pairs, build compact CSR/CSC of the reduced rows.
// Pre: cb.erc/eall hold the scan output in row-major order, cb.rptr the per-row counts
// (m+1 entries, last = 0), cb.nnz_all set.  Returns ST_OK or ST_TOO_LARGE.
template <class C>
CAVE_HD int32_t build_cone(C& c, Arena& ar, ConeBuild& cb) {
  const int d = cb.d, m = cb.m;
  const int NT = C::NT;
  CAVE_T0();
  // 1. row counts -> row pointers
  c.exclusive_scan_u32(cb.rptr, m + 1);
  cb.ucnt = ar.get_top<uint32_t>(d);
  cb.rs2 = ar.get_top<float>(m > 0 ? m : 1);
  cb.rowtag = ar.get_top<uint8_t>(m > 0 ? m : 1);
  const uint32_t nlong_cap = cb.nnz_all / kLongRow + 1u;
  uint32_t* longrows = ar.get_top<uint32_t>(nlong_cap);
  cb.usign = ar.get<uint8_t>(d);
  if (ar.ovf) return ST_TOO_LARGE;
  for (int k = c.tid(); k < d; k += NT) cb.ucnt[k] = 0;
  c.sync();
  // 2. per-row statistics and tags (one thread per row; long rows are left pending)
  for (int r = c.tid(); r < m; r += NT) {
    const uint32_t lo = cb.rptr[r], hi = cb.rptr[r + 1];
    if (hi - lo > kLongRow) { cb.rowtag[r] = ROW_PENDING; continue; }
    float s1 = 0.f, s2 = 0.f;
    bool allpm1 = true;
    for (uint32_t e = lo; e < hi; ++e) {
      float v = cb.eall[e];
      s1 += fabsf(v);
      s2 += v * v;
      allpm1 = allpm1 && (fabsf(v) == 1.0f);
    }
    const uint8_t tag = classify_row(s1, s2, allpm1, hi - lo);
    cb.rowtag[r] = tag;
    cb.rs2[r] = s2;
    if ((tag & 0x0F) == ROW_UNIT) c.atomic_add_u32(&cb.ucnt[cb.erc[lo] & 0xffffu], cb.eall[lo] > 0.f ? 1u : 0x10000u);
  }
  c.sync();
  const uint32_t nlong = c.compact_mask_u8(cb.rowtag, m, 0xFF, ROW_PENDING, longrows);
  c.sync();
  for (uint32_t li = (uint32_t)c.wave_id(); li < nlong; li += (uint32_t)C::NWAVES) {  // one wave per long row, fixed tree
    const uint32_t r = longrows[li];
    const uint32_t lo = cb.rptr[r], hi = cb.rptr[r + 1];
    double s1 = 0.0, s2 = 0.0, bad = 0.0;
    for (uint32_t e = lo + (uint32_t)c.lane_id(); e < hi; e += (uint32_t)C::WL) {
      float v = cb.eall[e];
      s1 += (double)fabsf(v);
      s2 += (double)v * (double)v;
      if (fabsf(v) != 1.0f) bad = 1.0;
    }
    s1 = c.wave_sum(s1);
    s2 = c.wave_sum(s2);
    bad = c.wave_max(bad);
    if (c.lane_id() == 0) {
      cb.rowtag[r] = classify_row((float)s1, (float)s2, bad == 0.0, hi - lo);
      cb.rs2[r] = (float)s2;
    }
  }
  c.sync();
  uint32_t nvp = 0, nva = 0, ngen = 0;
  for (int r = c.tid(); r < m; r += NT) {
    const uint8_t tag = cb.rowtag[r];
    nvp += (tag & 0x0F) != ROW_DROP;
    nva += (tag & ROW_AVG_VALID) != 0;
    ngen += (tag & 0x0F) == ROW_GENERAL;
  }
  cb.n_valid_proj = (int)c.reduce_add_u32(nvp);
  cb.n_valid_avg = (int)c.reduce_add_u32(nva);
  cb.p_raw = (int)c.reduce_add_u32(ngen);
  for (int k = c.tid(); k < d; k += NT)
    cb.usign[k] = (uint8_t)(((cb.ucnt[k] & 0xffffu) ? 1 : 0) | ((cb.ucnt[k] >> 16) ? 2 : 0));
  CAVE_ACC(11);
  // 3. ordered list of general rows
  const int pr = cb.p_raw;
  cb.vraw = ar.get_top<uint32_t>(pr > 0 ? pr : 1);
  cb.vnorm = ar.get_top<float>(pr > 0 ? pr : 1);
  uint32_t* twin = ar.get_top<uint32_t>(pr > 0 ? pr : 1);
  uint8_t* keep = ar.get_top<uint8_t>(pr > 0 ? pr : 1);
  uint16_t* rowvar = ar.get_top<uint16_t>(m > 0 ? m : 1);
  cb.cptr = ar.get<uint32_t>(d + 1);
  if (ar.ovf) return ST_TOO_LARGE;
  c.compact_mask_u8(cb.rowtag, m, 0x0F, ROW_GENERAL, cb.vraw);
  c.sync();
  // 4. pair detection by (signature(+a), signature(-a)); signatures live in scratch released below
  const uint32_t top_mark = ar.top;
  uint64_t* hp = ar.get_top<uint64_t>(pr > 0 ? pr : 1);
  uint64_t* hn = ar.get_top<uint64_t>(pr > 0 ? pr : 1);
  if (ar.ovf) return ST_TOO_LARGE;
  for (int i = c.tid(); i < pr; i += NT) {
    uint32_t r = cb.vraw[i];
    uint32_t lo = cb.rptr[r], hi = cb.rptr[r + 1];
    // signature of the row and of its negation: length + the first kHashPrefix entries (two 32-bit
    // FNV-1a streams each).  Candidates are verified entry by entry below, so a prefix is enough.
    uint32_t a0 = 0x811c9dc5u ^ (hi - lo), a1 = 0x9747b28cu + (hi - lo), b0 = a0, b1 = a1;
    const uint32_t hend = (hi - lo < kHashPrefix) ? hi : lo + kHashPrefix;
    for (uint32_t e = lo; e < hend; ++e) {
      uint32_t u = f2u(cb.eall[e]), col = cb.erc[e] & 0xffffu;
      uint32_t xp = u ^ (col * 0x9e3779b1u), xn = (u ^ 0x80000000u) ^ (col * 0x9e3779b1u);
      // xor-shift after each multiply so the sign bit (bit 31) reaches the low bits
      a0 = (a0 ^ xp) * 0x01000193u; a0 ^= a0 >> 15;
      a1 = (a1 + xp) * 0x85ebca6bu; a1 ^= a1 >> 13;
      b0 = (b0 ^ xn) * 0x01000193u; b0 ^= b0 >> 15;
      b1 = (b1 + xn) * 0x85ebca6bu; b1 ^= b1 >> 13;
    }
    hp[i] = ((uint64_t)a1 << 32) | a0;
    hn[i] = ((uint64_t)b1 << 32) | b0;
    cb.vnorm[i] = sqrtf(cb.rs2[r]);
  }
  c.sync();
  // TEAM lanes share one row's scan over all signatures (lane `sub` takes j = sub, sub+TEAM, ...); the
  // counts and, because a match only counts when it is unique, the SUM of matching indices are combined
  // with the team reduction
  {
    constexpr int TEAM = C::TEAM;
    constexpr int RPP = NT / TEAM;
    const int sub = c.tid() % TEAM;
    for (int base = 0; base < pr; base += RPP) {
      const int i = base + c.tid() / TEAM;
      const bool valid = i < pr;
      const uint64_t a = valid ? hp[i] : 0ull, b = valid ? hn[i] : 0ull;
      double same = 0.0, opp = 0.0, csum = 0.0;
      if (valid)
        for (int j = sub; j < pr; j += TEAM) {
          const uint64_t hj = hp[j];
          same += (hj == a) ? 1.0 : 0.0;
          if (hj == b) { opp += 1.0; csum += (double)j; }
        }
      same = c.team_reduce_sum(same);
      opp = c.team_reduce_sum(opp);
      csum = c.team_reduce_sum(csum);
      // exact verification (same columns, negated values), entries shared by the team as well
      const bool try_it = valid && same == 1.0 && opp == 1.0;
      const uint32_t cand = try_it ? (uint32_t)csum : 0u;
      double bad = 0.0;
      if (try_it) {
        const uint32_t r = cb.vraw[i], q = cb.vraw[cand];
        const uint32_t lo = cb.rptr[r], hi = cb.rptr[r + 1];
        const uint32_t lo2 = cb.rptr[q], hi2 = cb.rptr[q + 1];
        if ((hi - lo) != (hi2 - lo2)) bad = 1.0;
        else
          for (uint32_t e = (uint32_t)sub; e < hi - lo; e += (uint32_t)TEAM)
            if (((cb.erc[lo + e] & 0xffffu) != (cb.erc[lo2 + e] & 0xffffu)) || (cb.eall[lo + e] != -cb.eall[lo2 + e])) bad = 1.0;
      }
      bad = c.team_reduce_sum(bad);
      if (valid && sub == 0) twin[i] = (try_it && bad == 0.0) ? cand : 0xffffffffu;
    }
  }
  c.sync();
  for (int i = c.tid(); i < pr; i += NT) {
    uint32_t t = twin[i];
    bool paired = (t != 0xffffffffu) && (twin[t] == (uint32_t)i);
    // keep the lower-indexed twin as the free variable, drop the other
    keep[i] = (uint8_t)((paired && t < (uint32_t)i) ? 0 : (paired ? 2 : 1));
  }
  c.sync();
  ar.top = top_mark;  // release signature scratch
  CAVE_ACC(12);
  // 5. reduced variable list (ordered): reuse twin[] as the compacted index list
  uint32_t* vidx = twin;  // overwritten below only after keep[] has been fully derived
  cb.p = (int)c.compact_nonzero_u8(keep, pr, vidx);
  c.sync();
  const int p = cb.p;
  cb.mptr = ar.get<uint32_t>((uint32_t)p + 1u);
  cb.vkind = ar.get<uint8_t>(p > 0 ? p : 1);
  cb.vinv = ar.get_top<float>(p > 0 ? p : 1);
  if (ar.ovf) return ST_TOO_LARGE;
  for (int r = c.tid(); r < m; r += NT) rowvar[r] = 0xffffu;
  if (c.tid() == 0) cb.mptr[p] = 0u;
  c.sync();
  uint32_t notpm1 = 0;
  for (int i = c.tid(); i < p; i += NT) {
    const uint32_t src = vidx[i];
    const uint32_t r = cb.vraw[src];
    rowvar[r] = (uint16_t)i;
    cb.mptr[i] = cb.rptr[r + 1] - cb.rptr[r];
    cb.vkind[i] = (uint8_t)(keep[src] == 2 ? 1 : 0);
    cb.vinv[i] = (keep[src] != 2 && (cb.rowtag[r] & ROW_AVG_VALID)) ? 1.0f / fmaxf(cb.vnorm[src], (float)kNormClamp) : 0.f;
    notpm1 += (cb.rowtag[r] & ROW_PM1) ? 0u : 1u;
  }
  notpm1 = c.reduce_add_u32(notpm1);
  cb.pm1 = (notpm1 == 0u) && d <= 0x7fff && p <= 0x7fff;
  c.sync();
  cb.nnzM = c.exclusive_scan_u32(cb.mptr, p + 1);
  // 6. compact CSR + CSC of the reduced rows
  const uint32_t nz = cb.nnzM > 0 ? cb.nnzM : 1u;
  cb.mcol = ar.get<uint16_t>(nz);
  cb.cvar = ar.get<uint16_t>(nz);
  cb.mval = cb.pm1 ? nullptr : ar.get<float>(nz);
  cb.cvalc = cb.pm1 ? nullptr : ar.get<float>(nz);
  const uint32_t top_mark2 = ar.top;
  uint32_t* fill = ar.get_top<uint32_t>(d);
  if (ar.ovf) return ST_TOO_LARGE;
  for (int k = c.tid(); k <= d; k += NT) cb.cptr[k] = 0;
  for (int k = c.tid(); k < d; k += NT) fill[k] = 0;
  c.sync();
  const bool pm1 = cb.pm1;
  for (uint32_t e = c.tid(); e < cb.nnz_all; e += NT) {  // one thread per non-zero of the instance
    const uint32_t rc = cb.erc[e], r = rc >> 16, col = rc & 0xffffu;
    const uint32_t var = rowvar[r];
    if (var == 0xffffu) continue;
    const float val = cb.eall[e];
    const uint32_t dst = cb.mptr[var] + (e - cb.rptr[r]);
    cb.mcol[dst] = (uint16_t)(pm1 ? (col | (val < 0.f ? 0x8000u : 0u)) : col);
    if (!pm1) cb.mval[dst] = val;
    c.atomic_add_u32(&cb.cptr[col], 1u);
  }
  c.sync();
  c.exclusive_scan_u32(cb.cptr, d + 1);
  c.sync();
  for (uint32_t e = c.tid(); e < cb.nnz_all; e += NT) {
    const uint32_t rc = cb.erc[e], r = rc >> 16, col = rc & 0xffffu;
    const uint32_t var = rowvar[r];
    if (var == 0xffffu) continue;
    const float val = cb.eall[e];
    const uint32_t pos = cb.cptr[col] + c.atomic_inc_ret_u32(&fill[col]);
    cb.cvar[pos] = (uint16_t)(pm1 ? (var | (val < 0.f ? 0x8000u : 0u)) : var);
    if (!pm1) cb.cvalc[pos] = val;
  }
  c.sync();
  // the atomic cursor leaves each column in arbitrary order: insertion-sort it by reduced-row index
  // (one thread per column; columns are short) so every later summation order is fixed
  const uint32_t vmask = pm1 ? 0x7fffu : 0xffffu;
  for (int k = c.tid(); k < d; k += NT) {
    const uint32_t lo = cb.cptr[k], hi = cb.cptr[k + 1];
    for (uint32_t a = lo + 1; a < hi; ++a) {
      const uint16_t kv = cb.cvar[a];
      const float kf = pm1 ? 0.f : cb.cvalc[a];
      uint32_t b = a;
      while (b > lo && (cb.cvar[b - 1] & vmask) > (kv & vmask)) {
        cb.cvar[b] = cb.cvar[b - 1];
        if (!pm1) cb.cvalc[b] = cb.cvalc[b - 1];
        --b;
      }
      cb.cvar[b] = kv;
      if (!pm1) cb.cvalc[b] = kf;
    }
  }
  c.sync();
  ar.top = top_mark2;  // release fill[]
  CAVE_ACC(13);
  return ST_OK;
}

// Per-instance average of unit-normalised valid rows (src/cave.py:222-228), one thread per
// coordinate over the CSC of the reduced rows.  The two rows of a +a/-a pair contribute
// a/|a| - a/|a| = 0 and are skipped (the reference adds and subtracts them in float32, which
// leaves round-off of the order of 1e-8); they still count in the number of valid rows.
template <class C>
CAVE_HD void compute_avg(C& c, const ConeBuild& cb, float* avg) {
  const float invn = 1.0f / (float)(cb.n_valid_avg > 1 ? cb.n_valid_avg : 1);
  for (int k = c.tid(); k < cb.d; k += C::NT) {
    float a = (float)(cb.ucnt[k] & 0xffffu) - (float)(cb.ucnt[k] >> 16);
    for (uint32_t e = cb.cptr[k]; e < cb.cptr[k + 1]; ++e) {
      const uint32_t x = cb.cvar[e];
      const uint32_t var = cb.pm1 ? (x & 0x7fffu) : x;
      const float val = cb.pm1 ? ((x & 0x8000u) ? -1.0f : 1.0f) : cb.cvalc[e];
      a += val * cb.vinv[var];
    }
    avg[k] = a * invn;
  }
  c.sync();
}

// --------------------------------------------------------------------- solver

// entry e of the reduced CSR / CSC: index and value (value = sign bit of the index when v.pm1)
// (PM1 is a compile-time copy of v.pm1 so the hot loops carry no mode test)
// SP = 1: the caller knows the cone to be in GLOBAL memory (large-cone path: the packed store or the workspace; the far
// CSC of the diet layout) -> global_* loads; 0: wherever the view points (generic pointer: flat_* inside real calls)
template <bool PM1, int SP = 0>
CAVE_HD void csr_entry(const SolveView& v, uint32_t e, uint32_t& col, double& val) {
  const uint32_t x = space_cast<SP>(v.mcol)[e];
  if (PM1) { col = x & 0x7fffu; val = (x & 0x8000u) ? -1.0 : 1.0; }
  else { col = x; val = (double)space_cast<SP>(v.mval)[e]; }
}
template <bool PM1, int SP = 0>
CAVE_HD void csc_entry(const SolveView& v, uint32_t e, uint32_t& var, double& val) {
  const uint32_t x = space_cast<SP>(v.cvar)[e];
  if (PM1) { var = x & 0x7fffu; val = (x & 0x8000u) ? -1.0 : 1.0; }
  else { var = x; val = (double)space_cast<SP>(v.cvalc)[e]; }
}

CAVE_HD double clip_unit(double r, uint8_t u) {
  // residual left after the best multipliers of the +e_k / -e_k rows (closed form), branch-free:
  //   r - clamp(r, lo, hi),  lo = -inf if a -e_k row exists else 0,  hi = +inf if a +e_k row exists else 0
  // (u = 0: r;  u = 1: min(r, 0);  u = 2: max(r, 0);  u = 3: 0;  exact, since the clamp returns r or 0)
  const double lo = (u & 2) ? -HUGE_VAL : 0.0;
  const double hi = (u & 1) ? HUGE_VAL : 0.0;
  return r - fmin(fmax(r, lo), hi);
}

// does coordinate k carry curvature at residual r?  (D_kk of the generalised Hessian: the clip is the
// identity on u = 0 coordinates, one-sided on u = 1 / 2, and absorbs everything on u = 3)
CAVE_HD bool active_unit(double r, uint8_t u) {
  return (u == 0) | ((u == 1) & (r < 0.0)) | ((u == 2) & (r > 0.0));
}

// out[k] = base[k] - (M^T th)[k]   (base = y, or null for 0);  one CSC gather pass
template <class C, bool PM1>
CAVE_HD void gather_mt(C& c, const SolveView& v, const float* base, const double* th, double sgn, double* out) {
  for (int k = c.tid(); k < v.d; k += C::NT) {
    double r = base ? (double)base[k] : 0.0;
    for (uint32_t e = v.cptr[k]; e < v.cptr[k + 1]; ++e) {
      uint32_t var;
      double val;
      csc_entry<PM1>(v, e, var, val);
      r += sgn * val * th[var];
    }
    out[k] = r;
  }
  c.sync();
}

// rc = Pi(r);  returns 1/2 || rc ||^2
template <class C, int R = 1>
CAVE_HD double refresh_clipped(C& c, const SolveView& v, const double* r, double* rc) {
  double acc = 0.0;
  strided_batched<R, C::NT>(c.tid(), v.d, [&](int k) { return Ld2{r[k], (double)v.usign[k]}; },
                            [&](int k, const Ld2& x) {
                              const double t = clip_unit(x.a, (uint8_t)x.b);
                              rc[k] = t;
                              acc += t * t;
                            });
  double f = 0.5 * c.reduce_sum(acc);
  c.sync();
  return f;
}

// g = -M rc.  Rows are shared by TEAM adjacent lanes (fixed reduction tree); rows longer than
// kLongRow entries are summed by one whole wave each.

template <class C, bool PM1> CAVE_NOINLINE void dense_gradient(C& c_, const SolveView& v_, const double* rc, double* g_);
template <class C, bool PM1> CAVE_HD void gradient_long_rows_streamed(C& c, const SolveView& v, const double* rc, double* g);
template <class C, bool PM1> CAVE_NOINLINE void gradient_short_rows_streamed(C& c, const SolveView& v, const double* rc, double* g);

template <class C, bool PM1, bool STREAMED = false>
CAVE_HD void gradient(C& c, const SolveView& v, const double* rc, double* g) {
  constexpr int TEAM = C::TEAM;
  constexpr int RPP = C::NT / TEAM;  // rows per pass
  if constexpr (STREAMED && C::WL > 1 && TEAM == 4) {  // large-cone path: the cone is in global memory
    if (v.gcol_bound > 0.0) {  // dense reduced systems (TSP-100 class): column-wise, fixed point (cone_dense.h)
      dense_gradient<C, PM1>(c, v, rc, g);
      return;
    }
    gradient_short_rows_streamed<C, PM1>(c, v, rc, g);
    gradient_long_rows_streamed<C, PM1>(c, v, rc, g);
    c.sync();
    return;
  }
  const int sub = c.tid() % TEAM;
  for (int base = 0; base < v.p; base += RPP) {
    const int i = base + c.tid() / TEAM;
    const bool valid = i < v.p;
    double part = 0.0;
    uint32_t lo = 0, hi = 0;
    if (valid) { lo = v.mptr[i]; hi = v.mptr[i + 1]; }
    const bool is_long = (hi - lo) > kLongRow;
    if (valid && !is_long)
      for (uint32_t e = lo + sub; e < hi; e += TEAM) {
        uint32_t col;
        double val;
        csr_entry<PM1>(v, e, col, val);
        part -= val * rc[col];
      }
    part = c.team_reduce_sum(part);
    if (valid && !is_long && sub == 0) g[i] = part;
  }
  if constexpr (STREAMED && C::WL > 1) gradient_long_rows_streamed<C, PM1>(c, v, rc, g);
  else
  for (int li = c.wave_id(); li < v.nlong; li += C::NWAVES) {  // one wave per long row
    const int i = (int)v.longrow[li];
    const uint32_t lo = v.mptr[i], hi = v.mptr[i + 1];
    double part = 0.0;
    for (uint32_t e = lo + (uint32_t)c.lane_id(); e < hi; e += (uint32_t)C::WL) {
      uint32_t col;
      double val;
      csr_entry<PM1>(v, e, col, val);
      part -= val * rc[col];
    }
    part = c.wave_sum(part);
    if (c.lane_id() == 0) g[i] = part;
  }
  c.sync();
}

// ------------------------------------------------------------ small +-1 cones on ONE wave ("lite" form)
//
// One Newton iteration is a chain of short dependent phases.  Measured on TSP-20 (p ~ 24 reduced rows, d = 190,
// ~530 non-zeros; tools/diag/run_stamps.py, tools/micro/lite_bench.hip): 32-36 k cycles per iteration, the same
// on one wave and on four.  The time goes into dependent LDS round trips (~70 cycles each: row pointer -> entry
// -> operand, once per entry of a dynamic-length loop) and into instruction issue (~5 cycles per instruction
// for a lone wave), not into barriers or arithmetic.  For cones with +-1 entries, at most 32 reduced rows, at
// most 8 entries per column, at most 1536 non-zeros and d <= 256 the one-wave kernels therefore run the solver
// (SoloCtx, ctx_block.h) over two regular index structures built once per instance, in which every load of a
// phase is independent of the others and NOTHING is predicated or branched on inside a phase:
//   ell   [d][8] u16   the column of coordinate k: reduced row | sign << 15; unused slots hold row 32, a dummy
//                      multiplier that is always 0 (ONE 16-byte LDS read per coordinate; M^T x, Hessian updates)
//   csr16 [chn8 / 8][64][8] u16   CSR entry e = lane * chn8 + c: col | sign << 15; unused slots hold column d,
//                      a dummy coordinate whose residual is always 0.  M x by PREFIX SUMS: each lane adds up its
//                      contiguous run, a wave scan of the lane totals gives the sum of the lanes before it, and row i
//                      is P(last entry of i) - P(last entry of i - 1).  The LAST entry of a row carries the row in
//                      bits 9..13 and a flag in bit 14: its lane stores the running sum under the row's number (every
//                      other entry stores into the lane's dump slot: no predication), so the prefix table is 33 + 64
//                      doubles instead of one per entry (8-12 KB: the room that lets cones of up to 1536 non-zeros in).
struct LiteCone {
  const uint32_t* ell;    // [4 * d]            (16-byte aligned)
  const uint32_t* csr16;  // [4 * 64 * chn8 / 8 ... ] = 32 * chn8 words (16-byte aligned)
  double* rs;             // [33 + 64 + 65]     rs[i]: running sum of its lane at the last entry of row i (rs[32] spare),
                          //                    rs[33 + lane]: dump slots; then lbase[65]: sum of the lanes before lane l
  const uint8_t* rl;      // [p]                lane that holds the last entry of row i
  int chn8;               // CSR entries per lane: 8, 16 or 24
  int cmax;               // largest column count
};
static constexpr int kLiteMaxRows = 32, kLiteMaxD = 256, kLiteMaxCol = 8, kLiteMaxChunk = 24;
static constexpr uint32_t kLiteDummyRow = 32;

CAVE_HOSTDEV uint32_t lite_chunk(uint32_t nnz) { return nnz <= 512u ? 8u : (nnz <= 1024u ? 16u : 24u); }
CAVE_HOSTDEV uint32_t lite_lds_bytes(int d, uint32_t nnz) {
  const uint32_t chn8 = lite_chunk(nnz);
  return 16u * (uint32_t)d + 128u * chn8 + 8u * (33u + 64u + 65u) + 40u + 64u;
}

// Build the lite structures (all threads of the context).  Returns false when the cone does not qualify or the
// arena has no room; the caller then runs the general solver.
template <class C>
CAVE_HD bool lite_build(C& c, Arena& ar, const SolveView& v, LiteCone& L) {
  const int NT = C::NT;
  const int d = v.d, p = v.p;
  if (!v.pm1 || p < 1 || p > kLiteMaxRows || d > kLiteMaxD) return false;
  const uint32_t nnz = v.mptr[p];
  if (nnz == 0u || nnz > 64u * (uint32_t)kLiteMaxChunk) return false;
  const uint32_t chn8 = lite_chunk(nnz);
  uint32_t* ell = ar.try_get<uint32_t, 16u>(4u * (uint32_t)d);
  uint32_t* csr16 = ell ? ar.try_get<uint32_t, 16u>(32u * chn8) : nullptr;
  double* rs = csr16 ? ar.try_get<double>(33u + 64u + 65u) : nullptr;
  uint8_t* rl = rs ? ar.try_get<uint8_t>(40u) : nullptr;
  if (!rl) return false;
  // columns -> ELL rows of 8 (unused slots: the dummy row)
  uint32_t over = 0;
  double cm = 0.0;
  for (int k = c.tid(); k < d; k += NT) {
    const uint32_t lo = v.cptr[k], cnt = v.cptr[k + 1] - lo;
    over += cnt > (uint32_t)kLiteMaxCol ? 1u : 0u;
    cm = fmax(cm, (double)cnt);
    uint32_t x[8];
#pragma unroll
    for (uint32_t e = 0; e < 8; ++e) x[e] = v.cvar[lo + (e < cnt ? e : 0u)];  // unconditional loads
#pragma unroll
    for (uint32_t e = 0; e < 8; ++e) x[e] = e < cnt ? x[e] : kLiteDummyRow;
#pragma unroll
    for (int q = 0; q < 4; ++q) ell[4 * k + q] = x[2 * q] | (x[2 * q + 1] << 16);
  }
  // CSR entries, 8 per 16-byte word group: entry e = lane * chn8 + c at u16 index ((c / 8) * 64 + lane) * 8 + c % 8.
  // The stored sign is that of -M (the gradient is g = -M rc).
  uint16_t* c16 = reinterpret_cast<uint16_t*>(csr16);
  for (uint32_t idx = c.tid(); idx < 64u * chn8; idx += NT) {
    const uint32_t g8 = idx / 512u, ln = (idx / 8u) & 63u, u = idx & 7u;
    const uint32_t e = ln * chn8 + g8 * 8u + u;
    const uint32_t x = v.mcol[e < nnz ? e : 0u];
    c16[idx] = (uint16_t)(e < nnz ? (x ^ 0x8000u) : (uint32_t)d);
  }
  over = c.reduce_add_u32(over);
  L.cmax = (int)c.reduce_max(cm);
  c.sync();
  // the last entry of every row: row number + flag into its index word, its lane into rl (rows are never empty: a
  // reduced row has at least two entries)
  uint32_t empty = 0;
  for (int i = c.tid(); i < p; i += NT) {
    const uint32_t lo = v.mptr[i], hi = v.mptr[i + 1];
    if (hi <= lo) { empty = 1u; continue; }
    const uint32_t e = hi - 1u, ln = e / chn8, cc = e - ln * chn8;
    const uint32_t idx = ((cc / 8u) * 64u + ln) * 8u + (cc & 7u);
    c16[idx] = (uint16_t)(c16[idx] | 0x4000u | ((uint32_t)i << 9));
    rl[i] = (uint8_t)ln;
  }
  empty = c.reduce_add_u32(empty);
  if (c.tid() == 0) rs[33u + 64u + 64u] = 0.0;
  c.sync();
  if (over || empty) return false;
  L.ell = ell; L.csr16 = csr16; L.rs = rs; L.rl = rl; L.chn8 = (int)chn8;
  return true;
}

#if defined(CAVE_GPU_CODE)
// +-x with the sign taken from bit 15 of a 16-bit entry (no select: the sign bit is xor-ed in)
__device__ __forceinline__ double lite_signed(double x, uint32_t entry) {
  return __hiloint2double(__double2hiint(x) ^ (int)((entry & 0x8000u) << 16), __double2loint(x));
}
// inclusive prefix sum across the 64 lanes (fixed tree)
__device__ __forceinline__ double wave_inclusive_scan_f64(double v) {
  v += dpp_f64<0x111, 0xf>(0.0, v);  // row_shr:1
  v += dpp_f64<0x112, 0xf>(0.0, v);  // row_shr:2
  v += dpp_f64<0x114, 0xf>(0.0, v);  // row_shr:4
  v += dpp_f64<0x118, 0xf>(0.0, v);  // row_shr:8
  v += dpp_f64<0x142, 0xa>(0.0, v);  // row_bcast:15 -> rows 1,3
  v += dpp_f64<0x143, 0xc>(0.0, v);  // row_bcast:31 -> rows 2,3
  return v;
}

// (M^T x)[k] for one coordinate: one 16-byte index read, 8 independent operand reads (x[32] = 0 is the dummy)
__device__ __forceinline__ double lite_col_dot(const LiteCone& L, int k, const double* x) {
  const uint4 t = *reinterpret_cast<const uint4*>(L.ell + 4 * k);
  const uint32_t w[4] = {t.x, t.y, t.z, t.w};
  double val[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) val[e] = x[(w[e >> 1] >> ((e & 1) * 16)) & 63u];
  double acc = 0.0;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc += lite_signed(val[e], w[e >> 1] >> ((e & 1) * 16));
  return acc;
}

// out[k] = base[k] + sgn * (M^T x)[k] for the coordinates of this wave (x: [33] doubles in LDS, x[32] = 0)
template <class C>
__device__ __forceinline__ void lite_gather(C& c, const LiteCone& L, int d, const float* base, const double* x, double sgn,
                                            double* out) {
  for (int k = c.tid(); k < d; k += C::NT) out[k] = (base ? (double)base[k] : 0.0) + sgn * lite_col_dot(L, k, x);
  c.sync();
}

// g = -M rc by prefix sums (rc[d] = 0 is the dummy coordinate).  Each lane forms the running sums of its own run and
// stores the one at a row's last entry under the row's number; the sum of the lanes before it (one wave scan of the
// lane totals) goes to lbase[lane] and is added when a row's two end points are read: (run + base) in a fixed order.
template <class C>
__device__ __forceinline__ void lite_gradient(C& c, const LiteCone& L, int p, const double* rc, double* g) {
  const int lane = c.lane_id();
  double run = 0.0;
  double* dump = L.rs + 33 + lane;
#pragma unroll
  for (int g8 = 0; g8 < kLiteMaxChunk / 8; ++g8) {
    if (g8 * 8 < L.chn8) {  // wave-uniform; the loads of a group are all issued before the first add
      const uint4 t = *reinterpret_cast<const uint4*>(L.csr16 + 4 * (g8 * 64 + lane));
      const uint32_t w[4] = {t.x, t.y, t.z, t.w};
      double val[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) val[u] = rc[(w[u >> 1] >> ((u & 1) * 16)) & 0x1ffu];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t x = w[u >> 1] >> ((u & 1) * 16);
        run += lite_signed(val[u], x);
        double* q = (x & 0x4000u) ? L.rs + ((x >> 9) & 31u) : dump;
        *q = run;
      }
    }
  }
  double* lbase = L.rs + 33 + 64;
  lbase[lane] = wave_inclusive_scan_f64(run) - run;  // sum of the lanes before this one
  c.sync();
  if (lane < p) {
    const int l1 = L.rl[lane], l0 = L.rl[lane > 0 ? lane - 1 : 0];
    const double hi = lbase[l1] + L.rs[lane];
    const double lo = lane > 0 ? lbase[l0] + L.rs[lane - 1] : 0.0;
    g[lane] = hi - lo;
  }
  c.sync();
}
// H += (w_k - w_k_old) m_k m_k^T for the coordinates whose smoothed weight changed (see solve_cone_impl), lite
// form: four coordinates per lane, all of their operands loaded before the first weight is computed; each
// column comes in one 16-byte read and all lanes walk its (e1, e2 <= e1) pairs together, stopping at the longest
// column that changed (no loads inside, so branching costs nothing there).  Only the LOWER triangle of H is
// kept (columns are sorted by reduced row: e2 < e1 means b < a); gj_solve<.., LOWER> reads the rest transposed.
template <class C>
__device__ __forceinline__ void lite_hessian(C& c, const LiteCone& L, const SolveView& v, SolveWork& w, const double* r,
                                             double mu, double inv_mu) {
  constexpr int KC = kLiteMaxD / 64;
  const int lane = c.lane_id(), d = v.d, ldh = w.ldh;
  double rk[KC];
  float wo[KC];
  uint32_t us[KC];
#pragma unroll
  for (int s = 0; s < KC; ++s) {
    const int k = lane + 64 * s, kc = k < d ? k : d - 1;
    rk[s] = r[kc];
    wo[s] = w.wold[kc];
    us[s] = v.usign[kc];
  }
  const float imu = (float)inv_mu;
#pragma unroll
  for (int s = 0; s < KC; ++s) {
    if (64 * s >= d) continue;  // wave-uniform: no coordinate in this slot
    const int k = lane + 64 * s;
    const uint32_t u = us[s];
    const float t = (u == 2u) ? (float)rk[s] : -(float)rk[s];  // > 0 on the side that carries residual
    const float z = t * imu;  // the weight is a heuristic, quantised anyway: float is plenty
    float wn = t > 0.0f ? 1.0f : 0.0f;
    if (mu > 0.0 && fabsf(z) < 4.0f) wn = floorf(8.0f * (1.0f + z * __builtin_amdgcn_rsqf(1.0f + z * z)) + 0.5f) * (1.0f / 16.0f);
    wn = (u == 0u) ? 1.0f : (u == 3u) ? 0.0f : wn;
    const bool changed = k < d && wn != wo[s];
    if (__ballot(changed) == 0ull) continue;  // wave-uniform
    if (changed) w.wold[k] = wn;
    const double dw = (double)wn - (double)wo[s];
    const uint4 t4 = *reinterpret_cast<const uint4*>(L.ell + 4 * (k < d ? k : d - 1));
    const uint32_t cw[4] = {t4.x, t4.y, t4.z, t4.w};
    static_for<0, 8>([&](auto e1c) {
      constexpr int e1 = decltype(e1c)::value;
      const uint32_t x1 = (cw[e1 >> 1] >> ((e1 & 1) * 16)) & 0xffffu;
      const bool on1 = changed && (x1 & 0x7fffu) != kLiteDummyRow;
      if (e1 < L.cmax && __ballot(on1) != 0ull) {
        const uint32_t a = x1 & 0x7fffu;
        const double va = (x1 & 0x8000u) ? -dw : dw;
        double* Ha = w.H + a * ldh;
        if (on1) c.atomic_add_f64(Ha + a, dw);
        static_for<0, e1>([&](auto e2c) {
          constexpr int e2 = decltype(e2c)::value;
          const uint32_t x2 = (cw[e2 >> 1] >> ((e2 & 1) * 16)) & 0xffffu;
          if (on1) c.atomic_add_f64(Ha + (x2 & 0x7fffu), (x2 & 0x8000u) ? -va : va);
        });
      }
    });
  }
}
// Model minimisation of one Newton iteration for a lite cone whose rows are ordered [free | bound] with a few bound
// rows (TSP-20: 20 degree equalities, then <= 5 subtour cuts with theta >= 0).  The register Gauss-Jordan eliminates
// the free rows ONCE (gj_partial: pivots k < nF out of every row), which leaves the Schur complement S of the bound
// rows and the reduced right-hand side; the active-set loop (ratio test, fix the blocking row, solve again) then runs
// on S alone -- an nI x nI system, ~1 k cycles per round instead of ~9 k for the full elimination -- and the free rows
// follow from x_F = x_g - X_I x_I.  The iterates of the bound rows are those of the full-space loop (the free rows are
// at their optimum for every trial point either way).  Returns whether tc differs from theta.
// This IS the model minimisation of the lite solver (cones with other row orders or more than 8 bound rows do not
// take the lite form): the one-wave kernel sits at 250 VGPRs, and carrying this next to the full-elimination loop --
// inlined or as a call -- spilled 32-80 VGPRs and made every instance slower.
template <class C>
__device__ __forceinline__ bool lite_model_step(C& c, const SolveView& v, SolveWork& w, const double* theta, double* tc,
                                                double reg_rel) {
  const int p = v.p, nF = w.ls_nF, nI = w.ls_nI, lane = c.lane_id();
  double* XS = w.ls_scr;          // [p * nI]  rows < nF: X_I, rows >= nF: S
  double* xg = w.step;            // [p]       rows < nF: x_g, rows >= nF: reduced right-hand side (idle otherwise)
  double* st = XS + p * nI;       // [nI]      the bound rows' multipliers the loop ends with
  double* rhs = w.g2;
  if (lane < p) rhs[lane] = -w.g[lane];
  c.sync();
  gj_partial<32, true>(lane, w.H, w.ldh, rhs, p, nF, reg_rel, XS, xg);
  c.sync();
  bool moved = false;
  constexpr int NM = 8;  // (the lite form takes at most 8 bound rows)
  const bool mine = lane < nI;
  const int row = nF + lane;
  for (int attempt = 0; attempt < 2 && !moved; ++attempt) {
    double gmin = 0.0;
    if (attempt == 1) {
      double gl = 0.0;
      if (mine && theta[row] <= 0.0) gl = fmin(gl, w.g[row]);
      gmin = -c.reduce_max(-gl);
      if (!(gmin < 0.0)) break;
    }
    // The active-set loop on the Schur complement, in registers (wave_prims.h tableau_exchange): lane i < nI holds row i
    // of [S + reg I | c], c = (reduced gradient at theta) - S theta_I, i.e. the model in ABSOLUTE coordinates -- its
    // minimiser over a free set, with the other rows at 0, does not depend on the working point.  Free rows are
    // exchanged in once; every round reads the minimiser off the constant column, takes the ratio test from the working
    // point towards it, and exchanges the blocking row(s) back out.  Same iterates as the round-by-round solves this
    // replaces (which cost the slowest instances of a TSP-20 batch ~10 k cycles per Newton iteration).
    const double th_i = mine ? theta[row] : 0.0;
    bool act = false, frozen = false;
    if (mine) {
      const bool at_bound = th_i <= 0.0;
      const bool release = attempt == 0 ? (w.g[row] < 0.0) : (w.g[row] <= gmin);
      act = at_bound && !release;
    }
    double T[NM + 1];
#pragma unroll
    for (int j = 0; j < NM; ++j) T[j] = (mine && j < nI) ? XS[row * nI + j] : ((j == lane) ? 1.0 : 0.0);
    const double dmax = max8_f64((mine && !act) ? XS[row * nI + lane] : 0.0);
    const double reg2 = reg_rel * dmax;
    double c0 = mine ? -xg[row] : 0.0;
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      if (j == lane && mine) T[j] += reg2;
      c0 -= T[j] * readlane_f64(th_i, j);  // (theta of rows >= nI reads as 0)
    }
    T[NM] = c0;
    double st_i = th_i;
    uint32_t fmask = (uint32_t)__ballot(mine && !act) & 0xffu;
    static_for<0, NM>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      if ((fmask >> J) & 1u)
        if (!tableau_exchange<NM, J>(T, lane)) frozen = frozen || (lane == J);
    });
    for (int inner = 0; inner <= nI; ++inner) {
      const bool fr = mine && !act && !frozen;
      const double t = fr ? T[NM] : (act ? 0.0 : st_i);
      double amin = 2.0;
      if (fr && t < 0.0) amin = st_i / (st_i - t);
      amin = min8_f64(amin);
      const bool blocked = amin < 1.0;
      const double a = blocked ? fmax(amin, 0.0) : 1.0;
      const bool fixnow = fr && blocked && t < 0.0 && st_i <= a * (st_i - t) * (1.0 + 1e-12);
      double tn = blocked ? st_i + a * (t - st_i) : t;
      if (fixnow || act) tn = 0.0;
      st_i = tn;
      act = act || fixnow;
      if (!blocked) break;
      fmask = (uint32_t)__ballot(fixnow) & 0xffu;
      static_for<0, NM>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if ((fmask >> J) & 1u) tableau_exchange<NM, J>(T, lane);
      });
    }
    if (mine) st[lane] = st_i;
    double du[NM];  // displacement of the bound rows, broadcast (wave-uniform code: not inside the lane tests below)
#pragma unroll
    for (int j = 0; j < NM; ++j) du[j] = readlane_f64(st_i - th_i, j);  // (lanes >= nI: 0)
    c.sync();
    double mv = 0.0;
    if (lane < p) {
      double t;
      if (lane < nF) {
        double x = xg[lane];
#pragma unroll
        for (int j = 0; j < NM; ++j)
          if (j < nI) x -= XS[lane * nI + j] * du[j];
        t = theta[lane] + x;
      } else t = st[lane - nF];
      tc[lane] = t;
      mv = fabs(t - theta[lane]);
    }
    moved = c.reduce_max(mv) > 0.0;
    c.sync();
  }
  return moved;
}
#endif  // CAVE_GPU_CODE

// phi'(alpha) and phi''(alpha) of phi(alpha) = 1/2 || Pi(r - alpha q) ||^2
template <class C, int R = 1>
CAVE_HD void dphi(C& c, const SolveView& v, const double* r, const double* q, double alpha, double* d1, double* d2) {
  double a1 = 0.0, a2 = 0.0;
  strided_batched<R, C::NT>(c.tid(), v.d, [&](int k) { return Ld3{q[k], r[k], (double)v.usign[k]}; },
                            [&](int, const Ld3& x) {
                              const double qk = x.a;
                              const double rr = x.b - alpha * qk;
                              const uint8_t u = (uint8_t)x.c;
                              a1 -= clip_unit(rr, u) * qk;
                              if (active_unit(rr, u)) a2 += qk * qk;
                            });
  *d1 = c.reduce_sum(a1);
  *d2 = c.reduce_sum(a2);
}

// theta += alpha * dv with exact zeros for variables parked at / blocked by a bound
template <class C, int R = 1>
CAVE_HD void update_theta(C& c, const SolveView& v, double* theta, const double* tc, const double* dv, double alpha,
                          double amax) {
  strided_batched<R, C::NT>(c.tid(), v.p, [&](int i) { return Ld4{theta[i], dv[i], tc[i], (double)v.vkind[i]}; },
                            [&](int i, const Ld4& x) {
                              double t = x.a + alpha * x.b;
                              if (x.d == 0.0) {
                                if ((alpha == 1.0 && x.c == 0.0) || t < 0.0 ||
                                    (alpha >= amax && x.b < 0.0 && x.a <= -amax * x.b * (1.0 + 1e-12)))
                                  t = 0.0;
                              }
                              theta[i] = t;
                            });
  c.sync();
}

// Safeguarded 1-D Newton iteration for the root of phi'(alpha) (monotone, piecewise linear) on
// [0, amax], started at the full step alpha = 1.  eval(alpha, d1, d2) returns phi' and phi''.
template <class EVAL>
CAVE_HD double exact_step(EVAL&& eval, double psi0, double amax) {
  double alpha = 1.0, lo = 0.0, hi = amax, d1 = 0.0, d2 = 0.0;
  // an inexact search is enough for the outer Newton iteration: stop once the slope has dropped to a
  // tenth of its initial value (tighter tolerances cost ~40 % more evaluations for no fewer iterations)
  const double psitol = CAVE_PSITOL * fabs(psi0);
  for (int ls = 0; ls < 60; ++ls) {
    eval(alpha, d1, d2);
#ifdef CAVE_TRACE
    printf("   ls %d alpha %.6e psi %.3e (psi0 %.3e) curv %.3e  [%g, %g]\n", ls, alpha, d1, psi0, d2, lo, hi);
#endif
    if (fabs(d1) <= psitol) break;
    if (d1 < 0.0) {
      lo = alpha;
      if (alpha >= amax) break;  // a bound blocks: stay at the end of the segment
    } else hi = alpha;
    double an = (d2 > 0.0) ? alpha - d1 / d2 : (d1 < 0.0 ? 2.0 * alpha : 0.5 * (lo + alpha));
    if (hi < 1e299) { if (!(an > lo && an < hi)) an = 0.5 * (lo + hi); }
    else if (!(an > lo)) an = 2.0 * alpha;
    if (an > amax) an = amax;
    if (hi - lo <= 1e-15 * hi) break;
    alpha = an;
  }
  return alpha;
}

}  // namespace cave
#include "cone_dense.h"
#include "cone_rb.h"
namespace cave {

// w.H = M W M^T as a symmetric band (H[j * ldh + t] = H(j + t, j)), W_kk = weight(k).  Accumulated in 64-bit FIXED
// POINT (the zero bit pattern is shared with the doubles): integer adds are associative, so the sums -- and with them
// the whole projection -- do not depend on the order in which lanes and waves arrive; two launches give the same bits
// (the fp64 atomics spread results by 1e-7).  in_lds: the whole band fits the (idle) LDS ring window and is
// accumulated there with LDS atomics, then copied out; otherwise straight into the workspace.
template <class C, bool PM1, class W>
CAVE_HD void band_hessian(C& c, const SolveView& v, SolveWork& w, bool in_lds, W&& weight) {
  constexpr int NT = C::NT;
  const int p = v.p, d = v.d, ldh = w.ldh;
  const auto cptr = space_cast<1>(v.cptr);  // large-cone path: the view is global memory
  auto accumulate = [&](auto Hacc, auto add) {
    for (int idx = c.tid(); idx < p * ldh; idx += NT) Hacc[idx] = 0.0;
    c.sync();
    // Four coordinates per thread at a time, their column extents and first two entries requested
    // together: on the packed path the cone is read straight from the store (HBM / L2), and one
    // dependent load per entry would cost a full memory latency each.
    constexpr int G = 4;
    for (int base = c.tid(); base < d; base += G * NT) {
      uint32_t lo[G], hi[G], a0[G], a1[G];
      double wk[G], x0[G], x1[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int k = base + u * NT;
        const bool in = k < d;
        const int kc = in ? k : d - 1;
        lo[u] = cptr[kc];
        hi[u] = in ? cptr[kc + 1] : lo[u];
        wk[u] = in ? weight(kc) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const uint32_t last = cptr[d] > 0u ? cptr[d] - 1u : 0u;  // clamp: loads are unconditional
        const uint32_t e0 = lo[u] < last ? lo[u] : last, e1 = lo[u] + 1u < last ? lo[u] + 1u : last;
        csc_entry<PM1, 1>(v, e0, a0[u], x0[u]);
        csc_entry<PM1, 1>(v, e1, a1[u], x1[u]);
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (!(wk[u] > 1e-14) || hi[u] == lo[u]) continue;
        add(Hacc + a0[u] * ldh, wk[u] * x0[u] * x0[u]);
        if (hi[u] - lo[u] >= 2u) {
          add(Hacc + a1[u] * ldh, wk[u] * x1[u] * x1[u]);
          add(Hacc + (a0[u] * ldh + (a1[u] - a0[u])), wk[u] * x1[u] * x0[u]);  // columns are sorted: a0 < a1
        }
        for (uint32_t e1 = lo[u] + 2u; e1 < hi[u]; ++e1) {  // columns with more than two entries (cut rows)
          uint32_t a, b;
          double v1, v2;
          csc_entry<PM1, 1>(v, e1, a, v1);
          const double va = wk[u] * v1;
          add(Hacc + a * ldh, va * v1);
          for (uint32_t e2 = lo[u]; e2 < e1; ++e2) {
            csc_entry<PM1, 1>(v, e2, b, v2);
            add(Hacc + (b * ldh + (a - b)), va * v2);
          }
        }
      }
    }
  };
  const double hsc = w.hscale, hiv = w.hinv;
  if (in_lds) {
    auto Hl = space_cast<3>(w.bwin);
    accumulate(Hl, [&](decltype(Hl) q, double x) {
      c.atomic_add_i64_lds(reinterpret_cast<typename SpacePtr<long long, 3>::type>(q), (long long)llrint(x * hsc));
    });
    c.sync();
    auto Hq = reinterpret_cast<typename SpacePtr<long long, 3>::type>(Hl);
    for (int idx = c.tid(); idx < p * ldh; idx += NT) w.H[idx] = (double)Hq[idx] * hiv;
  } else {
    accumulate(w.H, [&](double* q, double x) { c.atomic_add_i64(reinterpret_cast<long long*>(q), (long long)llrint(x * hsc)); });
    c.sync();
    long long* Hq = reinterpret_cast<long long*>(w.H);
    for (int idx = c.tid(); idx < p * ldh; idx += NT) {
      const long long qv = Hq[idx];
      w.H[idx] = (double)qv * hiv;
    }
  }
}

// does the context carry the diet layout of the packed operator (BlockCtx<4, true>)?
template <class C, class = void> struct ctx_diet : std::false_type {};
template <class C> struct ctx_diet<C, std::void_t<decltype(C::DIET_OK)>> : std::bool_constant<C::DIET_OK> {};

// does the context carry the lite index structures (SoloCtx)?
template <class C, class = void> struct ctx_lite : std::false_type {};
template <class C> struct ctx_lite<C, std::void_t<decltype(C::LITE)>> : std::bool_constant<C::LITE> {};

// ---- large-cone path: the cone is read from GLOBAL memory (the packed store or the workspace), every dependent load
// costs a memory latency, so the loops below request the operands of several rows / columns before using any.
// Long rows of g = -M rc (more than kLongRow entries: TSP degree rows, cut rows): one wave per row, U chunks of 64
// entries in flight, two rows at a time (TSP-100: a 99-entry degree row takes one step, a 4700-entry cut row 19
// steps instead of 74 dependent ones); fixed summation order.  (Short rows keep the 4-lanes-per-row loop of gradient().)
template <class C, bool PM1>
CAVE_HD void gradient_long_rows_streamed(C& c, const SolveView& v, const double* rc, double* g) {
  constexpr int U = 4, WL = C::WL, NW = C::NWAVES;
  const int lane = c.lane_id();
  const auto mptr = space_cast<1>(v.mptr);
  const uint32_t last = mptr[v.p] > 0u ? mptr[v.p] - 1u : 0u;
  // Rows of more than kTeamRow entries (the large subtour cuts of a TSP-100 cone: up to 4 700) are SHARED by the waves,
  // below: one wave per row left the wave that drew two of them with 50 steps where the others had 31 -- the gradient
  // of the slowest instances, i.e. of the kernel.
  constexpr uint32_t kTeam = (NW > 1) ? kTeamRow : 0xffffffffu;
  for (int l0 = c.wave_id(); l0 < v.nlong; l0 += 2 * NW) {
    const int l1 = l0 + NW;
    const bool two = l1 < v.nlong;
    const int i0 = (int)v.longrow[l0], i1 = two ? (int)v.longrow[l1] : i0;
    const uint32_t lo0 = mptr[i0], m0 = mptr[i0 + 1] - lo0;
    const uint32_t lo1 = mptr[i1], m1 = two ? mptr[i1 + 1] - lo1 : 0u;
    const uint32_t n0 = m0 > kTeam ? 0u : m0, n1 = m1 > kTeam ? 0u : m1;
    double part0 = 0.0, part1 = 0.0;
    for (uint32_t off = 0; off < n0 || off < n1; off += (uint32_t)(U * WL)) {
      uint32_t col[2][U];
      double val[2][U], x[2][U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t t = off + (uint32_t)(u * WL + lane);
        const uint32_t e0 = lo0 + t < last ? lo0 + t : last, e1 = lo1 + t < last ? lo1 + t : last;  // clamped, unconditional
        csr_entry<PM1, 1>(v, e0, col[0][u], val[0][u]);
        csr_entry<PM1, 1>(v, e1, col[1][u], val[1][u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { x[0][u] = rc[col[0][u]]; x[1][u] = rc[col[1][u]]; }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t t = off + (uint32_t)(u * WL + lane);
        part0 -= t < n0 ? val[0][u] * x[0][u] : 0.0;
        part1 -= t < n1 ? val[1][u] * x[1][u] : 0.0;
      }
    }
    part0 = c.wave_sum(part0);
    part1 = c.wave_sum(part1);
    if (lane == 0) {
      if (m0 <= kTeam) g[i0] = part0;
      if (two && m1 <= kTeam) g[i1] = part1;
    }
  }
  if constexpr (NW > 1) {
    for (int l = 0; l < v.nteam; ++l) {  // (workgroup-uniform: every wave walks the short list of such rows)
      const int i = (int)v.teamrow[l];
      const uint32_t lo = mptr[i], n = mptr[i + 1] - lo;
      double part = 0.0;
      for (uint32_t off = (uint32_t)(c.wave_id() * U * WL); off < n; off += (uint32_t)(NW * U * WL)) {
        uint32_t col[U];
        double val[U], x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t t = off + (uint32_t)(u * WL + lane);
          csr_entry<PM1, 1>(v, lo + t < last ? lo + t : last, col[u], val[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = rc[col[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t t = off + (uint32_t)(u * WL + lane);
          part -= t < n ? val[u] * x[u] : 0.0;
        }
      }
      part = c.reduce_sum(part);  // lanes, then waves, in a fixed order
      if (c.tid() == 0) g[i] = part;
    }
  }
}

// Short rows (at most kLongRow entries): ONE LANE PER ROW, R rows per thread in flight -- the extents of all of them,
// then their first four entries, then the operands: three memory latencies per R * NT rows, where the four-lanes-per-row
// loop of gradient() pays three per NT / 4 rows (30x30 grid, 900 rows on 128 threads: 29 passes, 134 k cycles per
// gradient -- 10 % of the kernel).  The four partial sums of a row and their order are those of that loop and its quad
// reduction (entries s, s + 4, .. in partial sum s; (p0 + p1) + (p2 + p3)), so the bits are the same.  A real call:
// its ~100 live registers would otherwise spill inside the Newton iteration.
template <class C, bool PM1>
CAVE_NOINLINE void gradient_short_rows_streamed(C& c_, const SolveView& v_, const double* rc, double* g) {
  CtxLocal<C> cl(c_);
  C& c = cl.c;
  const SolveView v = v_;
  constexpr int R = 8, Q = 4, NT = C::NT;
  static_assert(C::TEAM == Q, "partial sums follow the quad reduction of gradient()");
  const int p = v.p, tid = c.tid();
  const auto mptr = space_cast<1>(v.mptr);
  const uint32_t last = mptr[p] > 0u ? mptr[p] - 1u : 0u;
  for (int i0 = tid; i0 < p; i0 += R * NT) {
    uint32_t lo[R], n[R], col[R][Q];
    double val[R][Q], x[R][Q];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int i = i0 + u * NT;
      const int ic = i < p ? i : p - 1;
      lo[u] = mptr[ic];
      n[u] = mptr[ic + 1] - lo[u];
      if (i >= p || n[u] > kLongRow) n[u] = 0u;  // (long rows: gradient_long_rows_streamed)
    }
    uint32_t any = 0u;
#pragma unroll
    for (int u = 0; u < R; ++u) any |= n[u];
    if (any == 0u) {  // long or empty rows only (TSP degree / cut rows): nothing to fetch
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int i = i0 + u * NT;
        if (i < p && mptr[i + 1] == mptr[i]) g[i] = 0.0;
      }
      continue;
    }
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const uint32_t e = lo[u] + (uint32_t)q < last ? lo[u] + (uint32_t)q : last;  // clamped, unconditional
        csr_entry<PM1, 1>(v, e, col[u][q], val[u][q]);
      }
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
      for (int q = 0; q < Q; ++q) x[u][q] = rc[col[u][q]];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int i = i0 + u * NT;
      double part[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        part[q] = 0.0;
        if ((uint32_t)q < n[u]) part[q] -= val[u][q] * x[u][q];
      }
      for (uint32_t off = (uint32_t)Q; off < n[u]; off += (uint32_t)Q) {  // rows with more than four entries
        uint32_t c2[Q];
        double v2[Q], x2[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const uint32_t e = lo[u] + off + (uint32_t)q < last ? lo[u] + off + (uint32_t)q : last;
          csr_entry<PM1, 1>(v, e, c2[q], v2[q]);
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) x2[q] = rc[c2[q]];
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if (off + (uint32_t)q < n[u]) part[q] -= v2[q] * x2[q];
      }
      if (i < p && (n[u] != 0u || mptr[i + 1] - mptr[i] <= kLongRow)) g[i] = (part[0] + part[1]) + (part[2] + part[3]);
    }
  }
}

// out[k] = base[k] + sgn * (M^T th)[k], G columns per thread at a time (extents, then entries, then operands)
template <class C, bool PM1, int CP = 0>
CAVE_HD void gather_mt_streamed(C& c, const SolveView& v, const float* base, const double* th, double sgn, double* out) {
  constexpr int G = 4, E = 3, NT = C::NT;
  const int d = v.d;
  const auto cptr = space_cast<CP>(v.cptr);  // CP = 1: large-cone path; 0: diet layout (column pointers in LDS)
  const uint32_t last = cptr[d] > 0u ? cptr[d] - 1u : 0u;
  for (int k0 = c.tid(); k0 < d; k0 += G * NT) {
    uint32_t lo[G], cnt[G], a[G][E];
    double x[G][E], t[G][E], r[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int k = k0 + u * NT;
      const bool in = k < d;
      const int kc = in ? k : d - 1;
      lo[u] = cptr[kc];
      cnt[u] = in ? cptr[kc + 1] - lo[u] : 0u;
      r[u] = base ? (double)base[kc] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t ee = lo[u] + (uint32_t)e < last ? lo[u] + (uint32_t)e : last;
        csc_entry<PM1, 1>(v, ee, a[u][e], x[u][e]);
      }
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
      for (int e = 0; e < E; ++e) t[u][e] = th[a[u][e]];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int k = k0 + u * NT;
      if (k >= d) continue;
      double acc = r[u];
#pragma unroll
      for (int e = 0; e < E; ++e)
        if ((uint32_t)e < cnt[u]) acc += sgn * x[u][e] * t[u][e];
      for (uint32_t e0 = (uint32_t)E; e0 < cnt[u]; e0 += 4u) {  // columns with more than E entries (edges inside
        uint32_t var[4];                                         // several cuts): four at a time, loads first -- one
        double val[4], tv[4];                                    // entry per trip paid two memory latencies per entry
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
          const uint32_t ee = lo[u] + e0 + j < last ? lo[u] + e0 + j : last;
          csc_entry<PM1, 1>(v, ee, var[j], val[j]);
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) tv[j] = th[var[j]];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j)
          if (e0 + j < cnt[u]) acc += sgn * val[j] * tv[j];
      }
      out[k] = acc;
    }
  }
  c.sync();
}

// STREAMED: the large-cone path (cone arrays in global memory)
template <class C, bool PM1, bool STREAMED = false>
CAVE_HD void gradient_any(C& c, const SolveView& v, const double* rc, double* g) {
#if defined(CAVE_GPU_CODE)
  if constexpr (ctx_lite<C>::value) { lite_gradient(c, c.lite, v.p, rc, g); return; }
#endif
  gradient<C, PM1, STREAMED>(c, v, rc, g);
}
template <class C, bool PM1, bool STREAMED = false>
CAVE_HD void gather_any(C& c, const SolveView& v, const float* base, const double* th, double sgn, double* out) {
#if defined(CAVE_GPU_CODE)
  if constexpr (ctx_lite<C>::value) { lite_gather(c, c.lite, v.d, base, th, sgn, out); return; }
#endif
  if constexpr (STREAMED && C::WL > 1) gather_mt_streamed<C, PM1, 1>(c, v, base, th, sgn, out);
  else {
    if constexpr (ctx_diet<C>::value && PM1) {
      if (v.csc_far) { gather_mt_streamed<C, PM1>(c, v, base, th, sgn, out); return; }  // CSC in the store: batched loads
    }
    gather_mt<C, PM1>(c, v, base, th, sgn, out);
  }
}

// One Newton step = exact minimisation of the local quadratic model over the
// non-negativity constraints by a primal active-set inner loop (ratio test to
// the first blocking bound, fix it at zero, re-solve on the smaller face — the
// inner loop of Lawson-Hanson applied to the model), followed by an EXACT line
// search on the true piecewise-quadratic f along the feasible segment (and
// beyond it while no bound blocks): phi' is monotone piecewise-linear, so a
// safeguarded 1-D Newton iteration finds its root in a few O(d) passes.
// Iterates stay feasible, bound variables sit at exactly 0, and for a pure
// quadratic (no unit rows) the outer loop is a block active-set NNLS method.
//
// On return w.res holds the CLIPPED residual Pi(y - M^T theta).
//
// BAND = true (large-cone path): H is kept as a symmetric band (SolveWork::H, ldh = bw + 1) and the
// model systems are solved by solve_spd_band instead of the register-resident Gauss-Jordan.
template <class C, bool PM1, bool BAND>
CAVE_HD SolveResult solve_cone_impl(C& c, const SolveView& v, SolveWork& w, int max_iter, double tol) {
  const int NT = C::NT;
  const int p = v.p, d = v.d;
  SolveResult out;
  out.iters = 0;
  out.status = ST_OK;
  double* theta = w.theta;
  double* tc = w.ttry;  // working point of the inner loop
  double* r = w.res;    // UNCLIPPED residual y - M^T theta during the iteration
  double* rc = w.rc;    // its clipped image Pi(r)
  // Starting point: theta = 0, or (warm start) the multipliers a previous solve of the SAME cone ended with --
  // cones are static per instance and predictions drift slowly during training (src/dataset.py:72), so the old
  // active set is nearly right.  The projection is unique, so the result does not depend on the start.
  constexpr int RB = BAND ? 8 : 1;  // iterations of a strided loop whose loads are issued together (strided_batched)
  const bool warm = w.warm != nullptr;
  for (int i = c.tid(); i < p; i += NT) {
    double t0 = warm ? (double)w.warm[i] : 0.0;
    if (!(t0 == t0) || fabs(t0) > 1e30 || (!v.vkind[i] && t0 < 0.0)) t0 = 0.0;
    theta[i] = t0;
  }
  double yy = 0.0, ymax = 0.0;
  strided_batched<RB, C::NT>(c.tid(), d, [&](int k) { return (double)w.y[k]; }, [&](int k, double yk) {
    yy += yk * yk;
    r[k] = yk;
    ymax = fmax(ymax, fabs(yk));
  });
  yy = c.reduce_sum(yy);
  ymax = c.reduce_max(ymax);
  c.sync();
  double f = refresh_clipped<C, RB>(c, v, r, rc);
  // Rounding floor of the gradient test: g_i = -sum_k m_ik rc_k is a sum of terms of size up to |m_ik| max|y|, so a
  // projected gradient below a few ulps of (largest row 1-norm) * max|y| is zero to working precision.  Without it a
  // start that is already optimal to the float32 resolution of y (y in the polar of the cone: g0n ~ 6e-8) asked for
  // pgn <= 1e-11 * g0n, which no iterate can deliver -- flagged NOT_CONVERGED with the right projection in hand
  // (found by tools/fuzz/fuzz_gpu.py seed 701, round 3: 1 of ~270 k adversarial instances).
  double rmax1 = 0.0;
  for (int i = c.tid(); i < p; i += NT) {
    const uint32_t lo = v.mptr[i], hi = v.mptr[i + 1];
    double s1 = (double)(hi - lo);
    if constexpr (!PM1) {
      s1 = 0.0;
      // (long rows: a wave each, below -- one thread walking a 4 700-entry cut row of a TSP-100 cone through global
      //  memory was 0.5 ms of a 2.3 ms instance)
      if (C::WL == 1 || hi - lo <= kLongRow)
        for (uint32_t e = lo; e < hi; ++e) s1 += fabs((double)v.mval[e]);
    }
    rmax1 = fmax(rmax1, s1);
  }
  if constexpr (!PM1 && C::WL > 1) {
    for (int li = c.wave_id(); li < v.nlong; li += C::NWAVES) {
      const int i = (int)v.longrow[li];
      const uint32_t lo = v.mptr[i], hi = v.mptr[i + 1];
      double part = 0.0;
      for (uint32_t e0 = lo + (uint32_t)c.lane_id(); e0 < hi; e0 += 4u * (uint32_t)C::WL) {  // four entries in flight
        float mv[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) mv[j] = v.mval[e0 + j * (uint32_t)C::WL < hi ? e0 + j * (uint32_t)C::WL : hi - 1u];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j)
          if (e0 + j * (uint32_t)C::WL < hi) part += fabs((double)mv[j]);
      }
      rmax1 = fmax(rmax1, c.wave_sum(part));
    }
  }
  rmax1 = c.reduce_max(rmax1);
  const double gfloor = 8.0 * 2.220446049250313e-16 * rmax1 * ymax;
  double g0n = 0.0;
  if (warm && p > 0) {
    // the convergence test is relative to the projected gradient AT theta = 0 (what a cold start measures in its
    // first iteration), not at the warm point, where it is already small
    gradient_any<C, PM1, BAND>(c, v, rc, w.g);
    double gm = 0.0;
    for (int i = c.tid(); i < p; i += NT) gm = fmax(gm, fabs(v.vkind[i] ? w.g[i] : fmin(w.g[i], 0.0)));
    g0n = c.reduce_max(gm);
    gather_any<C, PM1, BAND>(c, v, w.y, theta, -1.0, r);
    f = refresh_clipped<C, RB>(c, v, r, rc);
  }
  const int ldh = w.ldh;
  bool dense_on = false;
  if constexpr (BAND) dense_on = w.dn.on;
  bool hgen_on = false;
  if constexpr (BAND) hgen_on = w.gen.on;
  bool tri = false;  // diet layout: H is the packed lower triangle
  if constexpr (ctx_diet<C>::value && !BAND) tri = w.tri;
  const int hsize = tri ? p * (p + 1) / 2 : p * ldh;
  if (!dense_on && !hgen_on) for (int idx = c.tid(); idx < hsize; idx += NT) w.H[idx] = 0.0;
  if constexpr (!BAND) for (int k = c.tid(); k < d; k += NT) w.wold[k] = 0.f;
  c.sync();
  double reg_rel = 1e-12;  // Levenberg shift relative to max diag(H); raised when a step stalls
  double cap07 = 1.0, sched01 = warm ? 0.0 : 1.0;  // 0.7^it and 0.1^it, kept as running products (pow() is ~300 instructions)
  bool converged = (p == 0);
  int it = 0;
  CAVE_T0();
  for (; p > 0 && it < max_iter; ++it, cap07 *= CAVE_MU_CAP, sched01 *= CAVE_BMU_DECAY) {
#if defined(CAVE_GPU_CODE) && defined(CAVE_LITE_TAIL_PRIO_IT)
    // step kernel: an instance still iterating after CAVE_LITE_TAIL_PRIO_IT rounds is the tail of its launch -- its wave
    // issues ahead of the pack waves it shares a SIMD with from here on
    if constexpr (ctx_lite<C>::value) if (it == CAVE_LITE_TAIL_PRIO_IT) __builtin_amdgcn_s_setprio(3);
#endif
    // gradient g = -M Pi(r) and projected-gradient norm
    CAVE_ACCF(22);
    gradient_any<C, PM1, BAND>(c, v, rc, w.g);
    CAVE_ACCF(16);
    // Zig-zag extrapolation.  On degenerate cones (duplicated generators, y inside the cone) the iteration can
    // settle into a two-cycle of active sets and crawl along a valley at a linear rate.  From iteration 10 on,
    // every second iteration first minimises f exactly along theta - theta(two iterations ago), the valley
    // direction; instances that converge normally never get here.
    if ((it & 1) == 0) {
      if (it >= 10) {
        double psi0 = 0.0, amax = 1e300;
        for (int i = c.tid(); i < p; i += NT) {
          const double di = theta[i] - w.told[i];
          w.dv[i] = di;
          psi0 += w.g[i] * di;
          if (!v.vkind[i] && di < 0.0) amax = fmin(amax, theta[i] / (-di));
        }
        psi0 = c.reduce_sum(psi0);
        amax = -c.reduce_max(-amax);
        c.sync();
        if (psi0 < 0.0 && amax > 0.0) {
          gather_any<C, PM1, BAND>(c, v, nullptr, w.dv, 1.0, w.q);
          const double alpha = exact_step([&](double a, double& d1, double& d2) { dphi<C, RB>(c, v, r, w.q, a, &d1, &d2); }, psi0, amax);
          for (int i = c.tid(); i < p; i += NT) {
            double t = theta[i] + alpha * w.dv[i];
            if (!v.vkind[i] && t < 0.0) t = 0.0;
            theta[i] = t;
          }
          c.sync();
          gather_any<C, PM1, BAND>(c, v, w.y, theta, -1.0, r);
          f = refresh_clipped<C, RB>(c, v, r, rc);
          gradient_any<C, PM1, BAND>(c, v, rc, w.g);
        }
      }
      strided_batched<RB, C::NT>(c.tid(), p, [&](int i) { return theta[i]; }, [&](int i, double t) { w.told[i] = t; });
      c.sync();
    }
    CAVE_ACCF(17);
    double pgmax = 0.0;
    strided_batched<RB, C::NT>(c.tid(), p, [&](int i) { return Ld3{w.g[i], theta[i], (double)v.vkind[i]}; },
                               [&](int, const Ld3& x) {
                                 const double pg = (x.c != 0.0 || x.b > 0.0) ? x.a : fmin(x.a, 0.0);
                                 pgmax = fmax(pgmax, fabs(pg));
                               });
    double pgn = c.reduce_max(pgmax);
    if (it == 0 && !(warm && g0n > 0.0)) g0n = pgn;
#ifdef CAVE_TRACE
    printf("it %d f %.10e pgn %.6e\n", it, f, pgn);
#endif
    CAVE_ACC(2);
    // Nearly-zero residual (y inside or on the cone): the caller's `rnorm < 1e-7` inside test
    // (src/cave.py:218) needs rnorm itself resolved, so the gradient test is tightened.
    const double tol_it = (f < 1e-8 * yy) ? 1e-4 * tol : tol;
    // Residual floor: rnorm = sqrt(2 f) <= 4.5e-8 settles that test for good (this iterate is in the cone, so
    // the true distance is smaller still) and bounds the error of proj by 9e-8 -- points inside a cone with
    // degenerate multipliers otherwise creep towards f = 0 at a linear rate.  The floor is absolute only for
    // |y| >= 1: a prediction of tiny norm (|y| < 4.5e-8 has f <= 1e-15 at theta = 0) must still be projected,
    // because the cosine target only sees the direction of proj.
    if (!(pgn > tol_it * g0n) || pgn <= gfloor || f <= 1e-30 * yy || f <= 1e-15 * fmin(1.0, yy)) { converged = true; break; }
    if constexpr (BAND) {
      // SMOOTHED generalised Hessian H = M W M^T.  W_kk in [0,1] is the derivative of
      // the CHKS smoothing of the one-sided clip at scale mu (1/2 at the kink, -> the 0/1 activity D_kk
      // as mu -> 0); mu shrinks tenfold per iteration from 0.1*max|y| but stays above 0.03*max|y| times
      // the relative projected gradient, so it vanishes with the error (superlinear end game) while
      // coordinates resting exactly on their kink keep weight 1/2 instead of flipping every iteration.
      // Coordinates about to switch on thus resist the step before they do: with the binary D the
      // active set of a grid shortest-path cone grows one graph layer per iteration (60-100
      // iterations on a 30x30 grid), with W it takes 7-10.  The gradient and the line search stay exact,
      // so every step still decreases f and the limit is the same projection.  H is rebuilt per
      // iteration (O(sum of squared column counts) atomics).
      // When the whole band fits the (idle) LDS ring window -- dense reduced systems, p <= bw + 1: TSP --
      // it is accumulated there with LDS atomics and copied out; otherwise straight into the workspace.
      const bool in_lds = w.band_hot && p <= ldh;
      double mu = (it < 6) ? CAVE_BMU0 * ymax * sched01 : 0.0;
      mu = fmax(mu, CAVE_BMU_COEF * ymax * fmin(pgn / g0n, cap07));  // capped: see the fast path below
      const double inv_mu = mu > 0.0 ? 1.0 / mu : 0.0;
      auto weight = [&](int k) -> double { return band_weight(v.usign[k], r[k], inv_mu); };
      if (dense_on) dense_hessian<C, PM1>(c, v, weight, w.dn);  // whole matrix in LDS, fixed point (cone_dense.h)
      else if (hgen_on) {  // no bound rows, one-wave elimination: it builds the rows it needs itself (cone_band.h)
        w.gen.mu = inv_mu;
        w.gen.r = r;
      } else {
      band_hessian<C, PM1>(c, v, w, in_lds, weight);
      }
    } else {
    // generalised Hessian H = M W M^T, kept incrementally: H += (w_k - w_k_old) m_k m_k^T for the coordinates
    // whose weight changed.  Far from its kink a coordinate has the 0/1 activity D_kk = [Pi(r)_k != 0]; within
    // 4 mu of it the weight is the CHKS-smoothed step quantised to 1/16, mu = 0.1 max|y| * (relative projected
    // gradient).  So mu -> 0 with the error (the end game is the plain semismooth Newton method), few
    // coordinates are ever fractional, and a coordinate resting on its kink keeps weight 1/2 instead of
    // flipping the active set every iteration (SP 5x5: 6.6 -> 3.9 iterations on average, worst 21 -> 7;
    // TSP-20: 5.87 -> 5.75, instances needing >= 8 iterations 52 -> 8 of 1024).
    {
      // (capped by 0.7^it: tied to the gradient alone the scale can hold itself up -- a stalled iteration keeps mu
      // large, and a large mu keeps some cones from converging; the cap is far above pgn/g0n on every
      // instance that converges normally)
      const double mu = CAVE_MU_COEF * ymax * fmin(pgn / g0n, cap07);
      const double inv_mu = mu > 0.0 ? 1.0 / mu : 0.0;
      bool done = false;
#if defined(CAVE_GPU_CODE)
      if constexpr (ctx_lite<C>::value) { lite_hessian(c, c.lite, v, w, r, mu, inv_mu); done = true; }
#endif
      if (!done) {
      // Several waves add into H: the sums are taken in 64-bit FIXED POINT (integer adds are associative, so the result
      // does not depend on which wave's add arrives first -- floating-point LDS atomics made two launches of the 2- /
      // 4-wave shapes differ in the last bits on cones with general entries: VERDICT r3).  H is kept as doubles; it is
      // converted in place around the update (the scale is a power of two and |q| < 2^62: the round trip is exact).
      // One wave: its LDS atomics execute in lane order, the floating-point form is already reproducible.
      constexpr bool FIXED = C::NWAVES > 1;
      const double hsc = w.hscale, hiv = w.hinv;
      auto Hq = reinterpret_cast<typename SpacePtr<long long, 3>::type>(space_cast<3>(w.H));
      if constexpr (FIXED) {
        for (int idx = c.tid(); idx < hsize; idx += NT) Hq[idx] = (long long)llrint(w.H[idx] * hsc);
        c.sync();
      }
      auto hadd = [&](uint32_t idx, double x) {
        if constexpr (FIXED) c.atomic_add_i64_lds(Hq + idx, (long long)llrint(x * hsc));
        else c.atomic_add_f64(&w.H[idx], x);
      };
      auto new_weight = [&](int k) -> float {
        const uint8_t u = v.usign[k];
        if (u == 0) return 1.0f;
        if (u == 3) return 0.0f;
        const double t = (u == 2) ? r[k] : -r[k];  // > 0 on the side that carries residual
        const float z = (float)(t * inv_mu);  // the weight is a heuristic, quantised anyway: float is plenty
        if (mu > 0.0 && fabsf(z) < 4.0f) return floorf(8.0f * (1.0f + z / sqrtf(1.0f + z * z)) + 0.5f) * (1.0f / 16.0f);
        return t > 0.0 ? 1.0f : 0.0f;
      };
      bool far_done = false;
#if defined(CAVE_GPU_CODE)
      if constexpr (ctx_diet<C>::value && PM1) {
        if (v.csc_far) {
          // diet layout: the columns are read from the packed store (global memory).  G coordinates per thread at a
          // time, the first E entries of each requested before any is used (one memory round trip per pass; a TSP
          // column holds its two degree rows + the cuts through the edge: <= 7), the packed triangle of H.
          constexpr int G = 4, E = 8;
          const uint32_t last = v.cptr[d] > 0u ? v.cptr[d] - 1u : 0u;
          for (int k0 = c.tid(); k0 < d; k0 += G * NT) {
            uint32_t lo[G], cnt[G], ent[G][E];
            double dwv[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
              const int k = k0 + u * NT;
              lo[u] = 0; cnt[u] = 0; dwv[u] = 0.0;
              if (k < d) {
                const float wn = new_weight(k), wo = w.wold[k];
                if (wn != wo) {
                  w.wold[k] = wn;
                  dwv[u] = (double)wn - (double)wo;
                  lo[u] = v.cptr[k];
                  cnt[u] = v.cptr[k + 1] - lo[u];
                }
              }
            }
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
              for (int e = 0; e < E; ++e) {
                const uint32_t ee = lo[u] + (uint32_t)e < last ? lo[u] + (uint32_t)e : last;  // clamped, unconditional
                ent[u][e] = v.cvar[ee];
              }
#pragma unroll
            for (int u = 0; u < G; ++u) {
              if (cnt[u] == 0u) continue;
              const double dw = dwv[u];
              auto pair = [&](uint32_t x1, uint32_t x2) {  // H(a, b) += dw s1 s2, a > b (columns are sorted by row)
                const uint32_t a = x1 & 0x7fffu, b = x2 & 0x7fffu;
                const double vv = ((x1 ^ x2) & 0x8000u) ? -dw : dw;
                if (tri) hadd(tri_idx(a, b), vv);
                else { hadd(a * (uint32_t)ldh + b, vv); hadd(b * (uint32_t)ldh + a, vv); }
              };
              auto diag = [&](uint32_t x1) {
                const uint32_t a = x1 & 0x7fffu;
                hadd(tri ? tri_idx(a, a) : a * (uint32_t)ldh + a, dw);
              };
              if (cnt[u] <= (uint32_t)E) {  // the prefetched entries, statically indexed (registers)
                static_for<0, E>([&](auto e1c) {
                  constexpr int e1 = decltype(e1c)::value;
                  if ((uint32_t)e1 < cnt[u]) {
                    diag(ent[u][e1]);
                    static_for<0, e1>([&](auto e2c) { pair(ent[u][e1], ent[u][decltype(e2c)::value]); });
                  }
                });
              } else {  // a column longer than the prefetch (an edge inside more than six cuts): entry by entry
                for (uint32_t e1 = 0; e1 < cnt[u]; ++e1) {
                  const uint32_t x1 = v.cvar[lo[u] + e1];
                  diag(x1);
                  for (uint32_t e2 = 0; e2 < e1; ++e2) pair(x1, v.cvar[lo[u] + e2]);
                }
              }
            }
          }
          far_done = true;
        }
      }
#endif
      if (!far_done)
      for (int k = c.tid(); k < d; k += NT) {
        const float wn = new_weight(k);
        const float wo = w.wold[k];
        if (wn == wo) continue;
        w.wold[k] = wn;
        const double dw = (double)wn - (double)wo;
        uint32_t lo = v.cptr[k], hi = v.cptr[k + 1];
        for (uint32_t e1 = lo; e1 < hi; ++e1) {
          uint32_t a, b;
          double v1, v2;
          csc_entry<PM1>(v, e1, a, v1);
          const double va = dw * v1;
          hadd(tri ? tri_idx(a, a) : a * (uint32_t)ldh + a, va * v1);
          for (uint32_t e2 = lo; e2 < e1; ++e2) {
            csc_entry<PM1>(v, e2, b, v2);
            double vv = va * v2;
            if (tri) hadd(tri_idx(a, b), vv);
            else { hadd(a * (uint32_t)ldh + b, vv); hadd(b * (uint32_t)ldh + a, vv); }
          }
        }
      }
      if constexpr (FIXED) {
        c.sync();
        for (int idx = c.tid(); idx < hsize; idx += NT) w.H[idx] = (double)Hq[idx] * hiv;
      }
      }
    }
    }
    c.sync();
    CAVE_ACC(3);
    // ---- model minimisation over theta >= 0 (attempt 0: free every bound variable with
    //      a negative multiplier; attempt 1, only if that made no move: free the most negative one)
    bool moved = false;
#if defined(CAVE_GPU_CODE)
    // lite form: rows ordered [free | a few bound rows]: one elimination per iteration, active set on the Schur complement
    if constexpr (ctx_lite<C>::value && !BAND) moved = lite_model_step(c, v, w, theta, tc, reg_rel);
    else
#endif
    if (dense_on) {
      // dense form: ONE factorisation of the rows without bounds, the active-set loop on the Schur complement of
      // the others (cone_dense.h)
      if constexpr (BAND) moved = dense_model_step(c, v, w.dn, theta, w.g, tc, reg_rel);
    } else
    for (int attempt = 0; attempt < 2 && !moved; ++attempt) {
      double gmin = 0.0;
      if (attempt == 1) {
        double gl = 0.0;
        for (int i = c.tid(); i < p; i += NT)
          if (!v.vkind[i] && theta[i] <= 0.0) gl = fmin(gl, w.g[i]);
        gmin = -c.reduce_max(-gl);
        if (!(gmin < 0.0)) break;
      }
      strided_batched<RB, C::NT>(c.tid(), p, [&](int i) { return Ld3{theta[i], w.g[i], (double)v.vkind[i]}; },
                                 [&](int i, const Ld3& x) {
                                   const bool at_bound = x.c == 0.0 && x.a <= 0.0;
                                   const bool release = attempt == 0 ? (x.b < 0.0) : (x.b <= gmin);
                                   w.act[i] = (uint8_t)((at_bound && !release) ? 1 : 0);
                                   tc[i] = x.a;
                                   w.dv[i] = x.b;  // dv doubles as the model gradient at tc
                                 });
      c.sync();
      CAVE_ACCF(18);
      for (int inner = 0; inner <= p; ++inner) {
        // rhs: -model gradient on free rows, "go to zero" on fixed rows
        double* rhs = w.g2;
        strided_batched<RB, C::NT>(c.tid(), p, [&](int i) { return Ld3{(double)w.act[i], tc[i], w.dv[i]}; },
                                   [&](int i, const Ld3& x) { rhs[i] = x.a != 0.0 ? -x.b : -x.c; });
        c.sync();
        CAVE_ACC(4);
        if constexpr (BAND) {
#if defined(CAVE_GPU_CODE)
          if (w.band_wave && w.rb.on) rb_solve(c, v, w, rhs, reg_rel);  // half the rows (cone_rb.h)
          else if (w.band_wave) {
            solve_spd_band_wave<C::NWAVES>(c.lane_id(), c.wave_id(), w.H, w.bw, rhs, w.act, p, reg_rel, w.bwin, w.bfac, w.bz, w.step,
                                           &w.gen
#ifdef CAVE_STAMPS
                                  , c.st
#endif
              );
          } else
#endif
          if (w.band_hot)
            solve_spd_band<C, true>(c, w.H, w.bw, rhs, w.act, p, reg_rel, w.bwin, w.bfac, w.bz, w.step, w.bstg, w.bch);
          else
            solve_spd_band<C, false>(c, w.H, w.bw, rhs, w.act, p, reg_rel, w.bwin, w.bfac, w.bz, w.step, w.bstg, w.bch);
        }
        else {
          bool solved = false;
          if constexpr (ctx_diet<C>::value) {
            if (tri) { c.solve_spd_tri(w.H, rhs, w.act, p, reg_rel, w.step); solved = true; }
          }
          if (!solved) c.solve_spd(w.H, ldh, rhs, w.act, p, reg_rel, w.step);
        }
        c.sync();
        CAVE_ACC(5);
        // ratio test to the first blocking bound
        double amin = 2.0;
        strided_batched<RB, C::NT>(c.tid(), p, [&](int i) { return Ld4{(double)v.vkind[i], (double)w.act[i], tc[i], w.step[i]}; },
                                   [&](int, const Ld4& x) {
                                     if (x.a == 0.0 && x.b == 0.0) {
                                       const double t = x.c + x.d;
                                       if (t < 0.0) amin = fmin(amin, x.c / (x.c - t));
                                     }
                                   });
        amin = -c.reduce_max(-amin);
        CAVE_ACCF(19);
        const bool blocked = amin < 1.0;
        const double a = blocked ? fmax(amin, 0.0) : 1.0;
        // model gradient update  gm += a * H step  (only needed if another inner round follows)
        if (blocked) {
          for (int i = c.tid(); i < p; i += NT) {
            double s = 0.0;
            if constexpr (BAND) {
              const int j0 = i - w.bw > 0 ? i - w.bw : 0, j1 = i + w.bw < p - 1 ? i + w.bw : p - 1;
              for (int j = j0; j <= j1; ++j) s += band_at(w.H, ldh, i, j) * w.step[j];
            } else {
              if constexpr (ctx_lite<C>::value) {  // lower triangle only
                for (int j = 0; j <= i; ++j) s += w.H[i * ldh + j] * w.step[j];
                for (int j = i + 1; j < p; ++j) s += w.H[j * ldh + i] * w.step[j];
              } else if (tri) {
                for (int j = 0; j < p; ++j) s += w.H[tri_idx((uint32_t)i, (uint32_t)j)] * w.step[j];
              } else {
                for (int j = 0; j < p; ++j) s += w.H[i * ldh + j] * w.step[j];
              }
            }
            rhs[i] = s;  // rhs is dead until the next inner round
          }
          c.sync();
        }
        struct LdU { double tc, st, vk, act, dv, rh; };
        strided_batched<RB, C::NT>(c.tid(), p,
                                   [&](int i) { return LdU{tc[i], w.step[i], (double)v.vkind[i], (double)w.act[i], w.dv[i], rhs[i]}; },
                                   [&](int i, const LdU& x) {
                                     const double t = x.tc + x.st;
                                     double tn = x.tc + a * x.st;
                                     bool act = x.act != 0.0;
                                     if (x.vk == 0.0 && !act && blocked && t < 0.0 && x.tc <= a * (x.tc - t) * (1.0 + 1e-12)) {
                                       tn = 0.0;
                                       act = true;
                                       w.act[i] = 1;
                                     }
                                     if (act) tn = 0.0;
                                     tc[i] = tn;
                                     if (blocked) w.dv[i] = x.dv + a * x.rh;
                                   });
        c.sync();
        CAVE_ACCF(20);
        if (!blocked) break;
      }
      double mv = 0.0;
      strided_batched<RB, C::NT>(c.tid(), p, [&](int i) { return Ld2{tc[i], theta[i]}; },
                                 [&](int, const Ld2& x) { mv = fmax(mv, fabs(x.a - x.b)); });
      moved = c.reduce_max(mv) > 0.0;
    }
    CAVE_ACC(4);
    if (!moved) { converged = !(pgn > 1e-6 * g0n) || pgn <= gfloor; break; }
    // ---- exact line search on the true f along dv = tc - theta
    double psi0 = 0.0, amax = 1e300;
    strided_batched<RB, C::NT>(c.tid(), p, [&](int i) { return Ld4{tc[i], theta[i], w.g[i], (double)v.vkind[i]}; },
                               [&](int i, const Ld4& x) {
                                 const double di = x.a - x.b;
                                 w.dv[i] = di;
                                 psi0 += x.c * di;
                                 if (x.d == 0.0 && di < 0.0) amax = fmin(amax, x.b / (-di));
                               });
    if constexpr (ctx_lite<C>::value) {  // one interleaved pass for both
      double nm = -amax;
      c.reduce_sum_max(psi0, nm);
      amax = -nm;
    } else {
      psi0 = c.reduce_sum(psi0);
      amax = -c.reduce_max(-amax);  // >= 1 because tc is feasible
    }
    if (amax < 1.0) amax = 1.0;
    c.sync();
    CAVE_ACCF(21);
    if (!(psi0 < 0.0)) { converged = !(pgn > 1e-6 * g0n) || pgn <= gfloor; break; }
    // q = M^T dv, so r(alpha) = r - alpha q.  When the cost dimension fits KREG coordinates per
    // thread, q and r stay in registers for the whole search and the residual update is fused in.
    double alpha, fn;
    bool fast = false;
    if constexpr (C::KREG > 0) fast = d <= C::KREG * NT;
    if (fast) {
      constexpr int K = C::KREG > 0 ? C::KREG : 1;
      double rk[K], qk[K];
      uint8_t uk[K];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int k = c.tid() + j * NT;
        rk[j] = 0.0; qk[j] = 0.0; uk[j] = 3;
        if (k < d) {
          rk[j] = r[k];
          uk[j] = v.usign[k];
          double q = 0.0;
#if defined(CAVE_GPU_CODE)
          if constexpr (ctx_lite<C>::value) {
            q = lite_col_dot(c.lite, k, w.dv);
          } else
#endif
          for (uint32_t e = v.cptr[k]; e < v.cptr[k + 1]; ++e) {
            uint32_t var;
            double val;
            csc_entry<PM1>(v, e, var, val);
            q += val * w.dv[var];
          }
          qk[j] = q;
        }
      }
      CAVE_ACC(6);
      alpha = exact_step([&](double a, double& d1, double& d2) {
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const double rr = rk[j] - a * qk[j];
          a1 -= clip_unit(rr, uk[j]) * qk[j];
          if (active_unit(rr, uk[j])) a2 += qk[j] * qk[j];
        }
        c.reduce_sum2(a1, a2);
        d1 = a1; d2 = a2;
      }, psi0, amax);
      CAVE_ACC(7);
      const bool fresh = (it & 7) == 7;  // periodic fresh residual bounds the drift of the increments
      double acc = 0.0;
      if (!fresh) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const int k = c.tid() + j * NT;
          if (k < d) {
            const double rn = rk[j] - alpha * qk[j];
            const double t = clip_unit(rn, uk[j]);
            r[k] = rn;
            rc[k] = t;
            acc += t * t;
          }
        }
      }
      update_theta(c, v, theta, tc, w.dv, alpha, amax);
      if (fresh) {
        gather_any<C, PM1, BAND>(c, v, w.y, theta, -1.0, r);
        fn = refresh_clipped(c, v, r, rc);
      } else {
        fn = 0.5 * c.reduce_sum(acc);
        c.sync();
      }
      CAVE_ACC(8);
    } else {
      gather_any<C, PM1, BAND>(c, v, nullptr, w.dv, 1.0, w.q);
      CAVE_ACC(6);
      alpha = exact_step([&](double a, double& d1, double& d2) { dphi<C, RB>(c, v, r, w.q, a, &d1, &d2); }, psi0, amax);
      update_theta<C, RB>(c, v, theta, tc, w.dv, alpha, amax);
      CAVE_ACC(7);
      if ((it & 7) == 7) gather_any<C, PM1, BAND>(c, v, w.y, theta, -1.0, r);
      else {
        strided_batched<RB, C::NT>(c.tid(), d, [&](int k) { return Ld2{r[k], w.q[k]}; },
                                   [&](int k, const Ld2& x) { r[k] = x.a - alpha * x.b; });
        c.sync();
      }
      fn = refresh_clipped<C, RB>(c, v, r, rc);
      CAVE_ACC(8);
    }
    // An exact line search along the Newton direction that no longer lowers f beyond
    // round-off means the Newton decrement is ~0: by the projection inequality
    // ||proj - proj*||^2 <= 2 (f - f*), so this is fp32-exact long before it triggers.
    const bool tiny_gain = !(f - fn > 1e-15 * f);
    f = fn;
    if (tiny_gain) {
      // Either we are done (projected gradient at round-off level), or the Newton direction
      // was dominated by a (near-)null direction of a rank-deficient Hessian: damp harder,
      // which turns the step towards steepest descent, and go on.
      // (The looser gradient test is only trusted once the step has been damped to 1e-6: with the 1e-12
      // shift a Hessian of condition 1e12+ gives a direction made of round-off, and "no gain along it"
      // says nothing about optimality.)
      // (A start that is optimal up to the float32 rounding of y -- g0n ~ 1e-9 of the gradient's natural scale -- makes
      //  the relative tests unreachable in double precision: with f stagnant along damped Newton directions, a projected
      //  gradient within 1e3 rounding floors of zero is accepted.  tools/fuzz/fuzz_gpu.py seed 3001, round 4: a 6 x 16
      //  +-1 cone with a duplicated row was flagged NOT_CONVERGED at pgn = 1e-12 with the projection exact to 2e-16.)
      if (reg_rel >= 1e-6 && (!(pgn > 1e-8 * g0n) || pgn <= 1e3 * gfloor)) { converged = true; ++it; break; }
      if (reg_rel >= 1e-2) { converged = !(pgn > 1e-8 * g0n) || pgn <= 1e3 * gfloor; ++it; break; }
      reg_rel *= 1e3;
    } else if (reg_rel > 1e-12) reg_rel *= 0.1;
  }
  // final residual straight from theta (the iteration updated r incrementally), clipped for the epilogue
  if (p > 0) {
    gather_any<C, PM1, BAND>(c, v, w.y, theta, -1.0, r);
    f = refresh_clipped<C, RB>(c, v, r, rc);
  }
  strided_batched<RB, C::NT>(c.tid(), d, [&](int k) { return rc[k]; }, [&](int k, double t) { r[k] = t; });
  c.sync();
  if (!converged) out.status = ST_NOT_CONVERGED;
  if (!(f == f) || !(yy == yy)) out.status = ST_BAD_INPUT;
  out.f = f;
  out.iters = it;
  return out;
}

template <class C, bool BAND = false>
CAVE_HD SolveResult solve_cone(C& c, const SolveView& v, SolveWork& w, int max_iter, double tol) {
  return v.pm1 ? solve_cone_impl<C, true, BAND>(c, v, w, max_iter, tol)
               : solve_cone_impl<C, false, BAND>(c, v, w, max_iter, tol);
}

// The large-cone path calls the Newton iteration as a REAL function: inlined next to the streaming scan and the
// cone build of the persistent large kernels it inherited (and added to) a register file that spilled 150-250
// VGPRs; as a function it gets its own allocation and only the call boundary saves registers.
template <class C>
CAVE_NOINLINE void solve_cone_band_call(C& c_, const SolveView& v, SolveWork& w, int max_iter, double tol, SolveResult* out) {
  *out = solve_cone<C, true>(c_, v, w, max_iter, tol);
}

// ------------------------------------------------- truncated interior-point projection (MODE_IPM)
//
// The reference's CaVE+ runs an interior-point QP solver (Clarabel) for max_iter = 3 iterations and uses the
// strictly interior iterate as the cone-aligned target (src/cave.py:213-214, 267-295).  Emulation (Clarabel is
// not in the image: parity unpinned, tested by properties): primal path-following on
//     min_{lam > 0}  1/2 || y - A^T lam ||^2  -  tau * sum_i log lam_i .
// The multiplier of a signed-unit row appears in one coordinate only and is eliminated in closed form: for a
// coordinate with a +e_k row and s = y_k - (M^T theta)_k,  mu = (s + sqrt(s^2 + 4 tau)) / 2 > 0  and the residual
// becomes rho(s) = (s - sqrt(s^2 + 4 tau)) / 2 -- the Chen-Harker-Kanzow-Smale smoothing of min(s, 0) (mirrored
// for -e_k).  What is left is smooth and strictly convex in the reduced multipliers theta:
//     F_tau(theta) = sum_k psi_tau(s_k)  -  tau * sum_{i inequality} log theta_i ,     psi_tau' = rho ,
//     grad = -M rho - tau / theta ,      Hessian = M diag(rho') M^T + tau diag(1 / theta^2) .
// The bounds theta_i >= 0 of the inequality rows are handled PRIMAL-DUAL (a primal barrier step jams at the
// boundary once tau is small): duals z_i > 0 with theta_i z_i = tau; one Newton step on the perturbed KKT system
//     (H + diag(z / theta)) dtheta = M rho + tau / theta ,   dz = tau / theta - z - (z / theta) dtheta ,
// fraction-to-the-boundary step lengths (0.995) for theta and z, then tau = 0.2 * mean(theta_i z_i) (Mehrotra-
// style centring from the current complementarity, also covering the unit rows through the same tau).
// Rows of a +a / -a pair and coordinates with both unit rows carry two multipliers whose barrier has no
// minimiser (only their difference enters); they are left free, as an interior-point code's regularisation would.
// On return w.res holds rho (so proj = y - rho = A^T lam of the iterate) and f = 1/2 ||rho||^2.
// BAND = true (large-cone path, round 3): the same steps on the band / dense forms of the Newton system -- the
// smoothed Hessian M diag(rho') M^T is accumulated by band_hessian (or dense_hessian: whole matrix in LDS), the barrier
// terms z_i / theta_i go onto its diagonal, and one band LDL^T (solve_spd_band / solve_spd_band_wave, no fixed rows)
// or one complete dense LDL^T gives the step; the weights rho' live in the search-direction scratch w.q (doubles).
template <class C, bool PM1, bool BAND = false>
CAVE_HD SolveResult solve_cone_ipm_impl(C& c, const SolveView& v, SolveWork& w, int steps) {
  const int NT = C::NT;
  const int p = v.p, d = v.d, ldh = w.ldh;
  double* theta = w.theta;
  double* r = w.res;
  double* rc = w.rc;
  using WT = std::conditional_t<BAND, double, float>;
  WT* wgt;
  if constexpr (BAND) wgt = w.q;
  else wgt = w.wold;
  SolveResult out;
  out.iters = 0;
  out.status = ST_OK;
  double yy = 0.0, ymax = 0.0;
  for (int k = c.tid(); k < d; k += NT) {
    yy += (double)w.y[k] * (double)w.y[k];
    ymax = fmax(ymax, fabs((double)w.y[k]));
  }
  yy = c.reduce_sum(yy);
  ymax = c.reduce_max(ymax);
  double tau = 0.1 * ymax * ymax;
  if (!(tau > 0.0)) tau = 1e-300;
  const double tau_min = fmax(1e-14 * ymax * ymax, 1e-300);
  double* z = w.ttry;
  uint32_t nineq = 0;
  for (int i = c.tid(); i < p; i += NT) {
    theta[i] = v.vkind[i] ? 0.0 : sqrt(tau);
    z[i] = v.vkind[i] ? 0.0 : sqrt(tau);
    w.act[i] = 0;
    nineq += v.vkind[i] ? 0u : 1u;
  }
  nineq = c.reduce_add_u32(nineq);
  c.sync();
  auto smooth_residual = [&](double t) {  // r -> rho (rc) and rho' (wgt); returns 1/2 ||rho||^2
    double acc = 0.0;
    for (int k = c.tid(); k < d; k += NT) {
      const uint8_t u = v.usign[k];
      const double s = r[k];
      const double q = sqrt(s * s + 4.0 * t);
      double rho = s, dr = 1.0;
      if (u == 1) { rho = 0.5 * (s - q); dr = 0.5 * (1.0 - s / q); }
      else if (u == 2) { rho = 0.5 * (s + q); dr = 0.5 * (1.0 + s / q); }
      else if (u == 3) { rho = 0.0; dr = 0.0; }
      rc[k] = rho;
      wgt[k] = (WT)dr;
      acc += rho * rho;
    }
    const double f = 0.5 * c.reduce_sum(acc);
    c.sync();
    return f;
  };
  int it = 0;
  for (; it < steps && p > 0; ++it) {
    gather_any<C, PM1, BAND>(c, v, w.y, theta, -1.0, r);
    smooth_residual(tau);
    gradient<C, PM1, BAND>(c, v, rc, w.g);  // g = -M rho
    if constexpr (BAND) {
      auto weight = [&](int k) -> double { return (double)wgt[k]; };
      if (w.dn.on) dense_hessian<C, PM1>(c, v, weight, w.dn);
      else band_hessian<C, PM1>(c, v, w, w.band_hot && p <= ldh, weight);
      c.sync();
      for (int i = c.tid(); i < p; i += NT) {
        double rhs = -w.g[i];
        if (!v.vkind[i]) {
          const double inv = 1.0 / theta[i];
          rhs += tau * inv;
          if (w.dn.on) w.dn.A[fold_base(p, (int)w.dn.pos[i])] += z[i] * inv;
          else w.H[i * ldh] += z[i] * inv;
        }
        if (w.dn.on) w.dn.z[w.dn.pos[i]] = rhs;
        else w.g2[i] = rhs;
      }
      c.sync();
      if (w.dn.on) {
        dense_factor(c, w.dn, p, 1e-12, p);
        dense_backsub(c, w.dn, p, p);
        for (int i = c.tid(); i < p; i += NT) w.step[i] = w.dn.x[w.dn.pos[i]];
      } else {
#if defined(CAVE_GPU_CODE)
        if (w.band_wave) solve_spd_band_wave<C::NWAVES>(c.lane_id(), c.wave_id(), w.H, w.bw, w.g2, w.act, p, 1e-12, w.bwin, w.bfac, w.bz, w.step, nullptr);
        else
#endif
        if (w.band_hot) solve_spd_band<C, true>(c, w.H, w.bw, w.g2, w.act, p, 1e-12, w.bwin, w.bfac, w.bz, w.step, w.bstg, w.bch);
        else solve_spd_band<C, false>(c, w.H, w.bw, w.g2, w.act, p, 1e-12, w.bwin, w.bfac, w.bz, w.step, w.bstg, w.bch);
      }
      c.sync();
    } else {
    for (int idx = c.tid(); idx < p * ldh; idx += NT) w.H[idx] = 0.0;  // (the zero bit pattern is shared with int64)
    c.sync();
    {
      constexpr bool FIXED = C::NWAVES > 1;  // several waves: fixed-point sums (see solve_cone_impl)
      const double hsc = w.hscale, hiv = w.hinv;
      auto Hq = reinterpret_cast<typename SpacePtr<long long, 3>::type>(space_cast<3>(w.H));
      auto hadd = [&](uint32_t idx, double x) {
        if constexpr (FIXED) c.atomic_add_i64_lds(Hq + idx, (long long)llrint(x * hsc));
        else c.atomic_add_f64(&w.H[idx], x);
      };
      for (int k = c.tid(); k < d; k += NT) {
        const double wk = (double)wgt[k];
        if (!(wk > 0.0)) continue;
        const uint32_t lo = v.cptr[k], hi = v.cptr[k + 1];
        for (uint32_t e1 = lo; e1 < hi; ++e1) {
          uint32_t a, b;
          double v1, v2;
          csc_entry<PM1>(v, e1, a, v1);
          const double va = wk * v1;
          hadd(a * ldh + a, va * v1);
          for (uint32_t e2 = lo; e2 < e1; ++e2) {
            csc_entry<PM1>(v, e2, b, v2);
            const double vv = va * v2;
            hadd(a * ldh + b, vv);
            hadd(b * ldh + a, vv);
          }
        }
      }
      c.sync();
      if constexpr (FIXED) {
        for (int idx = c.tid(); idx < p * ldh; idx += NT) w.H[idx] = (double)Hq[idx] * hiv;
        c.sync();
      }
    }
    for (int i = c.tid(); i < p; i += NT) {
      double rhs = -w.g[i];
      if (!v.vkind[i]) {
        const double inv = 1.0 / theta[i];
        rhs += tau * inv;
        w.H[i * ldh + i] += z[i] * inv;
      }
      w.g2[i] = rhs;
    }
    c.sync();
    c.solve_spd(w.H, ldh, w.g2, w.act, p, 1e-12, w.step);
    c.sync();
    }
    double ap = 1e300, ad = 1e300;
    for (int i = c.tid(); i < p; i += NT) {
      double dz = 0.0;
      if (!v.vkind[i]) {
        const double inv = 1.0 / theta[i];
        dz = tau * inv - z[i] - z[i] * inv * w.step[i];
        if (w.step[i] < 0.0) ap = fmin(ap, -theta[i] / w.step[i]);
        if (dz < 0.0) ad = fmin(ad, -z[i] / dz);
      }
      w.told[i] = dz;
    }
    ap = -c.reduce_max(-ap);
    ad = -c.reduce_max(-ad);
    const double alpha_p = fmin(1.0, 0.995 * ap), alpha_d = fmin(1.0, 0.995 * ad);
    double gap = 0.0;
    for (int i = c.tid(); i < p; i += NT) {
      theta[i] += alpha_p * w.step[i];
      z[i] += alpha_d * w.told[i];
      gap += theta[i] * z[i];
    }
    gap = c.reduce_sum(gap);
    tau = nineq > 0u ? 0.2 * gap / (double)nineq : 0.2 * tau;
    // floor: the smoothed problem must stay smooth enough for plain Newton steps (the smoothing error of the
    // iterate is ~sqrt(tau) = 1e-7 max|y| there, below float32 resolution)
    if (!(tau > tau_min)) tau = tau_min;
    c.sync();
  }
  gather_any<C, PM1, BAND>(c, v, w.y, theta, -1.0, r);
  const double f = smooth_residual(tau);
  for (int k = c.tid(); k < d; k += NT) r[k] = rc[k];
  c.sync();
  if (!(f == f) || !(yy == yy)) out.status = ST_BAD_INPUT;
  out.f = f;
  out.iters = it;
  return out;
}

template <class C>
CAVE_HD SolveResult solve_cone_ipm(C& c, const SolveView& v, SolveWork& w, int steps) {
  return v.pm1 ? solve_cone_ipm_impl<C, true>(c, v, w, steps) : solve_cone_ipm_impl<C, false>(c, v, w, steps);
}
// large-cone path (a real call, like solve_cone_band_call)
template <class C>
CAVE_NOINLINE void solve_cone_ipm_band_call(C& c_, const SolveView& v_, SolveWork& w_, int steps, SolveResult* out) {
  CtxLocal<C> cl(c_);
  C& c = cl.c;
  const SolveView v = v_;
  SolveWork w = w_;
  *out = v.pm1 ? solve_cone_ipm_impl<C, true, true>(c, v, w, steps) : solve_cone_ipm_impl<C, false, true>(c, v, w, steps);
  w_ = w;
}

// half bandwidth of M M^T in the reduced-row order: the widest span of reduced rows meeting in one
// column (CSC columns are sorted by reduced-row index)
template <class C>
CAVE_HD int band_halfwidth(C& c, const SolveView& v) {
  const uint32_t vmask = v.pm1 ? 0x7fffu : 0xffffu;
  double bw = 0.0;
  for (int k = c.tid(); k < v.d; k += C::NT) {
    const uint32_t lo = v.cptr[k], hi = v.cptr[k + 1];
    if (hi > lo + 1u) bw = fmax(bw, (double)((v.cvar[hi - 1] & vmask) - (v.cvar[lo] & vmask)));
  }
  return (int)c.reduce_max(bw);
}

// ------------------------------------------------------------------- epilogue

struct EpilogueOut {
  float* proj;    // [d] or null
  float* rnorm;   // scalar or null
  float* target;  // [d] or null
  float* loss;    // scalar or null
  float* grad;    // [d] or null   d loss_i / d pred
};

// y = sign*pred (LDS, float); res = clipped residual (fp64) or null when the
// projection was skipped (heuristic); avg = per-instance average normal or null.
// tvec: LDS scratch [d] doubles for the target.
template <class C>
CAVE_HD void epilogue(C& c, int mode, int d, float sign, float inner_ratio, bool empty_cone,
                      const float* y, const double* res, double f, const float* avg, double* tvec,
                      const EpilogueOut& o) {
  const int NT = C::NT;
  if (mode == MODE_AVG) {
    if (o.target) for (int k = c.tid(); k < d; k += NT) o.target[k] = avg[k];
    return;
  }
  const bool has_proj = (mode != MODE_HEURISTIC);
  float rn = 0.f;
  double pp = 0.0, ss = 0.0;
  for (int k = c.tid(); k < d; k += NT) {
    double yk = (double)y[k];
    ss += yk * yk;
    if (has_proj) {
      // empty cone: reference returns cp itself (src/cave.py:304-305)
      float pk = empty_cone ? y[k] : (float)(yk - res[k]);
      if (o.proj) o.proj[k] = pk;
      tvec[k] = (double)pk;
      pp += (double)pk * (double)pk;
    }
  }
  pp = c.reduce_sum(pp);
  ss = c.reduce_sum(ss);
  if (has_proj) {
    rn = empty_cone ? 0.f : (float)sqrt(2.0 * f);
    if (o.rnorm && c.tid() == 0) *o.rnorm = rn;
  }
  if (mode == MODE_PROJECT) return;
  c.sync();
  const double ns = sqrt(ss);
  const double ns_c = fmax(ns, kNormClamp);
  // target vector
  double tt = 0.0, st = 0.0;
  const double r = (double)inner_ratio;
  const double np_c = fmax(sqrt(pp), kNormClamp);
  const bool inside = rn < kInsideRnorm;  // src/cave.py:218
  for (int k = c.tid(); k < d; k += NT) {
    double t;
    if (mode == MODE_EXACT || mode == MODE_IPM) t = tvec[k] / np_c;              // :129, :211-214
    else if (mode == MODE_INNER) {
      double pn = tvec[k] / np_c;                                                // :211
      t = inside ? pn : (1.0 - r) * pn + r * (double)avg[k];                      // :216-219
    } else t = (1.0 - r) * ((double)y[k] / ns_c) + r * (double)avg[k];           // :202-204
    t = (double)(float)t;  // the reference target is a float32 tensor
    tvec[k] = t;
    tt += t * t;
    st += (double)y[k] * t;
  }
  tt = c.reduce_sum(tt);
  st = c.reduce_sum(st);
  c.sync();
  const double nt_c = fmax(sqrt(tt), kNormClamp);
  // F.cosine_similarity: sum_k (x/max(|x|,eps))_k (t/max(|t|,eps))_k  (src/cave.py:72)
  const double cosv = st / (ns_c * nt_c);
  if (o.loss && c.tid() == 0) *o.loss = (float)(1.0 - cosv);
  const double inv = 1.0 / (ns_c * nt_c);
  const double radial = (ns > kNormClamp) ? st / (ns_c * ns_c * ns * nt_c) : 0.0;
  for (int k = c.tid(); k < d; k += NT) {
    if (o.target) o.target[k] = (float)tvec[k];
    if (o.grad) {
      double dcos = tvec[k] * inv - radial * (double)y[k];  // d cos / d signed
      o.grad[k] = (float)(-(double)sign * dcos);            // loss = 1 - cos, signed = sign*pred
    }
  }
}

}  // namespace cave
