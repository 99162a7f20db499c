// cone_dense.h — Newton systems of DENSE reduced matrices too large for registers (large-cone path: TSP-100 has
// ~100 degree equalities + a handful of subtour cuts, and M W M^T over them is full).
//
// What the round-2 form cost on TSP-100 (profiles/r02_large_path_phase_shares.txt): the smoothed Hessian was
// accumulated with fp64 atomics (LDS, 16-64 lanes per address on the diagonal), copied to the global workspace, read
// back into an LDS window, factored, the factor copied out and streamed back for the substitution -- and every bound
// that blocked a step in the active-set inner loop started the factorisation over: at equal Newton counts the
// instances of one batch took between 1.7 and 7.2 ms.  Here:
//   * the whole matrix lives in LDS as the FOLDED upper triangle (row r and row p-1-r share one line of p+1 entries:
//     45 KB at p = 105 instead of 89 KB, so two workgroups fit a CU) and is factored in place; nothing of it ever
//     goes to global memory;
//   * it is accumulated in 64-bit FIXED POINT (integer adds are associative: the result does not depend on the
//     order in which the lanes arrive, two launches give the same bits), every lane owning a contiguous block of
//     columns so that the lanes of one instruction hit different diagonal entries;
//   * rows are eliminated in the order [rows that can never be at a bound (paired +a/-a rows: free multipliers) |
//     rows with theta >= 0].  Eliminating the first block leaves, in the trailing corner, the Schur complement
//     S = H_II - H_IF H_FF^-1 H_FI of the bound rows and, in z, the reduced right-hand side: the LDL^T is PAUSED
//     there.  The active-set inner loop of the Newton step (ratio test, fix the blocking row, solve again) then
//     runs on S alone -- nI x nI, a handful of rows, solved in registers -- and only its final answer is
//     substituted back through the factor of the first block.  One O(p^3) factorisation per Newton iteration however
//     many bounds block; the iterates of the bound rows are those of the full-space loop (the free rows are at
//     their optimum for every trial point either way).
// Semantics otherwise those of solve_spd_band: rows of H + reg_rel * max diag * I, a non-positive pivot drops its
// row (x_k = 0).
#pragma once
#include "cone_common.h"

namespace cave {

constexpr int kDenseMaxP = 128;     // the pivot block is held two columns per lane
constexpr int kDenseMaxBound = 32;  // bound rows: their Schur system is solved in registers (Ctx::solve_spd)
constexpr int kDenseNB = 4;         // pivots per elimination step

// Which reduced systems take this path: up to 128 rows whose band is full (p <= bw + 1) or too wide for the one-wave
// band elimination of cone_band.h (half bandwidth > 34); narrow bands keep the band solvers.
CAVE_HOSTDEV bool dense_shape(int p, int bw) { return p >= 1 && p <= kDenseMaxP && (p <= bw + 1 || bw >= 35); }

CAVE_HOSTDEV uint32_t fold_entries(int p) { return (uint32_t)((p + 1) / 2) * (uint32_t)(p + 1); }
// first entry (the diagonal) of row r of the folded upper triangle; entry (r, j >= r) sits at fold_base + (j - r)
CAVE_HD int fold_base(int p, int r) {
  const int h = (p + 1) / 2;
  return r < h ? r * (p + 1) : (p - 1 - r) * (p + 1) + (r + 1);
}
CAVE_HOSTDEV uint32_t dense_scratch_entries(int p) { return 2u * (uint32_t)kDenseNB * (uint32_t)(p + 4); }
// doubles of LDS the dense path needs for a reduced system of p rows, nI of them with bounds (+ 4 p bytes of indices)
CAVE_HOSTDEV uint64_t dense_lds_bytes(int p, int nI) {
  const uint64_t ldS = (uint64_t)(nI | 1);
  return 8ull * (fold_entries(p) + 3ull * (uint64_t)p + dense_scratch_entries(p) + (uint64_t)nI * ldS + 4ull * (uint64_t)nI) +
         4ull * (uint64_t)p + (uint64_t)nI + 128u;
}

// fixed-point scale for sums of products w * m_ak * m_bk, w in [0, 1]: |sum| <= vmax * rmax (largest entry times
// largest row 1-norm) must stay below 2^61
CAVE_HD double fixed_scale(double vmax, double rmax) {
  const double bound = fmax(vmax * rmax, 1e-300);
  int e = 0;
  frexp(bound, &e);  // bound < 2^e
  int sh = 61 - e;
  if (sh > 1000) sh = 1000;
  if (sh < -1000) sh = -1000;
  return ldexp(1.0, sh);
}

// elimination order: rows with free multipliers first, rows with bounds last (both in their stored order)
template <class C>
CAVE_HD void dense_order(C& c, const SolveView& v, DenseWork& dw, uint32_t* tmp) {
  const int p = v.p;
  const uint32_t nF = c.compact_mask_u8(v.vkind, p, 0xff, 1, tmp);
  c.sync();
  for (int q = c.tid(); q < (int)nF; q += C::NT) { dw.ord[q] = (uint16_t)tmp[q]; dw.pos[tmp[q]] = (uint16_t)q; }
  c.sync();
  const uint32_t nI = c.compact_mask_u8(v.vkind, p, 0xff, 0, tmp);
  c.sync();
  for (int q = c.tid(); q < (int)nI; q += C::NT) { dw.ord[nF + q] = (uint16_t)tmp[q]; dw.pos[tmp[q]] = (uint16_t)(nF + q); }
  c.sync();
  dw.nF = (int)nF;
  dw.nI = (int)nI;
  dw.ldS = (int)(nI | 1u);
}

// A = M W M^T (folded upper triangle, elimination order), accumulated in fixed point.
// (The pieces below are REAL calls, like the band solvers: inlined into the Newton iteration they inherit -- and add
//  to -- a register file that already spills; as functions each gets its own allocation.)
template <class C, bool PM1, class W>
CAVE_NOINLINE void dense_hessian(C& c_, const SolveView& v_, W weight, const DenseWork& dw_) {
  CtxLocal<C> cl(c_);
  C& c = cl.c;
  const SolveView v = v_;
  const DenseWork dw = dw_;
  constexpr int NT = C::NT;
  const int p = v.p, d = v.d;
  const int ne = (int)fold_entries(p);
  const auto cptr = space_cast<1>(v.cptr);  // large-cone path: the view is global memory
  auto Aq = space_cast<3>(reinterpret_cast<long long*>(dw.A));
  auto Ad = space_cast<3>(dw.A);
  auto pos = space_cast<3>(dw.pos);
  for (int idx = c.tid(); idx < ne; idx += NT) Aq[idx] = 0;
  c.sync();
  const double sc = dw.hscale;
  auto add = [&](int a, int b, double x) {  // a, b: positions
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    c.atomic_add_i64_lds(Aq + (fold_base(p, lo) + (hi - lo)), (long long)llrint(x * sc));
  };
  // Consecutive lanes take consecutive columns (coalesced loads: the cone is read from global memory -- the packed
  // store or the workspace), G columns per thread at a time with the column extents and the first E entries of each
  // requested together: one dependent load per entry would cost a full memory latency each.  Lanes whose columns
  // share a row (the edges of one TSP node) add to the same diagonal entry: the LDS serialises those adds (a few
  // hundred cycles per instruction, measured cheaper than the 8x line requests of a blocked column assignment), and
  // integer adds give the same sum in any order.
  // E = 8: an edge inside c subtour cuts has 2 + c entries; with E = 4 the entries past the fourth took a loop of
  // dependent global loads per PAIR, and the instances with three to five large cuts -- the slowest of a TSP-100 batch,
  // i.e. the kernel time -- spent 410 k cycles per Hessian there against 110 k for the median instance.
  constexpr int G = 4, E = 8;
  const uint32_t last = cptr[d] > 0u ? cptr[d] - 1u : 0u;
  for (int kb = c.tid(); kb < d; kb += G * NT) {
    uint32_t lo[G], cnt[G], a[G][E];
    double wk[G], x[G][E];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int k = kb + u * NT;
      const bool in = k < d;
      const int kc = in ? k : d - 1;
      lo[u] = cptr[kc];
      cnt[u] = in ? cptr[kc + 1] - lo[u] : 0u;
      wk[u] = in ? weight(kc) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t ee = lo[u] + (uint32_t)e < last ? lo[u] + (uint32_t)e : last;  // clamped: loads are unconditional
        csc_entry<PM1, 1>(v, ee, a[u][e], x[u][e]);
      }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (!(wk[u] > 1e-14) || cnt[u] == 0u) continue;
      int ap[E];
#pragma unroll
      for (int e = 0; e < E; ++e) ap[e] = (uint32_t)e < cnt[u] ? (int)pos[a[u][e]] : 0;
#pragma unroll
      for (int e1 = 0; e1 < E; ++e1) {
        if ((uint32_t)e1 >= cnt[u]) break;
        const double va = wk[u] * x[u][e1];
#pragma unroll
        for (int e2 = 0; e2 <= e1; ++e2) add(ap[e1], ap[e2], va * x[u][e2]);
      }
      for (uint32_t e1 = (uint32_t)E; e1 < cnt[u]; ++e1) {  // columns with more than E entries (edges inside several cuts)
        uint32_t a1, a2;
        double v1, v2;
        csc_entry<PM1, 1>(v, lo[u] + e1, a1, v1);
        const double va = wk[u] * v1;
        const int p1 = (int)pos[a1];
        for (uint32_t e2 = 0; e2 <= e1; ++e2) {
          csc_entry<PM1, 1>(v, lo[u] + e2, a2, v2);
          add(p1, (int)pos[a2], va * v2);
        }
      }
    }
  }
  c.sync();
  const double hi = dw.hinv;
  for (int idx = c.tid(); idx < ne; idx += NT) {
    const long long q = Aq[idx];
    Ad[idx] = (double)q * hi;
  }
  c.sync();
}

// g = -M rc summed COLUMN-wise (round 4).  The row-wise form gathers an 8-byte operand per entry from a line of its own
// -- the per-CU L1 serves about a line per cycle, 130 k gathers per pass of a TSP-100 cone: 78 us, the largest cut row on
// one wave -- while a pass over the columns reads rc and the column entries coalesced (the Hessian and M^T theta passes
// above take 25 - 65 us).  The sums go to g itself, in LDS, as 64-bit FIXED POINT (integer adds are associative: any
// arrival order gives the same bits), scaled by (largest |rc_k|) x (longest row) x (largest |entry|) to 2^61: an add
// rounds to 2^-61 of that bound, the sum of a 4 700-entry row to ~1e-12 of the largest residual -- the size of the
// gradient test's rounding floor.  g must live in LDS (p <= 128 on this path).
template <class C, bool PM1>
CAVE_NOINLINE void dense_gradient(C& c_, const SolveView& v_, const double* rc, double* g_) {
  CtxLocal<C> cl(c_);
  C& c = cl.c;
  const SolveView v = v_;
  constexpr int NT = C::NT;
  const int p = v.p, d = v.d;
  const auto cptr = space_cast<1>(v.cptr);
  auto Gq = space_cast<3>(reinterpret_cast<long long*>(g_));
  auto Gd = space_cast<3>(g_);
  double rmax = 0.0;
  strided_batched<8, NT>(c.tid(), d, [&](int k) { return rc[k]; }, [&](int, double x) { rmax = fmax(rmax, fabs(x)); });
  for (int i = c.tid(); i < p; i += NT) Gq[i] = 0;
  rmax = c.reduce_max(rmax);  // (a barrier: the zeroes are in place)
  const double sc = fixed_scale(rmax, v.gcol_bound), inv = 1.0 / sc;
  constexpr int G = 4, E = 4;
  const uint32_t last = cptr[d] > 0u ? cptr[d] - 1u : 0u;
  for (int kb = c.tid(); kb < d; kb += G * NT) {
    uint32_t lo[G], cnt[G], a[G][E];
    double rk[G], x[G][E];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int k = kb + u * NT;
      const bool in = k < d;
      const int kc = in ? k : d - 1;
      lo[u] = cptr[kc];
      cnt[u] = in ? cptr[kc + 1] - lo[u] : 0u;
      rk[u] = rc[kc];
    }
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t ee = lo[u] + (uint32_t)e < last ? lo[u] + (uint32_t)e : last;  // clamped: loads are unconditional
        csc_entry<PM1, 1>(v, ee, a[u][e], x[u][e]);
      }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (rk[u] == 0.0 || cnt[u] == 0u) continue;
      const double t = -rk[u] * sc;
#pragma unroll
      for (int e = 0; e < E; ++e)
        if ((uint32_t)e < cnt[u]) c.atomic_add_i64_lds(Gq + a[u][e], (long long)llrint(x[u][e] * t));
      for (uint32_t e = (uint32_t)E; e < cnt[u]; ++e) {  // columns with more than E entries (edges inside several cuts)
        uint32_t a1;
        double v1;
        csc_entry<PM1, 1>(v, lo[u] + e, a1, v1);
        c.atomic_add_i64_lds(Gq + a1, (long long)llrint(v1 * t));
      }
    }
  }
  c.sync();
  for (int i = c.tid(); i < p; i += NT) {
    const long long q = Gq[i];
    Gd[i] = (double)q * inv;
  }
  c.sync();
}

// Partial LDL^T: pivots 0 .. nF-1 of A + reg I (reg = reg_rel * largest diagonal entry), the right-hand side z
// eliminated alongside.  Afterwards rows < nF hold the rows of U (dinv: reciprocal pivots), rows >= nF the Schur
// complement, z[nF ..] the reduced right-hand side.  Four pivots per step: wave 0 eliminates the four pivot rows
// against each other in registers (lane t: columns k0 + t and k0 + t + 64) and publishes them and the multipliers;
// after one barrier every lane takes the same TWO COLUMNS of every trailing row, so the eight pivot-row operands are
// read once per step and a trailing row costs four broadcast reads, one read and one write of the target pair.
template <class C>
CAVE_NOINLINE void dense_factor(C& c_, const DenseWork& dw_, int p, double reg_rel, int npiv = -1) {
  CtxLocal<C> cl(c_);
  C& c = cl.c;
  const DenseWork dw = dw_;
  constexpr int NT = C::NT, NB = kDenseNB;
  const int nF = npiv >= 0 ? npiv : dw.nF;  // (npiv = p: complete factorisation, interior-point steps)
  auto A = space_cast<3>(dw.A);
  auto z = space_cast<3>(dw.z);
  auto dinv = space_cast<3>(dw.dinv);
  const int tid = c.tid();
  double md = 0.0;
  for (int i = tid; i < p; i += NT) md = fmax(md, A[fold_base(p, i)]);
  md = c.reduce_max(md);
  const double reg = reg_rel * md;
  for (int i = tid; i < p; i += NT) A[fold_base(p, i)] += reg;
  c.sync();
#if defined(CAVE_GPU_CODE)
  if constexpr (C::WL == 64) {
    const int lane = c.lane_id(), wave = c.wave_id();
    constexpr int NWV = NT / 64;
    const int pc = p + 4;
    auto scP = space_cast<3>(dw.scr), scQ = space_cast<3>(dw.scr) + NB * pc;
    CAVE_T0();
    for (int k0 = 0; k0 < nF; k0 += NB) {
      if (wave == 0) {
        double u[NB][2], zv[NB], inv[NB];
#pragma unroll
        for (int a = 0; a < NB; ++a) {  // all twelve loads, then the selects (pinned: see CAVE_PIN_F64)
          const int row = k0 + a;
          const int rb = fold_base(p, row < p ? row : p - 1);
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const int col = k0 + lane + 64 * g;
            const bool in = row < p && col >= row && col < p;
            u[a][g] = A[in ? rb + (col - row) : 0];
          }
          zv[a] = z[row < p ? row : p - 1];
        }
#pragma unroll
        for (int a = 0; a < NB; ++a) { CAVE_PIN_F64(u[a][0]); CAVE_PIN_F64(u[a][1]); CAVE_PIN_F64(zv[a]); }
#pragma unroll
        for (int a = 0; a < NB; ++a) {
          const int row = k0 + a;
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const int col = k0 + lane + 64 * g;
            const bool in = row < p && col >= row && col < p;
            u[a][g] = in ? u[a][g] : 0.0;
          }
          zv[a] = row < p ? zv[a] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < NB; ++a) {
          const double dk = readlane_f64(u[a][0], a);
          const bool ok = (k0 + a < nF) && dk > 1e-300;
          double iv = rcp_full(dk);  // v_rcp_f64 + two Newton steps (a third of the IEEE divide); no select in front
          iv = ok ? iv : 0.0;        // of it (cone_band.h): a dropped pivot gives inf / NaN there and 0 here
          inv[a] = iv;
#pragma unroll
          for (int b = a + 1; b < NB; ++b) {
            const double m = readlane_f64(u[a][0], b) * iv;
            u[b][0] = fma(-m, u[a][0], u[b][0]);
            u[b][1] = fma(-m, u[a][1], u[b][1]);
            zv[b] = fma(-m, zv[a], zv[b]);
          }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int t = lane + 64 * g;  // column k0 + t; t = p - k0, p - k0 + 1: zero columns
          if (t >= NB && t <= p - k0 + 1) {
            const bool real = t < p - k0;
#pragma unroll
            for (int a = 0; a < NB; ++a) {
              scP[a * pc + t] = real ? u[a][g] : 0.0;
              scQ[a * pc + t] = real ? u[a][g] * inv[a] : 0.0;
            }
          }
#pragma unroll
          for (int a = 0; a < NB; ++a) {  // the block's rows as the earlier pivots of the block have left them
            const int row = k0 + a, col = k0 + t;
            if (row < p && col >= row && col < p) A[fold_base(p, row) + (col - row)] = u[a][g];
          }
        }
        if (lane < NB && k0 + lane < p) {
          z[k0 + lane] = lane == 0 ? zv[0] : (lane == 1 ? zv[1] : (lane == 2 ? zv[2] : zv[3]));
          dinv[k0 + lane] = lane == 0 ? inv[0] : (lane == 1 ? inv[1] : (lane == 2 ? inv[2] : inv[3]));
        }
      }
      CAVE_ACCF(0);
      c.sync_lds();
      CAVE_ACCF(1);
      {
        const int cb = k0 + NB;       // first trailing row / column
        const int c0 = cb + 2 * lane; // this lane's columns c0, c0 + 1 (past p - 1: the zero columns)
        const int t0 = c0 - k0 < pc - 1 ? c0 - k0 : pc - 2;
        double p0[NB], p1[NB], zq[NB];
#pragma unroll
        for (int a = 0; a < NB; ++a) {
          p0[a] = scP[a * pc + t0];
          p1[a] = scP[a * pc + t0 + 1];
          zq[a] = z[k0 + a < p ? k0 + a : p - 1];
        }
#if defined(__HIPCC__) && !defined(CAVE_SIMT_EMUL) && !defined(CAVE_DENSE_NO_MFMA)
        // Rank-4 update of the trailing triangle on the matrix cores (round 4): v_mfma_f64_16x16x4_f64 takes a 16 x 16
        // tile C, the 16 x 4 block -Q^T of the multipliers and the 4 x 16 block P of the pivot rows, one instruction per
        // tile.  On MI355X the f64 MFMA has the VALU's flop rate -- the gain is in the LDS: a lane reads its two operands
        // and four entries of C and writes those four back (5 KB per tile), where the two-columns-per-lane form below
        // read four broadcast multipliers per ROW and lane, half of the lanes left of the diagonal idling (3.3x the
        // traffic; the update was 62 % of the factorisation and LDS-bound with two workgroups per compute unit).
        // Tiles (I <= J) of the trailing square go round the waves; A[l & 15][k = l >> 4], B[k = l >> 4][l & 15],
        // C: column l & 15, rows (l >> 4) + 4 i (the f64 layout: cdna_hip_programming.md).
        {
          typedef double v4d __attribute__((ext_vector_type(4)));
          const int nt = p - cb, T = (nt + 15) >> 4, ntile = T * (T + 1) / 2;
          const int lr = lane & 15, lg = lane >> 4;
          (void)c0; (void)t0; (void)p0; (void)p1;
          for (int ti = wave; ti < ntile; ti += NWV) {
            int I = 0, rem = ti;
            while (rem >= T - I) { rem -= T - I; ++I; }  // (wave-uniform: tile ti of the upper triangle, row-major)
            const int r0 = cb + 16 * I, cc0 = cb + 16 * (I + rem);
            const int ar = r0 + lr, bc = cc0 + lr;
            double av = -scQ[lg * pc + ((ar < p ? ar : p - 1) - k0)];  // (clamped, unconditional; rows / columns past
            double bv = scP[lg * pc + ((bc < p ? bc : p - 1) - k0)];    //  the matrix contribute zeros)
            av = ar < p ? av : 0.0;
            bv = bc < p ? bv : 0.0;
            v4d acc;
            int o[4];
            bool in[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = r0 + lg + 4 * i;
              in[i] = row < p && bc < p && bc >= row;
              o[i] = in[i] ? fold_base(p, row) + (bc - row) : 0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = A[o[i]];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = in[i] ? acc[i] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (in[i]) A[o[i]] = acc[i];
          }
        }
#else
        // RB trailing rows per pass, every load of the pass before its first store: rows are independent, and one
        // row at a time each iteration waited out its own LDS round trips (12 k cycles per block step on TSP-100)
        constexpr int RB = 4;
        for (int rr = cb + wave; rr < p; rr += RB * NWV) {
          double q[RB][NB], t0v[RB], t1v[RB];
          int o[RB];
          bool in0[RB], in1[RB];
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            const int r = rr + j * NWV;
            const int rc_ = r < p ? r : p - 1;
#pragma unroll
            for (int a = 0; a < NB; ++a) q[j][a] = scQ[a * pc + (rc_ - k0)];
            o[j] = fold_base(p, rc_) + (c0 - rc_);
            in0[j] = r < p && c0 >= r && c0 < p;
            in1[j] = r < p && c0 + 1 >= r && c0 + 1 < p;
            t0v[j] = A[in0[j] ? o[j] : 0];
            t1v[j] = A[in1[j] ? o[j] + 1 : 0];
          }
#pragma unroll
          for (int j = 0; j < RB; ++j) {
#pragma unroll
            for (int a = 0; a < NB; ++a) {
              t0v[j] = fma(-q[j][a], p0[a], t0v[j]);
              t1v[j] = fma(-q[j][a], p1[a], t1v[j]);
            }
          }
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            if (in0[j]) A[o[j]] = t0v[j];
            if (in1[j]) A[o[j] + 1] = t1v[j];
          }
        }
#endif
        for (int r = cb + tid; r < p; r += NT) {  // the right-hand side below the block
          double zz = z[r];
#pragma unroll
          for (int a = 0; a < NB; ++a) zz = fma(-scQ[a * pc + (r - k0)], zq[a], zz);
          z[r] = zz;
        }
      }
      CAVE_ACCF(9);
      c.sync_lds();
      CAVE_ACCF(5);
    }
    return;
  }
#endif
  // plain right-looking form (serial test build)
  for (int k = 0; k < nF; ++k) {
    const int kb = fold_base(p, k);
    const double dk = A[kb];
    const double inv = dk > 1e-300 ? 1.0 / dk : 0.0;
    if (tid == 0) dinv[k] = inv;
    const double zk = z[k];
    c.sync();
    for (int r = k + 1 + tid; r < p; r += NT) {
      const double m = A[kb + (r - k)] * inv;
      const int rb = fold_base(p, r);
      for (int j = r; j < p; ++j) A[rb + (j - r)] -= m * A[kb + (j - k)];
      z[r] -= m * zk;
    }
    c.sync();
  }
}

// x[q], q < nF, from the factor of the first block, the eliminated right-hand side and the given x[nF ..]
template <class C>
CAVE_NOINLINE void dense_backsub(C& c_, const DenseWork& dw_, int p, int npiv = -1) {
  CtxLocal<C> cl(c_);
  C& c = cl.c;
  const DenseWork dw = dw_;
  const int nF = npiv >= 0 ? npiv : dw.nF;
  auto A = space_cast<3>(dw.A);
  auto z = space_cast<3>(dw.z);
  auto x = space_cast<3>(dw.x);
  auto dinv = space_cast<3>(dw.dinv);
#if defined(CAVE_GPU_CODE)
  if constexpr (C::WL == 64) {
    // column oriented, on wave 0: lane l owns the partial sums of rows l and l + 64; every finished x_k is folded
    // into the rows above it with one fma per lane and row, so the loop-carried chain is a v_readlane pair and two
    // fp64 operations.  (Needs p <= 128.)
    if (c.wave_id() == 0) {
      const int lane = c.lane_id();
      const int r0 = lane, r1 = lane + 64;
      const int b0 = fold_base(p, r0 < p ? r0 : 0), b1 = fold_base(p, r1 < p ? r1 : 0);
      double acc0 = 0.0, acc1 = 0.0;
      for (int j = nF; j < p; ++j) {  // the bound rows' part of the solution is given
        const double xj = x[j];
        const double u0 = A[r0 < nF ? b0 + (j - r0) : 0], u1 = A[r1 < nF ? b1 + (j - r1) : 0];
        acc0 = fma(r0 < nF ? u0 : 0.0, xj, acc0);
        acc1 = fma(r1 < nF ? u1 : 0.0, xj, acc1);
      }
      for (int k = nF - 1; k >= 0; --k) {
        const double dk = dinv[k], zk = z[k];
        const double u0 = A[r0 < k ? b0 + (k - r0) : 0], u1 = A[r1 < k ? b1 + (k - r1) : 0];
        const double ak = (k & 64) ? readlane_f64(acc1, k & 63) : readlane_f64(acc0, k & 63);
        const double xk = dk * (zk - ak);
        acc0 = fma(r0 < k ? u0 : 0.0, xk, acc0);
        acc1 = fma(r1 < k ? u1 : 0.0, xk, acc1);
        if (lane == 0) x[k] = xk;
      }
    }
    c.sync_lds();
    return;
  }
#endif
  if (c.tid() == 0)
    for (int k = nF - 1; k >= 0; --k) {
      const int kb = fold_base(p, k);
      double s = z[k];
      for (int j = k + 1; j < p; ++j) s -= A[kb + (j - k)] * x[j];
      x[k] = dinv[k] * s;
    }
  c.sync();
}

// One model minimisation (the body of a Newton iteration between the Hessian and the line search) on the dense
// path.  theta: current multipliers, g: gradient (both in reduced-row order).  On return tc holds the minimiser of
// the model over theta >= 0 found by the active-set loop (as in solve_cone_impl: attempt 0 frees every bound
// variable with a negative gradient, attempt 1 -- only if that made no move -- the most negative one);
// returns whether tc differs from theta.  dw.A must hold the Hessian (dense_hessian).
template <class C>
CAVE_NOINLINE bool dense_model_step(C& c_, const SolveView& v_, const DenseWork& dw_, const double* theta, const double* g,
                              double* tc, double reg_rel) {
  CtxLocal<C> cl(c_);
  C& c = cl.c;
  const SolveView v = v_;
  const DenseWork dw = dw_;
  constexpr int NT = C::NT;
  const int p = v.p, nF = dw.nF, nI = dw.nI, ldS = dw.ldS;
  auto A = space_cast<3>(dw.A);
  auto z = space_cast<3>(dw.z);
  auto x = space_cast<3>(dw.x);
  auto ord = space_cast<3>(dw.ord);
  auto S = space_cast<3>(dw.S);
  auto sg = space_cast<3>(dw.sg);
  auto st = space_cast<3>(dw.st);
  auto ss = space_cast<3>(dw.ss);
  auto sr = space_cast<3>(dw.sr);
  auto sact = space_cast<3>(dw.sact);
  CAVE_T0();
  for (int q = c.tid(); q < p; q += NT) z[q] = -g[ord[q]];
  c.sync();
  dense_factor(c, dw, p, reg_rel);
  CAVE_ACC(11);
  // Schur complement of the bound rows as a full square (the register solver reads rows)
  for (int idx = c.tid(); idx < nI * nI; idx += NT) {
    const int i = idx / nI, j = idx - i * nI;
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    S[i * ldS + j] = A[fold_base(p, nF + lo) + (hi - lo)];
  }
  c.sync();
  bool moved = false;
  for (int attempt = 0; attempt < 2 && !moved; ++attempt) {
    double gmin = 0.0;
    if (attempt == 1) {
      double gl = 0.0;
      for (int i = c.tid(); i < nI; i += NT) {
        const int row = ord[nF + i];
        if (theta[row] <= 0.0) gl = fmin(gl, g[row]);
      }
      gmin = -c.reduce_max(-gl);
      if (!(gmin < 0.0)) break;
    }
    for (int i = c.tid(); i < nI; i += NT) {
      const int row = ord[nF + i];
      const bool at_bound = theta[row] <= 0.0;
      const bool release = attempt == 0 ? (g[row] < 0.0) : (g[row] <= gmin);
      sact[i] = (uint8_t)((at_bound && !release) ? 1 : 0);
      st[i] = theta[row];
      sg[i] = -z[nF + i];  // reduced model gradient at theta (the free rows at their optimum)
    }
    c.sync();
    for (int inner = 0; inner <= nI && nI > 0; ++inner) {
      for (int i = c.tid(); i < nI; i += NT) sr[i] = sact[i] ? -st[i] : -sg[i];
      c.sync();
#if defined(CAVE_GPU_CODE)
      // (the register Gauss-Jordan at the sizes this path admits -- nI <= kDenseMaxBound = 32 --, not the context's
      //  full dispatch up to 64 rows: a third of the code, half the registers)
      if constexpr (C::WL == 64) { if (c.wave_id() == 0) gj_solve_small<kDenseMaxBound>(c.lane_id(), dw.S, ldS, dw.sr, dw.sact, nI, reg_rel, dw.ss); }
      else
#endif
      c.solve_spd(dw.S, ldS, dw.sr, dw.sact, nI, reg_rel, dw.ss);
      c.sync();
      double amin = 2.0;
      for (int i = c.tid(); i < nI; i += NT)
        if (!sact[i]) {
          const double t = st[i] + ss[i];
          if (t < 0.0) amin = fmin(amin, st[i] / (st[i] - t));
        }
      amin = -c.reduce_max(-amin);
      const bool blocked = amin < 1.0;
      const double a = blocked ? fmax(amin, 0.0) : 1.0;
      if (blocked) {
        for (int i = c.tid(); i < nI; i += NT) {
          double s = 0.0;
          for (int j = 0; j < nI; ++j) s += S[i * ldS + j] * ss[j];
          sr[i] = s;
        }
        c.sync();
      }
      for (int i = c.tid(); i < nI; i += NT) {
        const double t = st[i] + ss[i];
        double tn = st[i] + a * ss[i];
        if (!sact[i] && blocked && t < 0.0 && st[i] <= a * (st[i] - t) * (1.0 + 1e-12)) {
          tn = 0.0;
          sact[i] = 1;
        }
        if (sact[i]) tn = 0.0;
        st[i] = tn;
        if (blocked) sg[i] += a * sr[i];
      }
      c.sync();
      if (!blocked) break;
    }
    for (int i = c.tid(); i < nI; i += NT) x[nF + i] = st[i] - theta[ord[nF + i]];
    c.sync();
    CAVE_ACC(10);
    dense_backsub(c, dw, p);
    CAVE_ACC(12);
    double mv = 0.0;
    for (int q = c.tid(); q < p; q += NT) {
      const int row = ord[q];
      const double t = q < nF ? theta[row] + x[q] : st[q - nF];
      tc[row] = t;
      mv = fmax(mv, fabs(t - theta[row]));
    }
    moved = c.reduce_max(mv) > 0.0;
    c.sync();
  }
  return moved;
}

}  // namespace cave
