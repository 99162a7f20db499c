// cone_band.h — SPD solve for reduced systems too large for registers (large-cone path).
//
// The generalised Hessian H = M D M^T has the sparsity of M M^T.  In the reduced-row order the
// cones of this domain give it a band: grid shortest-path cones (node-arc incidence rows in node
// order) have half bandwidth = grid width, TSP degree rows give a dense matrix (bw = p - 1, p ~ n).
// The solver is an LDL^T elimination of the band without pivoting, written against the Ctx interface:
// a ring window of bw+1 band rows (LDS when it fits) is updated by the whole workgroup, one barrier
// per pivot; eliminated rows stream out to `bfac`; the right-hand side is eliminated alongside; the
// back substitution streams `bfac` back in.
//
// Semantics are those of gj_solve (wave_prims.h): rows flagged `act` are identity rows (x = rhs),
// free rows are rows of H + reg_rel*max diag*I; a non-positive pivot drops its row/column (x_k = 0).
#pragma once
#include "cone_common.h"

namespace cave {

// entry (i, j) of the symmetric band, |i - j| <= bw
CAVE_HD double band_at(const double* Hb, int ld, int i, int j) {
  return (i >= j) ? Hb[j * ld + (i - j)] : Hb[i * ld + (j - i)];
}

// masked + shifted entry t of band row r as the elimination sees it
template <class PH, class PA>
CAVE_HD double band_row_entry(PH Hb, int ld, PA act, int p, double reg, int r, int t) {
  const int i = r + t;
  if (r >= p || i >= p) return 0.0;
  const bool fixed = act[r] || act[i];
  if (t == 0) return act[r] ? 1.0 : Hb[r * ld] + reg;
  return fixed ? 0.0 : Hb[r * ld + t];
}

struct TrueTag { static constexpr bool value = true; };
struct FalseTag { static constexpr bool value = false; };

// rows staged per chunk (see solve_spd_band): bounded by the per-thread prefetch registers
template <class C>
CAVE_HD int band_chunk_rows(int ld) {
  constexpr int RMAX = (C::NT >= 64) ? 16 : 4096;
  const int byregs = (RMAX * C::NT) / ld;
  return byregs < 32 ? byregs : 32;
}

// HOT: win, z, x, stg and act are in LDS (typed ds_* accesses, barriers that wait for LDS only), Hb and
// fac in the global workspace.  Nothing inside the two sequential loops waits for global memory: rows
// of H (then of the factor) are fetched a chunk of CH rows ahead into registers and parked in the LDS
// staging buffers `stg` [2][CH*ld] one chunk before they are needed; factor rows are stored and forgotten.
// (A real call, not inlined: inside the fully inlined Newton iteration its loops inherit a register file
// already spilling; as a function they get their own allocation.)
template <class C, bool HOT>
CAVE_NOINLINE void solve_spd_band(C& c_, const double* Hb_, int bw, const double* rhs, const uint8_t* act_, int p,
                            double reg_rel, double* win_, double* fac_, double* z_, double* x_, double* stg_, int CH) {
  CtxLocal<C> cl(c_);
  C& c = cl.c;
  constexpr int NT = C::NT;
  constexpr int RMAX = (NT >= 64) ? 16 : 4096;
  constexpr int HS = HOT ? 3 : 0;   // space of the hot arrays
  constexpr int GS = HOT ? 1 : 0;   // HOT is only used on the GPU path, where Hb / fac are workspace (global) memory
  const int ld = bw + 1;
  if (p <= 0) return;
  CAVE_T0();
  // `c` lives in memory here (a real call): read its fields once, the LDS-only barriers below are
  // compiler memory barriers and would otherwise reload them from scratch on every pivot
  const int tid = c.tid(), lane = c.lane_id(), wave = c.wave_id();
  auto Hb = space_cast<GS>(Hb_);
  auto fac = space_cast<GS>(fac_);
  auto act = space_cast<HS>(act_);
  auto win = space_cast<HS>(win_);
  auto z = space_cast<HS>(z_);
  auto x = space_cast<HS>(x_);
  auto stg = space_cast<HS>(stg_);
  double md = 0.0;
  uint32_t nfix = 0;
  for (int i = tid; i < p; i += NT) {
    if (!act[i]) md = fmax(md, Hb[i * ld]);
    else nfix++;
  }
  md = c.reduce_max(md);
  nfix = c.reduce_add_u32(nfix);
  const double reg = reg_rel * md;
  // right-hand side: fixed rows keep theirs, free rows move the fixed unknowns over
  for (int i = tid; i < p; i += NT) {
    double zi = rhs[i];
    if (nfix != 0u && !act[i]) {
      const int j0 = i - bw > 0 ? i - bw : 0, j1 = i + bw < p - 1 ? i + bw : p - 1;
      for (int j = j0; j <= j1; ++j)
        if (act[j]) zi -= ((i >= j) ? Hb[j * ld + (i - j)] : Hb[i * ld + (j - i)]) * rhs[j];
    }
    z[i] = zi;
  }
  if (bw == 0) {  // diagonal system
    for (int i = tid; i < p; i += NT) {
      const double dk = act[i] ? 1.0 : Hb[i] + reg;
      x[i] = (dk > 1e-300) ? z[i] / dk : 0.0;
    }
    c.sync();
    return;
  }
  const int csz = CH * ld;  // entries per staged chunk
  double regs[RMAX];
  // entries [e0, e0 + csz) of a global array (clipped at eend) -> registers / registers -> staging buffer b
  auto fetch = [&](decltype(Hb) src, int e0, int eend) {
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
      // unconditional load from a clamped index: a select around the load would make each load wait
      // for the previous one (entries past the range are parked but never read)
      const int e = e0 + tid + j * NT;
      regs[j] = src[e < eend ? e : eend - 1];
    }
  };
  auto fetch_h = [&](int r0) { fetch(Hb, r0 * ld, p * ld); };  // raw rows r0 .. r0+CH-1 of H (masked on insertion)
  auto park = [&](int b) {
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
      const int idx = tid + j * NT;
      if (idx < csz) stg[b * csz + idx] = regs[j];
    }
  };
  // ring window: band row r lives in slot r % ld; rows ld .. p-1 arrive through the staging buffers
  for (int idx = tid; idx < ld * ld; idx += NT) {
    const int r = idx / ld, t = idx - r * ld;
    win[idx] = band_row_entry(Hb, ld, act, p, reg, r, t);
  }
  const bool streaming = p > ld;
  if (streaming) {
    fetch_h(ld);
    park(0);
    fetch_h(ld + CH);  // chunk 1 stays in registers until chunk 0 starts being consumed
  }
  // One barrier per pivot.  Step k: (A) pivot k updates rows k+1 .. k+bw-1 of the window and z;
  // (B) row k-1, final since the previous barrier, is retired to the factor and its slot takes row
  // k+bw of H -- the one row pivot k reaches only in its diagonal entry, which the inserting thread
  // adjusts itself.  A and B touch disjoint slots.
  double inv_prev = 0.0;
  c.sync();
  CAVE_ACC(10);
  int nb_cached = -1, na_cached = -1, qn = 0, rn = 0, s_first = 0, t_first = 0, s_rest = 0, t_rest = 0;
  constexpr int U = 4;
  int ua[U], ub[U], ur[U];
  bool uok[U];
  const bool narrow = bw < NT && bw * (bw - 1) <= U * NT;  // every pair, z entry and row entry has its own thread
  int slot_k = 0;          // k % ld, kept incrementally
  int cpos = 0, cidx = 0;  // position of the row inserted at this step inside its chunk, chunk index
  // The body of a step is written load / compute / store: every LDS operand of the step (pivot, the first
  // U triangle pairs of this thread, its z entry, the retiring and the entering row entry) is requested
  // before the reciprocal pivot is formed, so the step costs about one LDS round trip plus the division
  // instead of one round trip per statement.  Remainders (wide bands) take the plain loops below.
  auto step = [&](const int k, auto steady_tag, auto narrow_tag) __attribute__((always_inline)) {
    // NARROW: bw < NT and the whole triangle fits the U pairs per thread -- no remainder loops at all
    constexpr bool NARROW = decltype(narrow_tag)::value;
    // STEADY: 2 <= k, k + 2 bw < p, not the first row of a staged chunk -- the band is full width, a row
    // retires and a row enters, nothing is recomputed or fetched: the common case, kept branch-light
    constexpr bool STEADY = decltype(steady_tag)::value;
    auto wk = win + slot_k * ld;
    const int nb = STEADY ? bw : (bw < p - 1 - k ? bw : p - 1 - k);
    // rows updated by phase A: s = 1 .. na (row k+bw is phase B's, except at k = 0 where it is already resident)
    const int na = STEADY ? bw - 1 : ((k == 0 || nb < bw) ? nb : bw - 1);
    if (!STEADY && (nb != nb_cached || na != na_cached)) {  // only at the start and in the last bw steps
      nb_cached = nb; na_cached = na;
      if (nb > 0) {
        qn = NT / nb; rn = NT - qn * nb;
        s_first = tid / nb; t_first = tid - s_first * nb;
      }
      // this thread's first U pairs of the triangle: operand offsets relative to the pivot row's slot
      int s0 = s_first, t0 = t_first;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uok[u] = (na > 0) && (s0 < na) && (t0 >= s0);
        const int sc = uok[u] ? s0 : 0, tc = uok[u] ? t0 : 0;  // clamped: loads are unconditional
        ua[u] = sc + 1;
        ub[u] = tc + 1;
        ur[u] = (sc + 1) * ld + (tc - sc);
        if (nb > 0) {
          s0 += qn;
          t0 += rn;
          if (t0 >= nb) { t0 -= nb; ++s0; }
        }
      }
      s_rest = s0; t_rest = t0;
    }
    // ---- loads
    const int base_k = slot_k * ld;
    const double dk = wk[0];
    const double zk = z[k];
    const double wbw = wk[bw];
    double pa[U], pb[U], pr[U];
    int poff[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int o = base_k + ur[u];
      if (o >= ld * ld) o -= ld * ld;
      poff[u] = o;
      pa[u] = wk[ua[u]];
      pb[u] = wk[ub[u]];
      pr[u] = win[o];
    }
    int s0 = s_rest, t0 = t_rest;
    const bool zown = 1 + tid <= nb;
    const double zw = wk[zown ? 1 + tid : 0];
    const double zz = z[zown ? k + 1 + tid : k];
    const bool retire = STEADY || k > 0;
    const bool ins = STEADY || (retire && streaming && (k - 1 + ld < p));
    auto wp = win + (slot_k == 0 ? bw : slot_k - 1) * ld;
    if (!STEADY && ins && cpos == 0) {
      // first row of chunk cidx: chunk cidx+1 (in registers) takes the buffer chunk cidx-1 has just left
      park((cidx + 1) & 1);
      fetch_h(ld + (cidx + 2) * CH);
    }
    auto src = stg + (cidx & 1) * csz + cpos * ld;
    // row rI = k + bw of H as the elimination sees it (band_row_entry on the staged raw row)
    auto entering = [&](int t) -> double {
      const int rI = k + bw, i = rI + t;
      if (!STEADY && (!ins || i >= p)) return 0.0;
      const double raw = src[t];
      return (t == 0) ? (act[rI] ? 1.0 : raw + reg) : ((act[rI] || act[i]) ? 0.0 : raw);
    };
    const bool town = retire && tid <= bw;
    const double rold = town ? wp[tid] : 0.0;
    double rnew = town ? entering(tid) : 0.0;
    // ---- compute
    const bool ok = dk > 1e-300;
    const double inv = ok ? 1.0 / dk : 0.0;
    // ---- stores
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (uok[u]) win[poff[u]] = pr[u] - pa[u] * inv * pb[u];
    if constexpr (!NARROW)
    while (s0 < na) {  // pairs beyond the first U per thread (wide bands), again U at a time: loads, then stores
      double qa[U], qb[U], qr[U];
      int qoff[U];
      bool qok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        qok[u] = (s0 < na) && (t0 >= s0);
        const int sc = qok[u] ? s0 : 0, tc = qok[u] ? t0 : 0;
        int slot = slot_k + sc + 1;
        if (slot >= ld) slot -= ld;
        qoff[u] = slot * ld + (tc - sc);
        qa[u] = wk[sc + 1];
        qb[u] = wk[tc + 1];
        qr[u] = win[qoff[u]];
        s0 += qn;
        t0 += rn;
        if (t0 >= nb) { t0 -= nb; ++s0; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (qok[u]) win[qoff[u]] = qr[u] - qa[u] * inv * qb[u];
    }
    if (zown) z[k + 1 + tid] = zz - zw * inv * zk;
    if constexpr (!NARROW)
      for (int sx = 1 + tid + NT; sx <= nb; sx += NT) z[k + sx] -= wk[sx] * inv * zk;
    if (retire) {
      if (town) {
        fac[(k - 1) * ld + tid] = (tid == 0) ? inv_prev : rold;
        if (tid == 0 && (STEADY || k + bw < p)) rnew -= wbw * inv * wbw;
        wp[tid] = rnew;
      }
      if constexpr (!NARROW)
        for (int t = tid + NT; t <= bw; t += NT) {
          fac[(k - 1) * ld + t] = wp[t];
          wp[t] = entering(t);
        }
      if (ins && ++cpos == CH) { cpos = 0; ++cidx; }
    }
    inv_prev = inv;
    if (++slot_k == ld) slot_k = 0;
    if constexpr (HOT) c.sync_lds();
    else c.sync();
  };
  if (!streaming) {
    // Dense system (p <= bw + 1: every row is resident in its own slot, nothing retires or enters): plain
    // right-looking LDL^T.  A 32-lane strip per row, rows dealt round-robin to the strips -- unit-stride
    // LDS accesses, no per-pair index arithmetic; the factor is copied out in one pass at the end.
    constexpr int TX = NT >= 64 ? 32 : 1, TY = NT / TX;
    const int tx = tid % TX, ty = tid / TX;
    bool blocked = false;
#if defined(CAVE_GPU_CODE)
    if constexpr (HOT && C::WL == 64) blocked = p <= 127 && 2 * CH * ld >= 8 * (p + 2);
    if constexpr (HOT && C::WL == 64) if (blocked) {
      // Four pivots per step, as in solve_spd_band_wave below: wave 0 eliminates the four pivot rows against each
      // other in registers (lane = column, two columns per lane: p <= 127) and publishes them and the multipliers
      // (component-major scratch in the idle staging buffers); after one barrier every wave takes whole trailing
      // rows, two adjacent entries per lane, four updates per entry from operands whose addresses it never has to
      // recompute.  One barrier pair per FOUR pivots and no dependent LDS round trip per entry (the row-at-a-time
      // loop below: ~4500 cycles per pivot on TSP-100).  Same fma sequence per entry: same bits.
      constexpr int NBD = 4, NWV = NT / 64;
      const int pc = p + 2;
      auto scP = stg, scQ = stg + NBD * pc;
      for (int k0 = 0; k0 < p; k0 += NBD) {
        if (wave == 0) {
          double u[NBD][2], zv[NBD], inv[NBD];
#pragma unroll
          for (int a = 0; a < NBD; ++a) {
            const int row = k0 + a;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int col = k0 + lane + 64 * g, off = col - row;
              const bool in = row < p && off >= 0 && col < p;
              const double v = win[in ? row * ld + off : 0];
              u[a][g] = in ? v : 0.0;
            }
            const double zr = z[row < p ? row : p - 1];
            zv[a] = row < p ? zr : 0.0;
          }
#pragma unroll
          for (int a = 0; a < NBD; ++a) {
            const double d = readlane_f64(u[a][0], a);
            const double iv = (d > 1e-300) ? 1.0 / d : 0.0;
            inv[a] = iv;
#pragma unroll
            for (int b = a + 1; b < NBD; ++b) {
              const double m = readlane_f64(u[a][0], b) * iv;
              u[b][0] = fma(-m, u[a][0], u[b][0]);
              u[b][1] = fma(-m, u[a][1], u[b][1]);
              zv[b] = fma(-m, zv[a], zv[b]);
            }
          }
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const int t = lane + 64 * g;  // column k0 + t; t = p - k0 is the zero column
            if (t >= NBD && t <= p - k0) {
#pragma unroll
              for (int a = 0; a < NBD; ++a) {
                scP[a * pc + t] = u[a][g];
                scQ[a * pc + t] = u[a][g] * inv[a];
              }
            }
#pragma unroll
            for (int a = 1; a < NBD; ++a) {  // rows k0+1 .. k0+3 as the later pivots have left them
              const int row = k0 + a, off = k0 + t - row;
              if (row < p && off >= 0 && k0 + t < p) win[row * ld + off] = u[a][g];
            }
          }
          if (lane < NBD && k0 + lane < p) z[k0 + lane] = lane == 0 ? zv[0] : (lane == 1 ? zv[1] : (lane == 2 ? zv[2] : zv[3]));
        }
        c.sync_lds();
        for (int r = k0 + NBD + wave; r < p; r += NWV) {
          const int tr = r - k0, c0 = r + 2 * lane;
          if (c0 < p) {
            const int o = r * ld + 2 * lane, t0 = c0 - k0;
            double q[NBD], p0[NBD], p1[NBD];
#pragma unroll
            for (int a = 0; a < NBD; ++a) {
              q[a] = scQ[a * pc + tr];
              p0[a] = scP[a * pc + t0];
              p1[a] = scP[a * pc + t0 + 1];
            }
            double r0 = win[o], r1 = win[o + 1];
#pragma unroll
            for (int a = 0; a < NBD; ++a) {
              r0 = fma(-q[a], p0[a], r0);
              r1 = fma(-q[a], p1[a], r1);
            }
            win[o] = r0;
            win[o + 1] = r1;
          }
        }
        for (int r = k0 + NBD + tid; r < p; r += NT) {
          double zz = z[r];
#pragma unroll
          for (int a = 0; a < NBD; ++a) zz = fma(-scQ[a * pc + (r - k0)], z[k0 + a < p ? k0 + a : p - 1], zz);
          z[r] = zz;
        }
        c.sync_lds();
      }
    }
#endif
    if (!blocked)
    for (int k = 0; k < p; ++k) {
      auto wk = win + k * ld;
      const double dk = wk[0];
      const double zk = z[k];
      const double inv = (dk > 1e-300) ? 1.0 / dk : 0.0;
      const int nb = p - 1 - k;
      for (int sft = 1 + ty; sft <= nb; sft += TY) {
        auto row = win + (k + sft) * ld;
        const double ls = wk[sft] * inv;
        for (int tt = sft + tx; tt <= nb; tt += TX) row[tt - sft] -= ls * wk[tt];
      }
      for (int sx = 1 + tid; sx <= nb; sx += NT) z[k + sx] -= wk[sx] * inv * zk;
      if constexpr (HOT) c.sync_lds();
      else c.sync();
    }
    for (int idx = tid; idx < p * ld; idx += NT) {
      const int k = idx / ld, t = idx - k * ld;
      const double v = win[idx];
      fac[idx] = (t == 0) ? ((v > 1e-300) ? 1.0 / v : 0.0) : v;
    }
  } else {
  for (int k = 0; k < p;) {
    if (streaming && k >= 2 && k + 2 * bw < p && cpos != 0) {
      int kend = k + (CH - cpos);  // up to the end of the staged chunk
      if (kend > p - 2 * bw) kend = p - 2 * bw;
      if (narrow) for (; k < kend; ++k) step(k, TrueTag{}, TrueTag{});
      else for (; k < kend; ++k) step(k, TrueTag{}, FalseTag{});
    } else {
      step(k, FalseTag{}, FalseTag{});
      ++k;
    }
  }
  {
    auto wp = win + ((p - 1) % ld) * ld;
    for (int t = tid; t <= bw; t += NT) fac[(p - 1) * ld + t] = (t == 0) ? inv_prev : wp[t];
  }
  }
  c.sync();  // full barrier: the factor rows are in (workgroup-visible) global memory now
  CAVE_ACC(11);
  // back substitution  x_k = inv_k * (z_k - sum_s fac[k][s] * x_{k+s}): factor rows come back through the
  // staging buffers, CH rows per barrier, and the first wave does the sequential part without barriers
  // chunk j holds rows klo .. khi (ascending), khi = p-1 - j*CH
  auto fetch_f = [&](int khi) {
    const int klo = khi - CH + 1 > 0 ? khi - CH + 1 : 0;
    fetch(fac, klo * ld, (khi + 1) * ld);
  };
  fetch_f(p - 1);
  int b = 0;
  double xw = 0.0;  // register window of the narrow-band back substitution
  for (int khi = p - 1; khi >= 0; khi -= CH, b ^= 1) {
    park(b);
    if (khi - CH >= 0) fetch_f(khi - CH);
    if constexpr (HOT) c.sync_lds();
    else c.sync();
    if (wave == 0 && C::WL > 1 && bw < C::WL) {
      // narrow band: the window x_{k+1} .. x_{k+bw} lives in registers (lane s-1 holds x_{k+s}) and moves up
      // one lane per row, so the loop-carried chain is one wave reduction, not an LDS round trip
      const int klo = khi - CH + 1 > 0 ? khi - CH + 1 : 0;
      if (khi == p - 1) xw = 0.0;
      int k = khi;
      // two rows' factor entries in flight
      auto row_entry = [&](int kk) -> double { return (kk >= klo && lane < bw) ? stg[b * csz + (kk - klo) * ld + 1 + lane] : 0.0; };
      // everything that does not depend on x -- the factor row, its reciprocal pivot, z_k -- is requested one
      // row ahead, so the loop-carried chain is just the wave reduction and the shift
      auto pivot_entry = [&](int kk) -> double { return kk >= klo ? stg[b * csz + (kk - klo) * ld] : 0.0; };
      auto z_entry = [&](int kk) -> double { return kk >= klo ? z[kk] : 0.0; };
      double fnext = row_entry(k), f0next = pivot_entry(k), znext = z_entry(k);
      for (; k >= klo; --k) {
        const double fcur = fnext, f0 = f0next, zk = znext;
        fnext = row_entry(k - 1);
        f0next = pivot_entry(k - 1);
        znext = z_entry(k - 1);
        const double part = c.wave_sum(fcur * xw);  // entries past the matrix end multiply x = 0
        const double xk = f0 * (zk - part);
        if (lane == 0) x[k] = xk;
        xw = c.wave_shift_up(xw);
        if (lane == 0) xw = xk;
      }
    } else if (wave == 0) {
      constexpr int WL = C::WL;
      const int klo = khi - CH + 1 > 0 ? khi - CH + 1 : 0;
      for (int k = khi; k >= klo; --k) {
        auto fk = stg + b * csz + (k - klo) * ld;
        const int nb = bw < p - 1 - k ? bw : p - 1 - k;
        double part = 0.0;
        for (int s = 1 + lane; s <= nb; s += WL) part += fk[s] * x[k + s];
        part = c.wave_sum(part);
        if (lane == 0) x[k] = fk[0] * (z[k] - part);
        if constexpr (HOT) c.wave_fence_lds();
        else c.wave_fence();
      }
    }
  }
  c.sync();
  CAVE_ACC(12);
}

// smoothed weight of coordinate k (see solve_cone_impl): derivative of the CHKS smoothing of the one-sided clip,
// 1/2 (1 + z / sqrt(1 + z^2)) with z = t / mu.  Takes 1 / mu (0: the binary weight) and a reciprocal square root: the
// weight is evaluated once per coordinate and iteration (TSP-100: 4950 of them; four per band row in band_gen_rows),
// and sqrt + two divisions were a quarter of the Hessian phase's instructions.
CAVE_HD double band_weight(uint8_t u, double rk, double inv_mu) {
  if (u == 0) return 1.0;
  if (u == 3) return 0.0;
  const double t = (u == 2) ? rk : -rk;  // > 0 on the side that carries residual
  if (inv_mu > 0.0) {
    const double zz = t * inv_mu;
#if defined(__HIPCC__) && !defined(CAVE_SIMT_EMUL)
    return 0.5 * (1.0 + zz * rsqrt(1.0 + zz * zz));
#else
    return 0.5 * (1.0 + zz / sqrt(1.0 + zz * zz));
#endif
  }
  return (t > 0.0) ? 1.0 : 0.0;
}

// ---- sizing of the one-wave form below (shared with the host code: cone_instance.h, cave_hip.hip)
constexpr int kBandBlock = 4;      // pivots eliminated per step
constexpr int kBandWaveDuos = 5;   // duos per lane
constexpr int kBandWaveRegs = 20;  // prefetch registers per lane (factor rows of the back substitution)

constexpr int kBandGroup = 32;     // rows built per call of the row producer
// Polls before a wave stops waiting for the other one.  Never reached in a working hand-over; if it is (a broken or
// starved protocol), the wave sets the instance's FAILURE WORD (word 4 of the flag area) and goes on, so that every wave
// still reaches its exit; the caller turns the word into CAVE_ST_NOT_CONVERGED (solve_and_finish) -- a wave that fell
// through has computed on rows the other wave had not delivered.  (Test builds shrink the limit and withhold a flag:
// tests/test_simt_emul.py.)
#ifndef CAVE_BAND_SPIN_LIMIT
#define CAVE_BAND_SPIN_LIMIT (1 << 22)
#endif
constexpr int kBandSpinLimit = CAVE_BAND_SPIN_LIMIT;
// The flag words are relaxed atomic LDS accesses ordered by compiler barriers only (CAVE_FLAG_LOAD / CAVE_FLAG_STORE +
// CAVE_WAVE_ORDER): valid because a wave's LDS operations execute in issue order and the two waves share one LDS --
// i.e. NOT in threadgroup-split mode (tgsplit), which these kernels are never compiled for.

CAVE_HOSTDEV int band_wave_duos(int bw) {
  int nd = 0;
  for (int len = 1; len <= bw; ++len) nd += (len + 1) / 2;
  return nd;
}
CAVE_HOSTDEV bool band_wave_fits(int bw, int p) {
  // Half bandwidths below kBandBlock stay on the team form.
  if (bw < kBandBlock || p <= bw + 1) return false;
  if (bw + 2 > 64) return false;  // one lane per entry of a row
  return band_wave_duos(bw) <= 64 * kBandWaveDuos;
}
// ring row stride: odd, so that consecutive rows start on different banks
CAVE_HOSTDEV int band_wave_stride(int bw) { return ((bw + 1) & 1) ? bw + 1 : bw + 2; }
// operand scratch of one block step: the pivot rows and the multipliers, kBandBlock entries per column, columns
// kBandBlock .. bw + kBandBlock + 1 (the last one is a zero column: second operand of a duo without a second entry)
CAVE_HOSTDEV uint32_t band_wave_scratch(int bw) { return 2u * (uint32_t)kBandBlock * (uint32_t)(bw + kBandBlock + 2); }
// row stride of the factor ring of the back substitution (an even stride would spread the per-lane reads over all
// banks -- 31 gives 4-way conflicts -- but measured no faster on the 30x30 batch, and costs LDS)
CAVE_HOSTDEV int band_wave_ring_stride(int bw) { return bw + 1; }
// Rows of the ring the elimination works in: the bw + kBandBlock rows a step touches plus at least one group the
// producer is ahead, a multiple of kBandGroup (a group never wraps; neither does a pivot block).
CAVE_HOSTDEV int band_wave_ring_rows(int bw) { return (bw + kBandBlock + 2 * kBandGroup - 1) / kBandGroup * kBandGroup; }
// LDS entries of the wave form (SolveWork::bwin): [row ring | two entries: the four words the producer and the
// eliminator meet through | operand scratch when p is too small for x to hold it]; the back substitution reuses the
// front of the region for its 64-row ring of factor rows (+ one zero entry).
CAVE_HOSTDEV uint32_t band_wave_flags_at(int bw) {  // (behind the row ring and behind the factor ring + its zero entry)
  const uint32_t a = (uint32_t)band_wave_ring_rows(bw) * (uint32_t)band_wave_stride(bw);
  const uint32_t b = 64u * (uint32_t)band_wave_ring_stride(bw) + 1u;
  return a > b ? a : b;
}
CAVE_HOSTDEV uint32_t band_wave_region(int bw, int p) {
  return band_wave_flags_at(bw) + 3u + ((uint32_t)p < band_wave_scratch(bw) ? band_wave_scratch(bw) : 0u);  // (3 doubles: four flag words + the failure word)
}

#if defined(CAVE_GPU_CODE)
// ------------------------------------------------------------------ narrow bands on ONE wave
// What the team form above costs on a 30x30 grid (bw = 30, p = 900; rocprof + the stamp build): every pivot is a
// workgroup barrier plus ~200 instructions in EACH of the four waves for 465 window updates -- the kernel is
// bound by instruction issue, not by LDS or HBM, and two resident workgroups per CU only slow each other down.
// Here ONE wave runs the whole elimination: no barriers (the LDS operations of a wave execute in order), the
// update triangle is dealt out as "duos" (two adjacent entries of one window row: one ds_read_b64 for the
// multiplier, two ds_read2_b64 for the pivot-row and target operands, one ds_write2_b64), the reciprocal pivot is
// v_rcp_f64 + two Newton steps, and all bw + 1 rows a pivot reaches are resident when it is processed, so the
// steady state has no special cases: rows past the end of the matrix enter as zeros.
// The back substitution is column oriented: lane (j mod 64) owns the partial sum of row j, every finished x_k is
// folded into the bw rows above it with one fma per lane, and the loop-carried chain is a v_readlane pair and three
// fp64 operations instead of a wave reduction per row (~470 -> ~60 cycles per row).
// Same semantics as solve_spd_band (identity rows, shift, dropped pivots).  Needs win[(bw+1) * stride],
// stg[2 * (bw+1)^2], z[p], x[p], act[p] in LDS and p > bw + 1.
__device__ __forceinline__ double rcp_full(double d) {  // 1/d to double accuracy for d > 0
  double inv = __builtin_amdgcn_rcp(d);
  inv = fma(fma(-d, inv, 1.0), inv, inv);
  inv = fma(fma(-d, inv, 1.0), inv, inv);
  return inv;
}

template <class T>
__device__ __forceinline__ T* uniform_ptr(T* q) {  // arguments of a real call arrive in VGPRs: make them scalar again
  const uint64_t v = (uint64_t)q;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (T*)(((uint64_t)hi << 32) | lo);
}

// Rows first .. first + n - 1 (n <= 64: lane l builds row first + l) of H = M W M^T into LDS at dst, row stride
// `stride` -- the band of a cone without bound rows is never materialised (BandGen, cone_common.h).  One lane owns one
// row and adds its contributions in a fixed order: deterministic, no atomics.  The generator's fields are read once
// into registers, the cone arrays are typed global pointers, and a row's entries are taken four at a time: their
// coordinates' weights and column extents, then up to three entries of each column, are requested together -- three
// dependent memory levels per batch instead of one per entry.  A REAL call: inlined into the elimination (which already
// spills) the in-flight operands cost 2 ms on the 30x30 batch.
// The same for the red-black Schur complement (cone_rb.h): row q from its RECIPE -- records (coordinate, coordinate, red
// row, offset, signs) written once per call -- and this iteration's weights: the records of a row, then every weight
// they name, are requested together (two memory levels per twelve records).
__device__ __forceinline__ void band_gen_rows_rb(const RbWork* rb_v, const int lane, const int nB, const int first, const int n,
                                                 typename SpacePtr<double, 3>::type dst, const int stride) {
  const RbWork* rb = uniform_ptr(rb_v);
  auto rp = space_cast<1>(uniform_ptr(rb->rp));
  auto rec = space_cast<1>(uniform_ptr(rb->rec));
  auto wt = space_cast<1>(uniform_ptr(rb->wt));
  auto hd = space_cast<1>(uniform_ptr(rb->hd));
  auto hdB = space_cast<1>(uniform_ptr(rb->hdB));
  if (lane >= n) return;
  const int q = first + lane;
  auto row = dst + lane * stride;
  for (int t = 0; t < stride; ++t) row[t] = 0.0;
  if (q >= nB) return;
  const uint32_t e0 = rp[q], e1 = rp[q + 1];
  const double hb = hdB[q];
  constexpr int EB = 12;
  for (uint32_t eb = e0; eb < e1; eb += (uint32_t)EB) {
    uint32_t w0[EB], w1[EB];
    double a1[EB], a2[EB], dr[EB];
#pragma unroll
    for (int a = 0; a < EB; ++a) {
      const uint32_t e = eb + (uint32_t)a < e1 ? eb + (uint32_t)a : e1 - 1u;  // clamped, unconditional
      w0[a] = rec[2u * e];
      w1[a] = rec[2u * e + 1u];
    }
#pragma unroll
    for (int a = 0; a < EB; ++a) {
      const uint32_t k1 = w0[a] & 0x7fffu, k2x = w0[a] >> 16;
      a1[a] = wt[k1];
      a2[a] = wt[k2x == 0xffffu ? k1 : (k2x & 0x7fffu)];
      dr[a] = hd[w1[a] & 0xffffu];
    }
#pragma unroll
    for (int a = 0; a < EB; ++a) {
      const uint32_t k2x = w0[a] >> 16, offs = (w1[a] >> 16) & 0xffu;
      const double h1 = (w0[a] & 0x8000u) ? -a1[a] : a1[a];
      const double add = (k2x == 0xffffu) ? h1 : -(h1 * dr[a]) * ((k2x & 0x8000u) ? -a2[a] : a2[a]);
      if (eb + (uint32_t)a < e1) row[offs] += add;
    }
  }
  row[0] += hb;
}

CAVE_NOINLINE __device__ void band_gen_rows(const BandGen* gen_v, const int lane, const int p_v, const int bw_v,
                                            const int first_v, const int n_v, double* dst_v, const int stride_v) {
  const BandGen* gen = uniform_ptr(gen_v);
  const int p = __builtin_amdgcn_readfirstlane(p_v), bw = __builtin_amdgcn_readfirstlane(bw_v);
  const int first = __builtin_amdgcn_readfirstlane(first_v), n = __builtin_amdgcn_readfirstlane(n_v);
  const int stride = __builtin_amdgcn_readfirstlane(stride_v);
  auto dst = space_cast<3>(uniform_ptr(dst_v));
  if (gen->rb) {  // (wave-uniform)
    band_gen_rows_rb(gen->rb, lane, p, first, n, dst, stride);
    return;
  }
  auto g_mptr = space_cast<1>(uniform_ptr(gen->mptr));
  auto g_mcol = space_cast<1>(uniform_ptr(gen->mcol));
  auto g_mval = space_cast<1>(uniform_ptr(gen->mval));
  auto g_cptr = space_cast<1>(uniform_ptr(gen->cptr));
  auto g_cvar = space_cast<1>(uniform_ptr(gen->cvar));
  auto g_cvalc = space_cast<1>(uniform_ptr(gen->cvalc));
  auto g_usign = space_cast<1>(uniform_ptr(gen->usign));
  const double* g_r = uniform_ptr(gen->r);
  const double g_mu = gen->mu;  // (1 / mu, see BandGen)
  const bool g_pm1 = gen->mval == nullptr;
  if (lane >= n) return;
  constexpr int EB = 4, CB = 3;
  const int r = first + lane;
  auto q = dst + lane * stride;
  for (int t = 0; t < stride; ++t) q[t] = 0.0;
  if (r >= p) return;
  const uint32_t elo = g_mptr[r], ehi = g_mptr[r + 1];
  const uint32_t elast = g_mptr[p] > 0u ? g_mptr[p] - 1u : 0u;
  for (uint32_t eb = elo; eb < ehi; eb += (uint32_t)EB) {
    uint32_t k[EB], clo[EB], cnt[EB], xj[EB][CB], us[EB];
    double v1[EB], rk[EB], v2[EB][CB];
#pragma unroll
    for (int a = 0; a < EB; ++a) {
      const uint32_t e = eb + (uint32_t)a < elast ? eb + (uint32_t)a : elast;  // clamped, unconditional
      const uint32_t xk = g_mcol[e];
      k[a] = g_pm1 ? (xk & 0x7fffu) : xk;
      v1[a] = g_pm1 ? ((xk & 0x8000u) ? -1.0 : 1.0) : (double)g_mval[e];
    }
#pragma unroll
    for (int a = 0; a < EB; ++a) {
      us[a] = g_usign[k[a]];
      rk[a] = g_r[k[a]];
      clo[a] = g_cptr[k[a]];
      cnt[a] = g_cptr[k[a] + 1u] - clo[a];
    }
#pragma unroll
    for (int a = 0; a < EB; ++a)
#pragma unroll
      for (int b = 0; b < CB; ++b) {
        const uint32_t e2 = clo[a] + (uint32_t)b < elast ? clo[a] + (uint32_t)b : elast;
        xj[a][b] = g_cvar[e2];
        v2[a][b] = g_pm1 ? 0.0 : (double)g_cvalc[e2];
      }
#pragma unroll
    for (int a = 0; a < EB; ++a) {
      if (eb + (uint32_t)a >= ehi) break;
      const double tk = band_weight((uint8_t)us[a], rk[a], g_mu) * v1[a];
#pragma unroll
      for (int b = 0; b < CB; ++b) {
        if ((uint32_t)b >= cnt[a]) break;
        const int j = (int)(g_pm1 ? (xj[a][b] & 0x7fffu) : xj[a][b]);
        const double vv = g_pm1 ? ((xj[a][b] & 0x8000u) ? -1.0 : 1.0) : v2[a][b];
        if (j >= r && j - r <= bw) q[j - r] += tk * vv;
      }
      for (uint32_t b = (uint32_t)CB; b < cnt[a]; ++b) {  // columns with more than CB entries
        const uint32_t x2 = g_cvar[clo[a] + b];
        const int j = (int)(g_pm1 ? (x2 & 0x7fffu) : x2);
        const double vv = g_pm1 ? ((x2 & 0x8000u) ? -1.0 : 1.0) : (double)g_cvalc[clo[a] + b];
        if (j >= r && j - r <= bw) q[j - r] += tk * vv;
      }
    }
  }
}

// The elimination advances kBandBlock = 4 pivots per step:
//   A  the four pivot rows are read into registers (lane t = column k + t, lane 63 = the right-hand side) and
//      eliminated against each other there (v_readlane broadcasts, no LDS traffic);
//   B  pivot rows P[0..3][t] and multipliers Q[0..3][t] = P * (1/d) go to an LDS scratch, the finished rows to the
//      factor (workspace);
//   C  every duo of the trailing triangle takes its four updates in one pass: 4 + 4 operand reads whose addresses
//      never change, one ds_read2_b64 / ds_write2_b64 of the target -- a quarter of the LDS round trips and
//      of the address arithmetic of a pivot-at-a-time loop;
//   D  the right-hand side below the block.
// Every entry receives the same fma sequence as in a pivot-at-a-time elimination, so the bits are the same.
//
// Rows live in ONE ring of RING = 64 or 96 rows (band_wave_ring_rows: the former window and staging buffers in one
// piece), row r in slot r mod RING, in the form the elimination sees them (shifted diagonal, identity rows of bound
// rows, zero rows past the end).  NW >= 2: wave 1 is the PRODUCER -- it builds rows kBandGroup at a time (from the cone:
// band_gen_rows, or from the materialised band) up to RING - (bw + 4) rows ahead of the pivot, wave 0 is the ELIMINATOR
// and runs A-D alone; they meet through two LDS words (rows produced / rows retired), not through barriers: the stamp
// build of round 3 showed the two-wave form of round 2 (wave 1 admits rows and shares C, two barriers per step) waiting
// for wave 1 in 2.4 k of the 4.5 k cycles of a step -- ~1.3 k per step to admit four rows and 11 k cycles per
// band_gen_rows call every eight steps, all on the critical path.  NW = 1: the one wave produces the next group when the
// step needs it.  The back substitution stays on wave 0.
template <int NW>
CAVE_NOINLINE __device__ void solve_spd_band_wave(const int lane, const int wave_v, const double* Hb_v, const int bw_v,
                                                  const double* rhs_v, const uint8_t* act_v, const int p_v,
                                                  const double reg_rel, double* win_v, double* fac_v, double* z_v,
                                                  double* x_v, const BandGen* gen_v = nullptr,
                                                  unsigned long long* stamps = nullptr, const int x_cap_v = -1) {
  constexpr int U = kBandWaveDuos, RMAX = kBandWaveRegs, NB = kBandBlock, G = kBandGroup;
  constexpr bool DUO = NW >= 2;  // a producer wave exists
  const int wave = __builtin_amdgcn_readfirstlane(wave_v);
  const int bw = __builtin_amdgcn_readfirstlane(bw_v), p = __builtin_amdgcn_readfirstlane(p_v);
  const bool w0 = wave == 0;
  double* win_ = uniform_ptr(win_v);
  auto win = space_cast<3>(win_);
  const int ld = bw + 1, wl = band_wave_stride(bw), R = bw + NB;
  const int RING = band_wave_ring_rows(bw), rsz = RING * wl;
  // the words the waves meet through.  Elimination: rows [0, flags[0]) have been built, rows [0, flags[1]) retired
  auto flags = reinterpret_cast<typename SpacePtr<int, 3>::type>(win + band_wave_flags_at(bw));
  if constexpr (DUO) {
    if (w0 && lane == 0) {
      CAVE_FLAG_STORE(flags + 0, 0);
      CAVE_FLAG_STORE(flags + 1, 0);
      CAVE_FLAG_STORE(flags + 2, 0x7fffffff);  // back substitution: lowest factor row in its ring (not started)
      CAVE_FLAG_STORE(flags + 3, 0x7fffffff);  // back substitution: the row being solved
    }
    CAVE_LDS_BARRIER();  // (also: whatever used this LDS before is done with it)
    if (wave >= 2) return;
  }
  const double* Hb_ = uniform_ptr(Hb_v);
  const double* rhs = uniform_ptr(rhs_v);
  const uint8_t* act_ = uniform_ptr(act_v);
  double* fac_ = uniform_ptr(fac_v);
  double* z_ = uniform_ptr(z_v);
  double* x_ = uniform_ptr(x_v);
  const BandGen* gen = uniform_ptr(gen_v);
  const bool hgen = gen != nullptr && gen->on;  // rows of H on demand (no bound rows): Hb is not read
#ifdef CAVE_STAMPS
  struct { unsigned long long* st; } c{stamps};
#endif
  CAVE_T0();
  auto Hb = space_cast<1>(Hb_);
  auto fac = space_cast<1>(fac_);
  auto act = space_cast<3>(act_);
  auto z = space_cast<3>(z_);
  auto x = space_cast<3>(x_);
  // operand scratch: P[a][t] = U[k+a][k+t], Q[a][t] = P[a][t] / d_a, t = NB .. bw + NB + 1 (last column: zeros).
  // Component-major: the lanes of a duo round read consecutive t, i.e. consecutive words (as [t][a] records the
  // same reads were 16-way bank conflicts: rocprof counted 1.0e9 conflict cycles per launch on the 30x30 batch)
  const int ncol = bw + NB + 2;
  // (x_cap: entries x really has -- the red-black form solves nB < p rows in the arrays of a p-row system, whose LDS
  // region was sized for p: the scratch stays where that sizing put it)
  const int x_cap = __builtin_amdgcn_readfirstlane(x_cap_v) >= 0 ? __builtin_amdgcn_readfirstlane(x_cap_v) : p;
  auto scrP = ((uint32_t)x_cap >= band_wave_scratch(bw)) ? x : win + (band_wave_flags_at(bw) + 3);
  auto scrQ = scrP + NB * ncol;
  double md = 0.0;
  uint32_t nfix = 0;
  if (!hgen) {
    for (int i = lane; i < p; i += 64) {
      if (!act[i]) md = fmax(md, Hb[i * ld]);
      else nfix++;
    }
    md = wave_max_f64(md);
    nfix = wave_sum_u32(nfix);
  } else md = gen->hdiag;
  const double reg = reg_rel * md;
  // rows first .. first + G - 1 into their slots (first is a multiple of G, RING too: a group never wraps)
  auto produce = [&](const int first, const int slot) __attribute__((always_inline)) {
    auto dst = win + slot * wl;
    if (hgen) {  // (a real call: band_gen_rows has its own register allocation; it zeroes wl entries per row)
      band_gen_rows(gen, lane, p, bw, first, G, (double*)dst, wl);
      CAVE_WAVE_ORDER();
      if (lane < G && first + lane < p) dst[lane * wl] += reg;
    } else {
      // lane = entry of the row, eight rows per batch.  All loads first, from clamped addresses, and the masks as
      // selects: written with an `if`, each row pays two dependent round trips (the loads are sunk into the branch)
      const int tl = lane <= bw ? lane : bw;
      constexpr int RB = 8;
      for (int r0 = 0; r0 < G; r0 += RB) {
        double raw[RB];
        uint32_t fI[RB], fJ[RB];
#pragma unroll
        for (int a = 0; a < RB; ++a) {
          const int rI = first + r0 + a;
          const int rc = rI < p ? rI : p - 1;
          raw[a] = Hb[rc * ld + tl];
          fI[a] = act[rc];
          fJ[a] = act[rI + tl < p ? rI + tl : p - 1];
        }
#pragma unroll
        for (int a = 0; a < RB; ++a) {
          const int rI = first + r0 + a;
          const bool fixed = (fI[a] | fJ[a]) != 0u;
          const double diag = fI[a] != 0u ? 1.0 : raw[a] + reg;
          double v = (lane == 0) ? diag : (fixed ? 0.0 : raw[a]);
          v = (rI + tl < p && lane <= bw) ? v : 0.0;
          if (lane < wl) dst[(r0 + a) * wl + lane] = v;
        }
      }
    }
  };
  // last pivot block: the elimination reads rows below p_last + R
  const int p_last = ((p - 1) / NB) * NB;
  if constexpr (DUO) {
    if (!w0) {  // ---- the producer
      const int p_end = (p_last + R + G - 1) / G * G;
      int slot = 0;
      for (int g = 0; g < p_end; g += G) {
        const int need = g + G - RING;  // rows [g, g + G) take the slots of rows [g - RING, g + G - RING)
        if (need > 0) {
          int spin = 0;
          for (; spin < kBandSpinLimit; ++spin) {
            if (__builtin_amdgcn_readfirstlane(CAVE_FLAG_LOAD(flags + 1)) >= need) break;
            CAVE_SPIN_PAUSE();
          }
          if (spin >= kBandSpinLimit && lane == 0) CAVE_FLAG_STORE(flags + 4, 1);  // hand-over failed: flag the instance
          CAVE_WAVE_ORDER();
        }
        produce(g, slot);
        CAVE_LDS_WAIT();
        slot += G;
        slot = slot >= RING ? 0 : slot;
#ifdef CAVE_TEST_WITHHOLD_FLAG  // (test builds: the producer "forgets" to announce one group of rows)
        if (lane == 0 && g != CAVE_TEST_WITHHOLD_FLAG * G) CAVE_FLAG_STORE(flags + 0, g + G);
#else
        if (lane == 0) CAVE_FLAG_STORE(flags + 0, g + G);
#endif
      }
      // ---- then the factor rows of the back substitution, top row first, into the 64-row ring (row r: slot r & 63).
      // Two batches of BR rows in registers (lane = entry): the loads of one are in flight while the other waits
      // for its slots -- row r may take the slot of row r + 64 once the substitution has passed that row.
      int spin = 0;
      for (; spin < kBandSpinLimit; ++spin) {  // the eliminator's stores have completed (its fence)
        if (__builtin_amdgcn_readfirstlane(CAVE_FLAG_LOAD(flags + 2)) <= p) break;
        CAVE_SPIN_PAUSE();
      }
      if (spin >= kBandSpinLimit && lane == 0) CAVE_FLAG_STORE(flags + 4, 1);  // hand-over failed: flag the instance
      CAVE_WAVE_ORDER();
      constexpr int BR = 16;
      const int rs = band_wave_ring_stride(bw);
      const int tl = lane < ld ? lane : bw;
      double ra[BR], rb[BR];
      auto loadb = [&](double* rg, const int top) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < BR; ++j) rg[j] = fac[(top - j > 0 ? top - j : 0) * ld + tl];
      };
      auto storeb = [&](const double* rg, const int top) __attribute__((always_inline)) {
        const int lowest = top - (BR - 1) > 0 ? top - (BR - 1) : 0;
        int spin = 0;
        for (; spin < kBandSpinLimit; ++spin) {
          if (__builtin_amdgcn_readfirstlane(CAVE_FLAG_LOAD(flags + 3)) < lowest + 64) break;
          CAVE_SPIN_PAUSE();
        }
        if (spin >= kBandSpinLimit && lane == 0) CAVE_FLAG_STORE(flags + 4, 1);  // hand-over failed: flag the instance
        CAVE_WAVE_ORDER();
        if (lane < ld) {
#pragma unroll
          for (int j = 0; j < BR; ++j)
            if (top - j >= 0) win[((top - j) & 63) * rs + lane] = rg[j];
        }
        CAVE_LDS_WAIT();
        if (lane == 0) CAVE_FLAG_STORE(flags + 2, lowest);
      };
      int top = p - 1;
      loadb(ra, top);
      while (top >= 0) {
        if (top - BR >= 0) loadb(rb, top - BR);
        storeb(ra, top);
        top -= BR;
        if (top < 0) break;
        if (top - BR >= 0) loadb(ra, top - BR);
        storeb(rb, top);
        top -= BR;
      }
      return;
    }
  }
  for (int i = lane; i < p; i += 64) {
    double zi = rhs[i];
    if (nfix != 0u && !act[i]) {
      const int j0 = i - bw > 0 ? i - bw : 0, j1 = i + bw < p - 1 ? i + bw : p - 1;
      for (int j = j0; j <= j1; ++j)
        if (act[j]) zi -= ((i >= j) ? Hb[j * ld + (i - j)] : Hb[i * ld + (j - i)]) * rhs[j];
    }
    z[i] = zi;
  }
  // this wave's duos: duo q in row-major order over rows s = 1 .. bw below the block (entries t = s, s+2, ...);
  // round u = duos 64u .. 64u + 63.  A lane past the end of the last round repeats a duo of the same round: two lanes
  // then store the same bits to the same address (all loads of a step precede its stores), which needs no predicate.
  int uq[U], up[U], ur[U];
  const int nd = band_wave_duos(bw);
  const int nu = (nd + 63) / 64;
  {
#pragma unroll
    for (int i = 0; i < U; ++i) {
      int nv = nd - 64 * i;
      nv = nv > 64 ? 64 : (nv < 1 ? 1 : nv);
      int q = 64 * i + (lane < nv ? lane : lane % nv), s = 1;
      q = q < nd ? q : 0;
      while (q >= (bw - s + 2) / 2) { q -= (bw - s + 2) / 2; ++s; }
      const int t = s + 2 * q;
      uq[i] = s + NB - 1;               // column k + NB - 1 + s in Q
      up[i] = t + NB - 1;               // column k + NB - 1 + t (and the next one) in P
      ur[i] = (s + NB - 1) * wl + (t - s);  // target, relative to the first pivot row's slot
    }
  }
  // zero column of the scratch, written once
  if (lane < NB) { scrP[lane * ncol + ncol - 1] = 0.0; scrQ[lane * ncol + ncol - 1] = 0.0; }
  CAVE_WAVE_ORDER();
  CAVE_ACC(10);
  auto eliminate = [&](auto nu_tag) __attribute__((always_inline)) {
    constexpr int NU = decltype(nu_tag)::value;
    int slot_k = 0;        // slot of row k (RING is a multiple of NB: a block never wraps)
    int have = 0, hslot = 0;  // rows built so far (DUO: as last seen in the producer's word)
    const bool zl = lane == 63;
    for (int k = 0; k < p; k += NB) {
      // rows k .. k + R - 1 are resident before the step touches them
      if constexpr (DUO) {
        if (have < k + R) {
          int spin = 0;
          for (; spin < kBandSpinLimit; ++spin) {
            have = __builtin_amdgcn_readfirstlane(CAVE_FLAG_LOAD(flags + 0));
            if (have >= k + R) break;
            CAVE_SPIN_PAUSE();
          }
          if (spin >= kBandSpinLimit && lane == 0) CAVE_FLAG_STORE(flags + 4, 1);  // hand-over failed: flag the instance
          CAVE_WAVE_ORDER();
        }
      } else {
        while (have < k + R) {
          produce(have, hslot);
          have += G;
          hslot += G;
          hslot = hslot >= RING ? 0 : hslot;
          CAVE_WAVE_ORDER();
        }
      }
      // ---- A: pivot rows k .. k+3 into registers; lane t holds column k + t (entry t - a of row k + a)
      double u[NB], uz[NB];
#pragma unroll
      for (int a = 0; a < NB; ++a) {  // all eight loads, then the selects (pinned: see CAVE_PIN_F64)
        const int off = lane - a;
        const bool in = off >= 0 && off <= bw;
        u[a] = win[(slot_k + a) * wl + (in ? off : 0)];
        uz[a] = z[k + a < p ? k + a : p - 1];
      }
#pragma unroll
      for (int a = 0; a < NB; ++a) { CAVE_PIN_F64(u[a]); CAVE_PIN_F64(uz[a]); }
      if constexpr (DUO) {  // the four slots are free once these loads have executed (a wave's LDS operations do so in order)
        if (lane == 0) CAVE_FLAG_STORE(flags + 1, k + NB);
      }
#pragma unroll
      for (int a = 0; a < NB; ++a) {
        const int off = lane - a;
        const bool in = off >= 0 && off <= bw;
        u[a] = zl ? (k + a < p ? uz[a] : 0.0) : (in ? u[a] : 0.0);
      }
      double inv[NB];
#pragma unroll
      for (int a = 0; a < NB; ++a) {
        const double d = readlane_f64(u[a], a);
        // (no select in FRONT of the reciprocal: a dropped pivot gives inf / NaN there and 0 here, and the chain of a
        // kept one loses a VALU -> SALU -> VALU round trip)
        double iv = rcp_full(d);
        iv = d > 1e-300 ? iv : 0.0;
        inv[a] = iv;
#pragma unroll
        for (int b = a + 1; b < NB; ++b) {
          const double m = readlane_f64(u[a], b) * iv;
          u[b] = fma(-m, u[a], u[b]);
        }
      }
      CAVE_ACCF(0);
      // ---- B: operands of the trailing update, finished rows
      if (lane >= NB && lane < ncol - 1) {
#pragma unroll
        for (int a = 0; a < NB; ++a) {
          scrP[a * ncol + lane] = u[a];
          scrQ[a * ncol + lane] = u[a] * inv[a];
        }
      }
      double zq[NB];
#pragma unroll
      for (int a = 0; a < NB; ++a) {
        zq[a] = readlane_f64(u[a], 63);
        const int off = lane - a;
        if (off >= 0 && off <= bw && k + a < p) fac[(k + a) * ld + off] = (off == 0) ? inv[a] : u[a];
      }
      if (zl) {
#pragma unroll
        for (int a = 0; a < NB; ++a)
          if (k + a < p) z[k + a] = u[a];
      }
      CAVE_WAVE_ORDER();
      CAVE_ACCF(1);
      // ---- C: trailing triangle
      const int base_k = slot_k * wl;
      double qv[NU > 0 ? NU : 1][NB], p0[NU > 0 ? NU : 1][NB], p1[NU > 0 ? NU : 1][NB], r0[NU > 0 ? NU : 1], r1[NU > 0 ? NU : 1];
      int poff[NU > 0 ? NU : 1];
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        int o = base_k + ur[i];
        o = o >= rsz ? o - rsz : o;
        poff[i] = o;
#pragma unroll
        for (int a = 0; a < NB; ++a) {
          qv[i][a] = scrQ[a * ncol + uq[i]];
          p0[i][a] = scrP[a * ncol + up[i]];
          p1[i][a] = scrP[a * ncol + up[i] + 1];
        }
        r0[i] = win[o];
        r1[i] = win[o + 1];
      }
      // ---- D: right-hand side below the block (lane i: row k + NB + i)
      const bool zown = lane < bw && k + NB + lane < p;
      double zqv[NB];
#pragma unroll
      for (int a = 0; a < NB; ++a) zqv[a] = scrQ[a * ncol + NB + (lane < bw ? lane : 0)];
      double zz = z[zown ? k + NB + lane : 0];
      // every load of the step precedes its stores -- in every lane: a lane past the end of a round repeats another
      // lane's duo and must read what that lane read, not what it has already written back
      CAVE_WAVE_ORDER();
#pragma unroll
      for (int i = 0; i < NU; ++i) {
#pragma unroll
        for (int a = 0; a < NB; ++a) {
          r0[i] = fma(-qv[i][a], p0[i][a], r0[i]);
          r1[i] = fma(-qv[i][a], p1[i][a], r1[i]);
        }
        win[poff[i]] = r0[i];
        win[poff[i] + 1] = r1[i];
      }
#pragma unroll
      for (int a = 0; a < NB; ++a) zz = fma(-zqv[a], zq[a], zz);
      if (zown) z[k + NB + lane] = zz;
      CAVE_ACCF(9);
      slot_k += NB;
      slot_k = slot_k >= RING ? 0 : slot_k;
      CAVE_WAVE_ORDER();
    }
  };
  switch (nu) {
    case 0: eliminate(std::integral_constant<int, 0>{}); break;
    case 1: eliminate(std::integral_constant<int, 1>{}); break;
    case 2: eliminate(std::integral_constant<int, 2>{}); break;
    case 3: eliminate(std::integral_constant<int, 3>{}); break;
    case 4: eliminate(std::integral_constant<int, 4>{}); break;
    default: eliminate(std::integral_constant<int, 5>{}); break;
  }
  // the factor rows go out through this wave's stores and come back through its loads
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  CAVE_ACC(11);
#ifdef CAVE_STAMPS
  c.st[13] += (unsigned long long)p;  // pivots eliminated (per-pivot cost = slot 11 / slot 13)
#endif
  // ---- back substitution  x_k = inv_k (z_k - sum_s U[k][k+s] x_{k+s}), column oriented.
  // Factor rows come back through a 64-row ring in LDS (row r in slot r & 63; one zero entry behind the ring), so lane l
  // finds every row it owns at l * stride: the entry of column k in its row is at offset s = k - row, and an offset past
  // the band is redirected to the zero entry.  Rows are fetched CHB at a time (registers: lane = entry,
  // register = row), a chunk ahead, and parked once the rows that used their slots are done.
  constexpr int CHB = RMAX;
  double regs[RMAX];
  const int rs = band_wave_ring_stride(bw), zero_at = 64 * rs;
  auto ring = win;  // (the row ring of the elimination is free now)
  ring[zero_at] = 0.0;
  const int tl = lane < ld ? lane : bw;
  auto fetch_b = [&](int chi) {  // rows chi, chi-1, .. of the factor
#pragma unroll
    for (int j = 0; j < CHB; ++j) {
      const int r = chi - j > 0 ? chi - j : 0;
      regs[j] = fac[r * ld + tl];
    }
  };
  auto park_b = [&](int chi) {
    if (lane < ld) {
#pragma unroll
      for (int j = 0; j < CHB; ++j)
        if (chi - j >= 0) ring[((chi - j) & 63) * rs + lane] = regs[j];
    }
  };
  int chi_next = p - 1;  // NW = 1: top row of the chunk in flight; NW >= 2: lowest row the producer has delivered
  if constexpr (DUO) {
    chi_next = p;
    if (lane == 0) { CAVE_FLAG_STORE(flags + 3, p - 1); CAVE_FLAG_STORE(flags + 2, p); }  // (in this order: starts the producer)
  } else {
    fetch_b(chi_next);
    while (chi_next >= 0 && chi_next + bw + NB >= p - 1) {
      park_b(chi_next);
      chi_next -= CHB;
      if (chi_next >= 0) fetch_b(chi_next);
    }
  }
  CAVE_WAVE_ORDER();
  {
    double acc = 0.0;  // partial sum of the row this lane owns (row = lane mod 64)
    const int rowbase = lane * rs;
    // rows k - bw - NB .. are in the ring before step k reads them
    auto admit = [&](int k) __attribute__((always_inline)) {
      if constexpr (DUO) {
        if (lane == 0) CAVE_FLAG_STORE(flags + 3, k);  // rows above k are done with: their slots are free
        const int need = k - bw - NB > 0 ? k - bw - NB : 0;
        if (chi_next > need) {
          int spin = 0;
          for (; spin < kBandSpinLimit; ++spin) {
            chi_next = __builtin_amdgcn_readfirstlane(CAVE_FLAG_LOAD(flags + 2));
            if (chi_next <= need) break;
            CAVE_SPIN_PAUSE();
          }
          if (spin >= kBandSpinLimit && lane == 0) CAVE_FLAG_STORE(flags + 4, 1);  // hand-over failed: flag the instance
          CAVE_WAVE_ORDER();
        }
      } else {
        if (chi_next >= 0 && k <= chi_next + bw + NB) {
          park_b(chi_next);
          chi_next -= CHB;
          if (chi_next >= 0) fetch_b(chi_next);
          CAVE_WAVE_ORDER();
        }
      }
    };
    // entry t of factor row r, zero past the band (scalar address: every lane reads the same word)
    auto entry = [&](int r, int t) -> double { return ring[t <= bw ? (r & 63) * rs + t : zero_at]; };
    int k = p - 1;
    // the p mod NB rows at the bottom, one at a time
    for (; k >= 0 && ((k + 1) % NB) != 0; --k) {
      admit(k);
      const int sft = 1 + ((k - 1 - lane) & 63);
      const double fcol = ring[sft < ld ? rowbase + sft : zero_at];
      const double xk = ring[(k & 63) * rs] * (z[k] - readlane_f64(acc, k & 63));
      const bool own = lane == (k & 63);
      acc = own ? 0.0 : fma(fcol, xk, acc);
      if (own) x[k] = xk;
      CAVE_WAVE_ORDER();
    }
    // then NB = 4 rows per step: the four unknowns from one batch of loads (the 4x4 triangle between them as
    // scalars), after which every lane folds all four into its row's partial sum -- same fma order as row by row
    for (; k >= NB - 1; k -= NB) {
      admit(k);
      const int s0 = 1 + ((k - 1 - lane) & 63);  // offset of column k in this lane's row; column k - j: s0 - j
      double fc[NB];
      bool fin[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int sj = s0 - j;
        fin[j] = sj >= 1 && sj <= bw;
        fc[j] = ring[rowbase + (fin[j] ? sj : 0)];
      }
      double f0[NB], zz[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) { f0[j] = ring[((k - j) & 63) * rs]; zz[j] = z[k - j]; }
      const double c10 = entry(k - 1, 1), c20 = entry(k - 2, 2), c21 = entry(k - 2, 1), c30 = entry(k - 3, 3),
                   c31 = entry(k - 3, 2), c32 = entry(k - 3, 1);
      const double x0 = f0[0] * (zz[0] - readlane_f64(acc, k & 63));
      double a1 = readlane_f64(acc, (k - 1) & 63), a2 = readlane_f64(acc, (k - 2) & 63), a3 = readlane_f64(acc, (k - 3) & 63);
      a1 = fma(c10, x0, a1);
      const double x1 = f0[1] * (zz[1] - a1);
      a2 = fma(c20, x0, a2);
      a2 = fma(c21, x1, a2);
      const double x2 = f0[2] * (zz[2] - a2);
      a3 = fma(c30, x0, a3);
      a3 = fma(c31, x1, a3);
      a3 = fma(c32, x2, a3);
      const double x3 = f0[3] * (zz[3] - a3);
      acc = fma(fin[0] ? fc[0] : 0.0, x0, acc);
      acc = fma(fin[1] ? fc[1] : 0.0, x1, acc);
      acc = fma(fin[2] ? fc[2] : 0.0, x2, acc);
      acc = fma(fin[3] ? fc[3] : 0.0, x3, acc);
      const int mine = (k - lane) & 63;  // < NB: this lane owned one of the four rows; its sum starts over
      acc = mine < NB ? 0.0 : acc;
      if (lane < NB) x[k - lane] = lane == 0 ? x0 : (lane == 1 ? x1 : (lane == 2 ? x2 : x3));
      CAVE_WAVE_ORDER();
    }
  }
  CAVE_WAVE_ORDER();
  CAVE_ACC(12);
}
#endif  // CAVE_GPU_CODE

}  // namespace cave
