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
CAVE_NOINLINE void solve_spd_band(C& c, const double* Hb_, int bw, const double* rhs, const uint8_t* act_, int p,
                            double reg_rel, double* win_, double* fac_, double* z_, double* x_, double* stg_, int CH) {
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

}  // namespace cave
