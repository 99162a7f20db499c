// cone_band.h — SPD solve for reduced systems too large for registers (large-cone path).
//
// The generalised Hessian H = M D M^T has the sparsity of M M^T.  In the reduced-row order the
// cones of this domain give it a band: grid shortest-path cones (node-arc incidence rows in node
// order) have half bandwidth = grid width, TSP degree rows give a dense matrix (bw = p - 1, p ~ n).
// The solver is an LDL^T elimination of the band without pivoting, written against the Ctx interface:
// a ring window of bw+1 band rows (LDS when it fits) is updated by the whole workgroup, one barrier
// per phase; eliminated rows stream out to `bfac`; the right-hand side is eliminated alongside; the
// back substitution runs over `bfac`.
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
CAVE_HD double band_row_entry(const double* Hb, int ld, const uint8_t* act, int p, double reg, int r, int t) {
  const int i = r + t;
  if (r >= p || i >= p) return 0.0;
  const bool fixed = act[r] || act[i];
  if (t == 0) return act[r] ? 1.0 : Hb[r * ld] + reg;
  return fixed ? 0.0 : Hb[r * ld + t];
}

template <class C>
CAVE_HD void solve_spd_band(C& c, const double* Hb, int bw, const double* rhs, const uint8_t* act, int p,
                            double reg_rel, double* win, double* fac, double* z, double* x) {
  const int NT = C::NT;
  const int ld = bw + 1;
  if (p <= 0) return;
  double md = 0.0;
  uint32_t nfix = 0;
  for (int i = c.tid(); i < p; i += NT) {
    if (!act[i]) md = fmax(md, Hb[i * ld]);
    else nfix++;
  }
  md = c.reduce_max(md);
  nfix = c.reduce_add_u32(nfix);
  const double reg = reg_rel * md;
  // right-hand side: fixed rows keep theirs, free rows move the fixed unknowns over
  for (int i = c.tid(); i < p; i += NT) {
    double zi = rhs[i];
    if (nfix != 0u && !act[i]) {
      const int j0 = i - bw > 0 ? i - bw : 0, j1 = i + bw < p - 1 ? i + bw : p - 1;
      for (int j = j0; j <= j1; ++j)
        if (act[j]) zi -= band_at(Hb, ld, i, j) * rhs[j];
    }
    z[i] = zi;
  }
  if (bw == 0) {  // diagonal system
    for (int i = c.tid(); i < p; i += NT) {
      const double dk = act[i] ? 1.0 : Hb[i] + reg;
      x[i] = (dk > 1e-300) ? z[i] / dk : 0.0;
    }
    c.sync();
    return;
  }
  // ring window: band row r lives in slot r % ld
  for (int idx = c.tid(); idx < ld * ld; idx += NT) {
    const int r = idx / ld, t = idx - r * ld;
    win[idx] = band_row_entry(Hb, ld, act, p, reg, r, t);
  }
  // One barrier per pivot.  Step k: (A) pivot k updates rows k+1 .. k+bw-1 of the window and z;
  // (B) row k-1, final since the previous barrier, is retired to the factor and its slot takes row
  // k+bw of H -- the one row pivot k reaches only in its diagonal entry, which the inserting thread
  // adjusts itself.  A and B touch disjoint slots.  The H row for step k+1 is loaded one step ahead.
  double pre = (c.tid() <= bw) ? band_row_entry(Hb, ld, act, p, reg, ld, c.tid()) : 0.0;  // row 0 + ld, for step 1
  double inv_prev = 0.0;
  c.sync();
  for (int k = 0; k < p; ++k) {
    double* wk = win + (k % ld) * ld;
    const double dk = wk[0];
    const bool ok = dk > 1e-300;
    const double inv = ok ? 1.0 / dk : 0.0;
    const int nb = bw < p - 1 - k ? bw : p - 1 - k;
    // rows updated by phase A: s = 1 .. na (row k+bw is phase B's, except at k = 0 where it is already resident)
    const int na = (k == 0 || nb < bw) ? nb : bw - 1;
    const double zk = z[k];
    if (na > 0) {
      // upper triangle s <= t of the na x nb block, flat index walked without per-element division
      const int qn = NT / nb, rn = NT - qn * nb;
      int s0 = c.tid() / nb, t0 = c.tid() - s0 * nb;
      while (s0 < na) {
        if (t0 >= s0) {
          const int sft = s0 + 1, tt = t0 + 1;
          double* row = win + ((k + sft) % ld) * ld;
          row[tt - sft] -= wk[sft] * inv * wk[tt];
        }
        s0 += qn;
        t0 += rn;
        if (t0 >= nb) { t0 -= nb; ++s0; }
      }
    }
    for (int s = 1 + c.tid(); s <= nb; s += NT) z[k + s] -= wk[s] * inv * zk;
    if (k > 0) {
      double* wp = win + ((k - 1) % ld) * ld;
      for (int t = c.tid(); t <= bw; t += NT) {
        fac[(k - 1) * ld + t] = (t == 0) ? inv_prev : wp[t];
        double nv = (t == c.tid()) ? pre : band_row_entry(Hb, ld, act, p, reg, k - 1 + ld, t);
        if (t == 0 && k + bw < p) nv -= wk[bw] * inv * wk[bw];
        wp[t] = nv;
      }
    }
    inv_prev = inv;
    if (c.tid() <= bw) pre = band_row_entry(Hb, ld, act, p, reg, k + ld, c.tid());  // for step k + 1
    c.sync();
  }
  {
    const double* wp = win + ((p - 1) % ld) * ld;
    for (int t = c.tid(); t <= bw; t += NT) fac[(p - 1) * ld + t] = (t == 0) ? inv_prev : wp[t];
  }
  c.sync();
  // back substitution  x_k = inv_k * (z_k - sum_s fac[k][s] * x_{k+s}), by the first wave alone:
  // no barriers, the next factor row is loaded while the current one is reduced
  if (c.wave_id() == 0) {
    constexpr int WL = C::WL;
    const int lane = c.lane_id();
    const int nchunk = (bw + WL - 1) / WL;  // lanes cover s = 1 + lane + j*WL
    if (nchunk <= 1) {
      double fnext = (1 + lane <= bw && p - 1 >= 0) ? fac[(p - 1) * ld + 1 + lane] : 0.0;
      for (int k = p - 1; k >= 0; --k) {
        const int nb = bw < p - 1 - k ? bw : p - 1 - k;
        const double fk = fnext;
        if (k > 0) fnext = (1 + lane <= bw) ? fac[(k - 1) * ld + 1 + lane] : 0.0;
        double part = (1 + lane <= nb) ? fk * x[k + 1 + lane] : 0.0;
        part = c.wave_sum(part);
        if (lane == 0) x[k] = fac[k * ld] * (z[k] - part);
        c.wave_fence();
      }
    } else {
      for (int k = p - 1; k >= 0; --k) {
        const int nb = bw < p - 1 - k ? bw : p - 1 - k;
        double part = 0.0;
        for (int s = 1 + lane; s <= nb; s += WL) part += fac[k * ld + s] * x[k + s];
        part = c.wave_sum(part);
        if (lane == 0) x[k] = fac[k * ld] * (z[k] - part);
        c.wave_fence();
      }
    }
  }
  c.sync();
}

}  // namespace cave
