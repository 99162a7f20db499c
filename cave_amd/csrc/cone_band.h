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
  // ring window: band row r lives in slot r % ld
  for (int idx = c.tid(); idx < ld * ld; idx += NT) {
    const int r = idx / ld, t = idx - r * ld;
    win[idx] = band_row_entry(Hb, ld, act, p, reg, r, t);
  }
  c.sync();
  for (int k = 0; k < p; ++k) {
    double* wk = win + (k % ld) * ld;
    const double dk = wk[0];
    const bool ok = dk > 1e-300;
    const double inv = ok ? 1.0 / dk : 0.0;
    const int nb = bw < p - 1 - k ? bw : p - 1 - k;
    const double zk = z[k];
    // trailing update of rows k+1 .. k+nb (upper triangle s <= t of the nb x nb block) and of z
    for (int idx = c.tid(); idx < nb * nb; idx += NT) {
      const int s0 = idx / nb, t0 = idx - s0 * nb;
      if (t0 < s0) continue;
      const int s = s0 + 1, t = t0 + 1;
      double* row = win + ((k + s) % ld) * ld;
      row[t - s] -= wk[s] * inv * wk[t];
    }
    for (int s = 1 + c.tid(); s <= nb; s += NT) z[k + s] -= wk[s] * inv * zk;
    c.sync();
    // retire row k to the factor, bring row k + ld into its slot
    for (int t = c.tid(); t <= bw; t += NT) {
      fac[k * ld + t] = (t == 0) ? inv : wk[t];
      wk[t] = band_row_entry(Hb, ld, act, p, reg, k + ld, t);
    }
    c.sync();
  }
  // back substitution  x_k = inv_k * (z_k - sum_s fac[k][s] * x_{k+s})
  for (int k = p - 1; k >= 0; --k) {
    const int nb = bw < p - 1 - k ? bw : p - 1 - k;
    double part = 0.0;
    for (int s = 1 + c.tid(); s <= nb; s += NT) part += fac[k * ld + s] * x[k + s];
    part = c.reduce_sum(part);
    if (c.tid() == 0) x[k] = fac[k * ld] * (z[k] - part);
    c.sync();
  }
}

}  // namespace cave
