/* nnls_oracle.c — CPU restatement of the reference's cone projection.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this; the cave_amd package never does.
 *
 * What it restates (paths relative to /root/reference):
 *   cave_oracle_project_nnls   _project_nnls            src/cave.py:298-309
 *       row drop  |ctr|.sum(axis=1) > 1e-7              src/cave.py:303
 *       empty cone -> (cp, 0.0)                         src/cave.py:304-305
 *       lam, rnorm = nnls(ctr.T, cp); p = lam @ ctr     src/cave.py:306-309
 *   cave_oracle_average_ctrs   _average_ctrs            src/cave.py:222-228
 *
 * The NNLS itself lives in SciPy (scipy.optimize.nnls; the reference README pins
 * SciPy 1.11.2, this container has 1.15.3), which is not part of /root/reference.
 * It is restated here from the published algorithm SciPy cites: Lawson & Hanson,
 * "Solving Least Squares Problems" (1974/1995) ch. 23, in the normal-equation
 * form of Bro & de Jong, "A fast non-negativity-constrained least squares
 * algorithm", J. Chemometrics 11 (1997): precompute G = A^T A and h = A^T b,
 * grow the passive set by the most positive dual, solve G_PP s = h_P, step back
 * to feasibility with the ratio test.  All arithmetic in double.
 *
 * This is deliberately a different algorithm from the HIP kernels (which use a
 * reduced semismooth Newton method): the projection onto a closed convex cone
 * is unique, so agreement of the two is a real check.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_MAXITER 1
#define ORACLE_NOMEM 2

/* ---- Cholesky factor of G[P,P], kept incrementally (row-major, leading dimension n) ---- */

/* compute row `a` of L for passive index P[a]; returns 0 if P[a] is numerically dependent */
static int chol_row(const double* G, int n, const int* P, int a, double* L) {
  const int j = P[a];
  double* row = L + (size_t)a * n;
  double ss = 0.0;
  for (int c = 0; c < a; ++c) {
    double v = G[(size_t)j * n + P[c]];
    const double* rc = L + (size_t)c * n;
    for (int k = 0; k < c; ++k) v -= row[k] * rc[k];
    row[c] = v / rc[c];
    ss += row[c] * row[c];
  }
  double dd = G[(size_t)j * n + j] - ss;
  if (!(dd > 1e-14 * G[(size_t)j * n + j])) return 0;
  row[a] = sqrt(dd);
  return 1;
}

static void chol_solve(const double* L, int n, int np, const double* rhs, double* out) {
  for (int a = 0; a < np; ++a) {
    double v = rhs[a];
    const double* row = L + (size_t)a * n;
    for (int k = 0; k < a; ++k) v -= row[k] * out[k];
    out[a] = v / row[a];
  }
  for (int a = np - 1; a >= 0; --a) {
    double v = out[a];
    for (int k = a + 1; k < np; ++k) v -= L[(size_t)k * n + a] * out[k];
    out[a] = v / L[(size_t)a * n + a];
  }
}

/* least squares on the passive set: normal equations + two steps of iterative refinement
 * with the residual formed from A itself (corrected semi-normal equations, Bjorck 1987),
 * so the accuracy is that of a QR solve.  Also returns r = b - A_P^T s in `r`. */
static void ls_passive(const double* A, int d, const double* b, const double* L, int n, const int* P, int np,
                       double* s, double* r, double* t, double* dl) {
  for (int a = 0; a < np; ++a) s[a] = 0.0;
  for (int pass = 0; pass < 3; ++pass) {
    for (int k = 0; k < d; ++k) r[k] = b[k];
    for (int a = 0; a < np; ++a) {
      const double* row = A + (size_t)P[a] * d;
      const double sa = s[a];
      if (sa != 0.0) for (int k = 0; k < d; ++k) r[k] -= sa * row[k];
    }
    for (int a = 0; a < np; ++a) {
      const double* row = A + (size_t)P[a] * d;
      double v = 0.0;
      for (int k = 0; k < d; ++k) v += row[k] * r[k];
      t[a] = v;
    }
    chol_solve(L, n, np, t, dl);
    for (int a = 0; a < np; ++a) s[a] += dl[a];
  }
  for (int k = 0; k < d; ++k) r[k] = b[k];
  for (int a = 0; a < np; ++a) {
    const double* row = A + (size_t)P[a] * d;
    for (int k = 0; k < d; ++k) r[k] -= s[a] * row[k];
  }
}

/* min_{x>=0} || A^T x - b ||, A given as n rows of length d (row i = generator i).
 * x: n multipliers (out).  Returns ORACLE_*. */
static int nnls_rows(const double* A, int n, int d, const double* b, double* x, int maxiter, int* iters_out) {
  double* G = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* L = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* s = (double*)malloc(sizeof(double) * n);
  double* t = (double*)malloc(sizeof(double) * n);
  double* dl = (double*)malloc(sizeof(double) * n);
  double* r = (double*)malloc(sizeof(double) * d);
  int* P = (int*)malloc(sizeof(int) * n);
  unsigned char* inP = (unsigned char*)calloc(n, 1);
  unsigned char* skip = (unsigned char*)calloc(n, 1);
  int rc = ORACLE_OK, np = 0, it = 0;
  if (!G || !L || !s || !t || !dl || !r || !P || !inP || !skip) { rc = ORACLE_NOMEM; goto done; }
  double anorm = 0.0, bnorm = 0.0;
  for (int i = 0; i < n; ++i) {
    double rs = 0.0;
    for (int j = 0; j <= i; ++j) {
      double v = 0.0;
      for (int k = 0; k < d; ++k) v += A[(size_t)i * d + k] * A[(size_t)j * d + k];
      G[(size_t)i * n + j] = v;
      G[(size_t)j * n + i] = v;
    }
    for (int k = 0; k < d; ++k) rs += fabs(A[(size_t)i * d + k]);
    if (rs > anorm) anorm = rs;
    x[i] = 0.0;
  }
  for (int k = 0; k < d; ++k) { bnorm += b[k] * b[k]; r[k] = b[k]; }
  bnorm = sqrt(bnorm);
  /* dual tolerance: round-off level of a_j . r */
  const double tol = 2.220446049250313e-16 * 100.0 * (double)(n > d ? n : d) * anorm * (bnorm > 0 ? bnorm : 1.0);
  const int rank_cap = n < d ? n : d;
  for (;;) {
    /* dual w = A r, r = b - A^T x */
    int jbest = -1;
    double wbest = tol;
    for (int i = 0; i < n; ++i) {
      if (inP[i] || skip[i]) continue;
      const double* row = A + (size_t)i * d;
      double v = 0.0;
      for (int k = 0; k < d; ++k) v += row[k] * r[k];
      if (v > wbest) { wbest = v; jbest = i; }
    }
#ifdef ORACLE_TRACE
    printf("add %d w %.3e np %d tol %.3e\n", jbest, wbest, np, tol);
#endif
    if (jbest < 0 || np >= rank_cap) break;
    if (it++ >= maxiter) { rc = ORACLE_MAXITER; break; }
    P[np] = jbest;
    if (!chol_row(G, n, P, np, L)) {
#ifdef ORACLE_TRACE
      printf("   dependent %d\n", jbest);
#endif
      skip[jbest] = 1; continue; } /* dependent on the passive set */
    np++;
    inP[jbest] = 1;
    int moved = 0;
    for (;;) {
      ls_passive(A, d, b, L, n, P, np, s, r, t, dl);
      int allpos = 1;
      for (int a = 0; a < np; ++a) if (!(s[a] > 0.0)) { allpos = 0; break; }
      if (allpos) {
        for (int a = 0; a < np; ++a) x[P[a]] = s[a];
        moved = 1;
        break;
      }
      double alpha = 2.0;
      for (int a = 0; a < np; ++a) {
        if (!(s[a] > 0.0)) {
          double xi = x[P[a]];
          double q = xi / (xi - s[a]);
          if (q < alpha) alpha = q;
        }
      }
      if (!(alpha <= 1.0)) alpha = 0.0;
#ifdef ORACLE_TRACE
      printf("   inner alpha %.3e np %d\n", alpha, np);
#endif
      if (alpha > 0.0) moved = 1;
      int first_out = -1, nn = 0;
      for (int a = 0; a < np; ++a) {
        int i = P[a];
        double xi = x[i];
        x[i] = xi + alpha * (s[a] - xi);
        int out = !(s[a] > 0.0) && (xi <= alpha * (xi - s[a]) * (1.0 + 1e-12));
        if (out) { x[i] = 0.0; inP[i] = 0; if (first_out < 0) first_out = a; }
        else P[nn++] = i;
      }
      np = nn;
      for (int a = first_out; a < np; ++a) {
        if (!chol_row(G, n, P, a, L)) { /* cannot happen for an independent set; keep going safely */
          L[(size_t)a * n + a] = sqrt(1e-11 * G[(size_t)P[a] * n + P[a]] + 1e-300);
        }
      }
      if (np == 0) { for (int k = 0; k < d; ++k) r[k] = b[k]; break; }
      if (it++ >= maxiter) { rc = ORACLE_MAXITER; goto done; }
    }
    /* residual for the next dual: r = b - A^T x */
    for (int k = 0; k < d; ++k) r[k] = b[k];
    for (int a = 0; a < np; ++a) {
      const double* row = A + (size_t)P[a] * d;
      for (int k = 0; k < d; ++k) r[k] -= x[P[a]] * row[k];
    }
    /* Lawson-Hanson guard: an entrant thrown straight out again without any movement had a
     * positive dual only by round-off; do not offer it again until x has moved. */
    if (!inP[jbest] && !moved) skip[jbest] = 1;
    else memset(skip, 0, (size_t)n);
  }
done:
  if (iters_out) *iters_out = it;
  free(G); free(L); free(s); free(t); free(dl); free(r); free(P); free(inP); free(skip);
  return rc;
}

/* _project_nnls (src/cave.py:298-309).  ctr: m x d float32 row-major; cp: d float32.
 * proj_out: d float32; rnorm_out: double (the reference returns a Python float). */
int cave_oracle_project_nnls(const float* ctr, int m, int d, const float* cp, float* proj_out, double* rnorm_out,
                             int* iters_out) {
  int n = 0;
  int* keep = (int*)malloc(sizeof(int) * (m > 0 ? m : 1));
  if (!keep) return ORACLE_NOMEM;
  for (int i = 0; i < m; ++i) {
    float s = 0.f; /* numpy sums float32 */
    for (int k = 0; k < d; ++k) s += fabsf(ctr[(size_t)i * d + k]);
    if (s > 1e-7f) keep[n++] = i; /* src/cave.py:303 */
  }
  if (iters_out) *iters_out = 0;
  if (n == 0) { /* src/cave.py:304-305 */
    for (int k = 0; k < d; ++k) proj_out[k] = cp[k];
    *rnorm_out = 0.0;
    free(keep);
    return ORACLE_OK;
  }
  double* A = (double*)malloc(sizeof(double) * (size_t)n * d);
  double* b = (double*)malloc(sizeof(double) * d);
  double* x = (double*)malloc(sizeof(double) * n);
  if (!A || !b || !x) { free(keep); free(A); free(b); free(x); return ORACLE_NOMEM; }
  for (int a = 0; a < n; ++a)
    for (int k = 0; k < d; ++k) A[(size_t)a * d + k] = (double)ctr[(size_t)keep[a] * d + k];
  for (int k = 0; k < d; ++k) b[k] = (double)cp[k];
  int rc = nnls_rows(A, n, d, b, x, 30 * n + 100, iters_out);
  double rn = 0.0;
  for (int k = 0; k < d; ++k) {
    double p = 0.0;
    for (int a = 0; a < n; ++a) p += x[a] * A[(size_t)a * d + k]; /* lam @ ctr, src/cave.py:308 */
    proj_out[k] = (float)p;
    rn += (p - b[k]) * (p - b[k]);
  }
  *rnorm_out = sqrt(rn);
  free(keep); free(A); free(b); free(x);
  return rc;
}

/* batch loop in the reference's shape: [worker(cp[i], ctrs[i]) for i in range(B)] (src/cave.py:257) */
int cave_oracle_batch_project(const float* ctrs, const float* cps, int64_t B, int m, int d, float* proj, float* rnorm,
                              int* status) {
  int worst = ORACLE_OK;
  for (int64_t i = 0; i < B; ++i) {
    double rn;
    int rc = cave_oracle_project_nnls(ctrs + (size_t)i * m * d, m, d, cps + (size_t)i * d, proj + (size_t)i * d, &rn, 0);
    rnorm[i] = (float)rn; /* rnorm_np is built as float32, src/cave.py:261 */
    if (status) status[i] = rc;
    if (rc > worst) worst = rc;
  }
  return worst;
}

/* _average_ctrs (src/cave.py:222-228) for one instance, float32 arithmetic like torch. */
void cave_oracle_average_ctrs(const float* ctr, int m, int d, float* avg) {
  for (int k = 0; k < d; ++k) avg[k] = 0.f;
  float nvalid = 0.f;
  for (int i = 0; i < m; ++i) {
    double s2 = 0.0;
    for (int k = 0; k < d; ++k) s2 += (double)ctr[(size_t)i * d + k] * (double)ctr[(size_t)i * d + k];
    float nrm = (float)sqrt(s2);
    if (!(nrm > 1e-7f)) continue; /* valid = norms > 1e-7 */
    float den = nrm > 1e-8f ? nrm : 1e-8f;
    for (int k = 0; k < d; ++k) avg[k] += ctr[(size_t)i * d + k] / den;
    nvalid += 1.f;
  }
  if (nvalid < 1.f) nvalid = 1.f;
  for (int k = 0; k < d; ++k) avg[k] /= nvalid;
}
