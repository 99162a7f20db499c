"""Size-independent parity check for one projection: a KKT certificate (TEST INFRASTRUCTURE).

p is the Euclidean projection of y onto K = cone{rows of A} iff, with r = y - p,
  (1) r is in the polar cone:      A r <= 0,
  (2) complementarity:             p . r = 0,
  (3) p is in K:                   p = lam @ A for some lam >= 0.
(1) and (2) are sparse mat-vecs.  For (3) complementarity lets us restrict lam to the rows with
(A r)_i = 0 and eliminate the signed-unit rows c*e_k in closed form: a coordinate k with such rows
on both sides is free, with one side only it constrains the sign of q_k, with none q_k = 0, where
q = p - lam_G @ A_G over the remaining ("general") rows.  What is left is a small LP feasibility
problem (HiGHS), independent of any NNLS code.  This is how the full-size cases (TSP-100, 30x30
grids), where the CPU oracle and SciPy need minutes to hours per instance, are checked.
"""
import numpy as np
import scipy.sparse as sp
from scipy.optimize import linprog


def kkt_certificate(A, y, p, tol=4e-6):
    """A: dense (m, d) array, or a scipy sparse matrix (coordinate-form cones of cave_amd.synth.coo_batch)."""
    if sp.issparse(A):
        As = sp.csr_matrix(A, dtype=np.float64)
        As = As[np.asarray(abs(As).sum(1)).ravel() > 1e-7]
    else:
        A = np.asarray(A, np.float32)
        keep = np.abs(A).sum(1) > 1e-7  # the reference's padded-row rule (src/cave.py:303)
        As = sp.csr_matrix(A[keep].astype(np.float64))
    y = np.asarray(y, np.float64)
    p = np.asarray(p, np.float64)
    d = y.size
    scale = max(1.0, float(np.abs(y).max()))
    r = y - p
    l1 = np.asarray(abs(As).sum(1)).ravel()
    w = As @ r
    dual = float((w / np.maximum(l1, 1.0)).max(initial=0.0)) / scale           # (1)
    comp = abs(float(p @ r)) / max(float(y @ y), 1.0)                            # (2)
    # (3) membership over the tight rows
    tight = w >= -tol * scale * np.maximum(l1, 1.0)
    nnz_row = np.diff(As.indptr)
    unit = tight & (nnz_row == 1)
    gen = tight & (nnz_row > 1)
    plus = np.zeros(d, bool)
    minus = np.zeros(d, bool)
    ui = np.flatnonzero(unit)
    cols = As.indices[As.indptr[ui]]
    vals = As.data[As.indptr[ui]]
    plus[cols[vals > 0]] = True
    minus[cols[vals < 0]] = True
    G = As[np.flatnonzero(gen)]
    eps = tol * scale
    # q = p - G^T lam;  need q_k >= -eps unless `minus`, q_k <= eps unless `plus`
    need_lo = ~minus  # q_k >= -eps  <=>  (G^T lam)_k <= p_k + eps
    need_hi = ~plus   # q_k <= eps   <=>  -(G^T lam)_k <= -p_k + eps
    GT = G.T.tocsr()
    if G.shape[0] == 0:
        member = bool((p[need_lo] >= -eps).all() and (p[need_hi] <= eps).all())
    else:
        A_ub = sp.vstack([GT[np.flatnonzero(need_lo)], -GT[np.flatnonzero(need_hi)]]).tocsr()
        b_ub = np.concatenate([p[need_lo] + eps, -p[need_hi] + eps])
        res = linprog(np.zeros(G.shape[0]), A_ub=A_ub, b_ub=b_ub, bounds=(0, None), method="highs")
        member = bool(res.status == 0)
    return {"dual": dual, "comp": comp, "member": member, "n_general_tight": int(G.shape[0])}


def assert_projection(A, y, p, tol=4e-6, what=""):
    c = kkt_certificate(A, y, p, tol)
    assert c["dual"] <= tol and c["comp"] <= tol and c["member"], (what, c)
    return c


def coo_to_sparse(item, d):
    """(rows, cols, vals, m) of cave_amd.synth.*_cone_coo -> scipy CSR (m, d)."""
    r, c, v, m = item
    return sp.csr_matrix((v.astype(np.float64), (r, c)), shape=(m, d))
