"""Throw-away numpy prototype of the reduced semismooth-Newton projection."""
import sys, time
import numpy as np
from scipy.optimize import nnls
sys.path.insert(0, "/root/repo")
from cave_amd import synth


def classify(A):
    """Dense (m,d) -> (upos, uneg, M, kind, n_valid)."""
    m, d = A.shape
    keep = np.abs(A).sum(1) > 1e-7
    A = A[keep]
    nnz = (A != 0).sum(1)
    upos = np.zeros(d, int); uneg = np.zeros(d, int)
    gen = []
    for i in range(len(A)):
        if nnz[i] == 1:
            k = np.nonzero(A[i])[0][0]
            if A[i, k] > 0: upos[k] += 1
            else: uneg[k] += 1
        else:
            gen.append(i)
    G = A[gen].astype(np.float64)
    # pair detection
    p = len(G)
    used = np.zeros(p, bool); rows = []; kind = []
    for i in range(p):
        if used[i]: continue
        tw = -1
        for j in range(i + 1, p):
            if not used[j] and np.array_equal(G[j], -G[i]):
                tw = j; break
        used[i] = True
        if tw >= 0:
            used[tw] = True; rows.append(G[i]); kind.append(1)
        else:
            rows.append(G[i]); kind.append(0)
    M = np.array(rows).reshape(len(rows), d)
    return upos, uneg, M, np.array(kind, int), len(A)


def clip(r, upos, uneg):
    both = (upos > 0) & (uneg > 0)
    res = np.where(both, 0.0, np.where(upos > 0, np.minimum(r, 0), np.where(uneg > 0, np.maximum(r, 0), r)))
    return res


DELTA=1e-10
MODE='proj'
def chol_skip_solve(Hm, b, tau=1e-14):
    Hm = Hm + DELTA*max(Hm.diagonal().max(),1e-300)*np.eye(len(b))
    n = len(b); L = np.zeros((n, n)); skip = np.zeros(n, bool)
    for j in range(n):
        v = Hm[j, j] - L[j, :j] @ L[j, :j]
        if v <= tau * max(Hm[j, j], 1e-300):
            skip[j] = True; L[j, :] = 0; L[:, j] = 0; L[j, j] = 1.0
            continue
        L[j, j] = np.sqrt(v)
        for i in range(j + 1, n):
            L[i, j] = (Hm[i, j] - L[i, :j] @ L[j, :j]) / L[j, j]
    bb = np.where(skip, 0.0, b)
    z = np.linalg.solve(L, bb)
    z[skip] = 0
    x = np.linalg.solve(L.T, z)
    x[skip] = 0
    return x


def project_ssn(A, y, max_it=100, verbose=False):
    upos, uneg, M, kind, nvalid = classify(A)
    y = y.astype(np.float64)
    if nvalid == 0:
        return y.copy(), 0.0, 0
    p = len(M)
    theta = np.zeros(p)
    nonneg = kind == 0
    def fval(th):
        r = y - M.T @ th
        res = clip(r, upos, uneg)
        return 0.5 * res @ res, res
    f, res = fval(theta)
    g0n = None
    for it in range(max_it):
        g = -(M @ res)
        pg = np.where(nonneg, theta - np.maximum(theta - g, 0), g)
        pgn = np.abs(pg).max() if p else 0.0
        if g0n is None: g0n = max(pgn, 1e-300)
        if verbose: print(it, f, pgn)
        if pgn <= 1e-9 * g0n or f <= 1e-30 * (y @ y):
            break
        eps = min(1e-3, pgn)
        act = nonneg & (theta <= eps) & (g > 0)
        F = ~act
        D = (res != 0).astype(float)
        H = (M * D) @ M.T
        dvec = -g.copy()  # active: gradient step
        if F.any():
            idx = np.where(F)[0]
            dvec[idx] = chol_skip_solve(H[np.ix_(idx, idx)], -g[idx])
        alpha = 1.0
        if MODE == 'ratio':
            dvec = np.where(act, 0.0, dvec)
            neg = nonneg & F & (dvec < 0)
            hit = np.zeros(p, bool)
            if neg.any():
                ratios = np.where(neg, theta / np.where(neg, -dvec, 1.0), np.inf)
                amax = ratios.min()
                if amax < 1.0:
                    alpha = amax; hit = neg & (ratios <= amax * (1 + 1e-12))
        ok = False
        for ls in range(40):
            th = theta + alpha * dvec
            th = np.where(nonneg, np.maximum(th, 0), th)
            if MODE == 'ratio':
                th = np.where(act, 0.0, th)
                if ls == 0: th = np.where(hit, 0.0, th)
            fn, resn = fval(th)
            if fn <= f + 1e-4 * g @ (th - theta):
                ok = True; break
            alpha *= 0.5
        if not ok:
            if verbose: print("line search failed")
            break
        theta, f, res = th, fn, resn
    proj = y - res
    return proj, np.sqrt(2 * f), it


def ref_nnls(A, y):
    A = A[np.abs(A).sum(1) > 1e-7]
    if len(A) == 0: return y.astype(np.float32), 0.0
    lam, rn = nnls(np.asfortranarray(A.T.astype(np.float64)), y.astype(np.float64), maxiter=10 * A.shape[0])
    return (lam @ A), rn


def run(name, ctrs, ys):
    worst = 0; its = []; t0 = 0; t1 = 0
    for A, y in zip(ctrs, ys):
        ta = time.time(); p0, r0 = ref_nnls(A, y); tb = time.time()
        p1, r1, it = project_ssn(A, y); tc = time.time()
        t0 += tb - ta; t1 += tc - tb
        err = np.abs(p0 - p1).max(); worst = max(worst, err, abs(r0 - r1)); its.append(it)
    print(f"{name}: worst err {worst:.2e} its mean {np.mean(its):.1f} max {np.max(its)} scipy {t0/len(ys)*1e3:.2f}ms ssn {t1/len(ys)*1e3:.2f}ms")


if __name__ == "__main__":
    import sys
    if len(sys.argv)>1: DELTA=float(sys.argv[1])
    if len(sys.argv)>2: MODE=sys.argv[2]
    c, y, _ = synth.tsp_batch(20, 32, 0); run("tsp20", c, -y)
    c, y, _ = synth.sp_batch(5, 5, 32, 0); run("sp5x5", c, -y)
    c, y = synth.generic_batch(64); run("generic", c, -y)
    rng = np.random.default_rng(3)
    run("rand+", rng.random((32, 15, 10)).astype(np.float32), -rng.random((32, 10)).astype(np.float32))
    run("wide m=40 d=10", rng.standard_normal((32, 40, 10)).astype(np.float32), rng.standard_normal((32, 10)).astype(np.float32))
    run("tall m=8 d=30", rng.standard_normal((32, 8, 30)).astype(np.float32), rng.standard_normal((32, 30)).astype(np.float32))
    # duplicates + pairs, no unit rows
    G = rng.standard_normal((32, 6, 12)).astype(np.float32)
    run("dups/pairs", np.concatenate([G, -G[:, :3], G[:, 2:5]], 1), rng.standard_normal((32, 12)).astype(np.float32))
    # inside cone: y = positive combination
    A = rng.standard_normal((32, 15, 10)).astype(np.float32)
    lam = rng.random((32, 15)).astype(np.float32)
    run("inside", A, np.einsum("bm,bmd->bd", lam, A))
    c, y, _ = synth.tsp_batch(50, 2, 0); run("tsp50", c, -y)
