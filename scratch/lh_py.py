import numpy as np
def lh(A, b, verbose=False):
    """Textbook Lawson-Hanson, A: (d, n) columns are generators."""
    d, n = A.shape
    x = np.zeros(n); P = np.zeros(n, bool)
    w = A.T @ (b - A @ x)
    tol = 10 * np.finfo(float).eps * np.abs(A).sum(0).max() * max(d, n)
    it = 0
    while (~P).any() and w[~P].max() > tol:
        j = np.where(~P, w, -np.inf).argmax()
        P[j] = True
        s = np.zeros(n); s[P] = np.linalg.lstsq(A[:, P], b, rcond=None)[0]
        if verbose: print("add", j, "w", w[j], "s_j", s[j])
        while s[P].min() <= 0:
            it += 1
            mask = P & (s <= 0)
            alpha = (x[mask] / (x[mask] - s[mask])).min()
            x = x + alpha * (s - x)
            P[P & (x <= 1e-15)] = False   # hmm
            x[~P] = 0
            s = np.zeros(n)
            if P.any(): s[P] = np.linalg.lstsq(A[:, P], b, rcond=None)[0]
            else: break
            if it > 10*n: raise RuntimeError("maxiter")
        x = s.copy()
        w = A.T @ (b - A @ x)
        it += 1
        if it > 10*n: raise RuntimeError("maxiter")
    return x, np.linalg.norm(A @ x - b)
if __name__ == "__main__":
    import sys
    z = np.load(sys.argv[1]); A=z["A"].astype(float); y=z["y"].astype(float)
    A = A[np.abs(A).sum(1)>1e-7]
    x, r = lh(A.T, y, verbose=True); print("rn", r)
