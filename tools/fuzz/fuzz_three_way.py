"""Three-way fuzz (CPU): serial build of the kernel code vs the Lawson-Hanson oracle vs SciPy nnls on
adversarial random cones (duplicates, +-pairs, unit rows, rank deficiency, points inside / on faces).
    python tools/fuzz/fuzz_three_way.py [seed] [seconds] [large]     (large: the large-cone path's code)
Used during round 1: 900k instances, 0 silent mismatches of the kernel code vs the oracle,
~0.5 % SciPy 1.15.3 answers that disagree with both."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from emul_lib import Emul
from oracle import cave_oracle as O
from scipy.optimize import nnls
E = Emul()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
T = float(sys.argv[2]) if len(sys.argv)>2 else 60
LARGE = len(sys.argv) > 3 and sys.argv[3] == "large"
n=0; bad_ours=0; bad_scipy=0; scipy_err=0; worst=0; itmax=0; by_kind={}
t0=time.time()
def kkt_gap(A, y, p):
    # p optimal iff A(y-p) <= 0 (dual feasible) and (y-p).p == 0
    r = y.astype(np.float64) - p.astype(np.float64)
    A = A.astype(np.float64)
    keep = np.abs(A).sum(1) > 1e-7
    w = A[keep] @ r
    sc = max(1.0, np.abs(y).max())
    return max(w.max(initial=0.0), abs(r @ p.astype(np.float64))) / sc**2
while time.time()-t0 < T:
    d = int(rng.integers(1, 24)); m = int(rng.integers(0, 40)); B = 8
    kind = int(rng.integers(0, 6))
    A = rng.standard_normal((B, m, d)).astype(np.float32)
    if kind == 1: A = np.abs(A)
    if kind == 2:
        A *= (rng.random((B, m, d)) < 0.3)
        for b in range(B):
            for r in range(m):
                u = rng.random()
                if u < 0.3:
                    A[b, r] = 0; A[b, r, rng.integers(0, d)] = rng.choice([-1.0, 1.0]) * rng.choice([1.0, 0.5, 2.0])
                elif u < 0.45 and r > 0: A[b, r] = -A[b, rng.integers(0, r)]
                elif u < 0.5 and r > 0: A[b, r] = A[b, rng.integers(0, r)]
                elif u < 0.55: A[b, r] = 0
    if kind == 3: A = np.round(A)
    if kind == 4: A[:, m//2:] = 0
    y = rng.standard_normal((B, d)).astype(np.float32)
    if kind == 5 and m > 0:
        lam = rng.random((B, m)).astype(np.float32); y = np.einsum("bm,bmd->bd", lam, A)
    if rng.random() < 0.1: y[:] = 0
    if LARGE: o = E.cone_dense_large(A, y, 0, sign=1.0, nnz_cap=max(m*d,64), band=max(m*m,1), lds_bytes=int(rng.choice([1024, 8192, 65536])))
    else: o = E.cone_dense(A, y, 0, sign=1.0, nnz_cap=max(m*d,64), lds_bytes=160*1024)
    try:
        po, ro = O.batch_project(y, A)
    except RuntimeError:  # the oracle's own Lawson-Hanson iteration cap (3n, as SciPy's): judge ours by its KKT residual
        oracle_raised = globals().get("oracle_raised", 0) + 1
        globals()["oracle_raised"] = oracle_raised
        for b in range(B):
            gap = kkt_gap(A[b], y[b], o["proj"][b])
            if o["status"][b] != 0 or not gap < 1e-5:
                bad_ours += 1
                print("OURS (oracle hit its iteration cap) kind", kind, m, d, "status", o["status"][b], "kkt", gap)
        n += B
        continue
    for b in range(B):
        n+=1
        sc = max(1.0, np.abs(y[b]).max())
        e1 = max(np.abs(po[b]-o["proj"][b]).max(), abs(ro[b]-o["rnorm"][b]))/sc
        if o["status"][b] != 0 or not e1 < 2e-6:
            bad_ours += 1
            if bad_ours <= 5:
                print("OURS-vs-ORACLE kind", kind, m, d, "status", o["status"][b], "err", e1, "kkt ours", kkt_gap(A[b],y[b],o["proj"][b]), "kkt oracle", kkt_gap(A[b],y[b],po[b]))
                np.savez(f"/tmp/bad_{bad_ours}.npz", A=A[b], y=y[b])
        else: worst=max(worst,e1)
        itmax=max(itmax,o["iters"][b])
        Ak = A[b][np.abs(A[b]).sum(1) > 1e-7].astype(np.float64)
        if len(Ak):
            try:
                lam, rs = nnls(np.asfortranarray(Ak.T), y[b].astype(np.float64)); ps = lam@Ak
                e2 = max(np.abs(ps-po[b]).max(), abs(rs-ro[b]))/sc
                if not e2 < 2e-6:
                    bad_scipy += 1; by_kind[kind]=by_kind.get(kind,0)+1
            except RuntimeError: scipy_err += 1
print(f"oracle hit its cap on {globals().get('oracle_raised', 0)} batches; " if globals().get("oracle_raised") else "", end="")
print("path counters [dense-LDL^T, ...]:", E.path_counters())
print(f"n {n}  ours!=oracle {bad_ours}  scipy!=oracle {bad_scipy} (by kind {by_kind}) scipy raised {scipy_err}  worst ours-oracle {worst:.2e} max iters {itmax}")
