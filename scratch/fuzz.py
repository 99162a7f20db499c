import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/scratch")
import numpy as np
from emul_lib import Emul
from cave_amd import synth
from proto_ssn import ref_nnls
E = Emul()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
worst_all = 0; nbad = 0; itmax = 0; n=0
t0=time.time()
while time.time()-t0 < float(sys.argv[2]) if len(sys.argv)>2 else 60:
    d = int(rng.integers(1, 24)); m = int(rng.integers(0, 40)); B = 8
    kind = rng.integers(0, 6)
    A = rng.standard_normal((B, m, d)).astype(np.float32)
    if kind == 1: A = np.abs(A)
    if kind == 2:  # sparse + unit rows + pairs
        A *= (rng.random((B, m, d)) < 0.3)
        for b in range(B):
            for r in range(m):
                u = rng.random()
                if u < 0.3: 
                    A[b, r] = 0; A[b, r, rng.integers(0, d)] = rng.choice([-1.0, 1.0]) * rng.choice([1.0, 0.5, 2.0])
                elif u < 0.45 and r > 0: A[b, r] = -A[b, rng.integers(0, r)]
                elif u < 0.5 and r > 0: A[b, r] = A[b, rng.integers(0, r)]
                elif u < 0.55: A[b, r] = 0
    if kind == 3: A = np.round(A)  # integer coefficients -> exact degeneracies
    if kind == 4: A[:, m//2:] = 0
    y = rng.standard_normal((B, d)).astype(np.float32)
    if kind == 5 and m > 0:  # inside cone
        lam = rng.random((B, m)).astype(np.float32); y = np.einsum("bm,bmd->bd", lam, A)
    if rng.random() < 0.1: y[:] = 0
    o = E.cone_dense(A, y, 0, sign=1.0, nnz_cap=max(m*d,64), lds_bytes=160*1024)
    for b in range(B):
        p0, r0 = ref_nnls(A[b], y[b])
        sc = max(1.0, np.abs(y[b]).max())
        err = max(np.abs(p0 - o["proj"][b]).max(), abs(r0 - o["rnorm"][b])) / sc
        n+=1
        if o["status"][b] != 0 or not err < 2e-6:
            nbad += 1
            if nbad <= 5:
                print("BAD kind", kind, "m,d", m, d, "status", o["status"][b], "err", err, "iters", o["iters"][b], "rn", r0, o["rnorm"][b])
                np.savez(f"/tmp/bad_{nbad}.npz", A=A[b], y=y[b])
        else: worst_all = max(worst_all, err)
        itmax = max(itmax, o["iters"][b])
print("n", n, "bad", nbad, "worst ok err", worst_all, "max iters", itmax)
