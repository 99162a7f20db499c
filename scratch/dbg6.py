import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/scratch")
import numpy as np, ctypes as C
from emul_lib import Emul
from oracle import cave_oracle as O
from scipy.optimize import nnls
np.set_printoptions(linewidth=200, precision=4, suppress=True)
E = Emul(); 
if len(sys.argv)>2: E.lib = C.CDLL("/tmp/_emul_trace.so")
z = np.load(sys.argv[1]); A=z["A"]; y=z["y"]
print("m,d", A.shape)
o = E.cone_dense(A[None], y[None], 0, sign=1.0, nnz_cap=max(A.size,64), lds_bytes=160*1024)
print("ours  rn", o["rnorm"], o["status"], o["iters"])
try:
    p,r = O.project_nnls(y, A); print("oracle rn", r)
except Exception as e: print("oracle", e)
Ak = A[np.abs(A).sum(1)>1e-7].astype(np.float64)
lam, rs = nnls(np.asfortranarray(Ak.T), y.astype(np.float64)); print("scipy rn", rs, "true resid of scipy x", np.linalg.norm(lam@Ak-y))
from scipy.optimize import lsq_linear
res = lsq_linear(Ak.T, y.astype(np.float64), bounds=(0,np.inf), method='bvls', tol=1e-14); print("bvls rn", np.linalg.norm(Ak.T@res.x - y))
