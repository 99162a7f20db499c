import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, ctypes as C
import emul_lib
from emul_lib import Emul
E = Emul(); E.lib = C.CDLL("/tmp/_emul_trace.so")
rng = np.random.default_rng(3)
rng.random((32, 15, 10)); rng.random((32, 10)); rng.standard_normal((32, 40, 10)); rng.standard_normal((32, 10)); rng.standard_normal((32, 8, 30)); rng.standard_normal((32, 30))
G = rng.standard_normal((32, 6, 12)); rng.standard_normal((32, 12))
A = rng.standard_normal((32, 15, 10)).astype(np.float32)
lam = rng.random((32, 15)).astype(np.float32)
Y = np.einsum("bm,bmd->bd", lam, A)
o = E.cone_dense(A[20:21], Y[20:21], 0, sign=1.0)
print(o["status"], o["iters"], o["rnorm"])
import os
E2 = Emul()
o = E2.cone_dense(A, Y, 0, sign=1.0)
bad = np.where(o["status"]!=0)[0]; print("bad", bad, o["iters"][bad])
if len(bad):
    b = bad[0]
    o = E.cone_dense(A[b:b+1], Y[b:b+1], 0, sign=1.0)
