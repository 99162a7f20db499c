import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/scratch")
import numpy as np, ctypes as C
from emul_lib import Emul
from proto_ssn import ref_nnls
E = Emul(); E.lib = C.CDLL("/tmp/_emul_trace.so")
z = np.load(sys.argv[1]); A=z["A"]; y=z["y"]
print(A, y)
o = E.cone_dense(A[None], y[None], 0, sign=1.0, nnz_cap=max(A.size,64), lds_bytes=160*1024)
print("ours", o["proj"], o["rnorm"], o["status"], o["iters"])
print("ref", ref_nnls(A,y))
