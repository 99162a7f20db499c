import sys; sys.path.insert(0,'/root/repo/scratch'); sys.path.insert(0,'/root/repo')
import proto_ssn
from proto_ssn import *
proto_ssn.DELTA=1e-10
rng = np.random.default_rng(3)
rng.random((32, 15, 10)); rng.random((32, 10)); rng.standard_normal((32, 40, 10)); rng.standard_normal((32, 10)); rng.standard_normal((32, 8, 30)); rng.standard_normal((32, 30))
G = rng.standard_normal((32, 6, 12)); rng.standard_normal((32, 12))
A = rng.standard_normal((32, 15, 10)).astype(np.float32)
lam = rng.random((32, 15)).astype(np.float32)
Y = np.einsum("bm,bmd->bd", lam, A)
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    project_ssn(A[20],Y[20],max_it=100,verbose=True)
lines = buf.getvalue().splitlines()
print("\n".join(lines[36:60])); print("..."); print("\n".join(lines[-5:]))
