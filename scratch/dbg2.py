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
for i,(a,yy) in enumerate(zip(A,Y)):
    p,r,it = project_ssn(a,yy)
    if it>=30:
        print("instance",i, r); project_ssn(a,yy,max_it=40,verbose=True); break
