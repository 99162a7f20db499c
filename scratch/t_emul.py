import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/scratch")
import numpy as np
from emul_lib import Emul
from cave_amd import synth
from proto_ssn import ref_nnls
E = Emul()
def run(name, ctrs, ys, **kw):
    t0=time.time(); o = E.cone_dense(ctrs, ys, 0, sign=1.0, **kw); t1=time.time()
    worst=0
    for b,(A,y) in enumerate(zip(ctrs,ys)):
        p0,r0 = ref_nnls(A,y)
        worst = max(worst, np.abs(p0-o["proj"][b]).max(), abs(r0-o["rnorm"][b]))
    print(f"{name}: worst {worst:.2e} status {np.bincount(o['status'])} iters mean {o['iters'].mean():.1f} max {o['iters'].max()}  emul {1e3*(t1-t0)/len(ys):.3f} ms/inst")
c, y, _ = synth.tsp_batch(20, 32, 0); run("tsp20", c, -y)
c, y, _ = synth.sp_batch(5, 5, 32, 0); run("sp5x5", c, -y)
c, y = synth.generic_batch(64); run("generic", c, -y)
rng = np.random.default_rng(3)
run("rand+", rng.random((32, 15, 10)).astype(np.float32), -rng.random((32, 10)).astype(np.float32))
run("wide m=40 d=10", rng.standard_normal((32, 40, 10)).astype(np.float32), rng.standard_normal((32, 10)).astype(np.float32))
run("tall m=8 d=30", rng.standard_normal((32, 8, 30)).astype(np.float32), rng.standard_normal((32, 30)).astype(np.float32))
G = rng.standard_normal((32, 6, 12)).astype(np.float32)
run("dups/pairs", np.concatenate([G, -G[:, :3], G[:, 2:5]], 1), rng.standard_normal((32, 12)).astype(np.float32))
A = rng.standard_normal((32, 15, 10)).astype(np.float32)
lam = rng.random((32, 15)).astype(np.float32)
run("inside", A, np.einsum("bm,bmd->bd", lam, A))
c, y, _ = synth.tsp_batch(50, 2, 0); run("tsp50", c, -y)
