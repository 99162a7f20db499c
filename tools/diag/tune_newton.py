"""Diagnostic (CPU): Newton iteration statistics of the per-instance solver on TSP-20 cones for other values of
the smoothing / line-search constants (serial test build of the kernel code, rebuilt per variant)."""
import ctypes as C, itertools, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import emul_lib
from cave_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
c, y, _ = synth.tsp_batch(20, n, seed=0)
y = y + np.random.default_rng(1).normal(0, 0.05, y.shape).astype(np.float32)
variants = [dict()] + [dict(CAVE_MU_COEF=a) for a in (0.15, 0.2, 0.25, 0.3)]
sp = synth.sp_batch(5, 5, 256, seed=3)
for v in variants:
    so = "/tmp/_emul_var_%d.so" % variants.index(v)
    flags = [f"-D{k}={val}" for k, val in v.items()]
    subprocess.run(["g++", "-std=c++17", "-fPIC", "-shared", "-w", "-O2", *flags, os.path.join(ROOT, "tests/emul/emul_abi.cpp"), "-o", so], check=True)
    E = emul_lib.Emul.__new__(emul_lib.Emul)
    E.lib = C.CDLL(so)
    o = E.cone_dense(c, y, 2)
    it = o["iters"]
    hist = np.bincount(it, minlength=13)[3:13]
    o2 = E.cone_dense(sp[0], sp[1], 2)
    print(f"{str(v):32s} TSP-20: mean {it.mean():.3f} max {it.max()} hist(3..12) {hist.tolist()} bad {int((o['status'] != 0).sum())} | SP5x5: mean {o2['iters'].mean():.3f} max {o2['iters'].max()}", flush=True)
    del E
    os.remove(so)
