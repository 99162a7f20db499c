"""GPU stress of the large-cone path at larger sizes than fuzz_gpu_large.py (SP up to 24x24, TSP up to 64, random
cones up to 260 generators in up to 220 dimensions), each instance certified by the KKT conditions
(tests/certificate.py) instead of the oracle, which would need minutes here.
    python tools/fuzz/fuzz_gpu_large_big.py [seed] [seconds]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from collections import Counter
from cave_amd import synth, qpsolver
from cave_amd.qpsolver import cone_op_dense
from certificate import kkt_certificate

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
T = float(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
t0 = time.time(); last = t0; n = 0; bad = 0; itmax = 0; worst = 0.0
st = Counter()
while time.time() - t0 < T:
    if time.time() - last > 60:
        last = time.time(); print(f"... {int(last - t0)} s, {n} instances, {bad} failures", flush=True)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        A, y, _ = synth.tsp_batch(int(rng.integers(30, 65)), 4, seed=int(rng.integers(1 << 30)))
    elif kind == 1:
        A, y, _ = synth.sp_batch(int(rng.integers(10, 25)), int(rng.integers(10, 25)), 4, seed=int(rng.integers(1 << 30)))
    else:
        d, m, B = int(rng.integers(60, 220)), int(rng.integers(70, 260)), 4
        A = rng.standard_normal((B, m, d)).astype(np.float32)
        if kind == 3: A *= rng.random((B, m, d)) < 0.1
        y = rng.standard_normal((B, d)).astype(np.float32)
    At, yt = torch.tensor(A, device="cuda"), torch.tensor(y, device="cuda")
    qpsolver._tier[(A.shape[1], A.shape[2])] = 2
    try:
        o = cone_op_dense(At, yt, 0, -1.0, 0.0, outputs=("proj", "rnorm"))
    except qpsolver.HipSolverError as ex:
        bad += 1; print("RAISED kind", kind, A.shape, str(ex)[:100]); continue
    p = o["proj"].cpu().numpy(); itmax = max(itmax, int(o["iters"].max()))
    for b in range(len(A)):
        c = kkt_certificate(A[b], -y[b], p[b])
        n += 1; st[kind] += 1
        worst = max(worst, c["dual"], c["comp"])
        if not (c["dual"] <= 4e-6 and c["comp"] <= 4e-6 and c["member"]):
            bad += 1; print("NOT CERTIFIED kind", kind, A.shape, c)
print(f"instances {n} failures {bad} worst KKT residual {worst:.2e} max iters {itmax} by kind {dict(st)}")
