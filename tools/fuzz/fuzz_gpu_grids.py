"""GPU stress of the red-black band reduction (cone_rb.h): shortest-path cones of random rectangular grids (18 .. 40 nodes a
side: the shapes that take the reduction, and some that do not) through the packed store at 1 / 2 / 4 waves per instance;
every projection KKT-certified (tests/certificate.py), two launches of a shape bit-identical.
    python tools/fuzz/fuzz_gpu_grids.py [seed] [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from cave_amd import synth
from cave_amd.dataset import ConeStore
from certificate import kkt_certificate

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
T = float(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
t0 = time.time(); last = t0; n = 0; bad = 0; nondet = 0; worst = 0.0; itmax = 0; shapes = 0
while time.time() - t0 < T:
    if time.time() - last > 60:
        last = time.time(); print(f"... {int(last - t0)} s, {n} instances, {bad} failures", flush=True)
    h, w = int(rng.integers(18, 41)), int(rng.integers(18, 41))
    B = 6
    A, y, _ = synth.sp_batch(h, w, B, seed=int(rng.integers(1 << 30)))
    yp = (y + rng.normal(0, 0.05, y.shape)).astype(np.float32)
    store = ConeStore.from_dense(torch.tensor(A, device="cuda"), chunk=3)
    ids = torch.arange(B, device="cuda"); pt = torch.tensor(yp, device="cuda")
    shapes += 1
    for waves in (2, 1, 4):
        store.large_waves = waves
        o = store.cone_op(ids, pt, 0, -1.0, 0.0, outputs=("proj", "rnorm"), check=False)
        o2 = store.cone_op(ids, pt, 0, -1.0, 0.0, outputs=("proj", "rnorm"), check=False)
        if not (torch.equal(o["proj"], o2["proj"]) and torch.equal(o["iters"], o2["iters"])):
            nondet += 1; print("NOT BIT-IDENTICAL", (h, w), waves)
        if bool((o["status"] != 0).any()):
            bad += 1; print("STATUS", (h, w), waves, o["status"].tolist()); continue
        itmax = max(itmax, int(o["iters"].max()))
        p = o["proj"].cpu().numpy()
        for b in range(B):
            c = kkt_certificate(A[b], -yp[b], p[b])
            n += 1
            worst = max(worst, c["dual"], c["comp"])
            if not (c["dual"] <= 4e-6 and c["comp"] <= 4e-6 and c["member"]):
                bad += 1; print("NOT CERTIFIED", (h, w), waves, b, c)
print(f"grid shapes {shapes} instances x wave shapes {n} failures {bad} nondeterministic {nondet} worst KKT residual {worst:.2e} max iters {itmax}")
sys.exit(1 if bad or nondet else 0)
