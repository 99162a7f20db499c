"""GPU stress of the large-cone path (run on an MI355X): structured and random cones, many with more than
64 reduced rows, dense and packed operators, against the CPU oracle; reports the spread between two launches
(the band Hessian is accumulated with floating-point atomics, so launches may differ in the last bits).
    python tools/fuzz/fuzz_gpu_large.py [seed] [seconds]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from collections import Counter
from cave_amd import synth, qpsolver
from cave_amd.qpsolver import cone_op_dense
from cave_amd.dataset import ConeStore
from oracle import cave_oracle as O

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
T = float(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
t0 = time.time(); n = 0; bad = 0; worst = 0.0; spread = 0.0; itmax = 0
st = Counter()
OUTS = ("proj", "rnorm", "target", "loss", "grad")
last_report = time.time()
while time.time() - t0 < T:
    if time.time() - last_report > 60:  # the GPU box kills a run that is silent for 7 minutes
        last_report = time.time()
        print(f"... {int(time.time() - t0)} s, {n} batches/instances so far, {bad} mismatches", flush=True)
    kind = int(rng.integers(0, 5))
    if kind == 0:
        A, y, _ = synth.tsp_batch(int(rng.integers(8, 36)), 12, seed=int(rng.integers(1 << 30)))
    elif kind == 1:
        A, y, _ = synth.sp_batch(int(rng.integers(3, 13)), int(rng.integers(3, 13)), 12, seed=int(rng.integers(1 << 30)))
    else:
        d, m, B = int(rng.integers(2, 110)), int(rng.integers(1, 130)), 8
        A = rng.standard_normal((B, m, d)).astype(np.float32)
        if kind == 3: A *= rng.random((B, m, d)) < 0.15
        if kind == 4:
            A = np.round(A * 0.7)
            if m > 3: A[:, 1] = -A[:, 0]
        y = rng.standard_normal((B, d)).astype(np.float32)
        if rng.random() < 0.2 and m > 0:
            y = np.einsum("bm,bmd->bd", rng.random((B, m)).astype(np.float32), A)  # inside the cone
    At, yt = torch.tensor(A, device="cuda"), torch.tensor(y, device="cuda")
    po, ro = O.batch_project(-y, A)
    qpsolver._tier[(A.shape[1], A.shape[2])] = 2
    try:
        o = cone_op_dense(At, yt, 2, -1.0, 0.2, outputs=OUTS)   # check=True: workspace slices grow until the cones fit
    except qpsolver.HipSolverError as ex:
        st[(kind, "raised")] += 1
        print("RAISED kind", kind, A.shape, str(ex)[:120])
        continue
    o2 = cone_op_dense(At, yt, 2, -1.0, 0.2, outputs=OUTS, check=False)
    stt = o["status"].cpu().numpy()
    for c in stt: st[(kind, int(c))] += 1
    ok = stt == 0
    n += len(stt)
    itmax = max(itmax, int(o["iters"].max()))
    if not ok.any(): continue
    p = o["proj"].cpu().numpy(); r = o["rnorm"].cpu().numpy()
    sc = np.maximum(1.0, np.abs(y).max(axis=1))[:, None]
    e = max(float((np.abs(p - po) / sc)[ok].max()), float((np.abs(r - ro) / np.maximum(1, ro))[ok].max()))
    worst = max(worst, e)
    spread = max(spread, float((o["proj"] - o2["proj"]).abs().max()))
    if e > 4e-6:
        bad += 1
        print("MISMATCH kind", kind, A.shape, "err", e)
    try:
        store = ConeStore.from_dense(At)
        pk = store.cone_op(torch.arange(len(A), device="cuda"), yt, 2, -1.0, 0.2, outputs=OUTS, check=False)
        if store.large:
            dd = float((pk["proj"] - o["proj"]).abs().max())
            if dd > 4e-6:
                bad += 1
                print("PACKED != DENSE kind", kind, A.shape, dd)
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print("PACK FAILED kind", kind, A.shape, repr(ex)[:200])
print(f"instances {n} mismatches {bad} worst err {worst:.2e} launch-to-launch spread {spread:.2e} max iters {itmax}")
print("status by (kind, code):", dict(sorted(st.items(), key=str)))
