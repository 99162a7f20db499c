"""Diagnostic A/B: solve-kernel time of the packed TSP-20 store at B = 1024 (HIP events), optionally with a variant
library (CAVE_SO).   python tools/diag/packed_ab.py [instances] [reps]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth
if os.environ.get("CAVE_SO"): _lib.LIB_PATH = os.path.abspath(os.environ["CAVE_SO"])
from cave_amd.dataset import ConeStore
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctrs, costs, _ = synth.tsp_batch(20, N, seed=0)
store = ConeStore.from_dense(torch.tensor(ctrs))
rng = np.random.default_rng(1234)
res = []
for r in range(4):
    ids = torch.tensor((np.arange(1024) + r * 1024) % N, device="cuda")
    pred = torch.tensor(costs[ids.cpu().numpy()] + rng.normal(0, 0.05, (1024, costs.shape[1])).astype(np.float32), device="cuda")
    for _ in range(3): o = store.cone_op(ids, pred, 2, -1.0, 0.2, check=False, outputs=("loss", "grad"))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); store.cone_op(ids, pred, 2, -1.0, 0.2, check=False, outputs=("loss", "grad")); b.record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
    res.append(float(np.median(t)))
    print(f"batch {r}: median {np.median(t):.1f} us (min {t.min():.1f}), iters mean {o['iters'].float().mean():.2f} max {int(o['iters'].max())}, status ok {bool((o['status']==0).all())}")
print("mean of medians %.1f us; lds %d" % (np.mean(res), store.lds_bytes))
