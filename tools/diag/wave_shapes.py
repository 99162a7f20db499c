"""Diagnostic: kernel time (HIP events) of the dense and packed operators per launch shape, TSP-20, B=1024,
rotating batches; production library."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth
from cave_amd.qpsolver import cone_op_dense
from cave_amd.dataset import ConeStore
_lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctrs, costs, _ = synth.tsp_batch(n, 1000, seed=0)
R = 4
rng = np.random.default_rng(1)
bat = []
for r in range(R):
    ids = (np.arange(B) + r * 257) % 1000
    bat.append((torch.tensor(ctrs[ids], device="cuda"), torch.tensor(costs[ids] + rng.normal(0, .05, costs[ids].shape).astype(np.float32), device="cuda"), torch.tensor(ids, device="cuda")))
store = ConeStore.from_dense(torch.tensor(ctrs, device="cuda"))
def timeit(fn, reps=40):
    for i in range(8): fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(i); b.record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
    return np.median(t), t.min()
from cave_amd import qpsolver
cone_op_dense(bat[0][0], bat[0][1], 2, -1.0, 0.2, outputs=("loss", "grad"))  # checked: settles the auto shape
print("auto dense form:", "split (slot pack + one-wave solve)" if qpsolver._split_ok.get((bat[0][0].shape[1], bat[0][0].shape[2])) else "fused")
for waves in (0, 1, 2, 4, 8):
    def fd(i):
        c, p, _ = bat[i % R]
        return cone_op_dense(c, p, 2, -1.0, 0.2, waves=waves, check=False, outputs=("loss", "grad"))
    def fp(i):
        c, p, ids = bat[i % R]
        store.waves = waves
        return store.cone_op(ids, p, 2, -1.0, 0.2, check=False, outputs=("loss", "grad"))
    o = fd(0); st = int((o["status"] != 0).sum())
    md, mnd = timeit(fd); mp, mnp = timeit(fp)
    print(f"waves {waves}: dense median {md:7.1f} us (min {mnd:7.1f})   packed median {mp:7.1f} us (min {mnp:7.1f})   status!=0: {st}  iters mean {o['iters'].float().mean():.2f} max {int(o['iters'].max())}", flush=True)
