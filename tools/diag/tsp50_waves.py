"""Diagnostic: TSP-50 (config 3 share) by waves per instance, dense tier-1 and packed."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth
from cave_amd.qpsolver import cone_op_dense, _grow_limits
from cave_amd.dataset import ConeStore
B = 512
ctrs, costs, _ = synth.tsp_batch(50, 64, seed=0)
ids_np = np.arange(B) % 64
c = torch.tensor(ctrs[ids_np], device="cuda"); p = torch.tensor(costs[ids_np], device="cuda") + 0.05 * torch.randn(B, costs.shape[1], device="cuda")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): o = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n, o
cap, lds = _grow_limits(c.shape[1], c.shape[2])
for w in (1, 2, 8):
    dt, o = timeit(lambda: cone_op_dense(c, p, 1, -1.0, 0.0, outputs=("loss", "grad"), nnz_cap=cap, lds_bytes=lds, waves=w, check=False))
    print(f"dense waves={w}: {dt*1e3:.2f} ms -> {B/dt:.0f} proj/s; status {torch.bincount(o['status']).tolist()}", flush=True)
st = ConeStore.from_dense(torch.tensor(ctrs, device="cuda"), chunk=64)
ids = torch.tensor(ids_np, device="cuda")
print("packed lds", st.lds_bytes, "rows", st.max_rows, "nnz", st.max_nnz)
for w in (1, 2, 8):
    st.waves = w
    dt, o = timeit(lambda: st.cone_op(ids, p, 1, -1.0, 0.0, outputs=("loss", "grad"), check=False))
    print(f"packed waves={w}: {dt*1e3:.2f} ms -> {B/dt:.0f} proj/s; status {torch.bincount(o['status']).tolist()}", flush=True)
