"""One-off measurement of BASELINE configs[2] shape on ONE GPU: TSP-50, CaVE Exact, batch per GPU
(default 512 distinct instances; the dense batch is 6.5 MB per instance)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cave_amd import _lib, synth
from cave_amd.qpsolver import cone_op_dense
from cave_amd.dataset import ConeStore
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t0 = time.time(); ctrs, costs, _ = synth.tsp_batch(50, B, seed=0); tg = time.time() - t0
c = torch.tensor(ctrs, device="cuda"); p = torch.tensor(costs, device="cuda")
res = {"B": B, "m_max": ctrs.shape[1], "d": ctrs.shape[2], "gen_s": tg}
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): o = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n, o
dt, o = timeit(lambda: cone_op_dense(c, p, _lib.MODE_EXACT, -1.0, 0.0, outputs=("loss", "grad")))
assert bool((o["status"] == 0).all())
res["dense_ms"] = dt * 1e3; res["dense_proj_per_s"] = B / dt; res["iters_mean"] = float(o["iters"].float().mean()); res["iters_max"] = int(o["iters"].max())
res["dense_GBps"] = ctrs.nbytes / dt / 1e9
t0 = time.time(); st = ConeStore.from_dense(c, chunk=256); torch.cuda.synchronize(); res["pack_s"] = time.time() - t0
ids = torch.arange(B, device="cuda")
dt, o = timeit(lambda: st.cone_op(ids, p, _lib.MODE_EXACT, -1.0, 0.0, outputs=("loss", "grad")))
assert bool((o["status"] == 0).all())
res["packed_ms"] = dt * 1e3; res["packed_proj_per_s"] = B / dt; res["store_MB"] = st.nbytes() / 1e6; res["packed_lds"] = st.lds_bytes; res["max_rows"] = st.max_rows; res["max_nnz"] = st.max_nnz
# CPU oracle on 2 instances
from oracle import cave_oracle as O
t0 = time.time(); O.batch_project(-costs[:2], ctrs[:2]); res["cpu_oracle_proj_per_s"] = 2 / (time.time() - t0)
print(json.dumps(res))
