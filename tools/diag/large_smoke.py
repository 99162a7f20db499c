"""Diagnostic: large-cone path on the GPU — agreement with the fast path / oracle and first timings."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth, qpsolver
from cave_amd.qpsolver import cone_op_dense
from cave_amd.dataset import ConeStore

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n

dev = torch.device("cuda")
# 1. TSP-20 through both paths
ctrs, pred, _ = synth.tsp_batch(20, 64, seed=0)
c = torch.tensor(ctrs, device=dev); p = torch.tensor(pred, device=dev)
fast = cone_op_dense(c, p, 0, -1.0)
qpsolver._tier[(c.shape[1], c.shape[2])] = 2
large = cone_op_dense(c, p, 0, -1.0)
qpsolver._tier.pop((c.shape[1], c.shape[2]))
print("tsp20 large vs fast: proj", (large["proj"] - fast["proj"]).abs().max().item(), "rnorm",
      (large["rnorm"] - fast["rnorm"]).abs().max().item(), "iters", large["iters"].max().item(), fast["iters"].max().item(), flush=True)
# 2. SP 12x12 vs oracle
from oracle import cave_oracle as O
ctrs, pred, _ = synth.sp_batch(12, 12, 8, seed=0)
c = torch.tensor(ctrs, device=dev); p = torch.tensor(pred, device=dev)
o = cone_op_dense(c, p, 0, -1.0)
po, ro = O.batch_project(-pred, ctrs)
print("sp12 tier", qpsolver._tier, "err", np.abs(o["proj"].cpu().numpy() - po).max(), np.abs(o["rnorm"].cpu().numpy() - ro).max(),
      "iters", o["iters"].tolist(), flush=True)
# 3. SP 30x30
B = 128
ctrs, pred, _ = synth.sp_batch(30, 30, B, seed=0)
c = torch.tensor(ctrs, device=dev); p = torch.tensor(pred, device=dev)
o = cone_op_dense(c, p, 0, -1.0)
print("sp30 status", torch.bincount(o["status"]).tolist(), "iters mean/max", o["iters"].float().mean().item(), o["iters"].max().item(),
      "hint", qpsolver._large_hint, flush=True)
t = timed(lambda: cone_op_dense(c, p, 0, -1.0))
print(f"sp30 dense-large B={B}: {t*1e3:.1f} ms/step -> {B/t:.0f} proj/s", flush=True)
st = ConeStore.from_dense(c)
ids = torch.arange(B, device=dev)
o2 = st.cone_op(ids, p, 0, -1.0)
print("sp30 packed large", st.large, "rows", st.max_rows, "band", st.band_entries, "diff", (o2["proj"] - o["proj"]).abs().max().item(), flush=True)
t = timed(lambda: st.cone_op(ids, p, 0, -1.0))
print(f"sp30 packed-large B={B}: {t*1e3:.1f} ms/step -> {B/t:.0f} proj/s", flush=True)
del c, st
# 4. TSP-100
B = 16
ctrs, pred, _ = synth.tsp_batch(100, B, seed=0)
c = torch.tensor(ctrs, device=dev); p = torch.tensor(pred, device=dev)
o = cone_op_dense(c, p, 0, -1.0)
print("tsp100 status", torch.bincount(o["status"]).tolist(), "iters", o["iters"].tolist(), "hint", qpsolver._large_hint, flush=True)
t = timed(lambda: cone_op_dense(c, p, 0, -1.0))
print(f"tsp100 dense-large B={B}: {t*1e3:.1f} ms/step -> {B/t:.0f} proj/s", flush=True)
st = ConeStore.from_dense(c, chunk=8)
ids = torch.arange(B, device=dev)
o2 = st.cone_op(ids, p, 0, -1.0)
print("tsp100 packed large", st.large, "rows", st.max_rows, "band", st.band_entries, "diff", (o2["proj"] - o["proj"]).abs().max().item(), flush=True)
t = timed(lambda: st.cone_op(ids, p, 0, -1.0))
print(f"tsp100 packed-large B={B}: {t*1e3:.1f} ms/step -> {B/t:.0f} proj/s", flush=True)
