"""Workload for rocprofv3: BASELINE configs 2 / 3 / 4 at their per-GPU batch on the packed store, every batch slot a
DISTINCT cone (coordinate-form generator, densified on the GPU a chunk at a time).
    python tools/diag/large_profile.py tsp50|tsp100|sp30 [B] [steps]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import synth, _lib
if os.environ.get("CAVE_SO"): _lib.LIB_PATH = os.path.abspath(os.environ["CAVE_SO"])  # a diagnostic variant of the library
from cave_amd.dataset import ConeStore
which = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if which == "sp30" else 512)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda")
kind, size, chunk = {"tsp100": ("tsp", 100, 4), "tsp50": ("tsp", 50, 32), "sp30": ("sp", (30, 30), 32)}[which]
MODE = _lib.MODE_EXACT if which == "tsp50" else _lib.MODE_INNER  # (configs[2] is CaVE Exact)
items, costs, _ = synth.coo_batch(kind, size, B, seed=0)
d = costs.shape[1]; m_max = max(it[3] for it in items)
store = ConeStore.from_chunks_lazy(lambda i: synth.densify_on(items[i:i + chunk], d, dev, m_max), list(range(0, B, chunk)))
ids = torch.arange(B, device=dev)
if os.environ.get("LARGE_WAVES"): store.large_waves = int(os.environ["LARGE_WAVES"])
torch.manual_seed(0)
pred = torch.tensor(costs, device=dev) + 0.05 * torch.randn(B, d, device=dev)
o = store.cone_op(ids, pred, MODE, -1.0, 0.2, outputs=("loss", "grad"))
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps): store.cone_op(ids, pred, MODE, -1.0, 0.2, check=False, outputs=("loss", "grad"))
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
for a, b in ev:
    a.record(); store.cone_op(ids, pred, MODE, -1.0, 0.2, check=False, outputs=("loss", "grad")); b.record()
torch.cuda.synchronize()
print("  kernel (HIP events) ms:", [round(a.elapsed_time(b), 3) for a, b in ev], "waves", store.large_waves or "auto", "lds",
      store.large_lds if store.large else (store.lds_bytes_diet or store.lds_bytes))
print(f"{which} B={B} distinct cones: {dt*1e3:.2f} ms/step, {B/dt:.0f} proj/s, iters mean {o['iters'].float().mean():.2f} max {int(o['iters'].max())}, "
      f"rows {store.max_rows} bw {store.max_bw if store.large else '-'} algorithmic bytes {store.algorithmic_bytes(ids)}")
