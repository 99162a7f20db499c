"""How the fused launch responds to the pack half's work: one launch = solve of a packed batch of 1024 TSP-20 cones +
pack of k cones of the next batch, k = 0 .. 1024 (rotating dense batches: HBM).  us per launch for each k.
    python tools/diag/step_pack_share.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from cave_amd import _lib, synth, qpsolver
if os.environ.get("CAVE_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["CAVE_LIB"])
dev = torch.device("cuda", 0); _lib.load()
R, B = 4, 1024
ctrs_np, costs_np, _ = synth.tsp_batch(20, R * B, seed=0)
rng = np.random.default_rng(1234)
dense = [torch.tensor(ctrs_np[r * B:(r + 1) * B], device=dev) for r in range(R)]
pred = [torch.tensor(costs_np[r * B:(r + 1) * B] + rng.normal(0, 0.05, (B, costs_np.shape[1])).astype(np.float32), device=dev) for r in range(R)]
d = dense[0].shape[2]
solve = [qpsolver.prepare_dense(dense[r]) for r in range(2)]       # two packed batches to solve from
sink = qpsolver._LiteSlots(dev, B, d)                              # where the pack half writes
out = {"loss": torch.empty(B, device=dev), "grad": torch.empty(B, d, device=dev)}
st, it = torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
def launch(i, k):
    s = solve[i % 2]
    qpsolver._launch_step(s.store, pred[i % 2], B, _lib.MODE_INNER, -1.0, 0.2, 0, out, st, it, dense[i % R][:k] if k else None, sink if k else None)
for k in (0, 256, 512, 768, 1024, 1536 if False else 1024):
    for i in range(10): launch(i, k)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 200
    for i in range(n): launch(i, k)
    torch.cuda.synchronize()
    print(f"solve 1024 + pack {k:5d}: {1e6 * (time.perf_counter() - t) / n:7.1f} us per launch", flush=True)
assert bool((st == 0).all())
