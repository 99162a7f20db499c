"""Diagnostic: TSP-50 (BASELINE configs[2], B = 512, CaVE Exact) on the LDS path: ordinary layout (107 KB of LDS, one
workgroup per compute unit, two rounds) against the diet layout (76 KB, two workgroups per compute unit, one round)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import synth, _lib
if os.environ.get("CAVE_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["CAVE_LIB"])
from cave_amd.dataset import ConeStore
dev = torch.device("cuda"); lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
items, costs, _ = synth.coo_batch("tsp", 50, B, seed=0)
d = costs.shape[1]; m_max = max(it[3] for it in items)
store = ConeStore.from_chunks_lazy(lambda i: synth.densify_on(items[i:i + 32], d, dev, m_max), list(range(0, B, 32)))
ids = torch.arange(B, device=dev)
g = torch.Generator(device="cpu").manual_seed(1)
pred = torch.tensor(costs, device=dev) + 0.05 * torch.randn(B, d, generator=g).to(dev)
print("lds ordinary", store.lds_bytes, "diet", store.lds_bytes_diet, "rows", store.max_rows, "nnz", store.max_nnz)
def timed(tag, mode):
    o = store.cone_op(ids, pred, mode, -1.0, 0.2, outputs=("proj", "rnorm", "loss", "grad"))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); store.cone_op(ids, pred, mode, -1.0, 0.2, check=False, outputs=("loss", "grad")); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)[5]
    print(f"{tag}: {t:.3f} ms, {B / t * 1e3:.0f} proj/s, iters mean {o['iters'].float().mean():.2f} max {int(o['iters'].max())}", flush=True)
    return o
for mode, name in ((_lib.MODE_EXACT, "exact"), (_lib.MODE_INNER, "inner")):
    store.diet_min_batch = 1 << 30
    a = timed(f"{name}, ordinary layout", mode)
    store.diet_min_batch = 0
    b = timed(f"{name}, diet layout    ", mode)
    print("   max |d proj|", float((a["proj"] - b["proj"]).abs().max()), "max |d grad|", float((a["grad"] - b["grad"]).abs().max()),
          "iters equal", bool(torch.equal(a["iters"], b["iters"])))
