"""Diagnostic: TSP-50 (BASELINE configs[2], B = 512, CaVE Exact) on the LDS path (default) and forced onto the large path
(dense LDL^T in LDS + Schur loop, cone read in place from the store)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import synth, _lib
from cave_amd.dataset import ConeStore
dev = torch.device("cuda"); lib = _lib.load()
B = 512
items, costs, _ = synth.coo_batch("tsp", 50, B, seed=0)
d = costs.shape[1]; m_max = max(it[3] for it in items)
store = ConeStore.from_chunks_lazy(lambda i: synth.densify_on(items[i:i + 16], d, dev, m_max), list(range(0, B, 16)))
ids = torch.arange(B, device=dev)
g = torch.Generator(device="cpu").manual_seed(1)
pred = torch.tensor(costs, device=dev) + 0.05 * torch.randn(B, d, generator=g).to(dev)
def timed(tag):
    o = store.cone_op(ids, pred, _lib.MODE_EXACT, -1.0, 0.2, outputs=("loss", "grad"))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); store.cone_op(ids, pred, _lib.MODE_EXACT, -1.0, 0.2, check=False, outputs=("loss", "grad")); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)[5]
    print(f"{tag}: {t:.3f} ms, {B / t * 1e3:.0f} proj/s, iters mean {o['iters'].float().mean():.2f} max {int(o['iters'].max())}, rows {store.max_rows}, lds {store.large_lds if store.large else store.lds_bytes}", flush=True)
    return o
o1 = timed("LDS path (default)")
store.large = True
store.band_entries, store.max_bw = store._max_band_entries()
store.large_lds = int(lib.cave_hip_packed_large_lds_bytes(int(store.max_rows), int(store.max_bw)))
for w in (0, 4, 2):
    store.large_waves = w
    o2 = timed(f"large path, waves {w or 'auto'}")
print("max |dgrad|", float((o1["grad"] - o2["grad"]).abs().max()), "max |dloss|", float((o1["loss"] - o2["loss"]).abs().max()))
