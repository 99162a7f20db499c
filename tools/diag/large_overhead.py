"""Diagnostic: host-side overhead of a large-path call (B = 1) and kernel-only time via HIP events."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import synth, _lib
from cave_amd.dataset import ConeStore
dev = torch.device("cuda")
ctrs, pred, _ = synth.tsp_batch(100, 8, seed=0)
st = ConeStore.from_dense(torch.tensor(ctrs, device=dev), chunk=4)
p0 = torch.tensor(pred, device=dev)
for B in (1, 8, 64, 256):
    ids = torch.arange(B, device=dev) % 8
    p = p0[ids] + 0.01 * torch.randn(B, p0.shape[1], device=dev)
    for _ in range(3): st.cone_op(ids, p, 2, -1.0, outputs=("loss", "grad"))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): st.cone_op(ids, p, 2, -1.0, outputs=("loss", "grad"), check=False)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t) / 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): st.cone_op(ids, p, 2, -1.0, outputs=("loss", "grad"), check=False)
    e1.record(); torch.cuda.synchronize()
    t = time.perf_counter(); _lib.large_slots(dev, B, 1 << 20); tm = time.perf_counter() - t
    print(f"B={B}: wall {wall*1e3:.2f} ms/step, GPU events {e0.elapsed_time(e1)/5:.2f} ms/step, large_slots() {tm*1e3:.3f} ms")
