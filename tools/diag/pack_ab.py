"""Diagnostic A/B: pack-kernel time (slot mode, 4 waves) on rotating TSP-20 batches (HIP events).
   [CAVE_SO=variant.so] python tools/diag/pack_ab.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth
if os.environ.get("CAVE_SO"): _lib.LIB_PATH = os.path.abspath(os.environ["CAVE_SO"])
from cave_amd import qpsolver
lib = _lib.load()
N, B = 4096, 1024
ctrs, costs, _ = synth.tsp_batch(20, N, seed=0)
batches = [torch.tensor(ctrs[r * B:(r + 1) * B], device="cuda") for r in range(4)]
_, m, d = batches[0].shape
ss = qpsolver._SlotStore(batches[0].device, B, d)
def pack(c):
    _lib.check(lib.cave_hip_pack_fill(_lib.ptr(c), B, m, d, 0, 0, 4, ss.ref, 0, _lib.ptr(ss.pack_status), _lib.current_stream()), "pack")
for i in range(8): pack(batches[i % 4])
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(80)]
for i, (a, b) in enumerate(ev):
    a.record(); pack(batches[i % 4]); b.record()
torch.cuda.synchronize()
t = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
print(f"pack kernel: median {np.median(t):.1f} us, min {t.min():.1f}, p90 {np.percentile(t, 90):.1f}; status ok {bool((ss.pack_status == 0).all())}; "
      f"{4 * m * d * B / np.median(t) / 1e6:.2f} TB/s of dense bytes")
