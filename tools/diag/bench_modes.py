"""Diagnostic: kernel time of the bench batch by mode / waves (HIP events)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth
from cave_amd.qpsolver import cone_op_dense
_lib.load()
ctrs_np, costs_np, _ = synth.tsp_batch(20, 1000, seed=0)
ids = np.arange(1024) % 1000
rng = np.random.default_rng(1234)
pred_np = costs_np[ids] + rng.normal(0, 0.05, size=costs_np[ids].shape).astype(np.float32)
c = torch.tensor(ctrs_np[ids], device="cuda"); p = torch.tensor(pred_np, device="cuda")
p0 = torch.tensor(costs_np[ids], device="cuda")
def timeit(fn, n=30):
    for _ in range(5): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
for name, pp in (("bench pred (noisy)", p), ("clean pred", p0)):
    for mode, outs in ((0, ("proj", "rnorm")), (2, ("loss", "grad")), (1, ("loss", "grad"))):
        row = []
        for waves in (1, 2, 4):
            o = cone_op_dense(c, pp, mode, -1.0, 0.2, waves=waves, check=False, outputs=outs)
            t = timeit(lambda: cone_op_dense(c, pp, mode, -1.0, 0.2, waves=waves, check=False, outputs=outs))
            row.append(f"w{waves}: {t:.1f} us (iters mean {o['iters'].float().mean():.2f} max {int(o['iters'].max())})")
        print(name, "mode", mode, " | ".join(row), flush=True)
