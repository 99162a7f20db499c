"""Diagnostic: warm start on the large-cone path (SP 30x30, packed store): iterations and time of a cold step and of
steps whose predictions drift a little (the training situation)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import synth
from cave_amd.dataset import ConeStore
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
drift = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
dev = torch.device("cuda")
c, y, _ = synth.sp_batch(30, 30, 256, seed=0)
st = ConeStore.from_dense(torch.tensor(c, device=dev), chunk=256)
ids = torch.arange(B, device=dev) % 256
torch.manual_seed(0)
p = torch.tensor(y, device=dev)[ids] + 0.05 * torch.randn(B, y.shape[1], device=dev)
def step(pp):
    torch.cuda.synchronize(); t = time.time()
    o = st.cone_op(ids, pp, 2, -1.0, outputs=("loss", "grad"))
    torch.cuda.synchronize()
    return o, (time.time() - t) * 1e3
o, t = step(p); o, t = step(p)
print(f"cold: {t:.2f} ms, iters mean {o['iters'].float().mean():.2f} max {int(o['iters'].max())}", flush=True)
ref = o["loss"].clone()
# ids repeat cones (B > 256 distinct): warm slots are per cone, so use distinct ids only when B <= 256
st.enable_warm_start()
for it in range(5):
    p = p + drift * torch.randn_like(p)
    o, t = step(p)
    print(f"warm step {it}: {t:.2f} ms, iters mean {o['iters'].float().mean():.2f} max {int(o['iters'].max())}, status ok {bool((o['status'] == 0).all())}", flush=True)
st.enable_warm_start(False)
o2, t = step(p)
print(f"cold again: {t:.2f} ms, iters mean {o2['iters'].float().mean():.2f}; |loss warm - cold| max {float((o['loss'] - o2['loss']).abs().max()):.2e}")
