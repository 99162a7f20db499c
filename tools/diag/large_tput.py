"""Diagnostic: throughput of the large-cone path with the GPU filled."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import synth, _lib
if os.environ.get('CAVE_SO'): _lib.LIB_PATH = os.path.abspath(os.environ['CAVE_SO'])
from cave_amd.dataset import ConeStore

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n

dev = torch.device("cuda")
which, nuniq, Bs = sys.argv[1], int(sys.argv[2]), [int(x) for x in sys.argv[3].split(",")]
if which == "sp30": ctrs, pred, _ = synth.sp_batch(30, 30, nuniq, seed=0)
else: ctrs, pred, _ = synth.tsp_batch(100, nuniq, seed=0)
c = torch.tensor(ctrs, device=dev); p0 = torch.tensor(pred, device=dev)
t0 = time.time(); st = ConeStore.from_dense(c, chunk=8 if which != "sp30" else 256); torch.cuda.synchronize()
print(f"{which}: pack {nuniq} instances {time.time()-t0:.2f}s; store {st.nbytes()/1e6:.1f} MB, rows {st.max_rows} bw {st.max_bw} lds {st.large_lds}", flush=True)
del c
if os.environ.get("LDS"): st.large_lds = int(os.environ["LDS"])
for B in Bs:
    torch.manual_seed(1234 + B)
    ids = torch.arange(B, device=dev) % nuniq
    p = p0[ids] + 0.01 * torch.randn(B, p0.shape[1], device=dev)
    o = st.cone_op(ids, p, 2, -1.0, outputs=("loss", "grad"))
    t = timed(lambda: st.cone_op(ids, p, 2, -1.0, outputs=("loss", "grad")))
    print(f"{which} packed-large B={B}: {t*1e3:.1f} ms/step -> {B/t:.0f} proj/s; iters mean {o['iters'].float().mean():.1f} max {o['iters'].max().item()}", flush=True)
