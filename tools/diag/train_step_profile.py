"""Diagnostic: host-side profile (cProfile) of the training step (Linear + CaVE+ + Adam, TSP-20, B=1024, packed cones,
lazy status check) + wall times of its pieces.  The step is host-bound: its kernels take ~0.2 ms."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import synth
from cave_amd.cave import EPO, innerConeAlignedCosine, flush_checks
from cave_amd.dataset import ConeStore, PackedBatch
dev = torch.device("cuda")
ctrs, costs, _ = synth.tsp_batch(20, 1024, seed=0)
ids = torch.arange(1024, device=dev)
store = ConeStore.from_dense(torch.tensor(ctrs))
class M: modelSense = EPO.MINIMIZE
x = torch.randn(1024, 10, device=dev)
reg = torch.nn.Linear(10, costs.shape[1]).to(dev)
opt = torch.optim.Adam(reg.parameters(), lr=1e-2, fused=True)
batch = PackedBatch(store, ids)
def t(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, kw in (("strict", None), ("lazy", {"check": "lazy"}), ("unchecked", {"check": False})):
    cave = innerConeAlignedCosine(M(), solver="hip", seed=0, solver_kwargs=kw)
    def full():
        loss = cave(reg(x), batch); opt.zero_grad(); loss.backward(); opt.step()
    def fwd():
        with torch.no_grad(): cave(reg(x), batch)
    def fwdbwd():
        loss = cave(reg(x), batch); loss.backward()
    def rest():
        loss = reg(x).mean(); opt.zero_grad(); loss.backward(); opt.step()
    print(f"{name:9s}: full step {t(full):.3f} ms | loss forward (no grad) {t(fwd):.3f} | fwd+bwd {t(fwdbwd):.3f} | "
          f"linear + mean + backward + Adam alone {t(rest):.3f}")
    flush_checks()
cave = innerConeAlignedCosine(M(), solver="hip", seed=0, solver_kwargs={"check": "lazy"})
def full():
    loss = cave(reg(x), batch); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(20): full()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): full()
torch.cuda.synchronize(); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(45); print(st.getvalue()[:9000])
