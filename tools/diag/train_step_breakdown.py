"""Diagnostic: where an eager training step (Linear + CaVE+ + Adam, TSP-20, B=1024, packed cones) spends its time."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import synth
from cave_amd.cave import EPO, innerConeAlignedCosine
from cave_amd.dataset import ConeStore, PackedBatch
dev = torch.device("cuda")
ctrs, costs, _ = synth.tsp_batch(20, 1000, seed=0)
ids = torch.arange(1024, device=dev) % 1000
store = ConeStore.from_dense(torch.tensor(ctrs))
class M: modelSense = EPO.MINIMIZE
x = torch.randn(1024, 10, device=dev)
reg = torch.nn.Linear(10, costs.shape[1]).to(dev)
opt = torch.optim.Adam(reg.parameters(), lr=1e-2)
batch = PackedBatch(store, ids)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, kw in (("check=True (default)", None), ("check=False", {"check": False})):
    cave = innerConeAlignedCosine(M(), solver="hip", seed=0, solver_kwargs=kw)
    def full():
        loss = cave(reg(x), batch); opt.zero_grad(); loss.backward(); opt.step()
    def fwd():
        with torch.no_grad(): cave(reg(x), batch)
    def fwdbwd():
        loss = cave(reg(x), batch); loss.backward()
    print(f"{name}: full step {t(full):.3f} ms | forward only {t(fwd):.3f} | fwd+bwd {t(fwdbwd):.3f} | linear fwd {t(lambda: reg(x)):.3f} | raw op {t(lambda: store.cone_op(ids, reg(x).detach(), 2, -1.0, 0.2, outputs=('loss','grad'), check=(kw is None))):.3f}")
