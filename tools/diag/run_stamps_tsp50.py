"""Diagnostic: per-phase cycle shares of cone_packed_kernel<BlockCtx<4,true>> on TSP-50 (B = 512), ordinary layout (one
workgroup per compute unit) and diet layout (two).  Stamps build; never quote its run time."""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcave_hip_stamps.so")
from cave_amd.dataset import ConeStore
lib = _lib.load()
dev = torch.device("cuda")
B = 512
items, costs, _ = synth.coo_batch("tsp", 50, B, seed=0)
d = costs.shape[1]; m_max = max(it[3] for it in items)
store = ConeStore.from_chunks_lazy(lambda i: synth.densify_on(items[i:i + 32], d, dev, m_max), list(range(0, B, 32)))
ids = torch.arange(B, device=dev)
g = torch.Generator(device="cpu").manual_seed(1)
pred = torch.tensor(costs, device=dev) + 0.05 * torch.randn(B, d, generator=g).to(dev)
names = ["-", "-", "grad+pgn", "hessian", "inner misc (rhs/ratio/matvec/update)", "solve_spd (GJ)", "ls setup + gather q", "ls dphi loop + theta update", "gather r + f"]
buf = (C.c_ulonglong * (16 * 8192))()
for tag, mb in (("ordinary layout", 1 << 30), ("diet layout", 0)):
    store.diet_min_batch = mb
    for _ in range(3): o = store.cone_op(ids, pred, 1, -1.0, 0.2, outputs=("loss", "grad"))
    lib.cave_hip_debug_stamps(buf, 8192)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 16).astype(np.float64)[:B]
    it = o["iters"].cpu().numpy().astype(np.float64)
    mean = a.mean(0)
    print(f"{tag}: iters mean {it.mean():.2f} max {it.max():.0f}; {mean[14]:.0f} cycles per instance (max {a[:,14].max():.0f}); {mean[15]/100:.1f} us mean, {a[:,15].max()/100:.1f} us max")
    for i, n in enumerate(names):
        if n != "-": print(f"  {n:45s} {mean[i]:10.0f}  {100*mean[i]/mean[14]:5.1f}%   per iteration {np.mean(a[:,i]/np.maximum(it,1)):8.0f}")
    loop = a[:, 2:9].sum(1)
    print(f"  outside the Newton loop {np.mean(a[:,14]-loop):.0f}")
    t0 = a[:, 0].min()
    st = (a[:, 0] - t0) / 100.0
    print(f"  start (us) quantiles {np.percentile(st,[0,50,90,100]).round(1).tolist()}; end {np.percentile(st + a[:,15]/100.0,[0,50,90,100]).round(1).tolist()}")
