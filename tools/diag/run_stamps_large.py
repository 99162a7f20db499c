"""Diagnostic: per-phase cycle shares of cone_packed_large_kernel (stamps build; never quote its run time)."""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMPS_SO", "libcave_hip_stamps.so"))
from cave_amd.dataset import ConeStore
lib = _lib.load()
names = ["scan+build", "load y/avg", "grad+pgn", "hessian", "inner misc", "solve_spd total", "ls setup + gather q",
         "ls dphi loop + theta update", "gather r + f", "epilogue(+solve total tail)", "  band prep (z, window)", "  band factor loop",
         "  band back-subst", "-"]
which = sys.argv[1] if len(sys.argv) > 1 else "sp30"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
if which == "sp30": ctrs, costs, _ = synth.sp_batch(30, 30, B, seed=0)
else: ctrs, costs, _ = synth.tsp_batch(100, B, seed=0)
c = torch.tensor(ctrs, device="cuda"); p = torch.tensor(costs, device="cuda")
st = ConeStore.from_dense(c, chunk=8)
if os.environ.get("LDS"): st.large_lds = int(os.environ["LDS"])
print("large_lds", st.large_lds)
ids = torch.arange(B, device="cuda")
for _ in range(2): o = st.cone_op(ids, p, 0, -1.0)
buf = (C.c_ulonglong * (16 * B))()
lib.cave_hip_debug_stamps(buf, B)
a = np.frombuffer(buf, dtype=np.uint64).reshape(B, 16).astype(np.float64)
mean = a.mean(0); tot = mean[14]
rt = a[:, 15] / 100.0
print("per-instance time (us) quantiles 0/50/90/99/100:", [round(float(np.quantile(rt, q))) for q in (0, .5, .9, .99, 1)], "slowest instances:", np.argsort(rt)[-5:].tolist(), "their iters", o["iters"].cpu().numpy()[np.argsort(rt)[-5:]].tolist())
print(f"{which} B={B}: iters mean {o['iters'].float().mean():.2f}; cycles/instance {tot:.0f} = {mean[15]/100:.1f} us")
if "fine" in os.environ.get("STAMPS_SO", "") and which != "sp30":
    print(f"  dense LDL^T fine stamps (cycles per instance): pivot block on wave 0 {mean[0]:.0f}  wait at barrier 1 {mean[1]:.0f}  trailing update {mean[9]:.0f}  wait at barrier 2 {mean[5]:.0f}")
if "fine" in os.environ.get("STAMPS_SO", "") and which == "sp30":
    print(f"  per pivot (fine stamps): block A {mean[0]/mean[13]:.0f}  B {mean[1]/mean[13]:.0f}  C+D {mean[9]/mean[13]:.0f}  E {mean[11]/mean[13]:.0f}")
if mean[13] > 0: print(f"  per pivot: factor {mean[11]/mean[13]:.0f} cycles, back-subst {mean[12]/mean[13]:.0f} cycles")
for i, n in enumerate(names):
    print(f"  {n:45s} {mean[i]:12.0f}  {100*mean[i]/tot:5.1f}%")

order = np.argsort(rt)
sel = list(order[-3:][::-1]) + [order[len(order) // 2]]
nnz = None
try:
    nnz = st.n_nnz.cpu().numpy() if hasattr(st, "n_nnz") else None
except Exception:
    pass
print("  per-instance phases (cycles): slowest three, then the median instance")
for b in sel:
    row = a[b]
    print(f"    inst {b:4d} [set-up {row[0]:8.0f} solver call {row[1]:9.0f} epilogue {row[9]:8.0f}] iters {int(o['iters'][b])} total {row[14]:9.0f}: grad {row[2]:8.0f} hess {row[3]:8.0f} inner(incl. solve) {row[4]:8.0f} [factor {row[11]:8.0f} backsub {row[12]:7.0f}] ls-setup {row[6]:7.0f} ls-loop {row[7]:7.0f} resid {row[8]:7.0f} other {row[14] - row[2] - row[3] - row[4] - row[5] - row[6] - row[7] - row[8]:8.0f}")
