"""Diagnostic: per-phase cycle shares of the split dense operator (pack kernel, 4 waves, slot mode + solve kernel,
1 wave) at TSP-20 / B = 1024.  Stamps build (tools/diag/build_stamps.sh); never quote its run time."""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMPS_SO", "libcave_hip_stamps.so"))
from cave_amd import qpsolver
lib = _lib.load()
B = 1024
ctrs, costs, _ = synth.tsp_batch(20, B, seed=0)
c = torch.tensor(ctrs, device="cuda"); p = torch.tensor(costs, device="cuda")
_, m, d = c.shape
ss = qpsolver._SlotStore(c.device, B, d)
buf = (C.c_ulonglong * (16 * B))()

def stamps():
    lib.cave_hip_debug_stamps(buf, B)
    return np.frombuffer(buf, dtype=np.uint64).reshape(B, 16).astype(np.float64).copy()

for _ in range(3):
    _lib.check(lib.cave_hip_pack_fill(_lib.ptr(c), B, m, d, 0, 0, 4, ss.ref, 0, _lib.ptr(ss.pack_status), _lib.current_stream()), "pack")
a = stamps()
mean = a.mean(0)
print(f"pack kernel (4 waves, slot mode): {mean[14]:.0f} cycles per instance (max {a[:,14].max():.0f})")
for i, n in [(10, "scan (HBM stream + compaction)"), (0, "scan_and_build total"), (11, "  classify rows"), (12, "  pairing"),
             (13, "  var list + CSR/CSC"), (1, "avg + store writes")]:
    print(f"  {n:40s} {mean[i]:10.0f}  {100 * mean[i] / mean[14]:5.1f}%")
t0 = a[:, 15].min()
st, en = (a[:, 15] - t0) / 100.0, (a[:, 9] - t0) / 100.0
q = [0, 50, 90, 99, 100]
print(f"  workgroup start (us after the first) quantiles {q}: {np.percentile(st, q).round(1).tolist()}")
print(f"  workgroup end   (us after the first start)         : {np.percentile(en, q).round(1).tolist()}")
print(f"  scan cycles quantiles: {np.percentile(a[:, 10], q).round().tolist()}; total cycles: {np.percentile(a[:, 14], q).round().tolist()}")
print(f"  clock: {np.median(a[:, 14] / np.maximum((a[:, 9] - a[:, 15]), 1) / 10):.3f} GHz (cycles / memrealtime)")
out = {k: torch.empty((B,) if k == "loss" else (B, d), dtype=torch.float32, device="cuda") for k in ("loss", "grad")}
status = torch.empty(B, dtype=torch.int32, device="cuda"); iters = torch.empty(B, dtype=torch.int32, device="cuda")
for _ in range(3):
    _lib.check(lib.cave_hip_cone_packed(ss.ref, None, _lib.ptr(p), B, 2, -1.0, 0.2, 0, ss.lds_bytes, 1, None, None, None,
                                        _lib.ptr(out["loss"]), _lib.ptr(out["grad"]), _lib.ptr(status), _lib.ptr(iters),
                                        _lib.current_stream()), "packed")
a = stamps()
it = iters.cpu().numpy().astype(np.float64)
mean = a.mean(0)
names = ["-", "load y/avg", "grad+pgn", "hessian", "inner misc", "solve_spd (GJ)", "ls setup + gather q", "ls dphi + theta", "gather r + f", "solve_and_finish total"]
print(f"solve kernel (1 wave): iters mean {it.mean():.2f} max {it.max():.0f}; {mean[14]:.0f} cycles per instance (max {a[:,14].max():.0f}), "
      f"{mean[14] / max(mean[15], 1) / 10:.3f} GHz")
for i, n in enumerate(names):
    print(f"  {n:40s} {mean[i]:10.0f}  {100 * mean[i] / mean[14]:5.1f}%")
solve = a[:, 2:9].sum(1)
print(f"  Newton loop cycles per iteration: mean {np.mean(solve / np.maximum(it, 1)):.0f}; prologue+epilogue {np.mean(a[:,14] - solve):.0f}")
w = int(a[:, 14].argmax())
print(f"  worst instance: {a[w,14]:.0f} cycles, {it[w]:.0f} iterations; quantiles of total cycles 50/90/99/100: "
      f"{np.percentile(a[:,14], [50, 90, 99, 100]).round().tolist()}")
order = np.argsort(-a[:, 14])[:10]
print("  slowest instances (cycles, iters, then per-phase grad/hess/inner/GJ/ls-setup/ls-loop/resid, outside-loop):")
for i in order:
    ph = a[i, 2:9]
    print(f"    {a[i,14]:8.0f} it {it[i]:2.0f}  " + " ".join(f"{x:7.0f}" for x in ph) + f"  out {a[i,14] - ph.sum():7.0f}  GJ/iter {a[i,5]/max(it[i],1):6.0f}")
t0 = a[:, 0].min()
st = (a[:, 0] - t0) / 100.0
print(f"  solve workgroup start quantiles {q} (us): {np.percentile(st, q).round(1).tolist()}; end: {np.percentile(st + a[:,15]/100.0, q).round(1).tolist()}")
