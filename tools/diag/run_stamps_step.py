"""Diagnostic: per-phase cycle shares and the workgroup timeline of the fused step kernel (cave_hip_cone_step) at
TSP-20 / B = 1024: pack-only launch, solve-only launch, fused launch.  Stamps build (tools/diag/build_stamps.sh);
never quote its run time."""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMPS_SO", "libcave_hip_stamps.so"))
from cave_amd import qpsolver
from cave_amd.qpsolver import prepare_dense, cone_op_prepared
lib = _lib.load()
B = 1024
ctrs, costs, _ = synth.tsp_batch(20, 2 * B, seed=0)
rng = np.random.default_rng(1234)
cA = torch.tensor(ctrs[:B], device="cuda"); cB = torch.tensor(ctrs[B:], device="cuda")
pA = torch.tensor(costs[:B] + rng.normal(0, 0.05, size=costs[:B].shape).astype(np.float32), device="cuda")
N = 8192
buf = (C.c_ulonglong * (16 * N))()
q = [0, 50, 90, 99, 100]

def stamps():
    lib.cave_hip_debug_stamps(buf, N)
    return np.frombuffer(buf, dtype=np.uint64).reshape(N, 16).astype(np.float64).copy()

def pack_report(a, t0, title):
    mean = a.mean(0)
    print(f"{title}: {mean[14]:.0f} cycles per workgroup (max {a[:,14].max():.0f})")
    for i, n in [(10, "scan (HBM stream + compaction)"), (0, "scan_and_build total"), (11, "  classify rows"), (12, "  pairing"),
                 (13, "  var list + CSR/CSC"), (1, "avg + lite structures + store writes")]:
        print(f"  {n:40s} {mean[i]:10.0f}  {100 * mean[i] / mean[14]:5.1f}%")
    st, en = (a[:, 15] - t0) / 100.0, (a[:, 9] - t0) / 100.0
    print(f"  workgroup start (us) quantiles {q}: {np.percentile(st, q).round(1).tolist()}")
    print(f"  workgroup end   (us)               : {np.percentile(en, q).round(1).tolist()}")
    print(f"  per-workgroup duration (us)        : {np.percentile(en - st, q).round(1).tolist()}")
    print(f"  scan cycles quantiles: {np.percentile(a[:, 10], q).round().tolist()}")

def solve_report(a, it, t0, title):
    mean = a.mean(0)
    names = ["-", "-", "grad+pgn", "hessian", "model step", "-", "ls setup + gather q", "ls dphi + theta", "gather r + f"]
    print(f"{title}: iters mean {it.mean():.2f} max {it.max():.0f}; {mean[14]:.0f} cycles per instance (max {a[:,14].max():.0f}), "
          f"{mean[14] / max(mean[15], 1) / 10:.3f} GHz")
    for i, n in enumerate(names):
        if n != "-": print(f"  {n:40s} {mean[i]:10.0f}  {100 * mean[i] / mean[14]:5.1f}%")
    loop = a[:, 2:9].sum(1)
    print(f"  Newton loop cycles per iteration: mean {np.mean(loop / np.maximum(it, 1)):.0f}; prologue+epilogue {np.mean(a[:,14] - loop):.0f}")
    st = (a[:, 0] - t0) / 100.0
    print(f"  start (us) quantiles {q}: {np.percentile(st, q).round(1).tolist()}; end: {np.percentile(st + a[:,15]/100.0, q).round(1).tolist()}")
    order = np.argsort(-a[:, 14])[:6]
    print("  slowest instances (cycles, iters, grad/hess/model/-/ls-setup/ls-loop/resid, outside-loop):")
    for i in order:
        ph = a[i, 2:9]
        print(f"    {a[i,14]:8.0f} it {it[i]:2.0f}  " + " ".join(f"{x:7.0f}" for x in ph) + f"  out {a[i,14] - ph.sum():7.0f}")

for _ in range(3): prepA = prepare_dense(cA)
a = stamps()[4096:4096 + B]
pack_report(a, a[:, 15].min(), "pack-only launch (four two-wave workgroups per CU)")
for _ in range(3): o = cone_op_prepared(prepA, pA, 2, -1.0, 0.2, check=False, outputs=("loss", "grad"))
a = stamps()[:B]
it = o["iters"].cpu().numpy().astype(np.float64)
solve_report(a, it, a[:, 0].min(), "solve-only launch")
for _ in range(3):
    prepA = prepare_dense(cA)
    o = cone_op_prepared(prepA.then(cB), pA, 2, -1.0, 0.2, check=False, outputs=("loss", "grad"))
a = stamps()
t0 = a[:B, 0].min()
solve_report(a[:B], o["iters"].cpu().numpy().astype(np.float64), t0, "FUSED launch, solve half")
pack_report(a[4096:4096 + B], t0, "FUSED launch, pack half")
