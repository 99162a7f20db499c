"""Diagnostic: per-phase cycle shares of cone_dense_kernel (stamps build; never quote its run time)."""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcave_hip_stamps.so")  # built by tools/diag/build_stamps.sh
from cave_amd.qpsolver import cone_op_dense
lib = _lib.load()
ctrs, costs, _ = synth.tsp_batch(20, 1024, seed=0)
c = torch.tensor(ctrs, device="cuda"); p = torch.tensor(costs, device="cuda")
names = ["scan+build", "load y/avg", "grad+pgn", "hessian", "inner misc (rhs/ratio/matvec/update)", "solve_spd (GJ)", "ls setup + gather q", "ls dphi loop + theta update", "gather r + f", "epilogue(+solve total tail)", "  scan only", "  classify rows", "  pairing", "  var list + CSC"]
for WAVES in [int(a) for a in sys.argv[1:]] or [4]:
    mode = 0
    outs = ("proj","rnorm")
    for _ in range(3): cone_op_dense(c, p, mode, -1.0, 0.2, outputs=outs, waves=WAVES)
    o = cone_op_dense(c, p, mode, -1.0, 0.2, outputs=outs, waves=WAVES)
    buf = (C.c_ulonglong * (16*1024))()
    lib.cave_hip_debug_stamps(buf, 1024)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.float64)
    it = o["iters"].cpu().numpy().astype(np.float64)
    mean = a.mean(0); tot = mean[0] + mean[1] + mean[9]
    print(f"waves {WAVES}: iters mean {it.mean():.2f} max {it.max():.0f}; cycles/instance {tot:.0f}")
    for i, n in enumerate(names):
        print(f"  {n:45s} {mean[i]:10.0f}  {100*mean[i]/tot:5.1f}%")
    solve = a[:, 2:9].sum(1)
    print(f"  Newton loop cycles per iteration: mean {np.mean(solve/np.maximum(it,1)):.0f}; worst instance total {a[:,14].max():.0f} ticks, its iters {it[a[:,14].argmax()]:.0f}")
    print(f"  GJ cycles per iteration {np.mean(a[:,5]/np.maximum(it,1)):.0f}; per-iteration others: grad {np.mean(a[:,2]/it):.0f} hess {np.mean(a[:,3]/it):.0f} inner-misc {np.mean(a[:,4]/it):.0f} ls-setup {np.mean(a[:,6]/it):.0f} ls-loop {np.mean(a[:,7]/it):.0f} resid {np.mean(a[:,8]/it):.0f}")
    print(f"  whole instance: {mean[14]:.0f} memtime ticks in {mean[15]/100:.1f} us (memrealtime) -> {mean[14]/(mean[15]/100)/1e3:.3f} GHz", flush=True)
