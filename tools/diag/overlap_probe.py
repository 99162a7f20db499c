"""Diagnostic: can the pack kernel of one batch run beside the solve kernel of another (two streams, no dependency)?"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth, qpsolver
from cave_amd.qpsolver import prepare_dense, cone_op_prepared
if len(sys.argv) > 1: qpsolver.SPLIT_NNZ = int(sys.argv[1])
if len(sys.argv) > 2: qpsolver.PIPE_PACK_WAVES = int(sys.argv[2])
dev = torch.device("cuda", 0); torch.cuda.set_device(0); _lib.load()
ctrs_np, costs_np, _ = synth.tsp_batch(20, 2048, seed=0)
A = torch.tensor(ctrs_np[:1024], device=dev); Bc = torch.tensor(ctrs_np[1024:], device=dev)
pred = torch.tensor(costs_np[:1024], device=dev)
prepA = prepare_dense(A); torch.cuda.synchronize()
side = qpsolver._side_streams[dev]
def solve(): return cone_op_prepared(prepA, pred, _lib.MODE_INNER, -1.0, 0.2, check=False, outputs=("loss", "grad"))
lib = _lib.load()
ss = prepare_dense(Bc).store; torch.cuda.synchronize()
def pack():
    with torch.cuda.stream(side):
        lib.cave_hip_pack_fill(_lib.ptr(Bc), 1024, Bc.shape[1], Bc.shape[2], 0, 0, qpsolver.PIPE_PACK_WAVES, ss.ref, 0, _lib.ptr(ss.pack_status), qpsolver.C_void(side.cuda_stream))
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
ts = timed(solve); tp = timed(pack)
def both_solve_first(): solve(); pack()
def both_pack_first(): pack(); solve()
print(f"SPLIT_NNZ {qpsolver.SPLIT_NNZ} pack waves {qpsolver.PIPE_PACK_WAVES}: solve alone {ts:.1f} us, pack alone {tp:.1f} us (each incl. a host sync), solve then pack on two streams {timed(both_solve_first):.1f} us, pack then solve {timed(both_pack_first):.1f} us")
