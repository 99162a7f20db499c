import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from cave_amd import _lib, synth
_lib.LIB_PATH = "/root/repo/tools/diag/libcave_hip_stamps.so"
from cave_amd.qpsolver import prepare_dense
lib = _lib.load()
B = 1024
N = 8192
buf = (C.c_ulonglong * (16 * N))()
def stamps():
    lib.cave_hip_debug_stamps(buf, N)
    return np.frombuffer(buf, dtype=np.uint64).reshape(N, 16).astype(np.float64).copy()
ctrs, costs, _ = synth.tsp_batch(20, B, seed=0)
real = torch.tensor(ctrs, device="cuda")
zeros = torch.zeros_like(real)
unit = real.clone(); unit[(real != 0).sum(2) > 1] = 0
gen = real.clone(); gen[(real != 0).sum(2) == 1] = 0
one = torch.zeros_like(real); one[:, 0, 0] = 1.0
for name, t in (("real", real), ("zeros", zeros), ("unit rows only", unit), ("general rows only", gen), ("one nonzero", one)):
    for _ in range(3): prepare_dense(t)
    a = stamps()[4096:4096 + B]
    print(f"{name:18s} scan cycles mean {a[:,10].mean():8.0f}  total {a[:,14].mean():8.0f}  nnz/inst {float((t != 0).sum()) / B:.0f}")
