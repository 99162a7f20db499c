"""Diagnostic: one banded inequality-row case through the dense operator (status, iterations, error vs the oracle)."""
import sys, os
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests")]
import numpy as np, torch
from cave_amd import _lib
if os.environ.get('CAVE_SO'): _lib.LIB_PATH = os.path.abspath(os.environ['CAVE_SO'])
from cave_amd.qpsolver import cone_op_dense
from cave_amd import qpsolver
from oracle import cave_oracle as O
from test_gpu_round2 import _banded_inequality_cones
from cave_amd import synth
for (m, width, shift, pairs) in [(200, 4, 1, 9), (200, 4, 1, 0), (90, 4, 1, 0), (90, 4, 1, 5), (200, 5, 1, 0), (200, 6, 1, 9), (150, 4, 1, 0), (-66, 3, 0, 0), (-3, 66, 0, 0)]:
    if m < 0:
        A, y, _ = synth.sp_batch(-m, width, 6, seed=3)
    else:
        A, y = _banded_inequality_cones(6, m, width, shift, seed=m + width, pairs=pairs)
    po, ro = O.batch_project(y, A)
    qpsolver._tier[(A.shape[1], A.shape[2])] = 2
    o = cone_op_dense(torch.tensor(A, device="cuda"), torch.tensor(y, device="cuda"), 0, 1.0, 0.0, check=False, outputs=("proj", "rnorm", "status", "iters") if False else ("proj", "rnorm"))
    print((m, width, shift, pairs), "status", o["status"].cpu().numpy().tolist(), "iters", o["iters"].cpu().numpy().tolist(),
          "err %.2e" % np.abs(o["proj"].cpu().numpy() - po).max(), flush=True)
