import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib
if os.environ.get("CAVE_SO"): _lib.LIB_PATH = os.path.abspath(os.environ["CAVE_SO"])
from cave_amd.qpsolver import cone_op_dense
z = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/diag/noconv_r03.npz")); A, y = z["A"], z["y"]
for w in ([1] if os.environ.get("CAVE_SO") else [1, 2, 4, 8]):
    o = cone_op_dense(torch.tensor(A[None], device="cuda"), torch.tensor(y[None], device="cuda"), 0, 1.0, 0.0, waves=w, check=False)
    torch.cuda.synchronize()
    print("gpu waves", w, "status", o["status"].cpu().numpy(), "iters", o["iters"].cpu().numpy(), "rnorm", o["rnorm"].cpu().numpy(), flush=True)
