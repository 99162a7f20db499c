"""Workload for rocprofv3 --pmc passes on the step kernel: 6 pack-only launches (grid 1024), then 6 solve-only launches
(grid 1023: told apart by grid size in the counter CSV), then 6 fused launches (grid 2047)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from cave_amd import _lib, synth
from cave_amd.qpsolver import prepare_dense, cone_op_prepared
B = 1024
ctrs, costs, _ = synth.tsp_batch(20, 2 * B, seed=0)
cA = torch.tensor(ctrs[:B], device="cuda"); cB = torch.tensor(ctrs[B:], device="cuda"); cS = cA[:B - 1].contiguous()
p = torch.tensor(costs[:B], device="cuda")
for _ in range(6): prepare_dense(cA)
pS = prepare_dense(cS)
for _ in range(6): cone_op_prepared(pS, p[:B - 1], 2, -1.0, 0.2, check=False, outputs=("loss", "grad"))
for _ in range(6):
    pr = prepare_dense(cS)
    cone_op_prepared(pr.then(cB), p[:B - 1], 2, -1.0, 0.2, check=False, outputs=("loss", "grad"))
torch.cuda.synchronize()
