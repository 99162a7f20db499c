"""Diagnostic: reduced-row counts around the path boundaries (32/33: 4-wave limit, 64/65: fast path / large path)
through the automatic dispatch, dense and packed, against the oracle."""
import sys, os
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests")]
import numpy as np, torch
from cave_amd import qpsolver
from cave_amd.qpsolver import cone_op_dense
from cave_amd.dataset import ConeStore
from oracle import cave_oracle as O
rng = np.random.default_rng(5)
for m in (31, 32, 33, 48, 63, 64, 65, 66, 80):
    d = 40
    A = (rng.standard_normal((6, m, d)) * (rng.random((6, m, d)) < 0.2)).astype(np.float32)
    A[:, :, 0] = 1.0; A[:, :, 1] = rng.standard_normal((6, m))
    y = rng.standard_normal((6, d)).astype(np.float32)
    At, yt = torch.tensor(A, device="cuda"), torch.tensor(y, device="cuda")
    o = cone_op_dense(At, yt, 2, -1.0, 0.2, outputs=("proj", "rnorm", "loss", "grad"))
    po, ro = O.batch_project(-y, A)
    e = np.abs(o["proj"].cpu().numpy() - po).max()
    st = ConeStore.from_dense(At)
    pk = st.cone_op(torch.arange(6, device="cuda"), yt, 2, -1.0, 0.2, outputs=("proj", "loss"))
    e2 = (pk["proj"] - o["proj"]).abs().max().item()
    print(f"m={m}: tier {qpsolver._tier.get((m, d), 0)} wide_ok {qpsolver._wide_ok.get((m, d))} err {e:.1e} packed(large={st.large}, rows {st.max_rows}) diff {e2:.1e} iters max {int(o['iters'].max())}")
