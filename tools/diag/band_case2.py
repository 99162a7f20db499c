"""Diagnostic: banded inequality-row cones through the packed store at each workgroup shape."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, os.path.join(R, "tests")]
import numpy as np, torch
from cave_amd import _lib
if os.environ.get('CAVE_SO'): _lib.LIB_PATH = os.path.abspath(os.environ['CAVE_SO'])
from cave_amd.dataset import ConeStore
from oracle import cave_oracle as O
from test_gpu_round2 import _banded_inequality_cones
for (m, width, shift, pairs) in [(200, 4, 1, 9), (200, 4, 1, 0), (90, 5, 1, 0), (130, 6, 1, 5)]:
    A, y = _banded_inequality_cones(6, m, width, shift, seed=m + width, pairs=pairs)
    po, ro = O.batch_project(y, A)
    st = ConeStore.from_dense(torch.tensor(A, device="cuda"), chunk=6)
    for lds in (st.large_lds, 65536):
        st.large_lds = lds
        for w in (4, 2, 1):
            st.large_waves = w
            o = st.cone_op(torch.arange(6, device="cuda"), torch.tensor(y, device="cuda"), 0, 1.0, check=False, outputs=("proj", "rnorm"))
            print((m, width, shift, pairs), "rows", st.max_rows, "bw", st.max_bw, "lds", lds, "waves", w, "status", o["status"].cpu().numpy().tolist(),
                  "iters", o["iters"].cpu().numpy().tolist(), "err %.2e" % np.abs(o["proj"].cpu().numpy() - po).max(), flush=True)
