"""Diagnostic: save / compare bit patterns of the large-cone path's outputs on a few grid shapes
(a change that only reorganises the band elimination must reproduce them exactly)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import synth, _lib
if os.environ.get('CAVE_SO'): _lib.LIB_PATH = os.path.abspath(os.environ['CAVE_SO'])
from cave_amd.dataset import ConeStore
mode, path = sys.argv[1], sys.argv[2]
out = {}
for (h, w, n) in ((30, 30, 64), (12, 12, 64), (9, 9, 64), (20, 7, 32), (6, 33, 32)):
    c, y, _ = synth.sp_batch(h, w, n, seed=1)
    st = ConeStore.from_dense(torch.tensor(c, device="cuda"), chunk=64)
    assert st.large, (h, w)
    ids = torch.arange(n, device="cuda")
    o = st.cone_op(ids, torch.tensor(y, device="cuda"), 2, -1.0, outputs=("loss", "grad", "proj"))
    for k in ("loss", "grad", "proj", "iters"):
        out[f"{h}x{w}_{k}"] = o[k].cpu().numpy()
    print(h, w, "bw", st.max_bw, "iters", float(o["iters"].float().mean()), flush=True)
if mode == "save":
    np.savez(path, **out)
else:
    ref = np.load(path)
    bad = 0
    for k in out:
        same = np.array_equal(ref[k].view(np.uint32) if ref[k].dtype == np.float32 else ref[k], out[k].view(np.uint32) if out[k].dtype == np.float32 else out[k])
        if not same:
            bad += 1
            d = np.abs(ref[k].astype(np.float64) - out[k].astype(np.float64)).max()
            print("DIFF", k, "max abs", d)
    print("bitwise identical" if bad == 0 else f"{bad} arrays differ")
