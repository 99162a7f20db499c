import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth
from cave_amd.qpsolver import cone_op_dense
from cave_amd.dataset import ConeStore
_lib.load()
ctrs, costs, _ = synth.tsp_batch(20, 4096, seed=0)
call = torch.tensor(ctrs, device="cuda"); pall = torch.tensor(costs, device="cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a,b in evs]))*1e3
for B in (256, 1024, 2048, 4096):
    c = call[:B].contiguous(); p = pall[:B].contiguous()
    store = ConeStore.from_dense(c)
    ids = torch.arange(B, device="cuda")
    row = []
    for mi in (1, 100):
        td = timeit(lambda: cone_op_dense(c, p, 0, -1.0, 0.2, max_iter=mi, check=False))
        td1 = timeit(lambda: cone_op_dense(c, p, 0, -1.0, 0.2, max_iter=mi, waves=1, check=False))
        td2 = timeit(lambda: cone_op_dense(c, p, 0, -1.0, 0.2, max_iter=mi, waves=2, check=False))
        store.waves = 4
        tp = timeit(lambda: store.cone_op(ids, p, 0, -1.0, 0.2, max_iter=mi, check=False))
        store.waves = 1
        tp1 = timeit(lambda: store.cone_op(ids, p, 0, -1.0, 0.2, max_iter=mi, check=False))
        store.waves = 0
        store.waves = 2
        tp2 = timeit(lambda: store.cone_op(ids, p, 0, -1.0, 0.2, max_iter=mi, check=False))
        store.waves = 0
        row.append((mi, round(td,1), round(td2,1), round(td1,1), round(tp,1), round(tp2,1), round(tp1,1)))
    ta = timeit(lambda: cone_op_dense(c, None, 4, check=False, outputs=("target",)))
    print(f"B={B}: (max_iter, dense4, dense2, dense1, packed4, packed2, packed1 us) {row}  avg-only(scan+build+avg) {ta:.1f} us")
