import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth, qpsolver
from cave_amd.qpsolver import prepare_dense, cone_op_prepared, cone_op_dense, stream_mark, PreparedCones
dev = torch.device("cuda", 0); torch.cuda.set_device(0); _lib.load()
ctrs_np, costs_np, _ = synth.tsp_batch(20, 4096, seed=0)
batches = [(torch.tensor(ctrs_np[r*1024:(r+1)*1024], device=dev), torch.tensor(costs_np[r*1024:(r+1)*1024], device=dev)) for r in range(4)]
state = {"prep": None}
kinds = []
def step(i, ordered):
    c, p = batches[i % 4]
    prep = state["prep"] if state["prep"] is not None else prepare_dense(c)
    mark = stream_mark(dev)
    kinds.append((type(prep).__name__, prep.stale() if isinstance(prep, PreparedCones) else None))
    if isinstance(prep, PreparedCones): o = cone_op_prepared(prep, p, _lib.MODE_INNER, -1.0, 0.2, check=False, outputs=("loss", "grad"))
    else: o = cone_op_dense(c, p, _lib.MODE_INNER, -1.0, 0.2, check=False, outputs=("loss", "grad"))
    state["prep"] = prepare_dense(batches[(i + 1) % 4][0], ready=mark if ordered else None)
    return o
for ordered in (True,):
    for i in range(10): step(i, ordered)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200): step(i, ordered)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("ordered", ordered, (time.perf_counter() - t0) / 200 * 1e6, "us/step; host enqueue", (t1 - t0) / 200 * 1e6, "us/step", kinds[-3:])
    # host cost of the non-pipelined step
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200): cone_op_dense(batches[i % 4][0], batches[i % 4][1], _lib.MODE_INNER, -1.0, 0.2, check=False, outputs=("loss", "grad"))
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("back to back:", (time.perf_counter() - t0) / 200 * 1e6, "us/step; host enqueue", (t1 - t0) / 200 * 1e6)
