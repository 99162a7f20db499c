import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth, qpsolver
from cave_amd.qpsolver import prepare_dense, cone_op_prepared, stream_mark, PreparedCones
dev = torch.device("cuda", 0); torch.cuda.set_device(0); lib = _lib.load()
ctrs_np, costs_np, _ = synth.tsp_batch(20, 4096, seed=0)
rng = np.random.default_rng(1234)
batches = [(torch.tensor(ctrs_np[r*1024:(r+1)*1024], device=dev), torch.tensor(costs_np[r*1024:(r+1)*1024] + rng.normal(0, 0.05, size=(1024, 190)).astype(np.float32), device=dev)) for r in range(4)]
main = torch.cuda.current_stream()
qpsolver.PIPE_SPACER_CYCLES = int(os.environ.get('SPACER', 0))
E = lambda: torch.cuda.Event(enable_timing=True)
state = {"prep": None}
log = []
def step(i, rec):
    c, p = batches[i % 4]
    prep = state["prep"] if state["prep"] is not None else prepare_dense(c)
    if os.environ.get("WAIT_FIRST") and isinstance(prep, PreparedCones): main.wait_event(prep.event)
    mark = stream_mark(dev)
    a, b = E(), E()
    if rec: a.record(main)
    o = cone_op_prepared(prep, p, _lib.MODE_INNER, -1.0, 0.2, check=False, outputs=("loss", "grad"))
    if rec: b.record(main)
    state["prep"] = prepare_dense(batches[(i + 1) % 4][0], ready=mark)
    pe = E()
    if rec:
        pe.record(qpsolver._side_streams[dev])
        log.append((a, b, pe))
for i in range(20): step(i, False)
base = E(); base.record(main)
for i in range(20, 32): step(i, True)
torch.cuda.synchronize()
print("period", (base.elapsed_time(log[-1][1]) - base.elapsed_time(log[1][1])) / (len(log) - 2) * 1e3, "us")
for a, b, pe in log[:4]:
    print(f"solve start(enqueue point) {base.elapsed_time(a)*1e3:8.1f}  solve end {base.elapsed_time(b)*1e3:8.1f}   pack(next) end {base.elapsed_time(pe)*1e3:8.1f}")
t = batches[0][0]
print("same ptr", qpsolver._as_device(t, dev).data_ptr() == t.data_ptr(), "pool", {k: len(v) for k, v in qpsolver._prep_pool.items()})
