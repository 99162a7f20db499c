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
prep0 = prepare_dense(batches[0][0]); side = qpsolver._side_streams[dev]
stores = [prepare_dense(batches[r][0]).store for r in range(3)]
torch.cuda.synchronize()
E = lambda: torch.cuda.Event(enable_timing=True)
dummy = torch.cuda.Event(); dummy.record(main); torch.cuda.synchronize()
log = []
def pack(r, slot):
    c = batches[r][0]
    a, b = E(), E()
    with torch.cuda.stream(side):
        a.record(side)
        lib.cave_hip_pack_fill(_lib.ptr(c), 1024, c.shape[1], c.shape[2], 0, 0, 4, stores[slot].ref, 0, _lib.ptr(stores[slot].pack_status), qpsolver.C_void(side.cuda_stream))
        b.record(side)
    return a, b
def solve(r, slot):
    prep = PreparedCones(batches[r][0], stores[slot], dummy, stores[slot].gen)
    a, b = E(), E()
    a.record(main)
    o = cone_op_prepared(prep, batches[r][1], _lib.MODE_INNER, -1.0, 0.2, check=False, outputs=("loss", "grad"))
    b.record(main)
    return a, b
# manual pipeline: solve(i) on main, pack(i+1) on side; cross-stream waits on the end events
base = E(); base.record(main)
pk = pack(0, 0); prev_pack_end = pk[1]; ev = []
prev_solve_start = None
for i in range(12):
    main.wait_event(prev_pack_end)           # solve(i) needs pack(i)
    s = solve(i % 4, i % 3)
    if prev_solve_start is not None: pass
    side.wait_event(s[0])                    # pack(i+1) may start once solve(i) has STARTED (its slot is another one)
    pk = pack((i + 1) % 4, (i + 1) % 3)
    prev_pack_end = pk[1]
    ev.append((s, pk))
torch.cuda.synchronize()
for i, (s, pk) in enumerate(ev):
    print(f"step {i}: solve {base.elapsed_time(s[0])*1e3:8.1f} .. {base.elapsed_time(s[1])*1e3:8.1f}   pack(next) {base.elapsed_time(pk[0])*1e3:8.1f} .. {base.elapsed_time(pk[1])*1e3:8.1f}")
