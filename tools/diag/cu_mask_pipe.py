"""Diagnostic: the dense operator as a two-stage pipeline on DISJOINT compute units -- the pack kernel of step i + 1 on
a stream whose CU mask holds `PACK_CUS` units, the solve kernel of step i on the others (hipExtStreamCreateWithCUMask).
    python tools/diag/cu_mask_pipe.py [pack CUs per group of 32] [layout: 0 = mask bit i = CU i, 1 = interleaved by 8]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth, qpsolver
from cave_amd.qpsolver import prepare_dense, cone_op_prepared, cone_op_dense, stream_mark, PreparedCones
per32 = int(sys.argv[1]) if len(sys.argv) > 1 else 6
layout = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if len(sys.argv) > 3: qpsolver.SPLIT_NNZ = int(sys.argv[3])
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[w * 32 + b]) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
_lib.load()
pack_bits = [False] * 256
for i in range(256):
    if layout == 0: sel = (i % 32) < per32          # first per32 of every 32 consecutive mask bits
    else: sel = (i // 8) < per32                    # mask bits interleaved over 8 XCDs: CU index within the XCD = i // 8
    pack_bits[i] = sel
solve_bits = [not b for b in pack_bits]
print("pack CUs", sum(pack_bits), "solve CUs", sum(solve_bits), "layout", layout, "SPLIT_NNZ", qpsolver.SPLIT_NNZ, flush=True)
ctrs_np, costs_np, _ = synth.tsp_batch(20, 4096, seed=0)
rng = np.random.default_rng(1234)
batches = []
for r in range(4):
    ids = np.arange(1024) + r * 1024
    pred = costs_np[ids] + rng.normal(0, 0.05, size=costs_np[ids].shape).astype(np.float32)
    batches.append((torch.tensor(ctrs_np[ids], device=dev), torch.tensor(pred, device=dev)))
mode = _lib.MODE_INNER
outs = ("loss", "grad")
def run(steps, pipelined, solve_stream, pack_stream):
    if pack_stream is not None: qpsolver._side_streams[dev] = pack_stream
    state = {"prep": None}
    def step(i):
        c, p = batches[i % 4]
        if pipelined:
            prep = state["prep"] if state["prep"] is not None else prepare_dense(c)
            mark = stream_mark(dev)
            if isinstance(prep, PreparedCones): o = cone_op_prepared(prep, p, mode, -1.0, 0.2, check=False, outputs=outs)
            else: o = cone_op_dense(c, p, mode, -1.0, 0.2, check=False, outputs=outs)
            state["prep"] = prepare_dense(batches[(i + 1) % 4][0], ready=mark)
        else:
            o = cone_op_dense(c, p, mode, -1.0, 0.2, check=False, outputs=outs)
        return o
    ctx = torch.cuda.stream(solve_stream) if solve_stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for i in range(20): o = step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps): o = step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    return dt, float(o["loss"].double().sum())
dt0, l0 = run(200, False, None, None)
print(f"back to back, no masks: {dt0*1e6:.1f} us/step (loss sum {l0:.6f})", flush=True)
ps, ss = masked_stream(pack_bits), masked_stream(solve_bits)
dt1, l1 = run(200, True, ss, ps)
print(f"pipelined on disjoint CUs: {dt1*1e6:.1f} us/step (loss sum {l1:.6f})", flush=True)
dt2, l2 = run(200, False, ss, None)
print(f"back to back on the solve CUs only: {dt2*1e6:.1f} us/step", flush=True)
