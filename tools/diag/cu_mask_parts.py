"""Diagnostic: pack kernel alone / solve kernel alone on streams restricted to a subset of the compute units."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cave_amd import _lib, synth, qpsolver
from cave_amd.qpsolver import prepare_dense, cone_op_prepared, PreparedCones
if len(sys.argv) > 1: qpsolver.SPLIT_NNZ = int(sys.argv[1])
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[w * 32 + b]) for w in range(8)])
    s = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words) == 0
    return torch.cuda.ExternalStream(s.value)
dev = torch.device("cuda", 0); torch.cuda.set_device(0); _lib.load()
ctrs_np, costs_np, _ = synth.tsp_batch(20, 4096, seed=0)
rng = np.random.default_rng(1234)
batches = []
for r in range(4):
    ids = np.arange(1024) + r * 1024
    pred = costs_np[ids] + rng.normal(0, 0.05, size=costs_np[ids].shape).astype(np.float32)
    batches.append((torch.tensor(ctrs_np[ids], device=dev), torch.tensor(pred, device=dev)))
def timed(fn, stream, n=100):
    with torch.cuda.stream(stream):
        for i in range(10): fn(i)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for i, (a, b) in enumerate(ev):
            a.record(stream); fn(i); b.record(stream)
        torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]
print("SPLIT_NNZ", qpsolver.SPLIT_NNZ)
for name, sel in (("all 256", lambda i: True), ("first 192 bits", lambda i: i < 192), ("first 208 bits", lambda i: i < 208), ("first 224 bits", lambda i: i < 224),
                  ("26 of every 32", lambda i: i % 32 < 26), ("bits with i%8 < 6 (192)", lambda i: i % 8 < 6), ("last 64 bits", lambda i: i >= 192), ("last 48 bits", lambda i: i >= 208),
                  ("6 of every 32 (48)", lambda i: i % 32 >= 26), ("8 of every 32 (64)", lambda i: i % 32 >= 24), ("i%8 >= 6 (64)", lambda i: i % 8 >= 6)):
    bits = [bool(sel(i)) for i in range(256)]
    st = masked_stream(bits)
    qpsolver._side_streams[dev] = st
    preps = []
    def pack(i):
        p = prepare_dense(batches[i % 4][0])
        torch.cuda.current_stream().wait_event(p.event)
        return p
    tp = timed(pack, st)
    with torch.cuda.stream(st):
        prep = prepare_dense(batches[0][0]); torch.cuda.synchronize()
        ts = timed(lambda i: cone_op_prepared(prep, batches[0][1], _lib.MODE_INNER, -1.0, 0.2, check=False, outputs=("loss", "grad")), st)
    print(f"{name:28s} CUs {sum(bits):3d}: pack {tp:7.1f} us   solve {ts:7.1f} us", flush=True)
