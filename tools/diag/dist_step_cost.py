"""Diagnostic: cost of the per-step [sum loss, count] all-reduce (RCCL, world size 1) beside the pipelined step."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from cave_amd import _lib, synth, qpsolver
from cave_amd.qpsolver import prepare_dense, cone_op_prepared, stream_mark
dev = torch.device("cuda", 0); torch.cuda.set_device(0); _lib.load()
ctrs_np, costs_np, _ = synth.tsp_batch(20, 4096, seed=0)
rng = np.random.default_rng(1234)
batches = [(torch.tensor(ctrs_np[r*1024:(r+1)*1024], device=dev), torch.tensor(costs_np[r*1024:(r+1)*1024] + rng.normal(0, 0.05, size=(1024, 190)).astype(np.float32), device=dev)) for r in range(4)]
reds = [torch.zeros(2, device=dev) for _ in range(4)]
red_stream = torch.cuda.Stream(device=dev)
def run(variant, n=200):
    q = []
    def step(i):
        if not q:
            q.append(prepare_dense(batches[i % 4][0])); q.append(prepare_dense(batches[(i + 1) % 4][0]))
        prep = q.pop(0); mark = stream_mark(dev)
        o = cone_op_prepared(prep, batches[i % 4][1], _lib.MODE_INNER, -1.0, 0.2, check=False, outputs=("loss", "grad"))
        q.append(prepare_dense(batches[(i + 2) % 4][0], ready=mark))
        r = reds[i % 4]
        if variant == "main-blocking":
            r[0] = o["loss"].sum(); r[1] = 1024.0; dist.all_reduce(r)
        elif variant in ("side-sum", "side-allreduce", "side-async"):
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(red_stream):
                red_stream.wait_event(ev)
                r[0] = o["loss"].sum(); r[1] = 1024.0
                if variant == "side-allreduce": dist.all_reduce(r)
                if variant == "side-async": dist.all_reduce(r, async_op=True)
        elif variant == "side-fused":
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(red_stream):
                red_stream.wait_event(ev)
                torch.sum(o["loss"], dim=0, out=r[0:1].view(()) ) if False else r[0:1].copy_(o["loss"].sum(dim=0, keepdim=True))
                dist.all_reduce(r)
    for i in range(20): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(i)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{variant:16s}: {(time.perf_counter() - t0) / n * 1e6:7.1f} us/step, host enqueue {(t1 - t0) / n * 1e6:6.1f}", flush=True)
for sp in (25000, 50000, 100000, 200000):
    qpsolver.PIPE_SPACER_CYCLES = sp
    print("spacer", sp, end=" ")
    run("none")
qpsolver.PIPE_SPACER_CYCLES = int(os.environ.get("SPACER", 25000))
reds_c = [torch.tensor([0.0, 1024.0], device=dev) for _ in range(4)]
def run2(n=200):
    q = []
    def step(i):
        if not q:
            q.append(prepare_dense(batches[i % 4][0])); q.append(prepare_dense(batches[(i + 1) % 4][0]))
        prep = q.pop(0); mark = stream_mark(dev)
        o = cone_op_prepared(prep, batches[i % 4][1], _lib.MODE_INNER, -1.0, 0.2, check=False, outputs=("loss", "grad"))
        q.append(prepare_dense(batches[(i + 2) % 4][0], ready=mark))
        ev = torch.cuda.Event(); ev.record()
        r = reds_c[i % 4]
        with torch.cuda.stream(red_stream):
            red_stream.wait_event(ev)
            torch.sum(o["loss"], dim=0, keepdim=True, out=r[0:1])
            dist.all_reduce(r)
    for i in range(20): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(i)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"device-side sum + all-reduce on side stream: {(time.perf_counter() - t0) / n * 1e6:7.1f} us/step, host enqueue {(t1 - t0) / n * 1e6:6.1f}", flush=True)
run2()
dist.destroy_process_group()
