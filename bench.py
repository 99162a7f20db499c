#!/usr/bin/env python
"""Hot-path benchmark: cone projections/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the fused hot path (dense scan -> projection -> CaVE+ target ->
cosine loss -> d loss/d pred) over one batch of synthetic TSP-20 cones, B = 1024 per GPU
(BASELINE.json configs[1]), inputs resident in HBM, through the C ABI (cave_hip_cone_dense).
N > 1: one process per GPU (torchrun), the batch shards by instance (weak scaling, no
data-path collective); the scalar loss is all-reduced over RCCL each step.
Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--tsp", type=int, default=20, help="TSP size n (d = n(n-1)/2)")
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU per step")
    ap.add_argument("--instances", type=int, default=1000, help="distinct instances in the dataset")
    ap.add_argument("--mode", default="inner", choices=["project", "exact", "inner"])
    ap.add_argument("--cpu-sample", type=int, default=256, help="instances timed on the host for cpu_baseline (0=skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip packed-path / train-step side measurements")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the side measurements at the instance sizes of BASELINE configs 3-5")
    return ap.parse_args()


def cpu_baseline(ctrs_np, costs_np, n_sample):
    """The CPU oracle (oracle/nnls_oracle.c, Lawson-Hanson, 1 thread) in the reference's loop shape."""
    from oracle import cave_oracle as O

    O.build()
    n = min(n_sample, len(ctrs_np))
    O.batch_project(-costs_np[:2], ctrs_np[:2])  # warm
    t0 = time.perf_counter()
    O.batch_project(-costs_np[:n], ctrs_np[:n])
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "projections/s", "cores": 1, "kind": "port",
           "sample": f"{n} TSP instances of the benchmark batch, serial loop, {dt:.1f} s"}
    try:  # informational: the reference's third-party solver in the reference's loop shape (src/cave.py:257,303-309)
        import numpy as np
        from scipy.optimize import nnls

        k = min(32, n)
        t0 = time.perf_counter()
        for i in range(k):
            A = ctrs_np[i][np.abs(ctrs_np[i]).sum(axis=1) > 1e-7]
            nnls(np.asfortranarray(A.T), -costs_np[i])
        out["scipy_nnls_projections_per_s_1core"] = k / (time.perf_counter() - t0)
    except Exception:  # noqa: BLE001
        pass
    return out


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    from cave_amd import _lib, synth
    from cave_amd.qpsolver import cone_op_dense

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world} (launch with torch.distributed.run)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.load()

    # ---- synthetic dataset: `instances` TSP-n cones, this rank's batch of ids
    ctrs_np, costs_np, _ = synth.tsp_batch(args.tsp, args.instances, seed=0)
    ids = (np.arange(args.batch) + rank * args.batch) % args.instances
    rng = np.random.default_rng(1234 + rank)
    pred_np = costs_np[ids] + rng.normal(0, 0.05, size=costs_np[ids].shape).astype(np.float32)
    ctrs = torch.tensor(ctrs_np[ids], device=dev)  # dense wire format, resident in HBM
    pred = torch.tensor(pred_np, device=dev)
    B, m_max, d = ctrs.shape
    mode = {"project": _lib.MODE_PROJECT, "exact": _lib.MODE_EXACT, "inner": _lib.MODE_INNER}[args.mode]
    outs = ("proj", "rnorm") if mode == _lib.MODE_PROJECT else ("loss", "grad")
    red = torch.zeros(2, device=dev)

    def step():
        o = cone_op_dense(ctrs, pred, mode, -1.0, 0.2, check=False, outputs=outs)
        if world > 1 and "loss" in o:  # global mean loss: all-reduce of [sum loss, count]
            red[0] = o["loss"].sum()
            red[1] = float(B)
            dist.all_reduce(red)
        return o

    cone_op_dense(ctrs, pred, mode, -1.0, 0.2, outputs=outs)  # one status-checked call: lets the wrapper settle its launch shape
    for _ in range(args.warmup):
        o = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        o = step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert bool((o["status"] == 0).all()), "solver reported failures"
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)

    # ---- dominant kernel duration (HIP events on the launch stream, kernel launches only)
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in kev:
        a.record()
        cone_op_dense(ctrs, pred, mode, -1.0, 0.2, check=False, outputs=outs)
        b.record()
    torch.cuda.synchronize()
    kern_ms = float(np.median([a.elapsed_time(b) for a, b in kev]))
    m_i = (np.abs(ctrs_np[ids]).sum(axis=2) > 0).sum(axis=1)
    # algorithmic bytes per launch, dense operator format: cone once, y once, outputs once
    out_bytes = (4 * d + 4) if mode == _lib.MODE_PROJECT else (4 * d + 4)
    alg_bytes = int((4 * m_i * d).sum() + B * (4 * d + out_bytes))
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    traffic = None  # HBM bytes per launch from the PMC passes (tools/diag/pmc_run.sh -> profiles/)
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
        if pm.get("workload") == f"tsp{args.tsp}_b{args.batch}_{args.mode}":
            traffic = pm.get("hbm_bytes_per_launch")
    except Exception:  # noqa: BLE001
        pass
    if rank == 0:
        res = {
            "metric": "cone projections/sec", "value": world * B * args.steps / dt, "unit": "projections/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"TSP-{args.tsp} DFJ tight cones, {args.instances} instances, batch {B}/GPU, "
                                   f"CaVE+ ({args.mode}) solver='hip', dense (B,m_max,d) wire format",
                       "batch_per_gpu": B, "d": d, "m_max": m_max, "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "cone_dense_kernel", "kernel_ms": kern_ms, "algorithmic_bytes": alg_bytes},
            "newton_iters_mean": float(o["iters"].float().mean()),
        }
        if not args.no_extras:
            res.update(extras(args, ctrs_np, costs_np, ids, pred, dev, mode, outs))
            if not args.no_other_configs and world == 1:
                res["other_configs"] = other_configs(dev)
        if args.cpu_sample > 0 and world == 1:
            res["cpu_baseline"] = cpu_baseline(ctrs_np[ids], pred_np, args.cpu_sample)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def extras(args, ctrs_np, costs_np, ids, pred, dev, mode, outs):
    """Side measurements (not `value`): device-resident packed store, and a full training step."""
    import numpy as np
    import torch

    from cave_amd.cave import EPO, innerConeAlignedCosine
    from cave_amd.dataset import ConeStore, PackedBatch

    out = {}
    store = ConeStore.from_dense(torch.tensor(ctrs_np))
    tid = torch.tensor(ids, device=dev)
    for _ in range(5):
        store.cone_op(tid, pred, mode, -1.0, 0.2, check=False, outputs=outs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 100
    for _ in range(K):
        o = store.cone_op(tid, pred, mode, -1.0, 0.2, check=False, outputs=outs)
    torch.cuda.synchronize()
    dtp = time.perf_counter() - t0
    assert bool((o["status"] == 0).all())
    out["packed_store"] = {"projections_per_s": len(ids) * K / dtp, "ms_per_step": 1e3 * dtp / K,
                           "store_bytes": store.nbytes(), "algorithmic_bytes": store.algorithmic_bytes(tid)}

    # training step in the shape of code_sample.py:23-59: linear predictor, CaVE+ loss, Adam(lr 1e-2)
    class _Model:
        modelSense = EPO.MINIMIZE

    p_feat = 10
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(len(ids), p_feat, generator=g).to(dev)
    reg = torch.nn.Linear(p_feat, pred.shape[1]).to(dev)
    opt = torch.optim.Adam(reg.parameters(), lr=1e-2)
    cave = innerConeAlignedCosine(_Model(), solver="hip", seed=0)
    batch = PackedBatch(store, tid)

    def train_step():
        loss = cave(reg(x), batch)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        train_step()
    torch.cuda.synchronize()
    out["train_step_ms"] = 1e3 * (time.perf_counter() - t0) / 50
    # same step, per-instance status examined one call later instead of right after the launch (no host sync per step)
    from cave_amd.cave import flush_checks

    cave_lazy = innerConeAlignedCosine(_Model(), solver="hip", seed=0, solver_kwargs={"check": "lazy"})

    def train_step_lazy():
        loss = cave_lazy(reg(x), batch)
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(5):
        train_step_lazy()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        train_step_lazy()
    flush_checks()
    torch.cuda.synchronize()
    out["train_step_lazy_check_ms"] = 1e3 * (time.perf_counter() - t0) / 50
    # the same step captured in a HIP graph (the C-ABI launch path does no allocation, attribute
    # change or host sync of its own when status checking is deferred)
    try:
        reg2 = torch.nn.Linear(p_feat, pred.shape[1]).to(dev)
        opt2 = torch.optim.Adam(reg2.parameters(), lr=1e-2, capturable=True)
        cave2 = innerConeAlignedCosine(_Model(), solver="hip", seed=0, solver_kwargs={"check": False})

        def step2():
            loss = cave2(reg2(x), batch)
            opt2.zero_grad(set_to_none=False)
            loss.backward()
            opt2.step()
            return loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step2()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            gl = step2()
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            graph.replay()
        torch.cuda.synchronize()
        out["train_step_graph_ms"] = 1e3 * (time.perf_counter() - t0) / 100
        out["train_step_graph_loss"] = float(gl.detach())
    except Exception as e:  # noqa: BLE001 - graph capture is an optional extra
        out["train_step_graph_error"] = repr(e)[:200]
    return out


def other_configs(dev):
    """Side measurements (not `value`): one GPU's share of BASELINE configs 3-5 on the packed store.
    A few unique synthetic cones per size (the dense form of TSP-100 is 102 MB per instance) are
    replicated to the per-GPU batch with independent predictions."""
    import torch

    from cave_amd import _lib, synth
    from cave_amd.dataset import ConeStore

    def run(name, gen, n_unique, B, mode, chunk):
        ctrs, costs, _ = gen(n_unique)
        c = torch.tensor(ctrs, device=dev)
        store = ConeStore.from_dense(c, chunk=chunk)
        del c
        ids = torch.arange(B, device=dev) % n_unique
        g = torch.Generator(device="cpu").manual_seed(1)
        pred = torch.tensor(costs, device=dev)[ids] + 0.05 * torch.randn(B, costs.shape[1], generator=g).to(dev)
        outs = ("loss", "grad")
        o = store.cone_op(ids, pred, mode, -1.0, 0.2, outputs=outs)
        torch.cuda.synchronize()
        K = 5
        t0 = time.perf_counter()
        for _ in range(K):
            store.cone_op(ids, pred, mode, -1.0, 0.2, check=False, outputs=outs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        return {"workload": name, "batch_per_gpu": B, "unique_cones": n_unique, "d": int(costs.shape[1]),
                "reduced_rows_max": store.max_rows, "path": "large (band LDL^T, global workspace)" if store.large else "fast (LDS)",
                "projections_per_s": B / dt, "ms_per_step": 1e3 * dt, "newton_iters_mean": float(o["iters"].float().mean()),
                "newton_iters_max": int(o["iters"].max())}

    out = []
    try:
        out.append(run("configs[2] TSP-50, CaVE Exact, 4096/8 GPUs", lambda n: synth.tsp_batch(50, n, seed=0), 32, 512,
                       _lib.MODE_EXACT, 32))
        out.append(run("configs[3] TSP-100, QP branch of CaVE Hybrid (inner_ratio 0.2), 2048/4 GPUs",
                       lambda n: synth.tsp_batch(100, n, seed=0), 8, 512, _lib.MODE_INNER, 4))
        out.append(run("configs[4] shortest path 30x30, CaVE+ (inner), 8192/8 GPUs",
                       lambda n: synth.sp_batch(30, 30, n, seed=0), 32, 1024, _lib.MODE_INNER, 32))
    except Exception as e:  # noqa: BLE001 - side measurement only
        out.append({"error": repr(e)[:300]})
    return out


if __name__ == "__main__":
    main()
