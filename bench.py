#!/usr/bin/env python
"""Hot-path benchmark: cone projections/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (dense scan -> cone build -> projection -> CaVE+ target ->
cosine loss -> d loss/d pred) over one batch of synthetic TSP-20 cones, B = 1024 per GPU
(BASELINE.json configs[1]), inputs resident in HBM, through the C ABI: ONE launch of cave_hip_cone_step
per step -- the Newton solve + loss + gradient of batch i and, in the same grid, the scan + cone build of
batch i+1 (whose cones a DataLoader has collated before the predictor has its prediction).
Successive steps rotate over `--rotate` different batches (default 4 x 183 MB > the 256 MB
Infinity Cache), so the cones of a step are read from HBM, not from a cache warmed by the step before.

N > 1: one process per GPU.  Either the driver starts the ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: RANK is set), or
`python bench.py --gpus N` starts them itself: the parent, which never touches the GPU, launches
`torch.distributed.run` as a child process, relays rank 0's JSON line and exits with the child's code.
The batch shards by instance (weak scaling, no data-path collective); the scalar loss is
all-reduced over RCCL each step.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PMC_SUMMARY = os.path.join("profiles", "r04_pmc_summary.json")
LARGE_PMC = {"tsp100": os.path.join("profiles", "r04_large_tsp100_pmc_summary.json"),
             "sp30": os.path.join("profiles", "r04_large_sp30_pmc_summary.json"),
             "tsp50": os.path.join("profiles", "r04_tsp50_pmc_summary.json")}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--tsp", type=int, default=20, help="TSP size n (d = n(n-1)/2)")
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU per step")
    ap.add_argument("--instances", type=int, default=4096,
                    help="distinct instances in the dataset (>= rotate * batch: every slot of every rotating batch is a "
                         "different cone; BASELINE configs[1] names 1 000, which would repeat cones inside a batch of 1024)")
    ap.add_argument("--rotate", type=int, default=4, help="distinct batches the timed steps cycle through")
    ap.add_argument("--mode", default="inner", choices=["project", "exact", "inner"])
    ap.add_argument("--cpu-sample", type=int, default=256, help="instances timed on the host for cpu_baseline (0=skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip packed-path / train-step side measurements")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the side measurements at the instance sizes of BASELINE configs 3-5")
    ap.add_argument("--no-large-cpu", action="store_true",
                    help="skip the one-instance CPU timing of the 30x30 grid in other_configs (about a minute of host time)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="time the back-to-back form as `value`: the general dense operator, pack kernel then solve kernel "
                         "per step (what a plain `module(pred, bctr)` call launches).  Default: the fused step -- one "
                         "launch per step holds the solve of batch i and the pack of batch i+1; every timed step still "
                         "does one pack and one solve.  The other form is timed too and reported beside it")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even at --gpus 1 (world size 1) and run the per-step "
                         "[sum loss, count] all-reduce and the sharded-store leg: the code path of an N-GPU run")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / JSON-relay check without a GPU (gloo, no kernels; CPU tests)")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------- self-launch

def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv) -> int:
    """Parent of a `python bench.py --gpus N` call (N > 1, RANK unset).  Nothing here touches HIP
    (no torch.cuda call, no dlopen of the extension): the ranks are children of a child process."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    out, err = proc.communicate()
    line = None
    for ln in out.splitlines():
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
    if proc.returncode != 0 or line is None:
        sys.stderr.write(err[-4000:])
        sys.stderr.write(f"\nbench.py: the {args.gpus}-rank run failed (exit code {proc.returncode}, "
                         f"JSON line {'found' if line else 'missing'})\n")
        return proc.returncode or 1
    print(line)
    return 0


def dry_run(args) -> int:
    """What a rank does without a GPU: rendezvous, barrier, max-over-ranks timing, rank 0 prints the line."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    red = torch.zeros(2)
    for _ in range(args.steps):
        red[0], red[1] = float(rank), float(args.batch)
        if world > 1:
            dist.all_reduce(red)
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "cone projections/sec", "value": None, "unit": "projections/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "instances_seen": float(red[1])}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# --------------------------------------------------------------------------- CPU baseline

def _oracle_chunk(job):
    from oracle import cave_oracle as O

    y, A = job
    O.batch_project(y, A)
    return len(y)


def _host_cores() -> int:
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes show
    every core of the host in the mask but grant a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:  # noqa: BLE001
            continue
    return n


def cpu_baseline(ctrs_np, signed_np, n_sample, label):
    """The CPU restatement of the reference's nnls path (oracle/nnls_oracle.c, Lawson-Hanson, fp64) in the
    reference's loop shape: serial (`processes=1`, src/cave.py:257) and over all host cores (the
    `processes=0` pool mode, src/cave.py:259; here a thread pool -- the C call releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import cave_oracle as O

    O.build()
    n = min(n_sample, len(ctrs_np))
    O.batch_project(signed_np[:1], ctrs_np[:1])  # warm
    t0 = time.perf_counter()
    O.batch_project(signed_np[:n], ctrs_np[:n])
    dt = time.perf_counter() - t0
    nproc = _host_cores()
    out = {"value": n / dt, "unit": "projections/s", "cores": 1, "kind": "port",
           "sample": f"{n} {label} instances of the benchmark batch, serial loop, {dt:.1f} s"}
    if nproc > 1:
        per = max(1, min(8, n // nproc))
        n_all = min(len(ctrs_np), max(n, per * nproc * 4))
        jobs = [(signed_np[i:i + per], ctrs_np[i:i + per]) for i in range(0, n_all, per)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nproc) as ex:
            done = sum(ex.map(_oracle_chunk, jobs))
        dta = time.perf_counter() - t0
        out["all_cores"] = {"value": done / dta, "cores": nproc,
                            "sample": f"{done} instances over {nproc} host threads, {dta:.1f} s"}
    # the reference's own third-party solvers, where the box has them
    try:
        import numpy as np
        from scipy.optimize import nnls

        k = min(32, n)
        t0 = time.perf_counter()
        for i in range(k):
            A = ctrs_np[i][np.abs(ctrs_np[i]).sum(axis=1) > 1e-7]
            nnls(np.asfortranarray(A.T), signed_np[i])
        out["scipy_nnls_projections_per_s_1core"] = k / (time.perf_counter() - t0)
    except Exception as e:  # noqa: BLE001
        out["scipy_nnls"] = f"unavailable ({type(e).__name__})"
    out["clarabel"] = clarabel_baseline(ctrs_np, signed_np, min(16, n))
    return out


def clarabel_baseline(ctrs_np, signed_np, k):
    """`_project_clarabel` with max_iter=3 (src/cave.py:267-295), the comparator the north star names.
    cvxpy + clarabel are not in this image; when they are missing the line says so and every ratio in
    this file is against the nnls path."""
    try:
        import clarabel  # noqa: F401
        import cvxpy as cp
    except Exception as e:  # noqa: BLE001
        return {"status": "unavailable", "reason": f"{type(e).__name__}: {e}"[:120],
                "note": "north-star ratio '>=100x over solver=clarabel' is reported against the nnls CPU path instead"}
    import numpy as np

    t0 = time.perf_counter()
    for i in range(k):
        A = ctrs_np[i][np.abs(ctrs_np[i]).sum(axis=1) > 1e-7]
        lam = cp.Variable(A.shape[0], nonneg=True)
        prob = cp.Problem(cp.Minimize(cp.sum_squares(A.T @ lam - signed_np[i])))
        prob.solve(solver=cp.CLARABEL, max_iter=3)
    return {"status": "timed", "projections_per_s_1core": k / (time.perf_counter() - t0), "max_iter": 3}


# --------------------------------------------------------------------------- main measurement

def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    if args.dry_run:
        sys.exit(dry_run(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    from cave_amd import _lib, synth
    from cave_amd.qpsolver import cone_op_dense

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}")
    use_dist = world > 1 or args.force_dist
    if use_dist:  # before any other GPU call of this process
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.load()

    # ---- synthetic dataset: `instances` TSP-n cones; this rank's R rotating batches of ids
    ctrs_np, costs_np, _ = synth.tsp_batch(args.tsp, args.instances, seed=0)
    R = max(1, args.rotate)
    rng = np.random.default_rng(1234 + rank)
    batches = []
    for r in range(R):
        # batch r of this rank: `batch` consecutive instances starting at r * batch (+ a per-rank offset); with
        # instances >= R * batch the R batches share no cone
        ids = (np.arange(args.batch) + r * args.batch + rank * (args.batch // 2 + 13)) % args.instances
        pred_np = costs_np[ids] + rng.normal(0, 0.05, size=costs_np[ids].shape).astype(np.float32)
        batches.append((ids, pred_np, torch.tensor(ctrs_np[ids], device=dev), torch.tensor(pred_np, device=dev)))
    ids, pred_np, ctrs, pred = batches[0]
    B, m_max, d = ctrs.shape
    mode = {"project": _lib.MODE_PROJECT, "exact": _lib.MODE_EXACT, "inner": _lib.MODE_INNER}[args.mode]
    outs = ("proj", "rnorm") if mode == _lib.MODE_PROJECT else ("loss", "grad")
    reds = [torch.zeros(2, device=dev) for _ in range(4)]
    count = torch.full((1,), float(args.batch), device=dev)
    red_stream = torch.cuda.Stream(device=dev) if use_dist else None

    from cave_amd.qpsolver import PreparedCones, cone_op_prepared, prepare_dense
    args.pipeline = not args.no_pipeline

    # Fused step (default; --no-pipeline: the general dense operator, two kernels back to back): the pack stage of
    # batch i+1 (stream its dense cones, build its reduced cones: depends on the cones only, which a DataLoader has
    # collated ahead of the predictor) rides in the launch of batch i's solve -- one grid, solve blocks first in the
    # dispatch order, one stream, no events.  Every timed step does exactly one pack and one solve (the pack of the last
    # timed step is for a batch nobody solves: extra work inside the timed region, none skipped).
    state = {"prep": None}

    def step(i, fused=None):
        _, _, c, p = batches[i % R]
        fused = args.pipeline if fused is None else fused
        if fused:
            prep = state["prep"]
            if prep is None:  # first step: nothing has packed this batch yet
                prep = prepare_dense(c)
            if isinstance(prep, PreparedCones):
                prep.then(batches[(i + 1) % R][2])
                o = cone_op_prepared(prep, p, mode, -1.0, 0.2, check=False, outputs=outs)
                state["prep"] = prep.next
            else:
                o = cone_op_dense(c, p, mode, -1.0, 0.2, check=False, outputs=outs)
        else:
            o = cone_op_dense(c, p, mode, -1.0, 0.2, check=False, outputs=outs)
        if use_dist and "loss" in o:  # global mean loss: all-reduce of [sum loss, count]
            # on a stream of its own behind this step's solve: the next solve depends on neither the sum nor the
            # collective (RCCL orders the all-reduce after the work of the stream that is current at the call)
            ev = torch.cuda.Event()
            ev.record()
            r = reds[i % len(reds)]
            with torch.cuda.stream(red_stream):
                red_stream.wait_event(ev)
                o["loss"].record_stream(red_stream)
                torch.sum(o["loss"], dim=0, keepdim=True, out=r[0:1])  # (device side: `r[0] = x.sum(); r[1] = float(B)`
                r[1:2].copy_(count)                                    #  copies scalars from pageable host memory, which
                dist.all_reduce(r)                                     #  waits for the stream: 240 us per step)
        return o

    # one status-checked call per rotating batch: lets the wrapper settle a launch shape that fits every cone of the run
    # (and, through a checked prepared call, whether the cones qualify for the fused step)
    for _, _, c_, p_ in batches:
        cone_op_dense(c_, p_, mode, -1.0, 0.2, outputs=outs)
        pr_ = prepare_dense(c_)
        if isinstance(pr_, PreparedCones):
            cone_op_prepared(pr_, p_, mode, -1.0, 0.2, outputs=outs)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    status_all = []
    for i in range(args.steps):
        o = step(i)
        if i >= args.steps - R:
            status_all.append(o["status"])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(bool((s == 0).all()) for s in status_all), "solver reported failures"
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)

    fused = bool(args.pipeline and isinstance(state["prep"], PreparedCones))

    # ---- the OTHER form, timed the same way (not `value`): ADVICE r3 -- always report both
    def timed_form(use_fused, n):
        state["prep"] = None
        for i in range(min(10, n)):
            step(i, use_fused)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            step(i, use_fused)
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t) / n
    other_us = timed_form(not args.pipeline, args.steps) if world == 1 or not use_dist else None

    # ---- dominant kernel duration: HIP events on the launch stream around single launches, rotating batches
    # (fused form: the one cone_step_kernel launch of a step -- solve of batch i + pack of batch i+1; back-to-back form:
    #  the two kernels of the general dense operator)
    # groups of KG launches between two events, the stream kept busy by a launch ahead of the first event: the figure is
    # the kernel's duration + the 2-3 us between two launches of a busy stream, not the ~6 us an idle stream needs to
    # start one (rocprofv3's per-kernel average of the same command: profiles/r04_kernel_stats.csv, fused launches only
    # in profiles/r04_kernel_stats_by_grid.txt)
    KG = 4
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2 * R)]
    state["prep"] = None
    step(0)
    torch.cuda.synchronize()
    n_k = 1
    for a, b in kev:
        step(n_k)
        n_k += 1
        a.record()
        for _ in range(KG):
            step(n_k)
            n_k += 1
        b.record()
    torch.cuda.synchronize()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in kev])) / KG
    from cave_amd import qpsolver

    split = qpsolver._split_ok.get((m_max, d)) is True
    if fused:
        kernels = ("cone_step_kernel<BlockCtx<2,true>> (ONE launch per step: one-wave Newton solves + fused loss/grad of batch i, "
                   "then two-wave scan + cone build workgroups of batch i+1)")
    elif split:
        kernels = ("cone_pack_kernel<4 waves> (slot mode: scan + cone build -> transient store) + "
                   "cone_packed_kernel<1 wave> (lite Newton solver + fused loss/grad)")
    else:
        kernels = "cone_dense_kernel (fused)"
    # algorithmic bytes per launch, dense operator format: cone once, y once, outputs once
    alg = []
    for bids, _, _, _ in batches:
        m_i = (np.abs(ctrs_np[bids]).sum(axis=2) > 0).sum(axis=1)
        alg.append(int((4 * m_i * d).sum() + B * (4 * d + 4 * d + 4)))
    alg_bytes = int(np.mean(alg))
    step_ms = 1e3 * dt / args.steps
    # roofline of the dominant kernel: algorithmic bytes per launch / its average launch duration, HIP events on the
    # launch stream around single launches (fused form: the step IS that one launch; the figure of the timed region,
    # launches back to back, is reported beside it)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    traffic, traffic_src = None, None  # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/diag/pmc_run.sh)
    try:
        pm = json.load(open(os.path.join(ROOT, PMC_SUMMARY)))
        if pm.get("workload") == f"tsp{args.tsp}_b{args.batch}_{args.mode}" and bool(pm.get("fused")) == fused:
            traffic, traffic_src = pm.get("hbm_bytes_per_launch"), PMC_SUMMARY
    except Exception:  # noqa: BLE001
        pass
    if rank == 0:
        res = {
            "metric": "cone projections/sec", "value": world * B * args.steps / dt, "unit": "projections/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"TSP-{args.tsp} DFJ tight cones, {args.instances} instances, batch {B}/GPU, "
                                   f"CaVE+ ({args.mode}) solver='hip', dense (B,m_max,d) wire format, "
                                   f"{R} rotating batches ({R * ctrs.numel() * 4 / 1e6:.0f} MB of cones per GPU)",
                       "batch_per_gpu": B, "d": d, "m_max": m_max, "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernels, "kernel_ms": kern_ms, "algorithmic_bytes": alg_bytes,
                         "frac_timed_region": alg_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel_ms_note": "kernel_ms: average of HIP-event times around single launches of the step "
                                           "(rotating batches); achieved / frac = algorithmic bytes / kernel_ms; "
                                           "frac_timed_region = algorithmic bytes / ms_per_step (launches back to back, "
                                           "host time included); per-kernel durations: profiles/r04_kernel_stats.csv",
                         "memory_level": f"HBM (the {R} rotating batches exceed the 256 MB Infinity Cache)" if
                         R * ctrs.numel() * 4 > 300e6 else "may be served by the Infinity Cache (working set < 256 MB)"},
            "newton_iters_mean": float(o["iters"].float().mean()), "newton_iters_max": int(o["iters"].max()),
            "pipeline": {"form": "fused step (one launch: solve of batch i + pack of batch i+1)" if fused else
                         "back to back (general dense operator: pack kernel, then solve kernel)",
                         "fused_us_per_step": 1e3 * step_ms if fused else other_us,
                         "back_to_back_us_per_step": other_us if fused else 1e3 * step_ms,
                         "streams": 1, "events": 0,
                         "note": "both forms are timed over the same number of steps; `value` is the form named in `form` "
                                 "(--no-pipeline selects the other).  The fused form needs the NEXT batch's cones at the "
                                 "loss call: cave_amd.dataset.prefetch(loader) or loss_fn.prepare(bctr, next_bctr)"},
        }
        if use_dist:
            res["process_group"] = {"backend": "nccl (RCCL)", "world_size": world,
                                    "per_step_collective": "all_reduce([sum loss, count]), 8 bytes"}
        # side measurements only in the single-process run: under N ranks they would keep rank 0 busy for tens of
        # seconds while the others wait in a collective
        if not args.no_extras and world == 1:
            res.update(extras(args, ctrs_np, costs_np, ids, pred, dev, mode, outs))
            if not args.no_other_configs:
                res["other_configs"] = other_configs(dev, not args.no_large_cpu)
        if args.cpu_sample > 0 and world == 1:
            res["cpu_baseline"] = cpu_baseline(ctrs_np[ids], -pred_np, args.cpu_sample, f"TSP-{args.tsp}")
            res["gpu_over_cpu_1core_nnls"] = res["value"] / res["cpu_baseline"]["value"]
    # every rank reaches the sharded-store leg (collectives inside) right after the timed loop; rank 0 prints last
    if use_dist:
        shard = sharded_store_leg(args, ctrs_np, costs_np, dev, mode, outs, rank, world)
        if rank == 0:
            res["sharded_packed_store"] = shard
    if rank == 0:
        print(json.dumps(res))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def sharded_store_leg(args, ctrs_np, costs_np, dev, mode, outs, rank, world):
    """Distributed side measurement (reported inside the JSON line): the dataset's ragged cones are dealt to ranks
    balanced by their non-zeros (ConeStore.from_ragged_shard) and every rank projects its whole shard."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from cave_amd.dataset import ConeStore

    ragged = [torch.from_numpy(c[np.abs(c).sum(axis=1) > 0]) for c in ctrs_np]
    store = ConeStore.from_ragged_shard(ragged, rank, world)
    ids = torch.arange(store.n, device=dev)
    pred = torch.tensor(costs_np[store.global_ids.numpy()], device=dev)
    for _ in range(5):
        store.cone_op(ids, pred, mode, -1.0, 0.2, check=False, outputs=outs)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(50):
        store.cone_op(ids, pred, mode, -1.0, 0.2, check=False, outputs=outs)
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0, float(store.n), float(store.nbytes())], device=dev, dtype=torch.float64)
    tm = t.clone()
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {"ranks": world, "instances": int(t[1]), "projections_per_s": float(t[1]) * 50 / float(tm[0]),
            "max_shard_bytes": int(tm[2]), "balance": "sum of non-zeros (LPT)"}


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
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in kev:
        a.record()
        store.cone_op(tid, pred, mode, -1.0, 0.2, check=False, outputs=outs)
        b.record()
    torch.cuda.synchronize()
    pk_ms = float(np.mean([a.elapsed_time(b) for a, b in kev]))
    pk_bytes = store.algorithmic_bytes(tid)
    out["packed_store"] = {"projections_per_s": len(ids) * K / dtp, "ms_per_step": 1e3 * dtp / K,
                           "store_bytes": store.nbytes(), "algorithmic_bytes": pk_bytes, "kernel_ms": pk_ms,
                           "roofline": {"bound": "hbm", "achieved": pk_bytes / (pk_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": pk_bytes / (pk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "kernel": "cone_packed_kernel"}}

    # training step in the shape of code_sample.py:23-59: linear predictor, CaVE+ loss, Adam(lr 1e-2).
    # Two implementations of the same optimizer are timed: torch.optim.Adam(fused=True) -- one kernel per step -- under the
    # plain keys, and the default (foreach: a dozen small kernels, ~0.1 ms of host time per step) under *_foreach_adam_*.
    class _Model:
        modelSense = EPO.MINIMIZE

    p_feat = 10
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(len(ids), p_feat, generator=g).to(dev)
    batch = PackedBatch(store, tid)
    from cave_amd.cave import flush_checks

    def time_train(kw, fused_adam, n=60):
        reg = torch.nn.Linear(p_feat, pred.shape[1]).to(dev)
        opt = torch.optim.Adam(reg.parameters(), lr=1e-2, fused=fused_adam)
        cave = innerConeAlignedCosine(_Model(), solver="hip", seed=0, solver_kwargs=kw)

        def train_step():
            loss = cave(reg(x), batch)
            opt.zero_grad()
            loss.backward()
            opt.step()

        for _ in range(10):
            train_step()
        torch.cuda.synchronize()
        # host-bound (the kernel is a quarter of the step): the fastest of five runs of n steps -- single runs on one box
        # spread by +-30 % with whatever else the host is doing; the minimum is what the step costs when nothing interferes
        runs = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(n):
                train_step()
            flush_checks()
            torch.cuda.synchronize()
            runs.append(1e3 * (time.perf_counter() - t0) / n)
        return min(runs)

    out["train_step_ms"] = time_train(None, True)
    out["train_step_foreach_adam_ms"] = time_train(None, False)
    # same step, per-instance status examined by a later call, once it has arrived (no host sync per step)
    out["train_step_lazy_check_ms"] = time_train({"check": "lazy"}, True)
    out["train_step_lazy_check_foreach_adam_ms"] = time_train({"check": "lazy"}, False)
    out["train_step_timing"] = "host-bound: each train_step_*ms is the fastest of five runs of 60 steps"
    out["train_step_optimizer"] = ("torch.optim.Adam(lr=1e-2, fused=True); *_foreach_adam_*: torch.optim.Adam(lr=1e-2), the "
                                   "default multi-tensor implementation (rounds 1-3 timed that one)")
    reg = torch.nn.Linear(p_feat, pred.shape[1]).to(dev)
    opt = torch.optim.Adam(reg.parameters(), lr=1e-2, fused=True)
    cave_lazy = innerConeAlignedCosine(_Model(), solver="hip", seed=0, solver_kwargs={"check": "lazy"})
    # the same steps with the store's warm start (multipliers of the previous solve of each instance): cones are
    # static and the predictor moves a little per Adam step
    wstore = ConeStore.from_dense(torch.tensor(ctrs_np))
    wstore.enable_warm_start()
    wbatch = PackedBatch(wstore, tid)

    def train_step_warm():
        loss = cave_lazy(reg(x), wbatch)
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(5):
        train_step_warm()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    its = []
    for _ in range(50):
        train_step_warm()
        its.append(wstore.last_iters)
    flush_checks()
    torch.cuda.synchronize()
    out["train_step_warm_start_lazy_check_ms"] = 1e3 * (time.perf_counter() - t0) / 50
    out["train_step_warm_start_newton_iters_mean"] = float(torch.stack(its).float().mean())
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in kev:
        a.record()
        wstore.cone_op(tid, reg(x).detach(), mode, -1.0, 0.2, check=False, outputs=outs)
        b.record()
    torch.cuda.synchronize()
    out["packed_store_warm_start_kernel_ms"] = float(np.mean([a.elapsed_time(b) for a, b in kev]))
    # the same step captured in a HIP graph (the C-ABI launch path does no allocation, attribute
    # change or host sync of its own when status checking is deferred)
    def time_graph(fused_adam):
        reg2 = torch.nn.Linear(p_feat, pred.shape[1]).to(dev)
        opt2 = torch.optim.Adam(reg2.parameters(), lr=1e-2, capturable=True, fused=fused_adam)
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
        return 1e3 * (time.perf_counter() - t0) / 100, float(gl.detach())

    for key, fused_adam in (("train_step_graph", True), ("train_step_graph_foreach_adam", False)):
        try:
            out[key + "_ms"], out[key + "_loss"] = time_graph(fused_adam)
        except Exception as e:  # noqa: BLE001 - graph capture is an optional extra
            out[key + "_error"] = repr(e)[:200]
    return out


def other_configs(dev, large_cpu=True):
    """Side measurements (not `value`): one GPU's share of BASELINE configs 3-5 on the packed store, every
    batch slot a DISTINCT synthetic cone (generated in coordinate form, densified on the GPU a chunk at a
    time and packed), each with its own roofline and a sub-sampled CPU figure."""
    import numpy as np
    import torch

    from cave_amd import _lib, synth
    from cave_amd.dataset import ConeStore

    def run(name, kind, size, B, mode, chunk, cpu_n, cpu_note, tag=None, cpu_recorded=None, hybrid_ratio=None):
        items, costs, _ = synth.coo_batch(kind, size, B, seed=0)
        d = int(costs.shape[1])
        m_max = max(it[3] for it in items)
        store = ConeStore.from_chunks_lazy(lambda i: synth.densify_on(items[i:i + chunk], d, dev, m_max),
                                           list(range(0, B, chunk)))
        ids = torch.arange(B, device=dev)
        g = torch.Generator(device="cpu").manual_seed(1)
        pred_np = costs + 0.05 * torch.randn(B, d, generator=g).numpy()
        pred = torch.tensor(pred_np, device=dev)
        outs = ("loss", "grad")
        o = store.cone_op(ids, pred, mode, -1.0, 0.2, outputs=outs)
        torch.cuda.synchronize()
        K = 5
        kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        t0 = time.perf_counter()
        for a, b in kev:
            a.record()
            store.cone_op(ids, pred, mode, -1.0, 0.2, check=False, outputs=outs)
            b.record()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        k_ms = float(np.mean([a.elapsed_time(b) for a, b in kev]))
        alg = store.algorithmic_bytes(ids)
        res = {"workload": name, "batch_per_gpu": B, "distinct_cones": B, "d": d,
               "reduced_rows_max": store.max_rows,
               "path": "large (band LDL^T, global workspace)" if store.large else "fast (LDS)",
               "projections_per_s": B / dt, "ms_per_step": 1e3 * dt, "newton_iters_mean": float(o["iters"].float().mean()),
               "newton_iters_max": int(o["iters"].max()),
               "roofline": {"bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                            "kernel": "cone_packed_large_kernel" if store.large else "cone_packed_kernel",
                            "kernel_ms": k_ms, "algorithmic_bytes": alg, "format": "packed store"}}
        if tag in LARGE_PMC:  # HBM bytes per launch of this very workload, from separate rocprofv3 --pmc passes
            try:
                pm = json.load(open(os.path.join(ROOT, LARGE_PMC[tag])))
                res["roofline"]["traffic"] = pm.get("hbm_bytes_per_launch")
                res["roofline"]["traffic_source"] = LARGE_PMC[tag] + " (tools/diag/pmc_run_large.sh, same batch)"
                res["roofline"]["traffic_note"] = pm.get("fetch_size_note")
                res["roofline"]["traffic_over_algorithmic"] = pm.get("hbm_bytes_per_launch") / alg
            except Exception:  # noqa: BLE001
                pass
        if hybrid_ratio is not None:
            # CaVE Hybrid (src/cave.py:197-204): one RNG draw per forward decides the branch of the whole batch -- the QP
            # branch timed above with probability solve_ratio, else the heuristic branch (no projection: one pass over
            # y and the stored average normal); the expected step is their mix
            kh = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
            store.cone_op(ids, pred, _lib.MODE_HEURISTIC, -1.0, 0.2, outputs=outs)
            for a, b in kh:
                a.record()
                store.cone_op(ids, pred, _lib.MODE_HEURISTIC, -1.0, 0.2, check=False, outputs=outs)
                b.record()
            torch.cuda.synchronize()
            h_ms = float(np.mean([a.elapsed_time(b) for a, b in kh]))
            res["hybrid"] = {"solve_ratio": hybrid_ratio, "qp_branch_ms": k_ms, "heuristic_branch_ms": h_ms,
                             "hybrid_expected_ms_per_step": hybrid_ratio * k_ms + (1.0 - hybrid_ratio) * h_ms,
                             "hybrid_expected_projections_per_s": B / (1e-3 * (hybrid_ratio * k_ms + (1.0 - hybrid_ratio) * h_ms)),
                             "note": "kernel times (HIP events); the branch is drawn once per forward for the whole batch "
                                     "(src/cave.py:201), so a step costs one or the other"}
        if store.large and B > 256:
            # the slowest instance beside the mean: 256 instances = one workgroup per compute unit, so the launch
            # time is the latency of its slowest instance; mean = whole-batch time x resident workgroups / batch
            sub = ids[:256]
            kv = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
            for a, b in kv:
                a.record()
                store.cone_op(sub, pred[:256], mode, -1.0, 0.2, check=False, outputs=outs)
                b.record()
            torch.cuda.synchronize()
            res["slowest_instance_ms_alone_on_its_cu"] = float(np.min([a.elapsed_time(b) for a, b in kv[1:]]))
            res["newton_iters_max_of_those_256"] = int(o["iters"][:256].max())
        if store.large:
            # the training situation (cones static per instance, predictions drifting from step to step): per-instance
            # multiplier cache on, three steps of drift to fill it, then timed steps that keep drifting
            store.enable_warm_start()
            gw = torch.Generator(device="cpu").manual_seed(2)
            drift = [torch.tensor(0.01 * torch.randn(B, d, generator=gw).numpy(), device=dev) for _ in range(3 + K)]
            pw = pred.clone()
            for j in range(3):
                pw = pw + drift[j]
                ow = store.cone_op(ids, pw, mode, -1.0, 0.2, check=False, outputs=outs)
            torch.cuda.synchronize()
            its, t0 = [], time.perf_counter()
            for j in range(K):
                pw = pw + drift[3 + j]
                ow = store.cone_op(ids, pw, mode, -1.0, 0.2, check=False, outputs=outs)
                its.append(ow["iters"])
            torch.cuda.synchronize()
            dtw = (time.perf_counter() - t0) / K
            assert bool((ow["status"] == 0).all())
            res["warm_start"] = {"ms_per_step": 1e3 * dtw, "projections_per_s": B / dtw,
                                 "newton_iters_mean": float(torch.stack(its).float().mean()),
                                 "newton_iters_max": int(torch.stack(its).max()),
                                 "note": "predictions drift by 0.01 * N(0,1) per step (cold start above: 0.05 * N(0,1) off the "
                                         "true costs); includes the elementwise update of the predictions"}
            store.enable_warm_start(False)
        if cpu_n > 0:
            from oracle import cave_oracle as O

            dense = synth.densify_on(items[:cpu_n], d, torch.device("cpu"), m_max).numpy()
            t0 = time.perf_counter()
            O.batch_project(-pred_np[:cpu_n], dense)
            dtc = time.perf_counter() - t0
            res["cpu_baseline"] = {"value": cpu_n / dtc, "unit": "projections/s", "cores": 1, "kind": "port",
                                   "sample": f"{cpu_n} instance(s) of this batch, serial, {dtc:.1f} s"}
        elif cpu_recorded is not None:
            res["cpu_baseline"] = dict(cpu_recorded)
        else:
            res["cpu_baseline"] = {"value": None, "unit": "projections/s", "cores": 1, "kind": "port", "sample": cpu_note}
        return res

    out = []
    specs = [
        dict(name="configs[2] TSP-50, CaVE Exact, 4096/8 GPUs", kind="tsp", size=50, B=512, mode=_lib.MODE_EXACT, chunk=32,
             cpu_n=2, cpu_note="", tag="tsp50"),
        # one TSP-100 projection takes scipy.optimize.nnls 26 minutes: recorded once (with the fixture), not re-timed per run
        dict(name="configs[3] TSP-100, QP branch of CaVE Hybrid (inner_ratio 0.2), 2048/4 GPUs", kind="tsp", size=100, B=512,
             mode=_lib.MODE_INNER, chunk=4, cpu_n=0, cpu_note="", tag="tsp100", hybrid_ratio=0.3,
             cpu_recorded={"value": 1.0 / (26 * 60), "unit": "projections/s", "cores": 1, "kind": "reference",
                           "sample": "RECORDED, not timed in this run: scipy.optimize.nnls (the reference's solver, "
                                     "src/cave.py:307) on 1 TSP-100 instance took 26 min when tests/golden/large.npz was "
                                     "generated (tests/golden/make_golden_large.py, this image's 8-core container)"}),
        # one 30x30 projection through the CPU port: ~45-75 s (SciPy: 44 s when the fixture was generated)
        dict(name="configs[4] shortest path 30x30, CaVE+ (inner), 8192/8 GPUs", kind="sp", size=(30, 30), B=1024,
             mode=_lib.MODE_INNER, chunk=32, cpu_n=1 if large_cpu else 0, tag="sp30",
             cpu_note="skipped (--no-large-cpu): one 30x30 instance takes the CPU port about a minute"),
    ]
    for spec in specs:
        try:
            out.append(run(**spec))
        except Exception as e:  # noqa: BLE001 - side measurement only
            out.append({"workload": spec["name"], "error": repr(e)[:300]})
    return out


if __name__ == "__main__":
    main()
