#!/usr/bin/env python
"""End-to-end CaVE training with `solver='hip'` — the loop of the reference's code_sample.py:23-60 (linear
predictor, Adam lr 1e-2, CaVE+ loss), no Gurobi / PyEPO needed.  Default: the grid shortest path at
BASELINE configs[0] sizes (5x5 grid, 100 instances, batch 32); `--problem tsp` is code_sample.py's own problem
(DFJ TSP) at a size the Held-Karp / HiGHS tight-cone builder of cave_amd/tight.py handles exactly.

    python examples/train_sp_cave.py [--grid 5 5] [--num-data 100] [--batch 32] [--epochs 10] [--packed]
    python examples/train_sp_cave.py --problem tsp --nodes 10 --packed --warm-start
    python examples/train_sp_cave.py --grid 30 30 --num-data 64 --batch 32 --epochs 3 --packed --inner ipm
    python examples/train_sp_cave.py --packed --graph          # the whole step (predictor, loss, backward, Adam) as one HIP graph
    python examples/train_sp_cave.py --problem tsp --prefetch  # dense cones: the next batch's pack rides in this batch's loss call
"""

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
from torch import nn
from torch.utils.data import DataLoader


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, nargs=2, default=[5, 5])
    ap.add_argument("--num-data", type=int, default=100)
    ap.add_argument("--num-feat", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--variant", default="inner", choices=["exact", "inner", "hybrid"])
    ap.add_argument("--packed", action="store_true", help="device-resident packed cones instead of dense padding")
    ap.add_argument("--problem", default="sp", choices=["sp", "tsp"])
    ap.add_argument("--nodes", type=int, default=10, help="TSP size (Held-Karp: <= 14)")
    ap.add_argument("--inner", default="push", choices=["push", "ipm"],
                    help="CaVE+ interior point: 'push' (nnls-style: exact projection pushed towards the average normal) or "
                         "'ipm' (truncated interior-point iterate, as the reference's Clarabel max_iter=3: src/cave.py:213-214)")
    ap.add_argument("--max-iter", type=int, default=3, help="interior-point steps of --inner ipm")
    ap.add_argument("--warm-start", action="store_true",
                    help="(with --packed) start each projection from the multipliers of the previous epoch")
    ap.add_argument("--graph", action="store_true",
                    help="(with --packed, not hybrid) capture predictor + loss + backward + Adam of a full batch in ONE HIP "
                         "graph and replay it per step (the C-ABI launch path allocates nothing and never syncs when the "
                         "status check is off); a ragged last batch runs eagerly")
    ap.add_argument("--prefetch", action="store_true",
                    help="(dense cones) wrap the DataLoader in cave_amd.dataset.prefetch: the loop body stays as it is and the "
                         "pack stage of batch i+1 rides in the launch of batch i's loss")
    ap.add_argument("--lazy-check", action="store_true", help="solver_kwargs check='lazy': no host sync per step")
    args = ap.parse_args(argv)
    if args.graph and (not args.packed or args.variant == "hybrid"):
        ap.error("--graph needs --packed and a variant without a per-call branch draw")

    from cave_amd.cave import EPO, exactConeAlignedCosine, innerConeAlignedCosine
    from cave_amd.dataset import ConeStore, PackedBatch, prefetch
    from cave_amd.tight import (SPConeDataset, TSPConeDataset, sp_gen_data, sp_regret, tsp_gen_data, tsp_regret)
    from torch.nn.utils.rnn import pad_sequence

    h, w = args.grid
    if args.problem == "tsp":
        feats, costs = tsp_gen_data(args.num_data, args.num_feat, args.nodes, deg=4, noise_width=0.5, seed=42)
        dataset = TSPConeDataset(feats, costs, args.nodes)
        print(f"TSP-{args.nodes}: {len(dataset)} instances, {sum(dataset.tight_cuts)} tight subtour cuts in all")
    else:
        feats, costs = sp_gen_data(args.num_data, args.num_feat, h, w, deg=4, noise_width=0.5, seed=135)
        dataset = SPConeDataset(feats, costs, h, w)
    dev = torch.device("cuda")

    class _Model:  # what the loss modules read from a PyEPO optModel
        modelSense = EPO.MINIMIZE

    kw = {}
    if args.inner != "push":
        kw["inner"] = args.inner
    if args.graph:
        kw["check"] = False   # (a captured step cannot read the status back; it is examined after each replay below)
    elif args.lazy_check:
        kw["check"] = "lazy"
    if args.variant == "exact":
        cave = exactConeAlignedCosine(_Model(), solver="hip", solver_kwargs=kw or None)
    elif args.variant == "inner":
        cave = innerConeAlignedCosine(_Model(), solver="hip", seed=0, max_iter=args.max_iter, solver_kwargs=kw or None)
    else:
        cave = innerConeAlignedCosine(_Model(), solver="hip", solve_ratio=0.3, inner_ratio=0.2, seed=0, solver_kwargs=kw or None)

    store = ConeStore.from_ragged(dataset.ctrs) if args.packed else None
    if store is not None and args.warm_start:
        store.enable_warm_start()

    def collate(batch):  # reference collate_fn (src/dataset.py:133-144) / its id-returning replacement
        idx = torch.as_tensor(batch, dtype=torch.int64)
        x, c = dataset.feats[idx], dataset.costs[idx]
        if args.packed:
            return x, c, idx
        return x, c, pad_sequence([dataset.ctrs[i] for i in batch], batch_first=True, padding_value=0.0)

    loader = DataLoader(list(range(len(dataset))), batch_size=args.batch, shuffle=True, collate_fn=collate,
                        generator=torch.Generator().manual_seed(0))
    d = dataset.costs.shape[1]
    torch.manual_seed(0)
    reg = nn.Linear(args.num_feat, d).to(dev)
    opt = torch.optim.Adam(reg.parameters(), lr=1e-2, capturable=args.graph)

    graph = None
    if args.graph:
        # static inputs of the captured step; three warm-up steps on a side stream (as torch.cuda.graph wants), the
        # capture, then parameters and Adam state back to their initial values IN PLACE (the graph holds their addresses)
        gx = torch.zeros(args.batch, args.num_feat, device=dev)
        gids = torch.zeros(args.batch, dtype=torch.int64, device=dev)
        gbatch = PackedBatch(store, gids)
        init = [p_.detach().clone() for p_ in reg.parameters()]

        def gstep():
            loss = cave(reg(gx), gbatch)
            opt.zero_grad(set_to_none=False)
            loss.backward()
            opt.step()
            return loss

        x0, _, i0 = next(iter(loader))
        if len(i0) == args.batch:
            gx.copy_(x0)
            gids.copy_(i0)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    gstep()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gloss = gstep()
            with torch.no_grad():
                for p_, p0 in zip(reg.parameters(), init):
                    p_.copy_(p0)
                    if p_.grad is not None:
                        p_.grad.zero_()
                for st_ in opt.state.values():
                    for v_ in st_.values():
                        if torch.is_tensor(v_):
                            v_.zero_()
            if store is not None and args.warm_start:
                store.reset_warm_start()
            loader = DataLoader(list(range(len(dataset))), batch_size=args.batch, shuffle=True, collate_fn=collate,
                                generator=torch.Generator().manual_seed(0))  # the same batch order as an eager run

    def regret():
        with torch.no_grad():
            cp = reg(dataset.feats.to(dev)).cpu().numpy()
        if args.problem == "tsp":
            return tsp_regret(cp, dataset.costs.numpy(), dataset.objs.numpy()[:, 0], args.nodes)
        return sp_regret(cp, dataset.costs.numpy(), dataset.objs.numpy()[:, 0], h, w)

    hist = [(0, float("nan"), regret())]
    print(f"epoch 0: regret {hist[0][2] * 100:.2f}%")
    t0 = time.time()
    iters_log = []
    for epoch in range(1, args.epochs + 1):
        tot = 0.0
        it_sum, it_max, it_n = 0.0, 0, 0
        for x, c, cones in (prefetch(loader) if args.prefetch and not args.packed else loader):
            if graph is not None and len(x) == args.batch:
                gx.copy_(x, non_blocking=True)
                gids.copy_(cones, non_blocking=True)
                graph.replay()
                loss = gloss
                if bool((store.last_status != 0).any()):
                    raise RuntimeError("solver='hip': a projection of the captured step failed")
            else:
                x = x.to(dev)
                cp = reg(x)
                loss = cave(cp, PackedBatch(store, cones) if args.packed else cones.to(dev))
                opt.zero_grad()
                loss.backward()
                opt.step()
            tot += float(loss.detach()) * len(x)
            if store is not None and getattr(store, "last_iters", None) is not None:
                li = store.last_iters
                it_sum, it_max, it_n = it_sum + float(li.sum()), max(it_max, int(li.max())), it_n + li.numel()
        hist.append((epoch, tot / len(dataset), regret()))
        extra = ""
        if it_n:
            iters_log.append((it_sum / it_n, it_max))
            extra = f"  Newton iterations mean {it_sum / it_n:.2f} max {it_max}"
        print(f"epoch {epoch}: loss {hist[-1][1]:.4f}  regret {hist[-1][2] * 100:.2f}%{extra}")
    main.iters_log = iters_log
    print(f"training time {time.time() - t0:.2f} s ({args.epochs} epochs, {len(dataset)} instances, batch {args.batch})")
    return hist


if __name__ == "__main__":
    main()
