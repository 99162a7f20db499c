#!/usr/bin/env python
"""End-to-end CaVE training on the grid shortest path with `solver='hip'` — the loop of the
reference's code_sample.py:23-60 (linear predictor, Adam lr 1e-2, CaVE+ loss), BASELINE configs[0]
sizes by default (5x5 grid, 100 instances, batch 32), no Gurobi / PyEPO needed.

    python examples/train_sp_cave.py [--grid 5 5] [--num-data 100] [--batch 32] [--epochs 10] [--packed]
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
    args = ap.parse_args(argv)

    from cave_amd.cave import EPO, exactConeAlignedCosine, innerConeAlignedCosine
    from cave_amd.dataset import ConeStore, PackedBatch
    from cave_amd.tight import SPConeDataset, sp_gen_data, sp_regret
    from torch.nn.utils.rnn import pad_sequence

    h, w = args.grid
    feats, costs = sp_gen_data(args.num_data, args.num_feat, h, w, deg=4, noise_width=0.5, seed=135)
    dataset = SPConeDataset(feats, costs, h, w)
    dev = torch.device("cuda")

    class _Model:  # what the loss modules read from a PyEPO optModel
        modelSense = EPO.MINIMIZE

    if args.variant == "exact":
        cave = exactConeAlignedCosine(_Model(), solver="hip")
    elif args.variant == "inner":
        cave = innerConeAlignedCosine(_Model(), solver="hip", seed=0)
    else:
        cave = innerConeAlignedCosine(_Model(), solver="hip", solve_ratio=0.3, inner_ratio=0.2, seed=0)

    store = ConeStore.from_ragged(dataset.ctrs) if args.packed else None

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
    opt = torch.optim.Adam(reg.parameters(), lr=1e-2)

    def regret():
        with torch.no_grad():
            cp = reg(dataset.feats.to(dev)).cpu().numpy()
        return sp_regret(cp, dataset.costs.numpy(), dataset.objs.numpy()[:, 0], h, w)

    hist = [(0, float("nan"), regret())]
    print(f"epoch 0: regret {hist[0][2] * 100:.2f}%")
    t0 = time.time()
    for epoch in range(1, args.epochs + 1):
        tot = 0.0
        for x, c, cones in loader:
            x = x.to(dev)
            cp = reg(x)
            loss = cave(cp, PackedBatch(store, cones) if args.packed else cones.to(dev))
            opt.zero_grad()
            loss.backward()
            opt.step()
            tot += float(loss.detach()) * len(x)
        hist.append((epoch, tot / len(dataset), regret()))
        print(f"epoch {epoch}: loss {hist[-1][1]:.4f}  regret {hist[-1][2] * 100:.2f}%")
    print(f"training time {time.time() - t0:.2f} s ({args.epochs} epochs, {len(dataset)} instances, batch {args.batch})")
    return hist


if __name__ == "__main__":
    main()
