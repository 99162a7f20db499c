"""Gurobi-free tight-cone construction for shortest-path instances (and a data generator).

The reference obtains, per training instance, the optimal vertex from Gurobi and then stacks the
normals of the constraints tight at it (`optDatasetConstrs._getSols` + `_extract_tight_normals`,
/root/reference src/dataset.py:74-106,147-215).  For PyEPO's grid shortest-path model the LP is a
min-cost path on a DAG (arcs go right or down), so the optimal vertex comes from one dynamic
program and every rule of `_extract_tight_normals` can be applied without a solver:

  * every explicit constraint is a flow-conservation EQUALITY, always tight -> `+a` block then `-a`
    block (src/dataset.py:182-184);
  * no lazy cuts (src/dataset.py:186-196 contributes nothing);
  * every arc variable sits at a bound of its [0,1] box -> `-e_k` rows for arcs at 0, then `+e_k`
    rows for arcs at 1 (src/dataset.py:198-211).

This is the "next" row f2 of SURVEY.md §8 and lets BASELINE configs[0] (SP 5x5, 100 instances,
batch 32) run end to end without Gurobi/PyEPO.  Host-side numpy; nothing here is on the hot path.
"""

from __future__ import annotations

import numpy as np

from .synth import sp_arcs

__all__ = ["sp_solve", "sp_tight_normals", "sp_gen_data", "SPConeDataset", "sp_regret"]


def sp_solve(cost: np.ndarray, h: int, w: int):
    """Shortest source->sink path on the h x w grid DAG.  Returns (sol (d,) float32 0/1, objective)."""
    arcs = sp_arcs(h, w)
    n = h * w
    dist = np.full(n, np.inf)
    pred = np.full(n, -1, dtype=np.int64)
    dist[0] = 0.0
    # arcs are emitted row by row, every arc goes to a larger node index: one pass in order is a valid DP
    for k, (a, b) in enumerate(arcs):
        nd = dist[a] + cost[k]
        if nd < dist[b]:
            dist[b] = nd
            pred[b] = k
    sol = np.zeros(len(arcs), dtype=np.float32)
    v = n - 1
    while v != 0:
        k = pred[v]
        sol[k] = 1.0
        v = arcs[k][0]
    return sol, float(dist[n - 1])


def sp_tight_normals(sol: np.ndarray, h: int, w: int) -> np.ndarray:
    """`_extract_tight_normals` for the grid shortest-path model at vertex `sol` (float32 (m, d))."""
    arcs = sp_arcs(h, w)
    d = len(arcs)
    # flow conservation row of node v: +1 on arcs entering v, -1 on arcs leaving v
    N = np.zeros((h * w, d), dtype=np.float32)
    N[arcs[:, 1], np.arange(d)] = 1.0
    N[arcs[:, 0], np.arange(d)] = -1.0
    low = np.where(sol <= 1e-5)[0]
    high = np.where(sol >= 1 - 1e-5)[0]
    low_rows = np.zeros((len(low), d), dtype=np.float32)
    low_rows[np.arange(len(low)), low] = -1.0
    high_rows = np.zeros((len(high), d), dtype=np.float32)
    high_rows[np.arange(len(high)), high] = 1.0
    return np.vstack([N, -N, low_rows, high_rows]).astype(np.float32)


def sp_gen_data(num_data: int, num_feat: int, h: int, w: int, deg: int = 4, noise_width: float = 0.5, seed: int = 135):
    """Features and arc costs in the style of PyEPO's shortest-path generator (polynomial of degree
    `deg` of a random linear map, multiplicative noise), as used by code_sample.py:26-29 for TSP."""
    rng = np.random.RandomState(seed)
    d = len(sp_arcs(h, w))
    Bm = rng.binomial(1, 0.5, (d, num_feat))
    x = rng.normal(0, 1, (num_data, num_feat))
    c = (x @ Bm.T / np.sqrt(num_feat) + 3.0) ** deg + 1.0
    c /= 3.5 ** deg
    c *= rng.uniform(1 - noise_width, 1 + noise_width, c.shape)
    return x.astype(np.float32), c.astype(np.float32)


class SPConeDataset:
    """`optDatasetConstrs` (src/dataset.py:26-130) for the grid shortest path, without Gurobi:
    feats, costs, sols, objs and the ragged list of tight-constraint normals `ctrs`."""

    def __init__(self, feats: np.ndarray, costs: np.ndarray, h: int, w: int):
        import torch

        self.h, self.w = h, w
        sols, objs, ctrs = [], [], []
        for c in costs:
            s, o = sp_solve(c, h, w)
            sols.append(s)
            objs.append([o])
            ctrs.append(sp_tight_normals(s, h, w))
        self.feats = torch.as_tensor(feats, dtype=torch.float32)
        self.costs = torch.as_tensor(costs, dtype=torch.float32)
        self.sols = torch.as_tensor(np.stack(sols), dtype=torch.float32)
        self.objs = torch.as_tensor(np.asarray(objs), dtype=torch.float32)
        self.ctrs = [torch.as_tensor(c, dtype=torch.float32) for c in ctrs]

    def __len__(self) -> int:
        return len(self.feats)

    def __getitem__(self, i: int):
        return self.feats[i], self.costs[i], self.sols[i], self.objs[i], self.ctrs[i]


def sp_regret(pred_costs: np.ndarray, true_costs: np.ndarray, true_objs: np.ndarray, h: int, w: int) -> float:
    """Normalised regret sum(c . w(c_hat) - z*) / sum(z*) (PyEPO `metric.regret` for a MIN problem)."""
    loss = 0.0
    for cp, c, z in zip(pred_costs, true_costs, true_objs):
        s, _ = sp_solve(cp, h, w)
        loss += float(c @ s) - float(z)
    return loss / float(np.abs(true_objs).sum() + 1e-7)
