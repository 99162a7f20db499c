"""Gurobi-free tight-cone construction for shortest-path and small TSP instances (and data generators).

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

Small TSP (DFJ model, src/model/tsp.py:54-60): the optimal tour comes from the Held-Karp dynamic program
(`tsp_solve`, exact, O(n^2 2^n), n <= 14); the subtour-elimination rows come from the SAME lazy rule the
reference's callback applies (first edge that closes a cycle in union-find order -> the cut of its component,
src/model/tsp.py:41-60), applied to the successive optima of the cut relaxation {degree rows, cuts so far,
binary x} (`tsp_dfj_cuts`, solved with SciPy's HiGHS MILP).  Which cuts a branch-and-cut run generates depends
on the solver's search path, so the cut SET is a restatement of the rule, not of Gurobi's path; the tight ones
among them are then stacked exactly as `_extract_tight_normals` does (src/dataset.py:182-211):
`+Deg`, `-Deg`, tight cuts in `<=` orientation, `-e_k` at 0, `+e_k` at 1.
"""

from __future__ import annotations

import numpy as np

from .synth import sp_arcs

__all__ = ["sp_solve", "sp_tight_normals", "sp_gen_data", "SPConeDataset", "sp_regret",
           "tsp_solve", "tsp_dfj_cuts", "tsp_tight_normals", "tsp_gen_data", "TSPConeDataset", "tsp_regret"]


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


# ------------------------------------------------------------------ small TSP (DFJ)

def _edge_index(n: int):
    from .synth import tsp_edges

    edges = tsp_edges(n)
    eid = -np.ones((n, n), dtype=np.int64)
    eid[edges[:, 0], edges[:, 1]] = np.arange(len(edges))
    eid[edges[:, 1], edges[:, 0]] = np.arange(len(edges))
    return edges, eid


def tsp_solve(cost: np.ndarray, n: int):
    """Optimal tour of the symmetric TSP with edge costs `cost` (lexicographic (i<j) order, d = n(n-1)/2) by
    the Held-Karp dynamic program.  Returns (sol (d,) float32 0/1 edge indicator, objective, tour list)."""
    if n > 14:
        raise ValueError("Held-Karp is meant for n <= 14 here (2^n n^2 work and memory)")
    edges, eid = _edge_index(n)
    D = np.zeros((n, n))
    D[edges[:, 0], edges[:, 1]] = cost
    D[edges[:, 1], edges[:, 0]] = cost
    m = n - 1  # nodes 1..n-1 as bits 0..m-1; node 0 is the depot
    full = 1 << m
    dp = np.full((full, m), np.inf)
    parent = np.full((full, m), -1, dtype=np.int64)
    for j in range(m):
        dp[1 << j, j] = D[0, j + 1]
    Dm = D[1:, 1:]
    for mask in range(1, full):
        row = dp[mask]
        if not np.isfinite(row).any():
            continue
        # extend every end node k of `mask` to every j outside it
        cand = row[:, None] + Dm  # (k, j)
        best_k = np.argmin(cand, axis=0)
        best = cand[best_k, np.arange(m)]
        for j in range(m):
            if mask >> j & 1:
                continue
            nm = mask | (1 << j)
            if best[j] < dp[nm, j]:
                dp[nm, j] = best[j]
                parent[nm, j] = best_k[j]
    last = dp[full - 1] + D[1:, 0]
    j = int(np.argmin(last))
    obj = float(last[j])
    tour, mask = [], full - 1
    while j >= 0:
        tour.append(j + 1)
        pj = int(parent[mask, j])
        mask ^= 1 << j
        j = pj
    tour = [0] + tour[::-1]
    sol = np.zeros(len(edges), dtype=np.float32)
    for a, b in zip(tour, tour[1:] + tour[:1]):
        sol[eid[a, b]] = 1.0
    return sol, obj, tour


def _first_subtour(sel_edges, n: int):
    """The reference callback's detection (src/model/tsp.py:47-56): union-find over the selected edges in
    order; the first edge that closes a cycle names the component to cut (None if that is the whole tour)."""
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for i, j in sel_edges:
        ri, rj = find(i), find(j)
        if ri == rj:
            comp = [k for k in range(n) if find(k) == ri]
            return comp if len(comp) < n else None
        parent[ri] = rj
    return None


def tsp_dfj_cuts(cost: np.ndarray, n: int, max_rounds: int = 200):
    """DFJ cutting loop: solve {degree rows, cuts so far, x binary} exactly (HiGHS), cut the first subtour found
    by the reference's rule, repeat.  Returns (sol (d,) 0/1, objective, list of node sets cut)."""
    from scipy.optimize import Bounds, LinearConstraint, milp
    from scipy.sparse import csr_matrix, vstack

    edges, eid = _edge_index(n)
    d = len(edges)
    deg = np.zeros((n, d))
    deg[edges[:, 0], np.arange(d)] = 1.0
    deg[edges[:, 1], np.arange(d)] = 1.0
    cons = [LinearConstraint(csr_matrix(deg), 2.0, 2.0)]
    cuts: list[list[int]] = []
    for _ in range(max_rounds):
        res = milp(np.asarray(cost, dtype=np.float64), constraints=cons, integrality=np.ones(d), bounds=Bounds(0, 1))
        if res.status != 0:
            raise RuntimeError(f"cut relaxation not solved (status {res.status})")
        x = np.round(res.x)
        sel = [(int(edges[k, 0]), int(edges[k, 1])) for k in np.flatnonzero(x > 0.5)]
        comp = _first_subtour(sel, n)
        if comp is None:
            return x.astype(np.float32), float(cost @ x), cuts
        row = np.zeros((1, d))
        for a in range(len(comp)):
            for b in range(a + 1, len(comp)):
                row[0, eid[comp[a], comp[b]]] = 1.0
        cons.append(LinearConstraint(csr_matrix(row), -np.inf, len(comp) - 1.0))
        cuts.append(comp)
    raise RuntimeError("DFJ cutting loop did not terminate")


def tsp_tight_normals(sol: np.ndarray, n: int, cuts, tol: float = 1e-5) -> np.ndarray:
    """`_extract_tight_normals` (src/dataset.py:147-215) for the DFJ model at vertex `sol`: degree equalities
    (`+a` block then `-a` block), the lazy cuts that are tight at `sol` (already `<=` rows), `-e_k` at 0, `+e_k` at 1."""
    edges, eid = _edge_index(n)
    d = len(edges)
    deg = np.zeros((n, d), dtype=np.float32)
    deg[edges[:, 0], np.arange(d)] = 1.0
    deg[edges[:, 1], np.arange(d)] = 1.0
    rows = [deg, -deg]
    lazy = []
    for comp in cuts:
        r = np.zeros(d, dtype=np.float32)
        for a in range(len(comp)):
            for b in range(a + 1, len(comp)):
                r[eid[comp[a], comp[b]]] = 1.0
        if abs((len(comp) - 1.0) - float(r @ sol)) < tol:  # src/dataset.py:192-193
            lazy.append(r)
    if lazy:
        rows.append(np.asarray(lazy, dtype=np.float32))
    low = np.where(sol <= tol)[0]
    high = np.where((sol >= 1 - tol) & ~(sol <= tol))[0]
    low_rows = np.zeros((len(low), d), dtype=np.float32)
    low_rows[np.arange(len(low)), low] = -1.0
    high_rows = np.zeros((len(high), d), dtype=np.float32)
    high_rows[np.arange(len(high)), high] = 1.0
    return np.vstack(rows + [low_rows, high_rows]).astype(np.float32)


def tsp_gen_data(num_data: int, num_feat: int, n: int, deg: int = 4, noise_width: float = 0.5, seed: int = 42):
    """Features and edge costs in the style of PyEPO's TSP generator used by code_sample.py:20: Euclidean
    distances between random node positions plus a polynomial of a random linear map of the features, with
    multiplicative noise.  Returns (feats (N, p) float32, costs (N, d) float32)."""
    from .synth import tsp_edges

    rng = np.random.RandomState(seed)
    edges = tsp_edges(n)
    d = len(edges)
    pos = np.r_[rng.uniform(-2, 2, (n // 2, 2)), rng.normal(0, 1, (n - n // 2, 2))]
    dist = np.linalg.norm(pos[edges[:, 0]] - pos[edges[:, 1]], axis=1)
    Bm = rng.binomial(1, 0.5, (d, num_feat))
    x = rng.normal(0, 1, (num_data, num_feat))
    time = (x @ Bm.T / np.sqrt(num_feat) + 3.0) ** deg / 3.0 ** (deg - 1)
    time *= rng.uniform(1 - noise_width, 1 + noise_width, time.shape)
    return x.astype(np.float32), (dist[None, :] * 3.0 + time).astype(np.float32)


class TSPConeDataset:
    """`optDatasetConstrs` (src/dataset.py:26-130) for the DFJ TSP model without Gurobi."""

    def __init__(self, feats: np.ndarray, costs: np.ndarray, n: int):
        import torch

        self.n = n
        sols, objs, ctrs, ncuts = [], [], [], []
        for c in costs:
            s, o, cuts = tsp_dfj_cuts(c, n)
            sols.append(s)
            objs.append([o])
            A = tsp_tight_normals(s, n, cuts)
            ctrs.append(A)
            ncuts.append(len(A) - 2 * n - len(s))
        self.feats = torch.as_tensor(feats, dtype=torch.float32)
        self.costs = torch.as_tensor(costs, dtype=torch.float32)
        self.sols = torch.as_tensor(np.stack(sols), dtype=torch.float32)
        self.objs = torch.as_tensor(np.asarray(objs), dtype=torch.float32)
        self.ctrs = [torch.as_tensor(c, dtype=torch.float32) for c in ctrs]
        self.tight_cuts = ncuts

    def __len__(self) -> int:
        return len(self.feats)

    def __getitem__(self, i: int):
        return self.feats[i], self.costs[i], self.sols[i], self.objs[i], self.ctrs[i]


def tsp_regret(pred_costs: np.ndarray, true_costs: np.ndarray, true_objs: np.ndarray, n: int) -> float:
    """Normalised regret sum(c . w(c_hat) - z*) / sum(z*) with w(.) from Held-Karp."""
    loss = 0.0
    for cp, c, z in zip(pred_costs, true_costs, true_objs):
        s, _, _ = tsp_solve(cp, n)
        loss += float(c @ s) - float(z)
    return loss / float(np.abs(true_objs).sum() + 1e-7)
