"""CPU tier: Gurobi-free tight-cone construction for the grid shortest path."""

import itertools

import numpy as np

from cave_amd import synth
from cave_amd.tight import SPConeDataset, sp_gen_data, sp_regret, sp_solve, sp_tight_normals
from oracle import cave_oracle as O


def _brute(cost, h, w):
    arcs = synth.sp_arcs(h, w)
    aid = {(int(a), int(b)): k for k, (a, b) in enumerate(arcs)}
    best = np.inf
    for downs in itertools.combinations(range(h + w - 2), h - 1):
        i = j = 0
        tot = 0.0
        for t in range(h + w - 2):
            u = i * w + j
            if t in downs:
                i += 1
            else:
                j += 1
            tot += cost[aid[(u, i * w + j)]]
        best = min(best, tot)
    return best


def test_dp_is_optimal():
    rng = np.random.default_rng(0)
    for h, w in ((2, 2), (3, 4), (4, 4), (5, 5)):
        for _ in range(5):
            c = rng.uniform(0.1, 2.0, len(synth.sp_arcs(h, w)))
            s, obj = sp_solve(c, h, w)
            assert abs(obj - _brute(c, h, w)) < 1e-9 and abs(float(c @ s) - obj) < 1e-6
            assert s.sum() == h + w - 2


def test_tight_cone_certifies_the_vertex():
    """test/test_dataset.py:64-79 restated: -mean(ctrs) as a cost vector re-yields the stored optimal
    solution, and the true cost's negation lies inside the cone (rnorm = 0: KKT at the optimum)."""
    h = w = 5
    x, c = sp_gen_data(12, 5, h, w)
    ds = SPConeDataset(x, c, h, w)
    assert len(ds) == 12 and ds.ctrs[0].shape == (2 * h * w + len(synth.sp_arcs(h, w)), 40)
    for i in range(len(ds)):
        A = ds.ctrs[i].numpy()
        s2, _ = sp_solve(-A.mean(axis=0), h, w)
        assert np.array_equal(s2, ds.sols[i].numpy())
        _, rn = O.project_nnls(-ds.costs[i].numpy(), A)
        assert rn < 1e-5
    assert abs(sp_regret(c, c, ds.objs.numpy()[:, 0], h, w)) < 1e-6  # float32 objective storage


# ------------------------------------------------------------------ small TSP (DFJ) without Gurobi

def _brute_force_tsp(cost, n):
    from itertools import permutations

    from cave_amd.synth import tsp_edges

    edges = tsp_edges(n)
    D = np.zeros((n, n))
    D[edges[:, 0], edges[:, 1]] = cost
    D[edges[:, 1], edges[:, 0]] = cost
    best = np.inf
    for perm in permutations(range(1, n)):
        t = (0,) + perm
        best = min(best, sum(D[t[i], t[(i + 1) % n]] for i in range(n)))
    return best


def test_held_karp_is_optimal_and_dfj_loop_agrees():
    from cave_amd.tight import tsp_dfj_cuts, tsp_gen_data, tsp_solve

    for n in (5, 7, 8):
        _, costs = tsp_gen_data(6, 5, n, seed=n)
        for c in costs:
            sol, obj, tour = tsp_solve(c, n)
            assert sorted(tour) == list(range(n)) and abs(float(c.astype(np.float64) @ sol) - obj) < 1e-4 and sol.sum() == n
            assert abs(obj - _brute_force_tsp(c, n)) < 1e-4
            sol2, obj2, cuts = tsp_dfj_cuts(c, n)
            assert abs(obj2 - obj) < 1e-4 and sol2.sum() == n


def test_tsp_tight_cone_checks_of_the_reference_dataset_test():
    """test/test_dataset.py:47-79 restated for the TSP: stored solutions are optimal (checkOptimal), and the
    negated mean of the tight normals, used as an objective, reproduces an optimal solution (checkBinding)."""
    from cave_amd.tight import TSPConeDataset, tsp_gen_data, tsp_solve

    n = 9
    feats, costs = tsp_gen_data(10, 5, n, seed=3)
    ds = TSPConeDataset(feats, costs, n)
    d = n * (n - 1) // 2
    assert sum(ds.tight_cuts) > 0, "the generator should produce instances whose DFJ loop needs cuts"
    for i in range(len(ds)):
        cost, sol, ctrs = ds.costs[i].numpy(), ds.sols[i].numpy(), ds.ctrs[i].numpy()
        _, obj, _ = tsp_solve(cost, n)
        assert np.isclose(obj, cost @ sol, rtol=1e-6)                              # checkOptimal
        sol2, _, _ = tsp_solve(-ctrs.mean(axis=0), n)
        assert np.isclose(cost @ sol, cost @ sol2, rtol=1e-6)                      # checkBinding
        # layout of _extract_tight_normals: +Deg, -Deg, cuts, -e_k, +e_k; every row tight at sol
        assert ctrs.shape[1] == d and np.array_equal(ctrs[:n], -ctrs[n:2 * n])
        k = ds.tight_cuts[i]
        rhs = np.r_[np.full(n, 2.0), np.full(n, -2.0), ctrs[2 * n:2 * n + k].sum(1) * 0 - 1 + np.array(
            [np.sqrt(2 * r.sum() + 0.25) + 0.5 for r in ctrs[2 * n:2 * n + k]]), np.zeros(int((sol < 0.5).sum())),
            np.ones(int((sol > 0.5).sum()))]
        assert np.allclose(ctrs @ sol, rhs, atol=1e-5)
        # the true cost lies in the polar of the tight cone's complement: -c is a non-negative combination
        # of the tight normals (LP feasibility), i.e. sol is optimal for the relaxation it was cut from
        from scipy.optimize import linprog

        res = linprog(np.zeros(len(ctrs)), A_eq=ctrs.T, b_eq=-cost, bounds=(0, None), method="highs")
        assert res.status == 0, i
