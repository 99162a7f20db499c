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
