"""Shared parity checks: any implementation of the hot path vs the committed reference outputs.

`impl(ctrs, costs, mode, sign, inner_ratio)` must return a dict with float32 numpy arrays
proj, rnorm, target, loss, grad (as produced by the C ABI).  Tolerances (stated here once):
  proj / target / grad : 2e-6 * max(1, |y|_inf)   (fp32 outputs of an fp64 solve vs SciPy fp64 -> fp32)
  rnorm                : 2e-6 * max(1, rnorm)
  loss                 : 2e-6 absolute            (north star asks for 1e-4)
"""

import numpy as np

MODE_PROJECT, MODE_EXACT, MODE_INNER, MODE_HEURISTIC, MODE_AVG = 0, 1, 2, 3, 4
VARIANTS = {"exact": MODE_EXACT, "inner": MODE_INNER, "heur": MODE_HEURISTIC}
TOL = 2e-6

CASES = [("generic", "generic"), ("generic", "setup"), ("structured", "sp5"), ("structured", "tsp20")]


def check_case(impl, golden, file, tag, inner_ratio=0.2):
    g = golden[file]
    ctrs, costs = g[f"{tag}_ctrs"], g[f"{tag}_costs"]
    for sense, sign in (("min", -1.0), ("max", 1.0)):
        ok = g[f"{tag}_{sense}_consistent"]  # instances where the reference's SciPy answer is self-consistent
        sc = np.maximum(1.0, np.abs(costs).max(axis=1))[:, None]
        o = impl(ctrs, costs, MODE_PROJECT, sign, 0.0)
        assert (o["status"] == 0).all()
        assert np.all(np.abs(o["proj"] - g[f"{tag}_{sense}_proj"])[ok] <= (TOL * sc * np.ones_like(o["proj"]))[ok])
        rn = g[f"{tag}_{sense}_rnorm"]
        assert np.all(np.abs(o["rnorm"] - rn)[ok] <= TOL * np.maximum(1.0, rn)[ok])
        # inside-the-cone decisions must agree exactly (src/cave.py:218)
        assert np.array_equal((o["rnorm"] < np.float32(1e-7))[ok], (rn < np.float32(1e-7))[ok])
        o = impl(ctrs, None, MODE_AVG, 1.0, 0.0)
        assert np.abs(o["target"] - g[f"{tag}_{sense}_avg"]).max() <= TOL
        for vname, mode in VARIANTS.items():
            o = impl(ctrs, costs, mode, sign, inner_ratio)
            sel = ok if mode != MODE_HEURISTIC else np.ones_like(ok)
            assert np.all(np.abs(o["loss"] - g[f"{tag}_{sense}_{vname}_loss"])[sel] <= TOL), (tag, sense, vname)
            assert np.all(np.abs(o["target"] - g[f"{tag}_{sense}_{vname}_target"])[sel] <= TOL * 4), (tag, sense, vname)
            gs = np.maximum(1.0, np.abs(g[f"{tag}_{sense}_{vname}_grad"]).max())
            assert np.all(np.abs(o["grad"] - g[f"{tag}_{sense}_{vname}_grad"])[sel] <= TOL * 4 * gs), (tag, sense, vname)
            if sel.all():
                assert abs(o["loss"].mean() - g[f"{tag}_{sense}_{vname}_mean"]) <= TOL
                assert abs(o["loss"].sum() - g[f"{tag}_{sense}_{vname}_sum"]) <= TOL * len(sel)


def check_regress(impl, regress):
    """tests/golden/regress.npz (made by tests/golden/make_regress.py with the reference itself): cones that once
    hit the Newton iteration cap, and tiny-norm predictions.  Tolerances are RELATIVE to max|y| here (the
    projection is positively homogeneous and these cases span |y| from 4 down to 2e-12)."""
    tags = sorted(k[:-2] for k in regress.files if k.endswith("_A"))
    assert len(tags) >= 9
    for tag in tags:
        A, y = regress[f"{tag}_A"], regress[f"{tag}_y"]
        o = impl(A[None], y[None], MODE_PROJECT, 1.0, 0.0)
        assert o["status"][0] == 0, (tag, o["status"])
        assert o["iters"][0] <= 40, (tag, o["iters"])
        sc = float(np.abs(y).max())
        assert np.abs(o["proj"][0] - regress[f"{tag}_proj"]).max() <= 4e-6 * sc, tag
        assert abs(o["rnorm"][0] - regress[f"{tag}_rnorm"]) <= 4e-6 * sc, tag
        # inside-the-cone decision of src/cave.py:218
        assert (o["rnorm"][0] < np.float32(1e-7)) == (regress[f"{tag}_rnorm"] < np.float32(1e-7)), tag
