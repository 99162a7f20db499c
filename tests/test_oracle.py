"""CPU tier: the oracle (oracle/) against every committed output of the reference itself."""

import numpy as np
import pytest

from golden_cases import CASES, MODE_AVG, MODE_EXACT, MODE_HEURISTIC, MODE_INNER, MODE_PROJECT, check_case
from oracle import cave_oracle as O


def oracle_impl(ctrs, costs, mode, sign, inner_ratio):
    B, m, d = ctrs.shape
    out = {"status": np.zeros(B, np.int32)}
    if mode == MODE_AVG:
        out["target"] = O.average_ctrs(ctrs)
        return out
    signed = np.float32(sign) * costs
    if mode == MODE_PROJECT:
        out["proj"], out["rnorm"] = O.batch_project(signed, ctrs)
        return out
    if mode == MODE_EXACT:
        t = O.exact_target(signed, ctrs)[0]
    elif mode == MODE_INNER:
        t = O.inner_target(signed, ctrs, inner_ratio)[0]
    else:
        t = O.heuristic_target(signed, ctrs, inner_ratio)
    out["target"] = t
    out["loss"] = O.cone_loss(costs, t, sign)
    out["grad"] = O.cone_loss_grad(costs, t, sign)
    return out


@pytest.mark.parametrize("file,tag", CASES)
def test_oracle_matches_reference_outputs(golden, file, tag):
    check_case(oracle_impl, golden, file, tag)


def test_oracle_tsp50(golden):
    from cave_amd import synth

    g = golden["tsp50"]
    c, y, _ = synth.tsp_batch(int(g["n"]), int(g["batch"]), seed=int(g["seed"]))
    proj, rnorm = O.batch_project(-y, c)
    assert np.abs(proj - g["proj"]).max() <= 4e-6
    assert np.abs(rnorm - g["rnorm"]).max() <= 4e-6


def test_empty_cone_and_padding(golden):
    g = golden["generic"]
    p, r = O.project_nnls(np.ones(4, np.float32), np.zeros((3, 4), np.float32))
    assert np.array_equal(p, g["empty_proj"]) and r == float(g["empty_rnorm"]) == 0.0
    # zero-padded rows change nothing (test/test_func.py:165-180)
    lo = O.ConeLossOracle(inner=True, solve_ratio=0, seed=42)
    a, _ = lo(g["pad_pred"], g["pad_full"])
    b, _ = lo(g["pad_pred"], g["pad_padded"])
    assert abs(a - g["pad_heur_full"]) <= 1e-6 and abs(b - g["pad_heur_padded"]) <= 1e-6 and abs(a - b) <= 1e-6
    # zero prediction -> finite loss (test/test_func.py:182-193)
    z, _ = O.ConeLossOracle()(np.zeros((2, 6), np.float32), g["zero_ctrs"])
    assert np.isfinite(z) and abs(z - g["zero_exact_loss"]) <= 1e-6


def test_branch_rng_and_hybrid_sequence(golden):
    g = golden["generic"]
    assert np.allclose(np.random.RandomState(42).uniform(size=3), g["rng42"], rtol=0, atol=0)
    lo = O.ConeLossOracle(inner=True, solve_ratio=0.5, seed=7)
    seq = [lo(g["hyb_pred"], g["hyb_ctrs"])[0] for _ in range(3)]
    assert np.abs(np.asarray(seq) - g["hyb_losses"]).max() <= 2e-6


def test_scipy_defect_witness(golden):
    """The reference's own answer on this input is self-inconsistent (SciPy 1.15.3 Cython nnls);
    the oracle returns the true projection (a 2-ray cone in R^2 can be checked by hand)."""
    g = golden["scipy_defect"]
    A, cp = g["A"], g["cp"]
    assert abs(float(g["ref_rnorm"]) - float(g["ref_true_resid"])) > 0.1  # reported != actual residual
    p, r = O.project_nnls(cp, A)
    # candidates: 0, projection on each ray, y itself if inside
    best = min(
        [np.linalg.norm(cp)] + [np.linalg.norm(cp - max(0.0, cp @ a / (a @ a)) * a) for a in A.astype(np.float64)])
    assert abs(r - best) <= 1e-7 and abs(np.linalg.norm(cp - p) - best) <= 1e-6
    assert best < float(g["ref_true_resid"]) - 0.1


def test_oracle_kkt_random():
    """Optimality certificate on random dense cones: dual feasibility + complementarity + primal membership."""
    rng = np.random.default_rng(5)
    for _ in range(200):
        m, d = int(rng.integers(1, 30)), int(rng.integers(1, 16))
        A = rng.standard_normal((m, d)).astype(np.float32)
        y = rng.standard_normal(d).astype(np.float32)
        p, r = O.project_nnls(y, A)
        res = y.astype(np.float64) - p
        assert (A.astype(np.float64) @ res).max() <= 1e-5
        assert abs(res @ p) <= 1e-5 * max(1.0, float(y @ y))
        assert abs(np.linalg.norm(res) - r) <= 1e-5
