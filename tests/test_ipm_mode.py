"""CAVE_MODE_INNER_IPM -- the truncated interior-point variant of CaVE+ (src/cave.py:213-214, 267-295).
Clarabel is not in the image, so parity is unpinned; these are the properties the mechanism must have:
the iterate is STRICTLY inside the cone (every multiplier > 0), it tends to the nnls projection as max_iter
grows, and the loss is finite on the reference's own test data (test/test_func.py:78-153).
CPU tier: the kernel code compiled for one serial lane (tests/emul); GPU tier: the HIP kernels via the module."""

import numpy as np
import pytest
from scipy.optimize import linprog

from emul_lib import Emul
from oracle import cave_oracle as O

MODE_IPM = 5


def interior_margin(A, p, eps):
    """min_t  s.t. |A^T lam - p|_inf <= t, lam >= eps   (t ~ 0 <=> p = A^T lam with every multiplier >= eps)."""
    A = A[np.abs(A).sum(1) > 1e-7].astype(np.float64)
    m, d = A.shape
    c = np.r_[np.zeros(m), 1.0]
    Aub = np.block([[A.T, -np.ones((d, 1))], [-A.T, -np.ones((d, 1))]])
    bub = np.r_[p, -p]
    res = linprog(c, A_ub=Aub, b_ub=bub, bounds=[(eps, None)] * m + [(0, None)], method="highs")
    assert res.status == 0
    return res.x[-1]


def _cases(golden):
    g, s = golden["generic"], golden["structured"]
    return {"generic": (g["generic_ctrs"], g["generic_costs"]), "setup": (g["setup_ctrs"], g["setup_costs"]),
            "sp5": (s["sp5_ctrs"][:8], s["sp5_costs"][:8]), "tsp20": (s["tsp20_ctrs"][:4], s["tsp20_costs"][:4])}


def check_ipm_properties(run, golden):
    for name, (ctrs, costs) in _cases(golden).items():
        po, ro = O.batch_project(-costs, ctrs)
        sc = float(np.abs(costs).max())
        errs = []
        for steps in (1, 3, 6, 12, 40):
            o = run(ctrs, costs, steps)
            assert (o["status"] == 0).all() and np.isfinite(o["loss"]).all() and np.isfinite(o["grad"]).all(), (name, steps)
            assert (o["loss"] >= -1e-6).all() and (o["loss"] <= 2 + 1e-6).all()
            errs.append(float(np.abs(o["proj"] - po).max() / sc))
            if steps == 3:  # the reference's default truncation: strictly interior, and not yet the projection
                for b in range(min(3, len(ctrs))):
                    assert interior_margin(ctrs[b], o["proj"][b].astype(np.float64), 1e-6 * sc) <= 1e-5 * sc, (name, b)
                res = np.linalg.norm(-costs.astype(np.float64) - o["proj"], axis=1)
                assert np.abs(res - o["rnorm"]).max() <= 1e-5 * max(1.0, res.max())  # rnorm is the iterate's residual
        assert errs[-1] <= 4e-6 and errs[-1] <= errs[1] and errs[2] <= errs[0], (name, errs)  # -> the nnls projection


def test_ipm_serial_kernel_code(golden):
    E = Emul()
    check_ipm_properties(lambda c, y, k: E.cone_dense(c, y, MODE_IPM, sign=-1.0, max_iter=k), golden)


def _large_cases():
    from cave_amd import synth

    c1, y1, _ = synth.sp_batch(12, 12, 2, seed=5)     # narrow band, 144 free rows
    c2, y2, _ = synth.tsp_batch(30, 2, seed=5)        # ~33 rows, the cut rows carry bounds (barrier terms)
    return {"sp12": (c1, y1), "tsp30": (c2, y2)}


def check_ipm_large(run):
    """The interior-point inner mode on the large-cone path: strictly interior by LP at the
    reference's max_iter = 3, convergence to the nnls projection at 40 steps (<= 4e-6), finite loss and gradient."""
    for name, (ctrs, costs) in _large_cases().items():
        po, _ = O.batch_project(-costs, ctrs)
        sc = float(np.abs(costs).max())
        errs = {}
        for steps in (3, 12, 40):
            o = run(ctrs, costs, steps)
            assert (o["status"] == 0).all() and np.isfinite(o["loss"]).all() and np.isfinite(o["grad"]).all(), (name, steps)
            errs[steps] = float(np.abs(o["proj"] - po).max() / sc)
            if steps == 3:
                assert interior_margin(ctrs[0], o["proj"][0].astype(np.float64), 1e-6 * sc) <= 1e-5 * sc, name
                assert (o["iters"] == 3).all()
        assert errs[40] <= 4e-6 and errs[40] <= errs[3], (name, errs)


def test_ipm_large_path_serial_kernel_code(golden):
    """Same properties through the large-cone code (band LDL^T / dense LDL^T per interior-point step), serial build:
    the small fixture cones forced onto that path, a 12x12 grid (band) and TSP-30 cones (bound rows)."""
    E = Emul()
    check_ipm_properties(lambda c, y, k: E.cone_dense_large(c, y, MODE_IPM, sign=-1.0, max_iter=k), golden)
    check_ipm_large(lambda c, y, k: E.cone_dense_large(c, y, MODE_IPM, sign=-1.0, max_iter=k))


@pytest.mark.gpu
def test_ipm_on_the_large_cone_path_gpu():
    """GPU: interior-point inner mode through cave_hip_cone_packed_large at 4 / 2 / 1 waves (one-wave band elimination,
    dense LDL^T), and the module on a packed store of 30x30 grids (BASELINE configs[4] names CaVE+)."""
    import torch

    from cave_amd import synth
    from cave_amd.cave import EPO, innerConeAlignedCosine
    from cave_amd.dataset import ConeStore, PackedBatch

    ALL = ("proj", "rnorm", "target", "loss", "grad")
    stores = {}
    for waves in (4, 2, 1):
        def run(c, y, k):
            key = c.shape
            if key not in stores:
                stores[key] = ConeStore.from_dense(torch.tensor(c, device="cuda"), chunk=2)
            st = stores[key]
            st.large, st.large_waves = True, waves
            if not st.band_entries:
                st.band_entries, st.max_bw = st._max_band_entries()
                st.large_lds = 64 * 1024
            o = st.cone_op(torch.arange(len(c), device="cuda"), torch.tensor(y, device="cuda"), MODE_IPM, -1.0, 0.0,
                           max_iter=k, outputs=ALL)
            return {kk: v.cpu().numpy() for kk, v in o.items()}
        check_ipm_large(run)

    class M:
        modelSense = EPO.MINIMIZE

    c, y, _ = synth.sp_batch(30, 30, 2, seed=3)
    store = ConeStore.from_dense(torch.tensor(c, device="cuda"), chunk=2)
    assert store.large
    batch = PackedBatch(store, torch.arange(2, device="cuda"))
    pred = torch.tensor(y, device="cuda")
    losses = {}
    for k in (3, 40):
        mod = innerConeAlignedCosine(M(), solver="hip", solver_kwargs={"inner": "ipm"}, max_iter=k, reduction="none")
        p = pred.clone().requires_grad_(True)
        l = mod(p, batch)
        l.sum().backward()
        assert torch.isfinite(l).all() and torch.isfinite(p.grad).all()
        losses[k] = l.detach()
    from cave_amd.cave import exactConeAlignedCosine

    exact = exactConeAlignedCosine(M(), solver="hip", reduction="none")(pred, batch)
    assert float((losses[40] - exact).abs().max()) <= 1e-5 and float((losses[3] - exact).abs().max()) > 1e-6


@pytest.mark.gpu
def test_ipm_hip_kernels_and_module(golden):
    import torch

    from cave_amd.cave import EPO, innerConeAlignedCosine
    from cave_amd.dataset import ConeStore, PackedBatch
    from cave_amd.qpsolver import cone_op_dense

    ALL = ("proj", "rnorm", "target", "loss", "grad")
    for waves in (0, 1, 4):
        def run(c, y, k):
            o = cone_op_dense(torch.tensor(c, device="cuda"), torch.tensor(y, device="cuda"), MODE_IPM, -1.0, 0.0,
                              max_iter=k, waves=waves, outputs=ALL)
            return {kk: v.cpu().numpy() for kk, v in o.items()}
        check_ipm_properties(run, golden)
    # same code on one serial lane: results agree to rounding
    E = Emul()
    for name, (ctrs, costs) in _cases(golden).items():
        a = E.cone_dense(ctrs, costs, MODE_IPM, sign=-1.0, max_iter=3)
        b = cone_op_dense(torch.tensor(ctrs, device="cuda"), torch.tensor(costs, device="cuda"), MODE_IPM, -1.0, 0.0, max_iter=3,
                          outputs=ALL)
        for k in ALL:
            assert np.abs(a[k] - b[k].cpu().numpy()).max() <= 2e-5 * max(1.0, np.abs(a[k]).max()), (name, k)

    class M:
        modelSense = EPO.MINIMIZE

    g = golden["structured"]
    ctrs, costs = torch.tensor(g["tsp20_ctrs"], device="cuda"), torch.tensor(g["tsp20_costs"], device="cuda")
    losses = {}
    for k in (1, 3, 30):
        mod = innerConeAlignedCosine(M(), solver="hip", solver_kwargs={"inner": "ipm"}, max_iter=k, reduction="none")
        p = costs.clone().requires_grad_(True)
        l = mod(p, ctrs)
        l.sum().backward()
        assert torch.isfinite(l).all() and torch.isfinite(p.grad).all()
        losses[k] = l.detach()
        store = ConeStore.from_dense(ctrs)
        lp = mod(costs, PackedBatch(store, torch.arange(len(ctrs), device="cuda")))
        assert float((lp - l.detach()).abs().max()) <= 2e-6
    # max_iter is honoured (the nnls-style arm ignores it), and many steps reproduce the exact-projection loss
    assert float((losses[1] - losses[3]).abs().max()) > 1e-4
    from cave_amd.cave import exactConeAlignedCosine

    exact = exactConeAlignedCosine(M(), solver="hip", reduction="none")(costs, ctrs)
    assert float((losses[30] - exact).abs().max()) <= 1e-5
    with pytest.raises(ValueError):
        innerConeAlignedCosine(M(), solver="hip", solver_kwargs={"inner": "simplex"})
