#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (solver='nnls') in this container.

Run from the repo root:   python tests/golden/make_golden.py
Needs /root/reference (read-only) and scipy; it is NOT run on the GPU box — the
fixtures it writes are committed, and tests only read those.

The reference imports PyEPO at module level (src/cave.py:17-18) and PyEPO is not
installed, so this script registers a ~30-line in-memory stand-in for the three
PyEPO names the loss modules use (`EPO`, `optModule`, `optModel`) before importing
`src.cave` from /root/reference.  Everything below `optModule` — `_batch_project`,
`_project_nnls`, `_average_ctrs`, the forward algebra — is the reference's own code
plus SciPy.  `reduction` and the unseeded branch RNG come from the stand-in, i.e.
they are parity-unpinned w.r.t. PyEPO.

Each projection the reference returns is also checked for self-consistency
(SciPy's reported rnorm vs ||cp - proj||): SciPy 1.15.3's Cython nnls returns
wrong answers on a small fraction of inputs (see tests/golden/scipy_defect.npz),
and a fixture must not freeze such an answer in as "golden".
"""

from __future__ import annotations

import os
import sys
import types
from enum import Enum

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def install_pyepo_standin():
    class EPO(Enum):
        MINIMIZE = 1
        MAXIMIZE = -1

    class optModel:  # noqa: N801
        pass

    class optModule(nn.Module):  # noqa: N801
        def __init__(self, optmodel, processes=1, solve_ratio=1.0, reduction="mean", dataset=None):
            super().__init__()
            self.optmodel = optmodel
            self.processes = processes
            self.pool = None
            self.solve_ratio = solve_ratio
            self.reduction = reduction
            self._branch_rng = np.random.RandomState()

        def _reduce(self, loss):
            return {"mean": loss.mean, "sum": loss.sum, "none": lambda: loss}[self.reduction]()

    pyepo = types.ModuleType("pyepo")
    pyepo.EPO = EPO
    func = types.ModuleType("pyepo.func")
    abc = types.ModuleType("pyepo.func.abcmodule")
    abc.optModule = optModule
    model = types.ModuleType("pyepo.model")
    opt = types.ModuleType("pyepo.model.opt")
    opt.optModel = optModel
    pyepo.func, func.abcmodule, pyepo.model, model.opt = func, abc, model, opt
    sys.modules.update({"pyepo": pyepo, "pyepo.func": func, "pyepo.func.abcmodule": abc,
                        "pyepo.model": model, "pyepo.model.opt": opt})
    return EPO


def main():
    EPO = install_pyepo_standin()
    sys.path.insert(0, REF)
    from src import cave as ref  # the reference, unmodified

    from cave_amd import synth

    class _M:
        def __init__(self, sense):
            self.modelSense = sense

    def run_modules(costs, ctrs, tag, out):
        """losses / grads / targets of the reference modules on one batch."""
        costs_t, ctrs_t = torch.as_tensor(costs), torch.as_tensor(ctrs)
        for sense_name, sense in (("min", EPO.MINIMIZE), ("max", EPO.MAXIMIZE)):
            sign = -1.0 if sense == EPO.MINIMIZE else 1.0
            signed = sign * costs_t
            proj, rnorm = ref._batch_project(signed, ctrs_t, "nnls", None, 1, None)
            out[f"{tag}_{sense_name}_proj"] = proj.numpy()
            out[f"{tag}_{sense_name}_rnorm"] = rnorm.numpy()
            # self-consistency of what SciPy returned
            true_r = (signed - proj).norm(dim=1).numpy()
            out[f"{tag}_{sense_name}_consistent"] = np.abs(true_r - rnorm.numpy()) <= 1e-5 * np.maximum(1.0, true_r)
            out[f"{tag}_{sense_name}_avg"] = ref._average_ctrs(ctrs_t).numpy()
            variants = {
                "exact": lambda red: ref.exactConeAlignedCosine(_M(sense), solver="nnls", reduction=red),
                "inner": lambda red: ref.innerConeAlignedCosine(_M(sense), solver="nnls", seed=42, reduction=red),
                "heur": lambda red: ref.innerConeAlignedCosine(_M(sense), solver="nnls", solve_ratio=0, seed=42,
                                                              reduction=red),
            }
            for vname, make in variants.items():
                mod = make("none")
                p = costs_t.clone().requires_grad_(True)
                loss = mod(p, ctrs_t)
                loss.sum().backward()
                out[f"{tag}_{sense_name}_{vname}_loss"] = loss.detach().numpy()
                out[f"{tag}_{sense_name}_{vname}_grad"] = p.grad.numpy()
                mod2 = make("none")
                with torch.no_grad():
                    out[f"{tag}_{sense_name}_{vname}_target"] = mod2._get_projection(signed, ctrs_t).numpy()
                out[f"{tag}_{sense_name}_{vname}_mean"] = np.float32(make("mean")(costs_t, ctrs_t).item())
                out[f"{tag}_{sense_name}_{vname}_sum"] = np.float32(make("sum")(costs_t, ctrs_t).item())

    # ---- (1) non-degenerate generic data of test/test_func.py:281-283
    out = {}
    torch.manual_seed(1)
    costs = torch.randn(8, 10)
    bctrs = torch.randn(8, 15, 10)
    out["generic_costs"], out["generic_ctrs"] = costs.numpy(), bctrs.numpy()
    run_modules(costs.numpy(), bctrs.numpy(), "generic", out)
    # ---- (2) the degenerate setUp data of test/test_func.py:34-43 (seed 0, rand)
    torch.manual_seed(0)
    costs = torch.rand(32, 10)
    bctrs = torch.rand(32, 15, 10)
    out["setup_costs"], out["setup_ctrs"] = costs.numpy(), bctrs.numpy()
    run_modules(costs.numpy(), bctrs.numpy(), "setup", out)
    # ---- (3) empty cone, (4) padded vs unpadded, zero prediction (test_func.py:165-193)
    p, r = ref._project_nnls(np.ones(4, np.float32), np.zeros((3, 4), np.float32), None)
    out["empty_proj"], out["empty_rnorm"] = p, np.float32(r)
    torch.manual_seed(0)
    pred = torch.rand(2, 6)
    ctrs_full = torch.rand(2, 5, 6)
    ctrs_padded = torch.cat([ctrs_full, torch.zeros(2, 10, 6)], dim=1)
    out["pad_pred"], out["pad_full"], out["pad_padded"] = pred.numpy(), ctrs_full.numpy(), ctrs_padded.numpy()
    m = ref.innerConeAlignedCosine(_M(EPO.MINIMIZE), solver="nnls", solve_ratio=0, seed=42)
    out["pad_heur_full"] = np.float32(m(pred, ctrs_full).item())
    out["pad_heur_padded"] = np.float32(m(pred, ctrs_padded).item())
    zp = torch.zeros(2, 6)
    zc = torch.rand(2, 3, 6)
    out["zero_ctrs"] = zc.numpy()
    out["zero_exact_loss"] = np.float32(ref.exactConeAlignedCosine(_M(EPO.MINIMIZE), solver="nnls")(zp, zc).item())
    # ---- (6) branch RNG stream (src/cave.py:195,201)
    out["rng42"] = np.random.RandomState(42).uniform(size=3)
    # hybrid: three consecutive forwards with solve_ratio 0.5, seed 7 (test_func.py:216-227 data)
    torch.manual_seed(0)
    pred = torch.rand(4, 6)
    ctrs = torch.rand(4, 5, 6) - 0.3
    out["hyb_pred"], out["hyb_ctrs"] = pred.numpy(), ctrs.numpy()
    hm = ref.innerConeAlignedCosine(_M(EPO.MINIMIZE), solver="nnls", solve_ratio=0.5, seed=7)
    out["hyb_losses"] = np.asarray([hm(pred, ctrs).item() for _ in range(3)], np.float32)
    np.savez_compressed(os.path.join(HERE, "generic.npz"), **out)
    print("generic.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if "loss" in k and "min" in k})

    # ---- (5) synthetic structured cones (SURVEY.md §8d generators, inputs stored too)
    out = {}
    c, y, w = synth.sp_batch(5, 5, 32, seed=0)
    out["sp5_ctrs"], out["sp5_costs"], out["sp5_sols"] = c, y, w
    run_modules(y, c, "sp5", out)
    c, y, w = synth.tsp_batch(20, 16, seed=0)
    out["tsp20_ctrs"], out["tsp20_costs"], out["tsp20_sols"] = c, y, w
    run_modules(y, c, "tsp20", out)
    np.savez_compressed(os.path.join(HERE, "structured.npz"), **out)
    for k in ("sp5_min_consistent", "tsp20_min_consistent"):
        print(k, out[k].all())

    # two TSP-50 instances: projection only (5.9 s each in SciPy); inputs regenerated from the seed
    c, y, _ = synth.tsp_batch(50, 2, seed=0)
    proj, rnorm = ref._batch_project(torch.as_tensor(-y), torch.as_tensor(c), "nnls", None, 1, None)
    np.savez_compressed(os.path.join(HERE, "tsp50.npz"), seed=0, n=50, batch=2, proj=proj.numpy(),
                        rnorm=rnorm.numpy())

    # ---- SciPy defect witness: the reference's own _project_nnls is wrong on this input
    A = np.array([[0.75700015, 0.68212354], [0.43556216, 0.13460635]], np.float32)
    cp = np.array([0.76166636, -0.14083835], np.float32)
    p, r = ref._project_nnls(cp, A, None)
    import scipy

    np.savez(os.path.join(HERE, "scipy_defect.npz"), A=A, cp=cp, ref_proj=p, ref_rnorm=np.float64(r),
             ref_true_resid=np.float64(np.linalg.norm(cp.astype(np.float64) - p.astype(np.float64))),
             scipy_version=np.array(scipy.__version__))
    print("scipy defect witness: reported rnorm", r, "true residual of returned proj",
          np.linalg.norm(cp - p), "scipy", scipy.__version__)


if __name__ == "__main__":
    main()
