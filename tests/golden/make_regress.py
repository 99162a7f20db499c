#!/usr/bin/env python
"""Generate tests/golden/regress.npz: instances that once defeated the GPU solver, with the REFERENCE's
own answers (solver='nnls' path, `_project_nnls`, src/cave.py:298-309).

Inputs (fixture data, kept in the .npz itself so this script can be re-run):
  * r1..r5  the five cones on which the round-1 fuzzer (tools/fuzz/fuzz_gpu.py) hit the Newton iteration cap
            before the zig-zag extrapolation and the residual floor (cone_core.h) -- degenerate cones with
            duplicated rows and the prediction inside the cone;
  * t1..t4  tiny-norm predictions (|y| from 1e-6 down to 1e-12) on a generic cone: a residual floor that is
            absolute in |y| would declare these converged at theta = 0;
  * p1      (round 3, tools/fuzz/fuzz_gpu.py seed 701) the NEGATIVE of a point of the cone: theta = 0 is optimal
            to the float32 resolution of y, the initial projected gradient is ~6e-8 and a convergence test relative
            to it alone can never be met -- the solver returned the right projection flagged NOT_CONVERGED until
            the gradient test got its rounding floor (cone_core.h, gfloor).
  * p2      (round 4, tools/fuzz/fuzz_gpu.py seed 3001) the same situation on a 6 x 16 +-1 cone with a duplicated row: the
            iteration stagnated at a projected gradient of 1e-12 (35 rounding floors) with f no longer decreasing, and
            was flagged NOT_CONVERGED with the projection exact to 2e-16 -- the stagnation exits now accept 1e3 floors.
Run from the repo root (needs /root/reference and scipy; NOT run on the GPU box):
    python tests/golden/make_regress.py
"""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]


def main():
    from make_golden import REF, install_pyepo_standin

    install_pyepo_standin()
    sys.path.insert(0, REF)
    from src import cave as ref  # the reference, unmodified

    path = os.path.join(HERE, "regress.npz")
    cases = {}
    if os.path.exists(path):
        old = np.load(path)
        for k in old.files:
            if k.endswith("_A") or k.endswith("_y"):
                cases[k] = old[k]
    else:
        for i in range(1, 6):
            z = np.load(os.path.join(ROOT, "gpurun_out", f"noconv_{i}.npz"))
            cases[f"r{i}_A"], cases[f"r{i}_y"] = z["A"], z["y"]
        rng = np.random.default_rng(77)
        A = rng.standard_normal((12, 7)).astype(np.float32)
        y0 = rng.standard_normal(7).astype(np.float32)
        for j, s in enumerate((1e-6, 3e-8, 1e-9, 1e-12), 1):
            cases[f"t{j}_A"], cases[f"t{j}_y"] = A, (y0 * np.float32(s)).astype(np.float32)
    if "p1_A" not in cases:  # added in round 3 (the prediction enters with the sign the kernel saw: sign * pred = -y)
        z = np.load(os.path.join(ROOT, "tools", "diag", "noconv_r03.npz"))
        cases["p1_A"], cases["p1_y"] = z["A"], (-z["y"]).astype(np.float32)
    if "p2_A" not in cases:  # added in round 4
        z = np.load(os.path.join(ROOT, "tools", "diag", "noconv_r04.npz"))
        cases["p2_A"], cases["p2_y"] = z["A"], (-z["y"]).astype(np.float32)
    out = dict(cases)
    for k in sorted(cases):
        if not k.endswith("_A"):
            continue
        tag = k[:-2]
        A, y = cases[k], cases[f"{tag}_y"]
        proj, rnorm = ref._project_nnls(y.copy(), A.copy(), None)
        proj = np.asarray(proj, dtype=np.float32)
        true_r = float(np.linalg.norm(y.astype(np.float64) - proj.astype(np.float64)))
        out[f"{tag}_proj"] = proj
        out[f"{tag}_rnorm"] = np.float32(rnorm)
        out[f"{tag}_consistent"] = np.bool_(abs(true_r - float(rnorm)) <= 1e-6 * max(1.0, np.abs(y).max()))
        print(tag, A.shape, "rnorm", float(rnorm), "|y|", float(np.linalg.norm(y)), "consistent", bool(out[f"{tag}_consistent"]))
    np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
