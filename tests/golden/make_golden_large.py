#!/usr/bin/env python
"""Generate tests/golden/large.npz: reference outputs (solver='nnls') for cones beyond the LDS-resident
solver — 12x12 and 30x30 grid shortest-path cones (SciPy needs about a minute per 30x30 instance) and,
with --tsp100, one TSP-100 instance (26 minutes in SciPy; the committed large.npz was made with it).  Inputs are regenerated from the seeds by cave_amd.synth.

    python tests/golden/make_golden_large.py [--tsp100]

Same import recipe and self-consistency check as make_golden.py; needs /root/reference, never run on
the GPU box.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, install_pyepo_standin  # noqa: E402


def main():
    install_pyepo_standin()
    sys.path.insert(0, REF)
    from src import cave as ref  # the reference, unmodified

    from cave_amd import synth

    out = {}

    def run(tag, c, y):
        t = time.time()
        proj, rnorm = ref._batch_project(torch.as_tensor(-y), torch.as_tensor(c), "nnls", None, 1, None)
        proj, rnorm = proj.numpy(), rnorm.numpy()
        true = np.linalg.norm((-y).astype(np.float64) - proj.astype(np.float64), axis=1)
        out[f"{tag}_proj"], out[f"{tag}_rnorm"] = proj, rnorm
        out[f"{tag}_consistent"] = np.abs(true - rnorm) <= 1e-4 * np.maximum(1.0, true)
        print(tag, c.shape, "%.1fs" % (time.time() - t), "consistent", out[f"{tag}_consistent"], flush=True)

    c, y, _ = synth.sp_batch(12, 12, 4, seed=0)
    run("sp12", c, y)
    c, y, _ = synth.sp_batch(30, 30, 1, seed=0)
    run("sp30", c, y)
    if "--tsp100" in sys.argv:
        c, y, _ = synth.tsp_batch(100, 1, seed=0)
        run("tsp100", c, y)
    np.savez_compressed(os.path.join(HERE, "large.npz"), **out)


if __name__ == "__main__":
    main()
