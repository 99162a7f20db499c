"""The coordinate-form generators produce exactly the cones of the dense ones (same seed, same RNG order)."""

import numpy as np
import torch

from cave_amd import synth


def test_coo_generators_equal_dense_generators():
    for kind, size, dense in (("tsp", 9, lambda B, s: synth.tsp_batch(9, B, seed=s)),
                              ("sp", (4, 5), lambda B, s: synth.sp_batch(4, 5, B, seed=s))):
        for seed in (0, 3):
            ctrs, costs, sols = dense(6, seed)
            items, costs2, sols2 = synth.coo_batch(kind, size, 6, seed=seed)
            assert np.array_equal(costs, costs2) and np.array_equal(sols, sols2)
            got = synth.densify_on(items, ctrs.shape[2], torch.device("cpu")).numpy()
            assert got.shape == ctrs.shape and np.array_equal(got, ctrs)
