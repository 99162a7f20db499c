import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    d = os.path.join(ROOT, "tests", "golden")
    return {
        "generic": np.load(os.path.join(d, "generic.npz")),
        "structured": np.load(os.path.join(d, "structured.npz")),
        "tsp50": np.load(os.path.join(d, "tsp50.npz")),
        "scipy_defect": np.load(os.path.join(d, "scipy_defect.npz")),
        "large": np.load(os.path.join(d, "large.npz")),
        "regress": np.load(os.path.join(d, "regress.npz")),
    }
