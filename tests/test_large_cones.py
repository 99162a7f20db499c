"""CPU tier for the large-cone path (BASELINE configs 4 and 5: TSP-100, 30x30 shortest path).

The per-instance code shared with the HIP kernels runs here on one serial lane (tests/emul); it is
checked against the reference's own outputs (tests/golden/large.npz) and, at the full instance sizes
where SciPy / the oracle take minutes to hours, by a KKT certificate (tests/certificate.py)."""

import numpy as np
import pytest

from certificate import assert_projection, kkt_certificate
from emul_lib import Emul
from golden_cases import MODE_PROJECT
from oracle import cave_oracle as O


@pytest.fixture(scope="module")
def emul():
    return Emul()


def test_certificate_accepts_projections_and_rejects_others():
    from cave_amd import synth

    for c, y in (synth.tsp_batch(20, 3, seed=0)[:2], synth.sp_batch(5, 5, 3, seed=0)[:2]):
        po, _ = O.batch_project(-y, c)
        for b in range(3):
            assert_projection(c[b], -y[b], po[b])
            bad = po[b].copy()
            bad[np.argmax(np.abs(bad))] += 1e-3                       # off the face: complementarity / membership fail
            k = kkt_certificate(c[b], -y[b], bad)
            assert k["dual"] > 4e-6 or k["comp"] > 4e-6 or not k["member"]
            k = kkt_certificate(c[b], -y[b], 0.5 * po[b])           # in the cone, not the nearest point
            assert k["dual"] > 4e-6 or k["comp"] > 4e-6
    g, yy = synth.generic_batch(6)[:2]
    po, _ = O.batch_project(yy, g)
    for b in range(6):
        assert_projection(g[b], yy[b], po[b])
        assert not kkt_certificate(g[b], yy[b], yy[b] + 1.0)["member"] or np.allclose(po[b], yy[b])


def test_large_path_code_vs_reference_fixture(emul, golden):
    from cave_amd import synth

    g = golden["large"]
    for tag, (h, n) in (("sp12", (12, 4)), ("sp30", (30, 1)), ("tsp100", (100, 1))):
        c, y, _ = synth.tsp_batch(h, n, seed=0) if tag == "tsp100" else synth.sp_batch(h, h, n, seed=0)
        o = emul.cone_dense_large(c, y, MODE_PROJECT, sign=-1.0)
        assert (o["status"] == 0).all() and o["iters"].max() <= 20
        ok = g[f"{tag}_consistent"]
        sc = max(1.0, np.abs(y).max())
        assert np.abs(o["proj"] - g[f"{tag}_proj"])[ok].max() <= 2e-6 * sc
        assert np.abs(o["rnorm"] - g[f"{tag}_rnorm"])[ok].max() <= 2e-6 * sc


@pytest.mark.parametrize("which", ["tsp100", "sp30"])
def test_full_size_instances_are_certified(emul, which):
    from cave_amd import synth

    c, y, _ = synth.tsp_batch(100, 2, seed=1) if which == "tsp100" else synth.sp_batch(30, 30, 3, seed=1)
    o = emul.cone_dense_large(c, y, MODE_PROJECT, sign=-1.0)
    assert (o["status"] == 0).all() and o["iters"].max() <= 20
    for b in range(c.shape[0]):
        cert = assert_projection(c[b], -y[b], o["proj"][b], what=(which, b))
        assert cert["n_general_tight"] >= (200 if which == "tsp100" else 1800)
