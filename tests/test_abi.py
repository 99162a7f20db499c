"""CPU tier: the C-ABI library builds for gfx950, loads, exports every symbol the header declares,
and the Python host layer enforces the reference's error behaviour.  No compute calls (no GPU here)."""

import os
import re
from unittest.mock import Mock

import pytest

from cave_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    _lib.build()
    lib = _lib.load_library()
    hdr = open(os.path.join(ROOT, "include", "cave_hip.h")).read()
    declared = set(re.findall(r"\b(cave_hip_\w+)\s*\(", hdr))
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cave_hip_version() == 10
    assert lib.cave_hip_device_count() >= 0
    assert int(re.search(r"#define CAVE_HIP_ABI_VERSION (\d+)", hdr).group(1)) == lib.cave_hip_version()


def test_graft_entry_build_runs():
    """The driver's "does it build" check (__graft_entry__.build): every library is up to date here, so this only walks
    the build steps and their assertions -- one of which compared the ABI version with a literal that two ABI bumps had
    left behind (found in round 4)."""
    import sys

    sys.path.insert(0, ROOT)
    import __graft_entry__

    __graft_entry__.build()


def test_default_limits_and_arg_validation():
    cap, lds = _lib.default_limits(235, 190)  # TSP-20
    assert 1500 <= cap <= 235 * 190 and 0 < lds <= 40 * 1024  # >= 4 workgroups per CU
    cap, lds = _lib.default_limits(15, 10)
    assert cap == 150 and lds <= 20 * 1024
    lib = _lib.load_library()
    # bad shapes are rejected before any launch (works without a GPU)
    assert lib.cave_hip_cone_dense(None, None, 1, 4, 0, 0, 1.0, 0.0, 0, 0, 0, 0, None, None, None, None, None, None, None, None) == -1
    assert b"bad shape" in lib.cave_hip_last_error()
    assert lib.cave_hip_cone_dense(None, None, 1, 4, 70000, 0, 1.0, 0.0, 0, 0, 0, 0, None, None, None, None, None, None, None, None) == -1
    assert lib.cave_hip_cone_dense(None, None, 1, 4, 4, 9, 1.0, 0.0, 0, 0, 0, 0, None, None, None, None, None, None, None, None) == -1
    assert lib.cave_hip_cone_dense(None, None, 0, 4, 4, 0, 1.0, 0.0, 0, 0, 0, 0, None, None, None, None, None, None, None, None) == 0  # B == 0
    # +-1 cones keep no value arrays; small ones (d <= 256, <= 32 rows) add the index structures of the one-wave solver
    assert 0 < lib.cave_hip_packed_lds_bytes(1225, 55, 3000, 1) < lib.cave_hip_packed_lds_bytes(1225, 55, 3000, 0)
    assert 0 < lib.cave_hip_packed_lds_bytes(190, 26, 700, 0) < lib.cave_hip_packed_lds_bytes(190, 26, 700, 1) <= 40 * 1024
    assert lib.cave_hip_packed_lds_bytes(190, 5000, 700, 0) == -1
    # large-cone entry points: sizing is monotone, bad workspaces / shapes are rejected before any launch
    s1 = lib.cave_hip_large_slice_bytes(5155, 4950, 40000, 16000)
    assert 0 < s1 < lib.cave_hip_large_slice_bytes(5155, 4950, 80000, 16000) < lib.cave_hip_large_slice_bytes(5155, 4950, 80000, 64000)
    assert lib.cave_hip_large_slice_bytes(5155, 0, 40000, 16000) == -1
    assert 0 < lib.cave_hip_packed_large_slice_bytes(1740, 900, 27900) < 1 << 20
    # v7: exact LDS of the band solver's hot arrays -- a narrow band (30x30 grid: 900 rows, half bandwidth 30) must
    # leave room for four workgroups per CU; a dense reduced system (TSP-100: 105 rows) lives in LDS as a FOLDED
    # triangle since round 3 (cone_dense.h) and must leave room for two
    grid = lib.cave_hip_packed_large_lds_bytes(900, 30)
    assert 0 < grid and 4 * grid <= 160 * 1024
    assert 48 * 1024 < lib.cave_hip_packed_large_lds_bytes(105, 104) <= 80 * 1024
    assert 80 * 1024 < lib.cave_hip_packed_large_lds_bytes(200, 199) <= 160 * 1024  # beyond 128 rows: the band window
    assert lib.cave_hip_packed_large_lds_bytes(-1, 3) < 0
    none8 = [None] * 8
    assert lib.cave_hip_cone_dense_large(None, None, 1, 4, 4, 0, 1.0, 0.0, 0, 64, 0, None, 1 << 20, 4, *none8) == -1  # null pointers
    assert lib.cave_hip_cone_dense_large(None, None, 1, 70000, 4, 0, 1.0, 0.0, 0, 64, 0, None, 1 << 20, 4, *none8) == -1
    assert b"bad shape" in lib.cave_hip_last_error()
    assert lib.cave_hip_cone_dense_large(None, None, 0, 4, 4, 0, 1.0, 0.0, 0, 64, 0, None, 0, 0, *none8) == 0  # B == 0
    assert lib.cave_hip_pack_large(None, 1, 4, 4, 64, None, 1 << 20, 4, None, None, None, 0, None, None) == -1
    assert lib.cave_hip_cone_packed_large(None, None, None, 1, 0, 1.0, 0.0, 0, 0, 0, None, 1 << 20, 4, *none8) == -1
    # v9 fused step: six workgroups per compute unit (four solve blocks + two two-wave pack blocks) must fit the LDS at
    # TSP-20; shapes beyond the one-wave solver are refused up front; bad arguments are rejected before any launch
    step = lib.cave_hip_step_lds_bytes(235, 190)
    assert 0 < step and 6 * step <= 160 * 1024
    assert 0 < lib.cave_hip_step_lds_bytes(60, 40) <= step
    assert lib.cave_hip_step_lds_bytes(235, 300) < 0 and lib.cave_hip_step_lds_bytes(40000, 190) < 0
    none7 = [None] * 7
    assert lib.cave_hip_cone_step(None, None, None, 0, 0, 1.0, 0.0, 0, 0, *none7, None, 0, 0, 190, None, None, None, None) == 0  # nothing to do
    assert lib.cave_hip_cone_step(None, None, None, 4, 2, 1.0, 0.2, 0, 0, *none7, None, 0, 0, 190, None, None, None, None) == -1
    assert b"cu_tickets" in lib.cave_hip_last_error()
    assert lib.cave_hip_lite_from_packed(None, None, None, None) == -1


def test_status_codes_match_header():
    hdr = open(os.path.join(ROOT, "include", "cave_hip.h")).read()
    for name, val in (("CAVE_ST_OK", _lib.ST_OK), ("CAVE_ST_NOT_CONVERGED", _lib.ST_NOT_CONVERGED),
                      ("CAVE_ST_TOO_LARGE", _lib.ST_TOO_LARGE), ("CAVE_ST_BAD_INPUT", _lib.ST_BAD_INPUT),
                      ("CAVE_MODE_PROJECT", _lib.MODE_PROJECT), ("CAVE_MODE_EXACT", _lib.MODE_EXACT),
                      ("CAVE_MODE_INNER", _lib.MODE_INNER), ("CAVE_MODE_HEURISTIC", _lib.MODE_HEURISTIC),
                      ("CAVE_MODE_AVG", _lib.MODE_AVG)):
        assert int(re.search(rf"#define {name} (\d+)", hdr).group(1)) == val


def _model(sense):
    m = Mock()
    m.modelSense = sense
    return m


def test_constructor_error_behaviour(monkeypatch):
    """test/test_func.py:195-214 restated for solver='hip'."""
    import torch

    from cave_amd.cave import EPO, exactConeAlignedCosine, innerConeAlignedCosine

    with pytest.raises(ValueError):
        exactConeAlignedCosine(_model(EPO.MINIMIZE), solver="bogus")
    for ref_solver in ("nnls", "clarabel", "apgd"):  # the reference's own backends are not shipped here
        with pytest.raises(ImportError):
            exactConeAlignedCosine(_model(EPO.MINIMIZE), solver=ref_solver)
    with pytest.raises(ImportError):  # the reference's default solver, as in code_sample.py:44
        innerConeAlignedCosine(_model(EPO.MINIMIZE), processes=1)
    import inspect

    sig = inspect.signature(innerConeAlignedCosine.__init__)  # src/cave.py:152-163
    assert list(sig.parameters)[1:] == ["optmodel", "solver", "solver_kwargs", "max_iter", "solve_ratio", "inner_ratio",
                                        "processes", "reduction", "seed"]
    assert sig.parameters["solver"].default == "clarabel" and sig.parameters["max_iter"].default == 3
    sig = inspect.signature(exactConeAlignedCosine.__init__)  # src/cave.py:93-100
    assert list(sig.parameters)[1:] == ["optmodel", "solver", "solver_kwargs", "processes", "reduction"]
    if not torch.cuda.is_available():
        with pytest.raises(ImportError):  # no device -> loud failure, no fallback (cf. src/cave.py:113-117)
            exactConeAlignedCosine(_model(EPO.MINIMIZE), solver="hip")
    monkeypatch.setattr(_lib, "load", lambda: None)
    with pytest.raises(ValueError):
        innerConeAlignedCosine(_model(EPO.MINIMIZE), solver="hip", solve_ratio=1.5)
    with pytest.raises(ValueError):
        innerConeAlignedCosine(_model(EPO.MINIMIZE), solver="hip", inner_ratio=-0.1)
    m = innerConeAlignedCosine(_model(EPO.MINIMIZE), solver="hip", solver_kwargs={"max_iter": 50}, seed=42)
    assert m.solver_kwargs == {"max_iter": 50} and m.max_iter == 3 and m.solver == "hip"
    # seeded branch RNG: same stream as the reference (src/cave.py:195,201)
    assert abs(m._branch_rng.uniform() - 0.3745401188473625) < 1e-15
    # sense handling (src/cave.py:62-67)
    from cave_amd.abcmodule import sense_sign

    assert sense_sign(EPO.MINIMIZE) == -1.0 and sense_sign(EPO.MAXIMIZE) == 1.0
    with pytest.raises(ValueError):
        sense_sign("sideways")


def test_no_oracle_or_emul_in_product_path():
    """The package must never import the oracle or the serial test build."""
    pkg = os.path.join(ROOT, "cave_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no oracle", "") or fn == "__init__.py" and False, fn
            assert "emul" not in src, fn
