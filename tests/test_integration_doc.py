"""INTEGRATION.md is tested text: the fenced ctypes stubs of its §1 and §6 are executed verbatim.

CPU tier: every `argtypes` list the document declares has exactly as many entries as the prototype in
include/cave_hip.h has parameters (a missing argument would shift every later pointer by one slot).
GPU tier: `project_hip` exactly as the document writes it, against the reference's own outputs."""

import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _doc_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 1. "):text.index("## 2. ")]
    blocks = re.findall(r"^```python\n(.*?)^```", sec, flags=re.S | re.M)
    sec6 = text[text.index("## 6. "):]
    blocks += re.findall(r"^```python\n(.*?)^```", sec6, flags=re.S | re.M)[:1]   # the fused-step stub (the loop after it is prose)
    return blocks


def _header_param_counts():
    hdr = open(os.path.join(ROOT, "include", "cave_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    out = {}
    for name, params in re.findall(r"\b(cave_hip_\w+)\s*\(([^)]*)\)\s*;", hdr):
        params = params.strip()
        out[name] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def _exec_doc():
    from cave_amd import _lib

    _lib.build()
    os.environ["CAVE_HIP_LIB"] = _lib.LIB_PATH
    ns = {}
    blocks = _doc_blocks()
    assert len(blocks) >= 2, "INTEGRATION.md §1 must hold the dense and the large-cone stub"
    for b in blocks:
        exec(compile(b, "INTEGRATION.md", "exec"), ns)  # noqa: S102 - the document is the test subject
    return ns


def test_doc_stubs_match_header_prototypes():
    ns = _exec_doc()
    counts = _header_param_counts()
    assert counts["cave_hip_cone_dense"] == 20 and counts["cave_hip_cone_dense_large"] == 22 and counts["cave_hip_cone_step"] == 24
    declared = 0
    lib = ns["_lib"]
    for name, n in counts.items():
        fn = getattr(lib, name)
        if fn.argtypes is None:
            continue
        declared += 1
        assert len(fn.argtypes) == n, f"INTEGRATION.md declares {len(fn.argtypes)} arguments for {name}, header has {n}"
    assert declared >= 5 and lib.cave_hip_cone_step.argtypes is not None
    # the calls written in the document pass as many arguments as they declare
    src = re.sub(r"#[^\n]*", "", "\n".join(_doc_blocks()))
    for name in ("cave_hip_cone_dense", "cave_hip_cone_dense_large", "cave_hip_large_slice_bytes", "cave_hip_cone_step"):
        for m in re.finditer(rf"_lib\.{name}\(", src):
            depth, i, args = 1, m.end(), 1
            while depth:
                ch = src[i]
                depth += ch in "(["
                depth -= ch in ")]"
                args += (ch == "," and depth == 1)
                i += 1
            assert args == counts[name], (name, args, counts[name])


def test_package_binding_matches_header_prototypes():
    """The same check for the binding the package itself uses (cave_amd/_lib.py)."""
    from cave_amd import _lib

    _lib.build()
    lib = _lib.load_library()
    for name, n in _header_param_counts().items():
        fn = getattr(lib, name)
        if fn.argtypes is not None:
            assert len(fn.argtypes) == n, (name, len(fn.argtypes), n)


@pytest.mark.gpu
def test_doc_project_hip_matches_reference_outputs(golden):
    import torch

    ns = _exec_doc()
    g = golden["generic"]
    for tag in ("generic", "setup"):
        ctrs, costs = g[f"{tag}_ctrs"], g[f"{tag}_costs"]
        for sense, sign in (("min", -1.0), ("max", 1.0)):
            ok = g[f"{tag}_{sense}_consistent"]
            proj, rnorm = ns["project_hip"](torch.tensor(ctrs, device="cuda"), torch.tensor(sign * costs, device="cuda"))
            torch.cuda.synchronize()
            sc = np.maximum(1.0, np.abs(costs).max(axis=1))[:, None]
            assert np.all((np.abs(proj.cpu().numpy() - g[f"{tag}_{sense}_proj"]) <= 2e-6 * sc)[ok])
            rn = g[f"{tag}_{sense}_rnorm"]
            assert np.all((np.abs(rnorm.cpu().numpy() - rn) <= 2e-6 * np.maximum(1.0, rn))[ok])
    # the tiers the document describes: > 64 reduced rows end on the large-cone entry point
    from oracle import cave_oracle as O

    rng = np.random.default_rng(5)
    A = rng.standard_normal((3, 80, 70)).astype(np.float32)
    y = rng.standard_normal((3, 70)).astype(np.float32)
    proj, rnorm = ns["project_hip"](torch.tensor(A, device="cuda"), torch.tensor(y, device="cuda"))
    po, ro = O.batch_project(y, A)
    assert np.abs(proj.cpu().numpy() - po).max() <= 4e-6 * max(1.0, np.abs(y).max())
    assert np.abs(rnorm.cpu().numpy() - ro).max() <= 4e-6 * max(1.0, ro.max())


@pytest.mark.gpu
def test_doc_cone_step_matches_reference_outputs(golden):
    """§6's `cone_step_hip` as written: pack-only launch, then launches that solve one store and pack the other -- CaVE+
    losses and gradients of the TSP-20 and SP 5x5 fixtures against the reference's own outputs."""
    import torch

    ns = _exec_doc()
    g = golden["structured"]
    for tag in ("tsp20", "sp5"):
        ctrs = torch.tensor(g[f"{tag}_ctrs"], device="cuda")
        costs = torch.tensor(g[f"{tag}_costs"], device="cuda")
        B, m, d = ctrs.shape
        assert ns["_lib"].cave_hip_step_lds_bytes(m, d) > 0
        (A, keepA), (Bs, keepB) = ns["lite_store"](B, d, ctrs.device), ns["lite_store"](B, d, ctrs.device)
        ns["cone_step_hip"](None, None, ctrs, A)                       # pack only
        for solve, nxt in ((A, Bs), (Bs, A), (A, None)):                # fused, fused, solve only
            loss, grad, status = ns["cone_step_hip"](solve, costs, ctrs if nxt is not None else None, nxt)
            torch.cuda.synchronize()
            assert bool((status == 0).all())
            assert np.abs(loss.cpu().numpy() - g[f"{tag}_min_inner_loss"]).max() <= 2e-6
            gs = max(1.0, float(np.abs(g[f"{tag}_min_inner_grad"]).max()))
            assert np.abs(grad.cpu().numpy() - g[f"{tag}_min_inner_grad"]).max() <= 8e-6 * gs
