"""CPU oracle for the CaVE cone-projection hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the cave_amd package never does (and must fail loudly without its
HIP extension rather than fall back to anything here).

It restates, with numpy float32/float64 and the plain-C Lawson-Hanson NNLS of
oracle/nnls_oracle.c, what the reference computes on its solver='nnls' path
(paths relative to /root/reference):

    project_nnls / batch_project   src/cave.py:231-264, 298-309
    average_ctrs                   src/cave.py:222-228
    exact_target                   src/cave.py:121-129
    inner_target                   src/cave.py:197-219
    cone_loss (forward)            src/cave.py:55-73
    cone_loss_grad                 autograd of src/cave.py:68-73 (torch F.cosine_similarity)

Pinning: tests/golden/*.npz hold outputs of the reference itself (imported from
/root/reference with a stand-in for the absent PyEPO base class, see
tests/golden/make_golden.py); tests/test_oracle.py checks this module against
every one of them.  `reduction` and the unseeded branch RNG come from PyEPO's
optModule, which is not in /root/reference: those two are parity-unpinned
w.r.t. PyEPO and are restated from the README's description.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile oracle/nnls_oracle.c -> oracle/liboracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "nnls_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-std=c11", "-shared", src, "-o", so, "-lm"], check=True)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.cave_oracle_project_nnls.restype = C.c_int
        _LIB.cave_oracle_batch_project.restype = C.c_int
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ projection

def project_nnls(cp: np.ndarray, ctr: np.ndarray):
    """`_project_nnls(cp, ctr, _)` -> (proj float32 (d,), rnorm float).  src/cave.py:298-309"""
    cp = np.ascontiguousarray(cp, dtype=np.float32)
    ctr = np.ascontiguousarray(ctr, dtype=np.float32).reshape(-1, cp.shape[0])
    m, d = ctr.shape
    proj = np.empty(d, np.float32)
    rn = C.c_double(0.0)
    it = C.c_int(0)
    rc = _lib().cave_oracle_project_nnls(_fp(ctr), C.c_int(m), C.c_int(d), _fp(cp), _fp(proj), C.byref(rn), C.byref(it))
    if rc != 0:
        raise RuntimeError("Maximum number of iterations reached.")  # what SciPy raises (src/cave.py:307)
    return proj, float(rn.value)


def batch_project(signed_cost: np.ndarray, tight_ctrs: np.ndarray):
    """`_batch_project(..., solver='nnls', processes=1)` -> (proj (B,d) f32, rnorm (B,) f32).  src/cave.py:231-264"""
    cps = np.ascontiguousarray(signed_cost, dtype=np.float32)
    ctrs = np.ascontiguousarray(tight_ctrs, dtype=np.float32)
    B, m, d = ctrs.shape
    proj = np.empty((B, d), np.float32)
    rnorm = np.empty(B, np.float32)
    status = np.zeros(B, np.int32)
    rc = _lib().cave_oracle_batch_project(_fp(ctrs), _fp(cps), C.c_int64(B), C.c_int(m), C.c_int(d), _fp(proj),
                                          _fp(rnorm), _fp(status))
    if rc != 0:
        raise RuntimeError("Maximum number of iterations reached.")
    return proj, rnorm


def average_ctrs(tight_ctrs: np.ndarray) -> np.ndarray:
    """`_average_ctrs` in float32, line by line.  src/cave.py:222-228"""
    t = np.asarray(tight_ctrs, dtype=np.float32)
    norms = np.sqrt((t.astype(np.float64) ** 2).sum(axis=2, keepdims=True)).astype(np.float32)
    valid = (norms > np.float32(1e-7)).astype(np.float32)
    unit = t / np.maximum(norms, np.float32(1e-8)) * valid
    n_valid = np.maximum(valid.sum(axis=1), np.float32(1.0))
    return (unit.sum(axis=1) / n_valid).astype(np.float32)


# ---------------------------------------------------------------- loss algebra

def _rownorm(x):
    return np.sqrt((x.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(np.float32)


def exact_target(signed_cost, tight_ctrs):
    """exactConeAlignedCosine._get_projection: proj / clamp(||proj||, 1e-8).  src/cave.py:121-129"""
    proj, rnorm = batch_project(signed_cost, tight_ctrs)
    return proj / np.maximum(_rownorm(proj), np.float32(1e-8)), proj, rnorm


def heuristic_target(signed_cost, tight_ctrs, inner_ratio):
    """heuristic branch.  src/cave.py:202-204"""
    s = np.asarray(signed_cost, dtype=np.float32)
    pred_norm = s / np.maximum(_rownorm(s), np.float32(1e-8))
    avg = average_ctrs(tight_ctrs)
    r = np.float32(inner_ratio)
    return (np.float32(1) - r) * pred_norm + r * avg


def inner_target(signed_cost, tight_ctrs, inner_ratio):
    """QP branch for solver='nnls' (push inside unless rnorm < 1e-7).  src/cave.py:206-219"""
    proj, rnorm = batch_project(signed_cost, tight_ctrs)
    proj_norm = proj / np.maximum(_rownorm(proj), np.float32(1e-8))
    avg = average_ctrs(tight_ctrs)
    r = np.float32(inner_ratio)
    pushed = (np.float32(1) - r) * proj_norm + r * avg
    inside = (rnorm < np.float32(1e-7))[:, None]
    return np.where(inside, proj_norm, pushed), proj, rnorm


def cosine_similarity(x, t, eps=1e-8):
    """torch F.cosine_similarity(x, t, dim=1): sum (x/max(|x|,eps)) * (t/max(|t|,eps)), in float64."""
    x = np.asarray(x, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    nx = np.maximum(np.sqrt((x * x).sum(axis=1)), eps)
    nt = np.maximum(np.sqrt((t * t).sum(axis=1)), eps)
    return (x * t).sum(axis=1) / (nx * nt)


def cone_loss(pred_cost, target, sign):
    """per-instance loss 1 - cos(sign*pred, target).  src/cave.py:68-72"""
    s = np.float32(sign) * np.asarray(pred_cost, dtype=np.float32)
    return (1.0 - cosine_similarity(s, target)).astype(np.float32)


def cone_loss_grad(pred_cost, target, sign, eps=1e-8):
    """d loss_i / d pred_i with the target held constant (it is computed under no_grad, src/cave.py:70-71)."""
    s = np.float64(sign) * np.asarray(pred_cost, dtype=np.float64)
    t = np.asarray(target, dtype=np.float64)
    ns = np.sqrt((s * s).sum(axis=1, keepdims=True))
    nt = np.sqrt((t * t).sum(axis=1, keepdims=True))
    ns_c = np.maximum(ns, eps)
    nt_c = np.maximum(nt, eps)
    st = (s * t).sum(axis=1, keepdims=True)
    dcos = t / (ns_c * nt_c)
    radial = np.where(ns > eps, st / (ns_c * ns_c * np.maximum(ns, 1e-300) * nt_c), 0.0)
    dcos = dcos - radial * s
    return (-np.float64(sign) * dcos).astype(np.float32)


def reduce(loss, reduction):
    """PyEPO optModule._reduce as documented in the reference README (README.md:75,90)."""
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    if reduction == "none":
        return loss
    raise ValueError(f"No reduction '{reduction}'.")


class ConeLossOracle:
    """exactConeAlignedCosine / innerConeAlignedCosine(solver='nnls') on numpy arrays."""

    def __init__(self, minimize=True, inner=False, solve_ratio=1.0, inner_ratio=0.2, reduction="mean", seed=None):
        self.sign = -1.0 if minimize else 1.0
        self.inner = inner
        self.solve_ratio = float(solve_ratio)
        self.inner_ratio = float(inner_ratio)
        self.reduction = reduction
        self.rng = np.random.RandomState(seed) if seed is not None else np.random.RandomState()

    def target(self, pred_cost, tight_ctrs):
        signed = np.float32(self.sign) * np.asarray(pred_cost, dtype=np.float32)
        if not self.inner:
            return exact_target(signed, tight_ctrs)[0]
        if self.rng.uniform() > self.solve_ratio:  # one draw per forward, src/cave.py:201
            return heuristic_target(signed, tight_ctrs, self.inner_ratio)
        return inner_target(signed, tight_ctrs, self.inner_ratio)[0]

    def __call__(self, pred_cost, tight_ctrs):
        t = self.target(pred_cost, tight_ctrs)
        loss = cone_loss(pred_cost, t, self.sign)
        grad = cone_loss_grad(pred_cost, t, self.sign)
        B = len(loss)
        if self.reduction == "mean":
            grad = grad / np.float32(B)
        return reduce(loss, self.reduction), grad
