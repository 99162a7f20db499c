"""CPU tier: the per-instance code shared with the HIP kernels (cave_amd/csrc/cone_core.h,
cone_instance.h) compiled for one serial lane with gcc (tests/emul), checked against the
reference's committed outputs and the oracle.  This validates the algorithm and host logic
without a GPU; the parity tests proper are the -m gpu tests, which go through the C ABI."""

import os
import subprocess
import sys

import numpy as np
import pytest

from emul_lib import Emul
from golden_cases import CASES, MODE_EXACT, MODE_INNER, MODE_PROJECT, check_case, check_regress
from oracle import cave_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emul():
    return Emul()


def emul_impl(E):
    def f(ctrs, costs, mode, sign, inner_ratio):
        return E.cone_dense(ctrs, costs, mode, sign=sign, inner_ratio=inner_ratio)
    return f


@pytest.mark.parametrize("file,tag", CASES)
def test_serial_kernel_code_matches_reference_outputs(emul, golden, file, tag):
    check_case(emul_impl(emul), golden, file, tag)


def test_serial_kernel_code_regression_fixtures(emul, golden):
    """Former iteration-cap instances and tiny-norm predictions (tests/golden/regress.npz)."""
    check_regress(emul_impl(emul), golden["regress"])


def test_packed_store_equals_dense(emul, golden):
    g = golden["structured"]
    for tag in ("sp5", "tsp20"):
        ctrs, costs = g[f"{tag}_ctrs"], g[f"{tag}_costs"]
        st, arrs, mr, mz = emul.pack(ctrs)
        ids = np.arange(len(ctrs))[::-1].copy()
        for mode in (MODE_PROJECT, MODE_EXACT, MODE_INNER):
            a = emul.cone_dense(ctrs[ids], costs[ids], mode)
            b = emul.cone_packed(st, arrs, mr, mz, ids, costs[ids], mode)
            for k in (("proj", "rnorm") if mode == MODE_PROJECT else ("proj", "rnorm", "target", "loss", "grad")):
                assert np.array_equal(a[k], b[k]), (tag, mode, k)
        # equality rows are paired: TSP-n has n free multipliers + cuts, SP h*w has h*w
        n_free = arrs["vkind"].sum()
        assert n_free == (25 * len(ctrs) if tag == "sp5" else 20 * len(ctrs))


def test_edge_cases(emul):
    # empty cone -> proj = y, rnorm = 0, exact loss 0 (src/cave.py:304-305)
    y = np.array([[1, -2, 3, 0.5]], np.float32)
    o = emul.cone_dense(np.zeros((1, 3, 4), np.float32), y, MODE_EXACT, sign=1.0)
    assert np.array_equal(o["proj"], y) and o["rnorm"][0] == 0 and abs(o["loss"][0]) < 1e-7
    # m_max == 0
    o = emul.cone_dense(np.zeros((2, 0, 4), np.float32), np.ones((2, 4), np.float32), MODE_PROJECT, sign=1.0)
    assert np.array_equal(o["proj"], np.ones((2, 4), np.float32)) and (o["status"] == 0).all()
    # zero prediction -> zero projection, loss exactly 1 (cosine eps), finite gradient
    A = np.random.default_rng(0).random((2, 3, 6)).astype(np.float32)
    o = emul.cone_dense(A, np.zeros((2, 6), np.float32), MODE_EXACT, sign=-1.0)
    assert np.all(o["proj"] == 0) and np.allclose(o["loss"], 1.0) and np.isfinite(o["grad"]).all()
    # whole-space cone (+e_k and -e_k for every k): projection is the identity
    I = np.eye(4, dtype=np.float32)
    o = emul.cone_dense(np.concatenate([I, -I])[None], y, MODE_PROJECT, sign=1.0)
    assert np.array_equal(o["proj"], y) and o["rnorm"][0] == 0
    # non-negative orthant with scaled unit rows
    o = emul.cone_dense((I * np.array([1, 2, .5, 3], np.float32))[None], y, MODE_PROJECT, sign=1.0)
    assert np.array_equal(o["proj"], np.maximum(y, 0)) and abs(o["rnorm"][0] - 2.0) < 1e-6
    # NaN input is reported, not silently propagated
    bad = y.copy(); bad[0, 1] = np.nan
    o = emul.cone_dense(np.ones((1, 2, 4), np.float32), bad, MODE_PROJECT, sign=1.0)
    assert o["status"][0] == 3 and np.isnan(o["rnorm"][0])
    # a cone that cannot fit the arena is reported as such
    big = np.random.default_rng(1).standard_normal((1, 80, 70)).astype(np.float32)
    o = emul.cone_dense(big, np.ones((1, 70), np.float32), MODE_PROJECT, sign=1.0, nnz_cap=80 * 70, lds_bytes=160 * 1024)
    assert o["status"][0] == 2


def test_random_cones_vs_oracle(emul):
    rng = np.random.default_rng(11)
    for trial in range(60):
        d, m, B = int(rng.integers(1, 20)), int(rng.integers(0, 36)), 6
        A = rng.standard_normal((B, m, d)).astype(np.float32)
        if trial % 3 == 1:
            A *= rng.random((B, m, d)) < 0.35
        if trial % 3 == 2 and m > 4:
            A[:, m // 2:] = 0
            A[:, 1] = -A[:, 0]
        y = rng.standard_normal((B, d)).astype(np.float32)
        o = emul.cone_dense(A, y, MODE_PROJECT, sign=1.0, nnz_cap=max(m * d, 64), lds_bytes=160 * 1024)
        po, ro = O.batch_project(y, A)
        assert (o["status"] == 0).all()
        sc = np.maximum(1.0, np.abs(y).max(axis=1))[:, None]
        assert np.all(np.abs(o["proj"] - po) <= 4e-6 * sc)
        assert np.all(np.abs(o["rnorm"] - ro) <= 4e-6 * np.maximum(1.0, ro))


def test_tsp50_serial(emul, golden):
    from cave_amd import synth

    g = golden["tsp50"]
    c, y, _ = synth.tsp_batch(int(g["n"]), int(g["batch"]), seed=int(g["seed"]))
    # 320 KiB arena: beyond a real workgroup's LDS, algorithm check only (see emul_abi.cpp)
    o = emul.cone_dense(c, -y, MODE_PROJECT, sign=1.0, nnz_cap=16000, lds_bytes=320 * 1024)
    assert (o["status"] == 0).all()
    assert np.abs(o["proj"] - g["proj"]).max() <= 4e-6 and np.abs(o["rnorm"] - g["rnorm"]).max() <= 4e-6


def test_asan_ubsan_clean():
    """Run the serial build under AddressSanitizer/UBSan (arena overruns show up here)."""
    import emul_lib

    so = emul_lib.build(asan=True)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    code = (
        "import sys; sys.path[:0]=[%r,%r]\n"
        "import numpy as np, ctypes as C, emul_lib\n"
        "from cave_amd import synth\n"
        "E = emul_lib.Emul.__new__(emul_lib.Emul); E.lib = C.CDLL(%r)\n"
        "c,y,_ = synth.tsp_batch(12, 4, 1)\n"
        "for mode in range(5): E.cone_dense(c, y, mode)\n"
        "st,arrs,mr,mz = E.pack(c); E.cone_packed(st,arrs,mr,mz,np.arange(4),y,2)\n"
        "r = np.random.default_rng(0); A = r.standard_normal((3,9,5)).astype(np.float32)\n"
        "E.cone_dense(A, r.standard_normal((3,5)).astype(np.float32), 2)\n"
        "E.cone_dense(A, r.standard_normal((3,5)).astype(np.float32), 0, nnz_cap=8)\n"  # overflow path
        "for mode in range(5): E.cone_dense_large(c, y, mode)\n"  # large-cone path: exact-size workspace slice
        "E.cone_dense_large(A, r.standard_normal((3,5)).astype(np.float32), 2, lds_bytes=1024)\n"
        "E.cone_dense_large(c, y, 0, slice_bytes=20000)\n"  # workspace too small -> TOO_LARGE, no overrun
        "c2,y2,_ = synth.sp_batch(9, 9, 2, 1)\n"
        "st,arrs,mr,mz = E.pack_large(c2); E.cone_packed_large(st,arrs,mr,np.arange(2),y2,2)\n"
        "print('asan-ok')\n" % (ROOT, os.path.join(ROOT, "tests"), so))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "asan-ok" in r.stdout, r.stderr[-3000:]


def test_band_solver_matches_dense_solve(emul):
    """cone_band.h: LDL^T band elimination == dense solve of the masked system (identity rows for
    fixed unknowns, H + reg*I on the free block), over random bandwidths, masks and sizes."""
    import ctypes as C
    E = emul
    rng = np.random.default_rng(5)
    for trial in range(200):
        p = int(rng.integers(1, 70))
        bw = int(rng.integers(0, p)) if trial % 3 else p - 1
        ld = bw + 1
        G = rng.standard_normal((p, p + 3))
        H = G @ G.T
        for i in range(p):
            for j in range(p):
                if abs(i - j) > bw:
                    H[i, j] = 0.0
        H += np.eye(p) * (np.abs(H).sum(1).max() if bw < p - 1 else 0.0)  # keep the truncated band SPD
        act = (rng.random(p) < 0.25).astype(np.uint8)
        rhs = rng.standard_normal(p)
        Hb = np.zeros((p, ld))
        for j in range(p):
            for t in range(ld):
                if j + t < p:
                    Hb[j, t] = H[j + t, j]
        x = np.zeros(p)
        reg_rel = 1e-10
        rc = E.lib.cave_emul_band_solve(Hb.ctypes.data_as(C.c_void_p), C.c_int32(bw), rhs.ctypes.data_as(C.c_void_p),
                                        act.ctypes.data_as(C.c_void_p), C.c_int32(p), C.c_double(reg_rel),
                                        x.ctypes.data_as(C.c_void_p))
        assert rc == 0
        free = act == 0
        Md = np.eye(p)
        reg = reg_rel * (H.diagonal()[free].max() if free.any() else 0.0)
        Md[free] = H[free]
        Md[free, free] += reg
        # fixed unknowns: x = rhs; free rows: H_FF x_F + H_FA x_A = rhs_F
        want = np.linalg.solve(Md, rhs)
        assert np.allclose(x, want, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(want).max())), (trial, p, bw)


def large_impl(E):
    def f(ctrs, costs, mode, sign, inner_ratio):
        return E.cone_dense_large(ctrs, costs, mode, sign=sign, inner_ratio=inner_ratio)
    return f


@pytest.mark.parametrize("file,tag", CASES)
def test_large_path_code_matches_reference_outputs(emul, golden, file, tag):
    """The large-cone path (global-workspace arena, band Newton systems, smoothed Hessian) must give the
    reference's outputs on the same fixtures as the fast path."""
    check_case(large_impl(emul), golden, file, tag)


def test_large_path_packed_equals_dense(emul):
    from cave_amd import synth
    ctrs, costs, _ = synth.sp_batch(9, 9, 6, seed=4)   # 81 reduced rows: beyond the register solver
    dense = emul.cone_dense_large(ctrs, costs, MODE_INNER, sign=-1.0)
    st, arrs, mr, mz = emul.pack_large(ctrs)
    assert mr == 81
    packed = emul.cone_packed_large(st, arrs, mr, np.arange(6)[::-1].copy(), costs[::-1].copy(), MODE_INNER, sign=-1.0)
    assert (dense["status"] == 0).all() and (packed["status"] == 0).all()
    for k in ("loss", "grad", "target"):
        assert np.allclose(dense[k][::-1], packed[k], rtol=0, atol=2e-6), k
    po, ro = O.batch_project(-costs, ctrs)
    pr = emul.cone_dense_large(ctrs, costs, MODE_PROJECT, sign=-1.0)
    assert np.abs(pr["proj"] - po).max() < 2e-6 and np.abs(pr["rnorm"] - ro).max() < 2e-6
    assert int(pr["iters"].max()) <= 15
