"""CPU tier: the GPU-ONLY code paths under a SIMT emulation (tests/emul/simt_abi.cpp + tests/emul/simt/hip/hip_runtime.h).

The one-wave "lite" Newton solver (cone_core.h), the one-/two-wave band elimination (cone_band.h), the blocked dense
LDL^T with the paused factorisation (cone_dense.h) and the wave / workgroup contexts under them are written with DPP,
readlane and ballot primitives and were, until round 3, compiled by hipcc only: the serial single-lane build of
tests/emul could not see them, and a race between two waves (half bandwidth 3, round 2) was found late, on the GPU.
Here the same sources are compiled by g++ against a shim in which every lane is a fiber and every cross-lane primitive
or barrier is a rendezvous; between two rendezvous the lanes run one after another, in round-robin or in a seeded
shuffled order.  A hand-over through LDS that the source does not order therefore computes wrong numbers here (two such
places in the band elimination, harmless under the hardware's lockstep, were found this way and are now marked
CAVE_WAVE_ORDER()).  The shuffled runs and the AddressSanitizer / UBSan build are what `test_asan_ubsan_clean` of the
serial build is for the rest of the code.  TEST INFRASTRUCTURE: nothing in cave_amd loads these builds."""

import os
import subprocess
import sys

import numpy as np
import pytest

from emul_lib import Emul, Simt, store_bandwidth
from golden_cases import MODE_INNER, MODE_PROJECT
from oracle import cave_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emul():
    return Emul()


@pytest.fixture(scope="module")
def simt():
    return Simt()


def _banded_inequality_cones(B, m, width, shift, seed, pairs=0):
    """Cones whose reduced system is a narrow band with INEQUALITY rows (rows held at their bound by the active-set
    loop): dense rows of `width` consecutive entries, each shifted by `shift` (same generator as tests/test_gpu_round2)."""
    rng = np.random.default_rng(seed)
    d = shift * (m - 1) + width
    A = np.zeros((B, m + pairs, d), np.float32)
    for b in range(B):
        for i in range(m):
            A[b, i, shift * i: shift * i + width] = rng.standard_normal(width).astype(np.float32)
        for j in range(pairs):
            A[b, m + j] = -A[b, (7 * j + 3) % m]
    y = rng.standard_normal((B, d)).astype(np.float32)
    return A, y


def test_lite_solver_under_emulation(emul, simt, golden):
    """cone_packed_kernel<WaveCtx> / <BlockCtx<2>>: the one-wave lite solver (ELL gathers, prefix-sum gradient, lite
    Hessian, batched-readlane Gauss-Jordan) on the TSP-20 fixture against the REFERENCE's outputs, one and two waves,
    round-robin and shuffled lane order; the path counter proves the lite form ran."""
    g = golden["structured"]
    ctrs, costs = g["tsp20_ctrs"][:6], g["tsp20_costs"][:6]
    st, arrs, mr, mn = emul.pack(ctrs)
    ids = np.arange(len(ctrs))
    sc = float(np.abs(costs).max())
    simt.path_counters()
    for waves, seed in ((1, 0), (1, 17), (2, 0), (2, 5)):
        o = simt.cone_packed(st, arrs, mr, mn, ids, costs, MODE_PROJECT, sign=-1.0, waves=waves, seed=seed)
        assert (o["status"] == 0).all(), (waves, seed)
        assert np.abs(o["proj"] - g["tsp20_min_proj"][:6]).max() <= 2e-6 * sc, (waves, seed)
        assert np.abs(o["rnorm"] - g["tsp20_min_rnorm"][:6]).max() <= 2e-6 * sc, (waves, seed)
    assert simt.path_counters()[2] == 4 * len(ctrs)
    # the four-wave shape carries the general solver (no lite form): same numbers
    o4 = simt.cone_packed(st, arrs, mr, mn, ids, costs, MODE_PROJECT, sign=-1.0, waves=4)
    assert np.abs(o4["proj"] - g["tsp20_min_proj"][:6]).max() <= 2e-6 * sc and simt.path_counters()[2] == 0
    # loss / gradient through the fused epilogue (CaVE+), vs the serial build of the same code
    a = simt.cone_packed(st, arrs, mr, mn, ids, costs, MODE_INNER, sign=-1.0, waves=1, seed=3)
    b = emul.cone_packed(st, arrs, mr, mn, ids, costs, MODE_INNER, sign=-1.0)
    for k in ("loss", "grad", "target"):
        assert np.abs(a[k] - b[k]).max() <= 2e-6, k


def test_lite_solver_regression_fixtures_under_emulation(emul, simt, golden):
    """The cones on which the round-1 GPU fuzzer once hit the iteration cap and the tiny-norm predictions
    (tests/golden/regress.npz, reference outputs), through the dense operator at one wave (lite solver where the cone
    qualifies, the general one-wave solver otherwise)."""
    from golden_cases import check_regress

    n = [0]

    def impl(A, y, mode, sign, inner_ratio):
        n[0] += 1
        return simt.cone_dense(A, y, mode, sign=sign, inner_ratio=inner_ratio, waves=1, seed=n[0] % 3)

    check_regress(impl, golden["regress"])


@pytest.mark.parametrize("waves", [1, 2, 4])
def test_band_elimination_under_emulation(emul, simt, golden, waves):
    """solve_spd_band_wave: grid shortest-path cones (12x12: 144 free rows, half bandwidth 12) against the
    reference's outputs, on the 1-wave / 2-wave (wave 0 eliminates, wave 1 admits rows) / 4-wave shapes, round-robin
    and shuffled."""
    from cave_amd import synth

    g = golden["large"]
    c, y, _ = synth.sp_batch(12, 12, 4, seed=0)
    st, arrs, mr, _ = emul.pack_large(c[:2])
    bw = store_bandwidth(arrs, 2, c.shape[2])
    assert (mr, bw) == (144, 12)
    sc = max(1.0, float(np.abs(y).max()))
    simt.path_counters()
    for seed in (0, 23):
        o = simt.cone_packed_large(st, arrs, mr, bw, np.arange(2), y[:2], MODE_PROJECT, sign=-1.0, waves=waves, seed=seed)
        assert (o["status"] == 0).all() and o["iters"].max() <= 12, (waves, seed, o["iters"])
        assert np.abs(o["proj"] - g["sp12_proj"][:2]).max() <= 2e-6 * sc, (waves, seed)
        assert np.abs(o["rnorm"] - g["sp12_rnorm"][:2]).max() <= 2e-6 * sc, (waves, seed)
    assert simt.path_counters()[1] == 4  # the one-wave band form ran for every instance


@pytest.mark.parametrize("side", [24, 20])
def test_red_black_reduction_of_the_band_under_emulation(emul, simt, side):
    """cone_rb.h: grids of 24 x 24 (576 free rows, half bandwidth 24; 288 of them stay in the band system) and 20 x 20 (the
    black half, 200 rows, is smaller than the operand scratch the solver was sized for: 208) on the two-wave shape -- the
    adjacency, the greedy independent set, the recipes, the producer building rows of the Schur complement, the closed-form
    red rows -- against the serial build of the same source (which has no one-wave band form and therefore factors the
    full band), round-robin and shuffled schedules."""
    from cave_amd import synth

    c, y, _ = synth.sp_batch(side, side, 1, seed=5)
    st, arrs, mr, _ = emul.pack_large(c)
    bw = store_bandwidth(arrs, 1, c.shape[2])
    assert (mr, bw) == (side * side, side)
    ref = emul.cone_packed_large(st, arrs, mr, np.arange(1), y, MODE_PROJECT, sign=-1.0)
    assert (ref["status"] == 0).all()
    # what ConeStore._fold_signs does when it finalises a store of this path: the signs of an all-+-1 instance go into
    # bit 15 of its stored indices (flags bit 1) -- the reduction takes such instances only
    assert (arrs["flags"] & 1).all()
    for idx, val in (("ccol", "cval"), ("cvar", "cvalc")):
        arrs[idx] |= (arrs[val] < 0).astype(np.uint16) << 15
    arrs["flags"] |= 2
    sc = max(1.0, float(np.abs(y).max()))
    simt.path_counters()
    for seed in (0, 31):
        o = simt.cone_packed_large(st, arrs, mr, bw, np.arange(1), y, MODE_PROJECT, sign=-1.0, waves=2, seed=seed)
        assert (o["status"] == 0).all() and o["iters"].max() <= 14, (seed, o["iters"])
        assert np.abs(o["proj"] - ref["proj"]).max() <= 2e-6 * sc, seed
        assert np.abs(o["rnorm"] - ref["rnorm"]).max() <= 2e-6 * sc, seed
    assert simt.path_counters()[5] == 2  # both runs took the reduction


def test_diet_layout_of_the_packed_operator_under_emulation(emul, simt, golden):
    """cone_packed_kernel<BlockCtx<4, true>> with the "diet" LDS layout (round 4: TSP-50 at two workgroups per compute
    unit -- H as a packed lower triangle, CSC entries and average normal read in place from the store, batched column
    prefetch in the Hessian update) on the TSP-50 fixture against the REFERENCE's outputs, round-robin and shuffled lane
    order, and against the ordinary layout of the same shape (same iteration counts)."""
    from cave_amd import synth

    g = golden["tsp50"]
    c, y, _ = synth.tsp_batch(int(g["n"]), int(g["batch"]), seed=int(g["seed"]))
    st, arrs, mr, mn = emul.pack_large(c)
    assert mr > 32 and bool((arrs["flags"] & 1).all())
    # the signs into bit 15 of the indices + flags bit 1 (what ConeStore._fold_signs does on the device)
    for idx, val in (("ccol", "cval"), ("cvar", "cvalc")):
        arrs[idx] |= ((arrs[val] < 0).astype(np.uint16) << 15).astype(arrs[idx].dtype)
    arrs["flags"] |= 2
    ids = np.arange(len(c))
    plain = simt.cone_packed(st, arrs, mr, mn, ids, -y, MODE_PROJECT, sign=1.0, waves=8)
    assert (plain["status"] == 0).all() and np.abs(plain["proj"] - g["proj"]).max() <= 4e-6
    for seed in (0, 31):
        o = simt.cone_packed(st, arrs, mr, mn, ids, -y, MODE_PROJECT, sign=1.0, waves=8, seed=seed, diet=True)
        assert (o["status"] == 0).all(), seed
        assert np.abs(o["proj"] - g["proj"]).max() <= 4e-6 and np.abs(o["rnorm"] - g["rnorm"]).max() <= 4e-6, seed
        assert np.array_equal(o["iters"], plain["iters"]) and np.abs(o["proj"] - plain["proj"]).max() <= 1e-7, seed
    oi = simt.cone_packed(st, arrs, mr, mn, ids, -y, MODE_INNER, sign=1.0, waves=8, diet=True)   # the average normal, in place
    pi = simt.cone_packed(st, arrs, mr, mn, ids, -y, MODE_INNER, sign=1.0, waves=8)
    for k in ("loss", "grad", "target"):
        assert np.abs(oi[k] - pi[k]).max() <= 1e-7, k


def test_band_hand_over_failure_is_reported(emul):
    """cone_band.h kBandSpinLimit (VERDICT r3 / ADVICE r3): the producer / eliminator waves of the band elimination meet
    through LDS words and every wait is bounded; a wait that runs out used to fall through silently -- the wave then
    computed on rows the other one had not delivered and the instance still reported CAVE_ST_OK.  A variant build
    withholds ONE "rows built" announcement of the producer and shrinks the limit: the eliminator's wait runs out, every
    wave still reaches its exit, and the instance reports CAVE_ST_NOT_CONVERGED with NaN outputs.  The same build with
    the announcement in place (one-wave shape: no hand-over) stays correct."""
    from emul_lib import Simt
    from cave_amd import synth

    c, y, _ = synth.sp_batch(12, 12, 2, seed=0)
    st, arrs, mr, _ = emul.pack_large(c)
    bw = store_bandwidth(arrs, 2, c.shape[2])
    broken = Simt(defines=("CAVE_BAND_SPIN_LIMIT=4000", "CAVE_TEST_WITHHOLD_FLAG=2"), tag="_withhold")
    broken.path_counters()
    o = broken.cone_packed_large(st, arrs, mr, bw, np.arange(2), y, MODE_PROJECT, sign=-1.0, waves=2)
    assert broken.path_counters()[1] == 2                 # the two-wave band form ran
    assert (o["status"] == 1).all(), o["status"]           # CAVE_ST_NOT_CONVERGED, not a silent CAVE_ST_OK
    assert np.isnan(o["proj"]).all() and np.isnan(o["rnorm"]).all()
    ok = broken.cone_packed_large(st, arrs, mr, bw, np.arange(2), y, MODE_PROJECT, sign=-1.0, waves=1)
    assert (ok["status"] == 0).all()                       # no producer wave, nothing withheld


@pytest.mark.parametrize("m,width,pairs", [(70, 5, 0), (90, 8, 5), (60, 14, 0)])
def test_band_elimination_with_bound_rows_under_emulation(emul, simt, m, width, pairs):
    """Banded INEQUALITY cones (half bandwidths 4 / 7 / 13): rows are held at their bound by the active-set loop, so
    identity rows travel through the blocked elimination, the row admission and the ring back substitution; two waves
    (the shape in which round 2's bw = 3 overlap lived) and one, shuffled lane order; vs the oracle."""
    A, y = _banded_inequality_cones(2, m, width, 1, seed=m + width, pairs=pairs)
    po, ro = O.batch_project(y, A)
    st, arrs, mr, _ = emul.pack_large(A)
    bw = store_bandwidth(arrs, 2, A.shape[2])
    assert bw == width - 1 and mr > bw + 1
    tol = 4e-6 * max(1.0, float(np.abs(y).max()))
    simt.path_counters()
    for waves, seed in ((2, 0), (2, 31), (1, 7)):
        o = simt.cone_packed_large(st, arrs, mr, bw, np.arange(2), y, MODE_PROJECT, sign=1.0, waves=waves, seed=seed)
        assert (o["status"] == 0).all(), (waves, seed)
        assert np.abs(o["proj"] - po).max() <= tol and np.abs(o["rnorm"] - ro).max() <= tol, (waves, seed)
    assert simt.path_counters()[1] == 6


@pytest.mark.parametrize("waves", [4, 2, 1])
def test_dense_ldl_with_paused_factorisation_under_emulation(emul, simt, waves):
    """cone_dense.h as the GPU runs it: fixed-point Hessian with LDS integer atomics, four pivots per step on wave 0
    + trailing update by all waves, the Schur system of the cut rows solved by the register Gauss-Jordan, column-
    oriented back substitution -- TSP-40 cones (40 free degree rows + up to 5 cut rows with theta >= 0) vs the oracle."""
    from cave_amd import synth

    c, y, _ = synth.tsp_batch(40, 3, seed=11)
    po, ro = O.batch_project(-y, c)
    st, arrs, mr, _ = emul.pack_large(c)
    bw = store_bandwidth(arrs, 3, c.shape[2])
    sc = max(1.0, float(np.abs(y).max()))
    simt.path_counters()
    outs = []
    for seed in (0, 41):
        o = simt.cone_packed_large(st, arrs, mr, bw, np.arange(3), y, MODE_PROJECT, sign=-1.0, waves=waves, seed=seed)
        assert (o["status"] == 0).all() and o["iters"].max() <= 12, (waves, seed)
        assert np.abs(o["proj"] - po).max() <= 2e-6 * sc and np.abs(o["rnorm"] - ro).max() <= 2e-6 * sc, (waves, seed)
        outs.append(o)
    assert simt.path_counters()[0] == 6
    # fixed-point accumulation: the order in which the lanes arrive does not change a bit of the result
    assert np.array_equal(outs[0]["proj"], outs[1]["proj"]) and np.array_equal(outs[0]["iters"], outs[1]["iters"])


def test_simt_build_is_asan_ubsan_clean():
    """The emulated GPU code under AddressSanitizer + UBSan: the LDS arena and the workspace slice are exact-size heap
    blocks, so an index past a window, a ring or a scratch row is reported here (lite solver, both band forms with
    bound rows, dense LDL^T)."""
    import emul_lib

    so = emul_lib.build_simt(asan=True)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    code = (
        "import sys; sys.path[:0]=[%r,%r]\n"
        "import numpy as np, ctypes as C, emul_lib\n"
        "from cave_amd import synth\n"
        "E = emul_lib.Emul()\n"
        "S = emul_lib.Simt.__new__(emul_lib.Simt); S.lib = C.CDLL(%r); S.lib.cave_simt_packed_large_slice_bytes.restype = C.c_int64\n"
        "c,y,_ = synth.tsp_batch(20, 2, 1)\n"
        "st,arrs,mr,mz = E.pack(c)\n"
        "for w in (1, 2, 4): S.cone_packed(st,arrs,mr,mz,np.arange(2),y,2,waves=w,seed=w)\n"
        "S.cone_dense(c, y, 2, waves=1); S.cone_dense(c, y, 0, waves=4, seed=2)\n"
        "c2,y2,_ = synth.sp_batch(9, 9, 2, 1)\n"
        "st,arrs,mr,mz = E.pack_large(c2); bw = emul_lib.store_bandwidth(arrs, 2, c2.shape[2])\n"
        "for w in (1, 2, 4): S.cone_packed_large(st,arrs,mr,bw,np.arange(2),y2,2,waves=w,seed=5*w)\n"
        "sys.path.insert(0, %r)\n"
        "from test_simt_emul import _banded_inequality_cones\n"
        "A,yb = _banded_inequality_cones(1, 60, 6, 1, 3, pairs=4)\n"
        "st,arrs,mr,mz = E.pack_large(A); bw = emul_lib.store_bandwidth(arrs, 1, A.shape[2])\n"
        "for w in (2, 1): S.cone_packed_large(st,arrs,mr,bw,np.arange(1),yb,0,sign=1.0,waves=w,seed=w)\n"
        "c3,y3,_ = synth.tsp_batch(36, 1, 2)\n"
        "st,arrs,mr,mz = E.pack_large(c3); bw = emul_lib.store_bandwidth(arrs, 1, c3.shape[2])\n"
        "for w in (4, 2): S.cone_packed_large(st,arrs,mr,bw,np.arange(1),y3,2,waves=w,seed=w)\n"
        "print('asan-ok', S.path_counters())\n" % (ROOT, os.path.join(ROOT, "tests"), so, os.path.join(ROOT, "tests")))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "asan-ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
