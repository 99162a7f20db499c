"""GPU tier, round-2 additions: regression fixtures over every launch shape, the packed store directly against
the reference's outputs, BASELINE configs 2-4 at their full per-GPU batch sizes on DISTINCT cones with KKT
certificates, determinism, and the shape caches of the host layer under mixed cone sizes."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from golden_cases import (CASES, MODE_EXACT, MODE_INNER, MODE_PROJECT, check_case, check_regress)

ALL = ("proj", "rnorm", "target", "loss", "grad")


def _dense_impl(**kw):
    import torch

    from cave_amd.qpsolver import cone_op_dense

    def impl(ctrs, costs, mode, sign, inner_ratio):
        c = torch.tensor(np.asarray(ctrs), device="cuda")
        p = None if costs is None else torch.tensor(np.asarray(costs), device="cuda")
        o = cone_op_dense(c, p, mode, sign, inner_ratio, outputs=ALL, **kw)
        return {k: v.cpu().numpy() for k, v in o.items()}
    return impl


def _packed_impl(waves=0, large=False):
    import torch

    from cave_amd.dataset import ConeStore

    cache = {}

    def impl(ctrs, costs, mode, sign, inner_ratio):
        ctrs = np.asarray(ctrs)
        key = (ctrs.shape, ctrs.tobytes()[:4096], float(ctrs.sum()))
        if key not in cache:
            cache.clear()
            cache[key] = ConeStore.from_dense(torch.tensor(ctrs, device="cuda"), chunk=7)
        store = cache[key]
        store.waves = waves
        ids = torch.arange(len(ctrs), device="cuda")
        p = None if costs is None else torch.tensor(np.asarray(costs), device="cuda")
        o = store.cone_op(ids, p, mode, sign, inner_ratio, outputs=ALL)
        return {k: v.cpu().numpy() for k, v in o.items()}
    return impl


@pytest.mark.parametrize("waves", [1, 2, 4, 8])
def test_regression_fixtures_dense(golden, waves):
    check_regress(_dense_impl(waves=waves), golden["regress"])


@pytest.mark.parametrize("waves", [1, 2, 4, 8])
def test_regression_fixtures_packed(golden, waves):
    check_regress(_packed_impl(waves=waves), golden["regress"])


def test_regression_fixtures_large_path(golden):
    from test_gpu_parity import _force_large

    base = _dense_impl()
    check_regress(lambda c, y, *a: _force_large(lambda: base(c, y, *a), np.asarray(c)), golden["regress"])


@pytest.mark.parametrize("waves", [0, 1, 4])
@pytest.mark.parametrize("file,tag", CASES)
def test_packed_store_matches_reference_outputs(golden, file, tag, waves):
    """ConeStore -> cave_hip_cone_packed directly against the reference's outputs (not via the dense path)."""
    check_case(_packed_impl(waves=waves), golden, file, tag)


def test_two_launches_are_bit_identical():
    """Fixed summation order everywhere: two launches on the same inputs agree bit for bit on every instance
    that reports CAVE_ST_OK, in every launch shape, dense and packed."""
    import torch

    from cave_amd import synth
    from cave_amd.dataset import ConeStore
    from cave_amd.qpsolver import cone_op_dense

    # generic cones with GENERAL (non +-1) entries: the 2- / 4- / 8-wave shapes accumulated their Hessian with
    # floating-point LDS atomics until round 3 (exact for +-1 cones only: every addend a multiple of 1/16); since round 4
    # in 64-bit fixed point -- the same bits whichever wave's add arrives first
    rng = np.random.default_rng(77)
    gen_c = (rng.standard_normal((256, 40, 24)) * (rng.random((256, 40, 24)) < 0.35)).astype(np.float32)
    gen_c[:, 28:] = 0.0
    gen_c[:, 20:28] = -gen_c[:, :8]            # +a / -a pairs: free multipliers among the bounded ones
    gen_y = rng.standard_normal((256, 24)).astype(np.float32)
    big_c = np.zeros((64, 40, 48), np.float32)                      # dense rows: 30 bounded multipliers, 1 440 entries
    big_c[:, :30] = rng.standard_normal((64, 30, 48)).astype(np.float32)
    big_y = rng.standard_normal((64, 48)).astype(np.float32)
    for ctrs, costs in (synth.tsp_batch(20, 300, seed=5)[:2], synth.sp_batch(5, 5, 200, seed=5)[:2], (gen_c, gen_y),
                        (big_c, big_y)):
        c, p = torch.tensor(ctrs, device="cuda"), torch.tensor(costs, device="cuda")
        store = ConeStore.from_dense(c)
        ids = torch.arange(len(ctrs), device="cuda")
        for waves in (1, 2, 4, 8):
            runs = []
            for rep in range(3):
                runs.append(cone_op_dense(c, p, MODE_INNER, -1.0, 0.2, waves=waves, check=False, outputs=ALL,
                                          nnz_cap=int(c.shape[1] * c.shape[2]) if c.shape[2] < 100 else 0,
                                          lds_bytes=64 * 1024 if c.shape[2] < 100 else 0))
                _ = torch.randn(2048, 2048, device="cuda") @ torch.randn(2048, 2048, device="cuda")  # other work in between
            assert torch.equal(runs[0]["status"], runs[2]["status"]) and torch.equal(runs[0]["iters"], runs[2]["iters"])
            for k in ALL:
                ok2 = (runs[0]["status"] == 0)
                assert torch.equal(runs[0][k][ok2], runs[2][k][ok2]), (waves, k, "third launch")
            ok = (runs[0]["status"] == 0) & (runs[1]["status"] == 0)
            assert torch.equal(runs[0]["status"], runs[1]["status"]) and bool(ok.any())
            for k in ALL:
                assert torch.equal(runs[0][k][ok], runs[1][k][ok]), (waves, k)
            assert torch.equal(runs[0]["iters"], runs[1]["iters"])
            store.waves = waves
            pr = [store.cone_op(ids, p, MODE_INNER, -1.0, 0.2, check=False, outputs=ALL) for _ in range(2)]
            for k in ALL:
                assert torch.equal(pr[0][k], pr[1][k]), ("packed", waves, k)


def _full_batch(kind, size, B, mode, chunk, n_cert, iters_max):
    import torch

    from certificate import assert_projection, coo_to_sparse
    from cave_amd import synth
    from cave_amd.dataset import ConeStore

    dev = torch.device("cuda")
    items, costs, _ = synth.coo_batch(kind, size, B, seed=11)
    d = costs.shape[1]
    m_max = max(it[3] for it in items)
    store = ConeStore.from_chunks_lazy(lambda i: synth.densify_on(items[i:i + chunk], d, dev, m_max), list(range(0, B, chunk)))
    assert store.n == B
    ids = torch.arange(B, device=dev)
    pred = torch.tensor(costs, device=dev)
    o = store.cone_op(ids, pred, MODE_PROJECT, -1.0, 0.0, outputs=("proj", "rnorm"))
    assert bool((o["status"] == 0).all())
    it = o["iters"].cpu().numpy()
    assert it.max() <= iters_max and it.min() >= 1, (it.min(), it.max())
    proj, rnorm = o["proj"].cpu().numpy(), o["rnorm"].cpu().numpy()
    y = -costs
    res = np.linalg.norm(y.astype(np.float64) - proj, axis=1)
    assert np.abs(res - rnorm).max() <= 4e-6 * max(1.0, res.max())                     # rnorm is the residual norm
    assert np.abs(((y - proj).astype(np.float64) * proj).sum(1)).max() <= 2e-5 * (np.linalg.norm(y, axis=1) ** 2).max()
    for b in np.linspace(0, B - 1, n_cert).astype(int):                               # KKT certificate on a sample
        assert_projection(coo_to_sparse(items[b], d), y[b], proj[b], what=(kind, size, int(b)))
    # the loss modes at the same size: finite, in [0, 2], gradient orthogonal to the prediction (cosine loss)
    o2 = store.cone_op(ids, pred, mode, -1.0, 0.2, outputs=("loss", "grad"))
    loss, grad = o2["loss"].cpu().numpy(), o2["grad"].cpu().numpy()
    assert np.isfinite(loss).all() and loss.min() >= -1e-6 and loss.max() <= 2.0 + 1e-6
    assert np.abs((grad.astype(np.float64) * costs).sum(1)).max() <= 1e-5
    return store


def test_config2_tsp50_full_batch_distinct_cones():
    """BASELINE configs[2]: TSP-50, per-GPU batch 4096/8 = 512, CaVE Exact."""
    store = _full_batch("tsp", 50, 512, MODE_EXACT, 64, 16, 30)
    assert not store.large


def test_config3_tsp100_full_batch_distinct_cones():
    """BASELINE configs[3]: TSP-100, per-GPU batch 2048/4 = 512, QP branch of CaVE Hybrid."""
    store = _full_batch("tsp", 100, 512, MODE_INNER, 4, 16, 30)
    assert store.large


def test_config4_sp30_full_batch_distinct_cones():
    """BASELINE configs[4]: shortest path 30x30, per-GPU batch 8192/8 = 1024, CaVE+."""
    store = _full_batch("sp", (30, 30), 1024, MODE_INNER, 32, 16, 30)
    assert store.large


def test_lazy_check_with_mixed_cone_sizes_under_one_shape():
    """The launch-shape caches are keyed by (m_max, d) only.  A shape that settled on four waves (<= 32
    reduced rows) and is later called lazily with bigger cones must not poison training: the failed instances
    contribute zero loss / gradient, the next call raises, the shape is forgotten, and the call after that
    (strict again) re-tiers and succeeds."""
    import torch

    from cave_amd import qpsolver
    from cave_amd.cave import EPO, exactConeAlignedCosine, flush_checks
    from cave_amd.qpsolver import HipSolverError
    from oracle import cave_oracle as O

    class _M:
        modelSense = EPO.MAXIMIZE

    rng = np.random.default_rng(21)
    m, d, B = 48, 30, 8

    def cones(rows):
        A = np.zeros((B, m, d), np.float32)
        A[:, :rows] = (rng.standard_normal((B, rows, d)) * (rng.random((B, rows, d)) < 0.25)).astype(np.float32)
        A[:, :rows, 0] = 1.0
        A[:, :rows, 1] = rng.standard_normal((B, rows))
        return A

    small, big = cones(20), cones(44)
    y = rng.standard_normal((B, d)).astype(np.float32)
    qpsolver.forget_shape(m, d)
    qpsolver._split_ok[(m, d)] = False  # the fused kernel's shape cache is what is under test here
    mod = exactConeAlignedCosine(_M(), solver="hip", solver_kwargs={"check": "lazy"}, reduction="none")
    p = torch.tensor(y, device="cuda", requires_grad=True)
    l0 = mod(p, torch.tensor(small, device="cuda"))          # first call for the shape: strict, settles on 4 waves
    assert qpsolver._wide_ok.get((m, d)) is True and (m, d) in qpsolver._settled
    want = O.cone_loss(y, O.exact_target(y, small)[0], 1.0)
    assert np.abs(l0.detach().cpu().numpy() - want).max() <= 2e-6
    l1 = mod(p, torch.tensor(big, device="cuda"))            # lazy launch with the cached 4-wave shape: does not fit
    l1.sum().backward()
    assert torch.isfinite(l1).all() and torch.isfinite(p.grad).all()   # masked, not NaN
    torch.cuda.synchronize()
    with pytest.raises(HipSolverError):
        mod(p, torch.tensor(small, device="cuda"))           # the verdict arrives here ...
    assert (m, d) not in qpsolver._settled and (m, d) not in qpsolver._wide_ok  # ... and the shape is forgotten
    qpsolver._split_ok[(m, d)] = False
    flush_checks()
    l2 = mod(p, torch.tensor(big, device="cuda"))            # strict again: falls back to a shape that fits
    want = O.cone_loss(y, O.exact_target(y, big)[0], 1.0)
    assert np.abs(l2.detach().cpu().numpy() - want).max() <= 2e-6
    flush_checks()


def test_split_form_of_the_dense_operator(golden):
    """Small cones: cone_op_dense runs 'pack into transient slots + one-wave packed solve' (qpsolver.launch_split).
    Same results as the reference fixtures, bit-identical to the packed store, and a batch with an instance
    beyond the slot capacity falls back to the fused kernel."""
    import torch

    from cave_amd import qpsolver, synth
    from cave_amd.dataset import ConeStore
    from cave_amd.qpsolver import cone_op_dense

    for file, tag in CASES:
        g = golden[file]
        qpsolver.forget_shape(*g[f"{tag}_ctrs"].shape[1:])
        check_case(_dense_impl(), golden, file, tag)
    ctrs, costs, _ = synth.tsp_batch(20, 200, seed=9)
    key = (ctrs.shape[1], ctrs.shape[2])
    qpsolver.forget_shape(*key)
    c, p = torch.tensor(ctrs, device="cuda"), torch.tensor(costs, device="cuda")
    a = cone_op_dense(c, p, MODE_INNER, -1.0, 0.2, outputs=ALL)
    assert qpsolver._split_ok.get(key) is True
    store = ConeStore.from_dense(c)
    b = store.cone_op(torch.arange(200, device="cuda"), p, MODE_INNER, -1.0, 0.2, outputs=ALL)
    for k in ALL:
        assert torch.equal(a[k], b[k]), k
    a2 = cone_op_dense(c, p, MODE_INNER, -1.0, 0.2, outputs=ALL, check=False)  # unchecked: uses what was learnt
    for k in ALL:
        assert torch.equal(a[k], a2[k]), k
    # one dense row block turns an instance into a non +-1 cone with 40 general rows: beyond the slots
    rng = np.random.default_rng(0)
    big = ctrs[:8].copy()
    big[3, :40, :30] = rng.standard_normal((40, 30)).astype(np.float32)
    qpsolver.forget_shape(*key)
    o = cone_op_dense(torch.tensor(big, device="cuda"), p[:8], MODE_PROJECT, -1.0, 0.0, outputs=("proj", "rnorm"))
    assert qpsolver._split_ok.get(key) is False and bool((o["status"] == 0).all())
    from oracle import cave_oracle as O

    po, ro = O.batch_project(-costs[:8], big)
    assert np.abs(o["proj"].cpu().numpy() - po).max() <= 4e-6 * np.abs(costs[:8]).max()
    qpsolver.forget_shape(*key)


def test_warm_start_matches_cold_start_and_saves_iterations(golden):
    """ConeStore.enable_warm_start(): projections started from the multipliers of the previous solve of the
    same instance equal cold-start projections (the projection is unique) and need fewer Newton iterations
    once the prediction has only drifted."""
    import torch

    from cave_amd.dataset import ConeStore

    g = golden["structured"]
    ctrs, costs = g["tsp20_ctrs"], g["tsp20_costs"]
    B = len(ctrs)
    c = torch.tensor(ctrs, device="cuda")
    cold, warm = ConeStore.from_dense(c), ConeStore.from_dense(c)
    warm.enable_warm_start()
    ids = torch.arange(B, device="cuda")
    rng = np.random.default_rng(4)
    sc = float(np.abs(costs).max())
    for waves in (0, 4):
        pred = torch.tensor(costs, device="cuda")
        cold.waves = warm.waves = waves
        warm.reset_warm_start()
        its = []
        for step in range(4):
            a = cold.cone_op(ids, pred, MODE_INNER, -1.0, 0.2, outputs=ALL)
            b = warm.cone_op(ids, pred, MODE_INNER, -1.0, 0.2, outputs=ALL)
            for k in ALL:
                tol = 4e-6 * (sc if k in ("proj", "rnorm") else 4.0)
                assert float((a[k] - b[k]).abs().max()) <= tol, (waves, step, k)
            its.append((float(a["iters"].float().mean()), float(b["iters"].float().mean())))
            if step == 0:  # first call: nothing cached yet, identical work; and the fixture itself
                assert torch.equal(a["iters"], b["iters"])
                assert np.abs(b["proj"].cpu().numpy() - g["tsp20_min_proj"]).max() <= 2e-6 * sc
            pred = pred + torch.tensor(rng.normal(0, 0.01, costs.shape).astype(np.float32), device="cuda")
        assert its[-1][1] <= 3.0 and its[-1][1] < its[-1][0] - 1.5, its   # <= 3 iterations once warm
    # a NaN prediction invalidates the entry instead of poisoning the cache
    bad = pred.clone()
    bad[2, 7] = float("nan")
    o = warm.cone_op(ids, bad, MODE_INNER, -1.0, 0.2, check=False, outputs=("loss",))
    assert int(o["status"][2]) == 3 and int(warm.t["warm_state"][2]) == 0 and int(warm.t["warm_state"][3]) == 1
    o = warm.cone_op(ids, pred, MODE_INNER, -1.0, 0.2, outputs=ALL)
    a = cold.cone_op(ids, pred, MODE_INNER, -1.0, 0.2, outputs=ALL)
    assert float((a["proj"] - o["proj"]).abs().max()) <= 4e-6 * sc


def test_training_harness_on_tsp_with_warm_start():
    """examples/train_sp_cave.py --problem tsp: the code_sample.py loop on DFJ TSP cones built without Gurobi
    (Held-Karp / HiGHS), packed store, warm start: regret drops and epochs >= 2 need <= 3 Newton iterations."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import train_sp_cave

    hist = train_sp_cave.main(["--problem", "tsp", "--nodes", "9", "--num-data", "64", "--batch", "32", "--epochs", "6",
                               "--packed", "--warm-start"])
    assert hist[-1][2] < hist[0][2] and hist[-1][1] < hist[1][1], hist
    log = train_sp_cave.main.iters_log
    assert all(m <= 3.0 for m, _ in log[1:]), log


def _close(a, b, tol=1e-6):
    import torch

    return bool(torch.isfinite(a).all()) and float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def test_prepared_form_fuses_pack_and_solve(golden):
    """loss_fn.prepare(bctr) / prep.then(next_bctr): the prediction-independent pack stage of the NEXT batch rides in
    the launch of this batch's solve (cave_hip_cone_step: one grid, solve blocks first).  Same numbers as the ordinary
    call, also through the modules, also after many reuses of the three lite stores, and a batch with a cone the
    one-wave solver does not take falls back."""
    import torch

    from cave_amd import qpsolver, synth
    from cave_amd.cave import EPO, innerConeAlignedCosine
    from cave_amd.qpsolver import PreparedCones, cone_op_dense, cone_op_prepared, prepare_dense

    ctrs, costs, _ = synth.tsp_batch(20, 64, seed=13)
    key = (ctrs.shape[1], ctrs.shape[2])
    qpsolver.forget_shape(*key)
    rng = np.random.default_rng(3)
    batches = [(torch.tensor(ctrs[rng.permutation(64)[:48]], device="cuda"),
                torch.tensor(costs[:48] + rng.normal(0, 0.1, (48, costs.shape[1])).astype(np.float32), device="cuda"))
               for _ in range(7)]
    want = [cone_op_dense(c, p, MODE_INNER, -1.0, 0.2, outputs=ALL) for c, p in batches]
    prep = prepare_dense(batches[0][0])
    assert isinstance(prep, PreparedCones)
    for i, (c, p) in enumerate(batches):
        prep.then(batches[(i + 1) % len(batches)][0])   # the next batch's pack rides in this batch's solve launch
        got = cone_op_prepared(prep, p, MODE_INNER, -1.0, 0.2, outputs=ALL)
        assert bool((got["status"] == 0).all()) and torch.equal(got["iters"], want[i]["iters"]), i
        for k in ALL:
            assert _close(got[k], want[i][k]), (i, k)
        prep = prep.next
        assert isinstance(prep, PreparedCones) and not prep.stale()
    # every mode of the step kernel against the general operator (PROJECT / EXACT / HEURISTIC / AVG)
    from cave_amd._lib import MODE_AVG, MODE_EXACT, MODE_HEURISTIC, MODE_PROJECT
    c, p = batches[1]
    for mode, outs in ((MODE_PROJECT, ("proj", "rnorm")), (MODE_EXACT, ALL), (MODE_HEURISTIC, ("target", "loss", "grad")),
                       (MODE_AVG, ("target",))):
        got = cone_op_prepared(prepare_dense(c), p, mode, -1.0, 0.2, outputs=outs)
        ref = cone_op_dense(c, p, mode, -1.0, 0.2, outputs=outs)
        for k in outs:
            assert _close(got[k], ref[k]), (mode, k)

    class M:
        modelSense = EPO.MINIMIZE

    mod = innerConeAlignedCosine(M(), solver="hip", seed=0)
    c, p = batches[2]
    pr = p.clone().requires_grad_(True)
    nxt = mod.prepare(c, batches[3][0])
    l1 = mod(pr, nxt)
    l1.backward()
    pr2 = p.clone().requires_grad_(True)
    l2 = mod(pr2, c)
    l2.backward()
    assert _close(l1.detach(), l2.detach()) and _close(pr.grad, pr2.grad)
    l3 = mod(batches[3][1], nxt.next)   # the batch packed beside that solve
    assert _close(l3, mod(batches[3][1], batches[3][0]))
    # a cone the lite form does not take: a dense row block makes instance 3 a non +-1 cone with 40 general rows
    big = ctrs[:8].copy()
    big[3, :40, :30] = rng.standard_normal((40, 30)).astype(np.float32)
    qpsolver.forget_shape(*key)
    bt, pt = torch.tensor(big, device="cuda"), torch.tensor(costs[:8], device="cuda")
    got = cone_op_prepared(prepare_dense(bt), pt, MODE_PROJECT, -1.0, 0.0, outputs=("proj", "rnorm"))
    ref = cone_op_dense(bt, pt, MODE_PROJECT, -1.0, 0.0, outputs=("proj", "rnorm"))
    assert bool((got["status"] == 0).all()) and torch.equal(got["proj"], ref["proj"])
    assert prepare_dense(bt) is bt  # the shape is now known not to fit: prepare declines
    qpsolver.forget_shape(*key)


def _banded_inequality_cones(B, m, width, shift, seed, pairs=0):
    """General (>=) rows whose supports slide along the cost vector: M M^T is a narrow band, the reduced rows
    carry theta >= 0 multipliers, so the active-set inner loop fixes rows (identity rows in the band solver);
    `pairs` rows are duplicated with the opposite sign (+a / -a pairs: free multipliers)."""
    rng = np.random.default_rng(seed)
    d = shift * (m - 1) + width
    A = np.zeros((B, m + pairs, d), np.float32)
    for i in range(m):
        A[:, i, i * shift:i * shift + width] = rng.standard_normal((B, width)).astype(np.float32)
    for j in range(pairs):
        A[:, m + j] = -A[:, (j * 7) % m]
    y = rng.standard_normal((B, d)).astype(np.float32)
    return A, y


@pytest.mark.gpu
@pytest.mark.parametrize("m,width,shift,pairs", [(90, 3, 1, 0), (130, 6, 2, 5), (200, 4, 1, 9), (90, 5, 1, 0), (131, 9, 2, 7), (91, 6, 1, 0), (70, 40, 3, 0), (67, 2, 1, 3)])
def test_band_solver_with_bound_rows_vs_oracle(m, width, shift, pairs):
    """The one-wave band elimination (cone_band.h) with rows held at their bound: banded cones of inequality rows
    (half bandwidths 1 .. 13 -- below 3 the team form of the solver runs --, more than 64 reduced rows, row counts not a multiple of the block of four), dense
    operator and packed store at 4-, 2- and 1-wave workgroups, against the CPU oracle."""
    import os

    import torch

    from cave_amd.qpsolver import cone_op_dense
    from cave_amd.dataset import ConeStore
    from oracle import cave_oracle as O

    A, y = _banded_inequality_cones(6, m, width, shift, seed=m + width, pairs=pairs)
    po, ro = O.batch_project(y, A)
    tol = 4e-6 * max(1.0, float(np.abs(y).max()))
    At, yt = torch.tensor(A, device="cuda"), torch.tensor(y, device="cuda")
    o = cone_op_dense(At, yt, 0, 1.0, 0.0)
    assert (o["status"].cpu().numpy() == 0).all()
    assert np.abs(o["proj"].cpu().numpy() - po).max() <= tol and np.abs(o["rnorm"].cpu().numpy() - ro).max() <= tol
    st = ConeStore.from_dense(At, chunk=6)
    assert st.large
    ids = torch.arange(6, device="cuda")
    for waves in (4, 2, 1):
        st.large_waves = waves  # ABI v8: the workgroup shape is an argument of cave_hip_cone_packed_large
        q = st.cone_op(ids, yt, 0, 1.0, outputs=("proj", "rnorm"))
        assert (q["status"].cpu().numpy() == 0).all(), waves
        assert np.abs(q["proj"].cpu().numpy() - po).max() <= tol, waves
        assert np.abs(q["rnorm"].cpu().numpy() - ro).max() <= tol, waves


def test_warm_start_on_the_large_cone_path():
    """The multiplier cache also serves cones on the large path (12x12 grid: 144 reduced rows, band solver): same
    projections as a cold start, fewer Newton iterations once the predictions only drift."""
    import torch

    from cave_amd import synth
    from cave_amd.dataset import ConeStore

    c, y, _ = synth.sp_batch(12, 12, 48, seed=4)
    st = ConeStore.from_dense(torch.tensor(c, device="cuda"), chunk=48)
    assert st.large
    ids = torch.arange(48, device="cuda")
    g = torch.Generator(device="cpu").manual_seed(0)
    p = torch.tensor(y, device="cuda") + torch.tensor(0.05 * torch.randn(y.shape, generator=g).numpy(), device="cuda")
    cold = st.cone_op(ids, p, 2, -1.0, outputs=ALL)
    st.enable_warm_start()
    st.cone_op(ids, p, 2, -1.0, outputs=ALL)
    p2 = p + torch.tensor(0.01 * torch.randn(y.shape, generator=g).numpy(), device="cuda")
    warm = st.cone_op(ids, p2, 2, -1.0, outputs=ALL)
    st.enable_warm_start(False)
    ref = st.cone_op(ids, p2, 2, -1.0, outputs=ALL)
    assert bool((warm["status"] == 0).all())
    scale = max(1.0, float(p2.abs().max()))
    for k in ("proj", "loss", "grad"):
        assert float((warm[k] - ref[k]).abs().max()) <= 4e-6 * scale, k
    assert float(warm["iters"].float().mean()) <= float(ref["iters"].float().mean()) - 1.0
    assert float(cold["iters"].float().mean()) >= float(warm["iters"].float().mean()) + 1.0
