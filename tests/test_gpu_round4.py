"""GPU tier, round-4 additions: the fused step kernel (ABI v9 cave_hip_cone_step) against the reference fixtures, the
device-resident store's lite slots, the non-blocking lazy status check, and the training example's --graph / --prefetch
modes."""

import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from golden_cases import MODE_AVG, MODE_EXACT, MODE_HEURISTIC, MODE_INNER, MODE_PROJECT, check_case

ALL = ("proj", "rnorm", "target", "loss", "grad")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _np(o):
    return {k: v.detach().cpu().numpy() for k, v in o.items()}


def test_step_kernel_matches_the_reference_fixtures(golden):
    """tests/golden/structured.npz (outputs of the reference itself: src/cave.py _batch_project / exact / inner / heuristic
    targets, losses, gradients; SP 5x5 and TSP-20) through the fused step: pack-only launch, then solve launches that
    also pack the same cones again into the other store -- both halves of the kernel are exercised by every call."""
    import torch

    from cave_amd.qpsolver import PreparedCones, cone_op_prepared, prepare_dense

    def impl(ctrs, costs, mode, sign, inner_ratio):
        c = torch.tensor(ctrs, device="cuda")
        p = None if costs is None else torch.tensor(costs, device="cuda")
        prep = prepare_dense(c)
        assert isinstance(prep, PreparedCones)
        prep.then(c)
        o = cone_op_prepared(prep, p if p is not None else torch.zeros(c.shape[0], c.shape[2], device="cuda"), mode, sign,
                             inner_ratio, outputs=ALL if mode != MODE_AVG else ("target",))
        o2 = cone_op_prepared(prep.next, p if p is not None else torch.zeros(c.shape[0], c.shape[2], device="cuda"), mode,
                              sign, inner_ratio, outputs=ALL if mode != MODE_AVG else ("target",))
        for k in o:   # the store packed beside a solve holds the same cones as the one packed alone
            assert torch.equal(o[k], o2[k]) or (torch.isnan(o[k]) == torch.isnan(o2[k])).all(), k
        return _np(o)

    check_case(impl, golden, "structured", "sp5")
    check_case(impl, golden, "structured", "tsp20")


def test_step_kernel_edge_cases():
    """Empty cones (all-zero blocks: proj = y, rnorm = 0, src/cave.py:304-305), padded rows, zero predictions, solve and
    pack halves of different sizes, a batch of one, unchecked calls on a never-packed store."""
    import torch

    from cave_amd import qpsolver, synth
    from cave_amd.qpsolver import PreparedCones, cone_op_dense, cone_op_prepared, prepare_dense

    ctrs, costs, _ = synth.tsp_batch(12, 40, seed=3)
    ctrs[5] = 0.0                     # empty cone
    ctrs[6, 24:] = 0.0                # a cone cut short: the 12 degree equalities (+a / -a pairs) only
    costs[7] = 0.0                    # zero prediction
    m, d = ctrs.shape[1:]
    qpsolver.forget_shape(m, d)
    c, p = torch.tensor(ctrs, device="cuda"), torch.tensor(costs, device="cuda")
    for mode in (MODE_PROJECT, MODE_EXACT, MODE_INNER, MODE_HEURISTIC):
        ref = cone_op_dense(c, p, mode, -1.0, 0.2, outputs=ALL, waves=2)
        prep = prepare_dense(c[:33])            # 33 instances packed alone ...
        assert isinstance(prep, PreparedCones)
        got = cone_op_prepared(prep.then(c[33:]), p[:33], mode, -1.0, 0.2, outputs=ALL)   # ... 7 beside their solve
        got2 = cone_op_prepared(prep.next, p[33:], mode, -1.0, 0.2, outputs=ALL)
        keys = ("proj", "rnorm") if mode == MODE_PROJECT else ("target", "loss", "grad") if mode == MODE_HEURISTIC else ALL
        for k in keys:
            both = torch.cat([got[k], got2[k]])
            assert float((both - ref[k]).abs().max()) <= 1e-6 * max(1.0, float(ref[k].abs().max())), (mode, k)
    # ten unpaired >= rows (more than the eight bound rows the one-wave solver takes): the checked call falls back to the
    # general operator, the shape is remembered as not qualifying until forgotten
    odd = ctrs.copy()
    odd[6, 10:] = 0.0
    co = torch.tensor(odd, device="cuda")
    got = cone_op_prepared(prepare_dense(co), p, MODE_PROJECT, -1.0, 0.0, outputs=("proj", "rnorm"))
    ref = cone_op_dense(co, p, MODE_PROJECT, -1.0, 0.0, outputs=("proj", "rnorm"), waves=2)
    assert bool((got["status"] == 0).all()) and float((got["proj"] - ref["proj"]).abs().max()) <= 1e-6
    assert prepare_dense(co) is co
    qpsolver.forget_shape(m, d)
    one = cone_op_prepared(prepare_dense(c[5:6]), p[5:6], MODE_PROJECT, 1.0, 0.0, outputs=("proj", "rnorm"))
    assert torch.equal(one["proj"], p[5:6]) and float(one["rnorm"]) == 0.0   # empty cone: the input comes back
    # a store that was never packed: every slot reports TOO_LARGE (state 0), nothing is computed from stale memory
    ss = qpsolver._LiteSlots(c.device, 8, d)
    out = {"proj": torch.zeros(8, d, device="cuda")}
    st, it = torch.empty(8, dtype=torch.int32, device="cuda"), torch.empty(8, dtype=torch.int32, device="cuda")
    qpsolver._launch_step(ss, p[:8].contiguous(), 8, MODE_PROJECT, 1.0, 0.0, 0, out, st, it, None, None)
    assert bool((st == 2).all()) and bool(torch.isnan(out["proj"]).all())


def test_step_kernel_full_batch_is_deterministic_and_certified():
    """BASELINE configs[1] at its full per-GPU batch (1024 distinct TSP-20 cones) through the fused chain: three passes
    give the same bits (the per-CU SIMD claims change which wave of a block solves, never what it computes), equal the
    general operator to 1e-6, and a sample of projections carries KKT certificates."""
    import torch

    from certificate import assert_projection
    from cave_amd import synth
    from cave_amd.qpsolver import cone_op_dense, cone_op_prepared, prepare_dense

    ctrs, costs, _ = synth.tsp_batch(20, 2048, seed=11)
    rng = np.random.default_rng(2)
    pred = costs + rng.normal(0, 0.05, costs.shape).astype(np.float32)
    A, Bc = torch.tensor(ctrs[:1024], device="cuda"), torch.tensor(ctrs[1024:], device="cuda")
    pa, pb = torch.tensor(pred[:1024], device="cuda"), torch.tensor(pred[1024:], device="cuda")
    runs = []
    for rep in range(3):
        prep = prepare_dense(A)
        oa = cone_op_prepared(prep.then(Bc), pa, MODE_INNER, -1.0, 0.2, outputs=ALL)
        ob = cone_op_prepared(prep.next.then(A), pb, MODE_INNER, -1.0, 0.2, outputs=ALL)
        runs.append((oa, ob))
        assert bool((oa["status"] == 0).all()) and bool((ob["status"] == 0).all())
        assert int(oa["iters"].max()) <= 12
    for k in ALL + ("iters",):
        for r in runs[1:]:
            assert torch.equal(runs[0][0][k], r[0][k]) and torch.equal(runs[0][1][k], r[1][k]), k
    ref = cone_op_dense(A, pa, MODE_INNER, -1.0, 0.2, outputs=ALL)
    for k in ALL:
        assert float((runs[0][0][k] - ref[k]).abs().max()) <= 1e-6 * max(1.0, float(ref[k].abs().max())), k
    proj = runs[0][1]["proj"].cpu().numpy()
    for i in range(0, 1024, 64):
        assert_projection(ctrs[1024 + i], -pred[1024 + i], proj[i], what=f"step kernel, instance {i}")


def test_packed_store_serves_small_cones_from_lite_slots(golden):
    """ConeStore: every qualifying instance gets its lite slot once (cave_hip_lite_from_packed); batches of ids then run
    the solve half of the step kernel.  Against the reference fixtures, against the general packed kernel (store.waves
    pins it), with repeated and permuted ids, and a store with a cone that does not qualify keeps the general kernel."""
    import torch

    from cave_amd import synth
    from cave_amd.dataset import ConeStore

    g = golden["structured"]
    for tag in ("sp5", "tsp20"):
        store = ConeStore.from_dense(torch.tensor(g[f"{tag}_ctrs"]))
        assert store.lite_slots is not None
        n = store.n

        def impl(ctrs, costs, mode, sign, inner_ratio):
            ids = torch.arange(n, device="cuda")
            p = None if costs is None else torch.tensor(costs, device="cuda")
            return _np(store.cone_op(ids, p, mode, sign, inner_ratio, outputs=ALL if mode != MODE_AVG else ("target",)))

        check_case(impl, golden, "structured", tag)
    ctrs, costs, _ = synth.tsp_batch(20, 300, seed=4)
    store = ConeStore.from_dense(torch.tensor(ctrs))
    assert store.lite_slots is not None
    rng = np.random.default_rng(0)
    ids = torch.tensor(rng.integers(0, 300, size=500), device="cuda")
    p = torch.tensor(costs[ids.cpu().numpy()] + rng.normal(0, 0.05, (500, costs.shape[1])).astype(np.float32), device="cuda")
    a = store.cone_op(ids, p, MODE_INNER, -1.0, 0.2, outputs=ALL)
    store.waves = 1   # the general one-wave packed kernel
    b = store.cone_op(ids, p, MODE_INNER, -1.0, 0.2, outputs=ALL)
    store.waves = 0
    assert torch.equal(a["iters"], b["iters"])
    for k in ALL:
        assert float((a[k] - b[k]).abs().max()) <= 1e-6 * max(1.0, float(b[k].abs().max())), k
    bad = store.cone_op(torch.tensor([0, 300, -1], device="cuda"), p[:3], MODE_PROJECT, 1.0, 0.0, check=False)
    assert bad["status"].tolist() == [0, 3, 3]   # slots out of range: BAD_INPUT, as the general kernel reports them
    # TSP-23 (d = 253): the solve-only launch needs 4 workgroups' worth of LDS per CU, not the 6 a launch with a pack
    # half is held to -- its store is still served from lite slots
    c23, y23, _ = synth.tsp_batch(23, 64, seed=9)
    s23 = ConeStore.from_dense(torch.tensor(c23))
    assert s23.lite_slots is not None
    i23 = torch.arange(64, device="cuda")
    a = s23.cone_op(i23, torch.tensor(y23, device="cuda"), MODE_INNER, -1.0, 0.2, outputs=ALL)
    s23.waves = 2
    b = s23.cone_op(i23, torch.tensor(y23, device="cuda"), MODE_INNER, -1.0, 0.2, outputs=ALL)
    assert bool((a["status"] == 0).all())
    for k in ALL:
        assert float((a[k] - b[k]).abs().max()) <= 1e-6 * max(1.0, float(b[k].abs().max())), k
    big = ctrs[:8].copy()
    big[3, :40, :30] = rng.standard_normal((40, 30)).astype(np.float32)
    gen = ConeStore.from_dense(torch.tensor(big))
    assert gen.lite_slots is None
    o = gen.cone_op(torch.arange(8, device="cuda"), torch.tensor(costs[:8], device="cuda"), MODE_PROJECT, 1.0, 0.0)
    assert bool((o["status"] == 0).all())


def test_lazy_check_does_not_wait_for_the_previous_launch():
    """check='lazy': a loss call examines the verdicts that HAVE arrived and never waits for the previous launch (round 3
    did: one step of run-ahead at most).  Behind a long-running kernel several launches stay pending; flush_checks()
    drains them; beyond LAZY_MAX_PENDING the oldest is awaited."""
    import torch

    from cave_amd import cave as cave_mod
    from cave_amd import synth
    from cave_amd.cave import EPO, flush_checks, innerConeAlignedCosine
    from cave_amd.dataset import ConeStore, PackedBatch

    class _M:
        modelSense = EPO.MINIMIZE

    ctrs, costs, _ = synth.tsp_batch(12, 32, seed=1)
    store = ConeStore.from_dense(torch.tensor(ctrs, device="cuda"))
    batch = PackedBatch(store, torch.arange(32, device="cuda"))
    lazy = innerConeAlignedCosine(_M(), solver="hip", seed=0, solver_kwargs={"check": "lazy"})
    p = torch.tensor(costs, device="cuda")
    flush_checks()
    big = torch.randn(8192, 8192, device="cuda")
    for _ in range(6):
        _ = big @ big                 # ~10 ms of GPU work ahead of the loss launches
    for _ in range(3):
        lazy(p, batch)
    assert 1 <= len(cave_mod._pending_checks) <= 3      # nobody waited for the matmuls
    for _ in range(cave_mod.LAZY_MAX_PENDING + 3):
        lazy(p, batch)
    assert len(cave_mod._pending_checks) <= cave_mod.LAZY_MAX_PENDING + 1
    flush_checks()
    assert not cave_mod._pending_checks


def _example(argv):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import train_sp_cave

    return train_sp_cave.main(argv)


def test_training_example_graph_mode_matches_eager():
    """examples/train_sp_cave.py --graph: predictor + loss + backward + Adam of a full batch replayed as ONE HIP graph;
    the loss curve and the regret equal those of the eager run (same batches, same initial weights, Adam state reset in
    place after the capture's warm-up steps)."""
    common = ["--grid", "5", "5", "--num-data", "96", "--batch", "32", "--epochs", "4", "--packed"]
    eager = _example(common)
    graph = _example(common + ["--graph"])
    assert len(eager) == len(graph) == 5
    for (e0, l0, r0), (e1, l1, r1) in zip(eager[1:], graph[1:]):
        assert abs(l0 - l1) <= 2e-5 * max(1.0, abs(l0)), (e0, l0, l1)
        assert abs(r0 - r1) <= 1e-3, (e0, r0, r1)
    assert graph[-1][2] < graph[0][2]


def test_training_example_prefetch_mode_matches_plain_loop():
    """--prefetch (dense TSP cones): cave_amd.dataset.prefetch around the DataLoader, loop body unchanged -- the same loss
    curve as the plain loop (a ragged last batch included)."""
    common = ["--problem", "tsp", "--nodes", "9", "--num-data", "72", "--batch", "32", "--epochs", "3"]
    plain = _example(common)
    pre = _example(common + ["--prefetch"])
    for (e0, l0, r0), (e1, l1, r1) in zip(plain[1:], pre[1:]):
        assert abs(l0 - l1) <= 2e-5 * max(1.0, abs(l0)), (e0, l0, l1)
        assert abs(r0 - r1) <= 1e-3, (e0, r0, r1)


def test_red_black_cache_of_the_packed_store():
    """ABI v10 rb_cache: grid shortest-path cones on the large path keep, per instance, what the red-black reduction of
    their band systems derives from the static cone (cone_rb.h).  The first projection builds it, later ones reuse it:
    same bits either way and without a cache; repeated ids in one batch (two workgroups building the same entry) too;
    projections carry KKT certificates."""
    import torch

    from certificate import assert_projection
    from cave_amd import synth
    from cave_amd.dataset import ConeStore

    c, y, _ = synth.sp_batch(24, 24, 6, seed=21)
    rng = np.random.default_rng(3)
    pred = (y + rng.normal(0, 0.05, y.shape)).astype(np.float32)
    store = ConeStore.from_dense(torch.tensor(c, device="cuda"), chunk=3)
    assert store.large and store.rb_cache is not None and int(store.rb_cache.max()) == 0
    ids = torch.tensor([0, 1, 2, 3, 4, 5, 2, 2, 0], device="cuda")
    p = torch.tensor(pred[ids.cpu().numpy()], device="cuda")
    first = store.cone_op(ids, p, MODE_PROJECT, -1.0, 0.0, outputs=("proj", "rnorm"))
    stride = store._c.rb_stride
    states = store.rb_cache.view(-1, stride)[:, :4].contiguous().view(torch.int32).flatten().tolist()
    assert states == [1] * 6   # every instance built its entry (state word: 1 = built)
    again = store.cone_op(ids, p, MODE_PROJECT, -1.0, 0.0, outputs=("proj", "rnorm"))
    plain = ConeStore.from_dense(torch.tensor(c, device="cuda"), chunk=3)
    plain._c.rb_cache, plain._c.rb_stride, plain.rb_cache = None, 0, None
    ref = plain.cone_op(ids, p, MODE_PROJECT, -1.0, 0.0, outputs=("proj", "rnorm"))
    for k in ("proj", "rnorm", "iters"):
        assert torch.equal(first[k], again[k]) and torch.equal(first[k], ref[k]), k
    proj = first["proj"].cpu().numpy()
    for j, i in enumerate(ids.tolist()):
        assert_projection(c[i], -pred[i], proj[j], what=f"24x24 grid, instance {i}")
