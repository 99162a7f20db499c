"""GPU tier, round-3 additions: the review items of round 2 (warm start with repeated ids, the prepared form's stream
ordering and store generations, RCCL at world size 1) and the large-cone path work of this round (determinism,
the `waves` argument of ABI v8, interior-point inner mode on large cones)."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from golden_cases import MODE_INNER, MODE_PROJECT

ALL = ("proj", "rnorm", "target", "loss", "grad")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_warm_start_with_repeated_ids():
    """A batch may name one store slot several times while the multiplier cache is on (bench.py did: 1024 ids over
    1000 instances).  The workgroups of the repeats read and rewrite that slot's state concurrently; the decision to
    start warm is taken once per workgroup, so every wave shape must finish, and the projection (unique) must equal
    the cold one on every copy.  LDS path at 1 / 2 / 4 waves, and the large-cone path."""
    import torch

    from cave_amd import synth
    from cave_amd.dataset import ConeStore

    rng = np.random.default_rng(5)
    for kind in ("tsp20", "sp12"):
        if kind == "tsp20":
            ctrs, costs, _ = synth.tsp_batch(20, 24, seed=9)
        else:
            ctrs, costs, _ = synth.sp_batch(12, 12, 24, seed=9)
        c = torch.tensor(ctrs, device="cuda")
        cold, warm = ConeStore.from_dense(c), ConeStore.from_dense(c)
        assert cold.large == (kind == "sp12")
        warm.enable_warm_start()
        ids = torch.tensor(rng.integers(0, 6, size=96), device="cuda")  # 96 slots of the batch over 6 instances
        sc = float(np.abs(costs).max())
        for waves in ((2, 4, 1) if kind == "tsp20" else (4, 2)):
            cold.waves = warm.waves = waves
            cold.large_waves = warm.large_waves = waves
            warm.reset_warm_start()
            pred = torch.tensor(costs[ids.cpu().numpy()], device="cuda")
            for step in range(3):
                a = cold.cone_op(ids, pred, MODE_INNER, -1.0, 0.2, outputs=ALL)
                b = warm.cone_op(ids, pred, MODE_INNER, -1.0, 0.2, outputs=ALL)
                assert bool((b["status"] == 0).all())
                for k in ALL:
                    tol = 4e-6 * (sc if k in ("proj", "rnorm") else 4.0)
                    assert float((a[k] - b[k]).abs().max()) <= tol, (kind, waves, step, k)
                pred = pred + torch.tensor(rng.normal(0, 0.01, pred.shape).astype(np.float32), device="cuda")


def test_prepared_form_chain_and_beyond_the_pool():
    """The fused step (qpsolver.prepare_dense / PreparedCones.then / cone_op_prepared) behind other work on the stream,
    over more batches than the pool has lite stores (each store is recycled while its predecessor's launch is still
    in flight: one stream orders them -- ADVICE r3: the side-stream form could hand out a store a running solve was
    reading), with more prepared batches HELD than the pool has stores (a stale one falls back to its dense tensor
    instead of solving another batch's cones), and on float64 cones."""
    import torch

    from cave_amd import qpsolver, synth
    from cave_amd.qpsolver import STEP_POOL, PreparedCones, cone_op_dense, cone_op_prepared, prepare_dense

    ctrs, costs, _ = synth.tsp_batch(20, 96, seed=21)
    qpsolver.forget_shape(ctrs.shape[1], ctrs.shape[2])  # whatever earlier tests concluded about this (m_max, d)
    rng = np.random.default_rng(8)
    batches = [(torch.tensor(ctrs[rng.permutation(96)[:40]], device="cuda"),
                torch.tensor(costs[:40] + rng.normal(0, 0.1, (40, costs.shape[1])).astype(np.float32), device="cuda"))
               for _ in range(2 * STEP_POOL + 1)]
    want = [cone_op_dense(c, p, MODE_INNER, -1.0, 0.2, outputs=ALL) for c, p in batches]

    def close(a, b):
        return float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))

    qpsolver._step_pool.clear()  # every store of this run is created inside the loop below
    big = torch.randn(4096, 4096, device="cuda")
    prep = prepare_dense(batches[0][0])
    for i, (c, p) in enumerate(batches):
        _ = big @ big  # a long-running kernel ahead of the step on the same stream
        assert isinstance(prep, PreparedCones) and not prep.stale()
        if i + 1 < len(batches):
            prep.then(batches[i + 1][0])
        got = cone_op_prepared(prep, p, MODE_INNER, -1.0, 0.2, check=False, outputs=ALL)  # (no host sync in the chain)
        for k in ALL:
            assert close(got[k], want[i][k]), (i, k)
        assert bool((got["status"] == 0).all())
        prep = prep.next
    assert prep is None
    # hold STEP_POOL + 2 prepared batches before consuming any
    held = [prepare_dense(c) for c, _ in batches[:STEP_POOL + 2]]
    assert sum(h.stale() for h in held) == 2 and held[0].stale() and not held[-1].stale()
    for i, h in enumerate(held):
        got = cone_op_prepared(h, batches[i][1], MODE_INNER, -1.0, 0.2, outputs=ALL)
        for k in ALL:
            assert close(got[k], want[i][k]), ("held", i, k)
    # float64 cones: converted by the call itself
    got = cone_op_prepared(prepare_dense(batches[0][0].double()), batches[0][1], MODE_INNER, -1.0, 0.2, outputs=ALL)
    for k in ALL:
        assert close(got[k], want[0][k]), ("f64", k)


def test_prefetch_wraps_a_loader_and_keeps_the_loop_body():
    """cave_amd.dataset.prefetch(loader): the loop body of code_sample.py:48-60 unchanged (fields moved with .cuda(),
    loss = cave(cp, bctr)); losses and gradients equal those of the plain loop, every batch after the first was packed
    by its predecessor's loss call, a ragged last batch and a loop that skips a batch both work."""
    import torch

    from cave_amd import synth
    from cave_amd.cave import EPO, innerConeAlignedCosine
    from cave_amd.dataset import prefetch
    from cave_amd.qpsolver import PreparedCones

    ctrs, costs, _ = synth.tsp_batch(20, 150, seed=5)
    x = torch.randn(150, 10)
    data = [(x[i:i + 64], torch.tensor(costs[i:i + 64]), torch.zeros(len(costs[i:i + 64]), 1), torch.zeros(len(costs[i:i + 64]), 1),
             torch.tensor(ctrs[i:i + 64])) for i in range(0, 150, 64)]  # batches of 64, 64, 22 (host tensors, as a DataLoader yields)

    class M:
        modelSense = EPO.MINIMIZE

    def run(loader, skip=None):
        torch.manual_seed(0)
        reg = torch.nn.Linear(10, costs.shape[1]).cuda()
        cave = innerConeAlignedCosine(M(), solver="hip", seed=3, solve_ratio=0.7)
        out, kinds = [], []
        for j, batch in enumerate(loader):
            xb, c, w, z, bctr = batch
            xb, c, w, z, bctr = xb.cuda(), c.cuda(), w.cuda(), z.cuda(), bctr.cuda()
            kinds.append(type(bctr).__name__)
            if j == skip:
                continue
            loss = cave(reg(xb), bctr)
            reg.zero_grad()
            loss.backward()
            out.append((float(loss.detach()), reg.weight.grad.clone()))
        return out, kinds

    plain, _ = run(data)
    fused, kinds = run(prefetch(data))
    assert kinds == ["PreparedCones"] * 3
    assert len(plain) == len(fused) == 3
    for (l0, g0), (l1, g1) in zip(plain, fused):
        assert abs(l0 - l1) <= 1e-6 and float((g0 - g1).abs().max()) <= 1e-6
    skipped, _ = run(prefetch(data), skip=1)   # batch 2's cones were attached to batch 1, which never ran its loss
    ref, _ = run(data, skip=1)
    for (l0, g0), (l1, g1) in zip(ref, skipped):
        assert abs(l0 - l1) <= 1e-6 and float((g0 - g1).abs().max()) <= 1e-6
    assert list(prefetch([])) == []


_RCCL_SCRIPT = r"""
import json, os, sys
sys.path[:0] = [{root!r}, os.path.join({root!r}, "tests")]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str({port}), RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
# the process group FIRST, before any other GPU call of this process (what an N-GPU rank does)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
import numpy as np
from cave_amd import synth
from cave_amd.cave import EPO, innerConeAlignedCosine
from cave_amd.dataset import ConeStore
from cave_amd.dist import global_mean_loss, same_branch_seed, allreduce_grads_sum
class M: modelSense = EPO.MINIMIZE
ctrs, costs, _ = synth.tsp_batch(12, 48, seed=2)
seed = same_branch_seed(4321)
assert seed == 4321
pred = torch.tensor(costs, device="cuda", requires_grad=True)
c = torch.tensor(ctrs, device="cuda")
mod = innerConeAlignedCosine(M(), solver="hip", seed=seed, reduction="none")
per = mod(pred, c)
g = global_mean_loss(per)            # the [sum loss, count] all-reduce over RCCL
g.backward()
grad_dist = pred.grad.clone()
pred.grad = None
mod2 = innerConeAlignedCosine(M(), solver="hip", seed=seed, reduction="mean")
l2 = mod2(pred, c)
l2.backward()
red = torch.stack([per.detach().sum(), torch.tensor(float(per.numel()), device="cuda")])
dist.all_reduce(red)
lin = torch.nn.Linear(3, 5).cuda()
lin(torch.ones(2, 3, device="cuda")).sum().backward()
w0 = lin.weight.grad.clone()
allreduce_grads_sum(lin.parameters())
ragged = [torch.from_numpy(x[np.abs(x).sum(axis=1) > 0]) for x in ctrs]
store = ConeStore.from_ragged_shard(ragged, 0, 1)
o = store.cone_op(torch.arange(store.n, device="cuda"), pred.detach(), 2, -1.0, 0.2, outputs=("loss",))
dist.barrier()
out = dict(loss_dist=float(g), loss_plain=float(l2), grad_diff=float((grad_dist - pred.grad).abs().max()),
           red=[float(red[0]), float(red[1])], wdiff=float((w0 - lin.weight.grad).abs().max()),
           shard_n=store.n, shard_loss=float(o["loss"].mean()), backend=dist.get_backend())
dist.destroy_process_group()
print("RCCL_RESULT " + json.dumps(out))
"""


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_process_group_at_world_size_one():
    """The code path an N-GPU run takes, on the one GPU of this box: `nccl` (= RCCL) initialised with one rank before
    any other GPU call, then dist.same_branch_seed, dist.global_mean_loss (the per-step [sum loss, count]
    all-reduce), the gradient sum-reduce and ConeStore.from_ragged_shard -- and the N = 1 values equal the
    non-distributed ones.  Runs in a child process (a fresh one: nothing has touched the GPU before the group)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT.format(root=ROOT, port=_free_port())], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_RESULT ")]
    assert line, r.stdout[-2000:]
    o = json.loads(line[-1][len("RCCL_RESULT "):])
    assert o["backend"] == "nccl"
    assert abs(o["loss_dist"] - o["loss_plain"]) <= 1e-6 and o["grad_diff"] <= 1e-7, o
    assert o["red"][1] == 48.0 and abs(o["red"][0] / 48.0 - o["loss_plain"]) <= 1e-5, o
    assert o["wdiff"] == 0.0 and o["shard_n"] == 48 and abs(o["shard_loss"] - o["loss_plain"]) <= 1e-5, o


def test_bench_force_dist_single_gpu():
    """`bench.py --gpus 1 --force-dist`: the bench's own distributed legs (process group, per-step all-reduce,
    barrier-bracketed timing, sharded packed store) at world size 1."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "5",
                        "--warmup", "2", "--cpu-sample", "0", "--instances", "1024", "--rotate", "1", "--batch", "256", "--no-extras"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert line, r.stdout[-2000:]
    res = json.loads(line[-1])
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["process_group"]["world_size"] == 1
    assert res["sharded_packed_store"]["instances"] == 1024 and res["sharded_packed_store"]["projections_per_s"] > 0


def _banded_inequality_cones(B, m, width, shift, seed, pairs=0):
    from test_simt_emul import _banded_inequality_cones as gen

    return gen(B, m, width, shift, seed, pairs)


@pytest.mark.parametrize("kind", ["sp12", "sp30", "tsp100", "banded", "tsp70"])
def test_large_path_two_launches_are_bit_identical(kind):
    """the large-cone path accumulated its band Hessian with floating-point atomics (global memory, or
    LDS for dense systems) and two launches on the same inputs differed by ~1e-7.  It now accumulates in 64-bit fixed
    point (integer adds are associative), builds the rows of all-free narrow bands on demand in a fixed order, and
    keeps dense systems in LDS: every output -- the iteration counts included -- must repeat bit for bit, at 4-, 2-
    and 1-wave workgroups (ABI v8 `waves`), with other work on the GPU in between."""
    import torch

    from cave_amd import synth
    from cave_amd.dataset import ConeStore

    dev = torch.device("cuda")
    if kind in ("sp12", "sp30", "tsp100", "tsp70"):
        ckind, size, B, chunk = {"sp12": ("sp", (12, 12), 12, 12), "sp30": ("sp", (30, 30), 6, 6),
                                 "tsp100": ("tsp", 100, 4, 2), "tsp70": ("tsp", 70, 6, 3)}[kind]
        items, costs, _ = synth.coo_batch(ckind, size, B, seed=17)
        d = int(costs.shape[1])
        m_max = max(it[3] for it in items)
        store = ConeStore.from_chunks_lazy(lambda i: synth.densify_on(items[i:i + chunk], d, dev, m_max),
                                           list(range(0, B, chunk)))
        pred, sign = torch.tensor(costs, device=dev), -1.0
    else:
        A, y = _banded_inequality_cones(6, 150, 7, 1, seed=5, pairs=6)
        store = ConeStore.from_dense(torch.tensor(A, device=dev), chunk=6)
        pred, sign, B = torch.tensor(y, device=dev), 1.0, 6
    assert store.large
    ids = torch.arange(B, device=dev)
    big = torch.randn(2048, 2048, device=dev)
    for waves in (4, 2, 1):
        store.large_waves = waves
        a = store.cone_op(ids, pred, MODE_INNER, sign, 0.2, outputs=ALL)
        _ = big @ big  # other work in between: different arrival order of the workgroups
        b = store.cone_op(ids.flip(0), pred.flip(0), MODE_INNER, sign, 0.2, outputs=ALL)  # another block -> instance map
        c = store.cone_op(ids, pred, MODE_INNER, sign, 0.2, outputs=ALL)
        assert bool((a["status"] == 0).all()), (kind, waves)
        for k in ALL + ("iters",):
            assert torch.equal(a[k], c[k]), (kind, waves, k)
            assert torch.equal(a[k], b[k].flip(0)), (kind, waves, k, "flipped")


def test_training_example_with_the_interior_point_inner_mode_on_30x30_grids():
    """VERDICT r2 item 7 'done means': innerConeAlignedCosine(..., solver_kwargs={'inner': 'ipm'}, max_iter=3) trains
    examples/train_sp_cave.py --grid 30 30 --packed (BASELINE configs[4]: shortest path 30x30, CaVE+): the large-cone
    path runs the truncated interior-point steps (one band LDL^T each), the loss falls and the regret does not rise."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import train_sp_cave

    hist = train_sp_cave.main(["--grid", "30", "30", "--num-data", "48", "--batch", "24", "--epochs", "4", "--packed",
                               "--inner", "ipm", "--max-iter", "3"])
    assert all(np.isfinite(h[1]) for h in hist[1:])
    assert hist[-1][1] < hist[1][1] and hist[-1][2] <= hist[0][2] + 1e-9, hist


@pytest.mark.gpu
def test_large_stores_carry_the_signs_in_their_indices():
    """The large-cone path reads the store in place in every Newton iteration.  For all-+-1 instances the host folds
    the sign of every entry into bit 15 of its 16-bit index (`ConeStore._fold_signs`, flags bit 1) and the kernels
    never load the fp32 value arrays: same outputs as the unfolded store, which a mixed store (one instance with a
    scaled row: not +-1, left as it is) checks instance by instance against the oracle."""
    import torch

    from cave_amd import synth
    from cave_amd.dataset import ConeStore
    from oracle import cave_oracle as O

    dev = torch.device("cuda")
    ctrs, costs, _ = synth.sp_batch(12, 12, 6, seed=3)
    ctrs = ctrs.copy()
    r = int(np.flatnonzero(np.abs(ctrs[5]).sum(axis=1) > 1.5)[0])
    ctrs[5, r] *= 2.0  # instance 5: one general row scaled (same cone, entries no longer +-1)
    store = ConeStore.from_dense(torch.tensor(ctrs, device=dev), chunk=6)
    assert store.large
    t = store.t
    flags = t["flags"].cpu().numpy()
    assert (flags[:5] & 3 == 3).all() and (flags[5] & 3) == 0
    nnz_off = t["nnz_off"].cpu().numpy()
    for idx, val in (("ccol", "cval"), ("cvar", "cvalc")):
        ix = t[idx].cpu().numpy().view(np.uint16)
        vv = t[val].cpu().numpy()
        z5 = int(nnz_off[5])
        assert ((ix[:z5] >> 15).astype(bool) == (vv[:z5] < 0)).all()          # folded instances: bit 15 = sign
        assert (ix[z5:int(nnz_off[6])] >> 15 == 0).all()                       # the general instance: plain indices
    o = store.cone_op(torch.arange(6, device=dev), torch.tensor(costs, device=dev), MODE_PROJECT, -1.0, 0.2,
                      outputs=("proj", "rnorm"))
    assert bool((o["status"] == 0).all())
    po, ro = O.batch_project(-costs, ctrs)  # (sign -1: the operator projects -pred)
    tol = 4e-6 * max(1.0, float(np.abs(costs).max()))
    assert np.abs(o["proj"].cpu().numpy() - po).max() <= tol and np.abs(o["rnorm"].cpu().numpy() - ro).max() <= tol
