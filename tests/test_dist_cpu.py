"""CPU tier: the N>1 path (instance sharding + scalar loss all-reduce + branch-seed agreement) with
gloo, world_size 2.  Per-instance losses come from the oracle here; on GPUs they come from the HIP kernel."""

import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _predictor(d):
    torch.manual_seed(7)
    return torch.nn.Linear(4, d).double()


def _features(n):
    g = torch.Generator().manual_seed(8)
    return torch.randn(n, 4, generator=g, dtype=torch.float64)


def _surrogate(pred, loss_np, grad_np):
    """Per-instance loss tensor with the oracle's value and the oracle's d loss_b / d pred_b."""
    g = torch.tensor(grad_np, dtype=pred.dtype)
    lin = (pred * g).sum(dim=1)
    return lin - lin.detach() + torch.tensor(loss_np, dtype=pred.dtype)


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cave_amd import synth
    from cave_amd.dist import global_mean_loss, same_branch_seed, shard_range
    from oracle import cave_oracle as O

    ctrs, costs, _ = synth.sp_batch(5, 5, 21, seed=4)  # 21 instances: uneven shards (11 + 10)
    lo, hi = shard_range(len(ctrs), rank, world)
    seed = same_branch_seed(1234 if rank == 0 else 999)  # rank 0 wins
    orc = O.ConeLossOracle(minimize=True, inner=True, solve_ratio=0.5, seed=seed, reduction="none")
    losses = []
    for _ in range(3):  # three steps: every rank must take the same QP/heuristic branch each time
        l, _ = orc(costs[lo:hi], ctrs[lo:hi])
        x = torch.tensor(l, dtype=torch.float64, requires_grad=True)
        g = global_mean_loss(x)
        g.backward()
        losses.append((float(g), x.grad.numpy().copy()))
    # predictor gradient: replicated nn.Linear under real DDP (gradient AVERAGE) + global_mean_loss must give
    # the gradient of the unsharded reduction='mean' loss
    pgrads = {}
    for mode in ("mean", "sum"):
        lin = _predictor(costs.shape[1])
        net = torch.nn.parallel.DistributedDataParallel(lin) if mode == "mean" else lin
        feats = _features(len(ctrs))[lo:hi]
        pred = net(feats)
        exact = O.ConeLossOracle(minimize=True, inner=False, reduction="none")
        l, gpred = exact(pred.detach().numpy(), ctrs[lo:hi])
        li = _surrogate(pred, l, gpred)
        global_mean_loss(li, grad_reduce=mode).backward()
        if mode == "sum":
            from cave_amd.dist import allreduce_grads_sum

            allreduce_grads_sum(lin.parameters())
        pgrads[mode] = [p.grad.numpy().copy() for p in lin.parameters()]
    q.put((rank, lo, hi, seed, losses, pgrads))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_loss_matches_unsharded():
    from cave_amd import synth
    from oracle import cave_oracle as O

    world, port = 2, 29000 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ctrs, costs, _ = synth.sp_batch(5, 5, 21, seed=4)
    assert [(r[1], r[2]) for r in res] == [(0, 11), (11, 21)]
    assert res[0][3] == res[1][3] == 1234
    full = O.ConeLossOracle(minimize=True, inner=True, solve_ratio=0.5, seed=1234, reduction="mean")
    for step in range(3):
        want, _ = full(costs, ctrs)
        got0, grad0 = res[0][4][step]
        got1, grad1 = res[1][4][step]
        assert abs(got0 - want) < 1e-6 and abs(got1 - want) < 1e-6  # both ranks hold the global mean
        # default convention: local gradient world/B_global, to be AVERAGED over ranks by DDP
        assert np.allclose(grad0, 2.0 / 21) and np.allclose(grad1, 2.0 / 21)
    # single-process gradient of the unsharded mean loss w.r.t. the predictor
    lin = _predictor(costs.shape[1])
    pred = lin(_features(len(ctrs)))
    l, gpred = O.ConeLossOracle(minimize=True, inner=False, reduction="none")(pred.detach().numpy(), ctrs)
    _surrogate(pred, l, gpred).mean().backward()
    want = [p.grad.numpy() for p in lin.parameters()]
    for r in res:
        for mode in ("mean", "sum"):
            for got, w in zip(r[5][mode], want):
                assert np.allclose(got, w, rtol=1e-9, atol=1e-12), (r[0], mode)


def test_weighted_shards_balance_by_nnz():
    from cave_amd.dist import weighted_shards

    rng = np.random.default_rng(0)
    w = rng.integers(100, 5000, size=1000)
    for world in (1, 2, 4, 8):
        parts = weighted_shards(w, world)
        assert sorted(np.concatenate(parts).tolist()) == list(range(1000))
        loads = [w[p].sum() for p in parts]
        assert max(loads) - min(loads) <= w.max()  # LPT: within one instance of even
    # equal weights -> equal counts
    assert [len(p) for p in weighted_shards(np.ones(1024), 8)] == [128] * 8
    # a by-count split of a skewed dataset would be far off; by-weight is not
    w2 = np.r_[np.full(100, 5000), np.full(900, 100)]
    loads = [w2[p].sum() for p in weighted_shards(w2, 4)]
    assert max(loads) / min(loads) < 1.05


def test_shard_range_balanced():
    from cave_amd.dist import shard_range

    for n in (0, 1, 7, 1024, 1000):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
