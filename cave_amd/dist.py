"""Data-parallel plumbing for the cone loss: instances are independent, so a batch shards by
instance with no data-path collective; the only exchange is the 8-byte [sum(loss), count]
all-reduce that turns per-rank losses into the global mean (RCCL over xGMI when the backend is
"nccl"; gloo in the CPU tests).  SURVEY.md §8e.

Gradient convention.  The predictor is replicated and its gradients are all-reduced by DDP, which
AVERAGES over ranks.  `global_mean_loss(..., grad_reduce="mean")` (the default) therefore scales the
local gradient by world/B_global, so that DDP's average equals the gradient of the unsharded
`reduction='mean'` loss (src/cave.py:73); with `grad_reduce="sum"` the local gradient is 1/B_global
per instance, for callers that SUM-reduce predictor gradients themselves (`allreduce_grads_sum`).
"""

from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["shard_range", "weighted_shards", "global_mean_loss", "same_branch_seed", "allreduce_grads_sum"]


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced-by-count [lo, hi) slice of n instances for this rank (sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def weighted_shards(weights, world: int) -> list[np.ndarray]:
    """Partition instances over `world` ranks balanced by total weight (sum of non-zeros of the packed
    cones, or m_i*d for the dense form: SURVEY.md §8e), not by count.

    Longest-processing-time greedy: instances in decreasing weight, each to the lightest rank so far
    (ties: lowest rank) -- max load <= 4/3 of optimal, and exactly even for equal weights.  Every rank
    computes the same partition from the same weights; each shard is returned as sorted instance indices.
    """
    w = np.asarray(weights, dtype=np.float64).ravel()
    order = np.argsort(-w, kind="stable")
    load = np.zeros(world)
    count = np.zeros(world, dtype=np.int64)
    owner = np.empty(len(w), dtype=np.int64)
    for i in order:
        r = int(np.lexsort((count, load))[0])  # lightest, then fewest instances
        owner[i] = r
        load[r] += w[i]
        count[r] += 1
    return [np.flatnonzero(owner == r) for r in range(world)]


def global_mean_loss(per_instance_loss: torch.Tensor, group=None, grad_reduce: str = "mean") -> torch.Tensor:
    """Mean over ALL ranks' instances of a (B_local,) loss vector; every rank gets the same value.

    grad_reduce="mean": d/d(loss_b) = world / B_global, to be followed by DDP's gradient AVERAGE;
    grad_reduce="sum":  d/d(loss_b) = 1 / B_global, to be followed by a gradient SUM.
    Either way the reduced predictor gradient is that of the unsharded mean loss (src/cave.py:73)."""
    if grad_reduce not in ("mean", "sum"):
        raise ValueError("grad_reduce must be 'mean' or 'sum'")
    buf = torch.stack([per_instance_loss.detach().sum(),
                       torch.tensor(float(per_instance_loss.numel()), device=per_instance_loss.device,
                                    dtype=per_instance_loss.dtype)])
    world = 1
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        world = dist.get_world_size(group)
    total, count = buf[0], buf[1]
    scale = float(world) if grad_reduce == "mean" else 1.0
    local = per_instance_loss.sum()
    # value = global mean; the gradient flows through the local sum only
    return ((local - local.detach()) * scale + total) / count


def allreduce_grads_sum(params, group=None) -> None:
    """SUM all-reduce of the predictor's gradients (the companion of grad_reduce="sum")."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for p in params:
        if p.grad is not None:
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group)


def same_branch_seed(seed: int, group=None) -> int:
    """Broadcast rank 0's seed so every rank's Hybrid module draws the same QP/heuristic branch
    per step (src/cave.py:195,201)."""
    t = torch.tensor([int(seed)], dtype=torch.int64)
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0, group=group)
    return int(t.item())
