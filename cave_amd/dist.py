"""Data-parallel plumbing for the cone loss: instances are independent, so a batch shards by
instance with no data-path collective; the only exchange is the 8-byte [sum(loss), count]
all-reduce that turns per-rank losses into the global mean (RCCL over xGMI when the backend is
"nccl"; gloo in the CPU tests).  The predictor's gradient all-reduce is ordinary DDP and not
handled here.  SURVEY.md §8e.
"""

from __future__ import annotations

import torch
import torch.distributed as dist

__all__ = ["shard_range", "global_mean_loss", "same_branch_seed"]


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of n instances for this rank (sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_mean_loss(per_instance_loss: torch.Tensor, group=None) -> torch.Tensor:
    """Mean over ALL ranks' instances of a (B_local,) loss vector.

    Differentiable w.r.t. the local losses: d(global mean)/d(loss_b) = 1 / B_global, which is what
    `reduction='mean'` over the unsharded batch gives (src/cave.py:73)."""
    buf = torch.stack([per_instance_loss.detach().sum(),
                       torch.tensor(float(per_instance_loss.numel()), device=per_instance_loss.device,
                                    dtype=per_instance_loss.dtype)])
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    total, count = buf[0], buf[1]
    # value = global mean; gradient flows through the local sum only
    local = per_instance_loss.sum()
    return (local - local.detach() + total) / count


def same_branch_seed(seed: int, group=None) -> int:
    """Broadcast rank 0's seed so every rank's Hybrid module draws the same QP/heuristic branch
    per step (src/cave.py:195,201)."""
    t = torch.tensor([int(seed)], dtype=torch.int64)
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0, group=group)
    return int(t.item())
