"""Multi-GPU partitioning of a batch of independent streams.

Every stream is independent (the reference resets all state per call,
yeast.c:1315-1336), so a batch is cut into contiguous shards, one per rank,
balanced by compressed bytes.  The decoded payload stays on the GPU that
produced it; the only exchange is an all-gather of the fixed 32-byte result
records (torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def balanced_shards(weights: Sequence[int], nshards: int) -> List[Tuple[int, int]]:
    """Contiguous [begin, end) index ranges whose weight sums are as even as a
    greedy prefix cut allows.  Every index is covered exactly once; shards may be
    empty when there are fewer items than shards."""
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    if nshards <= 0:
        raise ValueError("nshards must be positive")
    pre = np.concatenate([[0.0], np.cumsum(w)])
    total = pre[-1]
    cuts = [0]
    for k in range(1, nshards):
        target = total * k / nshards
        i = int(np.searchsorted(pre, target, side="left"))
        # choose the nearer of the two neighbouring prefix sums
        if i > 0 and (i > n or abs(pre[i - 1] - target) <= abs(pre[min(i, n)] - target)):
            i -= 1
        i = min(max(i, cuts[-1]), n)
        cuts.append(i)
    cuts.append(n)
    return [(cuts[k], cuts[k + 1]) for k in range(nshards)]


def all_gather_records(local, dist=None, group=None, sizes=None):
    """All-gather per-stream 32-byte records.  `local` is a uint8 torch tensor
    of shape [n_local * 32] (CPU with gloo, GPU with nccl).  Returns a uint8
    tensor holding the records of rank 0, rank 1, ... in order.  `sizes` (bytes
    per rank), when the caller knows the shards, saves the exchange of counts."""
    import torch
    if dist is None:
        import torch.distributed as dist
    world = dist.get_world_size(group)
    # (no shortcut for a group of one: the collective runs, so a one-GPU box exercises RCCL too)
    if sizes is None:
        count = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
        counts = [torch.zeros_like(count) for _ in range(world)]
        dist.all_gather(counts, count, group=group)
        sizes = [int(c.item()) for c in counts]
    elif len(sizes) != world or sizes[dist.get_rank(group)] != local.numel():
        raise ValueError("all_gather_records: sizes do not describe this group's shards")
    cap = max(sizes)
    if all(s == cap for s in sizes):
        out = torch.empty(cap * world, dtype=torch.uint8, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros(cap, dtype=torch.uint8, device=local.device)
    padded[:local.numel()] = local
    parts = [torch.empty(cap, dtype=torch.uint8, device=local.device) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)])
