"""Batch-axis sharding over the GPUs of one node.

Every QP is independent, so the data path needs no collective: rank r solves
the contiguous slice ``shard_range(B, r, world)`` on its own GPU.  One
all-gather of the schedules (RCCL over xGMI when the process group's backend is
"nccl"; "gloo" on CPU for the tests) leaves the whole job's result on every
rank.  One process per GPU, launched by torch.distributed.run.
"""
from __future__ import annotations

from typing import Callable, Sequence, Tuple

import numpy as np


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of ``total`` items for ``rank``."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def solve_sharded(
    n_problems: int,
    solve_local: Callable[[int, int], Tuple[np.ndarray, np.ndarray]],
    group=None,
    device=None,
):
    """Run ``solve_local(lo, hi) -> (x (hi-lo, N, T), status (hi-lo,))`` on this
    rank's shard and all-gather.  Returns ``(x (n_problems, N, T), status)`` on
    every rank.  Shards may be ragged; they are padded to the largest shard for
    the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        x, st = solve_local(0, n_problems)
        to_np = lambda a: a.cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
        return to_np(x), to_np(st)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n_problems, rank, world)
    x, st = solve_local(lo, hi)
    cap = max(shard_range(n_problems, r, world)[1] - shard_range(n_problems, r, world)[0] for r in range(world))
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    # a shard that is already a tensor on the collective's device (DeviceBatch.x after solve_device, or the x_dev sink
    # of acnqp_solve_batches) enters the all-gather from HBM; numpy shards are uploaded once
    as_tensor = lambda a, dt: (a.to(device=device, dtype=dt) if torch.is_tensor(a)
                               else torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt))
    x, st = as_tensor(x, torch.float64), as_tensor(st, torch.int32)
    xs = torch.zeros((cap,) + tuple(x.shape[1:]), dtype=torch.float64, device=device)
    ss = torch.zeros((cap,), dtype=torch.int32, device=device)
    xs[: hi - lo] = x
    ss[: hi - lo] = st
    xg = torch.empty((world * cap,) + x.shape[1:], dtype=torch.float64, device=device)
    sg = torch.empty((world * cap,), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(xg, xs, group=group)
    dist.all_gather_into_tensor(sg, ss, group=group)
    xg, sg = xg.cpu().numpy(), sg.cpu().numpy()
    keep = np.concatenate(
        [np.arange(r * cap, r * cap + (shard_range(n_problems, r, world)[1] - shard_range(n_problems, r, world)[0]))
         for r in range(world)]
    )
    return xg[keep], sg[keep]
