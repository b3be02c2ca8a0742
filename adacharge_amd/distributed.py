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


def site_major_layout(batch_sizes: Sequence[int]):
    """Problem offsets of a site-major job (BASELINE.json configs[3]: S demand scenarios x K sites, all problems of
    site k contiguous): ``offsets[k] .. offsets[k + 1]`` are site k's problems."""
    off = np.zeros(len(batch_sizes) + 1, dtype=np.int64)
    np.cumsum(np.asarray(batch_sizes, np.int64), out=off[1:])
    return off


def solve_sites_sharded(site_batches, solve_site: Callable = None, options=None, group=None, device=None, local_device: int = 0):
    """configs[3] as a sharded job: ``site_batches[k]`` is the ProblemBatch of site k (its own site matrix and EVSE
    count, one horizon for all).  The job's problems are ordered site-major and cut into contiguous rank shards
    (``shard_range``): with as many ranks as sites every rank owns exactly one site and its site matrix stays resident
    on that GPU; with fewer ranks a rank owns several whole or partial sites, each through its own ``SiteHandle``.
    No collective on the data path; ONE all-gather of the schedules, padded to the widest site (N_max) so that every
    rank ends with ``x (total, N_max, Tm)`` and ``status (total,)``.

    ``solve_site(k, sub_batch) -> (x (n, N_k, Tm), status (n,))`` (tensors on the collective's device, or arrays);
    default: the HIP path -- ``acnqp_solve_batch_device`` on HBM-resident buffers of GPU ``local_device``, the result
    tensor going into the all-gather as it is."""
    import torch

    sizes = [b.B for b in site_batches]
    off = site_major_layout(sizes)
    n_max = max(b.N for b in site_batches)
    t_max = max(b.Tm for b in site_batches)
    handles = {}

    def hip_launch(k, sub):
        """Upload site k's problems and launch its solve on a stream of its own; returns the device batch (results valid
        after the device has been synchronised)."""
        from .backend import DeviceBatch, SiteHandle, default_options

        if k not in handles:
            handles[k] = (SiteHandle(sub.site, local_device), torch.cuda.Stream(device=local_device))
        h, stream = handles[k]
        dev = DeviceBatch(sub, torch.device("cuda", local_device))        # copies on the current stream
        stream.wait_stream(torch.cuda.current_stream(local_device))
        h.solve_device(dev, options if options is not None else default_options(), stream=stream.cuda_stream)
        return dev

    def solve_local(lo, hi):
        # A rank that owns several sites launches them ALL before it waits (one stream per site): the sites' kernels
        # share the GPU, so the rank ends with its slowest site instead of the sum -- on the one-GPU rehearsal of
        # configs[3] the congested site's stragglers (65 ms for 1,024 scenarios) hide the other seven sites entirely.
        parts = []
        for k, b in enumerate(site_batches):
            a, e = max(lo, int(off[k])), min(hi, int(off[k + 1]))
            if a >= e:
                continue
            sub = b.subset(slice(a - int(off[k]), e - int(off[k])))
            parts.append((e - a, hip_launch(k, sub) if solve_site is None else solve_site(k, sub)))
        if solve_site is None and parts:
            torch.cuda.synchronize(local_device)
        xs, ss = [], []
        for n, res in parts:
            x, st = (res.x, res.status) if solve_site is None else res
            x = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
            st = st if torch.is_tensor(st) else torch.from_numpy(np.ascontiguousarray(st))
            pad = torch.zeros((n, n_max, t_max), dtype=torch.float64, device=x.device)
            pad[:, : x.shape[1], : x.shape[2]] = x
            xs.append(pad)
            ss.append(st.to(torch.int32))
        if not xs:
            return np.zeros((0, n_max, t_max)), np.zeros(0, np.int32)
        return torch.cat(xs), torch.cat(ss)

    try:
        return solve_sharded(int(off[-1]), solve_local, group=group, device=device)
    finally:
        for h, _ in handles.values():
            h.close()
