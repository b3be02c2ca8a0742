"""N > 1 path on CPU: world_size-2 gloo processes exercise the shard + all-gather plumbing
(adacharge_amd/distributed.py) with a stand-in per-shard solve (no GPU here)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from adacharge_amd.distributed import shard_range, solve_sharded


def test_shard_range_partitions_everything():
    for total in (0, 1, 7, 256, 8192):
        for world in (1, 2, 3, 8):
            pieces = [shard_range(total, r, world) for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
            sizes = [hi - lo for lo, hi in pieces]
            assert max(sizes) - min(sizes) <= 1


def _fake_solve(lo, hi):
    idx = np.arange(lo, hi)
    x = idx[:, None, None] * np.ones((1, 3, 4)) + np.arange(4)[None, None, :] * 0.25
    return x, (idx % 3 + 1).astype(np.int32)


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, st = solve_sharded(total, _fake_solve)
    q.put((rank, x, st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_solve_sharded_gloo_world2(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + total
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_x, want_st = _fake_solve(0, total)
    for rank, x, st in got:
        assert x.shape == want_x.shape
        assert np.array_equal(x, want_x) and np.array_equal(st, want_st)


# ---- configs[3] as a sharded job: site-major order, ragged site widths, fake per-site solve -----------------------
class _FakeBatch:
    """The three attributes solve_sites_sharded reads, plus subset()."""

    def __init__(self, k, B, N, Tm, lo=0):
        self.k, self.B, self.N, self.Tm, self.lo = k, B, N, Tm, lo

    def subset(self, sl):
        a, e, _ = sl.indices(self.B)
        return _FakeBatch(self.k, e - a, self.N, self.Tm, self.lo + a)


def _fake_site(k, sub):
    idx = np.arange(sub.lo, sub.lo + sub.B)
    x = (1000 * k + idx)[:, None, None] + np.arange(sub.N)[None, :, None] * 0.5 + np.arange(sub.Tm)[None, None, :] * 0.01
    return x, ((idx + k) % 3 + 1).astype(np.int32)


SITES = [(5, 40), (3, 64), (4, 52), (6, 48)]   # (scenarios, EVSEs): ragged on both axes


def _site_worker(rank, world, port, q):
    from adacharge_amd.distributed import solve_sites_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batches = [_FakeBatch(k, B, N, 12) for k, (B, N) in enumerate(SITES)]
    x, st = solve_sites_sharded(batches, solve_site=_fake_site)
    q.put((rank, x, st))
    dist.barrier()
    dist.destroy_process_group()


def test_site_major_sharding_gloo_world2_ragged_sites():
    from adacharge_amd.distributed import site_major_layout

    assert site_major_layout([5, 3, 4, 6]).tolist() == [0, 5, 8, 12, 18]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 400)
    procs = [ctx.Process(target=_site_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_x = np.zeros((18, 64, 12))
    want_st = np.zeros(18, np.int32)
    o = 0
    for k, (B, N) in enumerate(SITES):
        x, st = _fake_site(k, _FakeBatch(k, B, N, 12))
        want_x[o:o + B, :N] = x
        want_st[o:o + B] = st
        o += B
    for rank, x, st in got:   # rank 0 owns sites 0, 1 and one scenario of site 2; rank 1 the rest: both end with the whole job
        assert np.array_equal(x, want_x) and np.array_equal(st, want_st)
