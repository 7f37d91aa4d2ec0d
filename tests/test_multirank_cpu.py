"""world_size-2 gloo test of the N > 1 path's host logic: shard bounds and the new-geometry all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, I, N, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multioptpy_amd.parallel import shard_bounds, gather_band
    first, count = shard_bounds(I, world, rank)
    rng = np.random.default_rng(0)
    full = rng.normal(size=(I, N, 3))                 # what a whole-band step would produce
    mine = torch.from_numpy(full[first:first + count].copy())
    got = gather_band(mine, I)
    ok = bool(np.array_equal(got.numpy(), full))
    q.put((rank, first, count, ok))
    dist.destroy_process_group()


def test_shard_bounds():
    from multioptpy_amd.parallel import shard_bounds
    assert [shard_bounds(32, 8, r) for r in range(8)] == [(4 * r, 4) for r in range(8)]
    assert shard_bounds(32, 1, 0) == (0, 32)
    with pytest.raises(ValueError):
        shard_bounds(30, 4, 0)


def test_gather_band_world2_gloo():
    world, I, N = 2, 8, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, I, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, 0, 4, True), (1, 4, 4, True)]
