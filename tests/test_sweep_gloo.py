"""CPU, world_size 2 on gloo: the multi-GPU sharding/reduction logic of ria_amd/sweep.py.
The per-chunk compute is replaced by a deterministic function of the GLOBAL trial index, so the test
checks what matters for N>1: disjoint complete sharding, rank-0 descriptor broadcast, counter
all-reduce, and independence of the result from the number of ranks."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ria_amd import sweep


def fake_chunk(point, seed, pi, start, n):
    idx = np.arange(start, start + n, dtype=np.int64)
    h = (idx * 2654435761 + seed + 97 * pi + int(point.snr_db * 10)) % 1000
    err = (h < 100 + 50 * point.channel)
    return np.array([n, int(err.sum()), int((h % 7 == 0).sum()), int(h.sum()), int((h % 13).sum()), n], np.int64)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pts = [sweep.SweepPoint(2, 20.0), sweep.SweepPoint(0, 12.0)] if rank == 0 else [sweep.SweepPoint(9, -1.0)]
    seed = 1234 if rank == 0 else 999  # only rank 0's descriptor may count
    total, got = sweep.run_sweep(pts, 10000, seed, fake_chunk, torch.device("cpu"), chunk=512)
    q.put((rank, total, [(p.channel, p.snr_db) for p in got]))
    dist.destroy_process_group()


def test_sharding_is_complete_and_disjoint():
    for world in (1, 2, 3, 8):
        seen = np.zeros(10000, int)
        for r in range(world):
            for s, n in sweep.shard_range(10000, r, world, 512):
                seen[s:s + n] += 1
        assert (seen == 1).all()


def test_world_size_2_matches_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pts = [sweep.SweepPoint(2, 20.0), sweep.SweepPoint(0, 12.0)]
    single, _ = sweep.run_sweep(pts, 10000, 1234, fake_chunk, torch.device("cpu"), chunk=512)
    for rank, total, got in res:
        assert got == [(2, 20.0), (0, 12.0)], "rank 0's grid must be the one every rank runs"
        assert np.array_equal(total, single), f"rank {rank}: reduced counters differ from the 1-process run"
    assert single[:, 0].tolist() == [10000, 10000]
