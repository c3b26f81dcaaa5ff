"""Monte-Carlo BER/FER sweep driver (BASELINE.json configs 3/5): embarrassingly parallel over GPUs.

Trials are sharded over ranks with NO data-path collective; torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests) is used only to broadcast the sweep descriptor/seed and to
all-reduce the error counters at the end.  Every per-trial seed is a function of
(base seed, global trial index), so the counters do not depend on the number of ranks.
"""
from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist

COUNTERS = ("frames", "frame_err", "cw_err", "byte_err", "iters_sum", "attempts_sum")


@dataclass
class SweepPoint:
    channel: int      # 0 awgn, 1 good, 2 moderate, 3 poor, 4 flutter
    snr_db: float


def shard_range(n_trials, rank, world, chunk):
    """Global trial chunks owned by `rank`: chunk c (of size `chunk`) goes to rank c % world."""
    out = []
    c = 0
    for start in range(0, n_trials, chunk):
        if c % world == rank:
            out.append((start, min(chunk, n_trials - start)))
        c += 1
    return out


def broadcast_descriptor(base_seed, points, device):
    """Rank 0's (seed, grid) is authoritative: tens of bytes, one broadcast."""
    buf = torch.zeros(2 + 2 * 64, dtype=torch.float64, device=device)
    if not dist.is_initialized() or dist.get_rank() == 0:
        buf[0] = float(base_seed)
        buf[1] = float(len(points))
        for i, p in enumerate(points):
            buf[2 + 2 * i] = float(p.channel)
            buf[3 + 2 * i] = float(p.snr_db)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(buf, 0)
    n = int(buf[1].item())
    pts = [SweepPoint(int(buf[2 + 2 * i].item()), float(buf[3 + 2 * i].item())) for i in range(n)]
    return int(buf[0].item()), pts


def reduce_counters(local, device):
    """local: int64 [n_points, len(COUNTERS)] -> summed over ranks (one all-reduce of a few hundred bytes)."""
    t = torch.as_tensor(local, dtype=torch.int64, device=device).clone()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def run_point_gpu(engine, point, base_seed, point_index, start, n):
    """One chunk of trials of one sweep point on this rank's GPU. Returns the counter row."""
    first = point_index * (1 << 32) + start          # disjoint global frame indices per point
    info = engine.make_frames(base_seed, start, n)
    x = engine.tx(info, peak=0.8)
    engine.channel_(x, point.channel, point.snr_db, base_seed + 7919 * (point_index + 1), first_frame=first)
    out, st = engine.rx(x)
    s = engine.decode_status(st)
    ok = s["cw_ok"].all(axis=1) & s["frame_valid"].astype(bool)
    same = (out == info).all(dim=1).cpu().numpy()
    byte_err = int((out != info).sum().item())
    return np.array([n, int((~(ok & same)).sum()), int((s["cw_ok"] == 0).sum()), byte_err,
                     int(s["iterations"].sum()), int(s["attempts"].sum())], dtype=np.int64)


def run_sweep(points, n_trials, base_seed, run_chunk, device, chunk=4096):
    """Generic driver: `run_chunk(point, base_seed, point_index, start, n) -> counter row`."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    seed, pts = broadcast_descriptor(base_seed, points, device)
    local = np.zeros((len(pts), len(COUNTERS)), dtype=np.int64)
    for pi, p in enumerate(pts):
        for start, n in shard_range(n_trials, rank, world, chunk):
            local[pi] += run_chunk(p, seed, pi, start, n)
    return reduce_counters(local, device), pts
