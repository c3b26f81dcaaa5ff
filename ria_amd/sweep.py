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
    engine.channel_exact_(x, point.channel, point.snr_db, base_seed + 7919 * (point_index + 1), first_frame=first)
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


# ------------------------------------------------------------------------------------------------------------
# Adaptive ladder (BASELINE.json config 5): the reference picks (waveform, modulation, rate, spreading) from
# SNR and fading index (protocol::recommendWaveformAndRate, waveform_selection.hpp:112-222 -> ria_link_recommend);
# MC-DPSK rungs carry one R1/4 codeword per frame and retransmit failed codewords with HARQ chase combining
# (fec::ChaseCache, <= 4 receptions, LLR sum); OFDM rungs are single shot.
LADDER_COUNTERS = ("frames", "frame_err", "cw_err", "byte_err", "iters_sum", "transmissions")
# typical measured fading index of the ITU-R F.1487 presets (thresholds 0.15 / 0.65 / 1.10, waveform_selection.hpp:49-61)
PRESET_FADING = {0: 0.05, 1: 0.45, 2: 0.90, 3: 1.20, 4: 1.40}
_MOD_NAMES = {0: "DBPSK", 1: "BPSK", 2: "DQPSK", 3: "QPSK", 4: "D8PSK", 6: "QAM16", 7: "QAM32", 8: "QAM64", 10: "QAM256"}
_RATE_NAMES = {0: "R1_4", 1: "R1_3", 2: "R1_2", 3: "R2_3", 4: "R3_4", 5: "R5_6"}


def ladder_mode(lib, snr_db, fading):
    import ctypes as C
    from . import capi
    o = capi.LinkRecommendation()
    lib.ria_link_recommend(float(snr_db), float(fading), C.byref(o))
    return o


def run_ladder_chunk(engines, point, base_seed, point_index, start, n, max_tx=4, n_payloads=32):
    """One chunk of trials of one (channel, SNR) point with the mode the ladder picks.
    engines: callable (mod_name, rate_name) -> RxEngine (cached by the caller)."""
    from . import capi
    lib = capi.load()
    rec = ladder_mode(lib, point.snr_db, PRESET_FADING.get(point.channel, 0.9))
    mod, rate = _MOD_NAMES[rec.modulation], _RATE_NAMES[rec.code_rate]
    e = engines(mod, rate)
    if rec.waveform != 4:   # OFDM-CHIRP rung: frame = 4 codewords, single shot
        row = run_point_gpu(e, point, base_seed, point_index, start, n)
        return np.array([row[0], row[1], row[2], row[3], row[4], n], dtype=np.int64)
    # MC-DPSK rung: one R1/4 codeword (162 information bits) per frame, 10 carriers
    bps = 1 if rec.modulation == 0 else 2
    sp = int(rec.spreading)
    dev = e.device
    rng = np.random.default_rng([int(base_seed) & 0x7fffffff, point_index, start])
    info = rng.integers(0, 256, (n_payloads, 21), dtype=np.uint8)
    info[:, -1] &= 0xC0                                   # k = 162 bits: the 21st byte carries 2 information bits
    coded = e.ldpc_encode(info)
    frames = np.stack([e.mcdpsk_modulate(coded[i], int(rec.num_carriers), bps, sp) for i in range(n_payloads)])
    pick = torch.from_numpy(rng.integers(0, n_payloads, n)).to(dev)
    clean = torch.from_numpy(frames).to(dev)[pick]                      # [n, samples]
    info_t = torch.from_numpy(info).to(dev)[pick]
    chan_seed = (int(base_seed) * 1000003 + point_index * 7919) & 0x7fffffff
    acc = torch.zeros((n, 648), dtype=torch.float32, device=dev)
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    decoded = torch.zeros(n, dtype=torch.uint8, device=dev)
    out = torch.zeros((n, 21), dtype=torch.uint8, device=dev)
    iters_sum, transmissions = 0, 0
    for tx_no in range(max_tx):
        todo = (decoded == 0).nonzero().flatten()
        if todo.numel() == 0:
            break
        transmissions += int(todo.numel())
        # the reference channel model on the MC-DPSK audio (any frame length): a fresh stream per (transmission, trial)
        x = clean[todo].contiguous()
        e.channel_exact_(x, point.channel, point.snr_db, chan_seed + 104729 * tx_no, first_frame=start + int(todo[0].item()))
        llr, _ = e.mcdpsk_demod(x, int(rec.num_carriers), bps, sp)
        soft = llr[:, :648].contiguous()
        a, c = acc[todo].contiguous(), cnt[todo].contiguous()
        e.chase_combine(a, c, soft)                                     # first reception copies, later ones add
        acc[todo], cnt[todo] = a, c
        o, ok, it = e.ldpc_decode(a, 50, 0.9375)
        iters_sum += int(it.to(torch.int64).sum().item())
        good = ok.bool() & (o == info_t[todo]).all(dim=1)
        out[todo] = o
        decoded[todo] = torch.where(good, torch.ones_like(decoded[todo]), decoded[todo])
    fail = int((decoded == 0).sum().item())
    byte_err = int((out != info_t).sum().item())
    return np.array([n, fail, fail, byte_err, iters_sum, transmissions], dtype=np.int64)
