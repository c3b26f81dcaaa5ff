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


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)).astype(np.uint64)
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)).astype(np.uint64)
    return x ^ (x >> np.uint64(31))


def trial_seed32(base_seed, point_index, tx_no, trials):
    """mt19937 seed of the channel of one transmission of one Monte-Carlo trial: a hash of (base seed, sweep point,
    transmission number, GLOBAL trial index) folded to 32 bits - nothing else.  A trial's noise therefore does not
    depend on the batch it is computed in, on which other trials already decoded, or on the number of GPUs."""
    with np.errstate(over="ignore"):
        t = np.asarray(trials, dtype=np.uint64)
        h = _splitmix64(np.uint64(int(base_seed) & 0xFFFFFFFFFFFFFFFF) ^ _splitmix64(np.uint64(point_index) * np.uint64(0x10001) + np.uint64(tx_no) * np.uint64(0x1000000001)))
        h = _splitmix64(h ^ _splitmix64(t))
    return ((h >> np.uint64(32)) ^ (h & np.uint64(0xFFFFFFFF))).astype(np.uint32)


def run_point_gpu(engine, point, base_seed, point_index, start, n):
    """One chunk of trials of one sweep point on this rank's GPU. Returns the counter row."""
    info = engine.make_frames(base_seed, start, n)
    x = engine.tx(info, peak=0.8)
    engine.channel_exact_seeded_(x, point.channel, point.snr_db, trial_seed32(base_seed, point_index, 0, np.arange(start, start + n)))
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
# nominal fading index of the ITU-R F.1487 presets (thresholds 0.15 / 0.65 / 1.10, waveform_selection.hpp:49-61): only a
# label for tables; the sweep feeds the ladder the demodulator's measured value (measured_fading_index)
PRESET_FADING = {0: 0.05, 1: 0.45, 2: 0.90, 3: 1.20, 4: 1.40}
_MOD_NAMES = {0: "DBPSK", 1: "BPSK", 2: "DQPSK", 3: "QPSK", 4: "D8PSK", 6: "QAM16", 7: "QAM32", 8: "QAM64", 10: "QAM256"}
_RATE_NAMES = {0: "R1_4", 1: "R1_3", 2: "R1_2", 3: "R2_3", 4: "R3_4", 5: "R5_6"}


def ladder_mode(lib, snr_db, fading):
    import ctypes as C
    from . import capi
    o = capi.LinkRecommendation()
    lib.ria_link_recommend(float(snr_db), float(fading), C.byref(o))
    return o


def run_harq_trials(e, carriers, bps, spreading, kind, snr_db, info21, seeds, want_crc=False):
    """The MC-DPSK data-codeword chain with HARQ chase combining on the GPU, per transmission exactly what
    StreamingDecoder::decodeMCDPSKFrame does for a codeword >= 1 (streaming_decoder.cpp:2758-2800):
      audio -> channel (seeds[i, t]) -> MC-DPSK demodulator -> robustDecodeSingleCW of the fresh soft bits;
      on failure ChaseCache::store (copy / add) and, from the second reception on, robustDecodeSingleCW of the sum.
    info21 uint8 [n, 21] (162 information bits), seeds uint32 [n, max_tx].  Returns a dict shaped like the one the tests record from
    the reference with (tests/golden/harq_trials.npz): tx_to_success [n] (0 = never), decoded [n, 20], fading [n, max_tx],
    tries [n, max_tx, 2], iterations_sum, and with want_crc the zlib.crc32 of every reception's soft bits / cache sum."""
    import zlib
    dev = e.device
    info21 = np.ascontiguousarray(info21, np.uint8)
    seeds = np.ascontiguousarray(seeds, np.uint32)
    n, max_tx = seeds.shape
    uniq, inv = np.unique(info21, axis=0, return_inverse=True)
    coded = e.ldpc_encode(uniq)
    frames = e.mcdpsk_modulate_batch(coded, carriers, bps, spreading)     # the few distinct messages, modulated on the device
    clean = frames[torch.from_numpy(np.asarray(inv).reshape(-1)).to(dev)]
    acc = torch.zeros((n, 648), dtype=torch.float32, device=dev)
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    tts = np.zeros(n, np.int32)
    decoded = np.zeros((n, 20), np.uint8)
    llr_crc = np.zeros((n, max_tx), np.uint32); acc_crc = np.zeros((n, max_tx), np.uint32)
    tries = np.zeros((n, max_tx, 2), np.int32); fading = np.zeros((n, max_tx), np.float32)
    iters_sum = 0
    for t in range(max_tx):
        todo = np.nonzero(tts == 0)[0]
        if len(todo) == 0:
            break
        td = torch.from_numpy(todo).to(dev)
        x = clean[td].contiguous()
        e.channel_exact_seeded_(x, kind, snr_db, seeds[todo, t])
        llr, st = e.mcdpsk_demod(x, carriers, bps, spreading)
        fading[todo, t] = st["fading_index"]
        soft = llr[:, :648].contiguous()
        o, ok, it, tr = e.ldpc_decode_robust(soft)
        ok_h, tr_h = ok.cpu().numpy().astype(bool), tr.cpu().numpy()
        iters_sum += int(it.to(torch.int64).sum().item())
        tries[todo, t, 0] = tr_h
        out_h = o.cpu().numpy()
        if want_crc:
            sh = soft.cpu().numpy()
            llr_crc[todo, t] = [zlib.crc32(sh[q].tobytes()) for q in range(len(todo))]
        failed = np.nonzero(~ok_h)[0]
        if len(failed):
            fd = td[torch.from_numpy(failed).to(dev)]
            a, c = acc[fd].contiguous(), cnt[fd].contiguous()
            e.chase_combine(a, c, soft[torch.from_numpy(failed).to(dev)].contiguous())   # ChaseCache::store: first reception copies, later ones add
            acc[fd], cnt[fd] = a, c
            if want_crc:
                ah = a.cpu().numpy()
                acc_crc[todo[failed], t] = [zlib.crc32(ah[q].tobytes()) for q in range(len(failed))]
            if t > 0:                                                          # getCombineCount > 1
                o2, ok2, it2, tr2 = e.ldpc_decode_robust(a)
                iters_sum += int(it2.to(torch.int64).sum().item())
                ok2_h = ok2.cpu().numpy().astype(bool)
                tries[todo[failed], t, 1] = tr2.cpu().numpy()
                ok_h[failed] = ok2_h
                out_h[failed[ok2_h]] = o2.cpu().numpy()[ok2_h]
        won = np.nonzero(ok_h)[0]
        tts[todo[won]] = t + 1
        decoded[todo[won]] = out_h[won][:, :20]
    res = {"tx_to_success": tts, "decoded": decoded, "fading": fading, "tries": tries, "iterations_sum": iters_sum}
    if want_crc:
        res["llr_crc"], res["acc_crc"] = llr_crc, acc_crc
    return res


def trial_payloads(base_seed, point_index, trials, n_payloads=32):
    """info bytes of every trial: one of n_payloads 162-bit messages of this point, picked by a hash of the global trial index"""
    rng = np.random.default_rng([int(base_seed) & 0x7fffffff, int(point_index), 0x51])
    pool = rng.integers(0, 256, (n_payloads, 21), dtype=np.uint8)
    pool[:, -1] &= 0xC0                                   # k = 162 bits: the 21st byte carries 2 information bits
    pick = (trial_seed32(base_seed, point_index, 255, trials) % np.uint32(n_payloads)).astype(np.int64)
    return pool[pick]


def measured_fading_index(engines, point, base_seed, point_index, n_probe=48):
    """Fading index the ladder is fed with: the demodulator's own (MultiCarrierDPSKDemodulator::getFadingIndex, what the
    host reads after the handshake frames), median over a fixed probe set of MC-DPSK DBPSK frames through this point's
    channel.  The probe set is a function of (seed, point) only, so every rank computes the same value."""
    e = engines("DQPSK", "R1_4")
    probe = np.arange(n_probe)
    info = trial_payloads(base_seed, point_index, probe, 4)
    seeds = trial_seed32(base_seed, point_index, 254, probe)[:, None]
    uniq, inv = np.unique(info, axis=0, return_inverse=True)
    coded = e.ldpc_encode(uniq)
    frames = e.mcdpsk_modulate_batch(coded, 10, 1, 1)
    x = frames[torch.from_numpy(np.asarray(inv).reshape(-1)).to(e.device)].contiguous()
    e.channel_exact_seeded_(x, point.channel, point.snr_db, seeds[:, 0])
    _, st = e.mcdpsk_demod(x, 10, 1, 1)
    return float(np.median(st["fading_index"]))


def run_ladder_chunk(engines, point, base_seed, point_index, start, n, max_tx=4, n_payloads=32, fading=None):
    """One chunk of trials of one (channel, SNR) point with the mode the ladder picks.
    engines: callable (mod_name, rate_name) -> RxEngine (cached by the caller).  fading: the fading index handed to
    recommendWaveformAndRate (None: measured by the demodulator on this point's probe set)."""
    from . import capi
    lib = capi.load()
    if fading is None:
        fading = measured_fading_index(engines, point, base_seed, point_index)
    rec = ladder_mode(lib, point.snr_db, fading)
    mod, rate = _MOD_NAMES[rec.modulation], _RATE_NAMES[rec.code_rate]
    e = engines(mod, rate)
    if rec.waveform != 4:   # OFDM-CHIRP rung: frame = 4 codewords, single shot
        row = run_point_gpu(e, point, base_seed, point_index, start, n)
        return np.array([row[0], row[1], row[2], row[3], row[4], n], dtype=np.int64)
    # MC-DPSK rung: one R1/4 data codeword (162 information bits) per trial, 10 carriers, HARQ chase combining
    e = engines("DQPSK", "R1_4") if rate != "R1_4" else e
    bps = 1 if rec.modulation == 0 else 2
    trials = np.arange(start, start + n)
    info = trial_payloads(base_seed, point_index, trials, n_payloads)
    seeds = np.stack([trial_seed32(base_seed, point_index, t, trials) for t in range(max_tx)], axis=1)
    r = run_harq_trials(e, int(rec.num_carriers), bps, int(rec.spreading), point.channel, point.snr_db, info, seeds)
    good = (r["tx_to_success"] > 0) & (r["decoded"] == info[:, :20]).all(axis=1)
    transmissions = int(np.where(r["tx_to_success"] > 0, r["tx_to_success"], max_tx).sum())
    fail = int((~good).sum())
    byte_err = int((r["decoded"] != info[:, :20]).sum())
    return np.array([n, fail, fail, byte_err, r["iterations_sum"], transmissions], dtype=np.int64)


# ------------------------------------------------------------------------------------------------------------
# Acquisition grid (BASELINE.json config 4): ZC + dual-chirp preambles in noise over a CFO x SNR grid.
#
# Every buffer is built ON THE DEVICE by the library's reference-identical impairments, from a recipe that is a
# function of global identifiers only (seed, grid point, kind, global preamble index), so that (a) a CPU checker can
# rebuild the very same buffer from the recipe (the tests do, with the CPU checker) and compare every sample and every
# detector field bit for bit, and (b) the counters do not depend on chunking or on the number of GPUs:
#   preamble (ria_gpu_zc_preamble / ria_gpu_chirp_preamble, the reference generators' audio)
#   -> cfo_model "tx":        SimulatedChannel::applyTxCFO of the transmission (tools/cli_simulator.cpp:298-341, the
#                             analytic-signal rotation SURVEY.md 8d names for config 4; ria_gpu_tx_cfo_batch), placed at
#                             `offset` in a silent buffer, then the AWGN WattersonChannel (noise sigma from the rms of
#                             the non-zero samples; ria_gpu_channel_exact_seeded_batch with the recipe's mt19937 seed)
#   -> cfo_model "watterson": the clean preamble placed in the buffer, then the AWGN WattersonChannel with
#                             Config::cfo_hz = the grid CFO (its own applyCFO, hf_channel.hpp:182-241;
#                             ria_gpu_channel_exact_cfo_batch).
def acq_recipe(seed, grid_index, kind_index, indices, max_off):
    """(offsets int64, mt19937 channel seeds uint32) of the preambles `indices` (global numbers) of one (grid point, kind)"""
    idx = np.asarray(indices, dtype=np.uint64)
    point = int(grid_index) * 8 + int(kind_index)
    seeds = trial_seed32(seed, point, 0, idx)
    offs = (trial_seed32(seed, point, 1, idx).astype(np.int64)) % np.int64(max_off + 1)
    return offs, seeds


def make_acq_buffers(engine, pre, buf_len, offs, seeds, snr_db, cfo_hz, cfo_model="tx"):
    """pre: 1-D preamble tensor on the engine's device; offs / seeds: the recipe of each buffer (acq_recipe).
    Returns float32 [n, buf_len] on the device."""
    dev = pre.device
    n, L = len(offs), pre.numel()
    seg = pre[None, :].contiguous()
    if cfo_model == "tx":
        seg = engine.tx_cfo(seg, float(cfo_hz))
    buf = torch.zeros((n, buf_len), dtype=torch.float32, device=dev)
    offs_t = torch.from_numpy(np.ascontiguousarray(offs, np.int64)).to(dev)
    buf.scatter_(1, offs_t[:, None] + torch.arange(L, device=dev)[None, :], seg.expand(n, -1).contiguous())   # placement only
    if cfo_model == "tx":
        engine.channel_exact_seeded_(buf, 0, snr_db, seeds)
    else:
        engine.channel_exact_cfo_(buf, 0, snr_db, seeds, cfo_hz=float(cfo_hz))
    return buf


ACQ_GRID = [(c, s) for c in (-50.0, -25.0, 0.0, 25.0, 50.0) for s in (-10.0, -5.0, 0.0, 5.0, 10.0)]
ACQ_COUNTERS = ("n", "zc_detected", "zc_timing_ok", "chirp_success", "chirp_timing_ok", "chirp_cfo_ok")
ZC_ROOT_MASK_ALL, ZC_THRESHOLD, CHIRP_THRESHOLD = 15, 0.3, 0.15


def acq_tally(kind, res, offs, pre_len, cfo_hz):
    """Counter contributions of one chunk from the detector's result records: (n, detected / success, timing ok[, cfo ok])"""
    offs = np.asarray(offs, np.int64)
    if kind == "zc":
        det = res["detected"].astype(bool)
        ok = np.abs(res["start_sample"].astype(np.int64) - (offs + pre_len)) <= 4
        return np.array([len(offs), det.sum(), (det & ok).sum(), 0, 0, 0], np.int64)
    suc = res["success"].astype(bool)
    ok = np.abs(res["up_chirp_start"].astype(np.int64) - offs) <= 2
    cok = np.abs(res["cfo_hz"] - np.float32(cfo_hz)) <= 1.0
    return np.array([0, 0, 0, suc.sum(), (suc & ok).sum(), (suc & cok).sum()], np.int64)


def run_acquisition_grid(engine, dev, cdev, preambles, seed, kinds=None, grid=None, sync=None, cfo_model="tx"):
    """Preambles are dealt to the ranks in fixed-size chunks (chunk c of a (grid point, kind) goes to rank c % world); the
    recipe of a buffer depends on its global index only, so the counters do not depend on the number of ranks.  The only
    collectives are the all-reduce of the counters and of the wall times (cdev: where those few bytes live - the GPU for
    RCCL, the CPU for gloo).
    kinds: [(name, preamble tensor, buffer length, largest offset, chunk)]; returns (counters [len(grid), 6], [t_zc, t_chirp])."""
    import time
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    grid = ACQ_GRID if grid is None else grid
    sync = sync or (lambda: None)
    cnt = np.zeros((len(grid), 6), np.int64)
    t_zc = t_ch = 0.0
    for gi, (cfo, snr) in enumerate(grid):
        for ki, (kind, pre, buf_len, max_off, chunk) in enumerate(kinds):
            for start, n in shard_range(preambles, rank, world, chunk):
                offs, seeds = acq_recipe(seed, gi, ki, np.arange(start, start + n), max_off)
                buf = make_acq_buffers(engine, pre, buf_len, offs, seeds, snr, cfo, cfo_model)
                sync()
                t0 = time.perf_counter()
                if kind == "zc":
                    # ZC alone is unambiguous to +-23.6 Hz (zc_sync.hpp:55-58): the chirp's CFO is handed to it as known_cfo
                    r = engine.sync_zc(buf, ZC_THRESHOLD, ZC_ROOT_MASK_ALL, torch.full((n,), cfo, dtype=torch.float32, device=dev))
                    t_zc += time.perf_counter() - t0
                else:
                    r = engine.sync_chirp(buf, CHIRP_THRESHOLD)
                    t_ch += time.perf_counter() - t0
                cnt[gi] += acq_tally(kind, r, offs, pre.numel(), cfo)
    cnt = torch.from_numpy(cnt).to(cdev)
    tt = torch.tensor([t_zc, t_ch], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return cnt.cpu().numpy(), tt.cpu().numpy()
