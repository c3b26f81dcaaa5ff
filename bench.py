#!/usr/bin/env python3
"""bench.py — frames/s demodulated + LDPC-decoded on MI355X (BASELINE.json metric).

One "step" = one pass of the RX hot path (ria_gpu_rx_batch: samples -> demod -> de-interleave ->
LDPC min-sum incl. the reference's retry cascade and CRC recovery -> payload bytes) over one batch
of synthetic frames that is already resident in HBM.  Workload at N=1: BASELINE.json configs[2]
"OFDM QAM16 R1/2 + LDPC min-sum, 100k frames, Watterson moderate fading" (the configuration the
metric is quoted on).  Frames are synthesised on the GPU by the library's own TX + channel kernels
(untimed; the channel kernel reproduces sim::WattersonChannel's random stream bit for bit, channel seed =
base + global frame index as in tools/test_waveform_simple.cpp).  N>1: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`; frames
are sharded over ranks (weak scaling, per-GPU batch fixed), RCCL only broadcasts the seed and
all-reduces counters/timing.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_FRAME = 18432 * 4 + 160          # samples in + payload out (fused path)
ALGO_BYTES_DEMOD = 18432 * 4 + 2632 * 4     # demod kernel alone: samples in + LLRs out
ALGO_BYTES_DECODE = 2592 * 4 + 160          # decode kernel alone: LLRs in + payload out
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec
CLOCK_HZ = 2.4e9                             # MI355X_MICROARCH.md: max clock
VALU_PEAK_INSTS = 256 * 4 * CLOCK_HZ / 2     # wave64 VALU instructions/s: 1024 SIMD-32 units, 2 cycles per wave64 instruction (the guide's
                                             # wave-scheduling section; equals its 157.3 TFLOP/s vector peak); ONE wave alone issues every 4
LDS_PEAK_CYCLES = 256 * CLOCK_HZ             # LDS-array cycles/s: one array per CU (SQ_LDS_IDX_ACTIVE counts its busy cycles)
RECOVERY_STAGE = ("recovery_list_kernel", "recovery_stage1_kernel", "recovery_fill_kernel", "recovery_stage2_kernel", "decode_fault_kernel")
DEMOD_STAGE = ("demod_fft_kernel", "demod_decide_kernel", "demod_walk_kernel", "demod_est_kernel", "demod_frames_kernel")
DECODE_STAGE = ("fast_primary_kernel", "fast_mark_kernel", "fast_stage_kernel", "fast_phase0_kernel", "fast_chain_kernel",
                "fast_cascade_kernel", "fast_finalize_kernel", "frame_validate_kernel", "dual_phase0_kernel", "dual_cascade_kernel")


def host_cores():
    """CPUs this process may use: the affinity mask, capped by the cgroup CPU quota where one is set."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def matching_profile(suffix, src_hash):
    """newest profiles/*<suffix> whose recorded kernel-source hash is the current one, or (None, reason)"""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)), key=os.path.getmtime, reverse=True)
    seen = 0
    for p in cands:
        try:
            d = json.load(open(p))
        except (OSError, ValueError):
            continue
        meta = d.get("_meta")
        if not meta:
            continue
        seen += 1
        if meta.get("source_sha256") == src_hash:
            return d, os.path.relpath(p, ROOT)
    return None, (f"no profiles/*{suffix} was measured on the current kernel sources (sha256 {src_hash[:12]}; "
                  f"{seen} tagged profile(s) are of other builds)")


def cpu_baseline(frames_host, seconds_budget=20.0):
    """Reference CPU path on the host cores, bounded sample of the same workload (rank 0, N=1 only).
    Prefers the compiled unmodified reference (oracle/_ref, kind 'reference'): one OFDMChirpWaveform per thread, configured
    once and reset() before every frame as gui::StreamingDecoder does (streaming_decoder.cpp:723; SURVEY.md 8d "one instance
    per std::thread"), then process -> getSoftBits -> v2::decodeFixedFrame, all inside one C call per frame.  Falls back to the
    C restatement (kind 'port').  The oracle is used here only as the timed baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    cores = host_cores()   # all host cores this process may use, count stated in the result
    kind = "port"
    ref = None
    if po.Ref.available():
        try:
            ref = po.Ref()
            kind = "reference"
        except OSError:
            ref = None
    orc = po.Oracle()

    def run_one(handle, x):
        if ref is not None:
            ref.rx_frame(handle, x)
        else:
            llr, _ = orc.rx_process(po.QAM16, po.R1_2, x)
            orc.decode_fixed_frame(llr, po.R1_2, True, 188, flags=7)

    warm = ref.rx_open(po.QAM16, po.R1_2) if ref is not None else None
    run_one(warm, frames_host[0])  # static-table warm-up before threading (frame_interleaver.cpp:13-48)
    if ref is not None:
        ref.rx_close(warm)
    n = len(frames_host)
    nxt = [0]
    done = [0]
    lock = threading.Lock()
    t_end = [0.0]

    def worker():
        handle = ref.rx_open(po.QAM16, po.R1_2) if ref is not None else None   # this thread's own waveform object
        while True:
            with lock:
                i = nxt[0]
                if i >= n or time.perf_counter() > t_end[0]:
                    break
                nxt[0] += 1
            run_one(handle, frames_host[i])
            with lock:
                done[0] += 1
        if ref is not None:
            ref.rx_close(handle)

    t0 = time.perf_counter()
    t_end[0] = t0 + seconds_budget
    ths = [threading.Thread(target=worker) for _ in range(cores)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": round(done[0] / dt, 2), "unit": "frames/s", "cores": cores, "kind": kind,
            "sample": f"{done[0]} frames from the head of the timed workload's last batch (QAM16 R1/2, Watterson moderate 20 dB), "
                      f"{dt:.1f} s wall on {cores} threads, one kept OFDMChirpWaveform per thread (reset per frame), process + full "
                      f"decodeFixedFrame incl. retry cascade and CRC recovery"}


def main(argv=None, engine_factory=None):
    """engine_factory: tests/test_bench_gloo.py passes a CPU stand-in for RxEngine to run this function's N > 1 control
    flow (seed broadcast, global frame indices, barrier, max-over-ranks time, counter all-reduce, rank-0 JSON line) with
    world_size 2 on gloo; None = the real engine on this rank's GPU."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=100000, help="frames per step per GPU (BASELINE.json configs[2]: 100k frames)")
    ap.add_argument("--channel", type=int, default=2, help="0 awgn 1 good 2 moderate 3 poor 4 flutter")
    ap.add_argument("--snr", type=float, default=20.0)
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--steps-only", action="store_true",
                    help="profiling passes: only the warm-up + timed steps run on the GPU (no per-kernel timing section, no CPU leg), "
                         "so that a counter summed over the run divides by warmup + steps")
    args = ap.parse_args(argv)
    stub = engine_factory is not None
    if stub:
        args.steps_only = True

    import torch
    import torch.distributed as dist
    from ria_amd import capi
    from ria_amd.engine import RxEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RIA_BENCH_REHEARSE=1: every rank on cuda:0 with the gloo backend - a rehearsal of the N > 1 control flow on a
    # one-GPU box (the driver's real runs use one GPU per rank over RCCL)
    rehearse = os.environ.get("RIA_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    own_group = False
    # RIA_BENCH_FORCE_DIST=1: initialise the process group even for ONE rank, so that the RCCL calls of the N > 1 path
    # (broadcast of the seed, all-reduce of counters and time, barrier) run on real hardware on a one-GPU box
    force_dist = os.environ.get("RIA_BENCH_FORCE_DIST") == "1"
    if force_dist and world == 1 and not stub and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        own_group = True
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        own_group = True
        if stub or rehearse:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cpu") if stub else torch.device("cuda", local)
    if not stub:
        torch.cuda.set_device(dev)
    device_sync = (lambda: None) if stub else torch.cuda.synchronize
    cdev = torch.device("cpu") if (rehearse or stub) else dev      # where the few-byte collectives live

    # seed broadcast (the only data-path-adjacent collective: tens of bytes over xGMI)
    collectives = world > 1 or (dist.is_available() and dist.is_initialized())
    seed_t = torch.tensor([args.seed], dtype=torch.int64, device=cdev)
    if collectives:
        dist.broadcast(seed_t, 0)
    seed = int(seed_t.item())

    B = args.batch
    e = engine_factory(B) if stub else RxEngine("QAM16", "R1_2", device=local, max_batch=B)
    n_sets = args.steps + args.warmup
    # every step gets its own input batch up to 16 resident batches (16 x 7.4 GB at the default size); longer runs cycle
    # through them - the work per step is the same, nothing is cached between steps
    n_pool = min(n_sets, 16)
    batches, infos = [], []
    for s in range(n_pool):
        first = (s * world + rank) * B                   # global frame index: results independent of N
        info = e.make_frames(seed, first, B)
        x = e.tx(info, peak=0.8)
        e.channel_exact_(x, args.channel, args.snr, seed, first_frame=first)   # the reference's own mt19937 stream: seed + frame
        batches.append(x)
        infos.append(info)
    device_sync()
    out = (torch.empty((B, e.geo.info_bytes_per_frame), dtype=torch.uint8, device=dev),
           torch.zeros((B, 20), dtype=torch.uint8, device=dev))

    def barrier():
        device_sync()
        if collectives:
            dist.barrier()
        device_sync()

    for s in range(args.warmup):
        e.rx(batches[s % n_pool], out=out)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.warmup, n_sets):
        e.rx(batches[s % n_pool], out=out)
    barrier()
    elapsed = time.perf_counter() - t0

    # correctness counters of the last step (outside the timed region)
    st = e.decode_status(out[1])
    frames_ok = int((st["cw_ok"].all(axis=1) & st["frame_valid"].astype(bool)).sum())
    bytes_ok = int((out[0] == infos[(n_sets - 1) % n_pool]).all(dim=1).sum().item())
    cnt = torch.tensor([B * args.steps, frames_ok, bytes_ok, int(st["iterations"].sum())], dtype=torch.int64, device=cdev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if collectives:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    total_frames = int(cnt[0].item())

    # live per-kernel durations with HIP events on the stream the kernels are launched on
    # (torch's current stream is the stream handed to the C ABI).
    t_demod = t_decode = None
    if not args.steps_only:
        x = batches[-1]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        llr, _ = e.demod(x, want_status=False)
        torch.cuda.synchronize()
        reps = 3
        for rep in range(reps + 1):          # the first pass is a warm-up (allocator, clocks) and is not counted
            if rep <= 1:
                t_demod = t_decode = 0.0
            ev[0].record()
            llr, _ = e.demod(x, want_status=False)
            ev[1].record()
            ev[2].record()
            e.decode(llr, flags=capi.DECODE_FULL)     # everything the fused call runs after the demodulator, CRC recovery included
            ev[3].record()
            torch.cuda.synchronize()
            t_demod += ev[0].elapsed_time(ev[1]) / reps
            t_decode += ev[2].elapsed_time(ev[3]) / reps
        del llr
    roof = None
    if t_decode is not None:
        step_s = elapsed / args.steps
        dec = t_decode >= t_demod
        dom = ("decode stage (fast_primary/mark/stage/phase0/chain/cascade/finalize + frame_validate + recovery_* kernels: one "
               "ria_gpu_decode_batch call)") if dec else "demodulator (demod_fft/decide/walk/est kernels: one ria_gpu_demod_batch call)"
        dur_ms, algo = (t_decode, ALGO_BYTES_DECODE * B) if dec else (t_demod, ALGO_BYTES_DEMOD * B)
        stage = (DECODE_STAGE + RECOVERY_STAGE) if dec else DEMOD_STAGE
        hbm_achieved = algo / (dur_ms * 1e-3) / 1e9
        # HBM traffic, LDS-array cycles and instruction counts cannot be measured from inside the process: they come from the
        # rocprofv3 --pmc passes of THIS command (tools/measure_round.sh) summarised under profiles/ - and only from a summary
        # whose recorded kernel-source hash equals the sources this run was built from; otherwise null with the reason.
        from ria_amd.srchash import csrc_sha256
        src = csrc_sha256()
        traffic = traffic_all = traffic_src = lds = valu_issue = None
        in_stage = lambda k, names: k != "_meta" and any(n in k for n in names)   # noqa: E731
        tr, tr_path = matching_profile("_hbm_traffic_pmc.json", src)
        if tr is not None and tr["_meta"].get("frames_per_launch") == B:
            traffic = int(sum(v["hbm_bytes_per_launch"] * v.get("launches_per_step", 1) for k, v in tr.items() if in_stage(k, stage)))
            traffic_all = int(sum(v["hbm_bytes_per_launch"] * v.get("launches_per_step", 1) for k, v in tr.items() if in_stage(k, DEMOD_STAGE + DECODE_STAGE + RECOVERY_STAGE)))
            traffic_src = tr_path + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, RIA_NO_SPLIT: one launch = one whole step; FETCH_SIZE x2 for gfx950)"
        else:
            traffic_src = tr_path if tr is None else f"{tr_path} was measured at {tr['_meta'].get('frames_per_launch')} frames per launch, this run at {B}"
        sq, sq_path = matching_profile("_sq_utilisation_pmc.json", src)
        if sq is not None and sq["_meta"].get("frames_per_step") == B:
            m = sq["_meta"]
            per_step = lambda field, names: sum(v[field] * v["launches"] for k, v in sq.items() if in_stage(k, names)) / m["steps_counted"]   # noqa: E731
            dec_insts = per_step("valu_insts", DECODE_STAGE + RECOVERY_STAGE)
            all_insts = per_step("valu_insts", DECODE_STAGE + RECOVERY_STAGE + DEMOD_STAGE)
            valu_issue = {"peak_insts_per_s": VALU_PEAK_INSTS, "unit": "wave64 VALU instructions (SQ_INSTS_VALU)",
                          "decode_stage": {"insts_per_step": int(dec_insts), "ms": round(t_decode, 3),
                                           "frac": round(dec_insts / (t_decode * 1e-3) / VALU_PEAK_INSTS, 4)},
                          "fused_step": {"insts_per_step": int(all_insts), "ms": round(step_s * 1e3, 3),
                                         "frac": round(all_insts / step_s / VALU_PEAK_INSTS, 4)},
                          "model": "2 cycles per wave64 VALU instruction per SIMD-32 (MI355X_MICROARCH.md wave scheduling = the 157.3 TFLOP/s "
                                   "vector peak); a single wave alone issues one per 4 cycles (frac_single_wave_model = 2 x frac)",
                          "source": sq_path + " (instruction counts; the times are this run's)"}
            if "lds_idx_active" in next(v for k, v in sq.items() if k != "_meta"):
                st_names = stage
                idx = per_step("lds_idx_active", st_names)
                conf = per_step("lds_bank_conflict", st_names)
                c = next((v for k, v in sq.items() if "fast_cascade_kernel" in k), None)
                lds = {"idx_active_cycles_per_step": int(idx), "bank_conflict_cycles_per_step": int(conf),
                       "bank_conflict_share": round(conf / max(idx, 1.0), 4),
                       "achieved_Gcycles_per_s": round(idx / (dur_ms * 1e-3) / 1e9, 2), "peak_Gcycles_per_s": LDS_PEAK_CYCLES / 1e9,
                       "frac": round(idx / (dur_ms * 1e-3) / LDS_PEAK_CYCLES, 4),
                       "cascade_kernel_alone": None if c is None else {"lds_busy_frac": c["lds_busy_frac"], "lds_bank_conflict_share": c["lds_bank_conflict_share"],
                                                                      "valu_issue_frac_2cyc": c.get("valu_issue_frac_2cyc")},
                       "source": sq_path + " (SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT summed over the stage's kernels; the time is this run's)"}
        else:
            valu_issue = {"source": sq_path if sq is None else f"{sq_path}: other batch size"}
        hbm = {"kernel": dom, "achieved": round(hbm_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_achieved / HBM_PEAK_GBS, 6),
               "algorithmic_bytes_per_launch": algo, "traffic": traffic}
        fused_gbs = ALGO_BYTES_FRAME * B / step_s / 1e9
        hbm_fused = {"kernel": "whole fused step (ria_gpu_rx_batch: samples in, payload out)", "achieved": round(fused_gbs, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(fused_gbs / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_step": ALGO_BYTES_FRAME * B, "traffic": traffic_all}
        if dec and lds is not None:
            # what binds the dominant stage is the LDS array (gathers / stores of the min-sum messages), not HBM or MFMA: the
            # roofline object states THAT unit; the HBM figures the contract defines stay beside it (hbm, hbm_fused_step)
            roof = {"bound": "lds", "kernel": dom, "achieved": lds["achieved_Gcycles_per_s"], "peak": lds["peak_Gcycles_per_s"],
                    "unit": "G LDS-array cycles/s (256 CUs x 2.4 GHz)", "frac": lds["frac"], "traffic": traffic}
        else:
            roof = {"bound": "hbm", "kernel": dom, "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac"], "traffic": traffic}
        roof.update({"traffic_source": traffic_src, "hbm": hbm, "hbm_fused_step": hbm_fused, "lds": lds, "valu_issue": valu_issue,
                     "kernel_ms": {"demod_kernels": round(t_demod, 3), "decode_kernels": round(t_decode, 3)},
                     "limiter": "LDS array + FP32 VALU issue, not HBM (SURVEY.md 8d: the fused chain's compulsory traffic is 0.1 % of HBM peak at 1 M frames/s)",
                     "kernel_source_sha256": src})

    if rank == 0:
        res = {
            "metric": "frames/s demod+LDPC-decoded, OFDM QAM16 R1/2 1024-FFT",
            "value": round(total_frames / elapsed, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "OFDM QAM16 R1/2 + LDPC min-sum, Watterson moderate fading 20 dB, "
                            f"{total_frames} frames ({B} per step per GPU), full decodeFixedFrame",
                "frame_samples": int(e.geo.frame_samples), "batch_per_gpu": B,
                "channel": args.channel, "snr_db": args.snr,
                "frames_decoded_last_step": int(cnt[1].item()), "frames_bytes_equal_tx_last_step": int(cnt[2].item()),
                "fused_path_GBps": round(total_frames / elapsed * ALGO_BYTES_FRAME / 1e9, 2),
            },
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline and not args.steps_only:
            # bounded sample of the same workload: the last batch, as many of its frames as the host cores finish in ~20 s
            res["cpu_baseline"] = cpu_baseline(batches[-1][:24576].cpu().numpy(), seconds_budget=20.0)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if own_group:
        dist.destroy_process_group()
    return res if rank == 0 else None


if __name__ == "__main__":
    main()
