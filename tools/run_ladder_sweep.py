#!/usr/bin/env python3
"""BASELINE.json config 5: adaptive-ladder BER/FER sweep (MC-DPSK 4x spread ... OFDM QAM64 R3/4), SNR -14..30 dB,
HARQ chase combining on the MC-DPSK rungs.  Trials are sharded over the ranks with no data-path collective
(ria_amd/sweep.py); RCCL only all-reduces the counters of a point (a few dozen bytes) when the point is complete.

  python tools/run_ladder_sweep.py --trials 65536 --out sweep.jsonl          # one GPU
  python tools/run_ladder_sweep.py --trials 65536 --out sweep.jsonl --resume # carries on after the last completed point
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_ladder_sweep.py --trials 1048576 --out sweep.jsonl

One JSON line per completed point goes to --out as soon as the point is done (rank 0), so an interrupted sweep keeps its
counters: a point's outcome is a function of (seed, point, global trial index) only, whatever the chunking, the number of
GPUs, or the session it ran in.  The last line printed on stdout is the summary: trials/s per rung and where the wall time went.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from ria_amd import capi, sweep
from ria_amd.engine import RxEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=16384)
    ap.add_argument("--snr-min", type=float, default=-14.0)
    ap.add_argument("--snr-max", type=float, default=30.0)
    ap.add_argument("--snr-step", type=float, default=2.0)
    ap.add_argument("--channels", type=str, default="0,1,2", help="0 awgn 1 good 2 moderate")
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--chunk", type=int, default=16384, help="trials per engine call")
    ap.add_argument("--out", type=str, default="", help="JSON-lines file: one line per completed point")
    ap.add_argument("--resume", action="store_true", help="skip the points --out already holds (same seed / trials)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    lib = capi.load()
    cache = {}

    def engines(mod, rate):   # one engine per (modulation, rate) for the whole run
        if (mod, rate) not in cache:
            cache[(mod, rate)] = RxEngine(mod, rate, device=local, max_batch=args.chunk)
        return cache[(mod, rate)]

    points = [sweep.SweepPoint(int(c), float(s)) for c in args.channels.split(",")
              for s in np.arange(args.snr_min, args.snr_max + 1e-6, args.snr_step)]
    done = {}
    if args.resume and args.out and os.path.exists(args.out):
        for line in open(args.out):
            try:
                r = json.loads(line)
            except ValueError:
                continue
            if r.get("seed") == args.seed and r.get("trials") == args.trials:
                done[r["point"]] = r
    t_run = time.perf_counter()
    t_probe = t_chunks = 0.0
    by_rung = {}
    rows = []
    for pi, p in enumerate(points):
        if pi in done:
            rows.append(done[pi]); continue
        t0 = time.perf_counter()
        fad = sweep.measured_fading_index(engines, p, args.seed, pi)          # the demodulator's own index: a function of (seed, point)
        rec = sweep.ladder_mode(lib, p.snr_db, fad)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        local_cnt = np.zeros(len(sweep.LADDER_COUNTERS), np.int64)
        for start, n in sweep.shard_range(args.trials, rank, world, args.chunk):
            local_cnt += sweep.run_ladder_chunk(engines, p, args.seed, pi, start, n, fading=fad)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tot = sweep.reduce_counters(local_cnt[None, :], dev)[0]
        t_probe += t1 - t0; t_chunks += t2 - t1
        rung = f"wf{int(rec.waveform)}_mod{int(rec.modulation)}_rate{int(rec.code_rate)}_x{int(rec.spreading)}"
        g = by_rung.setdefault(rung, [0, 0.0])
        g[0] += args.trials; g[1] += t2 - t1
        row = {"point": pi, "seed": args.seed, "trials": args.trials, "channel": p.channel, "snr_db": p.snr_db, "measured_fading_index": round(fad, 3),
               "waveform": int(rec.waveform), "modulation": int(rec.modulation), "code_rate": int(rec.code_rate), "spreading": int(rec.spreading),
               "frames": int(tot[0]), "frame_err": int(tot[1]), "fer": round(float(tot[1]) / max(1, int(tot[0])), 6),
               "mean_transmissions": round(float(tot[5]) / max(1, int(tot[0])), 4), "seconds": round(t2 - t0, 3)}
        rows.append(row)
        if rank == 0 and args.out:
            with open(args.out, "a") as f:
                f.write(json.dumps(row) + "\n")
    dt = time.perf_counter() - t_run
    if rank == 0:
        fresh = len(points) - len(done)
        print(json.dumps({"config": "adaptive ladder sweep with HARQ", "n_gpus": world, "trials_per_point": args.trials, "points": len(points),
                          "points_run_now": fresh, "seconds": round(dt, 2), "trials_per_s": round(fresh * args.trials / max(dt, 1e-9)),
                          "time_split_s": {"fading_probe_and_ladder": round(t_probe, 2), "trial_chunks": round(t_chunks, 2), "other": round(dt - t_probe - t_chunks, 2)},
                          "trials_per_s_by_rung": {k: round(v[0] / max(v[1], 1e-9)) for k, v in sorted(by_rung.items())},
                          "table": rows}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
