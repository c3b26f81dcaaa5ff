#!/usr/bin/env python3
"""BASELINE.json config 5: adaptive-ladder BER/FER sweep (MC-DPSK 4x spread ... OFDM QAM64 R3/4), SNR -14..30 dB,
HARQ chase combining on the MC-DPSK rungs.  Trials are sharded over the ranks with no data-path collective
(ria_amd/sweep.py); RCCL only broadcasts the descriptor and all-reduces the counters.

  python tools/run_ladder_sweep.py --trials 2048                     # one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_ladder_sweep.py --trials 65536
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from ria_amd import sweep
from ria_amd.engine import RxEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=1024)
    ap.add_argument("--snr-min", type=float, default=-14.0)
    ap.add_argument("--snr-max", type=float, default=30.0)
    ap.add_argument("--snr-step", type=float, default=2.0)
    ap.add_argument("--channels", type=str, default="0,1,2", help="0 awgn 1 good 2 moderate")
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--chunk", type=int, default=512)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    cache = {}

    def engines(mod, rate):
        if (mod, rate) not in cache:
            cache[(mod, rate)] = RxEngine(mod, rate, device=local, max_batch=args.chunk)
        return cache[(mod, rate)]

    points = [sweep.SweepPoint(int(c), float(s)) for c in args.channels.split(",")
              for s in np.arange(args.snr_min, args.snr_max + 1e-6, args.snr_step)]
    t0 = time.perf_counter()
    rows = []
    for i in range(0, len(points), 64):   # the descriptor broadcast carries 64 points at a time
        part = points[i:i + 64]
        r, _ = sweep.run_sweep(part, args.trials, args.seed, lambda p, seed, pi, start, n: sweep.run_ladder_chunk(engines, p, seed, i + pi, start, n),
                               dev, chunk=args.chunk)
        rows.append(r)
    total = np.concatenate(rows)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if (dist.get_rank() if dist.is_initialized() else 0) == 0:
        from ria_amd import capi
        lib = capi.load()
        table = []
        for p, r in zip(points, total):
            fad = sweep.measured_fading_index(engines, p, args.seed, len(table))
            rec = sweep.ladder_mode(lib, p.snr_db, fad)
            table.append({"channel": p.channel, "snr_db": p.snr_db, "measured_fading_index": round(fad, 3), "waveform": int(rec.waveform), "modulation": int(rec.modulation),
                          "code_rate": int(rec.code_rate), "spreading": int(rec.spreading), "frames": int(r[0]), "frame_err": int(r[1]),
                          "fer": round(float(r[1]) / max(1, int(r[0])), 5), "mean_transmissions": round(float(r[5]) / max(1, int(r[0])), 3)})
        print(json.dumps({"config": "adaptive ladder sweep with HARQ", "n_gpus": world, "trials_per_point": args.trials,
                          "points": len(points), "seconds": round(dt, 2), "trials_per_s": round(len(points) * args.trials / dt), "table": table}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
