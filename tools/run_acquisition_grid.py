#!/usr/bin/env python3
"""BASELINE.json config 4: ZC + dual-chirp acquisition over a CFO x SNR grid, preambles sharded over the ranks.
Synthetic buffers, built on the device by the library's reference-identical impairments (ria_amd/sweep.py): the
preamble (bit-identical to the reference's generator) shifted by the simulator's applyTxCFO (cli_simulator.cpp:298-341;
--cfo-model watterson: WattersonChannel's own cfo_hz / applyCFO instead), at a recipe-derived offset in a silent buffer
(ZC: 4 512 samples, chirp: 120 000), through the AWGN WattersonChannel.  Reports detection rate,
timing / CFO error and preambles/s.

  python tools/run_acquisition_grid.py --preambles 20000
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_acquisition_grid.py --preambles 1000000
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
    ap.add_argument("--preambles", type=int, default=20000, help="per grid point and per kind, over all ranks")
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--cfo-model", choices=("tx", "watterson"), default="tx")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    e = RxEngine("QAM16", "R1_2", device=local)
    zc = torch.from_numpy(e.zc_preamble(5)).to(dev)
    ch = torch.from_numpy(e.chirp_preamble()).to(dev)
    kinds = [("zc", zc, 4512, 2000, 32768), ("chirp", ch, 120000, 62400, 1024)]
    c, tt = sweep.run_acquisition_grid(e, dev, dev, args.preambles, args.seed, kinds, sync=torch.cuda.synchronize, cfo_model=args.cfo_model)
    grid = sweep.ACQ_GRID
    if rank == 0:
        total = int(c[:, 0].sum())
        table = [{"cfo_hz": g[0], "snr_db": g[1], "n": int(r[0]), "zc_detected": int(r[1]), "zc_timing_ok": int(r[2]),
                  "chirp_success": int(r[3]), "chirp_timing_ok": int(r[4]), "chirp_cfo_ok": int(r[5])} for g, r in zip(grid, c)]
        print(json.dumps({"config": "ZC + dual-chirp acquisition grid", "n_gpus": world, "preambles_per_point": int(c[0, 0]),
                          "zc_preambles_per_s": round(total / float(tt[0])), "chirp_preambles_per_s": round(total / float(tt[1])), "table": table}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
