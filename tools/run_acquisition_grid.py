#!/usr/bin/env python3
"""BASELINE.json config 4: ZC + dual-chirp acquisition over a CFO x SNR grid, preambles sharded over the ranks.
Synthetic buffers: the preamble (host-synthesised once, bit-identical to the reference's generator) at a random
offset in a noise-padded buffer (ZC: 4 512 samples, chirp: 120 000), CFO applied by analytic-signal rotation,
AWGN from the preamble's rms.  Reports detection rate, timing / CFO error and preambles/s.

  python tools/run_acquisition_grid.py --preambles 20000
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_acquisition_grid.py --preambles 1000000
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from ria_amd.engine import RxEngine


def analytic(x):
    X = torch.fft.fft(x.double(), dim=-1)
    n = x.shape[-1]
    h = torch.zeros(n, dtype=torch.float64, device=x.device)
    h[0] = 1; h[1:(n + 1) // 2] = 2
    if n % 2 == 0:
        h[n // 2] = 1
    return torch.fft.ifft(X * h, dim=-1)


def make_buffers(pre, n, buf_len, max_off, snr_db, cfo_hz, gen, dev):
    """pre: 1-D preamble tensor on dev.  Returns (buffers float32 [n, buf_len], offsets)."""
    L = pre.numel()
    seg = pre
    if cfo_hz != 0.0:
        t = torch.arange(L, device=dev, dtype=torch.float64) / 48000.0
        seg = (analytic(pre) * torch.exp(2j * np.pi * cfo_hz * t)).real.float()
    rms = pre[pre != 0].pow(2).mean().sqrt()
    sigma = rms * 10.0 ** (-snr_db / 20.0)
    buf = torch.randn((n, buf_len), generator=gen, device=dev) * sigma
    offs = torch.randint(0, max_off + 1, (n,), generator=gen, device=dev)
    idx = offs[:, None] + torch.arange(L, device=dev)[None, :]
    buf.scatter_add_(1, idx, seg[None, :].expand(n, -1).contiguous())
    return buf.contiguous(), offs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preambles", type=int, default=20000, help="per grid point and per kind, over all ranks")
    ap.add_argument("--seed", type=int, default=20261004)
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
    per_rank = (args.preambles + world - 1) // world
    grid = [(c, s) for c in (-50.0, -25.0, 0.0, 25.0, 50.0) for s in (-10.0, -5.0, 0.0, 5.0, 10.0)]
    # counters per grid point: n, zc detected, zc |start error| <= 4, chirp success, chirp |start error| <= 2, chirp |cfo error| <= 1 Hz
    cnt = torch.zeros((len(grid), 6), dtype=torch.int64, device=dev)
    t_zc = t_ch = 0.0
    for gi, (cfo, snr) in enumerate(grid):
        gen = torch.Generator(device=dev); gen.manual_seed(args.seed * 1000 + gi * 64 + rank)
        for kind, pre, buf_len, max_off, chunk in (("zc", zc, 4512, 2000, 32768), ("chirp", ch, 120000, 62400, 1024)):
            for start in range(0, per_rank, chunk):
                n = min(chunk, per_rank - start)
                # ZC alone is unambiguous to +-23.6 Hz (zc_sync.hpp:55-58): the chirp's CFO is handed to it as known_cfo
                buf, offs = make_buffers(pre, n, buf_len, max_off, snr, cfo, gen, dev)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                if kind == "zc":
                    r = e.sync_zc(buf, 0.3, 15, torch.full((n,), cfo, dtype=torch.float32, device=dev))
                    t_zc += time.perf_counter() - t0
                    det = torch.from_numpy(r["detected"].astype(np.int64)).to(dev)
                    ok = (torch.from_numpy(r["start_sample"].astype(np.int64)).to(dev) - (offs + 2512)).abs() <= 4
                    cnt[gi, 0] += n; cnt[gi, 1] += det.sum(); cnt[gi, 2] += (det.bool() & ok).sum()
                else:
                    r = e.sync_chirp(buf, 0.15)
                    t_ch += time.perf_counter() - t0
                    suc = torch.from_numpy(r["success"].astype(np.int64)).to(dev)
                    ok = (torch.from_numpy(r["up_chirp_start"].astype(np.int64)).to(dev) - offs).abs() <= 2
                    cok = torch.from_numpy(np.abs(r["cfo_hz"] - cfo) <= 1.0).to(dev)
                    cnt[gi, 3] += suc.sum(); cnt[gi, 4] += (suc.bool() & ok).sum(); cnt[gi, 5] += (suc.bool() & cok).sum()
    tt = torch.tensor([t_zc, t_ch], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        c = cnt.cpu().numpy(); total = int(c[:, 0].sum())
        table = [{"cfo_hz": g[0], "snr_db": g[1], "n": int(r[0]), "zc_detected": int(r[1]), "zc_timing_ok": int(r[2]),
                  "chirp_success": int(r[3]), "chirp_timing_ok": int(r[4]), "chirp_cfo_ok": int(r[5])} for g, r in zip(grid, c)]
        print(json.dumps({"config": "ZC + dual-chirp acquisition grid", "n_gpus": world, "preambles_per_point": int(c[0, 0]),
                          "zc_preambles_per_s": round(total / float(tt[0])), "chirp_preambles_per_s": round(total / float(tt[1])), "table": table}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
