"""CPU, world_size 2 on gloo: the N > 1 control flow of bench.py itself and of the config-4 acquisition-grid driver
(ria_amd/sweep.py run_acquisition_grid, what tools/run_acquisition_grid.py runs), with the GPU engine replaced by a
deterministic CPU stand-in whose outputs are functions of the GLOBAL frame / buffer content only.  What must hold for
N > 1: every rank works on its own global indices, rank 0's seed is the one used, the counters all-reduce to the sums
of the 1-rank runs over the same global indices, the time is the max over ranks, only rank 0 prints.
(The config-5 driver's sharding is sweep.run_sweep, covered by tests/test_sweep_gloo.py.)"""
import json
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Geo:
    info_bytes_per_frame, frame_samples = 160, 16


DSTAT = np.dtype([("cw_ok", "u1", 4), ("iterations", "<u2", 4), ("attempts", "u1", 4), ("frame_valid", "u1"),
                  ("needs_recovery", "u1"), ("reserved", "u1", 2)])


class StubEngine:
    """RxEngine stand-in: frame f of seed s carries bytes hash(s, f); the 'channel' keeps the global frame index in the
    samples; rx 'decodes' a frame unless hash(seed, global index) % 5 == 0."""
    geo = Geo()

    def __init__(self, batch):
        self.batch = batch
        self.rx_calls = []

    @staticmethod
    def _h(seed, idx):
        return (idx.astype(np.int64) * 2654435761 + seed * 40503) % 1000003

    def make_frames(self, seed, first, n):
        idx = np.arange(first, first + n)
        b = (self._h(seed, idx)[:, None] + np.arange(160)[None, :]) % 251
        return torch.from_numpy(b.astype(np.uint8))

    def tx(self, info, peak=0.8):
        return torch.zeros((info.shape[0], 16), dtype=torch.float32)

    def channel_exact_(self, x, kind, snr, seed, first_frame=0):
        x[:, 0] = torch.arange(first_frame, first_frame + x.shape[0], dtype=torch.float32)
        x[:, 1] = float(seed % 1000)
        self.seed = seed
        return x

    def rx(self, x, out=None):
        idx = x[:, 0].numpy().astype(np.int64)
        self.rx_calls.append((int(idx[0]), len(idx)))
        good = (self._h(self.seed, idx) % 5) != 0
        info = self.make_frames(self.seed, int(idx[0]), len(idx))
        info[torch.from_numpy(~good)] = 0
        st = np.zeros(len(idx), DSTAT)
        st["cw_ok"][good] = 1
        st["frame_valid"][good] = 1
        st["iterations"][:] = (idx % 7)[:, None].astype(np.uint16)
        out[0].copy_(info)
        out[1].copy_(torch.from_numpy(st.view(np.uint8).reshape(len(idx), 20)))
        return out

    def decode_status(self, st):
        return st.numpy().view(DSTAT).reshape(-1)


def _bench_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench
    made = []

    def factory(batch):
        made.append(StubEngine(batch))
        return made[-1]
    # rank 1 is started with a different --seed: only rank 0's may count
    res = bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "500", "--seed", str(777 if rank == 0 else 5)], factory)
    q.put((rank, res, made[0].rx_calls, made[0].seed))


def _spawn(target, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


def test_bench_control_flow_world_size_2():
    (r0, res0, calls0, seed0), (r1, res1, calls1, seed1) = _spawn(_bench_worker, 2)
    assert res1 is None and res0 is not None, "only rank 0 reports"
    assert seed0 == seed1 == 777, "rank 0's seed is broadcast"
    B, steps, warm = 500, 3, 1
    # global frame indices: step s of rank r starts at (s * world + r) * B - disjoint, complete, independent of the rank count
    assert [c[0] for c in calls0] == [(s * 2 + 0) * B for s in range(steps + warm)]
    assert [c[0] for c in calls1] == [(s * 2 + 1) * B for s in range(steps + warm)]
    assert res0["n_gpus"] == 2 and res0["scaling"] == "weak" and res0["steps"] == steps and res0["warmup"] == warm
    assert res0["config"]["workload"].count(str(2 * B * steps)) == 1          # whole-job frames = all ranks
    assert abs(res0["value"] - 2 * B * steps / (res0["ms_per_step"] * steps * 1e-3)) < 0.05 * res0["value"]
    # counters of the last step: the sum over the two ranks' last batches
    exp_ok = 0
    for r in range(2):
        idx = np.arange(((steps + warm - 1) * 2 + r) * B, ((steps + warm - 1) * 2 + r) * B + B)
        exp_ok += int(((StubEngine._h(777, idx) % 5) != 0).sum())
    assert res0["config"]["frames_decoded_last_step"] == exp_ok == res0["config"]["frames_bytes_equal_tx_last_step"]
    assert res0["cpu_baseline"] is None and res0["vs_baseline"] is None
    json.dumps(res0)


class StubSync:
    """CPU stand-ins for the engine calls of the acquisition grid: the impairments keep the library's contract (a
    function of the recipe only: per-buffer seed, CFO), the detectors 'detect' from the buffer content only"""
    ZC = np.dtype([("detected", "<i4"), ("start_sample", "<i4")])
    CH = np.dtype([("success", "<i4"), ("up_chirp_start", "<i4"), ("cfo_hz", "<f4")])

    def tx_cfo(self, seg, cfo_hz):
        return seg * float(1.0 + cfo_hz / 1000.0)

    def channel_exact_seeded_(self, buf, kind, snr_db, seeds):
        s = torch.from_numpy((np.asarray(seeds, np.uint32) % 1000).astype(np.float32) / 1000.0)
        ramp = torch.arange(buf.shape[1], dtype=torch.float32)[None, :] % 7
        buf += (s[:, None] - 0.5) * 10.0 ** (-snr_db / 20.0) * (ramp - 3.0) * 0.05
        return buf

    def sync_zc(self, buf, thr, mask, cfo):
        r = np.zeros(buf.shape[0], self.ZC)
        e = buf.abs().sum(dim=1).numpy()
        r["detected"] = (e * 1000).astype(np.int64) % 3 != 0
        r["start_sample"] = buf.abs().argmax(dim=1).numpy() + 7
        return r

    def sync_chirp(self, buf, thr):
        r = np.zeros(buf.shape[0], self.CH)
        e = buf.abs().sum(dim=1).numpy()
        r["success"] = (e * 1000).astype(np.int64) % 4 != 0
        r["up_chirp_start"] = buf.abs().argmax(dim=1).numpy()
        r["cfo_hz"] = (e % 3).astype(np.float32)
        return r


def _acq(world_rank=None):
    from ria_amd import sweep
    dev = torch.device("cpu")
    pre_a = torch.sin(torch.arange(40, dtype=torch.float32) * 0.7) * 3
    pre_b = torch.cos(torch.arange(64, dtype=torch.float32) * 0.3) * 3
    kinds = [("zc", pre_a, 128, 60, 64), ("chirp", pre_b, 256, 100, 32)]
    grid = [(-25.0, 0.0), (0.0, 5.0), (25.0, 10.0)]
    return sweep.run_acquisition_grid(StubSync(), dev, dev, 300, 99, kinds, grid=grid)


def _acq_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c, tt = _acq()
    q.put((rank, c, tt))
    dist.destroy_process_group()


def test_acquisition_grid_counters_do_not_depend_on_the_rank_count():
    single, _ = _acq()
    assert single[:, 0].tolist() == [300, 300, 300] and (single[:, 1] > 0).all() and (single[:, 3] > 0).all()
    res = _spawn(_acq_worker, 2)
    for rank, c, tt in res:
        assert np.array_equal(c, single), f"rank {rank}: reduced counters differ from the 1-process run"
    res3 = _spawn(_acq_worker, 3)
    assert np.array_equal(res3[0][1], single)
