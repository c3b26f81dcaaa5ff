"""ctypes bindings for the CPU checkers.  TEST INFRASTRUCTURE ONLY.

`Oracle`  -> oracle/libria_oracle.so  (our C restatement, travels everywhere)
`Ref`     -> oracle/_ref/libria_ref.so (the compiled unmodified reference; build container only)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

DBPSK, BPSK, DQPSK, QPSK, D8PSK, QAM8, QAM16, QAM32, QAM64, QAM256 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 10
R1_4, R1_3, R1_2, R2_3, R3_4, R5_6, R7_8 = range(7)
NCAR, CW_BITS, FRAME_BITS, MAX_EDGES = 59, 648, 2592, 4096

_f = C.POINTER(C.c_float)
_u8 = C.POINTER(C.c_uint8)
_i = C.POINTER(C.c_int)


def fp(a):
    return a.ctypes.data_as(_f)


def up(a):
    return a.ctypes.data_as(_u8)


def ip(a):
    return a.ctypes.data_as(_i)


class Geom(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "mod", "rate", "pilot_spacing", "n_pilot", "n_data", "bits_per_carrier", "bits_per_symbol",
        "n_data_symbols", "frame_samples", "n_llr", "info_bits", "bytes_per_cw", "max_iter")] + [
        ("all_idx", C.c_int * NCAR), ("is_pilot", C.c_int * NCAR),
        ("data_idx", C.c_int * NCAR), ("pilot_idx", C.c_int * NCAR),
        ("data_logical", C.c_int * NCAR), ("pilot_logical", C.c_int * NCAR),
        ("sync_re", C.c_float * NCAR), ("sync_im", C.c_float * NCAR),
        ("pilot_seq", C.c_float * NCAR),
        ("interp_lo", C.c_int * NCAR), ("interp_hi", C.c_int * NCAR),
        ("interp_alpha", C.c_float * NCAR)]


class Ldpc(C.Structure):
    _fields_ = [("rate", C.c_int), ("k", C.c_int), ("m", C.c_int), ("n", C.c_int), ("n_edges", C.c_int),
                ("row_ptr", C.c_int * (CW_BITS + 1)), ("edge_var", C.c_int * MAX_EDGES)]


class Mt(C.Structure):   # ro_mt: std::mt19937
    _fields_ = [("s", C.c_uint32 * 624), ("idx", C.c_int)]


class RxAux(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "snr_db", "cfo_hz", "fading_index", "noise_variance", "lts_phase_slope", "snr_linear",
        "corr_phase", "snr_symbol_count")] + [("h", C.c_float * (2 * NCAR))]


def build_oracle():
    so = os.path.join(HERE, "libria_oracle.so")
    srcs = [os.path.join(HERE, f) for f in ("ria_oracle.c", "ria_oracle_sync.c", "ria_oracle.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    return so


def acq_buffer(checker, pre, buf_len, off, seed, snr_db, cfo_hz, cfo_model="tx"):
    """One config-4 acquisition buffer on the CPU, from the same recipe ria_amd.sweep.make_acq_buffers follows on the
    device (checker: Oracle or Ref): preamble -> [applyTxCFO] -> placed at `off` in silence -> AWGN WattersonChannel
    (mt19937 `seed`), with Config::cfo_hz instead of the transmitter offset for cfo_model "watterson"."""
    buf = np.zeros(buf_len, np.float32)
    if cfo_model == "tx":
        seg, _ = checker.apply_tx_cfo(pre, cfo_hz)
        buf[off:off + len(pre)] = seg
        return checker.channel(0, snr_db, int(seed), buf)
    buf[off:off + len(pre)] = pre
    return checker.channel_cfo(0, snr_db, int(seed), buf, cfo_hz)[0]


class Oracle:
    """Our C restatement (oracle/ria_oracle.c)."""

    def __init__(self):
        self.lib = L = C.CDLL(build_oracle())
        L.ro_rx_process.argtypes = [C.POINTER(Geom), _f, C.c_int, C.c_float, C.c_longlong, _f, C.c_int,
                                    C.POINTER(RxAux)]
        L.ro_rx_process_flags.argtypes = [C.POINTER(Geom), _f, C.c_int, C.c_float, C.c_longlong, C.c_int, _f, C.c_int,
                                          C.POINTER(RxAux)]
        L.ro_burst_interleave.argtypes = [C.c_int, _u8, _u8]
        L.ro_burst_interleave.restype = None
        L.ro_burst_deinterleave.argtypes = [C.c_int, _f, C.c_int, _f]
        L.ro_burst_deinterleave.restype = None
        L.ro_channel.argtypes = [C.c_int, C.c_float, C.c_uint32, _f, C.c_int, _f]
        L.ro_channel_cfo.argtypes = [C.c_int, C.c_float, C.c_uint32, C.c_float, C.c_float, _f, C.c_int, _f, _f]
        L.ro_apply_tx_cfo.argtypes = [_f, C.c_int, C.c_float, _f, _f]
        L.ro_ldpc_decode.argtypes = [C.POINTER(Ldpc), _f, C.c_int, C.c_int, C.c_float, _u8, _i]
        L.ro_mt_seed.argtypes = [C.POINTER(Mt), C.c_uint32]
        L.ro_mt_next.argtypes = [C.POINTER(Mt)]
        L.ro_mt_next.restype = C.c_uint32
        L.ro_tool_add_noise.argtypes = [_f, C.c_int, C.c_float, C.POINTER(Mt)]
        L.ro_tool_apply_cfo.argtypes = [_f, C.c_int, C.c_float, C.c_float]
        L.ro_tool_chase_reception.argtypes = [_u8, C.c_float, C.POINTER(Mt), _f]
        L.ro_decode_fixed_frame.argtypes = [_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8, _u8, _i, _i]
        L.ro_crc16.restype = C.c_uint16
        L.ro_zc_generate.argtypes = [C.c_int, _f, C.c_int]
        L.ro_zc_detect.argtypes = [_f, C.c_int, C.c_float, C.c_int, C.c_float, _f]
        L.ro_chirp_generate.argtypes = [_f, C.c_int]
        L.ro_chirp_detect.argtypes = [_f, C.c_int, C.c_float, _f]
        L.ro_detect_data_sync.argtypes = [_f, C.c_int, C.c_float, C.c_float, _f]
        L.ro_cox_search.argtypes = [C.POINTER(Geom), _f, C.c_int, C.c_float, _f, _f]
        L.ro_cox_lts_template.argtypes = [C.POINTER(Geom), _f, _f]
        L.ro_mcdpsk_modulate.argtypes = [C.c_int, C.c_int, C.c_int, _u8, C.c_int, _f, C.c_int]
        L.ro_mcdpsk_demod.argtypes = [C.c_int, C.c_int, C.c_int, _f, C.c_int, C.c_float, C.c_float, _f, C.c_int, _f]
        self._geoms = {}
        self._codes = {}

    def zc_generate(self, root):
        out = np.zeros(4096, np.float32)
        n = self.lib.ro_zc_generate(root, fp(out), len(out))
        return out[:n].copy()

    def zc_detect(self, samples, threshold=0.3, root_mask=15, known_cfo=0.0):
        """-> float32[7] {detected, frame_type, start_sample, correlation, cfo_hz, snr_estimate, root}"""
        x = np.ascontiguousarray(samples, np.float32)
        out = np.zeros(7, np.float32)
        self.lib.ro_zc_detect(fp(x), len(x), threshold, root_mask, known_cfo, fp(out))
        return out

    def chirp_generate(self):
        out = np.zeros(60000, np.float32)
        n = self.lib.ro_chirp_generate(fp(out), len(out))
        return out[:n].copy()

    def detect_data_sync(self, samples, known_cfo=0.0, threshold=0.5):
        """-> float32[4] {detected, start_sample, correlation, burst_interleaved}"""
        x = np.ascontiguousarray(samples, np.float32)
        out = np.zeros(4, np.float32)
        self.lib.ro_detect_data_sync(fp(x), len(x), known_cfo, threshold, fp(out))
        return out

    def cox_search(self, samples, threshold=0.8, noise_floor=0.0, mod=QAM16, rate=R1_2):
        """OFDMDemodulator::searchForSync -> (float32[3] {found, first-LTS position, cfo_hz}, noise floor after)"""
        x = np.ascontiguousarray(samples, np.float32)
        out = np.zeros(3, np.float32)
        nf = np.array([noise_floor], np.float32)
        self.lib.ro_cox_search(C.byref(self.geom(mod, rate)), fp(x), len(x), threshold, fp(nf), fp(out))
        return out, float(nf[0])

    def cox_lts_template(self, mod=QAM16, rate=R1_2):
        tI, tQ = np.zeros(1152, np.float32), np.zeros(1152, np.float32)
        self.lib.ro_cox_lts_template(C.byref(self.geom(mod, rate)), fp(tI), fp(tQ))
        return tI, tQ

    def mcdpsk_modulate(self, nc, bps, spreading, data):
        data = np.ascontiguousarray(data, np.uint8)
        out = np.zeros(600000, np.float32)
        n = self.lib.ro_mcdpsk_modulate(nc, bps, spreading, up(data), len(data), fp(out), len(out))
        assert n > 0
        return out[:n].copy()

    def mcdpsk_demod(self, nc, bps, spreading, samples, cfo_hz=0.0, phase0=0.0):
        """-> (llr float32[n], aux float32[4] {cfo, fading, freq fading, temporal fading})"""
        x = np.ascontiguousarray(samples, np.float32)
        llr = np.zeros(8192, np.float32)
        aux = np.zeros(4, np.float32)
        n = self.lib.ro_mcdpsk_demod(nc, bps, spreading, fp(x), len(x), cfo_hz, phase0, fp(llr), len(llr), fp(aux))
        assert n > 0, n
        return llr[:n].copy(), aux

    def chirp_detect(self, samples, threshold=0.15):
        """-> float32[6] {success, up_start, down_start, cfo_hz, up_corr, down_corr}"""
        x = np.ascontiguousarray(samples, np.float32)
        out = np.zeros(6, np.float32)
        self.lib.ro_chirp_detect(fp(x), len(x), threshold, fp(out))
        return out

    def geom(self, mod, rate):
        key = (mod, rate)
        if key not in self._geoms:
            g = Geom()
            self.lib.ro_geom_init(C.byref(g), mod, rate)
            self._geoms[key] = g
        return self._geoms[key]

    def code(self, rate):
        if rate not in self._codes:
            c = Ldpc()
            self.lib.ro_ldpc_build(C.byref(c), rate)
            self._codes[rate] = c
        return self._codes[rate]

    def H_edges(self, rate):
        c = self.code(rate)
        return (np.array(c.row_ptr[:c.m + 1], dtype=np.int32),
                np.array(c.edge_var[:c.n_edges], dtype=np.int32), c.k, c.m)

    def make_frame(self, payload, seq, rate):
        g = self.geom(QAM16, rate)
        out = np.zeros(4 * g.bytes_per_cw, np.uint8)
        payload = np.ascontiguousarray(payload, np.uint8)
        self.lib.ro_make_frame(up(payload), len(payload), seq, rate, up(out))
        return out

    def encode_fixed_frame(self, info, rate, ch_il, bps):
        info = np.ascontiguousarray(info, np.uint8)
        out = np.zeros(324, np.uint8)
        self.lib.ro_encode_fixed_frame(up(info), len(info), rate, int(ch_il), bps, up(out))
        return out

    def modulate(self, mod, rate, coded):
        g = self.geom(mod, rate)
        coded = np.ascontiguousarray(coded, np.uint8)
        out = np.zeros(g.frame_samples + 1152, np.float32)
        n = self.lib.ro_modulate(C.byref(g), up(coded), len(coded), fp(out), len(out))
        assert n > 0
        return out[:n].copy()

    def tx_frame(self, mod, rate, payload, seq):
        g = self.geom(mod, rate)
        info = self.make_frame(payload, seq, rate)
        coded = self.encode_fixed_frame(info, rate, True, g.bits_per_symbol)
        return self.modulate(mod, rate, coded), info, coded

    def channel(self, kind, snr_db, seed, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        self.lib.ro_channel(kind, snr_db, seed, fp(x), len(x), fp(y))
        return y

    def channel_cfo(self, kind, snr_db, seed, x, cfo_hz=0.0, random_cfo_max_hz=0.0):
        """WattersonChannel with Config::cfo_hz / random_cfo_max_hz -> (samples, getActualCFO())"""
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        a = np.zeros(1, np.float32)
        self.lib.ro_channel_cfo(kind, snr_db, seed, cfo_hz, random_cfo_max_hz, fp(x), len(x), fp(y), fp(a))
        return y, float(a[0])

    def apply_tx_cfo(self, x, cfo_hz, phase=0.0):
        """SimulatedChannel::applyTxCFO -> (samples, phase accumulator after)"""
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        ph = np.array([phase], np.float32)
        self.lib.ro_apply_tx_cfo(fp(x), len(x), cfo_hz, fp(ph), fp(y))
        return y, float(ph[0])

    def mcdpsk_wf_tx(self, carriers, mod, rate, spreading, data_preamble, coded):
        """MCDPSKWaveform TX restated from the oracle's pieces (mc_dpsk_waveform.cpp:127-174)"""
        bps = 1 if mod == DBPSK else 2
        pre = self.zc_generate(5) if data_preamble else self.chirp_generate()
        return np.concatenate([pre, self.mcdpsk_modulate(carriers, bps, spreading, coded)])

    def mcdpsk_wf_rx(self, carriers, mod, rate, spreading, data_sync, samples, known_cfo=0.0, threshold=None, frame_len=0):
        """MCDPSKWaveform RX restated (mc_dpsk_waveform.cpp:176-338): detectSync / detectDataSync -> setFrequencyOffset(result
        CFO) -> process (demodulator after an external chirp detection) -> same tuple as Ref.mcdpsk_wf_rx"""
        x = np.ascontiguousarray(samples, np.float32)
        bps = 1 if mod == DBPSK else 2
        sync4, aux5 = np.zeros(4, np.float32), np.zeros(5, np.float32)
        if data_sync:
            r = self.zc_detect(x, 0.2 if threshold is None else threshold, 12, known_cfo)     # roots DATA | CONTROL
            sync4[:] = [r[0], r[2] if r[0] else -1.0, r[3], r[4]]
        else:
            r = self.chirp_detect(x, 0.15 if threshold is None else threshold)
            sync4[:] = [r[0], (r[2] + 24000 + 4800) if r[0] else r[1], max(r[4], r[5]), r[3]]
        start = int(sync4[1])
        if not sync4[0] or start < 0 or start >= len(x):
            return sync4, np.zeros(0, np.float32), aux5
        take = len(x) - start
        if frame_len > 0:
            take = min(take, frame_len)
        cfo = float(sync4[3])
        if data_sync and abs(known_cfo) > 0.01 and abs(np.float32(cfo) - np.float32(known_cfo)) > 1.0:   # streaming_decoder.cpp:903-917
            cfo = float(np.float32(known_cfo))
        aux5[3] = cfo
        if take <= 9 * 512:
            return sync4, np.zeros(0, np.float32), aux5
        llr, aux = self.mcdpsk_demod(carriers, bps, spreading, x[start:start + take], cfo, 0.0)
        aux5[:] = [1.0, aux[0], aux[1], cfo, 1.0]
        return sync4, llr, aux5

    def rx_process(self, mod, rate, samples, cfo_hz=0.0, abs_pos=0, burst_marker=False):
        g = self.geom(mod, rate)
        samples = np.ascontiguousarray(samples, np.float32)
        llr = np.zeros(8 * NCAR * 64, np.float32)
        aux = RxAux()
        n = self.lib.ro_rx_process_flags(C.byref(g), fp(samples), len(samples), cfo_hz, abs_pos, int(bool(burst_marker)),
                                         fp(llr), len(llr), C.byref(aux))
        return llr[:n].copy(), aux

    def burst_interleave(self, logical):
        """[N, 324] coded bytes -> physical [N, 324] (fec::BurstInterleaver::interleave)"""
        logical = np.ascontiguousarray(logical, np.uint8)
        out = np.zeros_like(logical)
        self.lib.ro_burst_interleave(logical.shape[0], up(logical), up(out))
        return out

    def burst_deinterleave(self, physical):
        """[N, >=2592] soft bits of the physical frames -> logical [N, 2592] (BurstInterleaver::deinterleave)"""
        physical = np.ascontiguousarray(physical, np.float32)
        out = np.zeros((physical.shape[0], 2592), np.float32)
        self.lib.ro_burst_deinterleave(physical.shape[0], fp(physical), physical.shape[1], fp(out))
        return out

    def ldpc_encode(self, rate, info):
        c = self.code(rate)
        info = np.ascontiguousarray(info, np.uint8)
        out = np.zeros(81, np.uint8)
        self.lib.ro_ldpc_encode(C.byref(c), up(info), len(info), up(out))
        return out

    def ldpc_decode(self, rate, llr, max_iter, factor):
        c = self.code(rate)
        llr = np.ascontiguousarray(llr, np.float32)
        out = np.zeros(81, np.uint8)
        it = C.c_int()
        ok = self.lib.ro_ldpc_decode(C.byref(c), fp(llr), len(llr), max_iter, factor, up(out), C.byref(it))
        return bool(ok), out[:(c.k + 7) // 8].copy(), it.value

    def robust_decode(self, rate, llr):
        """robustDecodeSingleCW (streaming_decoder.cpp:1028-1058) restated on the oracle's decoder:
        -> (ok, bytes of the last attempt, iterations, tries)"""
        mi = self.geom(QAM16, rate).max_iter          # getRecommendedIterations (ldpc_codec.hpp:86-95)
        ok, out, it, tries = False, None, 0, 0
        for factor in (0.9375, 0.875, 0.75, 0.625, 0.5):
            ok, out, it = self.ldpc_decode(rate, llr, mi, factor)
            tries += 1
            if ok:
                break
        return ok, out, it, tries

    def decode_fixed_frame(self, llr, rate, ch_deint, bps, flags=3):
        g = self.geom(QAM16, rate)
        llr = np.ascontiguousarray(llr, np.float32)
        data = np.zeros(4 * g.bytes_per_cw, np.uint8)
        ok = np.zeros(4, np.uint8)
        iters = np.zeros(4, np.int32)
        att = np.zeros(4, np.int32)
        self.lib.ro_decode_fixed_frame(fp(llr), len(llr), rate, int(ch_deint), bps, flags, up(data), up(ok),
                                       ip(iters), ip(att))
        return data, ok, iters, att

    # ---- the scenarios of the reference's own test programs, composed from the restatement (the compiled reference gives the
    # same arrays through Ref.tool_*: oracle/ref_shim_tools.cpp)
    ZC_TOOL_ROOTS = (1, 3, 5, 7)   # ZCFrameType PING, PONG, DATA, CONTROL (tools/test_zc_sync.cpp:313-323)

    def _rng(self, seed):
        m = Mt()
        self.lib.ro_mt_seed(C.byref(m), seed & 0xFFFFFFFF)
        return m

    def tool_add_noise(self, x, snr_db, rng):
        x = np.ascontiguousarray(x, np.float32).copy()
        self.lib.ro_tool_add_noise(fp(x), len(x), snr_db, C.byref(rng))
        return x

    def tool_zc_cases(self):
        """tools/test_zc_sync.cpp tests 0-4 (seed 42) -> dict(signals [50][4512], lengths, test, type, param, res7)"""
        pre = [self.zc_generate(r) for r in self.ZC_TOOL_ROOTS]
        pad = lambda t, n: np.concatenate([np.zeros(n, np.float32), pre[t], np.zeros(n, np.float32)])
        cases = []
        for t in range(4):
            cases.append((0, t, 0.0, pad(t, 500)))
        rng = self._rng(42)
        for t in range(4):
            cases.append((1, t, 20.0, self.tool_add_noise(pad(t, 1000), 20.0, rng)))
        snr = np.float32(-15.0)
        while snr <= np.float32(20.0):
            cases.append((2, 0, float(snr), self.tool_add_noise(pad(0, 500), float(snr), self._rng(42))))
            snr = np.float32(snr + np.float32(2.5))
        for cfo in (-15.0, -10.0, -5.0, 0.0, 5.0, 10.0, 15.0):
            x = pad(2, 500)
            self.lib.ro_tool_apply_cfo(fp(x), len(x), cfo, 48000.0)
            cases.append((3, 2, cfo, self.tool_add_noise(x, 15.0, self._rng(42))))
        for trial in range(5):
            for t in range(4):
                cases.append((4, t, 10.0, self.tool_add_noise(pad(t, 500), 10.0, self._rng(42 + trial * 100 + t))))
        n = len(cases)
        sig = np.zeros((n, 4512), np.float32)
        for i, c in enumerate(cases):
            sig[i, :len(c[3])] = c[3]
        res = np.stack([self.zc_detect(c[3], 0.2) for c in cases])
        return dict(signals=sig, lengths=np.array([len(c[3]) for c in cases], np.int32), test=np.array([c[0] for c in cases], np.int32),
                    type=np.array([c[1] for c in cases], np.int32), param=np.array([c[2] for c in cases], np.float32), res7=res)

    def tool_spreading_case(self, snr_db, spreading, seed):
        """tools/test_spreading.cpp testAtSNR: LDPCCodec::decode sees 650 soft bits, so LDPCDecoder::decodeSoft takes its
        multi-block branch (ldpc_decoder.cpp:305-392): the first 648 through min-sum at the decoder's DEFAULT factor 0.75 for
        up to 80 iterations, the hard decisions kept whether or not it converged (the program then only counts bit errors)"""
        rng = self._rng(seed)
        tx = np.array([self.lib.ro_mt_next(C.byref(rng)) & 0xFF for _ in range(40)], np.uint8)
        coded = self.ldpc_encode(R1_2, tx)
        sp = spreading if spreading in (2, 4) else 1
        frame = self.tool_add_noise(self.mcdpsk_modulate(10, 1, sp, coded), snr_db, rng)
        soft, _ = self.mcdpsk_demod(10, 1, sp, frame, 0.0, 0.0)
        _, dec, _ = self.ldpc_decode(R1_2, soft[:648], 80, 0.75)
        # the two soft bits beyond 648 go through decodeBP as a zero-padded block of their own (:394-406), and THAT decode's
        # verdict is what lastDecodeSuccess() reports for the call
        tail = np.zeros(648, np.float32); tail[:len(soft) - 648] = soft[648:]
        ok = int(self.ldpc_decode(R1_2, tail, 80, 0.75)[0]) if len(soft) > 648 else 1
        errs = int(np.unpackbits(dec[:40] ^ tx).sum()) if ok else 0
        return dict(tx=tx, frame=frame, soft=soft, decoded=dec[:40].copy() if ok else np.zeros(40, np.uint8), ok=ok, bit_errors=errs)

    def tool_chase_llrs(self):
        """tools/test_chase_cache.cpp tests 2-3 -> (llrs [400][648], ok [350]): LDPCCodec::decode of 648 soft bits is
        decodeBP at the decoder's default factor 0.75, 80 iterations"""
        coded = self.ldpc_encode(R1_2, np.arange(40, dtype=np.uint8))
        rng = self._rng(42)
        l = np.zeros((400, 648), np.float32)
        for i in range(400):
            self.lib.ro_tool_chase_reception(up(coded), 2.5 if i < 200 else 1.5, C.byref(rng), fp(l[i]))
        dec = lambda v: 1 if self.ldpc_decode(R1_2, np.ascontiguousarray(v, np.float32), 80, 0.75)[0] else 0
        ok = []
        for t in range(100):
            a, b = l[2 * t], l[2 * t + 1]
            ok += [dec(a), dec(a + b)]
        for t in range(50):
            a, b, c, d = l[200 + 4 * t: 204 + 4 * t]
            ok += [dec(a), dec(a + b), dec(((a + b) + c) + d)]
        return l, np.array(ok, np.uint8)

    def tool_zc_dbpsk_case(self, snr_db, seed):
        """tools/test_zc_dbpsk.cpp testAtSNR: silence + ZC (DATA) + MC-DPSK DBPSK frame + silence in noise -> ZCSync::detect ->
        the demodulator from start_sample with the ZC's CFO (81 expected bytes = 65 data symbols) -> 648 soft bits -> decodeBP at
        the decoder's default factor 0.75.  stage: 0 no sync, 1 bad start, 2 frame not ready, 4 decode failed, 5 decoded"""
        rng = self._rng(seed)
        tx = np.array([self.lib.ro_mt_next(C.byref(rng)) & 0xFF for _ in range(40)], np.uint8)
        frame = self.mcdpsk_modulate(10, 1, 1, self.ldpc_encode(R1_2, tx))
        z = np.zeros(500, np.float32)
        sig = self.tool_add_noise(np.concatenate([z, self.zc_generate(5), frame, z]), snr_db, rng)
        zc7 = self.zc_detect(sig, 0.2)
        out = dict(tx=tx, signal=sig, zc7=zc7, soft=np.zeros(648, np.float32), decoded=np.zeros(40, np.uint8), sync=int(zc7[0]), ok=0, bit_errors=0, stage=0)
        if not out["sync"]:
            return out
        start = int(zc7[2])
        out["stage"] = 1
        if start < 0 or start >= len(sig) - 1000:
            return out
        out["stage"] = 2
        need = (8 + 1 + 65) * 512
        dpsk = sig[start:len(sig) - 500]
        if len(dpsk) < need:
            return out
        soft, _ = self.mcdpsk_demod(10, 1, 1, dpsk[:need], float(zc7[4]), 0.0)
        out["stage"] = 4
        out["soft"] = soft[:648].copy()
        ok, dec, _ = self.ldpc_decode(R1_2, out["soft"], 80, 0.75)
        if not ok:
            return out
        out.update(stage=5, ok=1, decoded=dec[:40].copy(), bit_errors=int(np.unpackbits(dec[:40] ^ tx).sum()))
        return out

    def gather_table(self, bps, use_channel=True):
        t = np.zeros(4 * CW_BITS, np.int32)
        self.lib.ro_rx_gather_table(bps, int(use_channel), ip(t))
        return t


class Ref:
    """The compiled, unmodified reference (oracle/_ref/libria_ref.so). Build container only."""

    PATH = os.path.join(HERE, "_ref", "libria_ref.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = L = C.CDLL(self.PATH)
        L.ref_channel.argtypes = [C.c_int, C.c_float, C.c_uint32, _f, C.c_int, _f]
        L.ref_rx_process.argtypes = [C.c_int, C.c_int, _f, C.c_int, C.c_float, C.c_longlong, C.c_int, _f,
                                     C.c_int, _f, _f]
        L.ref_rx_process_nvis.argtypes = L.ref_rx_process.argtypes
        L.ref_ldpc_decode.argtypes = [C.c_int, _f, C.c_int, C.c_int, C.c_float, _u8, C.c_int, _i]
        L.ref_detect_data_sync.argtypes = [C.c_int, C.c_int, _f, C.c_int, C.c_float, C.c_float, _i, _f, _i]
        L.ref_chirp_detect.argtypes = [_f, C.c_int, C.c_float, _f]
        L.ref_chirp_generate.argtypes = [_f, C.c_int]
        L.ref_mcdpsk_modulate.argtypes = [C.c_int, C.c_int, C.c_int, _u8, C.c_int, _f, C.c_int]
        L.ref_mcdpsk_demod.argtypes = [C.c_int, C.c_int, C.c_int, _f, C.c_int, C.c_float, C.c_float, _f, C.c_int, _f]
        L.ref_zc_generate.argtypes = [C.c_int, _f, C.c_int]
        L.ref_zc_detect.argtypes = [_f, C.c_int, C.c_float, C.c_int, C.c_float, _f]
        L.ref_cox_search.argtypes = [C.c_int, C.c_int, _f, C.c_int, C.c_float, _f, _f]
        L.ref_cox_transmit.argtypes = [C.c_int, C.c_int, _u8, C.c_int, _f, C.c_int]
        L.ref_cox_lts_template.argtypes = [C.c_int, C.c_int, _f, _f]
        L.ref_cox_rx.argtypes = [C.c_int, C.c_int, _f, C.c_int, C.c_float, _f, _f, C.c_int, _f]
        _i64 = C.c_longlong
        L.ref_robust_decode.argtypes = [C.c_int, _f, C.c_int, _u8, C.c_int, _i, _i]
        L.ref_burst_tx.argtypes = [C.c_int, C.c_int, _u8, C.c_int, C.c_int, C.c_int, C.c_int, _f, C.c_int, _u8, _u8]
        L.ref_burst_rx.argtypes = [C.c_int, C.c_int, _f, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _i64, _f, _f, C.c_int,
                                   _f, _f, _f, _u8, _u8]
        _u32 = C.POINTER(C.c_uint32)
        L.ref_harq_trials.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _u8, _u32, C.c_int, C.c_int, _i, _u32, _u32, _i, _u8, _f]
        L.ref_burst_interleave.argtypes = [C.c_int, _u8, _u8]
        L.ref_burst_deinterleave.argtypes = [C.c_int, _f, _f]
        L.ref_channel_cfo.argtypes = [C.c_int, C.c_float, C.c_uint32, C.c_float, C.c_float, _f, C.c_int, _f, _f]
        L.ref_apply_tx_cfo.argtypes = [_f, C.c_int, C.c_float, _f, _f]
        L.ref_mcdpsk_wf_tx.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8, C.c_int, _f, C.c_int]
        L.ref_mcdpsk_wf_rx.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f, C.c_int, C.c_float, C.c_float, C.c_int, _f, _f, C.c_int, _f]
        L.ref_mcdpsk_wf_sizes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i]
        L.ref_mcdpsk_wf_sizes.restype = None
        L.ref_rx_open.argtypes = [C.c_int, C.c_int]
        L.ref_rx_open.restype = C.c_void_p
        L.ref_rx_close.argtypes = [C.c_void_p]
        L.ref_rx_close.restype = None
        L.ref_rx_frame.argtypes = [C.c_void_p, _f, C.c_int, C.c_float, _u8, _u8, _f, C.c_int]
        L.ref_tool_zc_cases.argtypes = [_f, C.c_int, _i, _i, _i, _f, _f, C.c_int]
        L.ref_tool_spreading_case.argtypes = [C.c_float, C.c_int, C.c_uint32, _u8, _f, C.c_int, _i, _f, C.c_int, _u8, _i]
        L.ref_tool_chase_llrs.argtypes = [_f, _u8]
        L.ref_tool_zc_dbpsk_case.argtypes = [C.c_float, C.c_uint32, _u8, _f, C.c_int, _f, _f, _u8, _i]
        L.ref_quiet()

    # ---- the scenarios of the reference's own test programs (oracle/ref_shim_tools.cpp)
    def tool_zc_cases(self):
        """tools/test_zc_sync.cpp tests 0-4 -> dict(signals [50][4512], lengths, test, type, param, res7)"""
        n_max, stride = 64, 4512
        sig = np.zeros((n_max, stride), np.float32)
        ln = np.zeros(n_max, np.int32); te = np.zeros(n_max, np.int32); ty = np.zeros(n_max, np.int32)
        pa = np.zeros(n_max, np.float32); res = np.zeros((n_max, 7), np.float32)
        n = self.lib.ref_tool_zc_cases(fp(sig), stride, ip(ln), ip(te), ip(ty), fp(pa), fp(res), n_max)
        assert n > 0, n
        return dict(signals=sig[:n].copy(), lengths=ln[:n].copy(), test=te[:n].copy(), type=ty[:n].copy(), param=pa[:n].copy(), res7=res[:n].copy())

    def tool_spreading_case(self, snr_db, spreading, seed):
        """tools/test_spreading.cpp testAtSNR -> dict(tx, frame, sizes, soft, decoded, ok, bit_errors)"""
        frame = np.zeros(160000, np.float32); soft = np.zeros(1024, np.float32)
        tx = np.zeros(40, np.uint8); dec = np.zeros(40, np.uint8); sizes = np.zeros(3, np.int32); out3 = np.zeros(3, np.int32)
        n = self.lib.ref_tool_spreading_case(snr_db, spreading, seed, up(tx), fp(frame), len(frame), ip(sizes), fp(soft), len(soft), up(dec), ip(out3))
        assert n > 0, n
        return dict(tx=tx, frame=frame[:n].copy(), sizes=sizes, soft=soft[:out3[2]].copy(), decoded=dec, ok=int(out3[0]), bit_errors=int(out3[1]))

    def tool_chase_llrs(self):
        """tools/test_chase_cache.cpp tests 2-3 -> (llrs [400][648], ok [350])"""
        l = np.zeros((400, 648), np.float32); ok = np.zeros(100 * 2 + 50 * 3, np.uint8)
        n = self.lib.ref_tool_chase_llrs(fp(l), up(ok))
        assert n == 400, n
        return l, ok

    def tool_zc_dbpsk_case(self, snr_db, seed):
        """tools/test_zc_dbpsk.cpp testAtSNR -> dict(tx, signal, zc7, soft, decoded, sync, ok, bit_errors, stage)"""
        sig = np.zeros(50000, np.float32); zc7 = np.zeros(7, np.float32); soft = np.zeros(648, np.float32)
        tx = np.zeros(40, np.uint8); dec = np.zeros(40, np.uint8); out4 = np.zeros(4, np.int32)
        n = self.lib.ref_tool_zc_dbpsk_case(snr_db, seed & 0xFFFFFFFF, up(tx), fp(sig), len(sig), fp(zc7), fp(soft), up(dec), ip(out4))
        assert n > 0, n
        return dict(tx=tx, signal=sig[:n].copy(), zc7=zc7, soft=soft, decoded=dec, sync=int(out4[0]), ok=int(out4[1]), bit_errors=int(out4[2]), stage=int(out4[3]))

    def channel_cfo(self, kind, snr_db, seed, x, cfo_hz=0.0, random_cfo_max_hz=0.0):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        a = np.zeros(1, np.float32)
        self.lib.ref_channel_cfo(kind, snr_db, seed, cfo_hz, random_cfo_max_hz, fp(x), len(x), fp(y), fp(a))
        return y, float(a[0])

    def apply_tx_cfo(self, x, cfo_hz, phase=0.0):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        ph = np.array([phase], np.float32)
        self.lib.ref_apply_tx_cfo(fp(x), len(x), cfo_hz, fp(ph), fp(y))
        return y, float(ph[0])

    def mcdpsk_wf_tx(self, carriers, mod, rate, spreading, data_preamble, coded):
        """MCDPSKWaveform: generatePreamble() / generateDataPreamble() + modulate(coded)"""
        coded = np.ascontiguousarray(coded, np.uint8)
        out = np.zeros(700000, np.float32)
        n = self.lib.ref_mcdpsk_wf_tx(carriers, mod, rate, spreading, int(data_preamble), up(coded), len(coded), fp(out), len(out))
        assert n > 0, n
        return out[:n].copy()

    def mcdpsk_wf_rx(self, carriers, mod, rate, spreading, data_sync, samples, known_cfo=0.0, threshold=None, frame_len=0):
        """One MCDPSKWaveform in StreamingDecoder's call order -> (sync4 {detected, start, corr, cfo}, soft bits,
        aux5 {ready, estimatedCFO, fading index, getFrequencyOffset, isSynced})"""
        x = np.ascontiguousarray(samples, np.float32)
        thr = (0.2 if data_sync else 0.15) if threshold is None else threshold
        sync4, aux5, llr = np.zeros(4, np.float32), np.zeros(5, np.float32), np.zeros(16384, np.float32)
        n = self.lib.ref_mcdpsk_wf_rx(carriers, mod, rate, spreading, int(data_sync), fp(x), len(x), known_cfo, thr, frame_len, fp(sync4), fp(llr), len(llr), fp(aux5))
        return sync4, llr[:n].copy(), aux5

    def mcdpsk_wf_sizes(self, carriers, mod, rate, spreading, num_cw):
        out = np.zeros(4, np.int32)
        self.lib.ref_mcdpsk_wf_sizes(carriers, mod, rate, spreading, num_cw, ip(out))
        return out

    def rx_open(self, mod, rate):
        """One kept OFDMChirpWaveform (configured once): the per-thread object of the CPU throughput baseline"""
        return self.lib.ref_rx_open(mod, rate)

    def rx_close(self, h):
        self.lib.ref_rx_close(h)

    def rx_frame(self, h, samples, cfo_hz=0.0, want_llr=False):
        """reset -> setFrequencyOffset -> process -> getSoftBits -> decodeFixedFrame on the kept object
        -> (codewords decoded or -1, data, ok[, llr])"""
        x = np.ascontiguousarray(samples, np.float32)
        data = np.zeros(4 * 68, np.uint8)
        ok = np.zeros(4, np.uint8)
        llr = np.zeros(8 * NCAR * 64, np.float32) if want_llr else None
        n = self.lib.ref_rx_frame(h, fp(x), len(x), cfo_hz, up(data), up(ok), fp(llr) if want_llr else None, len(llr) if want_llr else 0)
        return (n, data, ok, llr) if want_llr else (n, data, ok)

    def rx_process_kept(self, h, samples, cfo_hz=0.0):
        """process() + getSoftBits() on the kept object -> soft bits"""
        x = np.ascontiguousarray(samples, np.float32)
        llr = np.zeros(8 * NCAR * 64, np.float32)
        n = self.lib.ref_rx_frame(h, fp(x), len(x), cfo_hz, None, None, fp(llr), len(llr))
        return llr[:max(n, 0)].copy()

    def cox_search(self, samples, threshold=0.8, noise_floor=0.0, mod=QAM16, rate=R1_2):
        x = np.ascontiguousarray(samples, np.float32)
        out = np.zeros(3, np.float32)
        nf = np.array([noise_floor], np.float32)
        self.lib.ref_cox_search(mod, rate, fp(x), len(x), threshold, fp(nf), fp(out))
        return out, float(nf[0])

    def cox_transmit(self, coded=None, mod=QAM16, rate=R1_2):
        """Schmidl-Cox preamble (guard + 4 STS + 2 LTS) followed by the modulated coded bytes (optional)."""
        coded = np.zeros(0, np.uint8) if coded is None else np.ascontiguousarray(coded, np.uint8)
        out = np.zeros(200000, np.float32)
        n = self.lib.ref_cox_transmit(mod, rate, up(coded), len(coded), fp(out), len(out))
        assert n > 0, n
        return out[:n].copy()

    def cox_rx(self, samples, threshold=0.8, mod=QAM16, rate=R1_2):
        """OFDM-COX detectSync + process -> (float32[3] {found, position, cfo}, soft bits, float32[2] {snr_db, cfo after})"""
        x = np.ascontiguousarray(samples, np.float32)
        out, llr, aux = np.zeros(3, np.float32), np.zeros(8192, np.float32), np.zeros(2, np.float32)
        n = self.lib.ref_cox_rx(mod, rate, fp(x), len(x), threshold, fp(out), fp(llr), len(llr), fp(aux))
        return out, llr[:abs(n)].copy(), aux

    def cox_lts_template(self, mod=QAM16, rate=R1_2):
        tI, tQ = np.zeros(1152, np.float32), np.zeros(1152, np.float32)
        assert self.lib.ref_cox_lts_template(mod, rate, fp(tI), fp(tQ)) == 1152
        return tI, tQ

    def burst_interleave(self, logical):
        logical = np.ascontiguousarray(logical, np.uint8)
        out = np.zeros_like(logical)
        self.lib.ref_burst_interleave(logical.shape[0], up(logical), up(out))
        return out

    def burst_deinterleave(self, physical):
        physical = np.ascontiguousarray(physical[:, :2592], np.float32)
        out = np.zeros_like(physical)
        self.lib.ref_burst_deinterleave(physical.shape[0], fp(physical), fp(out))
        return out

    def harq_trials(self, nc, bps, spreading, kind, snr_db, info21, seeds):
        """MC-DPSK data codeword + HARQ chase combining through the reference's modulator / channel / demodulator /
        ChaseCache / LDPCDecoder (oracle/ref_shim.cpp ref_harq_trials).  info21 uint8 [n, 21], seeds uint32 [n, max_tx]."""
        info21 = np.ascontiguousarray(info21, np.uint8)
        seeds = np.ascontiguousarray(seeds, np.uint32)
        n, max_tx = seeds.shape
        tts = np.zeros(n, np.int32); lc = np.zeros((n, max_tx), np.uint32); ac = np.zeros((n, max_tx), np.uint32)
        tries = np.zeros((n, max_tx, 2), np.int32); dec = np.zeros((n, 20), np.uint8); fad = np.zeros((n, max_tx), np.float32)
        rc = self.lib.ref_harq_trials(nc, bps, spreading, kind, snr_db, up(info21), seeds.ctypes.data_as(C.POINTER(C.c_uint32)), n, max_tx,
                                      ip(tts), lc.ctypes.data_as(C.POINTER(C.c_uint32)), ac.ctypes.data_as(C.POINTER(C.c_uint32)), ip(tries), up(dec), fp(fad))
        assert rc == 0, rc
        return {"tx_to_success": tts, "llr_crc": lc, "acc_crc": ac, "tries": tries, "decoded": dec, "fading": fad}

    def burst_tx(self, mod, rate, infos, negate_first_lts=True):
        """One burst group as StreamingEncoder::encodeBurstLight builds it: infos uint8 [N, info_bytes] ->
        (samples, coded logical [N,324], coded physical [N,324])"""
        infos = np.ascontiguousarray(infos, np.uint8)
        n = infos.shape[0]
        out = np.zeros(n * 60000, np.float32)
        cl, cp = np.zeros((n, 324), np.uint8), np.zeros((n, 324), np.uint8)
        m = self.lib.ref_burst_tx(mod, rate, up(infos), infos.shape[1], infos.shape[1], n, int(negate_first_lts), fp(out), len(out),
                                  up(cl), up(cp))
        assert m > 0, m
        return out[:m].copy(), cl, cp

    def burst_rx(self, mod, rate, samples, n_frames, known_cfo=0.0, threshold=0.5, search_len=21000, abs_base=0, bpc=40):
        """One burst group through ONE OFDMChirpWaveform in StreamingDecoder's call order -> dict"""
        x = np.ascontiguousarray(samples, np.float32)
        sync4 = np.zeros(4, np.float32)
        llr = np.zeros((n_frames, 4096), np.float32)
        cfo_used, cfo_after = np.zeros(n_frames, np.float32), np.zeros(n_frames, np.float32)
        logical = np.zeros((n_frames, 2592), np.float32)
        dec = np.zeros((n_frames, 4 * bpc), np.uint8)
        ok = np.zeros((n_frames, 4), np.uint8)
        m = self.lib.ref_burst_rx(mod, rate, fp(x), len(x), search_len, n_frames, known_cfo, threshold, abs_base, fp(sync4), fp(llr),
                                  llr.shape[1], fp(cfo_used), fp(cfo_after), fp(logical), up(dec), up(ok))
        return {"n_soft": m, "sync": sync4, "llr": llr[:, :max(m, 0)].copy(), "cfo_used": cfo_used, "cfo_after": cfo_after,
                "logical": logical, "dec_data": dec, "dec_ok": ok}

    def zc_generate(self, root):
        out = np.zeros(4096, np.float32)
        n = self.lib.ref_zc_generate(root, fp(out), len(out))
        return out[:n].copy()

    def chirp_generate(self):
        out = np.zeros(60000, np.float32)
        n = self.lib.ref_chirp_generate(fp(out), len(out))
        return out[:n].copy()

    def chirp_detect(self, samples, threshold=0.15):
        x = np.ascontiguousarray(samples, np.float32)
        out = np.zeros(6, np.float32)
        self.lib.ref_chirp_detect(fp(x), len(x), threshold, fp(out))
        return out

    def detect_data_sync(self, samples, known_cfo=0.0, threshold=0.5, mod=QAM16, rate=R1_2):
        """-> float32[4] {detected, start_sample, correlation, burst_interleaved}"""
        x = np.ascontiguousarray(samples, np.float32)
        start, burst, corr = C.c_int(0), C.c_int(0), C.c_float(0)
        ok = self.lib.ref_detect_data_sync(mod, rate, fp(x), len(x), known_cfo, threshold, C.byref(start), C.byref(corr), C.byref(burst))
        return np.array([ok, start.value if ok else 0, corr.value, burst.value if ok else 0], np.float32)

    def mcdpsk_modulate(self, nc, bps, spreading, data):
        data = np.ascontiguousarray(data, np.uint8)
        out = np.zeros(600000, np.float32)
        n = self.lib.ref_mcdpsk_modulate(nc, bps, spreading, up(data), len(data), fp(out), len(out))
        assert n > 0
        return out[:n].copy()

    def mcdpsk_demod(self, nc, bps, spreading, samples, cfo_hz=0.0, phase0=0.0):
        x = np.ascontiguousarray(samples, np.float32)
        llr = np.zeros(8192, np.float32)
        aux = np.zeros(4, np.float32)
        n = self.lib.ref_mcdpsk_demod(nc, bps, spreading, fp(x), len(x), cfo_hz, phase0, fp(llr), len(llr), fp(aux))
        assert n > 0, n
        return llr[:n].copy(), aux

    def zc_detect(self, samples, threshold=0.3, root_mask=15, known_cfo=0.0):
        x = np.ascontiguousarray(samples, np.float32)
        out = np.zeros(7, np.float32)
        self.lib.ref_zc_detect(fp(x), len(x), threshold, root_mask, known_cfo, fp(out))
        return out

    def tx_frame(self, mod, rate, payload, seq, nvis=False):
        """nvis: through the OFDM-COX waveform object (the only one that accepts QAM256)"""
        payload = np.ascontiguousarray(payload, np.uint8)
        samples = np.zeros(80000, np.float32)
        info = np.zeros(4 * 68, np.uint8)
        coded = np.zeros(324, np.uint8)
        bps = C.c_int()
        n = (self.lib.ref_tx_frame_nvis if nvis else self.lib.ref_tx_frame)(mod, rate, up(payload), len(payload), seq, fp(samples), len(samples),
                                  up(info), len(info), up(coded), 324, C.byref(bps))
        assert n > 0
        return samples[:n].copy(), info, coded, bps.value

    def encode_fixed_frame(self, info, rate, ch_il, bps):
        info = np.ascontiguousarray(info, np.uint8)
        out = np.zeros(324, np.uint8)
        self.lib.ref_encode_fixed_frame(up(info), len(info), rate, int(ch_il), bps, up(out), 324)
        return out

    def modulate(self, mod, rate, coded):
        coded = np.ascontiguousarray(coded, np.uint8)
        out = np.zeros(80000, np.float32)
        n = self.lib.ref_modulate(mod, rate, up(coded), len(coded), fp(out), len(out))
        assert n > 0
        return out[:n].copy()

    def channel(self, kind, snr_db, seed, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        self.lib.ref_channel(kind, snr_db, seed, fp(x), len(x), fp(y))
        return y

    def rx_process(self, mod, rate, samples, cfo_hz=0.0, abs_pos=0, use_abs=True, nvis=False):
        samples = np.ascontiguousarray(samples, np.float32)
        llr = np.zeros(8 * NCAR * 64, np.float32)
        aux = np.zeros(8, np.float32)
        h = np.zeros(2 * NCAR, np.float32)
        n = (self.lib.ref_rx_process_nvis if nvis else self.lib.ref_rx_process)(mod, rate, fp(samples), len(samples), cfo_hz, abs_pos, int(use_abs),
                                    fp(llr), len(llr), fp(aux), fp(h))
        return llr[:abs(n)].copy(), aux, h, n > 0

    def ldpc_encode(self, rate, info):
        info = np.ascontiguousarray(info, np.uint8)
        out = np.zeros(256, np.uint8)
        n = self.lib.ref_ldpc_encode(rate, up(info), len(info), up(out), 256)
        return out[:n].copy()

    def ldpc_decode(self, rate, llr, max_iter, factor):
        llr = np.ascontiguousarray(llr, np.float32)
        out = np.zeros(256, np.uint8)
        it = C.c_int()
        n = self.lib.ref_ldpc_decode(rate, fp(llr), len(llr), max_iter, factor, up(out), 256, C.byref(it))
        return n > 0, out[:abs(n)].copy(), it.value

    def robust_decode(self, rate, llr):
        """robustDecodeSingleCW -> (ok, bytes of the last attempt, iterations, tries)"""
        llr = np.ascontiguousarray(llr, np.float32)
        out = np.zeros(256, np.uint8)
        it, tr = C.c_int(), C.c_int()
        n = self.lib.ref_robust_decode(rate, fp(llr), len(llr), up(out), 256, C.byref(it), C.byref(tr))
        return n > 0, out[:abs(n)].copy(), it.value, tr.value

    def decode_fixed_frame(self, llr, rate, ch_deint, bps):
        llr = np.ascontiguousarray(llr, np.float32)
        data = np.zeros(4 * 68, np.uint8)
        ok = np.zeros(4, np.uint8)
        self.lib.ref_decode_fixed_frame(fp(llr), len(llr), rate, int(ch_deint), bps, up(data), up(ok))
        return data, ok

    def channel_interleaver_inv(self, bps, total=648):
        out = np.zeros(total, np.int32)
        self.lib.ref_channel_interleaver_perm(bps, total, ip(out))
        return out
