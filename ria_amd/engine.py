"""Batched MI355X RX engine: thin Python over the C ABI (include/ria_gpu.h).

`RxEngine` owns one ria_gpu handle (one modulation/code-rate pair, like one configured IWaveform).
All heavy lifting happens in libria_gpu.so; torch only provides device buffers and streams.
There is no CPU fallback: constructing an engine without a HIP device raises.
"""
import ctypes as C

import numpy as np
import torch

from . import capi


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class RxEngine:
    def __init__(self, modulation="QAM16", code_rate="R1_2", device=0, max_batch=4096):
        if not torch.cuda.is_available():
            raise capi.RiaError("ria_amd needs a HIP device (MI355X); there is no CPU fallback")
        self.lib = capi.load()
        self.modulation = capi.MOD[modulation] if isinstance(modulation, str) else int(modulation)
        self.code_rate = capi.RATE[code_rate] if isinstance(code_rate, str) else int(code_rate)
        cfg = capi.Config()
        self.lib.ria_gpu_default_config(C.byref(cfg))
        cfg.device = device
        cfg.modulation = self.modulation
        cfg.code_rate = self.code_rate
        cfg.max_batch = max_batch
        h = C.c_void_p()
        rc = self.lib.ria_gpu_create(C.byref(cfg), C.byref(h))
        if rc != capi.RIA_OK:
            raise capi.RiaError(f"ria_gpu_create failed with status {rc}")
        self.h = h
        self.device = torch.device("cuda", device)
        self.geo = capi.Geometry()
        self._check(self.lib.ria_gpu_get_geometry(self.h, C.byref(self.geo)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.ria_gpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != capi.RIA_OK:
            raise capi.RiaError(f"libria_gpu status {rc}: {self.lib.ria_gpu_last_error(self.h).decode()}")

    def set_split_parts(self, parts):
        """How many parts ria_gpu_rx_batch cuts a large batch into (internal streams); 0 = library default."""
        self._check(self.lib.ria_gpu_set_option(self.h, capi.OPT_SPLIT_PARTS, int(parts)))

    def set_dual_decoder(self, mode):
        """1: two codewords per wave in the retry kernels, -1: one, 0: library default"""
        self._check(self.lib.ria_gpu_set_option(self.h, capi.OPT_DUAL_DECODER, int(mode)))

    # ---- helpers
    def _meta(self, n, cfo_hz, abs_pos, flags):
        if cfo_hz is None and abs_pos is None and flags is None:
            return None
        m = np.zeros(n, dtype=np.dtype([("cfo_hz", "<f4"), ("flags", "<u4"), ("abs_position", "<u8")]))
        if cfo_hz is not None:
            m["cfo_hz"] = cfo_hz
        if abs_pos is not None:
            m["abs_position"] = abs_pos
        if flags is not None:
            m["flags"] = flags
        return torch.from_numpy(m.view(np.uint8).reshape(n, 16)).to(self.device)

    @staticmethod
    def _status_array(t, dtype):
        return t.cpu().numpy().view(dtype).reshape(-1)

    FRAME_STATUS = np.dtype([("snr_db", "<f4"), ("cfo_hz", "<f4"), ("fading_index", "<f4"),
                             ("noise_variance", "<f4"), ("lts_phase_slope", "<f4"), ("snr_linear", "<f4"),
                             ("corr_phase", "<f4"), ("n_llr", "<i4")])
    DECODE_STATUS = np.dtype([("cw_ok", "u1", 4), ("iterations", "<u2", 4), ("attempts", "u1", 4),
                              ("frame_valid", "u1"), ("needs_recovery", "u1"), ("reserved", "u1", 2)])

    # ---- batched calls (device tensors in, device tensors out)
    def _frames_in(self, samples, offsets):
        """(n_frames, offsets tensor or None): either [n, frame_samples] rows, or one capture + uint64 sample offsets"""
        assert samples.dtype == torch.float32 and samples.is_contiguous()
        if offsets is None:
            assert samples.dim() == 2 and samples.shape[1] == self.geo.frame_samples
            return samples.shape[0], None
        off = np.ascontiguousarray(offsets, np.uint64)
        assert off.size == 0 or int(off.max()) + self.geo.frame_samples <= samples.numel(), "frame runs past the capture"
        return len(off), torch.from_numpy(off.view(np.int64)).to(self.device)

    def demod(self, samples, cfo_hz=None, abs_pos=None, flags=None, want_status=True, offsets=None):
        """samples: float32 [n_frames, frame_samples] on the GPU (or one capture + per-frame sample offsets)
        -> (llr [n, llrs_per_frame], status)"""
        n, off = self._frames_in(samples, offsets)
        llr = torch.empty((n, self.geo.llrs_per_frame), dtype=torch.float32, device=self.device)
        st = torch.zeros((n, 32), dtype=torch.uint8, device=self.device) if want_status else None
        meta = self._meta(n, cfo_hz, abs_pos, flags)
        self._check(self.lib.ria_gpu_demod_batch(self.h, _ptr(samples), _ptr(off), _ptr(meta), n, _ptr(llr), _ptr(st),
                                                 _stream_ptr()))
        return llr, st

    def decode(self, llr, flags=capi.DECODE_FULL):
        """llr: float32 [n_frames, >=2592] -> (info bytes [n, info_bytes_per_frame], status)"""
        n = llr.shape[0]
        assert llr.dtype == torch.float32 and llr.is_contiguous() and llr.shape[1] >= 2592
        info = torch.empty((n, self.geo.info_bytes_per_frame), dtype=torch.uint8, device=self.device)
        st = torch.zeros((n, 20), dtype=torch.uint8, device=self.device)
        self._check(self.lib.ria_gpu_decode_batch(self.h, _ptr(llr), llr.shape[1], n, flags, _ptr(info), _ptr(st),
                                                  _stream_ptr()))
        return info, st

    def rx(self, samples, flags=capi.DECODE_FULL, cfo_hz=None, abs_pos=None, meta_flags=None, want_llr=False,
           out=None, offsets=None):
        """Fused samples -> payload bytes. Returns (info, decode_status[, llr, frame_status])."""
        n, off = self._frames_in(samples, offsets)
        if out is None:
            info = torch.empty((n, self.geo.info_bytes_per_frame), dtype=torch.uint8, device=self.device)
            st = torch.zeros((n, 20), dtype=torch.uint8, device=self.device)
        else:
            info, st = out
        llr = fst = None
        if want_llr:
            llr = torch.empty((n, self.geo.llrs_per_frame), dtype=torch.float32, device=self.device)
            fst = torch.zeros((n, 32), dtype=torch.uint8, device=self.device)
        meta = self._meta(n, cfo_hz, abs_pos, meta_flags)
        self._check(self.lib.ria_gpu_rx_batch(self.h, _ptr(samples), _ptr(off), _ptr(meta), n, flags, _ptr(info), _ptr(st),
                                              _ptr(llr), _ptr(fst), _stream_ptr()))
        return (info, st, llr, fst) if want_llr else (info, st)

    def ldpc_decode(self, llr_rows, max_iterations, factor):
        """llr_rows: float32 [n_cw, 648] in decoder order."""
        n = llr_rows.shape[0]
        assert llr_rows.dtype == torch.float32 and llr_rows.is_contiguous() and llr_rows.shape[1] == 648
        nb = (self.geo.ldpc_k + 7) // 8
        out = torch.empty((n, nb), dtype=torch.uint8, device=self.device)
        ok = torch.empty(n, dtype=torch.uint8, device=self.device)
        it = torch.empty(n, dtype=torch.int16, device=self.device)
        self._check(self.lib.ria_gpu_ldpc_decode_batch(self.h, _ptr(llr_rows), n, int(max_iterations), float(factor),
                                                       _ptr(out), _ptr(ok), _ptr(it), _stream_ptr()))
        return out, ok, it

    def ldpc_decode_robust(self, llr_rows):
        """robustDecodeSingleCW over a batch: llr_rows float32 [n_cw, 648] -> (bytes, ok, iterations, tries)"""
        n = llr_rows.shape[0]
        assert llr_rows.dtype == torch.float32 and llr_rows.is_contiguous() and llr_rows.shape[1] == 648
        nb = (self.geo.ldpc_k + 7) // 8
        out = torch.empty((n, nb), dtype=torch.uint8, device=self.device)
        ok = torch.empty(n, dtype=torch.uint8, device=self.device)
        it = torch.empty(n, dtype=torch.int16, device=self.device)
        tries = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._check(self.lib.ria_gpu_ldpc_decode_robust_batch(self.h, _ptr(llr_rows), n, _ptr(out), _ptr(ok), _ptr(it),
                                                              _ptr(tries), _stream_ptr()))
        return out, ok, it, tries

    def ldpc_encode(self, info):
        """LDPCEncoder::encode on the host: info uint8 [n_cw, ceil(k/8)] -> coded uint8 [n_cw, 81]."""
        info = np.ascontiguousarray(info, np.uint8)
        out = np.zeros((info.shape[0], 81), np.uint8)
        self._check(self.lib.ria_gpu_ldpc_encode_host(self.h, info.ctypes.data, info.shape[0], out.ctypes.data))
        return out

    def make_frames(self, seed, first_seq, n):
        info = torch.empty((n, self.geo.info_bytes_per_frame), dtype=torch.uint8, device=self.device)
        self._check(self.lib.ria_gpu_make_frames(self.h, int(seed), int(first_seq), n, _ptr(info), _stream_ptr()))
        return info

    def tx(self, info, peak=0.8):
        n = info.shape[0]
        assert info.dtype == torch.uint8 and info.is_contiguous() and info.shape[1] == self.geo.info_bytes_per_frame
        s = torch.empty((n, self.geo.frame_samples), dtype=torch.float32, device=self.device)
        self._check(self.lib.ria_gpu_tx_batch(self.h, _ptr(info), n, float(peak), _ptr(s), _stream_ptr()))
        return s

    def channel_(self, samples, kind, snr_db, seed, first_frame=0):
        n = samples.shape[0]
        assert samples.dtype == torch.float32 and samples.is_contiguous()
        self._check(self.lib.ria_gpu_channel_batch(self.h, int(kind), float(snr_db), int(seed), int(first_frame),
                                                   _ptr(samples), n, _stream_ptr()))
        return samples

    ZC_RESULT = np.dtype([("detected", "<i4"), ("frame_type", "<i4"), ("start_sample", "<i4"), ("root_detected", "<i4"),
                          ("correlation", "<f4"), ("cfo_hz", "<f4"), ("snr_estimate", "<f4"), ("reserved", "<f4")])

    def sync_zc(self, buffers, threshold=0.3, root_mask=15, known_cfo=None):
        """ZCSync::detect over a batch: buffers float32 [n, buf_len] on the device -> structured array."""
        n, buf_len = buffers.shape
        assert buffers.dtype == torch.float32 and buffers.is_contiguous()
        out = torch.zeros((n, 32), dtype=torch.uint8, device=self.device)
        if known_cfo is not None:
            assert known_cfo.dtype == torch.float32 and known_cfo.numel() == n
        self._check(self.lib.ria_gpu_sync_zc_batch(self.h, _ptr(buffers), buf_len, buf_len, n, float(threshold),
                                                   int(root_mask), _ptr(known_cfo), _ptr(out), _stream_ptr()))
        return self._status_array(out, self.ZC_RESULT)

    def zc_preamble(self, root):
        out = np.zeros(4096, np.float32)
        n = self.lib.ria_gpu_zc_preamble(self.h, int(root), out.ctypes.data, len(out))
        if n < 0:
            raise capi.RiaError("zc_preamble: buffer too small")
        return out[:n].copy()

    CHIRP_RESULT = np.dtype([("success", "<i4"), ("up_chirp_start", "<i4"), ("down_chirp_start", "<i4"), ("cfo_hz", "<f4"),
                             ("up_correlation", "<f4"), ("down_correlation", "<f4"), ("reserved", "<i4", 2)])

    def sync_chirp(self, buffers, threshold=0.15):
        """ChirpSync::detectDualChirp over a batch: buffers float32 [n, buf_len] on the device -> structured array."""
        n, buf_len = buffers.shape
        assert buffers.dtype == torch.float32 and buffers.is_contiguous()
        out = torch.zeros((n, 32), dtype=torch.uint8, device=self.device)
        self._check(self.lib.ria_gpu_sync_chirp_batch(self.h, _ptr(buffers), buf_len, buf_len, n, float(threshold),
                                                      _ptr(out), _stream_ptr()))
        return self._status_array(out, self.CHIRP_RESULT)

    def chirp_preamble(self):
        out = np.zeros(60000, np.float32)
        n = self.lib.ria_gpu_chirp_preamble(self.h, out.ctypes.data, len(out))
        if n < 0:
            raise capi.RiaError("chirp_preamble: buffer too small")
        return out[:n].copy()

    LTS_RESULT = np.dtype([("detected", "<i4"), ("start_sample", "<i4"), ("correlation", "<f4"), ("cfo_hz", "<f4"),
                           ("burst_interleaved", "<i4"), ("reserved", "<i4", 3)])

    def sync_lts(self, buffers, known_cfo=None, threshold=0.5):
        """OFDMChirpWaveform::detectDataSync over a batch: buffers float32 [n, buf_len] on the device."""
        n, buf_len = buffers.shape
        assert buffers.dtype == torch.float32 and buffers.is_contiguous()
        out = torch.zeros((n, 32), dtype=torch.uint8, device=self.device)
        self._check(self.lib.ria_gpu_sync_lts_batch(self.h, _ptr(buffers), buf_len, buf_len, n, _ptr(known_cfo), float(threshold),
                                                    _ptr(out), _stream_ptr()))
        return self._status_array(out, self.LTS_RESULT)

    COX_RESULT = np.dtype([("found", "<i4"), ("start_sample", "<i4"), ("cfo_hz", "<f4"), ("noise_floor", "<f4"),
                           ("sts_position", "<i4"), ("reserved", "<i4", 3)])

    def sync_cox(self, buffers, threshold=0.8, noise_floor=None):
        """OFDMDemodulator::searchForSync (Schmidl-Cox, OFDM-COX waveform) over a batch: buffers float32
        [n, buf_len] on the device; noise_floor float32 [n] = the demodulator's tracker before the call."""
        n, buf_len = buffers.shape
        assert buffers.dtype == torch.float32 and buffers.is_contiguous()
        out = torch.zeros((n, 32), dtype=torch.uint8, device=self.device)
        self._check(self.lib.ria_gpu_sync_cox_batch(self.h, _ptr(buffers), buf_len, buf_len, n, float(threshold), _ptr(noise_floor),
                                                    _ptr(out), _stream_ptr()))
        return self._status_array(out, self.COX_RESULT)

    def cox_preamble(self):
        out = np.zeros(10000, np.float32)
        n = self.lib.ria_gpu_cox_preamble(self.h, out.ctypes.data, len(out))
        if n < 0:
            raise capi.RiaError("cox_preamble: buffer too small")
        return out[:n].copy()

    MCDPSK_STATUS = np.dtype([("cfo_hz", "<f4"), ("fading_index", "<f4"), ("freq_fading_index", "<f4"),
                              ("temporal_fading_index", "<f4"), ("training_cfo_residual", "<f4"), ("n_llr", "<i4"),
                              ("valid_symbols", "<i4"), ("reserved", "<i4")])

    def mcdpsk_demod(self, frames, carriers=10, bits_per_symbol=1, spreading=1, cfo_hz=None, phase0=None):
        """MC-DPSK demodulator over a batch: frames float32 [n, (9 + symbols)*512] -> (llr [n, n_llr], status)."""
        n, fs = frames.shape
        assert frames.dtype == torch.float32 and frames.is_contiguous()
        cfg = capi.McdpskConfig(carriers, bits_per_symbol, spreading, 0)
        n_llr = max(1, ((fs - 9 * 512) // 512) // spreading) * carriers * bits_per_symbol
        llr = torch.empty((n, n_llr), dtype=torch.float32, device=self.device)
        st = torch.zeros((n, 32), dtype=torch.uint8, device=self.device)
        self._check(self.lib.ria_gpu_mcdpsk_demod_batch(self.h, C.byref(cfg), _ptr(frames), fs, fs, n, _ptr(cfo_hz), _ptr(phase0),
                                                        _ptr(llr), n_llr, _ptr(st), _stream_ptr()))
        return llr, self._status_array(st, self.MCDPSK_STATUS)

    def mcdpsk_modulate(self, data, carriers=10, bits_per_symbol=1, spreading=1):
        data = np.ascontiguousarray(data, np.uint8)
        cfg = capi.McdpskConfig(carriers, bits_per_symbol, spreading, 0)
        out = np.zeros(600000, np.float32)
        n = self.lib.ria_gpu_mcdpsk_modulate_host(self.h, C.byref(cfg), data.ctypes.data, len(data), out.ctypes.data, len(out))
        if n < 0:
            raise capi.RiaError("mcdpsk_modulate failed")
        return out[:n].copy()

    def mcdpsk_modulate_batch(self, data, carriers=10, bits_per_symbol=1, spreading=1):
        """MultiCarrierDPSKModulator on the device: data uint8 [n, n_bytes] (device tensor or host array) -> float32 [n, samples]"""
        if not torch.is_tensor(data):
            data = torch.from_numpy(np.ascontiguousarray(data, np.uint8)).to(self.device)
        assert data.dtype == torch.uint8 and data.is_contiguous() and data.dim() == 2
        n, nb = data.shape
        bits = carriers * bits_per_symbol
        fs = (9 + ((nb * 8 + bits - 1) // bits) * spreading) * 512
        out = torch.empty((n, fs), dtype=torch.float32, device=self.device)
        cfg = capi.McdpskConfig(carriers, bits_per_symbol, spreading, 0)
        self._check(self.lib.ria_gpu_mcdpsk_modulate_batch(self.h, C.byref(cfg), _ptr(data), nb, n, _ptr(out), fs, _stream_ptr()))
        return out

    def chase_combine(self, acc, count, soft, decoded=None):
        """ChaseCache::store arithmetic in place on acc [n,648] / count [n] (int32); returns stored flags."""
        n = acc.shape[0]
        stored = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self._check(self.lib.ria_gpu_chase_combine_batch(self.h, _ptr(acc), _ptr(count), _ptr(decoded), _ptr(soft), n,
                                                         _ptr(stored), _stream_ptr()))
        return stored

    def channel_exact_(self, samples, kind, snr_db, seed, first_frame=0):
        """Reference-identical channel (mt19937 stream, seed + first_frame + f per frame); in place, any frame length."""
        n, fs = samples.shape
        assert samples.dtype == torch.float32 and samples.is_contiguous()
        self._check(self.lib.ria_gpu_channel_exact_batch(self.h, int(kind), float(snr_db), int(seed) & 0xffffffff, int(first_frame),
                                                         _ptr(samples), fs, fs, n, _stream_ptr()))
        return samples

    def channel_exact_seeded_(self, samples, kind, snr_db, seeds):
        """Reference-identical channel with one mt19937 seed per frame: seeds uint32 tensor/array [n]; in place."""
        n, fs = samples.shape
        assert samples.dtype == torch.float32 and samples.is_contiguous()
        if not torch.is_tensor(seeds):
            seeds = torch.from_numpy(np.ascontiguousarray(seeds, np.uint32).view(np.int32)).to(self.device)
        assert seeds.numel() == n and seeds.element_size() == 4 and seeds.is_contiguous()
        self._check(self.lib.ria_gpu_channel_exact_seeded_batch(self.h, int(kind), float(snr_db), _ptr(seeds), _ptr(samples), fs, fs, n,
                                                                _stream_ptr()))
        return samples

    def channel_exact_cfo_(self, samples, kind, snr_db, seeds, cfo_hz=None, random_cfo_max_hz=0.0):
        """Reference-identical channel incl. its CFO impairment (Config::cfo_hz per frame / random_cfo_max_hz + applyCFO);
        in place.  Returns the per-frame getActualCFO() tensor."""
        n, fs = samples.shape
        assert samples.dtype == torch.float32 and samples.is_contiguous()
        if not torch.is_tensor(seeds):
            seeds = torch.from_numpy(np.ascontiguousarray(seeds, np.uint32).view(np.int32)).to(self.device)
        assert seeds.numel() == n and seeds.element_size() == 4 and seeds.is_contiguous()
        if cfo_hz is not None and not torch.is_tensor(cfo_hz):
            cfo_hz = torch.from_numpy(np.full(n, cfo_hz, np.float32) if np.ndim(cfo_hz) == 0 else np.ascontiguousarray(cfo_hz, np.float32)).to(self.device)
        assert cfo_hz is None or (cfo_hz.dtype == torch.float32 and cfo_hz.numel() == n and cfo_hz.is_contiguous())
        actual = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._check(self.lib.ria_gpu_channel_exact_cfo_batch(self.h, int(kind), float(snr_db), _ptr(seeds), _ptr(cfo_hz), float(random_cfo_max_hz),
                                                             _ptr(actual), _ptr(samples), fs, fs, n, _stream_ptr()))
        return actual

    def tx_cfo(self, samples, cfo_hz, phase=None):
        """SimulatedChannel::applyTxCFO over a batch: samples float32 [n, len] on the device, cfo_hz scalar or [n];
        phase float32 [n] accumulator (updated in place) or None.  Returns the shifted samples."""
        n, L = samples.shape
        assert samples.dtype == torch.float32 and samples.is_contiguous()
        if not torch.is_tensor(cfo_hz):
            cfo_hz = torch.from_numpy(np.full(n, cfo_hz, np.float32) if np.ndim(cfo_hz) == 0 else np.ascontiguousarray(cfo_hz, np.float32)).to(self.device)
        assert cfo_hz.dtype == torch.float32 and cfo_hz.numel() == n and cfo_hz.is_contiguous()
        assert phase is None or (phase.dtype == torch.float32 and phase.numel() == n and phase.is_contiguous())
        out = torch.empty_like(samples)
        self._check(self.lib.ria_gpu_tx_cfo_batch(self.h, _ptr(samples), L, L, n, _ptr(cfo_hz), _ptr(phase), _ptr(out), L, _stream_ptr()))
        return out

    def burst_deinterleave(self, llr, burst_frames):
        """BurstInterleaver::deinterleave: llr float32 [n_groups*N, >=2592] physical -> logical (same shape)."""
        n, stride = llr.shape
        assert llr.dtype == torch.float32 and llr.is_contiguous() and n % burst_frames == 0
        out = torch.zeros_like(llr)
        self._check(self.lib.ria_gpu_burst_deinterleave_batch(self.h, _ptr(llr), stride, burst_frames, n // burst_frames, _ptr(out), _stream_ptr()))
        return out

    def burst_interleave(self, coded, burst_frames):
        """BurstInterleaver::interleave: coded uint8 [n_groups*N, 324] logical -> physical."""
        n = coded.shape[0]
        assert coded.dtype == torch.uint8 and coded.is_contiguous() and coded.shape[1] == 324 and n % burst_frames == 0
        out = torch.zeros_like(coded)
        self._check(self.lib.ria_gpu_burst_interleave_batch(self.h, _ptr(coded), burst_frames, n // burst_frames, _ptr(out), _stream_ptr()))
        return out

    def debug_math(self, op, a, b=None):
        out = torch.empty_like(a)
        self._check(self.lib.ria_gpu_debug_math(self.h, op, _ptr(a), _ptr(b), a.numel(), _ptr(out), _stream_ptr()))
        return out

    def frame_status(self, st):
        return self._status_array(st, self.FRAME_STATUS)

    def decode_status(self, st):
        return self._status_array(st, self.DECODE_STATUS)
