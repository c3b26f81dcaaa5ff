"""ctypes binding of include/ria_gpu.h.  Device memory comes from torch (plumbing only): every
wrapper takes CUDA/HIP tensors and passes raw device pointers across the C ABI."""
import ctypes as C
import os

from . import build as _build

RIA_OK = 0
MOD = {"DBPSK": 0, "BPSK": 1, "DQPSK": 2, "QPSK": 3, "D8PSK": 4, "QAM16": 6, "QAM32": 7, "QAM64": 8, "QAM256": 10}
RATE = {"R1_4": 0, "R1_3": 1, "R1_2": 2, "R2_3": 3, "R3_4": 4, "R5_6": 5}
DECODE_PHASE0, DECODE_PERTURB, DECODE_CRC_RECOVER, DECODE_FULL = 1, 2, 4, 7
DECODE_NO_CHANNEL_DEINTERLEAVE = 0x100
RX_DEMOD_ONLY = 0x200
OPT_SPLIT_PARTS = 1
OPT_DUAL_DECODER = 2

# every symbol include/ria_gpu.h declares
EXPORTS = [
    "ria_gpu_abi_version", "ria_gpu_default_config", "ria_gpu_create", "ria_gpu_destroy", "ria_gpu_last_error",
    "ria_gpu_get_geometry", "ria_gpu_set_option", "ria_gpu_demod_batch", "ria_gpu_decode_batch", "ria_gpu_ldpc_decode_batch", "ria_gpu_ldpc_decode_robust_batch",
    "ria_gpu_rx_batch", "ria_gpu_rx_frames_host", "ria_gpu_decode_frames_host", "ria_gpu_tx_batch", "ria_gpu_make_frames",
    "ria_gpu_channel_batch", "ria_gpu_channel_exact_batch", "ria_gpu_channel_exact_seeded_batch", "ria_gpu_debug_math", "ria_gpu_debug_queue_fault", "ria_gpu_sync_zc_batch", "ria_gpu_zc_preamble", "ria_gpu_sync_chirp_batch", "ria_gpu_chirp_preamble", "ria_gpu_mcdpsk_demod_batch",
    "ria_gpu_mcdpsk_modulate_host", "ria_gpu_chase_combine_batch", "ria_gpu_sync_lts_batch", "ria_gpu_sync_host", "ria_gpu_ldpc_encode_host", "ria_gpu_burst_deinterleave_batch", "ria_gpu_burst_interleave_batch",
    "ria_gpu_sync_cox_batch", "ria_gpu_cox_preamble", "ria_gpu_channel_exact_cfo_batch", "ria_gpu_tx_cfo_batch",
    "ria_gpu_mcdpsk_demod_host", "ria_gpu_ldpc_decode_robust_host", "ria_gpu_mcdpsk_modulate_batch",
    "ria_link_recommend", "ria_link_data_mode", "ria_link_ofdm_code_rate", "ria_link_cap_initial_rate",
]


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "device", "modulation", "code_rate", "fft_size",
                                          "num_carriers", "cyclic_prefix", "sample_rate", "center_freq",
                                          "max_batch")] + [("reserved", C.c_int32 * 6)]


class LinkRecommendation(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("waveform", "modulation", "code_rate", "spreading", "num_carriers")] + \
               [("estimated_throughput_bps", C.c_float)]


class McdpskConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_carriers", "bits_per_symbol", "spreading", "reserved")]


class Geometry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("pilot_spacing", "n_pilots", "n_data_carriers", "bits_per_carrier",
                                          "bits_per_symbol", "n_data_symbols", "samples_per_symbol",
                                          "frame_samples", "llrs_per_frame", "info_bits", "bytes_per_codeword",
                                          "info_bytes_per_frame", "ldpc_max_iterations", "ldpc_edges", "ldpc_k")] + \
               [("reserved", C.c_int32 * 1)]


class FrameMeta(C.Structure):
    _fields_ = [("cfo_hz", C.c_float), ("flags", C.c_uint32), ("abs_position", C.c_uint64)]


class FrameStatus(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("snr_db", "cfo_hz", "fading_index", "noise_variance", "lts_phase_slope",
                                          "snr_linear", "corr_phase")] + [("n_llr", C.c_int32)]


class DecodeStatus(C.Structure):
    _fields_ = [("cw_ok", C.c_uint8 * 4), ("iterations", C.c_uint16 * 4), ("attempts", C.c_uint8 * 4),
                ("frame_valid", C.c_uint8), ("needs_recovery", C.c_uint8), ("reserved", C.c_uint8 * 2)]


_lib = None


def library_path():
    return _build.LIB


def load(build_if_needed=True):
    """Loads libria_gpu.so; raises (never falls back) if it cannot be built or loaded."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (soname libamdhip64.so.7).  Import it first so that this
    # process holds ONE runtime: libria_gpu.so's DT_NEEDED then resolves to the copy torch loaded,
    # and device pointers / streams from torch are valid inside the library.
    import torch  # noqa: F401
    path = _build.build() if build_if_needed else _build.LIB
    path = os.environ.get("RIA_GPU_LIB", path)   # developer A/B switch: an alternative build of the same sources
    if not os.path.exists(path):
        raise RuntimeError("libria_gpu.so is missing: run `python -m ria_amd.build`")
    L = C.CDLL(path)
    vp, i32, u32, u64, f32 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float
    L.ria_gpu_abi_version.restype = i32
    L.ria_gpu_default_config.argtypes = [C.POINTER(Config)]
    L.ria_gpu_default_config.restype = None
    L.ria_gpu_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.ria_gpu_destroy.argtypes = [vp]
    L.ria_gpu_destroy.restype = None
    L.ria_gpu_last_error.argtypes = [vp]
    L.ria_gpu_last_error.restype = C.c_char_p
    L.ria_gpu_get_geometry.argtypes = [vp, C.POINTER(Geometry)]
    L.ria_gpu_set_option.argtypes = [vp, i32, i32]
    L.ria_gpu_demod_batch.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
    L.ria_gpu_decode_batch.argtypes = [vp, vp, i32, i32, u32, vp, vp, vp]
    L.ria_gpu_ldpc_decode_batch.argtypes = [vp, vp, i32, i32, f32, vp, vp, vp, vp]
    L.ria_gpu_ldpc_decode_robust_batch.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
    L.ria_gpu_rx_batch.argtypes = [vp, vp, vp, vp, i32, u32, vp, vp, vp, vp, vp]
    L.ria_gpu_rx_frames_host.argtypes = [vp, vp, vp, i32, u32, vp, vp, vp, vp]
    L.ria_gpu_decode_frames_host.argtypes = [vp, vp, i32, i32, u32, vp, vp]
    L.ria_gpu_tx_batch.argtypes = [vp, vp, i32, f32, vp, vp]
    L.ria_gpu_make_frames.argtypes = [vp, u64, i32, i32, vp, vp]
    L.ria_gpu_channel_batch.argtypes = [vp, i32, f32, u64, u64, vp, i32, vp]
    L.ria_gpu_debug_math.argtypes = [vp, i32, vp, vp, i32, vp, vp]
    L.ria_gpu_debug_queue_fault.argtypes = [vp]
    L.ria_gpu_channel_exact_batch.argtypes = [vp, i32, f32, u32, u64, vp, C.c_int64, i32, i32, vp]
    L.ria_gpu_channel_exact_seeded_batch.argtypes = [vp, i32, f32, vp, vp, C.c_int64, i32, i32, vp]
    L.ria_gpu_channel_exact_cfo_batch.argtypes = [vp, i32, f32, vp, vp, f32, vp, vp, C.c_int64, i32, i32, vp]
    L.ria_gpu_mcdpsk_modulate_batch.argtypes = [vp, vp, vp, i32, i32, vp, C.c_int64, vp]
    L.ria_gpu_mcdpsk_demod_host.argtypes = [vp, vp, vp, i32, f32, f32, vp, i32, vp]
    L.ria_gpu_ldpc_decode_robust_host.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    L.ria_gpu_tx_cfo_batch.argtypes = [vp, vp, C.c_int64, i32, i32, vp, vp, vp, C.c_int64, vp]
    L.ria_gpu_sync_zc_batch.argtypes = [vp, vp, C.c_int64, i32, i32, f32, u32, vp, vp, vp]
    L.ria_gpu_zc_preamble.argtypes = [vp, i32, vp, i32]
    L.ria_gpu_sync_chirp_batch.argtypes = [vp, vp, C.c_int64, i32, i32, f32, vp, vp]
    L.ria_gpu_chirp_preamble.argtypes = [vp, vp, i32]
    L.ria_gpu_mcdpsk_demod_batch.argtypes = [vp, vp, vp, C.c_int64, i32, i32, vp, vp, vp, i32, vp, vp]
    L.ria_gpu_mcdpsk_modulate_host.argtypes = [vp, vp, vp, i32, vp, i32]
    L.ria_gpu_chase_combine_batch.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp]
    L.ria_gpu_sync_lts_batch.argtypes = [vp, vp, C.c_int64, i32, i32, vp, f32, vp, vp]
    L.ria_gpu_sync_cox_batch.argtypes = [vp, vp, C.c_int64, i32, i32, f32, vp, vp, vp]
    L.ria_gpu_cox_preamble.argtypes = [vp, vp, i32]
    L.ria_gpu_sync_host.argtypes = [vp, i32, vp, i32, f32, f32, u32, vp]
    L.ria_gpu_ldpc_encode_host.argtypes = [vp, vp, i32, vp]
    L.ria_gpu_burst_deinterleave_batch.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.ria_gpu_burst_interleave_batch.argtypes = [vp, vp, i32, i32, vp, vp]
    L.ria_link_recommend.argtypes = [f32, f32, C.POINTER(LinkRecommendation)]
    L.ria_link_recommend.restype = None
    L.ria_link_data_mode.argtypes = [f32, i32, f32, C.POINTER(LinkRecommendation)]
    L.ria_link_data_mode.restype = None
    L.ria_link_ofdm_code_rate.argtypes = [f32, f32]
    L.ria_link_cap_initial_rate.argtypes = [f32, f32, i32]
    for name in EXPORTS:
        if name not in ("ria_gpu_default_config", "ria_gpu_destroy", "ria_gpu_last_error", "ria_link_recommend", "ria_link_data_mode"):
            getattr(L, name).restype = i32
    _lib = L
    return L


class RiaError(RuntimeError):
    pass
