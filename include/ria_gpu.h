/* include/ria_gpu.h — C ABI of libria_gpu.so: the MI355X (gfx950) RX signal chain for the RIA modem.
 *
 * This is the drop-in boundary.  Every entry point replaces one call the unmodified
 * host code (gui::StreamingDecoder, tools/cli_simulator, tools/test_waveform_simple) makes today;
 * the reference interface each one stands in for is cited as file:line relative to the reference
 * repository.  INTEGRATION.md shows the adaptor class a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain C, no exceptions: every function returns RIA_OK (0) or a negative ria_status; the text
 *     of the last error of a handle is available from ria_gpu_last_error().
 *   - "dev" pointers are device (HBM) addresses valid on the handle's GPU; "host" variants copy
 *     over PCIe themselves.  `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     device-pointer calls are asynchronous on that stream, host-pointer calls return when done.
 *   - a handle is bound to one (modulation, code rate) pair like one configured IWaveform object
 *     (waveform_interface.hpp:69 configure()); it is not thread-safe, like the reference
 *     (streaming_decoder.cpp:719 holds waveform_mutex_ around every call).
 *   - enum values are the reference's own (include/ultra/types.hpp:28-39, :91-100).
 */
#ifndef RIA_GPU_H
#define RIA_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RIA_GPU_ABI_VERSION 1

typedef enum ria_status {
    RIA_OK = 0,
    RIA_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
    RIA_ERR_NO_DEVICE = -2,    /* no gfx950 device or HIP runtime failure at create */
    RIA_ERR_HIP = -3,          /* a HIP call failed (see ria_gpu_last_error) */
    RIA_ERR_UNSUPPORTED = -4   /* valid request this build does not implement yet */
} ria_status;

/* Modulation (types.hpp:28-39) and CodeRate (types.hpp:91-100) */
enum { RIA_MOD_DBPSK = 0, RIA_MOD_BPSK = 1, RIA_MOD_DQPSK = 2, RIA_MOD_QPSK = 3, RIA_MOD_D8PSK = 4,
       RIA_MOD_QAM16 = 6, RIA_MOD_QAM32 = 7, RIA_MOD_QAM64 = 8, RIA_MOD_QAM256 = 10 };
enum { RIA_RATE_1_4 = 0, RIA_RATE_1_3 = 1, RIA_RATE_1_2 = 2, RIA_RATE_2_3 = 3, RIA_RATE_3_4 = 4,
       RIA_RATE_5_6 = 5 };

/* Immutable per-handle configuration: the subset of ultra::ModemConfig (types.hpp:193-289) the
 * OFDM-CHIRP RX path reads.  Zero-initialise, then ria_gpu_default_config(). */
typedef struct ria_gpu_config {
    int32_t abi_version;     /* RIA_GPU_ABI_VERSION */
    int32_t device;          /* HIP device ordinal */
    int32_t modulation;      /* RIA_MOD_* */
    int32_t code_rate;       /* RIA_RATE_* */
    int32_t fft_size;        /* 1024 */
    int32_t num_carriers;    /* 59 */
    int32_t cyclic_prefix;   /* 128 (CyclicPrefixMode::LONG at FFT 1024) */
    int32_t sample_rate;     /* 48000 */
    int32_t center_freq;     /* 1500 */
    int32_t max_batch;       /* frames per call the workspace is sized for */
    int32_t reserved[6];
} ria_gpu_config;

/* Frame geometry derived from the configuration (ofdm_chirp_waveform.cpp:616-648,
 * ofdm_link_adaptation.hpp:26-70, frame_v2.hpp:671-691). */
typedef struct ria_gpu_geometry {
    int32_t pilot_spacing, n_pilots, n_data_carriers;
    int32_t bits_per_carrier, bits_per_symbol;
    int32_t n_data_symbols;      /* data symbols in a fixed 4-codeword frame */
    int32_t samples_per_symbol;  /* 1152 */
    int32_t frame_samples;       /* (2 LTS + n_data_symbols) * 1152 = 18432 for QAM16 R1/2 */
    int32_t llrs_per_frame;      /* n_data_symbols * bits_per_symbol = 2632 */
    int32_t info_bits, bytes_per_codeword, info_bytes_per_frame; /* 324, 40, 160 */
    int32_t ldpc_max_iterations; /* ldpc_codec.hpp:86-95 */
    int32_t ldpc_edges;
    int32_t ldpc_k;              /* information bits of the LDPC code itself (ldpc_decoder.cpp:21-36): what the single-codeword
                                    decoders return, ceil(ldpc_k / 8) bytes.  Equals info_bits except at R1/3, whose table entry
                                    uses the (324, 324) code while the frame layer counts 216 bits (27 bytes) per codeword */
    int32_t reserved[1];
} ria_gpu_geometry;

/* Per-frame input of the demodulator: the three setters the host calls before process()
 * (streaming_decoder.cpp:896 setAbsoluteTrainingPosition, :1347 setFrequencyOffset). */
typedef struct ria_frame_meta {
    float    cfo_hz;          /* IWaveform::setFrequencyOffset */
    uint32_t flags;           /* bit0: first LTS symbol is negated (burst marker, ofdm_chirp_waveform.cpp:421-440) */
    uint64_t abs_position;    /* IWaveform::setAbsoluteTrainingPosition */
} ria_frame_meta;

/* Per-frame output of the demodulator: what the host reads back through IWaveform
 * (estimatedSNR :151, estimatedCFO :154, getFadingIndex :159) plus estimator taps for parity tests. */
typedef struct ria_frame_status {
    float snr_db;            /* OFDMDemodulator::getEstimatedSNR */
    float cfo_hz;            /* corrected CFO fed back to the waveform (ofdm_chirp_waveform.cpp:457-464) */
    float fading_index;      /* last_fading_index */
    float noise_variance;    /* LTS noise variance */
    float lts_phase_slope;
    float snr_linear;
    float corr_phase;        /* freq_correction_phase after the last sample */
    int32_t n_llr;           /* soft bits produced (0 if process() would have returned false) */
} ria_frame_status;

/* Per-frame output of decodeFixedFrame (frame_v2.hpp:637-664 CodewordStatus). */
typedef struct ria_decode_status {
    uint8_t  cw_ok[4];        /* CodewordStatus::decoded */
    uint16_t iterations[4];   /* LDPCDecoder::lastIterations() of the accepted (or last) attempt */
    uint8_t  attempts[4];     /* 1 = first decode, 2..5 phase 0, 6.. retry phases 1-6 */
    uint8_t  frame_valid;     /* 1: header+frame CRC verified on the reassembled frame */
    uint8_t  needs_recovery;  /* 1: all codewords converged but the frame CRC failed (LDPC false
                                 positive, frame_v2.cpp:1564-1880): ria_gpu_decode_* finishes it */
    uint8_t  reserved[2];     /* reserved[1] == 0xEE: an internal work queue of this call broke its bound; then every frame of the
                                 call is reported failed (cw_ok 0, frame_valid 0, zero bytes) and the host-buffer forms return RIA_ERR_HIP */
} ria_decode_status;

/* decode flags */
#define RIA_DECODE_PHASE0      0x1u  /* min-sum factor diversity retries   (frame_v2.cpp:1398-1413) */
#define RIA_DECODE_PERTURB     0x2u  /* stochastic retry phases 1-6        (frame_v2.cpp:1415-1546) */
#define RIA_DECODE_CRC_RECOVER 0x4u  /* CRC-guided false-positive recovery (frame_v2.cpp:1564-1880) */
#define RIA_DECODE_FULL        0x7u  /* exactly v2::decodeFixedFrame */
#define RIA_DECODE_NO_CHANNEL_DEINTERLEAVE 0x100u
#define RIA_RX_DEMOD_ONLY      0x200u  /* ria_gpu_rx_frames_host: process() + getSoftBits() only, no decode (info / decode
                                          status pointers may be NULL, llr_out_host must not be) */

typedef struct ria_gpu* ria_gpu_handle;

/* ---- lifecycle ------------------------------------------------------------------------------ */
int  ria_gpu_abi_version(void);
void ria_gpu_default_config(ria_gpu_config* cfg);
/* replaces: std::make_unique<OFDMChirpWaveform>(config) + configure(mod, rate)
 * (streaming_decoder.cpp:2299,2328; ofdm_chirp_waveform.cpp:81-107) */
int  ria_gpu_create(const ria_gpu_config* cfg, ria_gpu_handle* out);
void ria_gpu_destroy(ria_gpu_handle h);
const char* ria_gpu_last_error(ria_gpu_handle h);
int  ria_gpu_get_geometry(ria_gpu_handle h, ria_gpu_geometry* out);

/* Execution knobs of a handle.  None of them changes a result; the parity tests run the fused call under every
 * value and compare.
 *   RIA_OPT_SPLIT_PARTS  ria_gpu_rx_batch cuts a batch of >= 4096 frames into this many parts that run on internal
 *                        streams (1..4; 1 = one stream, no overlap; 0 = library default (3), which the environment
 *                        variables RIA_SPLIT_PARTS / RIA_NO_SPLIT may override). */
#define RIA_OPT_SPLIT_PARTS 1
/*   RIA_OPT_DUAL_DECODER the retry kernels (phase 0, cascade) decode two codewords per wavefront on an interleaved LDS
 *                        image (ldpc_dual.hip.h): 1 = on, -1 = off, 0 = library default (off).  An experiment record that
 *                        measured slower: only in builds made with -DRIA_WITH_DUAL_DECODER; elsewhere 1 is RIA_ERR_UNSUPPORTED. */
#define RIA_OPT_DUAL_DECODER 2
int  ria_gpu_set_option(ria_gpu_handle h, int option, int value);

/* ---- RX: demodulate  (IWaveform::process + getSoftBits, waveform_interface.hpp:124,135;
 *          OFDMChirpWaveform::process ofdm_chirp_waveform.cpp:391-468) ------------------------- */
/* samples_dev: frame f starts at samples_dev + (frame_offsets_dev ? frame_offsets_dev[f] : f*frame_samples)
 * and must hold frame_samples floats from the first LTS sample on.  meta_dev may be NULL (cfo 0).
 * llr_out_dev: n_frames * llrs_per_frame floats.  status_dev may be NULL. */
int ria_gpu_demod_batch(ria_gpu_handle h, const float* samples_dev, const uint64_t* frame_offsets_dev,
                        const ria_frame_meta* meta_dev, int n_frames,
                        float* llr_out_dev, ria_frame_status* status_dev, void* stream);

/* ---- RX: decode  (protocol::v2::decodeFixedFrame, frame_v2.hpp:848, frame_v2.cpp:1335-1883) --- */
/* llr_dev: frame f at llr_dev + f*llr_stride (first 2592 used).  info_out_dev: n_frames *
 * info_bytes_per_frame.  status_dev: n_frames entries. */
int ria_gpu_decode_batch(ria_gpu_handle h, const float* llr_dev, int llr_stride, int n_frames,
                         uint32_t flags, uint8_t* info_out_dev, ria_decode_status* status_dev, void* stream);

/* Single-codeword decoder (LDPCDecoder::decodeSoft, include/ultra/fec.hpp:48-81): n_cw rows of 648
 * LLRs already in decoder order; out: n_cw * ceil(ldpc_k/8) bytes (ria_gpu_geometry.ldpc_k); ok/iters: n_cw entries. */
int ria_gpu_ldpc_decode_batch(ria_gpu_handle h, const float* llr_dev, int n_cw, int max_iterations,
                              float min_sum_factor, uint8_t* out_dev, uint8_t* ok_dev,
                              uint16_t* iters_dev, void* stream);

/* robustDecodeSingleCW (src/gui/modem/streaming_decoder.cpp:1028-1058; the per-codeword decoder of the MC-DPSK and
 * control-frame paths, :1290,:1454,:2620): a fresh LDPCDecoder at getRecommendedIterations(rate), min-sum factor
 * 0.9375, then 0.875 / 0.75 / 0.625 / 0.5 until one converges.  n_cw rows of 648 LLRs in decoder order; out: n_cw *
 * ceil(ldpc_k/8) bytes (the last attempt's hard bits; the reference returns them only when ok); tries_dev (nullable):
 * decodes made, 1..5; iters_dev: lastIterations() of the last one. */
int ria_gpu_ldpc_decode_robust_batch(ria_gpu_handle h, const float* llr_dev, int n_cw, uint8_t* out_dev, uint8_t* ok_dev,
                                     uint16_t* iters_dev, uint8_t* tries_dev, void* stream);

/* ---- RX: fused samples -> payload (process + getSoftBits + decodeFixedFrame in one pass) ------ */
/* llr_out_dev and demod_status_dev may be NULL. */
int ria_gpu_rx_batch(ria_gpu_handle h, const float* samples_dev, const uint64_t* frame_offsets_dev,
                     const ria_frame_meta* meta_dev, int n_frames, uint32_t flags,
                     uint8_t* info_out_dev, ria_decode_status* decode_status_dev,
                     float* llr_out_dev, ria_frame_status* demod_status_dev, void* stream);

/* Host-buffer forms for the single-frame IWaveform adaptor (n_frames small): the caller's buffers are ordinary host
 * memory; the library stages them through a pinned + device block it keeps for the life of the handle (no allocation
 * per call), runs on its own stream and returns when the results are in the caller's buffers. */
int ria_gpu_rx_frames_host(ria_gpu_handle h, const float* samples_host, const ria_frame_meta* meta_host,
                           int n_frames, uint32_t flags, uint8_t* info_out_host,
                           ria_decode_status* decode_status_host, float* llr_out_host,
                           ria_frame_status* demod_status_host);

/* Host-buffer decodeFixedFrame (llr_host: n_frames rows of llr_stride floats, first 2592 used). */
int ria_gpu_decode_frames_host(ria_gpu_handle h, const float* llr_host, int llr_stride, int n_frames, uint32_t flags,
                               uint8_t* info_out_host, ria_decode_status* status_host);

/* ---- TX synthesis for Monte-Carlo sweeps (v2::encodeFixedFrame frame_v2.cpp:1285-1328 +
 *      OFDMModulator::generateTrainingSymbols/modulate modulator.cpp:534-583, :348-477) ---------- */
/* info_dev: n_frames * info_bytes_per_frame (already serialized frames, zero padded);
 * samples_out_dev: n_frames * frame_samples.  peak_normalize: scale every frame to this peak
 * (0 = leave the modulator's output_scale 40 level; tools/test_waveform_simple.cpp:365-371 uses 0.8) */
int ria_gpu_tx_batch(ria_gpu_handle h, const uint8_t* info_dev, int n_frames, float peak_normalize,
                     float* samples_out_dev, void* stream);

/* Builds serialized v2 data frames (makeFixedDataFrame("TEST","RX",seq,payload).serialize(),
 * frame_v2.cpp:1890-1912, :502-554) with payload bytes drawn from a counter RNG: seq = first_seq + f. */
int ria_gpu_make_frames(ria_gpu_handle h, uint64_t seed, int first_seq, int n_frames,
                        uint8_t* info_out_dev, void* stream);

/* ---- channel simulator (sim::WattersonChannel, src/sim/hf_channel.hpp:35-303, presets :411-488)
 * kind: 0 awgn, 1 good, 2 moderate, 3 poor, 4 flutter.  In place on n_frames * frame_samples.
 * Frame f uses the counter-RNG stream (seed, first_frame + f): results do not depend on how frames
 * are split over calls or GPUs.  Statistical (not bit) parity with the reference's mt19937 stream. */
int ria_gpu_channel_batch(ria_gpu_handle h, int kind, float snr_db, uint64_t seed, uint64_t first_frame,
                          float* samples_dev, int n_frames, void* stream);

/* ---- acquisition: Zadoff-Chu preamble (sync::ZCSync, src/sync/zc_sync.hpp) -----------------------
 * ria_gpu_sync_zc_batch replaces ZCSync::detect(samples, threshold, debug=false, root_mask, known_cfo_hz)
 * (zc_sync.hpp:192-391) for n_buffers capture buffers of buf_len samples each (buffer b starts at
 * samples_dev + b*stride).  root_mask bits 0..3 = PING/PONG/DATA/CONTROL roots 1/3/5/7 (ZC_ROOT_MASK_*).
 * known_cfo_dev: per-buffer known CFO in Hz, or NULL for 0.  Results are bit-identical to the reference's
 * ZCSyncResult for every field (snr_estimate included).  buf_len <= 1048576: buffers up to 16384 samples are mixed down
 * into the workgroup's LDS (the batched acquisition sweeps), longer ones (the host's connected-mode search windows of
 * 31 000 - 48 000 samples, streaming_decoder.cpp:424-431) into a device workspace the handle keeps. */
typedef struct ria_zc_result {
    int32_t detected;        /* ZCSyncResult::detected */
    int32_t frame_type;      /* ZCFrameType: 0 PING 1 PONG 2 DATA 3 CONTROL 255 UNKNOWN */
    int32_t start_sample;    /* first sample after the preamble, -1 if not detected */
    int32_t root_detected;   /* best root even below threshold, -1 if none */
    float correlation;
    float cfo_hz;
    float snr_estimate;
    float reserved;
} ria_zc_result;             /* 32 bytes */
int ria_gpu_sync_zc_batch(ria_gpu_handle h, const float* samples_dev, int64_t stride, int buf_len, int n_buffers,
                          float threshold, uint32_t root_mask, const float* known_cfo_dev,
                          ria_zc_result* out_dev, void* stream);
/* ZCSync::generatePreambleForRoot (zc_sync.hpp:133-190) into a HOST buffer (2512 samples); returns the
 * sample count, or -needed if max_n is too small.  Bit-identical audio. */
int ria_gpu_zc_preamble(ria_gpu_handle h, int root, float* out_host, int max_n);

/* ---- acquisition: dual chirp (sync::ChirpSync, src/sync/chirp_sync.hpp) ----------------------------
 * ria_gpu_sync_chirp_batch replaces ChirpSync::detectDualChirp(samples, threshold) (chirp_sync.hpp:352-512)
 * with the OFDM-CHIRP configuration (300 -> 2700 Hz, 500 ms, 100 ms gap, dual chirp; getChirpConfig,
 * ofdm_chirp_waveform.cpp:46-56) for n_buffers buffers of buf_len samples (buffer b at samples_dev + b*stride).
 * Every field is bit-identical to the reference's DualChirpResult. */
typedef struct ria_chirp_result {
    int32_t success;
    int32_t up_chirp_start;      /* CFO-corrected, -1 if not detected */
    int32_t down_chirp_start;
    float cfo_hz;
    float up_correlation;
    float down_correlation;
    int32_t reserved[2];
} ria_chirp_result;              /* 32 bytes */
int ria_gpu_sync_chirp_batch(ria_gpu_handle h, const float* samples_dev, int64_t stride, int buf_len, int n_buffers,
                             float threshold, ria_chirp_result* out_dev, void* stream);
/* ChirpSync::generate (chirp_sync.hpp:61-108): the 57 600-sample dual-chirp preamble into a HOST buffer;
 * returns the sample count or -needed. */
int ria_gpu_chirp_preamble(ria_gpu_handle h, float* out_host, int max_n);

/* ---- acquisition: LTS light sync (OFDMChirpWaveform::detectDataSync, ofdm_chirp_waveform.cpp:207-384) -------
 * Training-only preamble of connected-mode DATA frames: energy gate, Hilbert-65 analytic signal, one-symbol
 * autocorrelation (coarse step 8, early exit above 0.95, +-4 refinement), burst-interleave marker.
 * Replaces detectDataSync(samples, result, known_cfo_hz, threshold) for n_buffers buffers; every field of
 * SyncResult it sets is bit-identical (start_sample = first sample of the first LTS symbol). */
typedef struct ria_lts_result {
    int32_t detected;
    int32_t start_sample;
    float correlation;
    float cfo_hz;               /* = known_cfo_hz (SyncResult::cfo_hz) */
    int32_t burst_interleaved;  /* wasBurstInterleaved() */
    int32_t reserved[3];
} ria_lts_result;               /* 32 bytes */
int ria_gpu_sync_lts_batch(ria_gpu_handle h, const float* samples_dev, int64_t stride, int buf_len, int n_buffers,
                           const float* known_cfo_dev, float threshold, ria_lts_result* out_dev, void* stream);

/* ---- Schmidl-Cox acquisition of the OFDM-COX waveform
 * Replaces OFDMDemodulator::searchForSync(samples, out_position, out_cfo_hz, threshold)
 * (src/ofdm/demodulator.cpp:1450-1542) as OFDMNvisWaveform::detectSync drives it
 * (src/waveform/ofdm_cox_waveform.cpp:125-158), for n_buffers capture buffers: energy gate with the
 * demodulator's noise-floor tracker (ofdm_sync.cpp:20-50), half-symbol Schmidl-Cox metric on the FFT-Hilbert
 * analytic signal (ofdm_sync.cpp:56-86,118-163) on the 64-sample search grid and the 8-sample plateau grid,
 * plateau rule (>= 15 of 38 points at >= 0.90), passband LTS fine timing with the earlier-LTS preference and
 * the 0.05 confirmation threshold (ofdm_sync.cpp:386-484), coarse CFO (ofdm_sync.cpp:230-261).  All fields
 * bit-identical.  noise_floor_dev (may be NULL = fresh demodulator, 0) holds Impl::noise_floor_energy per buffer
 * before the call; the value after the call is returned in the result.  buf_len <= 240000
 * (MAX_BUFFER_SAMPLES). */
typedef struct ria_cox_result {
    int32_t found;
    int32_t start_sample;       /* first sample of the first LTS symbol (SyncResult::start_sample) */
    float cfo_hz;
    float noise_floor;          /* Impl::noise_floor_energy after the search */
    int32_t sts_position;       /* Schmidl-Cox plateau peak the LTS refinement started from */
    int32_t reserved[3];
} ria_cox_result;               /* 32 bytes */
int ria_gpu_sync_cox_batch(ria_gpu_handle h, const float* samples_dev, int64_t stride, int buf_len, int n_buffers,
                           float threshold, const float* noise_floor_dev, ria_cox_result* out_dev, void* stream);
/* OFDMModulator::generatePreamble (src/ofdm/modulator.cpp:479-532) for the handle's configuration: one symbol
 * of silence, 4 STS, 2 LTS = 8064 samples, bit-identical.  Returns the sample count (negative = needed). */
int ria_gpu_cox_preamble(ria_gpu_handle h, float* out_host, int max_n);

/* Single-buffer convenience forms for the IWaveform adaptor (host memory in, result by value; they stage
 * through device memory and synchronise).  kind: 0 dual chirp (ria_chirp_result), 1 LTS light sync
 * (ria_lts_result), 2 ZC (ria_zc_result), 3 Schmidl-Cox (ria_cox_result); param = known CFO in Hz (kinds 1, 2) or
 * the initial noise floor (kind 3), root_mask only for kind 2. */
int ria_gpu_sync_host(ria_gpu_handle h, int kind, const float* samples_host, int n_samples, float threshold, float param,
                      uint32_t root_mask, void* result_out /* 32 bytes */);

/* ---- MC-DPSK demodulator (src/psk/multi_carrier_dpsk.hpp) and HARQ chase combine (src/fec/chase_cache.cpp)
 * ria_gpu_mcdpsk_demod_batch replaces MultiCarrierDPSKDemodulator as MCDPSKWaveform::process drives it after
 * an external chirp detection (setChirpDetected + process, multi_carrier_dpsk.hpp:797-896): each frame is
 * training (8 x 512) + reference (512) + data symbols, frame f at samples_dev + f*stride; per-frame CFO (Hz)
 * and initial CFO phase (rad) may be NULL (0).  LLRs come out as demodulateSoft returns them
 * ((frame_samples/512 - 9)/spreading * carriers * bits_per_symbol values), bit-identical. */
typedef struct ria_mcdpsk_config {
    int32_t num_carriers;      /* 3..20 (reference default 8; MC-DPSK modes use 10) */
    int32_t bits_per_symbol;   /* 1 DBPSK, 2 DQPSK */
    int32_t spreading;         /* 1, 2 or 4 (SpreadingMode) */
    int32_t reserved;
} ria_mcdpsk_config;
typedef struct ria_mcdpsk_status {
    float cfo_hz;                  /* getEstimatedCFO() after the frame */
    float fading_index;            /* getFadingIndex() */
    float freq_fading_index;
    float temporal_fading_index;
    float training_cfo_residual;   /* processTraining's estimate (not applied after an external chirp) */
    int32_t n_llr;
    int32_t valid_symbols;
    int32_t reserved;
} ria_mcdpsk_status;               /* 32 bytes */
int ria_gpu_mcdpsk_demod_batch(ria_gpu_handle h, const ria_mcdpsk_config* cfg, const float* samples_dev, int64_t stride,
                               int frame_samples, int n_frames, const float* cfo_hz_dev, const float* phase0_dev,
                               float* llr_out_dev, int llr_stride, ria_mcdpsk_status* status_dev, void* stream);
/* Host-buffer forms for the single-frame MC-DPSK plug-in adaptor (MCDPSKWaveform::process -> getSoftBits,
 * src/waveform/mc_dpsk_waveform.cpp:294-338; robustDecodeSingleCW, streaming_decoder.cpp:1028-1058): ordinary host memory
 * in and out, staged on the handle's own stream, return when done.  demod: one frame (training + reference + data,
 * n_samples >= 10 * 512); llr_out_host gets the soft bits as demodulateSoft returns them (count in status_out->n_llr). */
int ria_gpu_mcdpsk_demod_host(ria_gpu_handle h, const ria_mcdpsk_config* cfg, const float* samples_host, int n_samples, float cfo_hz,
                              float phase0, float* llr_out_host, int max_llr, ria_mcdpsk_status* status_out);
int ria_gpu_ldpc_decode_robust_host(ria_gpu_handle h, const float* llr_host, int n_cw, uint8_t* out_host, uint8_t* ok_host,
                                    uint16_t* iters_host /* nullable */, uint8_t* tries_host /* nullable */);
/* MultiCarrierDPSKModulator: generateTrainingSequence + generateReferenceSymbol + modulate(data) into a HOST
 * buffer (multi_carrier_dpsk.hpp:141-281); returns the sample count or -needed.  Bit-identical audio. */
int ria_gpu_mcdpsk_modulate_host(ria_gpu_handle h, const ria_mcdpsk_config* cfg, const uint8_t* data, int n_bytes,
                                 float* out_host, int max_n);
/* The same modulator for a batch on the device: data_dev = n_frames rows of n_bytes coded bytes, frame f written to
 * out_dev + f*out_stride ((9 + ceil(8 n_bytes / (carriers * bits)) * spreading) * 512 samples).  Bit-identical audio. */
int ria_gpu_mcdpsk_modulate_batch(ria_gpu_handle h, const ria_mcdpsk_config* cfg, const uint8_t* data_dev, int n_bytes, int n_frames,
                                  float* out_dev, int64_t out_stride, void* stream);
/* fec::ChaseCache::store arithmetic for n_cw codeword slots of 648 LLRs (chase_cache.cpp:27-88): count 0 ->
 * copy, else add; skipped when decoded_dev[cw] != 0 or count >= 4.  stored_out_dev (nullable) gets 1/0. */
int ria_gpu_chase_combine_batch(ria_gpu_handle h, float* acc_dev, int32_t* count_dev, const uint8_t* decoded_dev,
                                const float* soft_dev, int n_cw, uint8_t* stored_out_dev, void* stream);

/* ---- link adaptation ladder (host scalars; src/protocol/waveform_selection.hpp:49-104,112-222,250-314) -----
 * waveform: protocol::WaveformMode value (4 MC_DPSK, 5 OFDM_CHIRP); spreading 1/2/4. */
typedef struct ria_link_recommendation {
    int32_t waveform;
    int32_t modulation;
    int32_t code_rate;
    int32_t spreading;
    int32_t num_carriers;
    float estimated_throughput_bps;   /* 0 from ria_link_data_mode for OFDM (the reference does not report one) */
} ria_link_recommendation;
void ria_link_recommend(float snr_db, float fading_index, ria_link_recommendation* out);          /* recommendWaveformAndRate */
void ria_link_data_mode(float snr_db, int waveform, float fading_index, ria_link_recommendation* out);   /* recommendDataMode */
int ria_link_ofdm_code_rate(float snr_db, float fading_index);                                     /* selectOFDMCodeRate */
int ria_link_cap_initial_rate(float snr_db, float fading_index, int candidate_rate);               /* capInitialOFDMRate */

/* LDPCEncoder::encode for n_cw codewords on the host (src/fec/ldpc_encoder.cpp:193-257): info = n_cw * ceil(k/8)
 * bytes (MSB first), coded_out = n_cw * 81 bytes.  Used to synthesise MC-DPSK frames (the OFDM TX kernel
 * encodes on the device). */
int ria_gpu_ldpc_encode_host(ria_gpu_handle h, const uint8_t* info, int n_cw, uint8_t* coded_out);

/* The same channel with the REFERENCE's random stream: frame f is sim::WattersonChannel(cfg, seed32) with
 * seed32 = (uint32_t)(seed + first_frame + f) (std::mt19937 + std::normal_distribution<float>, five draws per
 * sample), so the output is bit-identical to the reference channel (and to tools/test_waveform_simple.cpp's
 * "channel seed = base + frame" convention).  Any frame length; frame f at samples_dev + f*stride; in place. */
int ria_gpu_channel_exact_batch(ria_gpu_handle h, int kind, float snr_db, uint32_t seed, uint64_t first_frame,
                                float* samples_dev, int64_t stride, int frame_samples, int n_frames, void* stream);

/* The same channel with one mt19937 seed PER FRAME (seeds_dev[f]): Monte-Carlo drivers derive the seed of a trial from
 * (base seed, sweep point, transmission number, global trial index), so that a trial's noise does not depend on how
 * trials are batched, compacted or spread over GPUs.  Frame f = sim::WattersonChannel(cfg, seeds_dev[f]), bit-identical. */
int ria_gpu_channel_exact_seeded_batch(ria_gpu_handle h, int kind, float snr_db, const uint32_t* seeds_dev, float* samples_dev,
                                       int64_t stride, int frame_samples, int n_frames, void* stream);

/* The same channel object with its CFO impairment (hf_channel.hpp:47-51 Config::cfo_hz / random_cfo_max_hz): frame f =
 * sim::WattersonChannel(cfg{cfo_hz = cfo_hz_dev[f] (NULL: 0), random_cfo_max_hz}, seeds_dev[f]).process(frame f).
 * random_cfo_max_hz > 0 replaces the configured offset by the constructor's uniform draw from the frame's own generator
 * (:97-102; it takes the first random word, so every later noise value shifts); the noise / fading pass is followed by
 * applyCFO (:172-174, :182-241: mix to baseband at 1500 Hz, 48-sample running-sum average, rotate, mix back) whenever
 * |offset| > 0.001 Hz and the frame has at least 256 samples.  actual_cfo_out_dev (nullable, n_frames floats) =
 * getActualCFO().  Bit-identical output. */
int ria_gpu_channel_exact_cfo_batch(ria_gpu_handle h, int kind, float snr_db, const uint32_t* seeds_dev, const float* cfo_hz_dev,
                                    float random_cfo_max_hz, float* actual_cfo_out_dev, float* samples_dev, int64_t stride,
                                    int frame_samples, int n_frames, void* stream);

/* The simulator's transmitter frequency offset: SimulatedChannel::applyTxCFO(samples, phase_acc)
 * (tools/cli_simulator.cpp:298-341; BASELINE.json config 4's "+-50 Hz CFO" by analytic-signal rotation) for n_buffers
 * transmissions of n_samples each (buffer b at samples_dev + b*stride, result at out_dev + b*out_stride, not in place):
 * FFT of the next power of two, frequency-domain Hilbert, inverse FFT, rotation by cfo_hz_dev[b] with the wrapped float
 * phase accumulator phase_inout_dev[b] (NULL: starts at 0, not returned), real part.  |cfo| < 0.001 Hz copies the
 * samples and leaves the accumulator alone, as the reference does.  n_samples <= 131072.  Bit-identical output. */
int ria_gpu_tx_cfo_batch(ria_gpu_handle h, const float* samples_dev, int64_t stride, int n_samples, int n_buffers,
                         const float* cfo_hz_dev, float* phase_inout_dev, float* out_dev, int64_t out_stride, void* stream);

/* ---- burst interleaver (fec::BurstInterleaver, src/fec/burst_interleaver.cpp:8-78) -----------------
 * A burst of N physical frames carries N logical frames byte-interleaved: physical[(N*b+f)/324][(N*b+f)%324] =
 * logical[f][b].  deinterleave works on the soft bits (8 per byte) of n_groups bursts of N frames each:
 * frame j of group g at llr_dev + (g*N + j)*llr_stride (first 2592 used), same layout out.  interleave is the
 * TX side on coded bytes (324 per frame).  N < 2 copies. */
int ria_gpu_burst_deinterleave_batch(ria_gpu_handle h, const float* physical_llr_dev, int llr_stride, int burst_frames,
                                     int n_groups, float* logical_llr_out_dev, void* stream);
int ria_gpu_burst_interleave_batch(ria_gpu_handle h, const uint8_t* logical_bytes_dev, int burst_frames, int n_groups,
                                   uint8_t* physical_bytes_out_dev, void* stream);

/* ---- debug / test hooks ----------------------------------------------------------------------- */
/* op: 0 sinf 1 cosf 2 logf 3 atan2f(a,b) 4 hypotf(a,b) 5 a/b 6 sqrtf(a); evaluates the device
 * math the kernels use on n arguments (tests compare against the host libm). */
int ria_gpu_debug_math(ria_gpu_handle h, int op, const float* a_dev, const float* b_dev, int n,
                       float* out_dev, void* stream);
/* 1 if, in the handle's last decode calls, a persistent work-queue wave left its loop through the iteration bound
 * instead of the queue's end (a broken queue loop terminates and is reported, it does not hang the GPU); 0 if not;
 * negative = error.  Synchronises the device. */
int ria_gpu_debug_queue_fault(ria_gpu_handle h);

#ifdef __cplusplus
}
#endif
#endif /* RIA_GPU_H */
