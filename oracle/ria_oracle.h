/* oracle/ria_oracle.h — CPU restatement of the RIA RX hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load this library; the product (ria_amd/, libria_gpu.so) never does.
 *
 * Every function restates a reference routine in plain C (cited file:line, paths relative to the
 * reference repository root).  It is pinned against the compiled, unmodified reference
 * (oracle/_ref/libria_ref.so, built by oracle/Makefile in the build container) by
 * oracle/check_against_ref.py and by the golden vectors committed under tests/golden/.
 */
#ifndef RIA_ORACLE_H
#define RIA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum values follow include/ultra/types.hpp:28-39 (Modulation) and :91-100 (CodeRate) */
enum { RO_DBPSK = 0, RO_BPSK = 1, RO_DQPSK = 2, RO_QPSK = 3, RO_D8PSK = 4, RO_QAM8 = 5,
       RO_QAM16 = 6, RO_QAM32 = 7, RO_QAM64 = 8, RO_QAM256 = 10 };
enum { RO_R1_4 = 0, RO_R1_3 = 1, RO_R1_2 = 2, RO_R2_3 = 3, RO_R3_4 = 4, RO_R5_6 = 5, RO_R7_8 = 6 };

#define RO_FFT 1024
#define RO_CP 128
#define RO_SYM (RO_FFT + RO_CP)
#define RO_NCAR 59
#define RO_CW_BITS 648
#define RO_FRAME_BITS 2592
#define RO_MAX_EDGES 4096

typedef struct ro_geom {
    int mod, rate;
    int pilot_spacing, n_pilot, n_data;
    int bits_per_carrier, bits_per_symbol; /* n_data * bits_per_carrier */
    int n_data_symbols;                    /* ceil(2592 / bits_per_symbol) */
    int frame_samples;                     /* (2 + n_data_symbols) * 1152 */
    int n_llr;                             /* n_data_symbols * bits_per_symbol */
    int info_bits, bytes_per_cw, max_iter; /* LDPC k, k/8, recommended iterations */
    int all_idx[RO_NCAR];                  /* logical carrier -> FFT bin */
    int is_pilot[RO_NCAR];
    int data_idx[RO_NCAR], pilot_idx[RO_NCAR]; /* FFT bins */
    int data_logical[RO_NCAR], pilot_logical[RO_NCAR];
    float sync_re[RO_NCAR], sync_im[RO_NCAR];  /* ZC-59 LTS sequence */
    float pilot_seq[RO_NCAR];                  /* +-1 */
    int interp_lo[RO_NCAR], interp_hi[RO_NCAR]; /* per data carrier: pilot ordinal or -1 */
    float interp_alpha[RO_NCAR];
} ro_geom;

int ro_geom_init(ro_geom* g, int mod, int rate);

/* ---- RNG (std::mt19937 + libstdc++ std::normal_distribution<float>) */
typedef struct ro_mt { uint32_t s[624]; int idx; } ro_mt;
void ro_mt_seed(ro_mt* m, uint32_t seed);
uint32_t ro_mt_next(ro_mt* m);
typedef struct ro_normal { float saved; int has_saved; } ro_normal;
float ro_normal_draw(ro_normal* n, ro_mt* m, float mean, float stddev);

/* ---- LDPC (src/fec/ldpc_encoder.cpp, src/fec/ldpc_decoder.cpp) */
typedef struct ro_ldpc {
    int rate, k, m, n, n_edges;
    int row_ptr[RO_CW_BITS + 1];
    int edge_var[RO_MAX_EDGES]; /* row-major edge list: variable index */
} ro_ldpc;
int ro_ldpc_build(ro_ldpc* c, int rate);
int ro_ldpc_encode(const ro_ldpc* c, const uint8_t* info, int n_info_bytes, uint8_t* coded81);
/* returns 1 on success; out gets ceil(k/8) bytes; *iters as LDPCDecoder::lastIterations() */
int ro_ldpc_decode(const ro_ldpc* c, const float* llr, int n_llr, int max_iter, float factor,
                   uint8_t* out, int* iters);

/* ---- interleavers */
int ro_channel_interleaver_step(int bits_per_symbol, int total);
void ro_rx_gather_table(int bits_per_symbol, int use_channel, int* table /*[4*648]*/);

/* ---- frame build / TX (src/protocol/frame_v2.cpp, src/ofdm/modulator.cpp) */
uint16_t ro_crc16(const uint8_t* d, int n);
int ro_make_frame(const uint8_t* payload, int payload_len, int seq, int rate, uint8_t* info_out);
int ro_encode_fixed_frame(const uint8_t* info, int n_info, int rate, int ch_interleave, int bps,
                          uint8_t* coded324);
int ro_modulate(const ro_geom* g, const uint8_t* coded, int n_coded, float* samples, int max_samples);

/* ---- channel (src/sim/hf_channel.hpp) kind: 0 awgn 1 good 2 moderate 3 poor 4 flutter */
int ro_channel(int kind, float snr_db, uint32_t seed, const float* in, int n, float* out);
/* the same with Config::cfo_hz / random_cfo_max_hz: the constructor's CFO draw (hf_channel.hpp:97-102) and applyCFO
 * (:182-241) after the noise; actual_cfo_out (nullable) = getActualCFO() */
int ro_channel_cfo(int kind, float snr_db, uint32_t seed, float cfo_hz, float random_cfo_max_hz, const float* in, int n, float* out,
                   float* actual_cfo_out);

/* ---- RX demod (src/ofdm/demodulator.cpp, channel_equalizer.cpp, soft_demap.hpp) */
typedef struct ro_rx_aux {
    float snr_db, cfo_hz, fading_index, noise_variance, lts_phase_slope, snr_linear,
          corr_phase, snr_symbol_count;
    float h[2 * RO_NCAR];
} ro_rx_aux;
int ro_rx_process(const ro_geom* g, const float* samples, int n, float cfo_hz, long long abs_pos,
                  float* llr_out, int max_llr, ro_rx_aux* aux);

/* the same with OFDMChirpWaveform's one-shot burst marker: flags bit0 = the first LTS symbol was negated on air
 * (ofdm_chirp_waveform.cpp:421-440) */
int ro_rx_process_flags(const ro_geom* g, const float* samples, int n, float cfo_hz, long long abs_pos, int flags,
                        float* llr_out, int max_llr, ro_rx_aux* aux);
/* fec::BurstInterleaver (src/fec/burst_interleaver.cpp:8-78) */
void ro_burst_interleave(int n_frames, const uint8_t* logical, uint8_t* physical);
void ro_burst_deinterleave(int n_frames, const float* physical, int stride, float* logical);

/* ---- decode (src/protocol/frame_v2.cpp:1335-1883) */
/* flags bit0: run retry cascade phase 0; bit1: phases 1-6; bit2: CRC false-positive recovery */
int ro_decode_fixed_frame(const float* llr, int n, int rate, int ch_deint, int bps, int flags,
                          uint8_t* data_out, uint8_t* ok_out, int* iters_out, int* attempts_out);

#ifdef __cplusplus
}
#endif
/* ---- acquisition correlators (ria_oracle_sync.c) ---- */
/* sync::ZCSync, src/sync/zc_sync.hpp: preamble synthesis and detection.
 * out7 = {detected, frame_type, start_sample, correlation, cfo_hz, snr_estimate, root_detected} */
int ro_zc_preamble_samples(void);
int ro_zc_generate(int root, float* out, int max_n);
int ro_zc_detect(const float* rx, int n, float threshold, int root_mask, float known_cfo_hz, float* out7);
/* sync::ChirpSync, src/sync/chirp_sync.hpp: dual-chirp synthesis (57 600 samples) and detectDualChirp.
 * out6 = {success, up_chirp_start, down_chirp_start, cfo_hz, up_correlation, down_correlation} */
int ro_chirp_generate(float* out, int max_n);
int ro_chirp_detect(const float* s, int n, float threshold, float* out6);
/* MultiCarrierDPSK (src/psk/multi_carrier_dpsk.hpp): training + reference + data audio, and the demodulator chain
 * of processGotChirp after an external chirp detection.  aux4 = {cfo, fading, frequency fading, temporal fading}. */
int ro_mcdpsk_modulate(int nc, int bps, int spreading, const uint8_t* data, int n_bytes, float* out, int max_n);
int ro_mcdpsk_demod(int nc, int bps, int spreading, const float* samples, int n, float cfo_hz, float phase0,
                    float* llr_out, int max_llr, float* aux4);
/* OFDMChirpWaveform::detectDataSync (LTS light sync); out4 = {detected, start_sample, correlation, burst_interleaved} */
int ro_detect_data_sync(const float* x, int n, float known_cfo_hz, float threshold, float* out4);
/* SimulatedChannel::applyTxCFO (tools/cli_simulator.cpp:298-341): the simulator's transmitter frequency offset - FFT of
 * the next power of two, frequency-domain Hilbert, inverse FFT, rotation by a wrapped float phase (in/out), real part */
int ro_apply_tx_cfo(const float* in, int n, float cfo_hz, float* phase_inout, float* out);
/* ---- helper arithmetic of the reference's own test programs (their scenarios are composed in pyoracle.py)
 * ro_tool_add_noise: addNoise of tools/test_zc_sync.cpp:22-39 == tools/test_spreading.cpp:15-30 (AWGN at an SNR against the
 *   mean power of ALL samples; a fresh std::normal_distribution<float>(0, sigma) per call on the caller's generator)
 * ro_tool_apply_cfo: applyCFO of tools/test_zc_sync.cpp:43-63 (HilbertTransform(127), src/dsp/filters.cpp:266-317, rotation
 *   by a float phase wrapped into (-pi, pi], real part)
 * ro_tool_chase_reception: generateNoisyCodeword + addNoise of tools/test_chase_cache.cpp:21-62 (coded bits as +-4, a
 *   reception is 2 (sign + n) snr with n ~ N(0, 1 / snr)) */
void ro_tool_add_noise(float* x, int n, float snr_db, ro_mt* rng);
void ro_tool_apply_cfo(float* x, int n, float cfo_hz, float sample_rate);
void ro_tool_chase_reception(const uint8_t* coded81, float snr_db, ro_mt* rng, float* llr648);
/* fec::ChaseCache::store arithmetic for one codeword slot (src/fec/chase_cache.cpp:27-88) */
int ro_chase_store(float* existing, int* combine_count, int decoded, const float* soft);

#endif
/* Schmidl-Cox acquisition: OFDMDemodulator::searchForSync (src/ofdm/demodulator.cpp:1450-1542, ofdm_sync.cpp).
 * out3 = {found, first-LTS position, cfo_hz}; noise_floor in/out = Impl::noise_floor_energy (may be NULL = 0). */
#ifdef __cplusplus
extern "C" {
#endif
void ro_cox_lts_template(const ro_geom* g, float* tI, float* tQ);
int ro_cox_search(const ro_geom* g, const float* x, int n, float threshold, float* noise_floor, float* out3);
#ifdef __cplusplus
}
#endif
