// oracle/ref_shim.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin C-ABI wrapper around the *unmodified* reference classes, compiled together
// with the reference's own source files where they lie under /root/reference by
// oracle/Makefile into oracle/_ref/libria_ref.so (git-ignored).  It exists only in
// the build container: it pins the CPU restatement in oracle/ria_oracle.c and
// generates the golden vectors under tests/golden/ (oracle/gen_golden.py).
//
// Nothing here is copied from the reference; it only *calls* it:
//   OFDMChirpWaveform   src/waveform/ofdm_chirp_waveform.cpp:391-468 (process)
//   v2::encodeFixedFrame/decodeFixedFrame   src/protocol/frame_v2.cpp:1285-1883
//   LDPCEncoder/LDPCDecoder   src/fec/ldpc_encoder.cpp, src/fec/ldpc_decoder.cpp
//   sim::WattersonChannel   src/sim/hf_channel.hpp:35-303
//   sync::ZCSync / sync::ChirpSync   src/sync/zc_sync.hpp, src/sync/chirp_sync.hpp

#include <algorithm>
#include <atomic>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <random>
#include <span>
#include <string>
#include <vector>

// The demodulator keeps its estimator state in a private pimpl; the golden
// fixtures need that state (H, noise variance, phase slope) as stage taps.
#define private public
#include "ultra/ofdm.hpp"
#undef private
#include "ofdm/demodulator_impl.hpp"

#include "ultra/fec.hpp"
#include "ultra/logging.hpp"
#include "ultra/types.hpp"
#include "fec/frame_interleaver.hpp"
#include "fec/burst_interleaver.hpp"
#include "fec/chase_cache.hpp"
#include "fec/ldpc_codec.hpp"
#include "protocol/frame_v2.hpp"
#include "sim/hf_channel.hpp"
#include "protocol/waveform_selection.hpp"
#include "sync/chirp_sync.hpp"
#include "sync/zc_sync.hpp"
#include "psk/multi_carrier_dpsk.hpp"
#define private public
#include "waveform/ofdm_chirp_waveform.hpp"
#include "waveform/ofdm_cox_waveform.hpp"
#undef private
#include "waveform/mc_dpsk_waveform.hpp"

using namespace ultra;

namespace {

ModemConfig named_config(int mod, int rate) {
    // SURVEY.md Appendix C / tools/test_waveform_simple.cpp:109-121
    ModemConfig cfg;
    cfg.sample_rate = 48000;
    cfg.center_freq = 1500;
    cfg.fft_size = 1024;
    cfg.num_carriers = 59;
    cfg.cp_mode = CyclicPrefixMode::LONG;
    cfg.use_pilots = true;
    cfg.modulation = static_cast<Modulation>(mod);
    cfg.code_rate = static_cast<CodeRate>(rate);
    return cfg;
}

}  // namespace

extern "C" {

void ref_quiet(void) { ultra::g_log_level = LogLevel::NONE; }

// ---------------------------------------------------------------- TX
// Builds one connected-mode data frame: light preamble (2 LTS) + modulated
// fixed 4-CW frame.  Returns sample count (or -needed if max_samples too small).
int ref_tx_frame(int mod, int rate, const uint8_t* payload, int payload_len, int seq,
                 float* samples_out, int max_samples,
                 uint8_t* frame_info_out, int max_info,   // 4*bytes_per_cw, zero padded
                 uint8_t* coded_out, int max_coded,       // 324 interleaved coded bytes
                 int* bits_per_symbol_out) {
    ref_quiet();
    ModemConfig cfg = named_config(mod, rate);
    OFDMChirpWaveform tx(cfg);
    tx.configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    CodeRate cr = static_cast<CodeRate>(rate);

    Bytes pl(payload, payload + payload_len);
    auto frame = protocol::v2::makeFixedDataFrame("TEST", "RX", static_cast<uint16_t>(seq), pl, cr);
    Bytes fd = frame.serialize();

    int pilots = (cfg.num_carriers + tx.config_.pilot_spacing - 1) / tx.config_.pilot_spacing;
    int data_carriers = cfg.num_carriers - pilots;
    size_t bps = data_carriers * getBitsPerSymbol(static_cast<Modulation>(mod));
    if (bits_per_symbol_out) *bits_per_symbol_out = static_cast<int>(bps);

    Bytes enc = protocol::v2::encodeFixedFrame(fd, cr, true, bps);
    Samples pre = tx.generateDataPreamble();
    Samples dat = tx.modulate(enc);

    size_t info_bytes = 4 * protocol::v2::getBytesPerCodeword(cr);
    if (frame_info_out) {
        Bytes padded = fd;
        padded.resize(info_bytes, 0);
        std::memcpy(frame_info_out, padded.data(), std::min<size_t>(info_bytes, max_info));
    }
    if (coded_out) std::memcpy(coded_out, enc.data(), std::min<size_t>(enc.size(), max_coded));

    int n = static_cast<int>(pre.size() + dat.size());
    if (n > max_samples) return -n;
    std::memcpy(samples_out, pre.data(), pre.size() * sizeof(float));
    std::memcpy(samples_out + pre.size(), dat.data(), dat.size() * sizeof(float));
    return n;
}

// Raw bytes → fixed-frame encode (no frame header/CRC added): LDPC x4 + interleave.
int ref_encode_fixed_frame(const uint8_t* info, int n_info, int rate, int ch_interleave, int bps,
                           uint8_t* coded_out, int max_coded) {
    ref_quiet();
    Bytes fd(info, info + n_info);
    Bytes enc = protocol::v2::encodeFixedFrame(fd, static_cast<CodeRate>(rate), ch_interleave != 0, bps);
    std::memcpy(coded_out, enc.data(), std::min<size_t>(enc.size(), max_coded));
    return static_cast<int>(enc.size());
}

// Coded bytes → audio: light preamble + modulate (no LDPC).
int ref_modulate(int mod, int rate, const uint8_t* coded, int n_coded, float* samples_out, int max_samples) {
    ref_quiet();
    ModemConfig cfg = named_config(mod, rate);
    OFDMChirpWaveform tx(cfg);
    tx.configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    Bytes enc(coded, coded + n_coded);
    Samples pre = tx.generateDataPreamble();
    Samples dat = tx.modulate(enc);
    int n = static_cast<int>(pre.size() + dat.size());
    if (n > max_samples) return -n;
    std::memcpy(samples_out, pre.data(), pre.size() * sizeof(float));
    std::memcpy(samples_out + pre.size(), dat.data(), dat.size() * sizeof(float));
    return n;
}

// ---------------------------------------------------------------- channel
// kind: 0 awgn, 1 good, 2 moderate, 3 poor, 4 flutter  (hf_channel.hpp:411-488)
int ref_channel(int kind, float snr_db, uint32_t seed, const float* in, int n, float* out) {
    ref_quiet();
    sim::WattersonChannel::Config c;
    switch (kind) {
        case 0: c = sim::itu_r_f1487::awgn(snr_db); break;
        case 1: c = sim::itu_r_f1487::good(snr_db); break;
        case 2: c = sim::itu_r_f1487::moderate(snr_db); break;
        case 3: c = sim::itu_r_f1487::poor(snr_db); break;
        default: c = sim::itu_r_f1487::flutter(snr_db); break;
    }
    sim::WattersonChannel ch(c, seed);
    Samples o = ch.process(SampleSpan(in, n));
    std::memcpy(out, o.data(), n * sizeof(float));
    return n;
}

// The same channel with the reference's CFO impairment: Config::cfo_hz / random_cfo_max_hz (hf_channel.hpp:47-51),
// the CFO draw of the constructor (:97-102, taken from rng_ BEFORE any noise is drawn) and applyCFO (:182-241: mix to
// baseband at 1500 Hz, 48-tap moving average, rotate, mix back).  actual_cfo_out = getActualCFO().
int ref_channel_cfo(int kind, float snr_db, uint32_t seed, float cfo_hz, float random_cfo_max_hz, const float* in, int n, float* out,
                    float* actual_cfo_out) {
    ref_quiet();
    sim::WattersonChannel::Config c;
    switch (kind) {
        case 0: c = sim::itu_r_f1487::awgn(snr_db); break;
        case 1: c = sim::itu_r_f1487::good(snr_db); break;
        case 2: c = sim::itu_r_f1487::moderate(snr_db); break;
        case 3: c = sim::itu_r_f1487::poor(snr_db); break;
        default: c = sim::itu_r_f1487::flutter(snr_db); break;
    }
    c.cfo_hz = cfo_hz;
    c.random_cfo_max_hz = random_cfo_max_hz;
    sim::WattersonChannel ch(c, seed);
    Samples o = ch.process(SampleSpan(in, n));
    std::memcpy(out, o.data(), n * sizeof(float));
    if (actual_cfo_out) *actual_cfo_out = ch.getActualCFO();
    return n;
}

// ---------------------------------------------------------------- RX, one kept object per caller thread
// The CPU throughput baseline ("one instance per std::thread"): ONE OFDMChirpWaveform configured once,
// reset() before every frame as StreamingDecoder does (streaming_decoder.cpp:723), then setFrequencyOffset -> process ->
// getSoftBits -> v2::decodeFixedFrame.  ref_rx_process above builds and configures a new object per call (two
// initComponents()), which is right for fixtures but inflates a per-frame time.
struct RefRx {
    std::unique_ptr<OFDMChirpWaveform> rx;
    CodeRate rate;
    int bps;
};
void* ref_rx_open(int mod, int rate) {
    ref_quiet();
    auto* h = new RefRx();
    ModemConfig cfg = named_config(mod, rate);
    h->rx = std::make_unique<OFDMChirpWaveform>(cfg);
    h->rx->configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    h->rate = static_cast<CodeRate>(rate);
    const int pilots = (cfg.num_carriers + h->rx->config_.pilot_spacing - 1) / h->rx->config_.pilot_spacing;
    h->bps = (cfg.num_carriers - pilots) * static_cast<int>(getBitsPerSymbol(static_cast<Modulation>(mod)));
    return h;
}
void ref_rx_close(void* hv) { delete static_cast<RefRx*>(hv); }
// returns the number of codewords decoded (0..4), -1 if process() produced no soft bits; llr_out nullable;
// data_out == NULL: demodulate only and return the number of soft bits
int ref_rx_frame(void* hv, const float* samples, int n, float cfo_hz, uint8_t* data_out, uint8_t* ok_out, float* llr_out, int max_llr) {
    auto* h = static_cast<RefRx*>(hv);
    h->rx->reset();
    h->rx->setFrequencyOffset(cfo_hz);
    bool ok = h->rx->process(SampleSpan(samples, n));
    std::vector<float> soft = h->rx->getSoftBits();
    if (llr_out) std::memcpy(llr_out, soft.data(), std::min<size_t>(soft.size(), max_llr) * sizeof(float));
    if (!data_out) return ok ? static_cast<int>(soft.size()) : -1;   // process() + getSoftBits() only
    if (!ok || soft.size() < 2592) return -1;
    auto st = protocol::v2::decodeFixedFrame(soft, h->rate, true, h->bps);
    size_t bpc = protocol::v2::getBytesPerCodeword(h->rate);
    int good = 0;
    for (int cw = 0; cw < 4; ++cw) {
        ok_out[cw] = st.decoded[cw] ? 1 : 0;
        std::memset(data_out + cw * bpc, 0, bpc);
        if (st.decoded[cw] && st.data[cw].size() >= bpc) { std::memcpy(data_out + cw * bpc, st.data[cw].data(), bpc); ++good; }
    }
    return good;
}

// ---------------------------------------------------------------- RX demod
// aux_out[0..7] = {snr_db, cfo_hz_out, fading_index, noise_variance, lts_phase_slope,
//                  estimated_snr_linear, freq_correction_phase, snr_symbol_count}
// h_out: 59 complex (re,im) channel_estimate in logical carrier order after the frame.
int ref_rx_process(int mod, int rate, const float* samples, int n, float cfo_hz,
                   long long abs_pos, int use_abs,
                   float* llr_out, int max_llr, float* aux_out, float* h_out) {
    ref_quiet();
    ModemConfig cfg = named_config(mod, rate);
    OFDMChirpWaveform rx(cfg);
    rx.configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    rx.reset();
    if (use_abs) rx.setAbsoluteTrainingPosition(static_cast<size_t>(abs_pos));
    rx.setFrequencyOffset(cfo_hz);
    bool ok = rx.process(SampleSpan(samples, n));
    std::vector<float> soft = rx.getSoftBits();
    int m = static_cast<int>(soft.size());
    if (llr_out) std::memcpy(llr_out, soft.data(), std::min(m, max_llr) * sizeof(float));
    auto* impl = rx.demodulator_->impl_.get();
    if (aux_out) {
        aux_out[0] = rx.estimatedSNR();
        aux_out[1] = rx.demodulator_->getFrequencyOffset();
        aux_out[2] = impl->last_fading_index;
        aux_out[3] = impl->noise_variance;
        aux_out[4] = impl->lts_phase_slope;
        aux_out[5] = impl->estimated_snr_linear;
        aux_out[6] = impl->freq_correction_phase;
        aux_out[7] = static_cast<float>(impl->snr_symbol_count);
    }
    if (h_out) {
        for (size_t i = 0; i < impl->all_carrier_fft_indices.size(); ++i) {
            Complex h = impl->channel_estimate[impl->all_carrier_fft_indices[i]];
            h_out[2 * i] = h.real();
            h_out[2 * i + 1] = h.imag();
        }
    }
    return ok ? m : -m;
}

// ---------------------------------------------------------------- decode
// data_out: 4*bytes_per_cw, ok_out: 4
int ref_decode_fixed_frame(const float* llr, int n, int rate, int ch_deint, int bps,
                           uint8_t* data_out, uint8_t* ok_out) {
    ref_quiet();
    std::vector<float> soft(llr, llr + n);
    CodeRate cr = static_cast<CodeRate>(rate);
    auto st = protocol::v2::decodeFixedFrame(soft, cr, ch_deint != 0, bps);
    size_t bpc = protocol::v2::getBytesPerCodeword(cr);
    int good = 0;
    for (int cw = 0; cw < 4; ++cw) {
        ok_out[cw] = st.decoded[cw] ? 1 : 0;
        std::memset(data_out + cw * bpc, 0, bpc);
        if (st.decoded[cw] && st.data[cw].size() >= bpc) {
            std::memcpy(data_out + cw * bpc, st.data[cw].data(), bpc);
            ++good;
        }
    }
    return good;
}

int ref_ldpc_decode(int rate, const float* llr, int n, int max_iter, float factor,
                    uint8_t* out, int max_out, int* iters) {
    LDPCDecoder dec(static_cast<CodeRate>(rate));
    dec.setMaxIterations(max_iter);
    dec.setMinSumFactor(factor);
    Bytes b = dec.decodeSoft(std::span<const float>(llr, n));
    std::memcpy(out, b.data(), std::min<size_t>(b.size(), max_out));
    if (iters) *iters = dec.lastIterations();
    return dec.lastDecodeSuccess() ? static_cast<int>(b.size()) : -static_cast<int>(b.size());
}

// robustDecodeSingleCW is a file-static function of streaming_decoder.cpp (:1028-1058); these are its statements on
// the reference's own LDPCDecoder: recommended iterations, factor 0.9375, then 0.875 / 0.75 / 0.625 / 0.5.
int ref_robust_decode(int rate, const float* llr, int n, uint8_t* out, int max_out, int* iters, int* tries) {
    CodeRate cr = static_cast<CodeRate>(rate);
    LDPCDecoder decoder(cr);
    decoder.setMaxIterations(fec::LDPCCodec::getRecommendedIterations(cr));
    decoder.setMinSumFactor(0.9375f);
    auto decoded = decoder.decodeSoft(std::span<const float>(llr, n));
    bool ok = decoder.lastDecodeSuccess();
    int t = 1;
    if (!ok) {
        static constexpr float factors[] = {0.875f, 0.75f, 0.625f, 0.5f};
        for (int retry = 0; retry < 4 && !ok; retry++) {
            decoder.setMinSumFactor(factors[retry]);
            decoded = decoder.decodeSoft(std::span<const float>(llr, n));
            ok = decoder.lastDecodeSuccess();
            ++t;
        }
    }
    std::memcpy(out, decoded.data(), std::min<size_t>(decoded.size(), max_out));
    if (iters) *iters = decoder.lastIterations();
    if (tries) *tries = t;
    return ok ? static_cast<int>(decoded.size()) : -static_cast<int>(decoded.size());
}

int ref_ldpc_encode(int rate, const uint8_t* in, int n, uint8_t* out, int max_out) {
    LDPCEncoder enc(static_cast<CodeRate>(rate));
    Bytes b = enc.encode(ByteSpan(in, n));
    std::memcpy(out, b.data(), std::min<size_t>(b.size(), max_out));
    return static_cast<int>(b.size());
}

int ref_recommended_iterations(int rate) {
    return fec::LDPCCodec::getRecommendedIterations(static_cast<CodeRate>(rate));
}

// perm_out[i] = ChannelInterleaver permutation_[i] via a one-hot probe.
int ref_channel_interleaver_perm(int bps, int total, int* inv_out) {
    ChannelInterleaver il(bps, total);
    std::vector<float> probe(total);
    for (int i = 0; i < total; ++i) probe[i] = static_cast<float>(i);
    auto d = il.deinterleave(std::span<const float>(probe));
    for (int i = 0; i < total; ++i) inv_out[i] = static_cast<int>(d[i]);  // dec_in[i] = rx[inv_out[i]]
    return total;
}

// ---------------------------------------------------------------- sync
// LTS light sync (ofdm_chirp_waveform.cpp:207-384)
int ref_detect_data_sync(int mod, int rate, const float* samples, int n, float known_cfo, float thr,
                         int* start_out, float* corr_out, int* burst_out) {
    ref_quiet();
    ModemConfig cfg = named_config(mod, rate);
    OFDMChirpWaveform rx(cfg);
    rx.configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    SyncResult r;
    bool ok = rx.detectDataSync(SampleSpan(samples, n), r, known_cfo, thr);
    *start_out = r.start_sample;
    *corr_out = r.correlation;
    if (burst_out) *burst_out = rx.wasBurstInterleaved() ? 1 : 0;
    return ok ? 1 : 0;
}

// Schmidl-Cox acquisition (demodulator.cpp:1450-1542).  noise_inout = Impl::noise_floor_energy before / after.
int ref_cox_search(int mod, int rate, const float* samples, int n, float thr, float* noise_inout, float* out3) {
    ref_quiet();
    OFDMNvisWaveform rx(named_config(mod, rate));
    rx.configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));   // pilot spacing as the waveform sets it
    if (noise_inout) rx.demodulator_->impl_->noise_floor_energy = *noise_inout;
    SyncResult r;
    bool ok = rx.detectSync(SampleSpan(samples, n), r, thr);                    // -> OFDMDemodulator::searchForSync
    if (noise_inout) *noise_inout = rx.demodulator_->impl_->noise_floor_energy;
    out3[0] = ok ? 1.0f : 0.0f; out3[1] = ok ? static_cast<float>(r.start_sample) : 0.0f; out3[2] = ok ? r.cfo_hz : 0.0f;
    return ok ? 1 : 0;
}
// Full Schmidl-Cox preamble (guard + 4 STS + 2 LTS, modulator.cpp:479-532) followed by one modulated frame of coded bytes.
int ref_cox_transmit(int mod, int rate, const uint8_t* coded, int n_coded, float* out, int max_n) {
    ref_quiet();
    OFDMNvisWaveform tx(named_config(mod, rate));
    tx.configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    Samples pre = tx.generatePreamble();
    Samples dat;
    if (n_coded > 0) dat = tx.modulate(Bytes(coded, coded + n_coded));
    int n = static_cast<int>(pre.size() + dat.size());
    if (n > max_n) return -n;
    std::memcpy(out, pre.data(), pre.size() * sizeof(float));
    if (!dat.empty()) std::memcpy(out + pre.size(), dat.data(), dat.size() * sizeof(float));
    return n;
}
int ref_cox_lts_template(int mod, int rate, float* tI, float* tQ) {
    ref_quiet();
    OFDMNvisWaveform rx(named_config(mod, rate));
    rx.configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    auto& im = *rx.demodulator_->impl_;
    int n = static_cast<int>(im.lts_passband_I.size());
    std::memcpy(tI, im.lts_passband_I.data(), n * sizeof(float));
    std::memcpy(tQ, im.lts_passband_Q.data(), n * sizeof(float));
    return n;
}
// OFDM-COX end to end on one capture buffer: detectSync (Schmidl-Cox) -> process(samples from the LTS on) -> soft bits
// (ofdm_cox_waveform.cpp:125-214).  out3 = {found, first-LTS position, cfo}; returns the soft-bit count (0 if not found).
int ref_cox_rx(int mod, int rate, const float* samples, int n, float thr, float* out3, float* llr_out, int max_llr, float* aux2) {
    ref_quiet();
    OFDMNvisWaveform rx(named_config(mod, rate));
    rx.configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    SyncResult r;
    bool ok = rx.detectSync(SampleSpan(samples, n), r, thr);
    out3[0] = ok ? 1.0f : 0.0f; out3[1] = ok ? static_cast<float>(r.start_sample) : 0.0f; out3[2] = ok ? r.cfo_hz : 0.0f;
    if (!ok) return 0;
    bool ready = rx.process(SampleSpan(samples + r.start_sample, n - r.start_sample));
    std::vector<float> soft = rx.getSoftBits();
    int m = static_cast<int>(soft.size());
    std::memcpy(llr_out, soft.data(), std::min(m, max_llr) * sizeof(float));
    if (aux2) { aux2[0] = rx.estimatedSNR(); aux2[1] = rx.estimatedCFO(); }
    return ready ? m : -m;
}
// Dual chirp (chirp_sync.hpp:352-512); out = {success, up_start, down_start, cfo_hz, up_corr, down_corr}
int ref_chirp_detect(const float* samples, int n, float thr, float* out6) {
    ref_quiet();
    ModemConfig cfg = named_config(static_cast<int>(Modulation::QAM16), static_cast<int>(CodeRate::R1_2));
    OFDMChirpWaveform w(cfg);
    auto r = w.chirp_sync_->detectDualChirp(SampleSpan(samples, n), thr);
    out6[0] = r.success ? 1.f : 0.f;
    out6[1] = static_cast<float>(r.up_chirp_start);
    out6[2] = static_cast<float>(r.down_chirp_start);
    out6[3] = r.cfo_hz;
    out6[4] = r.up_correlation;
    out6[5] = r.down_correlation;
    return r.success ? 1 : 0;
}

int ref_chirp_generate(float* out, int max_n) {
    ref_quiet();
    ModemConfig cfg = named_config(static_cast<int>(Modulation::QAM16), static_cast<int>(CodeRate::R1_2));
    OFDMChirpWaveform w(cfg);
    Samples s = w.chirp_sync_->generate();
    int n = static_cast<int>(s.size());
    if (n > max_n) return -n;
    std::memcpy(out, s.data(), n * sizeof(float));
    return n;
}

// sync::ZCSync (src/sync/zc_sync.hpp:133-190 generatePreambleForRoot, :192-391 detect)
int ref_zc_generate(int root, float* out, int max_n) {
    ref_quiet();
    sync::ZCSync z;
    Samples s = z.generatePreambleForRoot(root);
    int n = static_cast<int>(s.size());
    if (n > max_n) return -n;
    std::memcpy(out, s.data(), n * sizeof(float));
    return n;
}
int ref_zc_detect(const float* samples, int n, float thr, int root_mask, float known_cfo, float* out7) {
    ref_quiet();
    sync::ZCSync z;
    auto r = z.detect(SampleSpan(samples, n), thr, false, static_cast<uint8_t>(root_mask), known_cfo);
    out7[0] = r.detected ? 1.f : 0.f;
    out7[1] = static_cast<float>(static_cast<int>(r.frame_type));
    out7[2] = static_cast<float>(r.start_sample);
    out7[3] = r.correlation;
    out7[4] = r.cfo_hz;
    out7[5] = r.snr_estimate;
    out7[6] = static_cast<float>(r.root_detected);
    return r.detected ? 1 : 0;
}

// MultiCarrierDPSK (src/psk/multi_carrier_dpsk.hpp): modulator output training+ref+data, and the
// demodulator driven exactly as MCDPSKWaveform::process drives it after an external chirp detection
// (mc_dpsk_waveform.cpp:322-332 -> processGotChirp, multi_carrier_dpsk.hpp:797-896).
static MultiCarrierDPSKConfig mc_config(int carriers, int bps, int spreading) {
    MultiCarrierDPSKConfig c;
    c.num_carriers = carriers;
    c.bits_per_symbol = bps;
    c.spreading_mode = spreading == 4 ? SpreadingMode::TIME_4X : spreading == 2 ? SpreadingMode::TIME_2X : SpreadingMode::NONE;
    return c;
}
int ref_mcdpsk_modulate(int carriers, int bps, int spreading, const uint8_t* data, int n_bytes, float* out, int max_n) {
    ref_quiet();
    MultiCarrierDPSKModulator m(mc_config(carriers, bps, spreading));
    Samples tr = m.generateTrainingSequence();
    Samples rf = m.generateReferenceSymbol();
    Samples dt = m.modulate(Bytes(data, data + n_bytes));
    int n = static_cast<int>(tr.size() + rf.size() + dt.size());
    if (n > max_n) return -n;
    std::memcpy(out, tr.data(), tr.size() * sizeof(float));
    std::memcpy(out + tr.size(), rf.data(), rf.size() * sizeof(float));
    std::memcpy(out + tr.size() + rf.size(), dt.data(), dt.size() * sizeof(float));
    return n;
}
// aux4 = {estimated cfo after the frame, fading index, frequency fading index, temporal fading index}
int ref_mcdpsk_demod(int carriers, int bps, int spreading, const float* samples, int n, float cfo_hz, float phase0,
                     float* llr_out, int max_llr, float* aux4) {
    ref_quiet();
    MultiCarrierDPSKDemodulator d(mc_config(carriers, bps, spreading));
    d.setChirpDetected(cfo_hz);
    d.setCFOWithPhase(cfo_hz, phase0);
    bool ready = d.process(SampleSpan(samples, n));
    if (!ready) return -1;
    float fading = d.getFadingIndex(), ffi = d.getFrequencyFadingIndex(), tfi = d.getTemporalFadingIndex(), cfo = d.getEstimatedCFO();
    std::vector<float> sb = d.getSoftBits();
    int m = static_cast<int>(sb.size());
    if (m > max_llr) return -m;
    std::memcpy(llr_out, sb.data(), sb.size() * sizeof(float));
    aux4[0] = cfo; aux4[1] = fading; aux4[2] = ffi; aux4[3] = tfi;
    return m;
}

// protocol::recommendWaveformAndRate / recommendDataMode / selectOFDMCodeRate / capInitialOFDMRate
// (src/protocol/waveform_selection.hpp)
void ref_link_recommend(float snr, float fading, float* out6) {
    auto r = protocol::recommendWaveformAndRate(snr, fading);
    out6[0] = static_cast<float>(static_cast<int>(r.waveform)); out6[1] = static_cast<float>(static_cast<int>(r.modulation));
    out6[2] = static_cast<float>(static_cast<int>(r.rate));
    out6[3] = r.spreading == SpreadingMode::TIME_4X ? 4.f : r.spreading == SpreadingMode::TIME_2X ? 2.f : 1.f;
    out6[4] = static_cast<float>(r.num_carriers); out6[5] = r.estimated_throughput_bps;
}
void ref_link_data_mode(float snr, int waveform, float fading, float* out4) {
    Modulation mod = Modulation::DQPSK; CodeRate rate = CodeRate::R1_4; int nc = 10; SpreadingMode sp = SpreadingMode::NONE;
    protocol::recommendDataMode(snr, static_cast<protocol::WaveformMode>(waveform), mod, rate, fading, &nc, &sp);
    out4[0] = static_cast<float>(static_cast<int>(mod)); out4[1] = static_cast<float>(static_cast<int>(rate));
    out4[2] = sp == SpreadingMode::TIME_4X ? 4.f : sp == SpreadingMode::TIME_2X ? 2.f : 1.f; out4[3] = static_cast<float>(nc);
}
int ref_link_ofdm_code_rate(float snr, float fading) { return static_cast<int>(protocol::selectOFDMCodeRate(snr, fading)); }
int ref_link_cap_initial_rate(float snr, float fading, int cand) {
    return static_cast<int>(protocol::capInitialOFDMRate(snr, fading, static_cast<CodeRate>(cand)));
}

// fec::BurstInterleaver (src/fec/burst_interleaver.cpp:8-78): N physical frames of 324 bytes / 2592 soft bits
int ref_burst_interleave(int n_frames, const uint8_t* logical, uint8_t* physical) {
    std::vector<std::vector<uint8_t>> in(n_frames, std::vector<uint8_t>(324));
    for (int f = 0; f < n_frames; ++f) std::memcpy(in[f].data(), logical + f * 324, 324);
    auto out = fec::BurstInterleaver::interleave(in);
    for (int f = 0; f < n_frames; ++f) std::memcpy(physical + f * 324, out[f].data(), 324);
    return 0;
}
int ref_burst_deinterleave(int n_frames, const float* physical, float* logical) {
    std::vector<std::vector<float>> in(n_frames, std::vector<float>(2592));
    for (int f = 0; f < n_frames; ++f) std::memcpy(in[f].data(), physical + f * 2592, 2592 * sizeof(float));
    auto out = fec::BurstInterleaver::deinterleave(in);
    for (int f = 0; f < n_frames; ++f) std::memcpy(logical + f * 2592, out[f].data(), 2592 * sizeof(float));
    return 0;
}

// ---------------------------------------------------------------- the same two calls on the OFDM-COX waveform object
// OFDMChirpWaveform::configure maps QAM256 (and QAM8) to DQPSK (ofdm_chirp_waveform.cpp:82-89); OFDMNvisWaveform::configure
// takes any modulation (ofdm_cox_waveform.cpp:71-91) and its process() drives the same OFDMDemodulator::processPresynced
// (:164-214).  TX = the waveform's own OFDMModulator: 2 training symbols + modulate, as generateDataPreamble does.
int ref_tx_frame_nvis(int mod, int rate, const uint8_t* payload, int payload_len, int seq, float* samples_out, int max_samples,
                      uint8_t* frame_info_out, int max_info, uint8_t* coded_out, int max_coded, int* bits_per_symbol_out) {
    ref_quiet();
    OFDMNvisWaveform tx(named_config(mod, rate));
    tx.configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    CodeRate cr = static_cast<CodeRate>(rate);
    Bytes pl(payload, payload + payload_len);
    Bytes fd = protocol::v2::makeFixedDataFrame("TEST", "RX", static_cast<uint16_t>(seq), pl, cr).serialize();
    int pilots = (tx.config_.num_carriers + tx.config_.pilot_spacing - 1) / tx.config_.pilot_spacing;
    size_t bps = (tx.config_.num_carriers - pilots) * getBitsPerSymbol(static_cast<Modulation>(mod));
    if (bits_per_symbol_out) *bits_per_symbol_out = static_cast<int>(bps);
    Bytes enc = protocol::v2::encodeFixedFrame(fd, cr, true, bps);
    Samples pre = tx.modulator_->generateTrainingSymbols(2);
    Samples dat = tx.modulate(enc);
    size_t info_bytes = 4 * protocol::v2::getBytesPerCodeword(cr);
    if (frame_info_out) { Bytes padded = fd; padded.resize(info_bytes, 0); std::memcpy(frame_info_out, padded.data(), std::min<size_t>(info_bytes, max_info)); }
    if (coded_out) std::memcpy(coded_out, enc.data(), std::min<size_t>(enc.size(), max_coded));
    int n = static_cast<int>(pre.size() + dat.size());
    if (n > max_samples) return -n;
    std::memcpy(samples_out, pre.data(), pre.size() * sizeof(float));
    std::memcpy(samples_out + pre.size(), dat.data(), dat.size() * sizeof(float));
    return n;
}
int ref_rx_process_nvis(int mod, int rate, const float* samples, int n, float cfo_hz, long long abs_pos, int use_abs,
                        float* llr_out, int max_llr, float* aux_out, float* h_out) {
    ref_quiet();
    OFDMNvisWaveform rx(named_config(mod, rate));
    rx.configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    rx.reset();
    if (use_abs) rx.setAbsoluteTrainingPosition(static_cast<size_t>(abs_pos));
    rx.setFrequencyOffset(cfo_hz);
    bool ok = rx.process(SampleSpan(samples, n));
    std::vector<float> soft = rx.getSoftBits();
    int m = static_cast<int>(soft.size());
    if (llr_out) std::memcpy(llr_out, soft.data(), std::min(m, max_llr) * sizeof(float));
    auto* impl = rx.demodulator_->impl_.get();
    if (aux_out) {
        aux_out[0] = rx.demodulator_->getEstimatedSNR();
        aux_out[1] = rx.demodulator_->getFrequencyOffset();
        aux_out[2] = impl->last_fading_index;
        aux_out[3] = impl->noise_variance;
        aux_out[4] = impl->lts_phase_slope;
        aux_out[5] = impl->estimated_snr_linear;
        aux_out[6] = impl->freq_correction_phase;
        aux_out[7] = static_cast<float>(impl->snr_symbol_count);
    }
    if (h_out) {
        for (size_t i = 0; i < impl->all_carrier_fft_indices.size(); ++i) {
            Complex h = impl->channel_estimate[impl->all_carrier_fft_indices[i]];
            h_out[2 * i] = h.real();
            h_out[2 * i + 1] = h.imag();
        }
    }
    return ok ? m : -m;
}

// ---------------------------------------------------------------- HARQ trial chain of the MC-DPSK rungs (BASELINE config 5)
static uint32_t crc32_bytes(const void* p, size_t n) {   // zlib's CRC-32 (so that numpy-side zlib.crc32 checks it)
    static uint32_t tab[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; tab[i] = c; } init = true; }
    uint32_t c = 0xFFFFFFFFu; const uint8_t* b = static_cast<const uint8_t*>(p);
    for (size_t i = 0; i < n; ++i) c = tab[(c ^ b[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
static std::pair<bool, Bytes> shim_robust(const float* cw, CodeRate rate, int* tries) {   // = robustDecodeSingleCW, see ref_robust_decode
    LDPCDecoder decoder(rate);
    decoder.setMaxIterations(fec::LDPCCodec::getRecommendedIterations(rate));
    decoder.setMinSumFactor(0.9375f);
    auto decoded = decoder.decodeSoft(std::span<const float>(cw, 648));
    bool ok = decoder.lastDecodeSuccess();
    int t = 1;
    static constexpr float factors[] = {0.875f, 0.75f, 0.625f, 0.5f};
    for (int retry = 0; retry < 4 && !ok; retry++) { decoder.setMinSumFactor(factors[retry]); decoded = decoder.decodeSoft(std::span<const float>(cw, 648)); ok = decoder.lastDecodeSuccess(); ++t; }
    if (tries) *tries = t;
    Bytes data; if (ok) data.assign(decoded.begin(), decoded.end());
    return {ok, data};
}
// One data codeword (CW index 1 of a 2-CW frame) per trial, retransmitted until it decodes, as
// StreamingDecoder::decodeMCDPSKFrame treats a CW >= 1 (streaming_decoder.cpp:2758-2800): robust decode of the fresh
// reception; on failure ChaseCache::store + getCombined and, from the second reception on, robust decode of the sum.
// Per transmission t of trial i: audio = MultiCarrierDPSKModulator(training + reference + modulate(LDPC R1/4 codeword)),
// sim::WattersonChannel(preset, seeds[i*max_tx + t]), demodulator driven as after an external chirp detection.
// Outputs per (trial, tx): crc32 of the 648 soft bits, crc32 of the cache sum after store (0 = not stored), robust tries of
// the fresh and of the combined decode (0 = not made); per trial: transmissions to success (0 = never), decoded 20 bytes.
int ref_harq_trials(int nc, int bps, int spreading, int kind, float snr_db, const uint8_t* info21, const uint32_t* seeds,
                    int n_trials, int max_tx, int32_t* tx_to_success, uint32_t* llr_crc, uint32_t* acc_crc, int32_t* tries,
                    uint8_t* decoded20, float* fading_out) {
    ref_quiet();
    sim::WattersonChannel::Config cc;
    switch (kind) {
        case 0: cc = sim::itu_r_f1487::awgn(snr_db); break;
        case 1: cc = sim::itu_r_f1487::good(snr_db); break;
        case 2: cc = sim::itu_r_f1487::moderate(snr_db); break;
        case 3: cc = sim::itu_r_f1487::poor(snr_db); break;
        default: cc = sim::itu_r_f1487::flutter(snr_db); break;
    }
    LDPCEncoder enc(CodeRate::R1_4);
    fec::ChaseCache::Config chase_config;
    chase_config.log_combines = false;
    for (int i = 0; i < n_trials; ++i) {
        fec::ChaseCache cache(chase_config);
        fec::ChaseCacheKey key{static_cast<uint16_t>(i & 0xFFFF), 0x123456u, 0x654321u};
        Bytes coded = enc.encode(ByteSpan(info21 + static_cast<size_t>(i) * 21, 21));
        coded.resize(81);
        MultiCarrierDPSKModulator m(mc_config(nc, bps, spreading));
        Samples tr = m.generateTrainingSequence(), rf = m.generateReferenceSymbol(), dt = m.modulate(coded);
        Samples tx; tx.insert(tx.end(), tr.begin(), tr.end()); tx.insert(tx.end(), rf.begin(), rf.end()); tx.insert(tx.end(), dt.begin(), dt.end());
        tx_to_success[i] = 0;
        std::memset(decoded20 + static_cast<size_t>(i) * 20, 0, 20);
        for (int t = 0; t < max_tx; ++t) {
            const size_t q = static_cast<size_t>(i) * max_tx + t;
            llr_crc[q] = acc_crc[q] = 0; tries[2 * q] = tries[2 * q + 1] = 0;
            if (tx_to_success[i]) continue;
            sim::WattersonChannel ch(cc, seeds[q]);
            Samples y = ch.process(SampleSpan(tx.data(), tx.size()));
            MultiCarrierDPSKDemodulator d(mc_config(nc, bps, spreading));
            d.setChirpDetected(0.0f);
            d.setCFOWithPhase(0.0f, 0.0f);
            if (!d.process(SampleSpan(y.data(), y.size()))) return -1;
            if (fading_out) fading_out[q] = d.getFadingIndex();
            std::vector<float> soft = d.getSoftBits();
            if (soft.size() < 648) return -2;
            soft.resize(648);
            llr_crc[q] = crc32_bytes(soft.data(), 648 * sizeof(float));
            auto [ok, data] = shim_robust(soft.data(), CodeRate::R1_4, &tries[2 * q]);
            if (!ok) {
                cache.store(key, 1, soft, 2, protocol::v2::FrameType::DATA);
                auto combined = cache.getCombined(key, 1);
                if (combined && combined->size() == 648) {
                    acc_crc[q] = crc32_bytes(combined->data(), 648 * sizeof(float));
                    if (cache.getCombineCount(key, 1) > 1) {
                        auto [ok2, data2] = shim_robust(combined->data(), CodeRate::R1_4, &tries[2 * q + 1]);
                        if (ok2 && data2.size() >= 20) { ok = true; data = std::move(data2); cache.markDecoded(key, 1); }
                    }
                }
            }
            if (ok && data.size() >= 20) {
                std::memcpy(decoded20 + static_cast<size_t>(i) * 20, data.data(), 20);
                tx_to_success[i] = t + 1;
                cache.markDecoded(key, 1);
            }
        }
    }
    return 0;
}

// ---------------------------------------------------------------- MC-DPSK plug-in (src/waveform/mc_dpsk_waveform.cpp)
// The reference's MCDPSKWaveform object itself, driven as gui::StreamingDecoder / StreamingEncoder drive it.
static std::unique_ptr<MCDPSKWaveform> mc_waveform(int carriers, int mod, int rate, int spreading) {
    auto wf = std::make_unique<MCDPSKWaveform>(carriers);
    wf->configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    wf->setSpreadingMode(spreading == 4 ? SpreadingMode::TIME_4X : spreading == 2 ? SpreadingMode::TIME_2X : SpreadingMode::NONE);
    return wf;
}
// TX: generatePreamble() (dual chirp + training + reference; data_preamble 0) or generateDataPreamble() (ZC DATA root +
// training + reference; 1), then modulate(coded)
int ref_mcdpsk_wf_tx(int carriers, int mod, int rate, int spreading, int data_preamble, const uint8_t* coded, int n_coded, float* out, int max_n) {
    ref_quiet();
    auto wf = mc_waveform(carriers, mod, rate, spreading);
    Samples pre = data_preamble ? wf->generateDataPreamble() : wf->generatePreamble();
    Samples dat = wf->modulate(Bytes(coded, coded + n_coded));
    if (static_cast<int>(pre.size() + dat.size()) > max_n) return -static_cast<int>(pre.size() + dat.size());
    std::memcpy(out, pre.data(), pre.size() * sizeof(float));
    std::memcpy(out + pre.size(), dat.data(), dat.size() * sizeof(float));
    return static_cast<int>(pre.size() + dat.size());
}
// RX on ONE object: reset() -> detectSync (0) | detectDataSync(known_cfo) (1) -> setFrequencyOffset(sync CFO, host rule below) ->
// process(span from start_sample, frame_len samples or all that remain) -> getSoftBits
// (streaming_decoder.cpp:723,733,831,1347-1363).  sync4 = {detected, start_sample, correlation, cfo_hz};
// aux5 = {process() ready, estimatedCFO(), getFadingIndex(), getFrequencyOffset(), isSynced()}.  Returns the soft-bit count.
int ref_mcdpsk_wf_rx(int carriers, int mod, int rate, int spreading, int data_sync, const float* samples, int n, float known_cfo,
                     float threshold, int frame_len, float* sync4, float* llr_out, int max_llr, float* aux5) {
    ref_quiet();
    auto wf = mc_waveform(carriers, mod, rate, spreading);
    wf->reset();
    SyncResult r;
    bool ok = data_sync ? wf->detectDataSync(SampleSpan(samples, n), r, known_cfo, threshold) : wf->detectSync(SampleSpan(samples, n), r, threshold);
    sync4[0] = ok ? 1.0f : 0.0f; sync4[1] = static_cast<float>(r.start_sample); sync4[2] = r.correlation; sync4[3] = r.cfo_hz;
    for (int i = 0; i < 5; ++i) aux5[i] = 0.0f;
    if (!ok || r.start_sample < 0 || r.start_sample >= n) return 0;
    // the host's CFO rule between sync and process (streaming_decoder.cpp:903-917): when connected with an established CFO,
    // a measurement more than 1 Hz away from it is replaced by the established value
    float new_cfo = r.cfo_hz;
    if (data_sync && std::abs(known_cfo) > 0.01f && std::abs(new_cfo - known_cfo) > 1.0f) new_cfo = known_cfo;
    wf->setFrequencyOffset(new_cfo);
    int take = n - r.start_sample;
    if (frame_len > 0 && frame_len < take) take = frame_len;
    bool ready = wf->process(SampleSpan(samples + r.start_sample, take));
    std::vector<float> soft = wf->getSoftBits();
    aux5[0] = ready ? 1.0f : 0.0f; aux5[1] = wf->estimatedCFO(); aux5[2] = wf->getFadingIndex(); aux5[3] = wf->getFrequencyOffset();
    aux5[4] = wf->isSynced() ? 1.0f : 0.0f;
    std::memcpy(llr_out, soft.data(), std::min<size_t>(soft.size(), max_llr) * sizeof(float));
    return static_cast<int>(soft.size());
}
// sizing queries of the object: out4 = {getMinSamplesForFrame, getMinSamplesForCWCount(num_cw), getDataPreambleSamples, getPreambleSamples}
void ref_mcdpsk_wf_sizes(int carriers, int mod, int rate, int spreading, int num_cw, int* out4) {
    ref_quiet();
    auto wf = mc_waveform(carriers, mod, rate, spreading);
    out4[0] = wf->getMinSamplesForFrame(); out4[1] = wf->getMinSamplesForCWCount(num_cw); out4[2] = wf->getDataPreambleSamples(); out4[3] = wf->getPreambleSamples();
}

// ---------------------------------------------------------------- burst chain
// TX of one burst-interleaved group as StreamingEncoder::encodeBurstLight builds it (streaming_encoder.cpp:302-389):
// every frame LDPC-encoded + frame/channel-interleaved (encodeFixedFrame), the group byte-interleaved
// (BurstInterleaver::interleave), then [data preamble][modulated frame] per frame with the first LTS symbol of the
// group's first frame negated.  info: n_frames * info_stride bytes (serialized frames); returns the sample count.
int ref_burst_tx(int mod, int rate, const uint8_t* info, int info_stride, int n_info, int n_frames, int negate_first_lts,
                 float* samples_out, int max_samples, uint8_t* coded_logical_out, uint8_t* coded_physical_out) {
    ref_quiet();
    ModemConfig cfg = named_config(mod, rate);
    OFDMChirpWaveform tx(cfg);
    tx.configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    CodeRate cr = static_cast<CodeRate>(rate);
    int pilots = (cfg.num_carriers + tx.config_.pilot_spacing - 1) / tx.config_.pilot_spacing;
    size_t bps = (cfg.num_carriers - pilots) * getBitsPerSymbol(static_cast<Modulation>(mod));
    std::vector<Bytes> encoded;
    for (int f = 0; f < n_frames; ++f) {
        Bytes fd(info + static_cast<size_t>(f) * info_stride, info + static_cast<size_t>(f) * info_stride + n_info);
        encoded.push_back(protocol::v2::encodeFixedFrame(fd, cr, true, bps));
        if (coded_logical_out) std::memcpy(coded_logical_out + f * 324, encoded.back().data(), 324);
    }
    if (n_frames >= 2) encoded = fec::BurstInterleaver::interleave(encoded);
    std::vector<float> result;
    for (int f = 0; f < n_frames; ++f) {
        if (coded_physical_out) std::memcpy(coded_physical_out + f * 324, encoded[f].data(), 324);
        Samples pre = tx.generateDataPreamble();
        if (f == 0 && negate_first_lts) for (size_t j = 0; j < pre.size() / 2; ++j) pre[j] = -pre[j];
        Samples dat = tx.modulate(encoded[f]);
        result.insert(result.end(), pre.begin(), pre.end());
        result.insert(result.end(), dat.begin(), dat.end());
    }
    int n = static_cast<int>(result.size());
    if (n > max_samples) return -n;
    std::memcpy(samples_out, result.data(), result.size() * sizeof(float));
    return n;
}

// RX of one burst group on ONE waveform object in the order gui::StreamingDecoder drives it:
//   detectDataSync(search span)                                           streaming_decoder.cpp:723-733
//   setAbsoluteTrainingPosition(abs_base + start), setFrequencyOffset     :896, :1347
//   process(first frame)  [consumes the one-shot marker], getSoftBits     :1350-1363
//   wasBurstInterleaved() AFTER process() -> accumulation                 :1380-1407
//   per continuation frame: setFrequencyOffset(burst_cfo), process(block at burst_next_pos_), getSoftBits,
//   estimatedCFO with the 2 Hz drift clamp                                :3127-3208
//   BurstInterleaver::deinterleave + decodeFixedFrame per logical frame   :3210-3240 (decodeFrame :2821)
// sync4 = {detected, start_sample, correlation, marker latched AFTER the first process()}; cfo_used[f] is the CFO
// handed to setFrequencyOffset before frame f; cfo_after[f] = estimatedCFO() after it.  Returns soft bits per frame.
int ref_burst_rx(int mod, int rate, const float* samples, int n, int search_len, int n_frames, float known_cfo, float thr,
                 long long abs_base, float* sync4, float* llr_out, int llr_stride, float* cfo_used, float* cfo_after,
                 float* logical_out, uint8_t* dec_data, uint8_t* dec_ok) {
    ref_quiet();
    ModemConfig cfg = named_config(mod, rate);
    OFDMChirpWaveform rx(cfg);
    rx.configure(static_cast<Modulation>(mod), static_cast<CodeRate>(rate));
    CodeRate cr = static_cast<CodeRate>(rate);
    int pilots = (cfg.num_carriers + rx.config_.pilot_spacing - 1) / rx.config_.pilot_spacing;
    size_t bps = (cfg.num_carriers - pilots) * getBitsPerSymbol(static_cast<Modulation>(mod));
    SyncResult r;
    bool ok = rx.detectDataSync(SampleSpan(samples, std::min(n, search_len)), r, known_cfo, thr);
    sync4[0] = ok ? 1.f : 0.f; sync4[1] = static_cast<float>(r.start_sample); sync4[2] = r.correlation; sync4[3] = 0.f;
    if (!ok) return 0;
    const size_t frame_len = static_cast<size_t>(rx.getMinSamplesForFrame());
    size_t pos = static_cast<size_t>(r.start_sample);
    if (pos + frame_len * n_frames > static_cast<size_t>(n)) return -1;
    rx.setAbsoluteTrainingPosition(static_cast<size_t>(abs_base) + pos);
    float burst_cfo = known_cfo;
    std::vector<std::vector<float>> soft_buffer;
    int n_soft = 0;
    for (int f = 0; f < n_frames; ++f) {
        rx.setFrequencyOffset(burst_cfo);
        cfo_used[f] = burst_cfo;
        if (!rx.process(SampleSpan(samples + pos, frame_len))) return -2 - f;
        std::vector<float> soft = rx.getSoftBits();
        if (f == 0) sync4[3] = rx.wasBurstInterleaved() ? 1.f : 0.f;
        n_soft = static_cast<int>(soft.size());
        std::memcpy(llr_out + static_cast<size_t>(f) * llr_stride, soft.data(), std::min(n_soft, llr_stride) * sizeof(float));
        float corrected = rx.estimatedCFO();
        cfo_after[f] = corrected;
        float drift = corrected - burst_cfo;                       // MAX_(PILOT|BURST)_CFO_DRIFT_HZ = 2.0
        if (std::abs(drift) > 2.0f) corrected = burst_cfo + std::copysign(2.0f, drift);
        burst_cfo = corrected;
        soft_buffer.push_back(std::move(soft));
        pos += frame_len;
    }
    auto logical = fec::BurstInterleaver::deinterleave(soft_buffer);
    size_t bpc = protocol::v2::getBytesPerCodeword(cr);
    for (int f = 0; f < n_frames; ++f) {
        std::memcpy(logical_out + static_cast<size_t>(f) * 2592, logical[f].data(), 2592 * sizeof(float));
        auto st = protocol::v2::decodeFixedFrame(logical[f], cr, true, bps);
        for (int cw = 0; cw < 4; ++cw) {
            dec_ok[f * 4 + cw] = st.decoded[cw] ? 1 : 0;
            std::memset(dec_data + (static_cast<size_t>(f) * 4 + cw) * bpc, 0, bpc);
            if (st.decoded[cw] && st.data[cw].size() >= bpc) std::memcpy(dec_data + (static_cast<size_t>(f) * 4 + cw) * bpc, st.data[cw].data(), bpc);
        }
    }
    return n_soft;
}

}  // extern "C"
