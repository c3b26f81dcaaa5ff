// ria_amd/host/gpu_waveform.hpp — host-side mirror of the reference's plugin interface for the RX
// hot path, implemented over the C ABI (include/ria_gpu.h).  Header-only C++17.
//
// The class and method names, argument meaning and error behaviour follow
//   ultra::IWaveform            src/waveform/waveform_interface.hpp:47-220
//   ultra::OFDMChirpWaveform    src/waveform/ofdm_chirp_waveform.cpp
//   protocol::v2::decodeFixedFrame / CodewordStatus   src/protocol/frame_v2.hpp:637-664, :848
//   ultra::LDPCDecoder          include/ultra/fec.hpp:48-81
// so that the parity tests read like the reference's own tools (tools/test_waveform_simple.cpp).
// On the reference side a maintainer makes GpuOfdmChirpWaveform inherit ultra::IWaveform (the
// signatures already match) — see INTEGRATION.md.  There is no CPU fallback: a missing GPU or a
// failed call throws std::runtime_error from the constructor / returns false / decoded[i]=false
// exactly where the reference would.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ria_gpu.h"

namespace ria_host {

enum class Modulation : uint8_t { DBPSK = 0, BPSK = 1, DQPSK = 2, QPSK = 3, D8PSK = 4, QAM8 = 5, QAM16 = 6,
                                  QAM32 = 7, QAM64 = 8, QAM256 = 10 };          // types.hpp:28-39
enum class CodeRate : uint8_t { R1_4, R1_3, R1_2, R2_3, R3_4, R5_6, R7_8 };      // types.hpp:91-100
using Bytes = std::vector<uint8_t>;
struct SampleSpan { const float* ptr; size_t len; const float* data() const { return ptr; } size_t size() const { return len; } };

struct SyncResult {  // waveform_interface.hpp:37-44
    bool detected = false; int start_sample = -1; float correlation = 0.0f; float cfo_hz = 0.0f;
    float snr_estimate = 0.0f; bool has_training = false;
};

struct CodewordStatus {  // frame_v2.hpp:637-664 (the fields decodeFixedFrame fills)
    std::vector<bool> decoded;
    std::vector<Bytes> data;
    std::vector<int> iterations;   // extra: LDPCDecoder::lastIterations() per codeword
    bool allSuccess() const { for (bool d : decoded) if (!d) return false; return !decoded.empty(); }
    int countFailures() const { int n = 0; for (bool d : decoded) n += !d; return n; }
};

class GpuHandle {
public:
    GpuHandle(Modulation mod, CodeRate rate, int device = 0, int max_batch = 64) {
        ria_gpu_config cfg;
        ria_gpu_default_config(&cfg);
        cfg.device = device;
        cfg.modulation = static_cast<int>(mod);
        cfg.code_rate = static_cast<int>(rate);
        cfg.max_batch = max_batch;
        int rc = ria_gpu_create(&cfg, &h_);
        if (rc != RIA_OK) throw std::runtime_error("ria_gpu_create failed: status " + std::to_string(rc));
        ria_gpu_get_geometry(h_, &geo_);
    }
    ~GpuHandle() { ria_gpu_destroy(h_); }
    GpuHandle(const GpuHandle&) = delete;
    GpuHandle& operator=(const GpuHandle&) = delete;
    ria_gpu_handle get() const { return h_; }
    const ria_gpu_geometry& geo() const { return geo_; }
private:
    ria_gpu_handle h_ = nullptr;
    ria_gpu_geometry geo_{};
};

// One configured OFDM-CHIRP waveform object whose process() runs on the GPU (n_frames = 1).
class GpuOfdmChirpWaveform /* : public ultra::IWaveform on the reference side */ {
public:
    explicit GpuOfdmChirpWaveform(Modulation mod = Modulation::QAM16, CodeRate rate = CodeRate::R1_2, int device = 0, bool any_modulation = false)
        : device_(device) { if (any_modulation) configureImpl(mod, rate); else configure(mod, rate); }

    std::string getName() const { return "OFDM-CHIRP (MI355X)"; }
    // configure() rebuilds the internals like the reference does (ofdm_chirp_waveform.cpp:81-107)
    void configure(Modulation mod, CodeRate rate) { configureImpl(allowedOnChirp(mod) ? mod : Modulation::DQPSK, rate); }
    // OFDMChirpWaveform::configure accepts DBPSK/DQPSK/D8PSK/QPSK/BPSK/QAM16/QAM32/QAM64 and maps everything else
    // (QAM8, QAM256) to DQPSK with a warning (ofdm_chirp_waveform.cpp:82-89); OFDM-COX takes any modulation
    static bool allowedOnChirp(Modulation m) {
        return m == Modulation::DBPSK || m == Modulation::DQPSK || m == Modulation::D8PSK || m == Modulation::QPSK ||
               m == Modulation::BPSK || m == Modulation::QAM16 || m == Modulation::QAM32 || m == Modulation::QAM64;
    }
    void configureImpl(Modulation mod, CodeRate rate) {
        mod_ = mod; rate_ = rate;
        gpu_ = std::make_unique<GpuHandle>(mod, rate, device_);
        soft_bits_.clear();
        synced_ = false;
    }
    void setFrequencyOffset(float cfo_hz) { cfo_hz_ = cfo_hz; }                       // :73
    void setAbsoluteTrainingPosition(size_t pos) { abs_pos_ = pos; has_abs_ = true; } // :131
    // the reference keeps TWO flags (ofdm_chirp_waveform.hpp:136-140): the one-shot that process() consumes to undo the
    // LTS negation, and the latch StreamingDecoder reads AFTER process() (streaming_decoder.cpp:1380-1383)
    void setBurstInterleaved(bool v) { burst_marker_ = v; burst_latched_ = v; }
    Modulation getModulation() const { return mod_; }
    CodeRate getCodeRate() const { return rate_; }
    float getFrequencyOffset() const { return cfo_hz_; }

    // process(): samples start at the first LTS sample (ofdm_chirp_waveform.cpp:193-198).  Returns true
    // when at least one codeword worth of soft bits is ready, like processPresynced (demodulator.cpp:1413).
    bool process(SampleSpan samples) {
        const ria_gpu_geometry& g = gpu_->geo();
        soft_bits_.clear();
        if (samples.size() < static_cast<size_t>(g.samples_per_symbol)) return false;
        // The kernel consumes whole frames; shorter spans (the host's 1-CW peeks, Appendix B of
        // SURVEY.md) are zero padded and only the symbols actually present are reported.
        std::vector<float> buf(static_cast<size_t>(g.frame_samples), 0.0f);
        size_t n = std::min(samples.size(), buf.size());
        std::memcpy(buf.data(), samples.data(), n * sizeof(float));
        ria_frame_meta meta{cfo_hz_, burst_marker_ ? 1u : 0u, has_abs_ ? abs_pos_ : training_start_};
        burst_marker_ = false;  // one-shot (ofdm_chirp_waveform.cpp:423); burst_latched_ stays for wasBurstInterleaved()
        std::vector<float> llr(static_cast<size_t>(g.llrs_per_frame));
        ria_frame_status fs{};
        int rc = ria_gpu_rx_frames_host(gpu_->get(), buf.data(), &meta, 1, RIA_RX_DEMOD_ONLY, nullptr, nullptr, llr.data(), &fs);
        if (rc != RIA_OK) return false;
        size_t data_syms = samples.size() / g.samples_per_symbol;
        data_syms = data_syms >= 2 ? data_syms - 2 : 0;
        size_t n_llr = std::min<size_t>(llr.size(), data_syms * g.bits_per_symbol);
        soft_bits_.assign(llr.begin(), llr.begin() + n_llr);
        last_snr_ = fs.snr_db;
        cfo_hz_ = fs.cfo_hz;       // CFO feedback (ofdm_chirp_waveform.cpp:457-464)
        last_cfo_ = fs.cfo_hz;
        fading_index_ = fs.fading_index;
        synced_ = true;
        return soft_bits_.size() >= 648;
    }
    // detectSync(): dual-chirp acquisition (ofdm_chirp_waveform.cpp:162-205 on top of detectDualChirp)
    bool detectSync(SampleSpan samples, SyncResult& result, float threshold = 0.15f) {
        ria_chirp_result r{};
        if (ria_gpu_sync_host(gpu_->get(), 0, samples.data(), static_cast<int>(samples.size()), threshold, 0.0f, 0u, &r) != RIA_OK) return false;
        result.detected = r.success != 0;
        result.correlation = std::max(r.up_correlation, r.down_correlation);
        result.cfo_hz = r.cfo_hz;
        result.has_training = true;
        if (r.success) {
            synced_ = true; last_cfo_ = r.cfo_hz;
            result.start_sample = r.down_chirp_start + 24000 + 4800;   // training starts after the down chirp + gap
            training_start_ = static_cast<size_t>(result.start_sample);
        }
        return result.detected;
    }
    // detectDataSync(): LTS light sync of connected-mode DATA frames (ofdm_chirp_waveform.cpp:207-384)
    bool detectDataSync(SampleSpan samples, SyncResult& result, float known_cfo_hz = 0.0f, float threshold = 0.5f) {
        ria_lts_result r{};
        result.detected = false; result.correlation = 0.0f; result.cfo_hz = known_cfo_hz; result.has_training = true;
        if (ria_gpu_sync_host(gpu_->get(), 1, samples.data(), static_cast<int>(samples.size()), threshold, known_cfo_hz, 0u, &r) != RIA_OK) return false;
        result.correlation = r.correlation;
        burst_marker_ = false;      // both reset at the start of every detection attempt (ofdm_chirp_waveform.cpp:357-359)
        burst_latched_ = false;
        if (r.detected) {
            result.detected = true; result.start_sample = r.start_sample;
            training_start_ = static_cast<size_t>(r.start_sample);
            synced_ = true; last_cfo_ = known_cfo_hz;
            burst_marker_ = r.burst_interleaved != 0;
            burst_latched_ = burst_marker_;
        }
        return result.detected;
    }
    bool wasBurstInterleaved() const { return burst_latched_; }      // ofdm_chirp_waveform.hpp:101
    std::vector<float> getSoftBits() { return std::move(soft_bits_); }                // :135
    void reset() { soft_bits_.clear(); synced_ = false; has_abs_ = false; abs_pos_ = 0; }  // CFO preserved (:474-485)
    bool isSynced() const { return synced_; }
    bool hasData() const { return !soft_bits_.empty(); }
    float estimatedSNR() const { return last_snr_; }
    float estimatedCFO() const { return last_cfo_; }
    float getFadingIndex() const { return fading_index_; }
    int getPilotSpacing() const { return gpu_->geo().pilot_spacing; }
    int getCarrierCount() const { return 59; }
    int getSamplesPerSymbol() const { return gpu_->geo().samples_per_symbol; }
    int getDataPreambleSamples() const { return 2 * getSamplesPerSymbol(); }
    int getMinSamplesForCWCount(int num_cw) const {                                   // ofdm_chirp_waveform.cpp:616-648
        const ria_gpu_geometry& g = gpu_->geo();
        int frame_bits = num_cw * 648;
        int data_symbols = (frame_bits + g.bits_per_symbol - 1) / g.bits_per_symbol;
        return 2 * g.samples_per_symbol + data_symbols * g.samples_per_symbol;
    }
    int getMinSamplesForFrame() const { return getMinSamplesForCWCount(4); }
    const ria_gpu_geometry& geometry() const { return gpu_->geo(); }

protected:
    int device_;
    Modulation mod_ = Modulation::QAM16;
    CodeRate rate_ = CodeRate::R1_2;
    std::unique_ptr<GpuHandle> gpu_;
    std::vector<float> soft_bits_;
    float cfo_hz_ = 0.0f, last_cfo_ = 0.0f, last_snr_ = 0.0f, fading_index_ = 0.0f;
    size_t abs_pos_ = 0, training_start_ = 0;
    bool has_abs_ = false, synced_ = false, burst_marker_ = false, burst_latched_ = false;
};

// OFDM-COX: the same demodulator behind Schmidl-Cox acquisition (src/waveform/ofdm_cox_waveform.cpp).  process()
// is the OFDM-CHIRP one (ofdm_cox_waveform.cpp:164-214 drives the same processPresynced); detectSync() is
// OFDMDemodulator::searchForSync on the GPU, the demodulator's noise-floor tracker carried from call to call as
// the reference's member is (ofdm_sync.cpp:35-46); configure() starts a fresh demodulator (tracker = 0).
class GpuOfdmCoxWaveform : public GpuOfdmChirpWaveform {
public:
    explicit GpuOfdmCoxWaveform(Modulation mod = Modulation::QAM16, CodeRate rate = CodeRate::R1_2, int device = 0)
        : GpuOfdmChirpWaveform(mod, rate, device, true) {}
    std::string getName() const { return "OFDM-COX (MI355X)"; }
    void configure(Modulation mod, CodeRate rate) { configureImpl(mod, rate); noise_floor_ = 0.0f; }   // ofdm_cox_waveform.cpp:71-91
    std::vector<float> generatePreamble() {                                          // ofdm_cox_waveform.cpp:107-112
        std::vector<float> p(8 * static_cast<size_t>(gpu_->geo().samples_per_symbol));
        int n = ria_gpu_cox_preamble(gpu_->get(), p.data(), static_cast<int>(p.size()));
        p.resize(n > 0 ? static_cast<size_t>(n) : 0);
        return p;
    }
    bool detectSync(SampleSpan samples, SyncResult& result, float threshold = 0.8f) {  // ofdm_cox_waveform.cpp:125-158
        ria_cox_result r{};
        if (ria_gpu_sync_host(gpu_->get(), 3, samples.data(), static_cast<int>(samples.size()), threshold, noise_floor_, 0u, &r) != RIA_OK) return false;
        noise_floor_ = r.noise_floor;
        if (!r.found) return false;
        result.detected = true; result.start_sample = r.start_sample; result.cfo_hz = r.cfo_hz;
        result.has_training = true; result.correlation = 0.9f;
        cfo_hz_ = r.cfo_hz; last_cfo_ = r.cfo_hz; synced_ = true;
        training_start_ = static_cast<size_t>(r.start_sample);
        return true;
    }
private:
    float noise_floor_ = 0.0f;
};

// protocol::v2::decodeFixedFrame(interleaved_soft, rate, use_channel_deinterleave, bits_per_symbol)
// (frame_v2.hpp:848).  `wf` supplies the configured handle (rate and bits_per_symbol must match it).
inline CodewordStatus decodeFixedFrame(GpuOfdmChirpWaveform& wf, GpuHandle& gpu, const std::vector<float>& interleaved_soft,
                                       bool use_channel_deinterleave = true) {
    (void)wf;
    CodewordStatus st;
    st.decoded.assign(4, false);
    st.data.assign(4, Bytes());
    st.iterations.assign(4, 0);
    if (interleaved_soft.size() < 2592) return st;  // "not enough data": all failed (frame_v2.cpp:1343-1345)
    const ria_gpu_geometry& g = gpu.geo();
    // single-frame host path: upload LLRs through the samples-free decode entry point
    // (device staging is internal to the library via ria_gpu_decode_host below)
    std::vector<uint8_t> info(static_cast<size_t>(g.info_bytes_per_frame));
    ria_decode_status ds{};
    uint32_t flags = RIA_DECODE_FULL | (use_channel_deinterleave ? 0u : RIA_DECODE_NO_CHANNEL_DEINTERLEAVE);
    int rc = ria_gpu_decode_frames_host(gpu.get(), interleaved_soft.data(), static_cast<int>(interleaved_soft.size()), 1,
                                        flags, info.data(), &ds);
    if (rc != RIA_OK) return st;
    for (int cw = 0; cw < 4; ++cw) {
        st.decoded[cw] = ds.cw_ok[cw] != 0;
        st.iterations[cw] = ds.iterations[cw];
        if (st.decoded[cw]) st.data[cw].assign(info.begin() + cw * g.bytes_per_codeword, info.begin() + (cw + 1) * g.bytes_per_codeword);
    }
    return st;
}

}  // namespace ria_host
