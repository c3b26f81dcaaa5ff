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
#include <cmath>
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

// MC-DPSK: the low-SNR plug-in (src/waveform/mc_dpsk_waveform.cpp; SURVEY.md 2 row 7, configs 1 and 5's low rungs).
// Mirrors ultra::MCDPSKWaveform: 3..20 carriers, DBPSK / DQPSK, 1x / 2x / 4x time spreading, dual-chirp acquisition for
// PING / PONG (detectSync), Zadoff-Chu acquisition for connected DATA / CONTROL frames (detectDataSync), and the demodulator
// driven as after an external chirp detection (process).  Every call runs on the GPU through the C ABI; there is no CPU
// fallback.  The library handle is bound to the codeword rate the host decodes MC-DPSK codewords with (R1/4 by default).
enum class SpreadingMode : uint8_t { NONE = 0, TIME_2X = 1, TIME_4X = 2 };        // multi_carrier_dpsk.hpp:27-31
class GpuMcDpskWaveform /* : public ultra::IWaveform on the reference side */ {
public:
    explicit GpuMcDpskWaveform(int num_carriers = 8, int device = 0) : device_(device) {   // mc_dpsk_waveform.cpp:10-19
        num_carriers_ = std::max(3, std::min(20, num_carriers));
        gpu_ = std::make_unique<GpuHandle>(Modulation::DQPSK, CodeRate::R1_4, device_);
    }
    std::string getName() const { return "MC-DPSK (MI355X)"; }
    // configure(): DBPSK / DQPSK / D8PSK are accepted, anything else becomes DQPSK (mc_dpsk_waveform.cpp:80-112).  The GPU
    // demodulator implements 1 and 2 bits per symbol; with D8PSK (3 bits, not on the reference's ladder) process() fails loudly.
    void configure(Modulation mod, CodeRate rate) {
        modulation_ = (mod == Modulation::DQPSK || mod == Modulation::DBPSK || mod == Modulation::D8PSK) ? mod : Modulation::DQPSK;
        code_rate_ = rate;
        bits_per_symbol_ = (mod == Modulation::DBPSK) ? 1 : (mod == Modulation::D8PSK) ? 3 : 2;
        initComponents();
    }
    void setFrequencyOffset(float cfo_hz) { cfo_hz_ = cfo_hz; demod_cfo_ = cfo_hz; }     // :114-119 (demodulator_->setCFO)
    void setCarrierCount(int carriers) { num_carriers_ = std::max(3, std::min(20, carriers)); initComponents(); }   // :418-421
    void setSpreadingMode(SpreadingMode mode) { spreading_ = mode; initComponents(); }        // :423-430
    SpreadingMode getSpreadingMode() const { return spreading_; }
    Modulation getModulation() const { return modulation_; }
    CodeRate getCodeRate() const { return code_rate_; }
    float getFrequencyOffset() const { return cfo_hz_; }
    bool supportsDataPreamble() const { return true; }

    // ---- TX (bit-identical audio; used by tests and simulators)
    std::vector<float> generatePreamble() {                 // dual chirp + training + reference (multi_carrier_dpsk.hpp:126-139)
        std::vector<float> out(57600);
        int n = ria_gpu_chirp_preamble(gpu_->get(), out.data(), static_cast<int>(out.size()));
        out.resize(n > 0 ? static_cast<size_t>(n) : 0);
        std::vector<float> tr = trainingAndReference();
        out.insert(out.end(), tr.begin(), tr.end());
        return out;
    }
    std::vector<float> generateDataPreamble() {             // ZC (DATA root) + training + reference (mc_dpsk_waveform.cpp:136-167)
        std::vector<float> out(4096);
        int n = ria_gpu_zc_preamble(gpu_->get(), 5, out.data(), static_cast<int>(out.size()));
        out.resize(n > 0 ? static_cast<size_t>(n) : 0);
        std::vector<float> tr = trainingAndReference();
        out.insert(out.end(), tr.begin(), tr.end());
        return out;
    }
    std::vector<float> modulate(const Bytes& encoded) {     // data symbols only (MultiCarrierDPSKModulator::modulate)
        std::vector<float> all = modulateAll(encoded);
        const size_t skip = static_cast<size_t>(9 * 512);
        return all.size() > skip ? std::vector<float>(all.begin() + skip, all.end()) : std::vector<float>();
    }

    // ---- RX
    // detectSync(): dual-chirp detection; start_sample = first TRAINING sample, from the down chirp (mc_dpsk_waveform.cpp:176-225)
    bool detectSync(SampleSpan samples, SyncResult& result, float threshold = 0.15f) {
        ria_chirp_result r{};
        if (ria_gpu_sync_host(gpu_->get(), 0, samples.data(), static_cast<int>(samples.size()), threshold, 0.0f, 0u, &r) != RIA_OK) return false;
        result.detected = r.success != 0;
        result.start_sample = r.up_chirp_start;
        result.correlation = std::max(r.up_correlation, r.down_correlation);
        result.cfo_hz = r.cfo_hz;
        result.has_training = true;
        if (r.success) {
            synced_ = true; last_cfo_ = r.cfo_hz;
            result.start_sample = r.down_chirp_start + 24000 + 4800;
        }
        return result.detected;
    }
    // detectDataSync(): ZC preamble of connected DATA / CONTROL frames, roots DATA | CONTROL only, the receiver's known CFO
    // mixed out first; the reported CFO is the residual, the object's own estimate known + residual (mc_dpsk_waveform.cpp:227-292)
    bool detectDataSync(SampleSpan samples, SyncResult& result, float known_cfo_hz = 0.0f, float threshold = 0.2f) {
        if (std::fabs(known_cfo_hz) > 0.1f) setFrequencyOffset(known_cfo_hz);
        ria_zc_result r{};
        constexpr uint32_t DATA_CONTROL_ROOTS = (1u << 2) | (1u << 3);      // ZC_ROOT_MASK_DATA | ZC_ROOT_MASK_CONTROL
        if (ria_gpu_sync_host(gpu_->get(), 2, samples.data(), static_cast<int>(samples.size()), threshold, known_cfo_hz, DATA_CONTROL_ROOTS, &r) != RIA_OK)
            return false;
        result.detected = r.detected != 0;
        result.correlation = r.correlation;
        result.cfo_hz = r.cfo_hz;
        result.has_training = true;
        if (r.detected) {
            synced_ = true; connected_ = true;
            result.start_sample = r.start_sample;
            last_cfo_ = (std::fabs(known_cfo_hz) > 0.1f) ? known_cfo_hz + r.cfo_hz : r.cfo_hz;
            cfo_hz_ = last_cfo_;
        }
        return result.detected;
    }
    // process(): samples = training (8 x 512) + reference (512) + data; demodulator in its "chirp detected externally" state
    // with the object's CFO (mc_dpsk_waveform.cpp:294-338 -> multi_carrier_dpsk.hpp:797-895).  false = not enough samples yet.
    bool process(SampleSpan samples) {
        soft_bits_.clear();
        demod_cfo_ = cfo_hz_;                               // setChirpDetected(cfo_hz_)
        const size_t preamble = 9 * 512;
        if (samples.size() <= preamble) return false;       // the demodulator waits for at least one codeword of data
        if (bits_per_symbol_ == 3) throw std::runtime_error("GpuMcDpskWaveform: 3 bits per symbol (D8PSK) is not implemented on the GPU path");
        if (samples.size() < preamble + 512) return false;  // less than one data symbol: nothing to demodulate
        ria_mcdpsk_config cfg{num_carriers_, bits_per_symbol_, spreadingFactor(), 0};
        const int n = static_cast<int>(samples.size());
        const int max_llr = std::max(1, ((n - 9 * 512) / 512) / spreadingFactor()) * num_carriers_ * bits_per_symbol_;
        std::vector<float> llr(static_cast<size_t>(max_llr));
        ria_mcdpsk_status st{};
        if (ria_gpu_mcdpsk_demod_host(gpu_->get(), &cfg, samples.data(), n, cfo_hz_, 0.0f, llr.data(), max_llr, &st) != RIA_OK) return false;
        llr.resize(static_cast<size_t>(st.n_llr));
        soft_bits_ = std::move(llr);
        demod_cfo_ = st.cfo_hz; fading_index_ = st.fading_index;
        synced_ = true;
        return true;
    }
    std::vector<float> getSoftBits() { return std::move(soft_bits_); }
    void reset() { soft_bits_.clear(); synced_ = false; }   // CFO preserved (mc_dpsk_waveform.cpp:344-352)
    bool isSynced() const { return synced_; }
    bool hasData() const { return !soft_bits_.empty(); }
    float estimatedSNR() const { return last_snr_; }
    float estimatedCFO() const { return demod_cfo_; }       // demodulator_->getEstimatedCFO()
    float getFadingIndex() const { return fading_index_; }
    bool isFading() const { return fading_index_ > 0.65f; }
    int getCarrierCount() const { return num_carriers_; }
    int getSamplesPerSymbol() const { return 512; }
    int getPreambleSamples() const { return 2 * 24000 + 2 * 4800; }                        // chirp_sync_->getTotalSamples()
    int getDataPreambleSamples() const { return 2512 + 8 * 512 + 512; }                     // :405-416
    int getMinSamplesForFrame() const { return getMinSamplesForCWCount(1); }                // :432-455
    int getMinSamplesForCWCount(int num_cw) const {                                         // :457-475
        const int bits_per_symbol = num_carriers_ * bits_per_symbol_;
        const int data_symbols_per_cw = (648 + bits_per_symbol - 1) / bits_per_symbol;
        return 8 * 512 + 512 + num_cw * data_symbols_per_cw * 512 * spreadingFactor();
    }
    GpuHandle& handle() { return *gpu_; }

private:
    int spreadingFactor() const { return spreading_ == SpreadingMode::TIME_4X ? 4 : spreading_ == SpreadingMode::TIME_2X ? 2 : 1; }
    void initComponents() { soft_bits_.clear(); demod_cfo_ = 0.0f; fading_index_ = 0.0f; }  // a fresh demodulator; cfo_hz_ and synced_ are the waveform's own
    std::vector<float> modulateAll(const Bytes& encoded) {
        if (bits_per_symbol_ == 3) throw std::runtime_error("GpuMcDpskWaveform: D8PSK is not implemented on the GPU path");
        ria_mcdpsk_config cfg{num_carriers_, bits_per_symbol_, spreadingFactor(), 0};
        std::vector<float> out(static_cast<size_t>(9 * 512) + (encoded.size() * 8 + 1) * 512 * 4);
        static const uint8_t none = 0;
        int n = ria_gpu_mcdpsk_modulate_host(gpu_->get(), &cfg, encoded.empty() ? &none : encoded.data(), static_cast<int>(encoded.size()), out.data(),
                                             static_cast<int>(out.size()));
        out.resize(n > 0 ? static_cast<size_t>(n) : 0);
        return out;
    }
    std::vector<float> trainingAndReference() { std::vector<float> a = modulateAll(Bytes()); a.resize(std::min<size_t>(a.size(), 9 * 512)); return a; }

    int device_;
    int num_carriers_ = 8, bits_per_symbol_ = 2;
    SpreadingMode spreading_ = SpreadingMode::NONE;
    Modulation modulation_ = Modulation::DQPSK;
    CodeRate code_rate_ = CodeRate::R1_4;
    std::unique_ptr<GpuHandle> gpu_;
    std::vector<float> soft_bits_;
    float cfo_hz_ = 0.0f, last_cfo_ = 0.0f, last_snr_ = 0.0f, demod_cfo_ = 0.0f, fading_index_ = 0.0f;
    bool synced_ = false, connected_ = false;
};

// robustDecodeSingleCW(llr, 648, rate) (streaming_decoder.cpp:1028-1058): the per-codeword decoder of the MC-DPSK and
// control-frame paths - min-sum factor 0.9375, then 0.875 / 0.75 / 0.625 / 0.5 - on the GPU.  `gpu` must be bound to `rate`.
inline std::pair<bool, Bytes> robustDecodeSingleCW(GpuHandle& gpu, const float* llr648, int* tries = nullptr) {
    const int nb = (gpu.geo().ldpc_k + 7) / 8;
    Bytes out(static_cast<size_t>(nb));
    uint8_t ok = 0, tr = 0;
    if (ria_gpu_ldpc_decode_robust_host(gpu.get(), llr648, 1, out.data(), &ok, nullptr, &tr) != RIA_OK) return {false, Bytes()};
    if (tries) *tries = tr;
    return {ok != 0, ok ? out : Bytes()};
}

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
