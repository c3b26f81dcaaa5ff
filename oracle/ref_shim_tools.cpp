// oracle/ref_shim_tools.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// The scenarios of the reference's own test programs for this path, rebuilt here by CALLING the reference classes in the
// order those programs do, so that their inputs and the reference's outputs can be handed to the parity tests as arrays
// (the programs themselves only print PASS / FAIL lines):
//   tools/test_zc_sync.cpp     tests 0-4: ZC preambles in silence / noise / with a CFO        -> ref_tool_zc_cases
//   tools/test_spreading.cpp   testAtSNR: LDPC R1/2 + MC-DPSK DBPSK 1x / 2x / 4x in noise      -> ref_tool_spreading_case
//   tools/test_chase_cache.cpp tests 2-3: noisy BPSK codewords, LLR sums of 2 / 4 receptions   -> ref_tool_chase_llrs
//   tools/test_zc_dbpsk.cpp    testAtSNR: ZC preamble + MC-DPSK DBPSK + LDPC R1/2 in noise          -> ref_tool_zc_dbpsk_case
// oracle/Makefile also builds those programs unmodified (make tools); oracle/check_against_ref.py runs them and
// compares their printed tables with what these builders give.  std::mt19937 / std::normal_distribution are the library's;
// the few lines of signal arithmetic (noise scaling, CFO rotation) restate the cited helper functions of those programs.

#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "ultra/dsp.hpp"
#include "ultra/fec.hpp"
#include "ultra/logging.hpp"
#include "ultra/types.hpp"
#include "fec/ldpc_codec.hpp"
#include "psk/multi_carrier_dpsk.hpp"
#include "sync/zc_sync.hpp"

using namespace ultra;

namespace {

// tools/test_zc_sync.cpp:22-39 == tools/test_spreading.cpp:15-30: AWGN at an SNR against the mean power of ALL samples
void tool_add_noise(Samples& signal, float snr_db, std::mt19937& rng) {
    float sig_power = 0.0f;
    for (float s : signal) sig_power += s * s;
    sig_power /= signal.size();
    float snr_linear = std::pow(10.0f, snr_db / 10.0f);
    float noise_power = sig_power / snr_linear;
    float noise_std = std::sqrt(noise_power);
    std::normal_distribution<float> noise(0.0f, noise_std);
    for (float& s : signal) s += noise(rng);
}

// tools/test_zc_sync.cpp:43-63: Hilbert FIR (127 taps) -> rotation by a wrapped float phase -> real part
void tool_apply_cfo(Samples& signal, float cfo_hz, float sample_rate) {
    if (std::abs(cfo_hz) < 0.01f || signal.size() < 128) return;
    HilbertTransform hilbert(127);
    SampleSpan span(signal.data(), signal.size());
    auto analytic = hilbert.process(span);
    float phase = 0.0f;
    float phase_inc = 2.0f * M_PI * cfo_hz / sample_rate;
    for (size_t i = 0; i < signal.size() && i < analytic.size(); i++) {
        Complex rotation(std::cos(phase), std::sin(phase));
        Complex shifted = analytic[i] * rotation;
        signal[i] = shifted.real();
        phase += phase_inc;
        while (phase > M_PI) phase -= 2.0f * M_PI;
        while (phase < -M_PI) phase += 2.0f * M_PI;
    }
}

sync::ZCSync tool_zc() {   // test_zc_sync.cpp:313-323 (the values are the defaults of ZCConfig)
    sync::ZCConfig config;
    config.sequence_length = 127;
    config.upsample_factor = 8;
    config.num_repetitions = 2;
    config.root_ping = 1;
    config.root_pong = 3;
    config.root_data = 5;
    config.root_control = 7;
    return sync::ZCSync(config);
}

Samples padded(const Samples& preamble, size_t pad) {
    Samples signal(pad, 0.0f);
    signal.insert(signal.end(), preamble.begin(), preamble.end());
    signal.resize(signal.size() + pad, 0.0f);
    return signal;
}

}  // namespace

extern "C" {

// Every signal tools/test_zc_sync.cpp hands to ZCSync::detect(signal, 0.2f), in the program's order (seed 42):
//   test 0  4 frame types, 500 samples of silence either side, no noise                  (:262-297)
//   test 1  4 frame types, 1000 either side, 20 dB, ONE generator for the four           (:65-121)
//   test 2  PING, 500 either side, -15 ... 20 dB in steps of 2.5, a fresh generator each (:123-154)
//   test 3  DATA, CFO -15 ... 15 Hz in steps of 5 applied before 15 dB noise             (:156-203)
//   test 4  5 trials x 4 types at 10 dB, generator seed + 100 trial + type               (:205-238)
// signals [max_cases][stride], lengths / test / type / param (SNR or CFO) [max_cases]; res7 as ref_zc_detect's out7.
// Returns the number of cases (50).
int ref_tool_zc_cases(float* signals, int stride, int* lengths, int* test_id, int* types, float* param, float* res7, int max_cases) {
    ultra::g_log_level = LogLevel::NONE;
    sync::ZCSync zc = tool_zc();
    const uint32_t seed = 42;
    const sync::ZCFrameType all[] = {sync::ZCFrameType::PING, sync::ZCFrameType::PONG, sync::ZCFrameType::DATA, sync::ZCFrameType::CONTROL};
    int n = 0;
    auto emit = [&](const Samples& s, int test, sync::ZCFrameType t, float p) {
        if (n >= max_cases || static_cast<int>(s.size()) > stride) { n = -1000000; return; }
        std::memset(signals + static_cast<size_t>(n) * stride, 0, sizeof(float) * stride);
        std::memcpy(signals + static_cast<size_t>(n) * stride, s.data(), s.size() * sizeof(float));
        lengths[n] = static_cast<int>(s.size()); test_id[n] = test; types[n] = static_cast<int>(t); param[n] = p;
        auto r = zc.detect(s, 0.2f);
        float* o = res7 + 7 * n;
        o[0] = r.detected ? 1.f : 0.f; o[1] = static_cast<float>(static_cast<int>(r.frame_type)); o[2] = static_cast<float>(r.start_sample);
        o[3] = r.correlation; o[4] = r.cfo_hz; o[5] = r.snr_estimate; o[6] = static_cast<float>(r.root_detected);
        ++n;
    };
    for (auto t : all) emit(padded(zc.generatePreamble(t), 500), 0, t, 0.0f);
    {
        std::mt19937 rng(seed);
        for (auto t : all) { Samples s = padded(zc.generatePreamble(t), 1000); tool_add_noise(s, 20.0f, rng); emit(s, 1, t, 20.0f); }
    }
    for (float snr = -15.0f; snr <= 20.0f; snr += 2.5f) {
        std::mt19937 rng(seed);
        Samples s = padded(zc.generatePreamble(sync::ZCFrameType::PING), 500);
        tool_add_noise(s, snr, rng);
        emit(s, 2, sync::ZCFrameType::PING, snr);
    }
    for (float cfo : {-15.0f, -10.0f, -5.0f, 0.0f, 5.0f, 10.0f, 15.0f}) {
        std::mt19937 rng(seed);
        Samples s = padded(zc.generatePreamble(sync::ZCFrameType::DATA), 500);
        tool_apply_cfo(s, cfo, zc.getConfig().sample_rate);
        tool_add_noise(s, 15.0f, rng);
        emit(s, 3, sync::ZCFrameType::DATA, cfo);
    }
    for (int trial = 0; trial < 5; trial++)
        for (auto t : all) {
            std::mt19937 rng(seed + trial * 100 + static_cast<int>(t));
            Samples s = padded(zc.generatePreamble(t), 500);
            tool_add_noise(s, 10.0f, rng);
            emit(s, 4, t, 10.0f);
        }
    return n;
}

// tools/test_spreading.cpp:41-151 testAtSNR(snr_db, spreading, seed): 40 random bytes -> LDPCCodec R1/2 -> MC-DPSK DBPSK
// (level4_dbpsk, 10 carriers) training + reference + data -> noise -> processTraining / setReference / demodulateSoft
// -> LDPCCodec::decode.  frame_out [max_n], sizes3 = {training, reference, data} samples, soft_out [max_soft];
// out3 = {decoded, bit errors against the transmitted bytes, number of soft bits}.  Returns the frame length.
int ref_tool_spreading_case(float snr_db, int spreading, uint32_t seed, uint8_t* tx40, float* frame_out, int max_n, int* sizes3,
                            float* soft_out, int max_soft, uint8_t* decoded40, int* out3) {
    ultra::g_log_level = LogLevel::NONE;
    std::mt19937 rng(seed);
    MultiCarrierDPSKConfig cfg = mc_dpsk_presets::level4_dbpsk();
    cfg.spreading_mode = spreading == 4 ? SpreadingMode::TIME_4X : spreading == 2 ? SpreadingMode::TIME_2X : SpreadingMode::NONE;
    cfg.use_dual_chirp = false;
    MultiCarrierDPSKModulator mod(cfg);
    MultiCarrierDPSKDemodulator demod(cfg);
    fec::LDPCCodec ldpc(CodeRate::R1_2);
    const int data_bytes = 40;
    Bytes tx_data(data_bytes);
    for (int i = 0; i < data_bytes; i++) tx_data[i] = rng() & 0xFF;
    std::memcpy(tx40, tx_data.data(), data_bytes);
    Bytes encoded = ldpc.encode(tx_data);
    Samples training = mod.generateTrainingSequence();
    Samples ref = mod.generateReferenceSymbol();
    Samples data = mod.modulate(encoded);
    Samples frame;
    frame.insert(frame.end(), training.begin(), training.end());
    frame.insert(frame.end(), ref.begin(), ref.end());
    frame.insert(frame.end(), data.begin(), data.end());
    tool_add_noise(frame, snr_db, rng);
    const int n = static_cast<int>(frame.size());
    if (n > max_n) return -n;
    std::memcpy(frame_out, frame.data(), frame.size() * sizeof(float));
    sizes3[0] = static_cast<int>(training.size()); sizes3[1] = static_cast<int>(ref.size()); sizes3[2] = static_cast<int>(data.size());
    demod.processTraining(SampleSpan(frame.data(), training.size()));
    demod.setReference(SampleSpan(frame.data() + training.size(), ref.size()));
    std::vector<float> soft = demod.demodulateSoft(SampleSpan(frame.data() + training.size() + ref.size(), data.size()));
    out3[2] = static_cast<int>(soft.size());
    if (static_cast<int>(soft.size()) > max_soft) return -1;
    std::memcpy(soft_out, soft.data(), soft.size() * sizeof(float));
    auto [success, decoded] = ldpc.decode(soft);
    out3[0] = 0; out3[1] = 0;
    std::memset(decoded40, 0, data_bytes);
    if (success && decoded.size() >= tx_data.size()) {
        out3[0] = 1;
        std::memcpy(decoded40, decoded.data(), data_bytes);
        for (size_t i = 0; i < tx_data.size(); i++) out3[1] += __builtin_popcount(static_cast<unsigned>(tx_data[i] ^ decoded[i]));
    }
    return n;
}

// tools/test_chase_cache.cpp:21-62, :65-88, :154-262: ONE generator (42) feeds test 2 (100 trials x 2 receptions at 2.5 dB) and
// then test 3 (50 trials x 4 receptions at 1.5 dB); the codeword is LDPCCodec R1/2 of bytes 0..39 as +-4 LLRs, a reception is
// 2 (sign + n) snr with n ~ N(0, 1 / snr).  llrs [400][648] in generation order.  ok_out [100][2] = {single, sum of 2} then
// [50][3] = {1, sum of 2, sum of 4}: LDPCCodec::decode of the sums formed left to right as the program does.
int ref_tool_chase_llrs(float* llrs, uint8_t* ok_out) {
    ultra::g_log_level = LogLevel::NONE;
    fec::LDPCCodec codec;
    codec.setRate(CodeRate::R1_2);
    std::mt19937 rng(42);
    std::vector<uint8_t> test_data(40);
    for (size_t i = 0; i < test_data.size(); i++) test_data[i] = static_cast<uint8_t>(i & 0xFF);
    auto noisy = [&](float snr_db) {
        auto encoded_bytes = codec.encode(test_data);
        std::vector<float> l;
        l.reserve(648);
        for (uint8_t byte : encoded_bytes)
            for (int b = 7; b >= 0; b--) l.push_back(((byte >> b) & 1) ? -4.0f : 4.0f);
        while (l.size() < 648) l.push_back(4.0f);
        l.resize(648);
        float snr_linear = std::pow(10.0f, snr_db / 10.0f);
        float noise_std = 1.0f / std::sqrt(snr_linear);
        std::normal_distribution<float> noise(0.0f, noise_std);
        for (auto& llr : l) {
            float sign = (llr > 0) ? 1.0f : -1.0f;
            float received = sign + noise(rng);
            llr = 2.0f * received * snr_linear;
        }
        return l;
    };
    int w = 0, k = 0;
    auto put = [&](const std::vector<float>& l) { std::memcpy(llrs + static_cast<size_t>(w++) * 648, l.data(), 648 * sizeof(float)); };
    for (int t = 0; t < 100; t++) {
        auto l1 = noisy(2.5f), l2 = noisy(2.5f);
        put(l1); put(l2);
        ok_out[k++] = codec.decode(l1).first ? 1 : 0;
        std::vector<float> c(648);
        for (size_t i = 0; i < 648; i++) c[i] = l1[i] + l2[i];
        ok_out[k++] = codec.decode(c).first ? 1 : 0;
    }
    for (int t = 0; t < 50; t++) {
        auto l1 = noisy(1.5f), l2 = noisy(1.5f), l3 = noisy(1.5f), l4 = noisy(1.5f);
        put(l1); put(l2); put(l3); put(l4);
        ok_out[k++] = codec.decode(l1).first ? 1 : 0;
        std::vector<float> c2(648), c4(648);
        for (size_t i = 0; i < 648; i++) c2[i] = l1[i] + l2[i];
        ok_out[k++] = codec.decode(c2).first ? 1 : 0;
        for (size_t i = 0; i < 648; i++) c4[i] = l1[i] + l2[i] + l3[i] + l4[i];
        ok_out[k++] = codec.decode(c4).first ? 1 : 0;
    }
    return w;
}

// tools/test_zc_dbpsk.cpp:50-232 testAtSNR(snr_db, seed): 500 samples of silence + ZC preamble (DATA) + MC-DPSK DBPSK
// (level4_dbpsk) training / reference / LDPC R1/2 data + 500 of silence -> noise -> ZCSync::detect(0.2) -> the demodulator's
// process() from start_sample with the ZC's CFO and 81 expected bytes -> 648 soft bits -> LDPCCodec::decode.
// signal_out [max_n], zc7 as ref_zc_detect's out7, soft648, tx40 / decoded40; out4 = {sync detected, data decoded, bit errors,
// stage reached: 0 no sync, 1 bad start_sample, 2 frame not ready, 3 fewer than 648 soft bits, 4 decode failed, 5 decoded}.
// Returns the signal length.
int ref_tool_zc_dbpsk_case(float snr_db, uint32_t seed, uint8_t* tx40, float* signal_out, int max_n, float* zc7, float* soft648,
                           uint8_t* decoded40, int* out4) {
    ultra::g_log_level = LogLevel::NONE;
    std::mt19937 rng(seed);
    sync::ZCSync zc_sync = tool_zc();
    MultiCarrierDPSKConfig cfg = mc_dpsk_presets::level4_dbpsk();
    cfg.use_dual_chirp = false;
    MultiCarrierDPSKModulator mod(cfg);
    MultiCarrierDPSKDemodulator demod(cfg);
    fec::LDPCCodec ldpc(CodeRate::R1_2);
    const int data_bytes = 40;
    Bytes tx_data(data_bytes);
    for (int i = 0; i < data_bytes; i++) tx_data[i] = rng() & 0xFF;
    std::memcpy(tx40, tx_data.data(), data_bytes);
    Samples zc_preamble = zc_sync.generatePreamble(sync::ZCFrameType::DATA);
    Samples training = mod.generateTrainingSequence();
    Samples ref = mod.generateReferenceSymbol();
    Bytes encoded = ldpc.encode(tx_data);
    mod.reset();
    Samples data = mod.modulate(encoded);
    Samples sig(500, 0.0f);
    sig.insert(sig.end(), zc_preamble.begin(), zc_preamble.end());
    sig.insert(sig.end(), training.begin(), training.end());
    sig.insert(sig.end(), ref.begin(), ref.end());
    sig.insert(sig.end(), data.begin(), data.end());
    sig.resize(sig.size() + 500, 0.0f);
    tool_add_noise(sig, snr_db, rng);
    const int n = static_cast<int>(sig.size());
    if (n > max_n) return -n;
    std::memcpy(signal_out, sig.data(), sig.size() * sizeof(float));
    std::memset(soft648, 0, 648 * sizeof(float)); std::memset(decoded40, 0, data_bytes);
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    auto r = zc_sync.detect(SampleSpan(sig.data(), sig.size()), 0.2f, false);
    zc7[0] = r.detected ? 1.f : 0.f; zc7[1] = static_cast<float>(static_cast<int>(r.frame_type)); zc7[2] = static_cast<float>(r.start_sample);
    zc7[3] = r.correlation; zc7[4] = r.cfo_hz; zc7[5] = r.snr_estimate; zc7[6] = static_cast<float>(r.root_detected);
    out4[0] = r.detected ? 1 : 0;
    if (!r.detected) return n;
    out4[3] = 1;
    int start = r.start_sample;
    if (start < 0 || start >= static_cast<int>(sig.size()) - 1000) return n;
    out4[3] = 2;
    Samples dpsk(sig.begin() + start, sig.end() - 500);
    demod.reset();
    demod.setExpectedDataBytes(encoded.size());
    demod.setChirpDetected(r.cfo_hz);
    if (!demod.process(SampleSpan(dpsk.data(), dpsk.size()))) return n;
    out4[3] = 3;
    auto soft = demod.getSoftBits();
    if (soft.size() < 648) return n;
    out4[3] = 4;
    soft.resize(648);
    std::memcpy(soft648, soft.data(), 648 * sizeof(float));
    auto [success, decoded] = ldpc.decode(soft);
    if (!success || decoded.empty()) return n;
    out4[3] = 5;
    out4[1] = 1;
    std::memcpy(decoded40, decoded.data(), data_bytes);
    for (int i = 0; i < data_bytes; i++) out4[2] += __builtin_popcount(static_cast<unsigned>(tx_data[i] ^ decoded[i]));
    return n;
}

}  // extern "C"
