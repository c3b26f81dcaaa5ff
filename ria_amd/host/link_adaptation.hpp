// ria_amd/host/link_adaptation.hpp — the SNR / fading ladder that picks (waveform, modulation, code rate,
// spreading) for a link: protocol::recommendWaveformAndRate, recommendDataMode,
// selectOFDMCodeRate, capInitialOFDMRate (src/protocol/waveform_selection.hpp:49-104,112-222,250-314).
// Host-side scalar logic: the Monte-Carlo sweeps (config 5) ask it which mode to simulate at each grid point.
// Written as ordered threshold TABLES (first matching row wins); tests compare a dense (snr, fading) grid with
// outputs recorded from the reference (tests/golden/link_adaptation.npz).
#pragma once
#include <stdint.h>

#include "../../include/ria_gpu.h"

namespace ria {

constexpr int kWaveMcDpsk = 4, kWaveOfdmChirp = 5;   // protocol::WaveformMode (frame_v2.hpp:28-36)

// ---- selectOFDMCodeRate: (fading below, snr at least) -> rate, else R1/4
inline int select_ofdm_code_rate(float snr_db, float fading) {
    struct Row { float fading_lt, snr_ge; int rate; };
    static const Row rows[] = {{0.15f, 20.0f, RIA_RATE_3_4}, {0.65f, 20.0f, RIA_RATE_2_3}, {1.10f, 15.0f, RIA_RATE_1_2}};
    for (const Row& r : rows)
        if (fading < r.fading_lt && snr_db >= r.snr_ge) return r.rate;
    return RIA_RATE_1_4;
}

// ---- capInitialOFDMRate: one step down unless the chirp-era metrics are near ideal
inline int cap_initial_ofdm_rate(float snr_db, float fading, int candidate) {
    struct Row { int from, to; float fading_ge; float snr_lt; };
    static const Row rows[] = {{RIA_RATE_3_4, RIA_RATE_2_3, 0.05f, 24.0f}, {RIA_RATE_2_3, RIA_RATE_1_2, 0.45f, 24.0f}};
    for (const Row& r : rows)
        if (candidate == r.from) return (fading >= r.fading_ge || snr_db < r.snr_lt) ? r.to : candidate;
    return candidate;
}

inline float dqpsk_throughput(int rate) {
    return rate == RIA_RATE_3_4 ? 3900.0f : rate == RIA_RATE_2_3 ? 3200.0f : rate == RIA_RATE_1_2 ? 2300.0f : 1150.0f;
}
inline float qam16_throughput(int rate) {
    return rate == RIA_RATE_3_4 ? 4800.0f : rate == RIA_RATE_2_3 ? 4000.0f : rate == RIA_RATE_1_2 ? 3000.0f : 1500.0f;
}

// ---- MC-DPSK rungs by SNR: (snr below) -> modulation, spreading, throughput
struct McRung { float snr_lt; int modulation, spreading; float bps; };
inline const McRung* mc_rung(float snr_db) {
    static const McRung rungs[] = {{-7.0f, RIA_MOD_DBPSK, 4, 117.0f}, {-3.0f, RIA_MOD_DBPSK, 2, 235.0f},
                                   {5.0f, RIA_MOD_DBPSK, 1, 469.0f}, {1e30f, RIA_MOD_DQPSK, 1, 938.0f}};
    for (const McRung& r : rungs) if (snr_db < r.snr_lt) return &r;
    return &rungs[3];
}

// ---- recommendWaveformAndRate
inline ria_link_recommendation recommend_waveform_and_rate(float snr_db, float fading) {
    ria_link_recommendation o{};
    o.num_carriers = 10; o.spreading = 1;
    auto mc = [&](const McRung* r) { o.waveform = kWaveMcDpsk; o.modulation = r->modulation; o.code_rate = RIA_RATE_1_4; o.spreading = r->spreading; o.estimated_throughput_bps = r->bps; };
    auto ofdm = [&](int mod, int rate, float bps) { o.waveform = kWaveOfdmChirp; o.modulation = mod; o.code_rate = rate; o.estimated_throughput_bps = bps; };
    if (snr_db < 10.0f) { mc(mc_rung(snr_db)); return o; }
    const int ladder = select_ofdm_code_rate(snr_db, fading);
    if (fading < 0.15f) {            // AWGN: QAM by SNR
        if (snr_db >= 25.0f) ofdm(RIA_MOD_QAM64, RIA_RATE_3_4, 7200.0f);
        else if (snr_db >= 22.0f) ofdm(RIA_MOD_QAM32, RIA_RATE_3_4, 6000.0f);
        else if (snr_db >= 18.0f) ofdm(RIA_MOD_QAM16, ladder, qam16_throughput(ladder));
        else ofdm(RIA_MOD_DQPSK, ladder, dqpsk_throughput(ladder));
    } else if (fading < 0.65f) {     // good fading
        if (snr_db >= 22.0f) ofdm(RIA_MOD_QAM16, RIA_RATE_2_3, 4000.0f);
        else ofdm(RIA_MOD_DQPSK, ladder, dqpsk_throughput(ladder));
    } else if (fading < 1.10f) {     // moderate fading
        ofdm(RIA_MOD_DQPSK, ladder, dqpsk_throughput(ladder));
    } else {                         // heavy fading, SNR >= 10
        ofdm(RIA_MOD_DQPSK, RIA_RATE_1_4, 1150.0f);
    }
    return o;
}

// ---- recommendDataMode (waveform already negotiated)
inline ria_link_recommendation recommend_data_mode(float snr_db, int waveform, float fading) {
    ria_link_recommendation o{};
    o.waveform = waveform; o.num_carriers = 10; o.spreading = 1;
    if (waveform == kWaveMcDpsk) {
        const McRung* r = mc_rung(snr_db);
        o.modulation = r->modulation; o.code_rate = RIA_RATE_1_4; o.spreading = r->spreading; o.estimated_throughput_bps = r->bps;
        return o;
    }
    const int ladder = select_ofdm_code_rate(snr_db, fading);
    o.modulation = RIA_MOD_DQPSK; o.code_rate = ladder;
    if (fading < 0.15f) {
        if (snr_db >= 25.0f) { o.modulation = RIA_MOD_QAM64; o.code_rate = RIA_RATE_3_4; }
        else if (snr_db >= 22.0f) { o.modulation = RIA_MOD_QAM32; o.code_rate = RIA_RATE_3_4; }
        else if (snr_db >= 18.0f) { o.modulation = RIA_MOD_QAM16; }
    } else if (fading < 0.65f && snr_db >= 22.0f) {
        o.modulation = RIA_MOD_QAM16; o.code_rate = RIA_RATE_2_3;
    }
    return o;
}

}  // namespace ria
