// ria_amd/csrc/host_tables.hpp — host-side construction of the constant tables the kernels read.
//
// Product code (not the oracle): built once per handle in ria_gpu_create() and uploaded to HBM.
// Each builder cites the reference routine whose result it must equal; tests compare the uploaded
// tables against the oracle's.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <random>
#include <vector>

#include "../../include/ria_gpu.h"

namespace ria {

constexpr int kFFT = 1024;
constexpr int kCP = 128;
constexpr int kSym = kFFT + kCP;
constexpr int kCarriers = 59;
constexpr int kCwBits = 648;
constexpr int kFrameBits = 2592;
constexpr int kMaxRowDeg = 7;   // max_check_degree 6 + identity column (ldpc_decoder.cpp:87,124-129)

inline bool is_coherent(int mod) {
    return !(mod == RIA_MOD_DBPSK || mod == RIA_MOD_DQPSK || mod == RIA_MOD_D8PSK);
}
inline int bits_per_carrier(int mod) {  // types.hpp:42-56
    switch (mod) {
        case RIA_MOD_DBPSK: case RIA_MOD_BPSK: return 1;
        case RIA_MOD_DQPSK: case RIA_MOD_QPSK: return 2;
        case RIA_MOD_D8PSK: return 3;
        case RIA_MOD_QAM16: return 4;
        case RIA_MOD_QAM32: return 5;
        case RIA_MOD_QAM64: return 6;
        case RIA_MOD_QAM256: return 8;
        default: return 0;
    }
}
inline int pilot_spacing_for(int mod, int rate) {  // ofdm_link_adaptation.hpp:26-70
    if (is_coherent(mod)) {
        if (rate == RIA_RATE_5_6 || rate == 6) return 6;
        if (rate == RIA_RATE_3_4) return 8;
        return 5;
    }
    if (mod == RIA_MOD_D8PSK) return (rate == RIA_RATE_3_4 || rate == RIA_RATE_2_3 || rate == RIA_RATE_1_2) ? 8 : 10;
    return (rate == RIA_RATE_3_4) ? 15 : 10;
}
inline int info_bits_for(int rate) {  // frame_v2.hpp:671-681
    switch (rate) {
        case RIA_RATE_1_4: return 162;
        case RIA_RATE_1_3: return 216;
        case RIA_RATE_1_2: return 324;
        case RIA_RATE_2_3: return 432;
        case RIA_RATE_3_4: return 486;
        case RIA_RATE_5_6: return 540;
        default: return 162;
    }
}
inline int recommended_iterations(int rate) {  // ldpc_codec.hpp:86-95
    switch (rate) {
        case RIA_RATE_3_4: return 60;
        case RIA_RATE_2_3: return 70;
        case RIA_RATE_1_2: return 80;
        case RIA_RATE_1_3: return 60;
        default: return 50;
    }
}

// ---------------------------------------------------------------- carrier plan + sequences
struct CarrierPlan {
    int spacing = 0, n_pilot = 0, n_data = 0;
    int all_bin[kCarriers];      // logical carrier -> FFT bin          (demodulator.cpp:45-76)
    int is_pilot[kCarriers];
    int data_bin[kCarriers], pilot_bin[kCarriers];
    int data_logical[kCarriers], pilot_logical[kCarriers];
    float sync_re[kCarriers], sync_im[kCarriers];   // ZC-59 LTS         (demodulator.cpp:78-87)
    float pilot_seq[kCarriers];                     // BPSK pilots       (demodulator.cpp:89-94)
    int interp_lo[kCarriers], interp_hi[kCarriers]; // pilot ordinals    (demodulator.cpp:145-202)
    float interp_alpha[kCarriers];
};

inline CarrierPlan build_carrier_plan(int mod, int rate) {
    CarrierPlan p{};
    p.spacing = pilot_spacing_for(mod, rate);
    int neg = kCarriers / 2, pos = (kCarriers + 1) / 2, l = 0;
    for (int i = -neg; i <= pos; ++i) {
        if (i == 0) continue;
        int bin = (i + kFFT) % kFFT;
        bool isp = (l % p.spacing) == 0;
        p.all_bin[l] = bin;
        p.is_pilot[l] = isp;
        if (isp) { p.pilot_bin[p.n_pilot] = bin; p.pilot_logical[p.n_pilot++] = l; }
        else { p.data_bin[p.n_data] = bin; p.data_logical[p.n_data++] = l; }
        ++l;
    }
    for (int n = 0; n < kCarriers; ++n) {
        float phase = static_cast<float>(-M_PI * 1.0 * static_cast<double>(n) * static_cast<double>(n + 1) /
                                         static_cast<double>(kCarriers));
        p.sync_re[n] = std::cos(phase);
        p.sync_im[n] = std::sin(phase);
    }
    std::mt19937 rng(0x50494C54u);
    for (int i = 0; i < p.n_pilot; ++i) p.pilot_seq[i] = (rng() & 1u) ? 1.0f : -1.0f;
    int d = 0;
    for (int ci = 0; ci < kCarriers; ++ci) {
        if (p.is_pilot[ci]) continue;
        int lo = -1, hi = -1;
        for (int j = ci - 1; j >= 0; --j) if (p.is_pilot[j]) { lo = j; break; }
        for (int j = ci + 1; j < kCarriers; ++j) if (p.is_pilot[j]) { hi = j; break; }
        float alpha = 0.5f;
        if (lo >= 0 && hi >= 0) {
            float total = static_cast<float>(hi - lo);
            alpha = (total > 0) ? static_cast<float>(ci - lo) / total : 0.5f;
        }
        p.interp_lo[d] = lo >= 0 ? lo / p.spacing : -1;
        p.interp_hi[d] = hi >= 0 ? hi / p.spacing : -1;
        p.interp_alpha[d] = alpha;
        ++d;
    }
    return p;
}

// ---------------------------------------------------------------- FFT twiddles / NCO table
inline std::vector<float> build_twiddles() {  // fft.cpp:83-87, interleaved re,im for k < 512
    std::vector<float> tw(kFFT);
    for (int k = 0; k < kFFT / 2; ++k) {
        float angle = static_cast<float>(-2.0f * M_PI * static_cast<double>(k) / static_cast<double>(kFFT));
        tw[2 * k] = std::cos(angle);
        tw[2 * k + 1] = std::sin(angle);
    }
    return tw;
}
// The RX mixer restarts at phase 0 for every frame (demodulator.cpp:1270) and its float phase
// recurrence does not depend on the data, so one table of (cos, sin) serves all frames
// (filters.cpp:228-238).  n samples, interleaved.
inline std::vector<float> build_nco_table(int n, float freq = 1500.0f, float fs = 48000.0f) {
    std::vector<float> t(2 * static_cast<size_t>(n));
    float phase = 0.0f;
    float inc = static_cast<float>(2.0f * M_PI * static_cast<double>(freq) / static_cast<double>(fs));
    for (int i = 0; i < n; ++i) {
        t[2 * i] = std::cos(phase);
        t[2 * i + 1] = std::sin(phase);
        phase += inc;
        if (static_cast<double>(phase) > 2.0f * M_PI) phase = static_cast<float>(static_cast<double>(phase) - 2.0f * M_PI);
        if (phase < 0) phase = static_cast<float>(static_cast<double>(phase) + 2.0f * M_PI);
    }
    return t;
}

// ---------------------------------------------------------------- LDPC H = [H_data | I]
struct LdpcCode {
    int rate = 0, k = 0, m = 0, n = 0, n_edges = 0, max_col_deg = 0;
    std::vector<std::vector<int>> rows;          // variable indices per check, reference edge order
    // device layout (message slot of edge s of check i lives at s*m + i):
    std::vector<uint8_t> row_deg;                // [m]
    std::vector<uint16_t> row_var;               // [kMaxRowDeg][m], 0xFFFF padded
    std::vector<uint8_t> col_deg;                // [n]
    std::vector<uint16_t> col_slot;              // [max_col_deg][n] message addresses, ascending check order
};

inline LdpcCode build_ldpc(int rate) {  // ldpc_decoder.cpp:21-36, :65-138 (encoder twin ldpc_encoder.cpp:70-129)
    LdpcCode c;
    c.rate = rate;
    switch (rate) {
        case RIA_RATE_1_4: c.k = 162; c.m = 486; break;
        case RIA_RATE_1_2: c.k = 324; c.m = 324; break;
        case RIA_RATE_2_3: c.k = 432; c.m = 216; break;
        case RIA_RATE_3_4: c.k = 486; c.m = 162; break;
        case RIA_RATE_5_6: c.k = 540; c.m = 108; break;
        default: c.k = 324; c.m = 324; break;
    }
    c.n = c.k + c.m;
    int k = c.k, m = c.m;
    std::mt19937 rng(static_cast<uint32_t>(0x12345678 + rate));
    int target_var = std::max(3, (4 * m) / k);
    target_var = std::min(target_var, m / 2);
    const int max_check = 6;
    c.rows.assign(m, {});
    std::vector<int> deg(m, 0), avail;
    for (int j = 0; j < k; ++j) {
        avail.clear();
        for (int i = 0; i < m; ++i) if (deg[i] < max_check) avail.push_back(i);
        for (size_t i = avail.size(); i > 1; --i) {
            size_t r = rng() % i;
            std::swap(avail[i - 1], avail[r]);
        }
        int conn = std::min(target_var, static_cast<int>(avail.size()));
        for (int d = 0; d < conn; ++d) { c.rows[avail[d]].push_back(j); deg[avail[d]]++; }
    }
    for (int i = 0; i < m; ++i) if (c.rows[i].empty()) c.rows[i].push_back(static_cast<int>(rng() % k));
    for (int i = 0; i < m; ++i) c.rows[i].push_back(k + i);

    c.row_deg.assign(m, 0);
    c.row_var.assign(static_cast<size_t>(kMaxRowDeg) * m, 0xFFFF);
    std::vector<std::vector<uint16_t>> cols(c.n);
    for (int i = 0; i < m; ++i) {
        c.row_deg[i] = static_cast<uint8_t>(c.rows[i].size());
        for (size_t s = 0; s < c.rows[i].size(); ++s) {
            c.row_var[s * m + i] = static_cast<uint16_t>(c.rows[i][s]);
            cols[c.rows[i][s]].push_back(static_cast<uint16_t>(s * m + i));  // i ascending => check order
            c.n_edges++;
        }
    }
    for (auto& v : cols) c.max_col_deg = std::max(c.max_col_deg, static_cast<int>(v.size()));
    c.col_deg.assign(c.n, 0);
    c.col_slot.assign(static_cast<size_t>(c.max_col_deg) * c.n, 0);
    for (int j = 0; j < c.n; ++j) {
        c.col_deg[j] = static_cast<uint8_t>(cols[j].size());
        for (size_t d = 0; d < cols[j].size(); ++d) c.col_slot[d * c.n + j] = cols[j][d];
    }
    return c;
}

// ---------------------------------------------------------------- LDPC tables for the wave decoder
// Rows are re-ordered by decreasing degree so that the 64 rows one wavefront processes together in a
// "round" have (nearly) the same number of edges; position p = 64*round + lane handles check perm[p].
// The arithmetic is unaffected: variable sums still run in ascending ORIGINAL check order.
struct FastTables {
    int k = 0, m = 0, n_rounds = 0, n_col_rounds = 0, max_col_deg = 0;
    std::vector<uint16_t> perm;        // [m] position -> check index
    std::vector<uint8_t> row_ne;       // [m] information edges of the check at position p
    std::vector<uint16_t> row_var;     // [6][m] information variable of slot s at position p (0 if unused)
    std::vector<uint8_t> col_deg;      // [k]
    std::vector<uint16_t> col_slot;    // [max_col_deg][k] slot word index s*m + p, ascending check order
    uint8_t round_ne[8] = {0};         // max information edges per row round (wave-uniform loop bounds)
    uint8_t round_cd[16] = {0};        // max column degree per column round
};

inline FastTables build_fast_tables(const LdpcCode& c) {
    FastTables t;
    t.k = c.k; t.m = c.m;
    const int k = c.k, m = c.m;
    t.perm.resize(m);
    for (int i = 0; i < m; ++i) t.perm[i] = static_cast<uint16_t>(i);
    std::stable_sort(t.perm.begin(), t.perm.end(), [&](uint16_t a, uint16_t b) { return c.rows[a].size() > c.rows[b].size(); });
    std::vector<int> pos(m);
    for (int p = 0; p < m; ++p) pos[t.perm[p]] = p;
    t.row_ne.assign(m, 0);
    t.row_var.assign(static_cast<size_t>(6) * m, 0);
    t.n_rounds = (m + 63) / 64;
    for (int p = 0; p < m; ++p) {
        const auto& row = c.rows[t.perm[p]];
        int ne = static_cast<int>(row.size()) - 1;  // last edge is the identity column k+i
        t.row_ne[p] = static_cast<uint8_t>(ne);
        for (int s = 0; s < ne; ++s) t.row_var[static_cast<size_t>(s) * m + p] = static_cast<uint16_t>(row[s]);
        t.round_ne[p / 64] = std::max<uint8_t>(t.round_ne[p / 64], static_cast<uint8_t>(ne));
    }
    std::vector<std::vector<uint16_t>> cols(k);
    for (int i = 0; i < m; ++i) {  // ascending original check index
        const auto& row = c.rows[i];
        for (size_t s = 0; s + 1 < row.size(); ++s) cols[row[s]].push_back(static_cast<uint16_t>(s * m + pos[i]));
    }
    for (auto& v : cols) t.max_col_deg = std::max(t.max_col_deg, static_cast<int>(v.size()));
    t.col_deg.assign(k, 0);
    t.col_slot.assign(static_cast<size_t>(std::max(1, t.max_col_deg)) * k, 0);
    t.n_col_rounds = (k + 63) / 64;
    for (int j = 0; j < k; ++j) {
        t.col_deg[j] = static_cast<uint8_t>(cols[j].size());
        t.round_cd[j / 64] = std::max<uint8_t>(t.round_cd[j / 64], t.col_deg[j]);
        for (size_t d = 0; d < cols[j].size(); ++d) t.col_slot[d * k + j] = cols[j][d];
    }
    return t;
}

// ---------------------------------------------------------------- RX gather (both de-interleavers folded)
inline int channel_interleaver_step(int n, int total) {  // ldpc_decoder.cpp:552-577
    auto gcd = [](int a, int b) { while (b) { int t = b; b = a % b; a = t; } return a; };
    int target = n * 3;
    if (target >= total) target = total / 2;
    for (int s = target; s < total; ++s) if (gcd(s, total) == 1) return s;
    for (int s = n + 1; s < total; ++s) if (gcd(s, total) == 1) return s;
    return n + 1;
}
// table[cw*648 + i] = position in the 2592 interleaved soft bits of decoder input i of codeword cw
// (frame_interleaver.cpp:37-45 inverted, then ChannelInterleaver::deinterleave ldpc_decoder.cpp:617-625)
inline std::vector<uint16_t> build_rx_gather(int bps, bool use_channel) {
    std::vector<uint16_t> t(4 * kCwBits);
    int step = use_channel ? channel_interleaver_step(bps, kCwBits) : 1;
    for (int cw = 0; cw < 4; ++cw)
        for (int i = 0; i < kCwBits; ++i) {
            int bit = use_channel ? static_cast<int>((static_cast<long>(i) * step) % kCwBits) : i;
            t[cw * kCwBits + i] = static_cast<uint16_t>(bit * 4 + (cw + bit) % 4);
        }
    return t;
}

}  // namespace ria
