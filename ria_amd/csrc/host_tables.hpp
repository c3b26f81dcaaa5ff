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

// ---------------------------------------------------------------- register-resident decoder tables
// Layout the wave-per-codeword decoder (ldpc_fast.hip.h) works on.  Rows are sorted by decreasing number
// of information edges and cut into rounds of 64 rows of EQUAL degree (a new round starts whenever the
// degree changes, the tail lanes of a round stay idle); information columns are sorted by decreasing
// degree, 64 per round.  The arithmetic is unaffected: variable sums still run in ascending ORIGINAL
// check order.  All addresses are BYTE offsets inside the wave's LDS region:
//   c2v word of (round r, slot s, lane l) = 64*(row_off[r] + s) + l
//   total of the column at sorted position q = tot_word + q;  64 words of +0.0f at zero_word.
struct CoreTables {
    int k = 0, m = 0;
    std::vector<int> ne;               // information edges per row round
    std::vector<int> dv;               // max degree per column round
    int ts = 0, td = 0, tot_word = 0, zero_word = 0;
    std::vector<uint16_t> row_addr;    // [ts][64]
    std::vector<uint16_t> col_addr;    // [td][64]
    std::vector<uint16_t> check_at;    // [64*NR]  0xFFFF idle
    std::vector<uint16_t> col_at;      // [64*NC]  0xFFFF idle
    std::vector<uint16_t> col_pos;     // [k]
};

inline CoreTables build_core_tables(const LdpcCode& c) {
    CoreTables t;
    t.k = c.k; t.m = c.m;
    const int k = c.k, m = c.m;
    auto info_deg = [&](int i) { return static_cast<int>(c.rows[i].size()) - 1; };  // last edge: identity column k+i
    std::vector<uint16_t> rows(m);
    for (int i = 0; i < m; ++i) rows[i] = static_cast<uint16_t>(i);
    std::stable_sort(rows.begin(), rows.end(), [&](uint16_t a, uint16_t b) { return info_deg(a) > info_deg(b); });
    std::vector<int> pos(m);   // check -> position 64*round + lane
    for (int i = 0; i < m;) {
        const int d = info_deg(rows[i]);
        int j = i;
        while (j < m && info_deg(rows[j]) == d) ++j;
        for (int b = i; b < j; b += 64) {
            const int r = static_cast<int>(t.ne.size());
            t.ne.push_back(d);
            t.check_at.resize(static_cast<size_t>(64) * (r + 1), 0xFFFF);
            for (int l = 0; l < 64 && b + l < j; ++l) { t.check_at[64 * r + l] = rows[b + l]; pos[rows[b + l]] = 64 * r + l; }
        }
        i = j;
    }
    const int NR = static_cast<int>(t.ne.size());
    std::vector<int> row_off(NR + 1, 0);
    for (int r = 0; r < NR; ++r) row_off[r + 1] = row_off[r] + t.ne[r];
    t.ts = row_off[NR];
    // columns: c2v word list in ascending original check order
    std::vector<std::vector<uint16_t>> cols(k);
    for (int i = 0; i < m; ++i) {
        const auto& row = c.rows[i];
        const int r = pos[i] / 64, l = pos[i] % 64;
        for (size_t s = 0; s + 1 < row.size(); ++s) cols[row[s]].push_back(static_cast<uint16_t>(64 * (row_off[r] + static_cast<int>(s)) + l));
    }
    std::vector<uint16_t> order(k);
    for (int j = 0; j < k; ++j) order[j] = static_cast<uint16_t>(j);
    std::stable_sort(order.begin(), order.end(), [&](uint16_t a, uint16_t b) { return cols[a].size() > cols[b].size(); });
    const int NC = (k + 63) / 64;
    t.col_at.assign(static_cast<size_t>(64) * NC, 0xFFFF);
    t.col_pos.assign(k, 0);
    t.dv.assign(NC, 0);
    for (int q = 0; q < k; ++q) {
        t.col_at[q] = order[q];
        t.col_pos[order[q]] = static_cast<uint16_t>(q);
        t.dv[q / 64] = std::max(t.dv[q / 64], static_cast<int>(cols[order[q]].size()));
    }
    std::vector<int> col_off(NC + 1, 0);
    for (int r = 0; r < NC; ++r) col_off[r + 1] = col_off[r] + t.dv[r];
    t.td = col_off[NC];
    t.tot_word = 64 * t.ts;
    t.zero_word = t.tot_word + 64 * NC;
    const uint16_t zero_addr = static_cast<uint16_t>(4 * t.zero_word);
    t.row_addr.assign(static_cast<size_t>(64) * std::max(1, t.ts), zero_addr);
    for (int r = 0; r < NR; ++r)
        for (int l = 0; l < 64; ++l) {
            const uint16_t i = t.check_at[64 * r + l];
            if (i == 0xFFFF) continue;
            for (int s = 0; s < t.ne[r]; ++s)
                t.row_addr[static_cast<size_t>(64) * (row_off[r] + s) + l] = static_cast<uint16_t>(4 * (t.tot_word + t.col_pos[c.rows[i][s]]));
        }
    t.col_addr.assign(static_cast<size_t>(64) * std::max(1, t.td), zero_addr);
    for (int q = 0; q < k; ++q) {
        const auto& v = cols[order[q]];
        for (size_t d = 0; d < v.size(); ++d) t.col_addr[static_cast<size_t>(64) * (col_off[q / 64] + static_cast<int>(d)) + q % 64] = static_cast<uint16_t>(4 * v[d]);
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
