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
#include "devmath.h"

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

// ---------------------------------------------------------------- Schmidl-Cox preamble / LTS template (host, built once)
// FFT::inverse of the reference (fft.cpp:96-128): bit-reversal, radix-2 DIT with conjugated twiddles, 1/N scale.
inline void host_ifft1024(std::vector<float>& re, std::vector<float>& im) {
    const std::vector<float> tw = build_twiddles();
    const int size = kFFT;
    int j = 0;
    for (int i = 0; i < size - 1; ++i) {
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
        int k = size / 2;
        while (k <= j) { j -= k; k /= 2; }
        j += k;
    }
    for (int len = 2; len <= size; len *= 2) {
        const int half = len / 2, step = size / len;
        for (int i = 0; i < size; i += len)
            for (int k = 0; k < half; ++k) {
                const float wr = tw[2 * k * step], wi = -tw[2 * k * step + 1];
                const float br = re[i + k + half], bi = im[i + k + half];
                const float tr = wr * br - wi * bi, ti = wr * bi + wi * br;
                re[i + k + half] = re[i + k] - tr; im[i + k + half] = im[i + k] - ti;
                re[i + k] = re[i + k] + tr;        im[i + k] = im[i + k] + ti;
            }
    }
    const float scale = 1.0f / static_cast<float>(size);
    for (int i = 0; i < size; ++i) { re[i] *= scale; im[i] *= scale; }
}
// one OFDM training symbol with cyclic prefix, complex baseband; sts = even FFT bins of the data carriers only
// (modulator.cpp:298-329 createSchmidlCoxSTS) else the LTS (data carriers <- ZC-59 by data ordinal, pilots <- pilot
// sequence; modulator.cpp:217-270 / demodulator.cpp:108-128)
inline void build_training_symbol(const CarrierPlan& p, bool sts, std::vector<float>& re, std::vector<float>& im) {
    std::vector<float> fr(kFFT, 0.0f), fi(kFFT, 0.0f);
    for (int i = 0; i < p.n_data; ++i) {
        if (sts && (p.data_bin[i] % 2) != 0) continue;
        fr[p.data_bin[i]] = p.sync_re[i % kCarriers]; fi[p.data_bin[i]] = p.sync_im[i % kCarriers];
    }
    if (!sts) for (int i = 0; i < p.n_pilot; ++i) { fr[p.pilot_bin[i]] = p.pilot_seq[i]; fi[p.pilot_bin[i]] = 0.0f; }
    host_ifft1024(fr, fi);
    const int cp = kSym - kFFT;
    re.resize(kSym); im.resize(kSym);
    for (int i = 0; i < kSym; ++i) { const int k = (i < cp) ? kFFT - cp + i : i - cp; re[i] = fr[k]; im[i] = fi[k]; }
}
struct CoxTemplate { std::vector<float> tI, tQ; float energy_ref = 0.0f; };
// OFDMDemodulator::Impl::generateSequences (demodulator.cpp:108-141) + the template energy of refineLTSTiming
// (ofdm_sync.cpp:405-411)
inline CoxTemplate build_cox_template(const CarrierPlan& p) {
    CoxTemplate t;
    std::vector<float> re, im;
    build_training_symbol(p, false, re, im);
    const std::vector<float> nco = build_nco_table(kSym);
    t.tI.resize(kSym); t.tQ.resize(kSym);
    for (int i = 0; i < kSym; ++i) {
        const float c = nco[2 * i], s = nco[2 * i + 1];
        t.tI[i] = re[i] * c - im[i] * s;
        t.tQ[i] = re[i] * s + im[i] * c;
    }
    float e = 0.0f;
    for (int i = 0; i < kSym; ++i) { e += t.tI[i] * t.tI[i]; e += t.tQ[i] * t.tQ[i]; }
    t.energy_ref = e * 0.5f;
    return t;
}
// OFDMModulator::generatePreamble (modulator.cpp:479-532): one symbol of silence, the STS four times, the LTS twice;
// complexToReal (:272-282) runs the TX mixer once over the STS and once over the LTS (the copies repeat the samples).
inline std::vector<float> build_cox_preamble(const CarrierPlan& p, float output_scale = 40.0f) {
    std::vector<float> out(static_cast<size_t>(kSym), 0.0f);
    const std::vector<float> nco = build_nco_table(2 * kSym);
    for (int part = 0; part < 2; ++part) {
        std::vector<float> re, im, real(kSym);
        build_training_symbol(p, part == 0, re, im);
        for (int i = 0; i < kSym; ++i) {
            const float c = nco[2 * (part * kSym + i)], s = nco[2 * (part * kSym + i) + 1];
            real[i] = (re[i] * c - im[i] * s) * output_scale;
        }
        for (int r = 0; r < (part == 0 ? 4 : 2); ++r) out.insert(out.end(), real.begin(), real.end());
    }
    return out;
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
// of information edges and cut into rounds of 64; round r is unrolled for ne[r] = its largest degree, the
// slots s >= nm[r] (its smallest degree) are "mixed": lanes whose row is shorter are padded there (they
// read a zero word, their c2v word holds -FLT_MAX for good and their store goes to a dump word, so the
// padded edge is the neutral element of min/sign without a single extra instruction).  Information
// columns are sorted by decreasing degree, 64 per round, dv[r] = largest degree in the round; shorter
// columns add zero words.  The arithmetic is unaffected: variable sums still run in ascending ORIGINAL
// check order.  All addresses are BYTE offsets inside the wave's LDS region:
//   c2v word of (round r, slot s, lane l) = 64*(row_off[r] + s) + l
//   total of the column at position q     = tot_word + q
//   zero words (read-only +0.0f)          = zero_word + lane,   dump words = dump_word + lane.
// Which lane a row / column sits on inside its round, and which slot an edge takes inside its row, does
// not change any result (min, xor and the per-column sums do not see it) but decides the LDS bank
// conflicts of the two gathers (bank = word mod 32, conflicts are per half-wave): optimise_banks()
// anneals those three choices; the decode kernels are LDS-bound and VALU-bound at once, so a conflict
// is paid in full.
struct CoreTables {
    int k = 0, m = 0;
    std::vector<int> ne, nm;           // per row round: slots unrolled, smallest row degree
    std::vector<int> dv;               // max degree per column round
    int ts = 0, td = 0, n_mixed = 0, tot_word = 0, zero_word = 0, dump_word = 0, big_word = 0;
    std::vector<uint16_t> row_addr;    // [ts][64]
    std::vector<uint16_t> col_addr;    // [td][64]
    std::vector<uint16_t> check_at;    // [64*NR]  0xFFFF idle
    std::vector<uint16_t> col_at;      // [64*NC]  0xFFFF idle
    std::vector<uint16_t> col_pos;     // [k]
    int conflict_cost_before = 0, conflict_cost_after = 0, conflict_floor = 0;   // sum over half-wave gathers of the worst bank multiplicity
};

struct CoreLayout {
    int k = 0, m = 0, NR = 0, NC = 0;
    std::vector<int> ne, nm, dv, row_off, col_off;
    std::vector<int> check_at;                    // [64*NR] check or -1
    std::vector<int> row_pos;                     // [m]
    std::vector<std::vector<int>> row_cols;       // [m] information columns in slot order
    std::vector<std::vector<int>> row_d;          // [m] for each slot: index of this edge in its column's ascending-check list
    std::vector<int> col_at;                      // [64*NC] column or -1
    std::vector<int> col_pos;                     // [k]
    std::vector<std::vector<int>> col_checks;     // [k] ascending check order
};

namespace bankopt {
// one half-wave gather: LDS passes = worst bank multiplicity (distinct words per bank); word < 0: a padded slot or an idle
// lane, which reads a constant word (+0.0 on the column side, the "big" total on the row side) of which there is a copy in
// every bank: the table builder points all of them at the copy in a bank the real reads of this gather leave free (one
// address: a broadcast), so they cost nothing.  Returns kPassWeight * passes + sum of squared multiplicities: the second
// term only breaks ties, it gives the annealer a slope on the plateaus of the max.
constexpr int kPassWeight = 64;
inline int half_cost(const int* word, int lane0) {
    (void)lane0;
    int cnt[32] = {0};
    int seen[32][8];
    int worst = 1;
    for (int l = 0; l < 32; ++l) {
        if (word[l] < 0) continue;
        const int w = word[l];
        const int b = w & 31;
        bool dup = false;
        for (int i = 0; i < cnt[b] && i < 8; ++i) dup = dup || seen[b][i] == w;
        if (dup) continue;
        if (cnt[b] < 8) seen[b][cnt[b]] = w;
        ++cnt[b];
        if (cnt[b] > worst) worst = cnt[b];
    }
    int sq = 0;
    for (int b = 0; b < 32; ++b) sq += cnt[b] * cnt[b];
    return kPassWeight * worst + sq;
}
}  // namespace bankopt

struct BankOptimiser {
    CoreLayout& L;
    std::vector<int> g1, g2;   // cost per half-wave gather: g1[(row_off[r]+s)*2 + h], g2[(col_off[cr]+d)*2 + h]
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    explicit BankOptimiser(CoreLayout& l) : L(l) {}
    uint32_t next() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return static_cast<uint32_t>(rng >> 33); }
    int eval_g1(int r, int s, int h) const {
        int word[32];
        for (int l = 0; l < 32; ++l) {
            const int i = L.check_at[64 * r + 32 * h + l];
            word[l] = (i >= 0 && s < static_cast<int>(L.row_cols[i].size())) ? L.col_pos[L.row_cols[i][s]] : -1;
        }
        return bankopt::half_cost(word, 32 * h);
    }
    int eval_g2(int cr, int d, int h) const {
        int word[32];
        for (int l = 0; l < 32; ++l) {
            const int c = L.col_at[64 * cr + 32 * h + l];
            word[l] = -1;
            if (c >= 0 && d < static_cast<int>(L.col_checks[c].size())) {
                const int i = L.col_checks[c][d];
                int s = 0;
                while (L.row_cols[i][s] != c) ++s;
                word[l] = 64 * (L.row_off[L.row_pos[i] / 64] + s) + L.row_pos[i] % 64;
            }
        }
        return bankopt::half_cost(word, 32 * h);
    }
    int total() {
        g1.assign(static_cast<size_t>(2) * L.row_off[L.NR], 1);
        g2.assign(static_cast<size_t>(2) * std::max(1, L.col_off[L.NC]), 1);
        int t = 0;
        for (int r = 0; r < L.NR; ++r) for (int s = 0; s < L.ne[r]; ++s) for (int h = 0; h < 2; ++h) t += g1[(L.row_off[r] + s) * 2 + h] = eval_g1(r, s, h);
        for (int cr = 0; cr < L.NC; ++cr) for (int d = 0; d < L.dv[cr]; ++d) for (int h = 0; h < 2; ++h) t += g2[(L.col_off[cr] + d) * 2 + h] = eval_g2(cr, d, h);
        return t;
    }
    int floor_cost() const { return 2 * (L.row_off[L.NR] + L.col_off[L.NC]); }   // in passes
    int passes() const {
        int t = 0;
        for (int v : g1) t += v / bankopt::kPassWeight;
        for (int v : g2) t += v / bankopt::kPassWeight;
        return t;
    }
    // affected gather lists (ids: g1 -> id, g2 -> id + G1N)
    void touch_row(int i, std::vector<int>& out) const {
        if (i < 0) return;
        const int r = L.row_pos[i] / 64, h = (L.row_pos[i] % 64) / 32;
        for (int s = 0; s < L.ne[r]; ++s) out.push_back((L.row_off[r] + s) * 2 + h);
        for (size_t s = 0; s < L.row_cols[i].size(); ++s) {
            const int c = L.row_cols[i][s], q = L.col_pos[c];
            out.push_back(static_cast<int>(g1.size()) + (L.col_off[q / 64] + L.row_d[i][s]) * 2 + (q % 64) / 32);
        }
    }
    void touch_col_pos(int q, std::vector<int>& out) const {
        const int cr = q / 64, h = (q % 64) / 32;
        for (int d = 0; d < L.dv[cr]; ++d) out.push_back(static_cast<int>(g1.size()) + (L.col_off[cr] + d) * 2 + h);
        const int c = L.col_at[q];
        if (c < 0) return;
        for (int i : L.col_checks[c]) {
            int s = 0;
            while (L.row_cols[i][s] != c) ++s;
            const int p = L.row_pos[i];
            out.push_back((L.row_off[p / 64] + s) * 2 + (p % 64) / 32);
        }
    }
    int eval_id(int id) const {
        const int G1N = static_cast<int>(g1.size());
        if (id < G1N) {
            const int slot = id / 2, h = id % 2;
            int r = 0;
            while (L.row_off[r + 1] <= slot) ++r;
            return eval_g1(r, slot - L.row_off[r], h);
        }
        const int slot = (id - G1N) / 2, h = (id - G1N) % 2;
        int cr = 0;
        while (L.col_off[cr + 1] <= slot) ++cr;
        return eval_g2(cr, slot - L.col_off[cr], h);
    }
    int& stored(int id) { const int G1N = static_cast<int>(g1.size()); return id < G1N ? g1[id] : g2[id - G1N]; }

    void run(int moves) {
        int cur = total();
        const int G1N = static_cast<int>(g1.size());
        (void)G1N;
        std::vector<int> ids, newc;
        for (int it = 0; it < moves; ++it) {
            const uint32_t kind = next() % 10;
            ids.clear();
            // --- propose
            int a0 = -1, a1 = -1, i0 = -1, s0 = 0, s1 = 0;
            if (kind < 4) {            // swap the lanes of two rows (or a row and an idle lane) of one round
                const int r = static_cast<int>(next() % L.NR);
                a0 = 64 * r + static_cast<int>(next() % 64); a1 = 64 * r + static_cast<int>(next() % 64);
                if (a0 == a1 || (L.check_at[a0] < 0 && L.check_at[a1] < 0)) continue;
                touch_row(L.check_at[a0], ids); touch_row(L.check_at[a1], ids);
                auto apply = [&]() {
                    std::swap(L.check_at[a0], L.check_at[a1]);
                    if (L.check_at[a0] >= 0) L.row_pos[L.check_at[a0]] = a0;
                    if (L.check_at[a1] >= 0) L.row_pos[L.check_at[a1]] = a1;
                };
                apply();
                touch_row(L.check_at[a0], ids); touch_row(L.check_at[a1], ids);
                if (!accept(ids, newc, cur, it, moves)) apply();
            } else if (kind < 8) {     // swap the positions of two columns (degree must fit the round)
                a0 = static_cast<int>(next() % (64 * L.NC)); a1 = static_cast<int>(next() % (64 * L.NC));
                const int c0 = L.col_at[a0], c1 = L.col_at[a1];
                if (a0 == a1 || (c0 < 0 && c1 < 0)) continue;
                const int d0 = c0 < 0 ? 0 : static_cast<int>(L.col_checks[c0].size()), d1 = c1 < 0 ? 0 : static_cast<int>(L.col_checks[c1].size());
                if (d0 > L.dv[a1 / 64] || d1 > L.dv[a0 / 64]) continue;
                touch_col_pos(a0, ids); touch_col_pos(a1, ids);
                auto apply = [&]() {
                    std::swap(L.col_at[a0], L.col_at[a1]);
                    if (L.col_at[a0] >= 0) L.col_pos[L.col_at[a0]] = a0;
                    if (L.col_at[a1] >= 0) L.col_pos[L.col_at[a1]] = a1;
                };
                apply();
                touch_col_pos(a0, ids); touch_col_pos(a1, ids);
                if (!accept(ids, newc, cur, it, moves)) apply();
            } else {                   // swap two slots of one row
                i0 = static_cast<int>(next() % L.m);
                const int deg = static_cast<int>(L.row_cols[i0].size());
                if (deg < 2) continue;
                s0 = static_cast<int>(next() % deg); s1 = static_cast<int>(next() % deg);
                if (s0 == s1) continue;
                const int r = L.row_pos[i0] / 64, h = (L.row_pos[i0] % 64) / 32;
                ids.push_back((L.row_off[r] + s0) * 2 + h); ids.push_back((L.row_off[r] + s1) * 2 + h);
                auto apply = [&]() { std::swap(L.row_cols[i0][s0], L.row_cols[i0][s1]); std::swap(L.row_d[i0][s0], L.row_d[i0][s1]); };
                apply();
                if (!accept(ids, newc, cur, it, moves)) apply();
            }
        }
    }
    // evaluates the touched gathers in the (already modified) layout; keeps the move if it does not
    // raise the cost (plus a little early uphill tolerance).  Updates `cur` and the stored costs on accept.
    bool accept(std::vector<int>& ids, std::vector<int>& newc, int& cur, int it, int moves) {
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        newc.resize(ids.size());
        int before = 0, after = 0;
        for (size_t q = 0; q < ids.size(); ++q) { before += stored(ids[q]); after += newc[q] = eval_id(ids[q]); }
        const int delta = after - before;
        const bool early = it < moves / 2;
        const bool ok = delta <= 0 || (early && delta <= 4 && next() % 16 == 0);
        if (!ok) return false;
        for (size_t q = 0; q < ids.size(); ++q) stored(ids[q]) = newc[q];
        cur += delta;
        return true;
    }
};

inline CoreTables build_core_tables(const LdpcCode& c, int optimiser_moves = 120000) {
    CoreLayout L;
    L.k = c.k; L.m = c.m;
    const int k = c.k, m = c.m;
    auto info_deg = [&](int i) { return static_cast<int>(c.rows[i].size()) - 1; };  // last edge: identity column k+i
    std::vector<int> rows(m);
    for (int i = 0; i < m; ++i) rows[i] = i;
    std::stable_sort(rows.begin(), rows.end(), [&](int a, int b) { return info_deg(a) > info_deg(b); });
    L.NR = (m + 63) / 64;
    L.check_at.assign(static_cast<size_t>(64) * L.NR, -1);
    L.row_pos.assign(m, 0);
    L.ne.assign(L.NR, 0); L.nm.assign(L.NR, 99);
    for (int p = 0; p < m; ++p) {
        L.check_at[p] = rows[p]; L.row_pos[rows[p]] = p;
        L.ne[p / 64] = std::max(L.ne[p / 64], info_deg(rows[p]));
        L.nm[p / 64] = std::min(L.nm[p / 64], info_deg(rows[p]));
    }
    L.row_off.assign(L.NR + 1, 0);
    for (int r = 0; r < L.NR; ++r) L.row_off[r + 1] = L.row_off[r] + L.ne[r];
    L.row_cols.resize(m); L.row_d.resize(m); L.col_checks.resize(k);
    for (int i = 0; i < m; ++i) {   // ascending original check index
        L.row_cols[i].assign(c.rows[i].begin(), c.rows[i].end() - 1);
        L.row_d[i].resize(L.row_cols[i].size());
        for (size_t s = 0; s < L.row_cols[i].size(); ++s) {
            L.row_d[i][s] = static_cast<int>(L.col_checks[L.row_cols[i][s]].size());
            L.col_checks[L.row_cols[i][s]].push_back(i);
        }
    }
    std::vector<int> order(k);
    for (int j = 0; j < k; ++j) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return L.col_checks[a].size() > L.col_checks[b].size(); });
    L.NC = (k + 63) / 64;
    L.col_at.assign(static_cast<size_t>(64) * L.NC, -1);
    L.col_pos.assign(k, 0);
    L.dv.assign(L.NC, 0);
    for (int q = 0; q < k; ++q) {
        L.col_at[q] = order[q]; L.col_pos[order[q]] = q;
        L.dv[q / 64] = std::max(L.dv[q / 64], static_cast<int>(L.col_checks[order[q]].size()));
    }
    L.col_off.assign(L.NC + 1, 0);
    for (int r = 0; r < L.NC; ++r) L.col_off[r + 1] = L.col_off[r] + L.dv[r];

    CoreTables t;
    t.k = k; t.m = m;
    {
        BankOptimiser opt(L);
        (void)opt.total();
        t.conflict_cost_before = opt.passes();
        t.conflict_floor = opt.floor_cost();
        if (optimiser_moves > 0) opt.run(optimiser_moves);
        (void)opt.total();
        t.conflict_cost_after = opt.passes();
    }
    t.ne = L.ne; t.nm = L.nm; t.dv = L.dv;
    t.ts = L.row_off[L.NR]; t.td = L.col_off[L.NC];
    for (int r = 0; r < L.NR; ++r) t.n_mixed += L.ne[r] - L.nm[r];
    t.tot_word = 64 * t.ts;
    t.zero_word = t.tot_word + 64 * L.NC;
    t.dump_word = t.zero_word + 64;
    t.big_word = t.dump_word + 64;
    t.check_at.assign(L.check_at.size(), 0xFFFF);
    for (size_t p = 0; p < L.check_at.size(); ++p) if (L.check_at[p] >= 0) t.check_at[p] = static_cast<uint16_t>(L.check_at[p]);
    t.col_at.assign(L.col_at.size(), 0xFFFF);
    for (size_t q = 0; q < L.col_at.size(); ++q) if (L.col_at[q] >= 0) t.col_at[q] = static_cast<uint16_t>(L.col_at[q]);
    t.col_pos.resize(k);
    for (int j = 0; j < k; ++j) t.col_pos[j] = static_cast<uint16_t>(L.col_pos[j]);
    // Padded slots and idle lanes read a constant word; every one of the 64 big / zero words holds the same value, so the lanes
    // of a half-wave gather all take the copy in a bank its real reads leave free (-1 below, resolved per half wave)
    auto resolve_padding = [](std::vector<int>& w64, int const_word) {
        for (int h = 0; h < 2; ++h) {
            bool used[32] = {false};
            int real = 0;
            for (int l = 32 * h; l < 32 * h + 32; ++l) if (w64[l] >= 0) { used[w64[l] & 31] = true; ++real; }
            int b = 0;
            while (b < 31 && used[b]) ++b;                 // fewer than 32 real reads leave a bank free
            (void)real;
            for (int l = 32 * h; l < 32 * h + 32; ++l) if (w64[l] < 0) w64[l] = const_word + b;
        }
    };
    t.row_addr.assign(static_cast<size_t>(64) * std::max(1, t.ts), 0);
    for (int r = 0; r < L.NR; ++r)
        for (int s = 0; s < L.ne[r]; ++s) {
            std::vector<int> w64(64, -1);                  // padded slot (or idle lane): a big word (ldpc_fast.hip.h: kPadTotal)
            for (int l = 0; l < 64; ++l) {
                const int i = L.check_at[64 * r + l];
                if (i >= 0 && s < static_cast<int>(L.row_cols[i].size())) w64[l] = t.tot_word + L.col_pos[L.row_cols[i][s]];
            }
            resolve_padding(w64, t.big_word);
            for (int l = 0; l < 64; ++l) t.row_addr[static_cast<size_t>(64) * (L.row_off[r] + s) + l] = static_cast<uint16_t>(4 * w64[l]);
        }
    t.col_addr.assign(static_cast<size_t>(64) * std::max(1, t.td), 0);
    for (int cr = 0; cr < L.NC; ++cr)
        for (int d = 0; d < L.dv[cr]; ++d) {
            std::vector<int> w64(64, -1);                  // padded slot (or idle lane): a zero word
            for (int l = 0; l < 64; ++l) {
                const int cc = L.col_at[64 * cr + l];
                if (cc >= 0 && d < static_cast<int>(L.col_checks[cc].size())) {
                    const int i = L.col_checks[cc][d];
                    int s = 0;
                    while (L.row_cols[i][s] != cc) ++s;
                    w64[l] = 64 * (L.row_off[L.row_pos[i] / 64] + s) + L.row_pos[i] % 64;
                }
            }
            resolve_padding(w64, t.zero_word);
            for (int l = 0; l < 64; ++l) t.col_addr[static_cast<size_t>(64) * (L.col_off[cr] + d) + l] = static_cast<uint16_t>(4 * w64[l]);
        }
    return t;
}

// ---- shipped layouts -------------------------------------------------------------------------------
// The annealer is deterministic but slow to converge (R1/2: 120 conflict passes per iteration after 2 M moves, 110 after
// 32 M, 107 after 128 M; floor 98), so the layouts of the six rates are annealed offline by tools/gen_core_layouts.cpp and shipped as data
// (core_layouts.inc).  A shipped layout is used only after validate_core_tables() has checked it against the H the
// library generates: every address table entry must be the one the decoder's indexing scheme implies.
struct SavedCoreTables { int rate; int n; const uint16_t* data; };
// flat form: ts td n_mixed NR NC tot_word zero_word dump_word big_word cost_before cost_after floor | ne[NR] nm[NR] dv[NC]
//            | check_at[64 NR] col_at[64 NC] col_pos[k] row_addr[64 ts] col_addr[64 td]
inline std::vector<uint16_t> flatten_core_tables(const CoreTables& t) {
    std::vector<uint16_t> o;
    const int NR = static_cast<int>(t.ne.size()), NC = static_cast<int>(t.dv.size());
    for (int v : {t.ts, t.td, t.n_mixed, NR, NC, t.tot_word, t.zero_word, t.dump_word, t.big_word, t.conflict_cost_before, t.conflict_cost_after, t.conflict_floor})
        o.push_back(static_cast<uint16_t>(v));
    for (int v : t.ne) o.push_back(static_cast<uint16_t>(v));
    for (int v : t.nm) o.push_back(static_cast<uint16_t>(v));
    for (int v : t.dv) o.push_back(static_cast<uint16_t>(v));
    for (const auto* a : {&t.check_at, &t.col_at, &t.col_pos, &t.row_addr, &t.col_addr}) o.insert(o.end(), a->begin(), a->end());
    return o;
}
inline bool unflatten_core_tables(const LdpcCode& c, const uint16_t* d, int n, CoreTables& t) {
    if (n < 12) return false;
    int p = 0;
    t = CoreTables{};
    t.k = c.k; t.m = c.m;
    t.ts = d[p++]; t.td = d[p++]; t.n_mixed = d[p++];
    const int NR = d[p++], NC = d[p++];
    t.tot_word = d[p++]; t.zero_word = d[p++]; t.dump_word = d[p++]; t.big_word = d[p++];
    t.conflict_cost_before = d[p++]; t.conflict_cost_after = d[p++]; t.conflict_floor = d[p++];
    const long need = 12L + 2L * NR + NC + 64L * NR + 64L * NC + c.k + 64L * std::max(1, t.ts) + 64L * std::max(1, t.td);
    if (NR != (c.m + 63) / 64 || NC != (c.k + 63) / 64 || need != n) return false;
    auto take = [&](std::vector<int>& v, int cnt) { v.assign(d + p, d + p + cnt); p += cnt; };
    auto take16 = [&](std::vector<uint16_t>& v, int cnt) { v.assign(d + p, d + p + cnt); p += cnt; };
    take(t.ne, NR); take(t.nm, NR); take(t.dv, NC);
    take16(t.check_at, 64 * NR); take16(t.col_at, 64 * NC); take16(t.col_pos, c.k);
    take16(t.row_addr, 64 * std::max(1, t.ts)); take16(t.col_addr, 64 * std::max(1, t.td));
    return true;
}
// true iff the tables describe a valid layout of code c: permutations, round structure, and every gather address
inline bool validate_core_tables(const LdpcCode& c, const CoreTables& t) {
    const int k = c.k, m = c.m, NR = (m + 63) / 64, NC = (k + 63) / 64;
    if (t.k != k || t.m != m || static_cast<int>(t.ne.size()) != NR || static_cast<int>(t.nm.size()) != NR || static_cast<int>(t.dv.size()) != NC) return false;
    std::vector<int> row_off(NR + 1, 0), col_off(NC + 1, 0);
    int mixed = 0;
    for (int r = 0; r < NR; ++r) { if (t.nm[r] > t.ne[r] || t.ne[r] < 0) return false; row_off[r + 1] = row_off[r] + t.ne[r]; mixed += t.ne[r] - t.nm[r]; }
    for (int r = 0; r < NC; ++r) col_off[r + 1] = col_off[r] + t.dv[r];
    if (t.ts != row_off[NR] || t.td != col_off[NC] || t.n_mixed != mixed) return false;
    if (t.tot_word != 64 * t.ts || t.zero_word != t.tot_word + 64 * NC || t.dump_word != t.zero_word + 64 || t.big_word != t.dump_word + 64) return false;
    if (static_cast<int>(t.check_at.size()) != 64 * NR || static_cast<int>(t.col_at.size()) != 64 * NC || static_cast<int>(t.col_pos.size()) != k) return false;
    if (static_cast<int>(t.row_addr.size()) != 64 * std::max(1, t.ts) || static_cast<int>(t.col_addr.size()) != 64 * std::max(1, t.td)) return false;
    std::vector<int> row_pos(m, -1), seen_col(k, 0);
    for (int p = 0; p < 64 * NR; ++p) if (t.check_at[p] != 0xFFFF) { const int i = t.check_at[p]; if (i >= m || row_pos[i] >= 0) return false; row_pos[i] = p; }
    for (int i = 0; i < m; ++i) if (row_pos[i] < 0) return false;
    for (int q = 0; q < 64 * NC; ++q) if (t.col_at[q] != 0xFFFF) { const int j = t.col_at[q]; if (j >= k || seen_col[j] || t.col_pos[j] != q) return false; seen_col[j] = 1; }
    for (int j = 0; j < k; ++j) if (!seen_col[j]) return false;
    // information edges of every check (the last entry of c.rows[i] is its identity column k+i) and, per column, its checks ascending
    std::vector<std::vector<int>> col_checks(k);
    for (int i = 0; i < m; ++i) for (size_t e = 0; e + 1 < c.rows[i].size(); ++e) col_checks[c.rows[i][e]].push_back(i);
    std::vector<std::vector<int>> slot_col(m);       // column read by slot s of check i, from row_addr
    for (int i = 0; i < m; ++i) {
        const int p = row_pos[i], r = p / 64, l = p % 64, deg = static_cast<int>(c.rows[i].size()) - 1;
        if (deg > t.ne[r] || deg < t.nm[r]) return false;
        std::vector<int> want(c.rows[i].begin(), c.rows[i].end() - 1), got;
        for (int s = 0; s < t.ne[r]; ++s) {
            const int a = t.row_addr[static_cast<size_t>(64) * (row_off[r] + s) + l];
            if (a % 4) return false;
            const int w = a / 4;
            if (s < deg) {
                const int q = w - t.tot_word;
                if (q < 0 || q >= 64 * NC || t.col_at[q] == 0xFFFF) return false;
                got.push_back(t.col_at[q]);
            } else if (w < t.big_word || w >= t.big_word + 64) return false;   // a padded slot reads any of the big words
        }
        slot_col[i] = got;
        std::sort(want.begin(), want.end()); std::sort(got.begin(), got.end());
        if (want != got) return false;
    }
    for (int r = 0; r < NR; ++r) for (int l = 0; l < 64; ++l) if (t.check_at[64 * r + l] == 0xFFFF)
        for (int s = 0; s < t.ne[r]; ++s) { const int a = t.row_addr[static_cast<size_t>(64) * (row_off[r] + s) + l]; if (a % 4 || a / 4 < t.big_word || a / 4 >= t.big_word + 64) return false; }
    for (int cr = 0; cr < NC; ++cr) for (int l = 0; l < 64; ++l) {
        const int cc = t.col_at[64 * cr + l] == 0xFFFF ? -1 : t.col_at[64 * cr + l];
        const int deg = cc < 0 ? 0 : static_cast<int>(col_checks[cc].size());
        if (deg > t.dv[cr]) return false;
        for (int d = 0; d < t.dv[cr]; ++d) {
            const int a = t.col_addr[static_cast<size_t>(64) * (col_off[cr] + d) + l];
            if (d < deg) {
                const int i = col_checks[cc][d];
                int s = -1;
                for (size_t q = 0; q < slot_col[i].size(); ++q) if (slot_col[i][q] == cc) s = static_cast<int>(q);
                if (s < 0) return false;
                if (a != 4 * (64 * (row_off[row_pos[i] / 64] + s) + row_pos[i] % 64)) return false;
            } else if (a % 4 || a / 4 < t.zero_word || a / 4 >= t.zero_word + 64) return false;   // a padded slot reads any of the zero words
        }
    }
    return true;
}

#if __has_include("core_layouts.inc")
#include "core_layouts.inc"
#define RIA_HAVE_SAVED_LAYOUTS 1
#endif
// the shipped layout of a rate, if there is one and it is a valid layout of the H this build generates
inline bool load_saved_core_tables(const LdpcCode& c, CoreTables& t) {
#ifdef RIA_HAVE_SAVED_LAYOUTS
    for (const SavedCoreTables& sv : kSavedLayouts)
        if (sv.rate == c.rate) return unflatten_core_tables(c, sv.data, sv.n, t) && validate_core_tables(c, t);
#endif
    (void)c; (void)t;
    return false;
}

// ---------------------------------------------------------------- Zadoff-Chu preamble (sync::ZCSync)
// zc_sync.hpp:420-436 (generateZC, N = 127 odd), :147-157 (8x linear interpolation), :133-190 (preamble).
// The float functions are devmath.h's host build (bit-identical to glibc, tests/test_devmath_host.py).
constexpr int kZcChips = 127, kZcUpsample = 8, kZcRepSamples = kZcChips * kZcUpsample, kZcGapSamples = 480;
inline void build_zc_reference(int root, std::vector<float>& re, std::vector<float>& im) {   // [1016] interpolated reference
    float zr[kZcChips], zi[kZcChips];
    for (int n = 0; n < kZcChips; ++n) {
        float phase = static_cast<float>(-3.14159265358979323846 * root * n * (n + 1) / kZcChips);
        zr[n] = cosf_glibc(phase);
        zi[n] = sinf_glibc(phase);
    }
    re.resize(kZcRepSamples); im.resize(kZcRepSamples);
    for (int i = 0; i < kZcRepSamples; ++i) {
        float chip_pos = static_cast<float>(i) / kZcUpsample;
        int c = static_cast<int>(chip_pos);
        float frac = chip_pos - c;
        if (c < kZcChips - 1) {
            float a = 1.0f - frac;
            re[i] = zr[c] * a + zr[c + 1] * frac;
            im[i] = zi[c] * a + zi[c + 1] * frac;
        } else { re[i] = zr[c]; im[i] = zi[c]; }
    }
}
inline std::vector<float> build_zc_preamble(int root) {
    std::vector<float> re, im, out(2 * kZcRepSamples + kZcGapSamples, 0.0f);
    build_zc_reference(root, re, im);
    for (int rep = 0; rep < 2; ++rep)
        for (int i = 0; i < kZcRepSamples; ++i) {
            int g = rep * kZcRepSamples + i;
            float t = static_cast<float>(g) / 48000.0f;
            float ph = static_cast<float>(static_cast<double>(2.0f) * 3.14159265358979323846 * static_cast<double>(1500.0f) * static_cast<double>(t));
            out[g] = re[i] * cosf_glibc(ph) - im[i] * sinf_glibc(ph);
        }
    float max_amp = 0.0f;
    for (int i = 0; i < 2 * kZcRepSamples; ++i) max_amp = std::max(max_amp, std::fabs(out[i]));
    if (max_amp > 0.0f) { float scale = 0.8f / max_amp; for (int i = 0; i < 2 * kZcRepSamples; ++i) out[i] *= scale; }
    return out;
}

// ---------------------------------------------------------------- dual chirp (sync::ChirpSync)
// chirp_sync.hpp:853-900 (phases, templates, energies), :61-108 (generate), fft.cpp:83-87 (twiddles)
constexpr int kChirpLen = 24000, kChirpGap = 4800, kChirpFft = 131072;
inline float chirp_phase(float t, bool down) {
    const float T = 500.0f / 1000.0f, k = (2700.0f - 300.0f) / T;
    const float inner = down ? (2700.0f * t - 0.5f * k * t * t) : (300.0f * t + 0.5f * k * t * t);
    return static_cast<float>(static_cast<double>(2.0f) * 3.14159265358979323846 * static_cast<double>(inner));
}
struct ChirpTables {
    std::vector<float> tmpl;      // [4][24000] up sin, up cos, down sin, down cos
    float energy[2] = {0, 0};
    std::vector<float> tw;        // [65536][2]
};
inline ChirpTables build_chirp_tables() {
    ChirpTables t;
    t.tmpl.resize(static_cast<size_t>(4) * kChirpLen);
    for (int d = 0; d < 2; ++d) {
        float e = 0.0f;
        for (int i = 0; i < kChirpLen; ++i) {
            const float ph = chirp_phase(static_cast<float>(i) / 48000.0f, d == 1);
            const float sn = sinf_glibc(ph);
            t.tmpl[static_cast<size_t>(2 * d) * kChirpLen + i] = sn;
            t.tmpl[static_cast<size_t>(2 * d + 1) * kChirpLen + i] = cosf_glibc(ph);
            e += sn * sn;
        }
        t.energy[d] = e;
    }
    t.tw.resize(static_cast<size_t>(kChirpFft));
    for (int k = 0; k < kChirpFft / 2; ++k) {
        const float angle = static_cast<float>(static_cast<double>(-2.0f) * 3.14159265358979323846 * static_cast<double>(k) / static_cast<double>(kChirpFft));
        t.tw[2 * k] = cosf_glibc(angle);
        t.tw[2 * k + 1] = sinf_glibc(angle);
    }
    return t;
}
inline std::vector<float> build_chirp_preamble() {
    std::vector<float> out(static_cast<size_t>(2 * kChirpLen + 2 * kChirpGap), 0.0f);
    for (int i = 0; i < kChirpLen; ++i) {
        const float t = static_cast<float>(i) / 48000.0f;
        out[i] = 0.5f * sinf_glibc(chirp_phase(t, false));
        out[kChirpLen + kChirpGap + i] = 0.5f * sinf_glibc(chirp_phase(t, true));
    }
    return out;
}

// ---------------------------------------------------------------- MC-DPSK (multi_carrier_dpsk.hpp)
constexpr int kMcdSps = 512, kMcdTrain = 8;
inline std::vector<float> mcdpsk_freqs(int nc) {   // getCarrierFreqs :66-78
    std::vector<float> f(nc);
    if (nc == 1) { f[0] = (500.0f + 2500.0f) / 2.0f; return f; }
    const float spacing = (2500.0f - 500.0f) / (nc - 1);
    for (int i = 0; i < nc; ++i) f[i] = 500.0f + i * spacing;
    return f;
}
inline float mcdpsk_phase_inc(float freq) { return static_cast<float>(static_cast<double>(2.0f) * 3.14159265358979323846 * static_cast<double>(freq) / static_cast<double>(48000.0f)); }
// mixer[c][i] = std::polar(1.0f, -phase_i), phase_0 = 0, phase_{i+1} = phase_i + phase_inc (:931-946)
inline std::vector<float> build_mcdpsk_mixer(int nc) {
    std::vector<float> m(static_cast<size_t>(nc) * kMcdSps * 2);
    const std::vector<float> fr = mcdpsk_freqs(nc);
    for (int c = 0; c < nc; ++c) {
        const float inc = mcdpsk_phase_inc(fr[c]);
        float phase = 0.0f;
        for (int i = 0; i < kMcdSps; ++i) {
            m[(static_cast<size_t>(c) * kMcdSps + i) * 2] = 1.0f * cosf_glibc(-phase);
            m[(static_cast<size_t>(c) * kMcdSps + i) * 2 + 1] = 1.0f * sinf_glibc(-phase);
            phase += inc;
        }
    }
    return m;
}
inline std::vector<float> build_hilbert(int taps);
inline std::vector<float> build_hilbert127() { return build_hilbert(127); }
inline std::vector<float> build_hilbert(int taps) {   // HilbertTransform ctor, filters.cpp:266-291 (taps odd)
    const int M = (taps - 1) / 2;
    std::vector<float> c(taps);
    for (int n = 0; n < taps; ++n) {
        const int k = n - M;
        if (k == 0) c[n] = 0;
        else if (k % 2 != 0) c[n] = static_cast<float>(static_cast<double>(2.0f) / (3.14159265358979323846 * k));
        else c[n] = 0;
        const float w = static_cast<float>(static_cast<double>(2.0f) * 3.14159265358979323846 * n / (taps - 1));
        c[n] *= 0.42f - 0.5f * cosf_glibc(w) + 0.08f * cosf_glibc(2.0f * w);
    }
    return c;
}
// tables of the device modulator: carrier[c][i] = (cos, sin)(i * phase_inc_c), train[sym][c] = (cos, sin)(c * sym * pi / 2)
inline void build_mcdpsk_mod_tables(int nc, std::vector<float>& carrier, std::vector<float>& train) {
    const std::vector<float> fr = mcdpsk_freqs(nc);
    carrier.assign(static_cast<size_t>(nc) * kMcdSps * 2, 0.0f);
    train.assign(static_cast<size_t>(kMcdTrain) * nc * 2, 0.0f);
    for (int c = 0; c < nc; ++c) {
        const float inc = mcdpsk_phase_inc(fr[c]);
        for (int i = 0; i < kMcdSps; ++i) {
            const float t = i * inc;
            carrier[(static_cast<size_t>(c) * kMcdSps + i) * 2] = 1.0f * cosf_glibc(t);
            carrier[(static_cast<size_t>(c) * kMcdSps + i) * 2 + 1] = 1.0f * sinf_glibc(t);
        }
        for (int sym = 0; sym < kMcdTrain; ++sym) {
            const float po = static_cast<float>(static_cast<double>(c * sym) * 3.14159265358979323846 / static_cast<double>(2.0f));
            train[(static_cast<size_t>(sym) * nc + c) * 2] = 1.0f * cosf_glibc(po);
            train[(static_cast<size_t>(sym) * nc + c) * 2 + 1] = 1.0f * sinf_glibc(po);
        }
    }
}
// training + reference + data audio (:141-281)
inline std::vector<float> build_mcdpsk_frame(int nc, int bps, int spreading, const uint8_t* data, int n_bytes) {
    const std::vector<float> fr = mcdpsk_freqs(nc);
    const int bits_per_sym = nc * bps, n_bits = n_bytes * 8;
    const int n_data_sym = (n_bits + bits_per_sym - 1) / bits_per_sym;
    std::vector<float> out(static_cast<size_t>(kMcdTrain + 1 + n_data_sym * spreading) * kMcdSps, 0.0f);
    auto cmulr = [](float ar, float ai, float br, float bi) { return ar * br - ai * bi; };   // real part of a*b
    std::vector<float> pr(nc, 1.0f), pi(nc, 0.0f);
    for (int sym = 0; sym < kMcdTrain; ++sym)
        for (int c = 0; c < nc; ++c) {
            const float po = static_cast<float>(static_cast<double>(c * sym) * 3.14159265358979323846 / static_cast<double>(2.0f));
            const float tr = 1.0f * cosf_glibc(po), ti = 1.0f * sinf_glibc(po);
            const float inc = mcdpsk_phase_inc(fr[c]);
            for (int i = 0; i < kMcdSps; ++i) {
                const float t = i * inc;
                out[static_cast<size_t>(sym) * kMcdSps + i] += cmulr(tr, ti, 1.0f * cosf_glibc(t), 1.0f * sinf_glibc(t)) / nc;
            }
        }
    float* ref = out.data() + static_cast<size_t>(kMcdTrain) * kMcdSps;
    for (int c = 0; c < nc; ++c) {
        const float inc = mcdpsk_phase_inc(fr[c]);
        pr[c] = 1.0f; pi[c] = 0.0f;
        for (int i = 0; i < kMcdSps; ++i) {
            const float t = i * inc;
            ref[i] += cmulr(1.0f, 0.0f, 1.0f * cosf_glibc(t), 1.0f * sinf_glibc(t)) / nc;
        }
    }
    float* dat = ref + kMcdSps;
    std::vector<float> one(kMcdSps);
    int bit_idx = 0;
    for (int ds = 0; ds < n_data_sym; ++ds) {
        std::fill(one.begin(), one.end(), 0.0f);
        for (int c = 0; c < nc; ++c) {
            int sb = 0;
            for (int b = 0; b < bps; ++b) {
                const int bit = (bit_idx < n_bits) ? (data[bit_idx >> 3] >> (7 - (bit_idx & 7))) & 1 : 0;
                ++bit_idx;
                sb = (sb << 1) | bit;
            }
            float pc;
            if (bps == 2) {
                const float ph[4] = {static_cast<float>(3.14159265358979323846 / 4), static_cast<float>(3 * 3.14159265358979323846 / 4),
                                     static_cast<float>(-3 * 3.14159265358979323846 / 4), static_cast<float>(-3.14159265358979323846 / 4)};
                pc = ph[sb];
            } else {
                pc = sb ? static_cast<float>(3.14159265358979323846) : 0.0f;
            }
            const float dr = 1.0f * cosf_glibc(pc), di = 1.0f * sinf_glibc(pc);
            float cr = pr[c] * dr - pi[c] * di, ci = pr[c] * di + pi[c] * dr;
            const float a = hypotf_glibc(cr, ci);
            cr /= a; ci /= a;
            pr[c] = cr; pi[c] = ci;
            const float inc = mcdpsk_phase_inc(fr[c]);
            for (int i = 0; i < kMcdSps; ++i) {
                const float t = i * inc;
                one[i] += cmulr(cr, ci, 1.0f * cosf_glibc(t), 1.0f * sinf_glibc(t)) / nc;
            }
        }
        for (int rep = 0; rep < spreading; ++rep) std::copy(one.begin(), one.end(), dat + static_cast<size_t>(ds * spreading + rep) * kMcdSps);
    }
    return out;
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
