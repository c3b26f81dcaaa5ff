// ria_amd/csrc/demod_kernels.hip.h — OFDM-CHIRP frame demodulator for gfx950: audio samples -> LLRs.
//
// Replaces OFDMChirpWaveform::process (src/waveform/ofdm_chirp_waveform.cpp:391-468) and
// OFDMDemodulator::processPresynced (src/ofdm/demodulator.cpp:1250-1414) with everything they call:
//   toBaseband/extractSymbol      src/ofdm/channel_equalizer.cpp:99-187   (+ NCO filters.cpp:228-238)
//   FFT::forward (radix-2 DIT)    src/dsp/fft.cpp:96-140
//   estimateChannelFromLTS        channel_equalizer.cpp:193-643
//   updateChannelEstimate         channel_equalizer.cpp:645-1043
//   equalize / hardDecision       channel_equalizer.cpp:1259-1451, :1168-1230
//   demodulateSymbol + demappers  src/ofdm/demodulator.cpp:208-508, src/ofdm/soft_demap.hpp
//
// Mapping: one 256-thread workgroup per frame.
//   phase F  the 4 wavefronts each take one OFDM symbol at a time: 1024-sample tile staged through LDS,
//            downconverted with a host-built NCO table (the reference mixer restarts at phase 0 every
//            frame), 1024-point FFT as three in-register radix-16/16/4 passes whose butterflies are the
//            reference's radix-2 DIT butterflies (same twiddle table, same operand order, no FMA
//            contraction) so every output bin is bit-identical; only the 59 used bins are kept.
//   phase E  wavefront 0 runs the sequential estimator/equaliser/demapper with lane = logical carrier;
//            every reduction the reference performs as a left-to-right float loop is reproduced as an
//            ordered readlane sum, so branch scalars (residual-CFO gate, CPE gate, fade erasure, DD
//            acceptance) see the same bits as on the CPU.
// Transcendentals come from devmath.h (bit-identical to glibc).  Compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ria_gpu.h"
#include "devmath.h"
#include "host_tables.hpp"

namespace ria {

struct DemodConst {
    int mod, coherent, n_data, n_pilot, bits_per_carrier, bits_per_symbol, n_data_symbols, n_llr;
    unsigned long long data_mask, pilot_mask;
    float ce_margin;
    int bin[64];          // logical carrier -> FFT bin
    int kk[64];           // signed bin index
    int is_pilot[64];
    int ord[64];          // data ordinal or pilot ordinal
    float tx_re[64], tx_im[64];   // LTS value transmitted on this carrier
    int lo_lane[64], hi_lane[64]; // logical index of the neighbouring pilots of a data carrier, or -1
    float alpha[64];
};

inline float ce_margin_for(int mod) {  // soft_demap.hpp:309-332
    switch (mod) {
        case RIA_MOD_D8PSK: return 1.1f;
        case RIA_MOD_QAM16: return 1.2f;
        case RIA_MOD_QAM32: return 1.5f;
        case RIA_MOD_QAM64: return 1.8f;
        case RIA_MOD_QAM256: return 2.5f;
        default: return 1.0f;
    }
}

inline DemodConst build_demod_const(const CarrierPlan& p, int mod, const ria_gpu_geometry& g) {
    DemodConst k{};
    k.mod = mod;
    k.coherent = is_coherent(mod) ? 1 : 0;
    k.n_data = p.n_data;
    k.n_pilot = p.n_pilot;
    k.bits_per_carrier = g.bits_per_carrier;
    k.bits_per_symbol = g.bits_per_symbol;
    k.n_data_symbols = g.n_data_symbols;
    k.n_llr = g.llrs_per_frame;
    k.ce_margin = ce_margin_for(mod);
    int d = 0, pi = 0;
    for (int l = 0; l < kCarriers; ++l) {
        k.bin[l] = p.all_bin[l];
        k.kk[l] = (p.all_bin[l] <= kFFT / 2) ? p.all_bin[l] : p.all_bin[l] - kFFT;
        k.is_pilot[l] = p.is_pilot[l];
        k.lo_lane[l] = k.hi_lane[l] = -1;
        if (p.is_pilot[l]) {
            k.ord[l] = pi;
            k.tx_re[l] = p.pilot_seq[pi];
            k.tx_im[l] = 0.0f;
            k.pilot_mask |= 1ull << l;
            ++pi;
        } else {
            k.ord[l] = d;
            k.tx_re[l] = p.sync_re[d % kCarriers];
            k.tx_im[l] = p.sync_im[d % kCarriers];
            k.data_mask |= 1ull << l;
            k.lo_lane[l] = p.interp_lo[d] >= 0 ? p.pilot_logical[p.interp_lo[d]] : -1;
            k.hi_lane[l] = p.interp_hi[d] >= 0 ? p.pilot_logical[p.interp_hi[d]] : -1;
            k.alpha[l] = p.interp_alpha[d];
            ++d;
        }
    }
    return k;
}

struct DemodArgs {
    const DemodConst* k;
    const float2* twiddle;   // [512]
    const float2* nco;       // [frame_samples] (cos, sin) of the RX mixer
    const float* samples;
    const uint64_t* offsets; // nullable
    const ria_frame_meta* meta;  // nullable
    int n_frames;
    float* llr_out;
    int llr_stride;
    ria_frame_status* status;  // nullable
    unsigned long long* dbg;   // nullable: per-frame phase stamps (s_memtime), diagnostic builds of the bench only
};

// ---------------------------------------------------------------- complex helpers (reference semantics)
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {  // naive, no FMA
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cdivc(float2 x, float2 y) {  // libgcc __divsc3: double, one rounding
    double a = x.x, b = x.y, c = y.x, d = y.y;
    double den = c * c + d * d;
    return make_float2(static_cast<float>((a * c + b * d) / den), static_cast<float>((b * c - a * d) / den));
}
__device__ __forceinline__ float cabs_(float2 a) { return hypotf_glibc(a.x, a.y); }
__device__ __forceinline__ float cnorm(float2 a) { return a.x * a.x + a.y * a.y; }
__device__ __forceinline__ float carg_(float2 a) { return atan2f_glibc(a.y, a.x); }
__device__ __forceinline__ float2 cexpj(float ph) { return make_float2(cosf_glibc(ph), sinf_glibc(ph)); }
__device__ __forceinline__ float2 conj_(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float maxf_(float a, float b) { return (a < b) ? b : a; }  // std::max
__device__ __forceinline__ float minf_(float a, float b) { return (b < a) ? b : a; }  // std::min

// Ordered (left-to-right) float sums, N at a time: the reference accumulates e.g. `cpe_sum += ...` in a
// for-loop over carriers, and float addition is not associative, so the same order is kept.  Members
// scatter their N terms to LDS rows compacted by ordinal (member o -> T[k][o]); lane k then adds row k
// front to back (contiguous 16-byte LDS reads, `count` dependent adds) while the other N-1 sums run
// in the neighbouring lanes; results are broadcast with readlane.
// Entries at ordinals >= count hold +0.0 (the scratch is zeroed at kernel start, members only ever write ordinals
// < count, and the one call with a larger count clears its tail afterwards): the accumulator starts at +0.0 and can
// never become -0.0 (x + y is -0.0 only for two negative zeros), so the trailing "+ 0.0f" adds are exact identities and
// the row is summed without any per-term test.
constexpr int kSumRow = 68;
constexpr int kPilotRow = 20;   // the pilot-ordered sums (<= 16 terms) have their own scratch: rows 20 floats apart
template <int N, int MAXC, int ROW = kSumRow, int CHUNK = 0>
__device__ __forceinline__ void ordered_sums(const float (&terms)[N], bool member, int ord, int count, float* T,
                                             int lane, float (&out)[N]) {
    static_assert(N <= 8 && MAXC % 4 == 0 && MAXC <= ROW - 4, "scratch is [8][ROW]");
    (void)count;
    if (member) {
#pragma unroll
        for (int k = 0; k < N; ++k) T[k * ROW + ord] = terms[k];
    }
    wave_sync();
    // rows are 68 (20) floats apart: a stride of 64 (16) would put the 8 rows read by one ds_read_b128 on the same banks
    const float4* row = reinterpret_cast<const float4*>(T + (lane & 7) * ROW);
    float acc = 0.0f;
    if constexpr (CHUNK > 0 && (MAXC / 4) > CHUNK) {
        // low-register form (the estimator kernel of the split pipeline wants many waves per SIMD): CHUNK float4 in flight
        // while the previous CHUNK are added
        constexpr int NQ = MAXC / 4;
        float4 cur[CHUNK], nxt[CHUNK];
#pragma unroll
        for (int q = 0; q < CHUNK; ++q) cur[q] = row[q];
#pragma unroll
        for (int base = 0; base < NQ; base += CHUNK) {
#pragma unroll
            for (int q = 0; q < CHUNK; ++q) if (base + CHUNK + q < NQ) nxt[q] = row[base + CHUNK + q];
#pragma unroll
            for (int q = 0; q < CHUNK; ++q) {
                if (base + q < NQ) {
                    acc = acc + cur[q].x;
                    acc = acc + cur[q].y;
                    acc = acc + cur[q].z;
                    acc = acc + cur[q].w;
                }
            }
            asm volatile("" ::: "memory");   // keep the compiler from hoisting every load to the top again
#pragma unroll
            for (int q = 0; q < CHUNK; ++q) cur[q] = nxt[q];
        }
    } else {
        float4 v[MAXC / 4];
#pragma unroll
        for (int q = 0; q < MAXC / 4; ++q) v[q] = row[q];
#pragma unroll
        for (int q = 0; q < MAXC / 4; ++q) {
            acc = acc + v[q].x;
            acc = acc + v[q].y;
            acc = acc + v[q].z;
            acc = acc + v[q].w;
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
        out[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), k));
    wave_sync();
}
__device__ __forceinline__ float lane_read(float v, int l) { return __shfl(v, l); }

// ---------------------------------------------------------------- 1024-point FFT by one wavefront
constexpr int kFftBufFloats2 = 1088;  // 1024 + 64 padding slots (index i lives at i + (i >> 4))

__device__ __forceinline__ void bfly(float2& a, float2& b, float2 w) {  // fft.cpp:113-117
    float2 t = cmul(w, b);
    b = make_float2(a.x - t.x, a.y - t.y);
    a = make_float2(a.x + t.x, a.y + t.y);
}

// The 15 twiddles of stages 1-4 are the same for every lane and every symbol: loaded once per frame into scalar registers
// (inside the symbol loop the compiler has to use vector loads for them: the kernel has stored to global memory by then).
struct FftUniformTw { float2 w[15]; };   // stage s (1..4), k in [0, 2^(s-1)): w[(1 << (s - 1)) - 1 + k] = tw[k << (10 - s)]
__device__ __forceinline__ FftUniformTw fft_load_uniform_tw(const float2* __restrict__ tw) {
    FftUniformTw u;
#pragma unroll
    for (int s = 1; s <= 4; ++s)
#pragma unroll
        for (int k = 0; k < (1 << (s - 1)); ++k) {
            const float2 v = tw[k << (10 - s)];
            u.w[(1 << (s - 1)) - 1 + k] = make_float2(u2f(__builtin_amdgcn_readfirstlane(f2u(v.x))), u2f(__builtin_amdgcn_readfirstlane(f2u(v.y))));
        }
    return u;
}

// buf[0..1023] holds the time samples in natural order on entry (plain layout).  On exit the 59 used
// bins are written to Yrow[logical carrier].
// The twiddles of stages 5-10 depend on the lane only (not on the symbol): a wave that transforms many symbols keeps them
// in registers.  ws[(1 << (s - 5)) - 1 + q] = tw[(a + 16 q) << (10 - s)] for s = 5..8, a = lane & 15; w9 / w10a / w10b per
// output half t (b = lane + 64 t, t in {0, 3}) come from the table (L1 hits after the first symbol).
struct FftLaneTw { float2 ws[15]; };
__device__ __forceinline__ FftLaneTw fft_load_lane_tw(const float2* __restrict__ tw, int lane) {
    FftLaneTw L;
    const int a = lane & 15;
#pragma unroll
    for (int s = 5; s <= 8; ++s)
#pragma unroll
        for (int q = 0; q < (1 << (s - 5)); ++q) L.ws[(1 << (s - 5)) - 1 + q] = tw[(a + 16 * q) << (10 - s)];
    return L;
}
// Input tile in the bank-friendly order of the split pipeline: sample i of the symbol lives at fft_in_slot(i).  A lane of the
// mixer holds samples 4 l .. 4 l + 3 of each quarter and stores them as two 16-byte pairs {e0, e2} and {e1, e3} whose
// addresses are contiguous over the lanes (no bank conflict: the natural order puts the lanes' pairs 32 bytes apart, 2-way),
// and pass 1's bit-reversed gather then reads 32 consecutive slots per half wave (natural order: every other slot, 2-way).
__device__ __forceinline__ constexpr int fft_in_slot(int i) { return 2 * ((i >> 2) & 63) + ((i >> 1) & 1) + 128 * (i & 1) + 256 * (i >> 8); }
template <bool kLaneTw = false, bool kSlotOrder = false>
__device__ inline void fft1024_wave(float2* buf, const float2* __restrict__ tw, const FftUniformTw& utw, float2* Yrow, int lane, const FftLaneTw* ltw = nullptr) {
    float2 x[16];
    // pass 1: bit-reversed gather, stages 1-4 on indices 16*lane + r
    {
        int rl = __brev(static_cast<unsigned>(lane)) >> 26;  // 6-bit reverse
        const int rs = 2 * (rl >> 2) + ((rl >> 1) & 1) + 128 * (rl & 1);   // fft_in_slot(rl + 64 r4) = rs + 32 (r4 & 3) + 256 (r4 >> 2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int r4 = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
            if constexpr (kSlotOrder) x[r] = buf[rs + 32 * (r4 & 3) + 256 * (r4 >> 2)];
            else x[r] = buf[rl + 64 * r4];
        }
    }
    wave_sync();
#pragma unroll
    for (int s = 1; s <= 4; ++s) {
        const int half = 1 << (s - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((r & half) == 0) {
                int k = r & (half - 1);
                bfly(x[r], x[r + half], utw.w[half - 1 + k]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[17 * lane + r] = x[r];
    wave_sync();
    // pass 2: stages 5-8 on indices a + 16*r2 + 256*hi
    {
        const int a = lane & 15, hi = lane >> 4;
        const int base = a + 272 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = buf[base + 17 * r];
#pragma unroll
        for (int s = 5; s <= 8; ++s) {
            const int hr = 1 << (s - 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((r & hr) == 0) {
                    int k = a + 16 * (r & (hr - 1));
                    if constexpr (kLaneTw) bfly(x[r], x[r + hr], ltw->ws[hr - 1 + (r & (hr - 1))]);
                    else bfly(x[r], x[r + hr], tw[k << (10 - s)]);
                }
            }
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[base + 17 * r] = x[r];
    }
    wave_sync();
    // pass 3: stages 9-10 on indices b + 256*q; only bases that feed a used bin (b in 1..30 -> q=0
    // output, b in 227..255 -> q=3 output) are computed.
#pragma unroll
    for (int t = 0; t < 4; t += 3) {
        int b = lane + 64 * t;
        int pa = b + (b >> 4);
        float2 y0 = buf[pa], y1 = buf[pa + 272], y2 = buf[pa + 544], y3 = buf[pa + 816];
        const float2 w9 = tw[b << 1];
        bfly(y0, y1, w9);
        bfly(y2, y3, w9);
        bfly(y0, y2, tw[b]);
        bfly(y1, y3, tw[b + 256]);
        if (t == 0) { if (lane >= 1 && lane <= 30) Yrow[28 + lane] = y0; }      // bins 1..30  -> logical 29..58
        else        { if (lane >= 35) Yrow[lane - 35] = y3; }                    // bins 995..1023 -> logical 0..28
    }
    wave_sync();
}

// ---------------------------------------------------------------- soft demappers (soft_demap.hpp)
__device__ __forceinline__ float clip_llr(float llr) {  // :22-29
    float c = maxf_(-20.0f, minf_(20.0f, llr));
    if (fabs_(c) < 0.01f) c = (c >= 0.0f) ? 0.01f : -0.01f;
    return c;
}

__device__ inline int demap_symbol(int mod, float2 sym, float2 prev, float nv, float* o) {
    float I = sym.x, Q = sym.y;
    switch (mod) {
        case RIA_MOD_BPSK: o[0] = clip_llr(fdiv(-2.0f * I, nv)); return 1;
        case RIA_MOD_QPSK: {
            float sc = fdiv(-2.0f * 0.7071067811865476f, nv);
            o[0] = clip_llr(I * sc); o[1] = clip_llr(Q * sc); return 2;
        }
        case RIA_MOD_QAM16: {
            float sc = fdiv(2.0f, nv);
            const float T = 0.6324555320336759f;
            o[0] = clip_llr(-sc * I); o[1] = clip_llr(sc * (fabs_(I) - T));
            o[2] = clip_llr(-sc * Q); o[3] = clip_llr(sc * (fabs_(Q) - T));
            return 4;
        }
        case RIA_MOD_QAM32: {
            const float IL[4] = {-3, -1, 1, 3}, QL[8] = {-7, -5, -3, -1, 1, 3, 5, 7};
            const int IG[4] = {0, 1, 3, 2}, QG[8] = {0, 1, 3, 2, 6, 7, 5, 4};
            const float S = 0.1961161351381840f;
            float sf = fdiv(2.0f, nv);
            float m0[5], m1[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) { m0[b] = 1e10f; m1[b] = 1e10f; }
            for (int qi = 0; qi < 8; ++qi)
                for (int ii = 0; ii < 4; ++ii) {
                    float dr = I - IL[ii] * S, di = Q - QL[qi] * S;
                    float d2 = dr * dr + di * di;
                    int bits = (QG[qi] << 2) | IG[ii];
#pragma unroll
                    for (int b = 0; b < 5; ++b) {
                        if (bits & (1 << (4 - b))) { if (d2 < m1[b]) m1[b] = d2; }
                        else { if (d2 < m0[b]) m0[b] = d2; }
                    }
                }
#pragma unroll
            for (int b = 0; b < 5; ++b) o[b] = clip_llr(sf * (m1[b] - m0[b]));
            return 5;
        }
        case RIA_MOD_QAM64: {
            float sc = fdiv(2.0f, nv);
            const float D2 = 0.3086067f, D4 = 0.6172134f;
            o[0] = clip_llr(-sc * I); o[1] = clip_llr(sc * (fabs_(I) - D4)); o[2] = clip_llr(sc * (fabs_(fabs_(I) - D4) - D2));
            o[3] = clip_llr(-sc * Q); o[4] = clip_llr(sc * (fabs_(Q) - D4)); o[5] = clip_llr(sc * (fabs_(fabs_(Q) - D4) - D2));
            return 6;
        }
        case RIA_MOD_QAM256: {
            float sc = fdiv(2.0f, nv);
            const float D2 = 0.1290994f, D4 = 0.2581989f, D8 = 0.5163978f;
            float v2[2] = {I, Q};
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float x = v2[a];
                o[4 * a + 0] = clip_llr(-sc * x);
                o[4 * a + 1] = clip_llr(sc * (fabs_(x) - D8));
                o[4 * a + 2] = clip_llr(sc * (fabs_(fabs_(x) - D8) - D4));
                o[4 * a + 3] = clip_llr(sc * (fabs_(fabs_(fabs_(x) - D8) - D4) - D2));
            }
            return 8;
        }
        case RIA_MOD_DBPSK: {
            float2 diff = cmul(sym, conj_(prev));
            float pd = atan2f_glibc(diff.y, diff.x);
            float sp = cabs_(sym) * cabs_(prev);
            if (sp < 1e-6f) { o[0] = 0.0f; return 1; }
            float dnv = 2.0f * nv;
            float conf = fdiv(2.0f * sp, dnv);
            o[0] = clip_llr(conf * cosf_glibc(pd));
            return 1;
        }
        case RIA_MOD_DQPSK: {
            float2 diff = cmul(sym, conj_(prev));
            float dI = diff.x, dQ = diff.y, dm = cabs_(diff);
            if (dm < 1e-6f) { o[0] = 0.0f; o[1] = 0.0f; return 2; }
            float dnv = 2.0f * nv;
            float sp = cabs_(sym) * cabs_(prev);
            float snr = fdiv(sp, dnv);
            float sc = 2.0f * fsqrt(snr);
            const float pi_f = 3.14159265358979f;
            float ph = atan2f_glibc(dQ, dI);
            o[0] = clip_llr(sc * sinf_glibc(ph + fdiv(pi_f, 4.0f)));
            o[1] = clip_llr(fdiv(sc * (fabs_(dI) - fabs_(dQ)), dm));
            return 2;
        }
        case RIA_MOD_D8PSK: {   // soft_demap.hpp:239-263
            float2 diff = cmul(sym, conj_(prev));
            float pd = atan2f_glibc(diff.y, diff.x);
            float sp = cabs_(sym) * cabs_(prev);
            if (sp < 1e-6f) { o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f; return 3; }
            float dnv = 2.0f * nv;
            float conf = fdiv(sp, dnv);
            o[0] = clip_llr(conf * sinf_glibc(pd));
            o[1] = clip_llr(conf * sinf_glibc(2.0f * pd));
            o[2] = clip_llr(conf * sinf_glibc(4.0f * pd));
            return 3;
        }
        default: return 0;
    }
}

__device__ __forceinline__ float2 hard_decision(float2 s, int mod) {  // channel_equalizer.cpp:1168-1230
    switch (mod) {
        case RIA_MOD_BPSK: return make_float2(s.x > 0 ? 1.0f : -1.0f, 0.0f);
        case RIA_MOD_QAM16: {
            auto sl = [](float x) { return (x < -0.4f) ? -0.9487f : (x < 0.0f) ? -0.3162f : (x < 0.4f) ? 0.3162f : 0.9487f; };
            return make_float2(sl(s.x), sl(s.y));
        }
        case RIA_MOD_QAM32: {
            const float d = 0.1961161351381840f;
            float x = s.x, y = s.y;
            float I = (x < -2 * d) ? -3 * d : (x < 0) ? -d : (x < 2 * d) ? d : 3 * d;
            float Q = (y < -6 * d) ? -7 * d : (y < -4 * d) ? -5 * d : (y < -2 * d) ? -3 * d : (y < 0) ? -d
                    : (y < 2 * d) ? d : (y < 4 * d) ? 3 * d : (y < 6 * d) ? 5 * d : 7 * d;
            return make_float2(I, Q);
        }
        case RIA_MOD_QAM64: {
            const float d = 0.1543f;
            auto sl = [d](float y) {
                return (y < -6 * d) ? -7 * d : (y < -4 * d) ? -5 * d : (y < -2 * d) ? -3 * d : (y < 0) ? -d
                     : (y < 2 * d) ? d : (y < 4 * d) ? 3 * d : (y < 6 * d) ? 5 * d : 7 * d;
            };
            return make_float2(sl(s.x), sl(s.y));
        }
        default: return make_float2(s.x > 0 ? 0.7071f : -0.7071f, s.y > 0 ? 0.7071f : -0.7071f);
    }
}

// ---------------------------------------------------------------- the frame kernel
constexpr int kDemodThreads = 64;   // ONE wavefront per frame: no workgroup barriers, no idle waves in phase E
// The frame is streamed symbol by symbol (FFT of symbol s, then the estimator step that consumes it), so the LDS
// holds one FFT tile (8.7 KB), THREE rows of bins (the two training symbols + the current one) and the scratch of the
// ordered sums: 13.1 KB per frame whatever the frame length -> 12 frames per CU (3 waves per SIMD).
struct DemodShared {
    float cfo, theta0;          // current CFO and correction phase at frame start
    int rerun, pad_;
    float th_end, pad2_[3];     // correction phase after the last sample of the frame
    float sub_theta[16];        // correction phase at every 72nd sample of the symbol being staged
    float sums[8 * kSumRow];    // ordered_sums scratch [8][kSumRow], data- and carrier-ordered sums (16-byte aligned)
    float psums[8 * kPilotRow]; // ordered_sums scratch of the pilot-ordered sums
};

// One step of the CFO correction phase (channel_equalizer.cpp:139-144): th += inc, wrapped with
// double-precision pi.  For a float th, "th > M_PI" <=> th >= 0x40490fdb (the float just above pi),
// so the common no-wrap step is one add and one compare; the wrap itself is done in double as written.
__device__ __forceinline__ float cfo_phase_step(float th, float inc) {
    th += inc;
    const float kPiUp = u2f(0x40490fdbu);
    if (fabs_(th) >= kPiUp) {
        if (th > 0.0f) th = static_cast<float>(static_cast<double>(th) - 2.0f * 3.14159265358979323846);
        else th = static_cast<float>(static_cast<double>(th) + 2.0f * 3.14159265358979323846);
    }
    return th;
}

// Eight steps at once: the wrap happens once per ~1000+ samples, so run the adds speculatively and
// check the largest magnitude once; fall back to single steps only for the group that wraps.
__device__ __forceinline__ float cfo_phase_step8(float th, float inc) {
    const float kPiUp = u2f(0x40490fdbu);
    float t1 = th + inc, t2 = t1 + inc, t3 = t2 + inc, t4 = t3 + inc;
    float t5 = t4 + inc, t6 = t5 + inc, t7 = t6 + inc, t8 = t7 + inc;
    float m = fmaxf(fmaxf(fmaxf(fabs_(t1), fabs_(t2)), fmaxf(fabs_(t3), fabs_(t4))),
                    fmaxf(fmaxf(fabs_(t5), fabs_(t6)), fmaxf(fabs_(t7), fabs_(t8))));
    if (m >= kPiUp) {
#pragma unroll 1
        for (int i = 0; i < 8; ++i) th = cfo_phase_step(th, inc);
        return th;
    }
    return t8;
}

// Symbol s of the frame: CFO correction phases (if any), downconversion, FFT; the 59 used bins go to Yrow.
// th_walk (meaningful on lane 0): correction phase at the first sample of this symbol on entry, of the next on exit.
// Sample prefetch: while the estimator works on symbol s the FFT tile is idle, so the 1024 samples of the NEXT symbol are
// fetched straight into it by LDS-DMA (global_load_lds_dwordx4: 4 wave-instructions, no VGPRs): floats [kRawOff, kRawOff +
// 1024) of the tile, behind the 1152 floats the CFO path uses for its phases.  pf_sym (wave-uniform) = the symbol whose
// samples are in flight / have landed there, -1 none; the staging step then reads the tile instead of HBM (a frame's
// sample loads were 21 % of the kernel's time: measured by letting every frame read cache-resident samples).
constexpr int kRawOff = 1152;
__device__ __forceinline__ void demod_prefetch_symbol(const float* __restrict__ x, int s, float2* buf, int lane) {
    const float* xs = x + s * kSym + kCP;
    float* raw = reinterpret_cast<float*>(buf) + kRawOff;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xs + 4 * lane + 256 * c),
                                         (__attribute__((address_space(3))) void*)(raw + 256 * c), 16, 0, 0);
}

__device__ __forceinline__ void demod_fft_symbol(const DemodArgs& A, const float* __restrict__ x, int s, int next_s, float2* buf, float2* Yrow,
                                                 DemodShared* sh, int lane, float& th_walk, int& pf_sym, const FftUniformTw& utw) {
    const float cfo = sh->cfo;
    const bool use_cfo = fabs_(cfo) > 0.01f;
    const float inc = static_cast<float>(-2.0f * 3.14159265358979323846 * static_cast<double>(cfo) / 48000.0);
    const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
    if (pf_sym >= 0) {   // a prefetch is in flight or has landed: it must be complete before the tile is read or rewritten
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
    }
    float th_reg[16];
    if (use_cfo) {
        // channel_equalizer.cpp:132-144: float phase recurrence over EVERY sample (CP included), wrapped
        // with double-precision pi.  Inherently serial: lane 0 walks the symbol once, dropping a
        // marker every 72 samples; 16 lanes then re-walk 72 samples each into the (still free) tile.
        float* thb = reinterpret_cast<float*>(buf);
        if (lane == 0) {
            float th = th_walk;
            for (int q = 0; q < 16; ++q) {
                sh->sub_theta[q] = th;
#pragma unroll 1
                for (int g = 0; g < 9; ++g) th = cfo_phase_step8(th, inc);
            }
            th_walk = th;
        }
        wave_sync();
        if (lane < 16) {
            float th = sh->sub_theta[lane];
            for (int i = 0; i < 72; ++i) { thb[72 * lane + i] = th; th = cfo_phase_step(th, inc); }
        }
        wave_sync();
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) th_reg[4 * c + e] = thb[kCP + 4 * lane + 256 * c + e];
        wave_sync();
    }
    // stage + downconvert 1024 samples (cyclic prefix dropped): 16 B per lane per load, coalesced; from the tile when the
    // previous symbol's step prefetched them (all 16 floats of the lane are read before the tile is written below)
    const float* xs = x + s * kSym + kCP;
    const float2* osc = A.nco + s * kSym + kCP;
    const bool from_tile = (pf_sym == s);
    float xa[16];
    if (from_tile) {
        const float* raw = reinterpret_cast<const float*>(buf) + kRawOff;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(raw + 4 * lane + 256 * c);
            xa[4 * c] = v.x; xa[4 * c + 1] = v.y; xa[4 * c + 2] = v.z; xa[4 * c + 3] = v.w;
        }
        wave_sync();
    }
    pf_sym = -1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int j = 4 * lane + 256 * c;
        float xv[4];
        if (from_tile) { xv[0] = xa[4 * c]; xv[1] = xa[4 * c + 1]; xv[2] = xa[4 * c + 2]; xv[3] = xa[4 * c + 3]; }
        else if (aligned) { float4 v = *reinterpret_cast<const float4*>(xs + j); xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w; }
        else { xv[0] = xs[j]; xv[1] = xs[j + 1]; xv[2] = xs[j + 2]; xv[3] = xs[j + 3]; }
        const float4 o01 = *reinterpret_cast<const float4*>(osc + j), o23 = *reinterpret_cast<const float4*>(osc + j + 2);
        const float2 ov[4] = {make_float2(o01.x, o01.y), make_float2(o01.z, o01.w), make_float2(o23.x, o23.y), make_float2(o23.z, o23.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float2 m = make_float2(xv[e] * ov[e].x, xv[e] * -ov[e].y);  // samples[i] * conj(osc)
            if (use_cfo) m = cmul(m, cexpj(th_reg[4 * c + e]));
            buf[j + e] = m;
        }
    }
    wave_sync();
    fft1024_wave(buf, A.twiddle, utw, Yrow, lane);
    if (aligned && next_s >= 0) { demod_prefetch_symbol(x, next_s, buf, lane); pf_sym = next_s; }
}

__global__ __launch_bounds__(kDemodThreads) __attribute__((amdgpu_waves_per_eu(3))) void demod_frames_kernel(DemodArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DemodConst& K = *A.k;
    const int lane = threadIdx.x;
    const int frame = blockIdx.x;
    const int n_sym = 2 + K.n_data_symbols;
    float2* tiles = reinterpret_cast<float2*>(smem);                       // 1088 float2
    float2* Y = tiles + kFftBufFloats2;                                    // [3][64]: training symbols 0, 1 and the current data symbol
    DemodShared* sh = reinterpret_cast<DemodShared*>(Y + 3 * 64);

    const float* x = A.samples + (A.offsets ? A.offsets[frame] : static_cast<uint64_t>(frame) * n_sym * kSym);
    if (threadIdx.x == 0) {
        float cfo = 0.0f;
        double init = 0.0;
        uint32_t fl = 0;
        if (A.meta) {
            cfo = A.meta[frame].cfo_hz;
            fl = A.meta[frame].flags;
            // ofdm_chirp_waveform.cpp:402-411
            float ip = static_cast<float>(-2.0f * 3.14159265358979323846 * static_cast<double>(cfo) *
                                          static_cast<double>(A.meta[frame].abs_position) / 48000.0);
            while (static_cast<double>(ip) > 3.14159265358979323846) ip = static_cast<float>(static_cast<double>(ip) - 2.0f * 3.14159265358979323846);
            while (static_cast<double>(ip) < -3.14159265358979323846) ip = static_cast<float>(static_cast<double>(ip) + 2.0f * 3.14159265358979323846);
            init = ip;
        }
        (void)fl;
        sh->cfo = cfo;
        sh->theta0 = static_cast<float>(init);
        sh->rerun = 0;
    }
    // ordered-sum scratch: +0.0 everywhere (see ordered_sums)
    for (int i = lane; i < 8 * kSumRow; i += 64) sh->sums[i] = 0.0f;
    for (int i = lane; i < 8 * kPilotRow; i += 64) sh->psums[i] = 0.0f;
    wave_sync();
    const bool negate_lts0 = A.meta && (A.meta[frame].flags & 1u);

    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (A.dbg) t0 = __builtin_readcyclecounter();
    t1 = t0;
    // per-lane estimator state (wave 0; lane = logical carrier)
    float2 H = make_float2(1.0f, 0.0f);
    float noise_var = 0.1f, snr_lin = 1.0f, fading = 0.0f, slope = 0.0f;
    int snr_count = 0;
    const bool is_car = lane < kCarriers;
    const bool is_pil = is_car && K.is_pilot[lane];
    const bool is_dat = is_car && !K.is_pilot[lane];
    const int my_ord = K.ord[lane];          // data ordinal (data carriers) / pilot ordinal (pilots)
    float* T = sh->sums;
    float* TP = sh->psums;
    float th_walk = 0.0f;
    int pf_sym = -1;
    const FftUniformTw utw = fft_load_uniform_tw(A.twiddle);
    const float2 txv = make_float2(K.tx_re[lane], K.tx_im[lane]);
    const float lsign = negate_lts0 ? -1.0f : 1.0f;
    float2 H0 = make_float2(0, 0), H1 = make_float2(0, 0);

    // ================= LTS channel estimate (channel_equalizer.cpp:193-643), possibly twice
    for (int pass = 0; pass < 2; ++pass) {
        // ---- the two training symbols through the FFT with the current CFO (pass 1: mixer.reset(), phase
        // restored to its value at training start, corrected CFO: channel_equalizer.cpp:337-344)
        th_walk = sh->theta0;
        demod_fft_symbol(A, x, 0, 1, tiles, Y, sh, lane, th_walk, pf_sym, utw);
        demod_fft_symbol(A, x, 1, 2 < n_sym ? 2 : -1, tiles, Y + 64, sh, lane, th_walk, pf_sym, utw);   // (a re-run asks for symbol 0 instead: the prefetch is dropped)
        if (A.dbg && pass == 0) t1 = __builtin_readcyclecounter();
        {
            float2 y0 = Y[0 * 64 + lane], y1 = Y[1 * 64 + lane];
            y0 = make_float2(lsign * y0.x, lsign * y0.y);  // burst marker: first LTS was negated on air
            if (is_car) { H0 = cdivc(y0, txv); H1 = cdivc(y1, txv); }
            bool rerun = false;
            if (pass == 0) {
                bool v = is_dat && cabs_(H0) > 0.01f && cabs_(H1) > 0.01f;
                float2 diff = cmul(H1, conj_(H0));
                float mag = cabs_(diff);
                v = v && mag > 1e-6f;
                float tr = v ? fdiv(diff.x, mag) : 0.0f, ti = v ? fdiv(diff.y, mag) : 0.0f;
                float o2[2];
                { const float t2[2] = {tr, ti}; ordered_sums<2, 60>(t2, is_dat, my_ord, K.n_data, T, lane, o2); }
                const float sr = o2[0], si = o2[1];
                int cnt = __popcll(__ballot(v));
                if (cnt > 10) {
                    float avg = atan2f_glibc(si, sr);
                    float dur = fdiv(1152.0f, 48000.0f);
                    float res = static_cast<float>(static_cast<double>(avg) / (2.0f * 3.14159265358979323846 * static_cast<double>(dur)));
                    if (fabs_(res) > 0.3f && fabs_(res) < 5.0f) {
                        rerun = true;
                        if (lane == 0) { sh->cfo = sh->cfo + res; sh->rerun = 1; }
                    }
                }
            }
            (void)rerun;
        }
        wave_sync();
        if (!(pass == 0 && sh->rerun)) break;
    }

    if (A.dbg) t2 = __builtin_readcyclecounter();
    {
        // H := last LTS symbol
        H = is_car ? H1 : make_float2(1.0f, 0.0f);
        {   // phase slope over adjacent logical carriers
            float2 hn = make_float2(lane_read(H.x, lane + 1), lane_read(H.y, lane + 1));
            bool v = (lane < kCarriers - 1) && cabs_(H) > 0.01f && cabs_(hn) > 0.01f;
            float2 diff = cmul(hn, conj_(H));
            float mag = cabs_(diff);
            v = v && mag > 1e-6f;
            float tr = v ? fdiv(diff.x, mag) : 0.0f, ti = v ? fdiv(diff.y, mag) : 0.0f;
            float o2[2];
            { const float t2[2] = {tr, ti}; ordered_sums<2, 60>(t2, lane < kCarriers - 1, lane, kCarriers - 1, T, lane, o2); }
            // the only call ordered by carrier (58 terms): give the entries beyond the data ordinals their +0.0 back
            if (lane >= K.n_data) { T[lane] = 0.0f; T[kSumRow + lane] = 0.0f; }
            wave_sync();
            const float sr = o2[0], si = o2[1];
            int cnt = __popcll(__ballot(v));
            if (cnt > 0) slope = atan2f_glibc(fdiv(si, static_cast<float>(cnt)), fdiv(sr, static_cast<float>(cnt)));
        }
        {   // noise variance + SNR from H1 - H0
            bool v = is_dat && cabs_(H0) > 1e-6f && cabs_(H1) > 1e-6f;
            float2 d = make_float2(H1.x - H0.x, H1.y - H0.y);
            float tn = v ? cnorm(d) : 0.0f;
            float ts = v ? fdiv(cnorm(H0) + cnorm(H1), 2.0f) : 0.0f;
            float o2[2];
            { const float t2[2] = {tn, ts}; ordered_sums<2, 60>(t2, is_dat, my_ord, K.n_data, T, lane, o2); }
            const float ns = o2[0], ss = o2[1];
            int cnt = __popcll(__ballot(v));
            if (cnt > 0) {
                float nv = fdiv(ns, 4.0f * static_cast<float>(cnt));
                float sp = fdiv(ss, static_cast<float>(cnt));
                float snr = fdiv(sp, maxf_(nv, 1e-10f));
                snr = maxf_(3.16f, minf_(10000.0f, snr));
                noise_var = nv;
                snr_lin = snr;
            }
        }
        {   // fading index of |H| over data carriers
            float a = is_dat ? cabs_(H) : 0.0f;
            float o1[1];
            { const float t1[1] = {a}; ordered_sums<1, 60>(t1, is_dat, my_ord, K.n_data, T, lane, o1); }
            float mean = fdiv(o1[0], static_cast<float>(K.n_data));
            float dd_ = a - mean;
            { const float t1[1] = {dd_ * dd_}; ordered_sums<1, 60>(t1, is_dat, my_ord, K.n_data, T, lane, o1); }
            float var = fdiv(o1[0], static_cast<float>(K.n_data));
            fading = (mean > 0.01f) ? fdiv(fsqrt(var), mean) : 0.0f;
        }
        snr_count = 2;

        // ================= data symbols
        float2 prev_pilot = make_float2(0, 0);
        bool have_prev = false, have_dd = false, have_ema = false, have_dprev = false, cp_init = false;
        float dd = 0.0f, ema = 0.0f, var = 0.0f, cnv = 0.0f;
        float2 cp_corr = make_float2(1.0f, 0.0f), dprev = make_float2(1.0f, 0.0f);
        const int mod = K.mod;
        const bool coh = K.coherent != 0;
        const int kk = K.kk[lane];
        const int lo = K.lo_lane[lane], hi = K.hi_lane[lane];
        const float ia = K.alpha[lane];
        const float npf = static_cast<float>(K.n_pilot), ndf = static_cast<float>(K.n_data);

        // the LTS phase slope is fixed for the frame: de-slope / re-slope rotations once per lane
        const float ph_des = -slope * static_cast<float>(kk), ph_res = slope * static_cast<float>(kk);
        const float2 rot_des = make_float2(cosf_glibc(ph_des), sinf_glibc(ph_des));
        const float2 rot_res = make_float2(cosf_glibc(ph_res), sinf_glibc(ph_res));

        for (int ds = 0; ds < K.n_data_symbols; ++ds) {
            demod_fft_symbol(A, x, 2 + ds, 3 + ds < n_sym ? 3 + ds : -1, tiles, Y + 128, sh, lane, th_walk, pf_sym, utw);
            float2 y = Y[128 + lane];
            const bool first = (ds == 0);
            // ---------- updateChannelEstimate (channel_equalizer.cpp:645-1043)
            if (K.n_pilot > 0) {
                float alpha = first ? 1.0f : (coh ? 0.9f : 0.5f);
                float2 hls = is_pil ? cdivc(y, txv) : make_float2(0, 0);
                if (coh) {
                    cp_init = true;
                } else if (!cp_init) {
                    float o2[2];
                    { const float t2[2] = {hls.x, hls.y}; ordered_sums<2, 16, kPilotRow>(t2, is_pil, my_ord, K.n_pilot, TP, lane, o2); }
                    float2 hsum = make_float2(o2[0], o2[1]);
                    float2 havg = make_float2(fdiv(hsum.x, npf), fdiv(hsum.y, npf));
                    float am = cabs_(havg);
                    if (am > 0.01f) { cp_corr = make_float2(fdiv(havg.x, am), fdiv(-havg.y, am)); cp_init = true; }
                }
                hls = cmul(hls, cp_corr);
                // All pilot-ordered sums of this symbol in one go (CPE numerator/denominator, signal power,
                // temporal noise power, mean pilot magnitude): they only depend on hls, H and prev_pilot.
                const float hls_abs = cabs_(hls);
                float hm = cabs_(H);
                float2 ratio = cmul(hls, conj_(H));
                float mag = cabs_(ratio);
                bool v = coh && is_pil && hm > 0.01f && mag > 1e-6f;
                bool nvv = is_pil && have_prev && cnorm(prev_pilot) > 1e-6f && cnorm(hls) > 1e-6f;
                float2 dp = make_float2(hls.x - prev_pilot.x, hls.y - prev_pilot.y);
                float o6[6];
                {
                    const float t6[6] = {v ? fdiv(ratio.x, mag) * hm : 0.0f, v ? fdiv(ratio.y, mag) * hm : 0.0f,
                                         v ? hm : 0.0f, cnorm(hls), nvv ? cnorm(dp) : 0.0f, hls_abs};
                    ordered_sums<6, 16, kPilotRow>(t6, is_pil, my_ord, K.n_pilot, TP, lane, o6);
                }
                if (coh) {  // common phase error
                    const float cr = o6[0], ci = o6[1], ws = o6[2];
                    if (ws > 0.01f) {
                        float ph = atan2f_glibc(ci, cr);
                        if (fabs_(ph) > 0.001f) H = cmul(H, cexpj(ph));
                    }
                }
                float signal_power = fdiv(o6[3], npf);
                float noise_power_sum = o6[4];
                int noise_count = __popcll(__ballot(nvv));
                if (is_pil) {
                    if (coh) {
                        H = make_float2(alpha * hls.x + (1.0f - alpha) * H.x, alpha * hls.y + (1.0f - alpha) * H.y);
                    } else {
                        float nm = alpha * cabs_(hls) + (1.0f - alpha) * cabs_(H);
                        float ph = carg_(H);
                        H = make_float2(nm * cosf_glibc(ph), nm * sinf_glibc(ph));
                    }
                }
                if (noise_count == 0) { noise_power_sum = fdiv(signal_power, 31.6f); noise_count = 1; }
                prev_pilot = hls;
                have_prev = true;
                // interpolation between pilots
                if (coh) {
                    float2 des = cmul(H, rot_des);
                    float2 hl = make_float2(lane_read(des.x, lo < 0 ? 0 : lo), lane_read(des.y, lo < 0 ? 0 : lo));
                    float2 hu = make_float2(lane_read(des.x, hi < 0 ? 0 : hi), lane_read(des.y, hi < 0 ? 0 : hi));
                    if (is_dat) {
                        float2 ih;
                        if (lo >= 0 && hi >= 0)
                            ih = make_float2((1.0f - ia) * hl.x + ia * hu.x, (1.0f - ia) * hl.y + ia * hu.y);
                        else if (lo >= 0) ih = hl;
                        else ih = hu;
                        H = cmul(ih, rot_res);
                    }
                } else {
                    float am = cabs_(H);
                    float m1 = lane_read(am, lo < 0 ? 0 : lo), m2 = lane_read(am, hi < 0 ? 0 : hi);
                    if (is_dat) {
                        float im = 0.0f;
                        if (lo >= 0 && hi >= 0) im = (1.0f - ia) * m1 + ia * m2;
                        else if (lo >= 0) im = m1;
                        else if (hi >= 0) im = m2;
                        float ph = carg_(H);
                        H = make_float2(im * cosf_glibc(ph), im * sinf_glibc(ph));
                    }
                }
                if (coh && have_dd && snr_count >= 3 && is_dat) {
                    if (fabs_(dd) > 0.001f) H = cmul(H, cexpj(dd * 0.3f));
                }
                {   // fading index from pilot magnitudes
                    float mean = fdiv(o6[5], npf);
                    float df = hls_abs - mean;
                    float o1[1];
                    { const float t1[1] = {df * df}; ordered_sums<1, 16, kPilotRow>(t1, is_pil, my_ord, K.n_pilot, TP, lane, o1); }
                    float vv = fdiv(o1[0], npf);
                    fading = (mean > 0.01f) ? fdiv(fsqrt(vv), mean) : 0.0f;
                }
                if (noise_count > 0 && noise_power_sum > 0.0f && coh && noise_count > 1) {
                    float inst = fdiv(signal_power, maxf_(noise_var, 1e-6f));
                    inst = maxf_(0.1f, minf_(10000.0f, inst));
                    snr_lin = 0.3f * inst + (1.0f - 0.3f) * snr_lin;
                }
                snr_count++;
            }
            // ---------- equalize (channel_equalizer.cpp:1259-1451)
            float2 eq = make_float2(0, 0);
            {
                float hp = cnorm(H);
                float o1[1];
                { const float t1[1] = {hp}; ordered_sums<1, 60>(t1, is_dat, my_ord, K.n_data, T, lane, o1); }
                float avg = fdiv(o1[0], ndf);
                float thr = 0.25f * avg;
                if (!coh) {
                    float snv = noise_var;
                    if (snv < 1e-6f) snv = fdiv(avg, 31.6f);
                    float den = hp + snv;
                    if (den < 1e-10f) { eq = make_float2(0, 0); cnv = 100.0f; }
                    else {
                        float2 pr = cmul(y, conj_(H));
                        eq = make_float2(fdiv(pr.x, den), fdiv(pr.y, den));
                        cnv = fdiv(snv, hp + snv);
                    }
                    if (hp < thr) cnv = 100.0f;
                    cnv = maxf_(1e-6f, minf_(100.0f, cnv));
                } else {
                    float den = hp + noise_var;
                    if (den < 1e-10f) { eq = make_float2(0, 0); cnv = 100.0f; }
                    else {
                        float2 pr = cmul(conj_(H), y);
                        eq = make_float2(fdiv(pr.x, den), fdiv(pr.y, den));
                        cnv = maxf_(1e-6f, minf_(100.0f, fdiv(noise_var, den)));
                    }
                    if (hp < thr) cnv = 100.0f;
                    bool dd_ok = (mod == RIA_MOD_QPSK || mod == RIA_MOD_BPSK || mod == RIA_MOD_QAM16 ||
                                  mod == RIA_MOD_QAM32 || mod == RIA_MOD_QAM64);
                    if (dd_ok && snr_count >= 2) {
                        have_dd = true;
                        float mt = 0.3f, pt = 0.61f;
                        if (mod == RIA_MOD_QAM16) { mt = 0.25f; pt = 0.44f; }
                        else if (mod == RIA_MOD_QAM32 || mod == RIA_MOD_QAM64) { mt = 0.20f; pt = 0.35f; }
                        if (cabs_(eq) < mt) dd = 0.0f;
                        else {
                            float2 dec = hard_decision(eq, mod);
                            float pe = carg_(cmul(eq, conj_(dec)));
                            dd = (fabs_(pe) < pt) ? -pe : 0.0f;
                        }
                    }
                }
            }
            // ---------- demodulateSymbol (demodulator.cpp:208-508)
            {
                float mag = cabs_(eq);
                if (!have_ema) { ema = mag; var = 0.0f; have_ema = true; }
                else {
                    float delta = mag - ema;
                    ema += 0.3f * delta;
                    var += 0.3f * (delta * delta - var);
                }
                if (!coh && !have_dprev) { dprev = make_float2(1.0f, 0.0f); have_dprev = true; }
                float nv = cnv * K.ce_margin;
                float msq = ema * ema + 1e-6f;
                float nvar = fdiv(var, msq);
                nv *= (1.0f + 10.0f * nvar);
                float o[8];
                float2 sym = eq;
                if (mod == RIA_MOD_D8PSK && fading > 0.30f) {
                    // demodulateD8PSKTwoPass (demodulator.cpp:533-620): common phase error from the embedded DQPSK
                    // grid (three ordered sums over the data carriers), half of it removed before the demap; the
                    // corrected symbol becomes the next differential reference
                    const double kPi = 3.14159265358979323846;
                    float ts = 0.0f, tc = 0.0f, tw = 0.0f;
                    const float sp = cabs_(eq) * cabs_(dprev);
                    if (is_dat && sp > 0.1f) {
                        const float2 diff = cmul(eq, conj_(dprev));
                        const float phase = atan2f_glibc(diff.y, diff.x);
                        const float pmo = static_cast<float>(static_cast<double>(phase) - kPi / static_cast<double>(4.0f));
                        int quadrant = static_cast<int>(__builtin_round(static_cast<double>(pmo * 2.0f) / kPi));
                        quadrant = ((quadrant % 4) + 4) % 4;
                        const float expected = static_cast<float>(quadrant * kPi / static_cast<double>(2.0f) + kPi / static_cast<double>(4.0f));
                        float err = phase - expected;
                        while (static_cast<double>(err) > kPi) err = static_cast<float>(static_cast<double>(err) - 2 * kPi);
                        while (static_cast<double>(err) < -kPi) err = static_cast<float>(static_cast<double>(err) + 2 * kPi);
                        ts = sp * sinf_glibc(err); tc = sp * cosf_glibc(err); tw = sp;
                    }
                    float o3[3];
                    { const float t3[3] = {ts, tc, tw}; ordered_sums<3, 60>(t3, is_dat, my_ord, K.n_data, T, lane, o3); }   // weak carriers add +0.0
                    const float mean_error = (o3[2] > 0.1f) ? atan2f_glibc(o3[0], o3[1]) : 0.0f;
                    if (fabs_(mean_error) > 0.05f && fabs_(mean_error) < 0.26f) {
                        const float ce = mean_error * 0.5f;
                        sym = cmul(eq, make_float2(cosf_glibc(-ce), sinf_glibc(-ce)));
                    } else {
                        sym = cmul(eq, make_float2(1.0f, 0.0f));
                    }
                    snr_count++;   // demodulator.cpp:292
                }
                int nb = demap_symbol(mod, sym, dprev, nv, o);
                if (!coh) dprev = sym;
                if (is_dat) {
                    float* dst = A.llr_out + static_cast<size_t>(frame) * A.llr_stride + ds * K.bits_per_symbol + K.ord[lane] * K.bits_per_carrier;
                    for (int b = 0; b < nb; ++b) dst[b] = o[b];
                }
            }
        }
        if (lane == 0) sh->th_end = th_walk;
        wave_sync();
        if (A.status && lane == 0) {
            ria_frame_status st;
            // 10*log10f(x): the only transcendental on the status path that is not bit-pinned; it is
            // a display value (OFDMDemodulator::getEstimatedSNR) and is checked to 1e-5 relative.
            st.snr_db = 10.0f * (logf_glibc(snr_lin) * 0.43429448190325176f);
            st.cfo_hz = sh->cfo;
            st.fading_index = fading;
            st.noise_variance = noise_var;
            st.lts_phase_slope = slope;
            st.snr_linear = snr_lin;
            st.corr_phase = (fabs_(sh->cfo) > 0.01f) ? sh->th_end : sh->theta0;
            st.n_llr = K.n_llr;
            A.status[frame] = st;
        }
    }
    if (A.dbg && threadIdx.x == 0) {
        t3 = __builtin_readcyclecounter();
        A.dbg[frame * 4 + 0] = t1 - t0; A.dbg[frame * 4 + 1] = t2 - t1; A.dbg[frame * 4 + 2] = t3 - t2; A.dbg[frame * 4 + 3] = t3 - t0;
    }
}

// =====================================================================================================
// Split pipeline (the default): the same arithmetic in separate kernels, so that the two halves of the work run
// at the occupancy each wants instead of sharing one wave per frame.  Per chunk of frames:
//   demod_walk_kernel(initial)  frames that arrive with a CFO: the serial phase recurrence over the two training symbols
//   demod_fft_kernel(0, 2, 0)   the two training symbols with the caller's CFO
//   demod_decide_kernel         1 wave per frame, lane = carrier: the residual-CFO decision of estimateChannelFromLTS
//                               (channel_equalizer.cpp:304-382) -> the frame's final CFO; frames whose CFO changed are listed
//   demod_walk_kernel(final)    1 lane per frame with a CFO: the serial float phase recurrence over the whole frame
//                               (channel_equalizer.cpp:132-144), leaving its value at every 72nd sample
//   demod_fft_kernel(2, n-2, 1) downconversion + 1024-point FFT of the data symbols; persistent waves, one symbol index per
//                               wave, 59 bins per symbol to the workspace
//   demod_fft_kernel(0, 2, 1, list)  the training symbols of the listed frames again (channel_equalizer.cpp:337-344)
//   demod_est_kernel<MOD>       1 wave per frame, lane = carrier: the sequential estimator / equaliser / demapper over the
//                               stored bins (latency-bound ordered sums and libm calls: few registers, many waves per SIMD)
struct DemodWs {
    float2* Y;        // [chunk][n_sym][64]  the 59 used bins of every symbol (logical carrier order)
    float* cfo;       // [chunk]  CFO after the training estimate
    float* theta0;    // [chunk]  correction phase at the first sample
    float* th_end;    // [chunk]  correction phase after the last sample
    int* rerun;       // [chunk]  the training symbols are transformed again with the corrected CFO
    float* marks;     // [chunk][n_sym][16]  correction phase at every 72nd sample (frames with |CFO| > 0.01 Hz)
};
struct DemodSplitArgs {
    DemodArgs a;      // samples / offsets / meta / outputs of the whole call
    DemodWs ws;
    int first, n;     // this chunk: frames [first, first + n)
    int n_sym;
};

__device__ __forceinline__ void demod_frame_start(const DemodArgs& A, int frame, float& cfo, float& theta0) {
    cfo = 0.0f; theta0 = 0.0f;
    if (A.meta) {
        cfo = A.meta[frame].cfo_hz;
        // ofdm_chirp_waveform.cpp:402-411
        float ip = static_cast<float>(-2.0f * 3.14159265358979323846 * static_cast<double>(cfo) *
                                      static_cast<double>(A.meta[frame].abs_position) / 48000.0);
        while (static_cast<double>(ip) > 3.14159265358979323846) ip = static_cast<float>(static_cast<double>(ip) - 2.0f * 3.14159265358979323846);
        while (static_cast<double>(ip) < -3.14159265358979323846) ip = static_cast<float>(static_cast<double>(ip) + 2.0f * 3.14159265358979323846);
        theta0 = ip;
    }
}

// the serial phase walk of one symbol by one lane, leaving the phase at every 72nd sample (see cfo_phase_step8)
__device__ __forceinline__ float demod_walk_symbol(float th, float inc, float* marks16) {
    for (int q = 0; q < 16; ++q) {
        marks16[q] = th;
#pragma unroll 1
        for (int g = 0; g < 9; ++g) th = cfo_phase_step8(th, inc);
    }
    return th;
}

// Persistent waves, one symbol INDEX per wave: wave w of the grid transforms symbol s0 + (w % ns) of frames w / ns, w / ns + F,
// ... (F = grid / ns frames in flight).  What depends on the symbol index only - the 16 mixer values of the lane, and the
// twiddles, which depend on the lane only - stays in registers for the whole walk over the frames; the 16 samples of the next
// frame are fetched while the current one is transformed, so a symbol's transform never waits for HBM.
//   use_final = 0  the caller's CFO (frame meta), before the training estimate;  1  the frame's final CFO
//   list / n_list  nullable: only these frames (the re-run of the training symbols, channel_equalizer.cpp:337-344)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void demod_fft_kernel(DemodSplitArgs S, int s0, int ns, int use_final, const int* __restrict__ list, const int* __restrict__ n_list) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DemodArgs& A = S.a;
    const int lane = threadIdx.x;
    const int w = blockIdx.x, s = s0 + w % ns, F = gridDim.x / ns;
    int i = w / ns;
    const int count = list ? *n_list : S.n;
    if (i >= F || i >= count) return;
    float2* tile = reinterpret_cast<float2*>(smem);
    float2* Yl = tile + kFftBufFloats2;
    const FftUniformTw utw = fft_load_uniform_tw(A.twiddle);
    const FftLaneTw ltw = fft_load_lane_tw(A.twiddle, lane);
    float2 ov[16];                                        // conj-able mixer values of this lane's 16 samples of symbol s
    {
        const float2* osc = A.nco + s * kSym + kCP;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = 4 * lane + 256 * c;
            const float4 o01 = *reinterpret_cast<const float4*>(osc + j), o23 = *reinterpret_cast<const float4*>(osc + j + 2);
            ov[4 * c] = make_float2(o01.x, o01.y); ov[4 * c + 1] = make_float2(o01.z, o01.w);
            ov[4 * c + 2] = make_float2(o23.x, o23.y); ov[4 * c + 3] = make_float2(o23.z, o23.w);
        }
    }
    auto frame_ptr = [&](int q) { const int frame = S.first + q; return A.samples + (A.offsets ? A.offsets[frame] : static_cast<uint64_t>(frame) * S.n_sym * kSym) + s * kSym + kCP; };
    auto fetch = [&](int q, float (&xv)[16]) {
        const float* xs = frame_ptr(q);
        if ((reinterpret_cast<uintptr_t>(xs) & 15u) == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float4 v = *reinterpret_cast<const float4*>(xs + 4 * lane + 256 * c); xv[4 * c] = v.x; xv[4 * c + 1] = v.y; xv[4 * c + 2] = v.z; xv[4 * c + 3] = v.w; }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[4 * c + e] = xs[4 * lane + 256 * c + e];
        }
    };
    float xn[16];
    int q = list ? list[i] : i;
    fetch(q, xn);
    for (;;) {
        float (&xv)[16] = xn;                              // mixed into the tile first; the next frame's samples are fetched right after
        const int qc = q;
        float cfo = 0.0f;
        if (use_final) cfo = S.ws.cfo[qc];
        else if (A.meta) cfo = A.meta[S.first + qc].cfo_hz;
        const bool use_cfo = fabs_(cfo) > 0.01f;           // wave-uniform
        if (use_cfo) {
            // 16 lanes re-walk 72 samples each from their marker (demod_walk_kernel) into the UPPER half of the tile (floats
            // 1024..2175).  The mixed samples then fill the tile from the bottom, a quarter (256 samples per wave) at a time:
            // quarter c reads the phases at floats [1152 + 256 c, +256) and writes floats [512 c, +512), which only ever
            // covers phases of quarters already consumed - with all reads of a quarter ahead of its writes.
            const float inc = static_cast<float>(-2.0f * 3.14159265358979323846 * static_cast<double>(cfo) / 48000.0);
            const float* marks = S.ws.marks + (static_cast<size_t>(qc) * S.n_sym + s) * 16;
            float* thb = reinterpret_cast<float*>(tile) + 1024;
            if (lane < 16) {
                float th = marks[lane];
                for (int k = 0; k < 72; ++k) { thb[72 * lane + k] = th; th = cfo_phase_step(th, inc); }
            }
            wave_sync();
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 th = *reinterpret_cast<const float4*>(thb + kCP + 4 * lane + 256 * c);
                wave_sync();
                const float tv[4] = {th.x, th.y, th.z, th.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 m = make_float2(xv[4 * c + e] * ov[4 * c + e].x, xv[4 * c + e] * -ov[4 * c + e].y);
                    tile[2 * lane + 256 * c + (e >> 1) + 128 * (e & 1)] = cmul(m, cexpj(tv[e]));   // fft_in_slot(4 lane + 256 c + e)
                }
                wave_sync();
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e)   // samples[i] * conj(osc) at fft_in_slot(i), i = 4 lane + 256 c + e
                    tile[2 * lane + 256 * c + (e >> 1) + 128 * (e & 1)] = make_float2(xv[4 * c + e] * ov[4 * c + e].x, xv[4 * c + e] * -ov[4 * c + e].y);
        }
        i += F;
        const bool more = i < count;
        if (more) { q = list ? list[i] : i; fetch(q, xn); }   // in flight during the whole transform below
        wave_sync();
        fft1024_wave<true, true>(tile, A.twiddle, utw, Yl, lane, &ltw);
        S.ws.Y[(static_cast<size_t>(qc) * S.n_sym + s) * 64 + lane] = Yl[lane];
        if (!more) break;
        wave_sync();
    }
}

// residual CFO between the two training symbols (channel_equalizer.cpp:304-382): one wave per frame, lane = carrier
constexpr int kDecideLds = 8 * kSumRow * 4 + 16;
__global__ __launch_bounds__(64) void demod_decide_kernel(DemodSplitArgs S) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DemodArgs& A = S.a;
    const DemodConst& K = *A.k;
    const int lane = threadIdx.x, q = blockIdx.x, frame = S.first + q;
    float* T = reinterpret_cast<float*>(smem);
    for (int i = lane; i < 8 * kSumRow; i += 64) T[i] = 0.0f;
    wave_sync();
    float cfo, theta0;
    demod_frame_start(A, frame, cfo, theta0);
    const float2* Yf = S.ws.Y + static_cast<size_t>(q) * S.n_sym * 64;
    const bool negate_lts0 = A.meta && (A.meta[frame].flags & 1u);
    const float lsign = negate_lts0 ? -1.0f : 1.0f;
    const bool is_car = lane < kCarriers, is_dat = is_car && !K.is_pilot[lane];
    const float2 txv = make_float2(K.tx_re[lane], K.tx_im[lane]);
    float2 y0 = Yf[lane], y1 = Yf[64 + lane];
    y0 = make_float2(lsign * y0.x, lsign * y0.y);
    float2 H0 = make_float2(0, 0), H1 = make_float2(0, 0);
    if (is_car) { H0 = cdivc(y0, txv); H1 = cdivc(y1, txv); }
    bool v = is_dat && cabs_(H0) > 0.01f && cabs_(H1) > 0.01f;
    const float2 diff = cmul(H1, conj_(H0));
    const float mag = cabs_(diff);
    v = v && mag > 1e-6f;
    const float tr = v ? fdiv(diff.x, mag) : 0.0f, ti = v ? fdiv(diff.y, mag) : 0.0f;
    float o2[2];
    { const float t2[2] = {tr, ti}; ordered_sums<2, 60>(t2, is_dat, K.ord[lane], K.n_data, T, lane, o2); }
    const int cnt = __popcll(__ballot(v));
    int rerun = 0;
    if (cnt > 10) {
        const float avg = atan2f_glibc(o2[1], o2[0]);
        const float dur = fdiv(1152.0f, 48000.0f);
        const float res = static_cast<float>(static_cast<double>(avg) / (2.0f * 3.14159265358979323846 * static_cast<double>(dur)));
        if (fabs_(res) > 0.3f && fabs_(res) < 5.0f) { rerun = 1; cfo = cfo + res; }
    }
    if (lane == 0) {
        S.ws.cfo[q] = cfo; S.ws.theta0[q] = theta0; S.ws.th_end[q] = theta0;
        if (rerun) S.ws.rerun[1 + atomicAdd(reinterpret_cast<unsigned int*>(S.ws.rerun), 1u)] = q;   // rerun[0] = count, then the frames
    }
}

// The serial float phase recurrence (channel_equalizer.cpp:132-144), one lane per frame that has a CFO: n_walk symbols from
// the frame's first sample, the phase at every 72nd sample kept for the transforms.  initial: the caller's CFO over the two
// training symbols (before the estimate); else the final CFO over the whole frame.
__global__ __launch_bounds__(64) void demod_walk_kernel(DemodSplitArgs S, int initial) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= S.n) return;
    float cfo, theta0;
    if (initial) demod_frame_start(S.a, S.first + q, cfo, theta0);
    else { cfo = S.ws.cfo[q]; theta0 = S.ws.theta0[q]; }
    if (!(fabs_(cfo) > 0.01f)) return;
    const float inc = static_cast<float>(-2.0f * 3.14159265358979323846 * static_cast<double>(cfo) / 48000.0);
    float th = theta0;
    float* m = S.ws.marks + static_cast<size_t>(q) * S.n_sym * 16;
    const int n_walk = initial ? 2 : S.n_sym;
    for (int s = 0; s < n_walk; ++s) th = demod_walk_symbol(th, inc, m + 16 * s);
    if (!initial) S.ws.th_end[q] = th;
}

#ifndef RIA_EST_WAVES
#define RIA_EST_WAVES 4     // waves per SIMD the estimator kernel is held to (128 VGPRs, no spills; measured per 100 000 frames: 3 waves 6.6 ms, 4 waves 6.3, 5 (96 VGPRs, 92 B of scratch) 6.6, 6 waves 7.7)
#endif
constexpr int kEstLds = (8 * kSumRow + 8 * kPilotRow) * 4 + 16;
// MOD: the modulation as a compile-time constant - every instantiation carries one demapper and one decision-directed
// path instead of all of them (a run-time switch over the nine modulations needs 165 VGPRs, the QAM16 body 1/2 of that)
template <int MOD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RIA_EST_WAVES))) void demod_est_kernel(DemodSplitArgs S) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DemodArgs& A = S.a;
    const DemodConst& K = *A.k;
    const int lane = threadIdx.x;
    const int q = blockIdx.x, frame = S.first + q;
    float* T = reinterpret_cast<float*>(smem);
    float* TP = T + 8 * kSumRow;
    for (int i = lane; i < 8 * kSumRow; i += 64) T[i] = 0.0f;       // ordered-sum scratch: +0.0 everywhere (see ordered_sums)
    for (int i = lane; i < 8 * kPilotRow; i += 64) TP[i] = 0.0f;
    wave_sync();
    const float2* Yf = S.ws.Y + static_cast<size_t>(q) * S.n_sym * 64;
    const float cfo_final = S.ws.cfo[q];
    const bool negate_lts0 = A.meta && (A.meta[frame].flags & 1u);
    float2 H = make_float2(1.0f, 0.0f);
    float noise_var = 0.1f, snr_lin = 1.0f, fading = 0.0f, slope = 0.0f;
    int snr_count = 0;
    const bool is_car = lane < kCarriers;
    const bool is_pil = is_car && K.is_pilot[lane];
    const bool is_dat = is_car && !K.is_pilot[lane];
    const int my_ord = K.ord[lane];
    const float2 txv = make_float2(K.tx_re[lane], K.tx_im[lane]);
    const float lsign = negate_lts0 ? -1.0f : 1.0f;
    float2 H0 = make_float2(0, 0), H1 = make_float2(0, 0);
    {   // LS estimate on the two training symbols (channel_equalizer.cpp:257-288), final CFO
        float2 y0 = Yf[lane], y1 = Yf[64 + lane];
        y0 = make_float2(lsign * y0.x, lsign * y0.y);    // burst marker: first LTS was negated on air
        if (is_car) { H0 = cdivc(y0, txv); H1 = cdivc(y1, txv); }
    }
    float2 y_next = (K.n_data_symbols > 0) ? Yf[2 * 64 + lane] : make_float2(0, 0);
    {
        // H := last LTS symbol
        H = is_car ? H1 : make_float2(1.0f, 0.0f);
        {   // phase slope over adjacent logical carriers
            float2 hn = make_float2(lane_read(H.x, lane + 1), lane_read(H.y, lane + 1));
            bool v = (lane < kCarriers - 1) && cabs_(H) > 0.01f && cabs_(hn) > 0.01f;
            float2 diff = cmul(hn, conj_(H));
            float mag = cabs_(diff);
            v = v && mag > 1e-6f;
            float tr = v ? fdiv(diff.x, mag) : 0.0f, ti = v ? fdiv(diff.y, mag) : 0.0f;
            float o2[2];
            { const float t2[2] = {tr, ti}; ordered_sums<2, 60, kSumRow, 4>(t2, lane < kCarriers - 1, lane, kCarriers - 1, T, lane, o2); }
            // the only call ordered by carrier (58 terms): give the entries beyond the data ordinals their +0.0 back
            if (lane >= K.n_data) { T[lane] = 0.0f; T[kSumRow + lane] = 0.0f; }
            wave_sync();
            const float sr = o2[0], si = o2[1];
            int cnt = __popcll(__ballot(v));
            if (cnt > 0) slope = atan2f_glibc(fdiv(si, static_cast<float>(cnt)), fdiv(sr, static_cast<float>(cnt)));
        }
        {   // noise variance + SNR from H1 - H0
            bool v = is_dat && cabs_(H0) > 1e-6f && cabs_(H1) > 1e-6f;
            float2 d = make_float2(H1.x - H0.x, H1.y - H0.y);
            float tn = v ? cnorm(d) : 0.0f;
            float ts = v ? fdiv(cnorm(H0) + cnorm(H1), 2.0f) : 0.0f;
            float o2[2];
            { const float t2[2] = {tn, ts}; ordered_sums<2, 60, kSumRow, 4>(t2, is_dat, my_ord, K.n_data, T, lane, o2); }
            const float ns = o2[0], ss = o2[1];
            int cnt = __popcll(__ballot(v));
            if (cnt > 0) {
                float nv = fdiv(ns, 4.0f * static_cast<float>(cnt));
                float sp = fdiv(ss, static_cast<float>(cnt));
                float snr = fdiv(sp, maxf_(nv, 1e-10f));
                snr = maxf_(3.16f, minf_(10000.0f, snr));
                noise_var = nv;
                snr_lin = snr;
            }
        }
        {   // fading index of |H| over data carriers
            float a = is_dat ? cabs_(H) : 0.0f;
            float o1[1];
            { const float t1[1] = {a}; ordered_sums<1, 60, kSumRow, 4>(t1, is_dat, my_ord, K.n_data, T, lane, o1); }
            float mean = fdiv(o1[0], static_cast<float>(K.n_data));
            float dd_ = a - mean;
            { const float t1[1] = {dd_ * dd_}; ordered_sums<1, 60, kSumRow, 4>(t1, is_dat, my_ord, K.n_data, T, lane, o1); }
            float var = fdiv(o1[0], static_cast<float>(K.n_data));
            fading = (mean > 0.01f) ? fdiv(fsqrt(var), mean) : 0.0f;
        }
        snr_count = 2;

        // ================= data symbols
        float2 prev_pilot = make_float2(0, 0);
        bool have_prev = false, have_dd = false, have_ema = false, have_dprev = false, cp_init = false;
        float dd = 0.0f, ema = 0.0f, var = 0.0f, cnv = 0.0f;
        float2 cp_corr = make_float2(1.0f, 0.0f), dprev = make_float2(1.0f, 0.0f);
        constexpr int mod = MOD;
        constexpr bool coh = (MOD == RIA_MOD_BPSK || MOD == RIA_MOD_QPSK || MOD == RIA_MOD_QAM16 || MOD == RIA_MOD_QAM32 || MOD == RIA_MOD_QAM64 || MOD == RIA_MOD_QAM256);
        const int kk = K.kk[lane];
        const int lo = K.lo_lane[lane], hi = K.hi_lane[lane];
        const float ia = K.alpha[lane];
        const float npf = static_cast<float>(K.n_pilot), ndf = static_cast<float>(K.n_data);

        // the LTS phase slope is fixed for the frame: de-slope / re-slope rotations once per lane
        const float ph_des = -slope * static_cast<float>(kk), ph_res = slope * static_cast<float>(kk);
        const float2 rot_des = make_float2(cosf_glibc(ph_des), sinf_glibc(ph_des));
        const float2 rot_res = make_float2(cosf_glibc(ph_res), sinf_glibc(ph_res));

        for (int ds = 0; ds < K.n_data_symbols; ++ds) {
            const float2 y = y_next;
            if (ds + 1 < K.n_data_symbols) y_next = Yf[(3 + ds) * 64 + lane];   // the next symbol's bins are on their way while this one is worked on
            const bool first = (ds == 0);
            // ---------- updateChannelEstimate (channel_equalizer.cpp:645-1043)
            if (K.n_pilot > 0) {
                float alpha = first ? 1.0f : (coh ? 0.9f : 0.5f);
                float2 hls = is_pil ? cdivc(y, txv) : make_float2(0, 0);
                if (coh) {
                    cp_init = true;
                } else if (!cp_init) {
                    float o2[2];
                    { const float t2[2] = {hls.x, hls.y}; ordered_sums<2, 16, kPilotRow>(t2, is_pil, my_ord, K.n_pilot, TP, lane, o2); }
                    float2 hsum = make_float2(o2[0], o2[1]);
                    float2 havg = make_float2(fdiv(hsum.x, npf), fdiv(hsum.y, npf));
                    float am = cabs_(havg);
                    if (am > 0.01f) { cp_corr = make_float2(fdiv(havg.x, am), fdiv(-havg.y, am)); cp_init = true; }
                }
                hls = cmul(hls, cp_corr);
                // All pilot-ordered sums of this symbol in one go (CPE numerator/denominator, signal power,
                // temporal noise power, mean pilot magnitude): they only depend on hls, H and prev_pilot.
                const float hls_abs = cabs_(hls);
                float hm = cabs_(H);
                float2 ratio = cmul(hls, conj_(H));
                float mag = cabs_(ratio);
                bool v = coh && is_pil && hm > 0.01f && mag > 1e-6f;
                bool nvv = is_pil && have_prev && cnorm(prev_pilot) > 1e-6f && cnorm(hls) > 1e-6f;
                float2 dp = make_float2(hls.x - prev_pilot.x, hls.y - prev_pilot.y);
                float o6[6];
                {
                    const float t6[6] = {v ? fdiv(ratio.x, mag) * hm : 0.0f, v ? fdiv(ratio.y, mag) * hm : 0.0f,
                                         v ? hm : 0.0f, cnorm(hls), nvv ? cnorm(dp) : 0.0f, hls_abs};
                    ordered_sums<6, 16, kPilotRow>(t6, is_pil, my_ord, K.n_pilot, TP, lane, o6);
                }
                if (coh) {  // common phase error
                    const float cr = o6[0], ci = o6[1], ws = o6[2];
                    if (ws > 0.01f) {
                        float ph = atan2f_glibc(ci, cr);
                        if (fabs_(ph) > 0.001f) H = cmul(H, cexpj(ph));
                    }
                }
                float signal_power = fdiv(o6[3], npf);
                float noise_power_sum = o6[4];
                int noise_count = __popcll(__ballot(nvv));
                if (is_pil) {
                    if (coh) {
                        H = make_float2(alpha * hls.x + (1.0f - alpha) * H.x, alpha * hls.y + (1.0f - alpha) * H.y);
                    } else {
                        float nm = alpha * cabs_(hls) + (1.0f - alpha) * cabs_(H);
                        float ph = carg_(H);
                        H = make_float2(nm * cosf_glibc(ph), nm * sinf_glibc(ph));
                    }
                }
                if (noise_count == 0) { noise_power_sum = fdiv(signal_power, 31.6f); noise_count = 1; }
                prev_pilot = hls;
                have_prev = true;
                // interpolation between pilots
                if (coh) {
                    float2 des = cmul(H, rot_des);
                    float2 hl = make_float2(lane_read(des.x, lo < 0 ? 0 : lo), lane_read(des.y, lo < 0 ? 0 : lo));
                    float2 hu = make_float2(lane_read(des.x, hi < 0 ? 0 : hi), lane_read(des.y, hi < 0 ? 0 : hi));
                    if (is_dat) {
                        float2 ih;
                        if (lo >= 0 && hi >= 0)
                            ih = make_float2((1.0f - ia) * hl.x + ia * hu.x, (1.0f - ia) * hl.y + ia * hu.y);
                        else if (lo >= 0) ih = hl;
                        else ih = hu;
                        H = cmul(ih, rot_res);
                    }
                } else {
                    float am = cabs_(H);
                    float m1 = lane_read(am, lo < 0 ? 0 : lo), m2 = lane_read(am, hi < 0 ? 0 : hi);
                    if (is_dat) {
                        float im = 0.0f;
                        if (lo >= 0 && hi >= 0) im = (1.0f - ia) * m1 + ia * m2;
                        else if (lo >= 0) im = m1;
                        else if (hi >= 0) im = m2;
                        float ph = carg_(H);
                        H = make_float2(im * cosf_glibc(ph), im * sinf_glibc(ph));
                    }
                }
                if (coh && have_dd && snr_count >= 3 && is_dat) {
                    if (fabs_(dd) > 0.001f) H = cmul(H, cexpj(dd * 0.3f));
                }
                {   // fading index from pilot magnitudes
                    float mean = fdiv(o6[5], npf);
                    float df = hls_abs - mean;
                    float o1[1];
                    { const float t1[1] = {df * df}; ordered_sums<1, 16, kPilotRow>(t1, is_pil, my_ord, K.n_pilot, TP, lane, o1); }
                    float vv = fdiv(o1[0], npf);
                    fading = (mean > 0.01f) ? fdiv(fsqrt(vv), mean) : 0.0f;
                }
                if (noise_count > 0 && noise_power_sum > 0.0f && coh && noise_count > 1) {
                    float inst = fdiv(signal_power, maxf_(noise_var, 1e-6f));
                    inst = maxf_(0.1f, minf_(10000.0f, inst));
                    snr_lin = 0.3f * inst + (1.0f - 0.3f) * snr_lin;
                }
                snr_count++;
            }
            // ---------- equalize (channel_equalizer.cpp:1259-1451)
            float2 eq = make_float2(0, 0);
            {
                float hp = cnorm(H);
                float o1[1];
                { const float t1[1] = {hp}; ordered_sums<1, 60, kSumRow, 4>(t1, is_dat, my_ord, K.n_data, T, lane, o1); }
                float avg = fdiv(o1[0], ndf);
                float thr = 0.25f * avg;
                if (!coh) {
                    float snv = noise_var;
                    if (snv < 1e-6f) snv = fdiv(avg, 31.6f);
                    float den = hp + snv;
                    if (den < 1e-10f) { eq = make_float2(0, 0); cnv = 100.0f; }
                    else {
                        float2 pr = cmul(y, conj_(H));
                        eq = make_float2(fdiv(pr.x, den), fdiv(pr.y, den));
                        cnv = fdiv(snv, hp + snv);
                    }
                    if (hp < thr) cnv = 100.0f;
                    cnv = maxf_(1e-6f, minf_(100.0f, cnv));
                } else {
                    float den = hp + noise_var;
                    if (den < 1e-10f) { eq = make_float2(0, 0); cnv = 100.0f; }
                    else {
                        float2 pr = cmul(conj_(H), y);
                        eq = make_float2(fdiv(pr.x, den), fdiv(pr.y, den));
                        cnv = maxf_(1e-6f, minf_(100.0f, fdiv(noise_var, den)));
                    }
                    if (hp < thr) cnv = 100.0f;
                    bool dd_ok = (mod == RIA_MOD_QPSK || mod == RIA_MOD_BPSK || mod == RIA_MOD_QAM16 ||
                                  mod == RIA_MOD_QAM32 || mod == RIA_MOD_QAM64);
                    if (dd_ok && snr_count >= 2) {
                        have_dd = true;
                        float mt = 0.3f, pt = 0.61f;
                        if (mod == RIA_MOD_QAM16) { mt = 0.25f; pt = 0.44f; }
                        else if (mod == RIA_MOD_QAM32 || mod == RIA_MOD_QAM64) { mt = 0.20f; pt = 0.35f; }
                        if (cabs_(eq) < mt) dd = 0.0f;
                        else {
                            float2 dec = hard_decision(eq, mod);
                            float pe = carg_(cmul(eq, conj_(dec)));
                            dd = (fabs_(pe) < pt) ? -pe : 0.0f;
                        }
                    }
                }
            }
            // ---------- demodulateSymbol (demodulator.cpp:208-508)
            {
                float mag = cabs_(eq);
                if (!have_ema) { ema = mag; var = 0.0f; have_ema = true; }
                else {
                    float delta = mag - ema;
                    ema += 0.3f * delta;
                    var += 0.3f * (delta * delta - var);
                }
                if (!coh && !have_dprev) { dprev = make_float2(1.0f, 0.0f); have_dprev = true; }
                float nv = cnv * K.ce_margin;
                float msq = ema * ema + 1e-6f;
                float nvar = fdiv(var, msq);
                nv *= (1.0f + 10.0f * nvar);
                float o[8];
                float2 sym = eq;
                if (mod == RIA_MOD_D8PSK && fading > 0.30f) {
                    // demodulateD8PSKTwoPass (demodulator.cpp:533-620): common phase error from the embedded DQPSK
                    // grid (three ordered sums over the data carriers), half of it removed before the demap; the
                    // corrected symbol becomes the next differential reference
                    const double kPi = 3.14159265358979323846;
                    float ts = 0.0f, tc = 0.0f, tw = 0.0f;
                    const float sp = cabs_(eq) * cabs_(dprev);
                    if (is_dat && sp > 0.1f) {
                        const float2 diff = cmul(eq, conj_(dprev));
                        const float phase = atan2f_glibc(diff.y, diff.x);
                        const float pmo = static_cast<float>(static_cast<double>(phase) - kPi / static_cast<double>(4.0f));
                        int quadrant = static_cast<int>(__builtin_round(static_cast<double>(pmo * 2.0f) / kPi));
                        quadrant = ((quadrant % 4) + 4) % 4;
                        const float expected = static_cast<float>(quadrant * kPi / static_cast<double>(2.0f) + kPi / static_cast<double>(4.0f));
                        float err = phase - expected;
                        while (static_cast<double>(err) > kPi) err = static_cast<float>(static_cast<double>(err) - 2 * kPi);
                        while (static_cast<double>(err) < -kPi) err = static_cast<float>(static_cast<double>(err) + 2 * kPi);
                        ts = sp * sinf_glibc(err); tc = sp * cosf_glibc(err); tw = sp;
                    }
                    float o3[3];
                    { const float t3[3] = {ts, tc, tw}; ordered_sums<3, 60, kSumRow, 4>(t3, is_dat, my_ord, K.n_data, T, lane, o3); }   // weak carriers add +0.0
                    const float mean_error = (o3[2] > 0.1f) ? atan2f_glibc(o3[0], o3[1]) : 0.0f;
                    if (fabs_(mean_error) > 0.05f && fabs_(mean_error) < 0.26f) {
                        const float ce = mean_error * 0.5f;
                        sym = cmul(eq, make_float2(cosf_glibc(-ce), sinf_glibc(-ce)));
                    } else {
                        sym = cmul(eq, make_float2(1.0f, 0.0f));
                    }
                    snr_count++;   // demodulator.cpp:292
                }
                int nb = demap_symbol(mod, sym, dprev, nv, o);
                if (!coh) dprev = sym;
                if (is_dat) {
                    float* dst = A.llr_out + static_cast<size_t>(frame) * A.llr_stride + ds * K.bits_per_symbol + K.ord[lane] * K.bits_per_carrier;
                    for (int b = 0; b < nb; ++b) dst[b] = o[b];
                }
            }
        }
        if (A.status && lane == 0) {
            ria_frame_status st;
            // 10*log10f(x): the only transcendental on the status path that is not bit-pinned; it is
            // a display value (OFDMDemodulator::getEstimatedSNR) and is checked to 1e-5 relative.
            st.snr_db = 10.0f * (logf_glibc(snr_lin) * 0.43429448190325176f);
            st.cfo_hz = cfo_final;
            st.fading_index = fading;
            st.noise_variance = noise_var;
            st.lts_phase_slope = slope;
            st.snr_linear = snr_lin;
            st.corr_phase = S.ws.th_end[q];
            st.n_llr = K.n_llr;
            A.status[frame] = st;
        }
    }
}

inline int demod_fft_lds_bytes() { return kFftBufFloats2 * 8 + 64 * 8 + 16 * 4 + 16; }
inline int demod_fused_lds_bytes() { return kFftBufFloats2 * 8 + 3 * 64 * 8 + static_cast<int>(sizeof(DemodShared)) + 16; }
inline void launch_demod_fused(const DemodArgs& A, hipStream_t s) {   // the one-wave-per-frame form (A/B reference: RIA_DEMOD_FUSED=1)
    hipLaunchKernelGGL(demod_frames_kernel, dim3(A.n_frames), dim3(kDemodThreads), demod_fused_lds_bytes(), s, A);
}
inline hipError_t demod_set_attributes() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(demod_frames_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, demod_fused_lds_bytes());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(demod_fft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, demod_fft_lds_bytes());
}
// Frames per chunk of the split pipeline.  The bins of a chunk (n_sym x 512 B per frame) go out to HBM and come back once
// (+22 % on the 74 KB of samples a frame brings in; the demodulator runs at a tenth of the HBM rate), so chunks are as large
// as the workspace allows: the serial phase walk of the frames with a CFO is a fixed latency per chunk.
inline int demod_chunk_frames(int n_sym) { const int c = (320 << 20) / (n_sym * 512); return c < 1 ? 1 : c; }   // 40 960 frames of the named shape
inline size_t demod_ws_bytes(int n_sym) {
    const size_t c = static_cast<size_t>(demod_chunk_frames(n_sym));
    return c * n_sym * 64 * sizeof(float2) + c * 4 * sizeof(float) + c * n_sym * 16 * sizeof(float) + 512;
}
inline void launch_demod(const DemodArgs& A, const ria_gpu_geometry& g, int mod, void* ws, hipStream_t s) {
    const int n_sym = 2 + g.n_data_symbols, chunk = demod_chunk_frames(n_sym);
    DemodSplitArgs S{};
    S.a = A; S.n_sym = n_sym;
    unsigned char* p = static_cast<unsigned char*>(ws);
    S.ws.Y = reinterpret_cast<float2*>(p); p += static_cast<size_t>(chunk) * n_sym * 64 * sizeof(float2);
    S.ws.cfo = reinterpret_cast<float*>(p); p += static_cast<size_t>(chunk) * sizeof(float);
    S.ws.theta0 = reinterpret_cast<float*>(p); p += static_cast<size_t>(chunk) * sizeof(float);
    S.ws.th_end = reinterpret_cast<float*>(p); p += static_cast<size_t>(chunk) * sizeof(float);
    S.ws.rerun = reinterpret_cast<int*>(p); p += static_cast<size_t>(chunk + 1) * sizeof(int);   // [0] = count, then the listed frames
    p = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(p) + 255) & ~uintptr_t(255));
    S.ws.marks = reinterpret_cast<float*>(p);
    constexpr int kFftWaves = 256 * 12;                   // persistent grid = what is resident at once: 3 waves per SIMD (168 VGPRs)
    auto fft = [&](int s0, int ns, int use_final, const int* list, const int* n_list, int n_units) {
        int F = kFftWaves / ns;
        if (F > n_units) F = n_units;
        if (F < 1) F = 1;
        hipLaunchKernelGGL(demod_fft_kernel, dim3(F * ns), dim3(64), demod_fft_lds_bytes(), s, S, s0, ns, use_final, list, n_list);
    };
    const int n_chunks = (A.n_frames + chunk - 1) / chunk, even = (A.n_frames + n_chunks - 1) / n_chunks;   // equal chunks: every chunk pays the fixed latencies once
    for (int first = 0; first < A.n_frames; first += even) {
        S.first = first; S.n = (A.n_frames - first < even) ? A.n_frames - first : even;
        (void)hipMemsetAsync(S.ws.rerun, 0, sizeof(int), s);
        if (A.meta) hipLaunchKernelGGL(demod_walk_kernel, dim3((S.n + 63) / 64), dim3(64), 0, s, S, 1);
        fft(0, 2, 0, nullptr, nullptr, S.n);                                             // training symbols, caller's CFO
        hipLaunchKernelGGL(demod_decide_kernel, dim3(S.n), dim3(64), kDecideLds, s, S);
        hipLaunchKernelGGL(demod_walk_kernel, dim3((S.n + 63) / 64), dim3(64), 0, s, S, 0);
        fft(2, n_sym - 2, 1, nullptr, nullptr, S.n);                                     // data symbols, final CFO
        fft(0, 2, 1, S.ws.rerun + 1, S.ws.rerun, S.n);                                   // training symbols again where the CFO changed
        switch (mod) {
            case RIA_MOD_DBPSK: hipLaunchKernelGGL(demod_est_kernel<RIA_MOD_DBPSK>, dim3(S.n), dim3(64), kEstLds, s, S); break;
            case RIA_MOD_BPSK: hipLaunchKernelGGL(demod_est_kernel<RIA_MOD_BPSK>, dim3(S.n), dim3(64), kEstLds, s, S); break;
            case RIA_MOD_DQPSK: hipLaunchKernelGGL(demod_est_kernel<RIA_MOD_DQPSK>, dim3(S.n), dim3(64), kEstLds, s, S); break;
            case RIA_MOD_QPSK: hipLaunchKernelGGL(demod_est_kernel<RIA_MOD_QPSK>, dim3(S.n), dim3(64), kEstLds, s, S); break;
            case RIA_MOD_D8PSK: hipLaunchKernelGGL(demod_est_kernel<RIA_MOD_D8PSK>, dim3(S.n), dim3(64), kEstLds, s, S); break;
            case RIA_MOD_QAM32: hipLaunchKernelGGL(demod_est_kernel<RIA_MOD_QAM32>, dim3(S.n), dim3(64), kEstLds, s, S); break;
            case RIA_MOD_QAM64: hipLaunchKernelGGL(demod_est_kernel<RIA_MOD_QAM64>, dim3(S.n), dim3(64), kEstLds, s, S); break;
            case RIA_MOD_QAM256: hipLaunchKernelGGL(demod_est_kernel<RIA_MOD_QAM256>, dim3(S.n), dim3(64), kEstLds, s, S); break;
            default: hipLaunchKernelGGL(demod_est_kernel<RIA_MOD_QAM16>, dim3(S.n), dim3(64), kEstLds, s, S); break;
        }
    }
}

}  // namespace ria
