// ria_amd/csrc/tx_kernels.hip.h — frame synthesis and channel simulation kernels for Monte-Carlo sweeps.
//
//   make_frames_kernel  v2::makeFixedDataFrame(...).serialize()          frame_v2.cpp:1890-1912, :502-554
//   tx_frames_kernel    v2::encodeFixedFrame                             frame_v2.cpp:1285-1328
//                       LDPCEncoder::encode                              ldpc_encoder.cpp:193-257
//                       OFDMModulator::generateTrainingSymbols/modulate  modulator.cpp:534-583, :348-477
//                       createOFDMSymbol/complexToReal (IFFT, CP, upmix x40) modulator.cpp:217-283
//                       -> bit-identical audio to the reference TX for the same info bytes
//   channel_exact_kernel  sim::WattersonChannel::process                 hf_channel.hpp:107-177, :267-284
//                       the reference's own mt19937 -> normal_distribution<float> stream: bit-identical output
//   channel_kernel      the same model on a counter-based RNG (Philox4x32-10): order-independent and faster,
//                       STATISTICAL parity only
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ria_gpu.h"
#include "demod_kernels.hip.h"
#include "host_tables.hpp"
#include "ldpc_kernels.hip.h"

namespace ria {

struct TxConst {
    int mod, n_data, n_pilot, bits_per_carrier, bits_per_symbol, n_data_symbols;
    int k, m, bytes_per_cw;
    int data_bin[64], pilot_bin[64];
    float pilot_seq[64];
    float sync_re[64], sync_im[64];   // by data ordinal
    uint8_t row_deg[648];
    uint16_t row_var[7 * 648];        // [7][m]
    uint16_t scatter[4 * 648];        // coded bit i of codeword cw -> position in the 2592-bit stream
};

inline TxConst build_tx_const(const CarrierPlan& p, const LdpcCode& c, int mod, const ria_gpu_geometry& g) {
    TxConst t{};
    t.mod = mod;
    t.n_data = p.n_data; t.n_pilot = p.n_pilot;
    t.bits_per_carrier = g.bits_per_carrier; t.bits_per_symbol = g.bits_per_symbol;
    t.n_data_symbols = g.n_data_symbols;
    t.k = c.k; t.m = c.m; t.bytes_per_cw = g.bytes_per_codeword;
    for (int i = 0; i < p.n_data; ++i) { t.data_bin[i] = p.data_bin[i]; t.sync_re[i] = p.sync_re[i % kCarriers]; t.sync_im[i] = p.sync_im[i % kCarriers]; }
    for (int i = 0; i < p.n_pilot; ++i) { t.pilot_bin[i] = p.pilot_bin[i]; t.pilot_seq[i] = p.pilot_seq[i]; }
    for (int i = 0; i < c.m; ++i) t.row_deg[i] = c.row_deg[i];
    for (size_t i = 0; i < c.row_var.size(); ++i) t.row_var[i] = c.row_var[i];
    auto sc = build_rx_gather(g.bits_per_symbol, true);
    for (size_t i = 0; i < sc.size(); ++i) t.scatter[i] = sc[i];
    return t;
}

// ---------------------------------------------------------------- counter RNG
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0, p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
        uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0, n1 = static_cast<uint32_t>(p1);
        uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1, n3 = static_cast<uint32_t>(p0);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(uint32_t u) { return (static_cast<float>(u >> 8) + 0.5f) * (1.0f / 16777216.0f); }
// two independent N(0,1) from two uniforms (Box-Muller)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float* n0, float* n1) {
    float r = sqrtf(-2.0f * __logf(u01(a)));
    float s, c;
    __sincosf(6.283185307179586f * u01(b), &s, &c);
    *n0 = r * c; *n1 = r * s;
}

// ---------------------------------------------------------------- frame builder
__global__ __launch_bounds__(64) void make_frames_kernel(const TxConst* __restrict__ T, const uint16_t* crc_bit,
                                                         const uint16_t* crc_init, uint64_t seed, int first_seq,
                                                         int n_frames, uint8_t* __restrict__ out) {
    __shared__ uint8_t fr[4 * 68];
    const int lane = threadIdx.x, frame = blockIdx.x;
    const int total = 4 * T->bytes_per_cw, plen = total - 19;
    const int seq = (first_seq + frame) & 0xFFFF;
    for (int b = lane; b < total; b += 64) {
        uint32_t r[4];
        philox4x32(static_cast<uint32_t>(frame + static_cast<uint32_t>(first_seq)), static_cast<uint32_t>(b >> 2), 0x5249u, 0,
                   static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), r);
        fr[b] = (b >= 17 && b < 17 + plen) ? static_cast<uint8_t>(r[b & 3] >> 11) : 0;
    }
    wave_sync();
    if (lane == 0) {
        // hashCallsign("TEST") = 0x..., ("RX"): djb2-xor over upper-case chars, 24 bits (frame_v2.cpp:78-84)
        uint32_t hs = 5381, hd = 5381;
        const char s1[4] = {'T', 'E', 'S', 'T'}, s2[2] = {'R', 'X'};
        for (int i = 0; i < 4; ++i) hs = ((hs << 5) + hs) ^ static_cast<uint8_t>(s1[i]);
        for (int i = 0; i < 2; ++i) hd = ((hd << 5) + hd) ^ static_cast<uint8_t>(s2[i]);
        hs &= 0xFFFFFF; hd &= 0xFFFFFF;
        fr[0] = 0x55; fr[1] = 0x4C; fr[2] = 0x30; fr[3] = 0x01;
        fr[4] = static_cast<uint8_t>(seq >> 8); fr[5] = static_cast<uint8_t>(seq);
        fr[6] = static_cast<uint8_t>(hs >> 16); fr[7] = static_cast<uint8_t>(hs >> 8); fr[8] = static_cast<uint8_t>(hs);
        fr[9] = static_cast<uint8_t>(hd >> 16); fr[10] = static_cast<uint8_t>(hd >> 8); fr[11] = static_cast<uint8_t>(hd);
        fr[12] = 4;
        fr[13] = static_cast<uint8_t>(plen >> 8); fr[14] = static_cast<uint8_t>(plen);
    }
    wave_sync();
    uint32_t hc = crc16_wave(fr, 15, crc_bit, crc_init, lane);
    if (lane == 0) { fr[15] = static_cast<uint8_t>(hc >> 8); fr[16] = static_cast<uint8_t>(hc); }
    wave_sync();
    uint32_t fc = crc16_wave(fr, total - 2, crc_bit, crc_init, lane);
    if (lane == 0) { fr[total - 2] = static_cast<uint8_t>(fc >> 8); fr[total - 1] = static_cast<uint8_t>(fc); }
    wave_sync();
    for (int b = lane; b < total; b += 64) out[static_cast<size_t>(frame) * total + b] = fr[b];
}

// ---------------------------------------------------------------- inverse FFT by one wavefront (fft.cpp:96-128, inverse)
// buf[0..1023] natural-order frequency bins (plain layout) -> buf[i + (i>>4)] time samples (padded layout)
__device__ inline void ifft1024_wave(float2* buf, const float2* __restrict__ tw, int lane) {
    float2 x[16];
    {
        int rl = __brev(static_cast<unsigned>(lane)) >> 26;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int r4 = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
            x[r] = buf[rl + 64 * r4];
        }
    }
    wave_sync();
#pragma unroll
    for (int s = 1; s <= 4; ++s) {
        const int half = 1 << (s - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if ((r & half) == 0) bfly(x[r], x[r + half], conj_(tw[(r & (half - 1)) << (10 - s)]));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[17 * lane + r] = x[r];
    wave_sync();
    {
        const int a = lane & 15, hi = lane >> 4, base = a + 272 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = buf[base + 17 * r];
#pragma unroll
        for (int s = 5; s <= 8; ++s) {
            const int hr = 1 << (s - 5);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & hr) == 0) bfly(x[r], x[r + hr], conj_(tw[(a + 16 * (r & (hr - 1))) << (10 - s)]));
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[base + 17 * r] = x[r];
    }
    wave_sync();
    const float scale = 1.0f / 1024.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int b = lane + 64 * t, pa = b + (b >> 4);
        float2 y0 = buf[pa], y1 = buf[pa + 272], y2 = buf[pa + 544], y3 = buf[pa + 816];
        float2 w9 = conj_(tw[b << 1]);
        bfly(y0, y1, w9); bfly(y2, y3, w9);
        bfly(y0, y2, conj_(tw[b])); bfly(y1, y3, conj_(tw[b + 256]));
        buf[pa] = make_float2(y0.x * scale, y0.y * scale);
        buf[pa + 272] = make_float2(y1.x * scale, y1.y * scale);
        buf[pa + 544] = make_float2(y2.x * scale, y2.y * scale);
        buf[pa + 816] = make_float2(y3.x * scale, y3.y * scale);
    }
    wave_sync();
}

__device__ __forceinline__ float2 map_bits(uint32_t bits, int mod) {  // modulator.cpp:27-116
    switch (mod) {
        case RIA_MOD_BPSK: return make_float2((bits & 1) ? 1.0f : -1.0f, 0.0f);
        case RIA_MOD_QAM16: {
            const float lv[4] = {-3, -1, 3, 1};
            const float s = 0.3162277660168379f;
            return make_float2(lv[(bits >> 2) & 3] * s, lv[bits & 3] * s);
        }
        case RIA_MOD_QAM32: {
            const float s = 0.1961161351381840f;
            const float IL[4] = {-3, -1, 1, 3}, QL[8] = {-7, -5, -3, -1, 1, 3, 5, 7};
            const int IG[4] = {0, 1, 3, 2}, QG[8] = {0, 1, 3, 2, 6, 7, 5, 4};
            int qb = (bits >> 2) & 7, ib = bits & 3, qi = 0, ii = 0;
            for (int i = 0; i < 4; ++i) if (IG[i] == ib) { ii = i; break; }
            for (int i = 0; i < 8; ++i) if (QG[i] == qb) { qi = i; break; }
            return make_float2(IL[ii] * s, QL[qi] * s);
        }
        case RIA_MOD_QAM64: {
            const float lv[8] = {-7, -5, -1, -3, 7, 5, 1, 3};
            const float s = 0.1543033499620919f;
            return make_float2(lv[(bits >> 3) & 7] * s, lv[bits & 7] * s);
        }
        case RIA_MOD_QAM256: {
            const float lv[16] = {-15, -13, -9, -11, -1, -3, -7, -5, 15, 13, 9, 11, 1, 3, 7, 5};
            const float s = 0.0645497224367903f;
            return make_float2(lv[(bits >> 4) & 15] * s, lv[bits & 15] * s);
        }
        default: {
            const float s = 0.7071067811865476f;
            return make_float2((bits & 2) ? s : -s, (bits & 1) ? s : -s);
        }
    }
}

// One workgroup per frame.  LDS: 4 IFFT tiles + bit stream + differential state.
__global__ __launch_bounds__(256) void tx_frames_kernel(const TxConst* __restrict__ T, const float2* __restrict__ tw,
                                                        const float2* __restrict__ nco, const uint8_t* __restrict__ info,
                                                        int n_frames, float peak, int frame_samples,
                                                        float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* tiles = reinterpret_cast<float2*>(smem);
    uint8_t* bits = reinterpret_cast<uint8_t*>(tiles + 4 * kFftBufFloats2);   // [2592] interleaved coded bits
    uint8_t* cwbits = bits + 2592;                                            // [4][648] codeword bits
    float2* dstate = reinterpret_cast<float2*>(cwbits + 2592);                // [n_sym][64] differential symbols
    float* red = reinterpret_cast<float*>(dstate + 64 * 64);                  // [8] reduction scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frame = blockIdx.x;
    const int k = T->k, m = T->m, bpc = T->bytes_per_cw, mod = T->mod;
    const uint8_t* fi = info + static_cast<size_t>(frame) * 4 * bpc;
    // ---- LDPC encode: information bits, then parity[i] = XOR of the information bits in check i
    for (int idx = tid; idx < 4 * 648; idx += 256) {
        int cw = idx / 648, j = idx - cw * 648;
        int bit = 0;
        if (j < k && j < bpc * 8) bit = (fi[cw * bpc + (j >> 3)] >> (7 - (j & 7))) & 1;
        cwbits[idx] = static_cast<uint8_t>(bit);
    }
    __syncthreads();
    for (int idx = tid; idx < 4 * m; idx += 256) {
        int cw = idx / m, i = idx - cw * m;
        int deg = T->row_deg[i], p = 0;
        for (int s = 0; s < deg - 1; ++s) p ^= cwbits[cw * 648 + T->row_var[s * m + i]];
        cwbits[cw * 648 + k + i] = static_cast<uint8_t>(p);
    }
    __syncthreads();
    for (int idx = tid; idx < 4 * 648; idx += 256) bits[T->scatter[idx]] = cwbits[idx];
    __syncthreads();
    // ---- differential pre-pass (sequential over symbols, parallel over carriers)
    const int n_sym = 2 + T->n_data_symbols, bps = T->bits_per_symbol, bcar = T->bits_per_carrier;
    const bool diff = (mod == RIA_MOD_DBPSK || mod == RIA_MOD_DQPSK || mod == RIA_MOD_D8PSK);
    if (diff && tid < T->n_data) {
        float2 prev = make_float2(1.0f, 0.0f);
        for (int d = 0; d < T->n_data_symbols; ++d) {
            int b0 = d * bps + tid * bcar;
            if (b0 < 2592) {
                uint32_t v = 0;
                for (int b = 0; b < bcar; ++b) v = (v << 1) | ((b0 + b < 2592) ? bits[b0 + b] : 0);
                float2 pc;
                if (mod == RIA_MOD_DBPSK) pc = (v & 1) ? make_float2(-1, 0) : make_float2(1, 0);
                else if (mod == RIA_MOD_D8PSK) {   // natural-binary octant + 22.5 deg offset (modulator.cpp D8PSK branch)
                    const float pi_f = 3.14159265358979f;
                    const float angle = static_cast<float>(v & 7) * fdiv(pi_f, 4.0f) + fdiv(pi_f, 8.0f);
                    pc = make_float2(cosf_glibc(angle), sinf_glibc(angle));
                } else { const float pr[4] = {1, 0, -1, 0}, pi[4] = {0, 1, 0, -1}; pc = make_float2(pr[v & 3], pi[v & 3]); }
                prev = cmul(prev, pc);
                dstate[d * 64 + tid] = prev;
            } else dstate[d * 64 + tid] = make_float2(0, 0);
        }
    }
    __syncthreads();
    // ---- symbols: build spectrum, IFFT, cyclic prefix, upmix, x40
    float* fo = out + static_cast<size_t>(frame) * frame_samples;
    float2* buf = tiles + wave * kFftBufFloats2;
    float lmax = 0.0f;
    for (int s = wave; s < n_sym; s += 4) {
        for (int i = lane; i < 1024; i += 64) buf[i] = make_float2(0, 0);
        wave_sync();
        if (lane < T->n_data) {
            float2 v;
            if (s < 2) v = make_float2(T->sync_re[lane], T->sync_im[lane]);
            else if (diff) v = dstate[(s - 2) * 64 + lane];
            else {
                int b0 = (s - 2) * bps + lane * bcar;
                if (b0 < 2592) {
                    uint32_t bv = 0;
                    for (int b = 0; b < bcar; ++b) bv = (bv << 1) | ((b0 + b < 2592) ? bits[b0 + b] : 0);
                    v = map_bits(bv, mod);
                } else v = make_float2(0, 0);
            }
            buf[T->data_bin[lane]] = v;
        }
        if (lane < T->n_pilot) buf[T->pilot_bin[lane]] = make_float2(T->pilot_seq[lane], 0.0f);
        wave_sync();
        ifft1024_wave(buf, tw, lane);
        for (int i = lane; i < kSym; i += 64) {
            int j = (i < kCP) ? (kFFT - kCP + i) : (i - kCP);
            float2 v = buf[j + (j >> 4)];
            float2 o = nco[s * kSym + i];
            float re = v.x * o.x - v.y * o.y;   // (complex_signal * mixer.next()).real()
            float smp = re * 40.0f;
            fo[s * kSym + i] = smp;
            lmax = fmaxf(lmax, fabsf(smp));
        }
        wave_sync();
    }
    if (peak > 0.0f) {   // tools/test_waveform_simple.cpp:365-371: scale = 0.8 / max|s|
        for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
        if (lane == 0) red[wave] = lmax;
        __syncthreads();
        float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (mx > 0.0f) {
            float sc = fdiv(peak, mx);
            __threadfence_block();
            for (int i = tid; i < frame_samples; i += 256) fo[i] = fo[i] * sc;
        }
    }
}

inline int tx_lds_bytes() { return 4 * kFftBufFloats2 * 8 + 2592 * 2 + 64 * 64 * 8 + 64; }

inline void launch_make_frames(const TxConst* T, const uint16_t* crc_bit, const uint16_t* crc_init, uint64_t seed,
                               int first_seq, int n_frames, const ria_gpu_geometry&, uint8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(make_frames_kernel, dim3(n_frames), dim3(64), 0, s, T, crc_bit, crc_init, seed, first_seq,
                       n_frames, out);
}
inline void launch_tx(const TxConst* T, const float2* tw, const float2* nco, const uint8_t* info, int n_frames,
                      float peak, const ria_gpu_geometry& g, float* out, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tx_frames_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, tx_lds_bytes());
        attr = true;
    }
    hipLaunchKernelGGL(tx_frames_kernel, dim3(n_frames), dim3(256), tx_lds_bytes(), s, T, tw, nco, info, n_frames, peak,
                       g.frame_samples, out);
}

// ---------------------------------------------------------------- channel
// One workgroup per frame; the frame is staged in LDS (in-place operation with a delayed tap).
__global__ __launch_bounds__(256) void channel_kernel(int kind, float noise_gain, float delay_ms, float doppler_hz,
                                                      float g1, float g2, uint64_t seed, uint64_t first_frame,
                                                      float* __restrict__ samples, int frame_samples) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* xs = reinterpret_cast<float*>(smem);                // [frame_samples]
    float* red = xs + frame_samples;                           // [16]
    float4* aff = reinterpret_cast<float4*>(red + 16);         // [256] chunk-end fading states (f1, f2)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t frame = first_frame + blockIdx.x;
    float* fs = samples + static_cast<size_t>(blockIdx.x) * frame_samples;
    float p = 0.0f, c = 0.0f;
    for (int i = tid; i < frame_samples; i += 256) {
        float v = fs[i];
        xs[i] = v;
        if (fabsf(v) > 1e-6f) { p += v * v; c += 1.0f; }
    }
    for (int o = 32; o > 0; o >>= 1) { p += __shfl_xor(p, o); c += __shfl_xor(c, o); }
    if (lane == 0) { red[wave] = p; red[4 + wave] = c; }
    __syncthreads();
    p = red[0] + red[1] + red[2] + red[3];
    c = red[4] + red[5] + red[6] + red[7];
    const float rms = (c > 0.0f) ? sqrtf(p / c) : 0.1f;
    const float nstd = rms * noise_gain;                       // rms * 10^(-snr/20)
    const bool fading = kind != 0;
    const int delay = static_cast<int>(delay_ms * 48000 / 1000.0f);
    const uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
    const uint32_t fr_lo = static_cast<uint32_t>(frame), fr_hi = static_cast<uint32_t>(frame >> 32);
    const int chunk = (frame_samples + 255) / 256;
    const int n0 = tid * chunk, n1 = min(frame_samples, n0 + chunk);
    float alpha = 0.0f, a = 1.0f, gin = 0.0f;
    float4 st = make_float4(1.0f, 0.0f, 1.0f, 0.0f);           // fading1 (re, im), fading2 (re, im)
    if (fading) {
        alpha = 1.0f - expf(-6.283185307179586f * (doppler_hz / 48000.0f));
        a = 1.0f - alpha;
        gin = alpha * sqrtf(1.0f / alpha);
        // pass 1: zero-state response of this chunk and its decay
        float4 z = make_float4(0, 0, 0, 0);
        for (int n = n0; n < n1; ++n) {
            uint32_t r[4];
            float w0, w1, w2, w3;
            philox4x32(static_cast<uint32_t>(n), fr_lo, fr_hi, 1u, k0, k1, r);
            box_muller(r[0], r[1], &w0, &w1);
            box_muller(r[2], r[3], &w2, &w3);
            z.x = a * z.x + gin * w0; z.y = a * z.y + gin * w1; z.z = a * z.z + gin * w2; z.w = a * z.w + gin * w3;
        }
        aff[tid] = z;
        __syncthreads();
        if (tid == 0) {  // 256-step serial combine of the affine maps (cheap next to the per-sample work)
            float ac = powf(a, static_cast<float>(chunk));
            float4 s = make_float4(1.0f, 0.0f, 1.0f, 0.0f);
            for (int t = 0; t < 256; ++t) {
                float4 z2 = aff[t];
                aff[t] = s;  // state at the START of chunk t
                int len = min(frame_samples, (t + 1) * chunk) - t * chunk;
                float d = (len == chunk) ? ac : powf(a, static_cast<float>(len > 0 ? len : 0));
                s = make_float4(d * s.x + z2.x, d * s.y + z2.y, d * s.z + z2.z, d * s.w + z2.w);
            }
        }
        __syncthreads();
        st = aff[tid];
    }
    // pass 2: regenerate the same draws and emit
    for (int n = n0; n < n1; ++n) {
        uint32_t r[4];
        float h1 = 1.0f, h2 = 1.0f;
        if (fading) {
            float w0, w1, w2, w3;
            philox4x32(static_cast<uint32_t>(n), fr_lo, fr_hi, 1u, k0, k1, r);
            box_muller(r[0], r[1], &w0, &w1);
            box_muller(r[2], r[3], &w2, &w3);
            st.x = a * st.x + gin * w0; st.y = a * st.y + gin * w1; st.z = a * st.z + gin * w2; st.w = a * st.w + gin * w3;
            h1 = sqrtf(st.x * st.x + st.y * st.y);
            h2 = sqrtf(st.z * st.z + st.w * st.w);
        }
        float s = xs[n], o;
        if (fading && delay > 0) {
            float dl = (n >= delay) ? xs[n - delay] : 0.0f;
            o = s * g1 * h1 + dl * g2 * h2;
        } else {
            o = s * h1;
        }
        float nz0, nz1;
        philox4x32(static_cast<uint32_t>(n), fr_lo, fr_hi, 2u, k0, k1, r);
        box_muller(r[0], r[1], &nz0, &nz1);
        fs[n] = o + nstd * nz0;
    }
}

inline void launch_channel(int kind, float snr_db, uint64_t seed, uint64_t first_frame, float* samples, int n_frames,
                           int frame_samples, hipStream_t s) {
    float delay_ms = 0, doppler = 0, g1 = 1.0f, g2 = 0.0f;  // hf_channel.hpp:411-488
    switch (kind) {
        case 1: delay_ms = 0.5f; doppler = 0.1f; g1 = g2 = 0.707f; break;
        case 2: delay_ms = 1.0f; doppler = 0.5f; g1 = g2 = 0.707f; break;
        case 3: delay_ms = 2.0f; doppler = 1.0f; g1 = g2 = 0.707f; break;
        case 4: delay_ms = 0.5f; doppler = 10.0f; g1 = g2 = 0.707f; break;
        default: break;
    }
    float noise_gain = powf(10.0f, -snr_db / 20.0f);
    int lds = frame_samples * 4 + 64 + 256 * 16 + 64;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(channel_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(channel_kernel, dim3(n_frames), dim3(256), lds, s, kind, noise_gain, delay_ms, doppler, g1, g2,
                       seed, first_frame, samples, frame_samples);
}

// ---------------------------------------------------------------- channel, reference-identical
// sim::WattersonChannel::process with the reference's own random stream (hf_channel.hpp:107-177, :267-284):
// std::mt19937(seed) -> std::normal_distribution<float> (libstdc++ Marsaglia polar, pairs), five draws per
// sample in the order imag/real of tap 1, imag/real of tap 2, AWGN.  One wavefront per frame:
//   * the normal stream is produced a twist (312 candidate pairs) at a time by all lanes, acceptance ranked
//     with ballot prefix counts so that the accepted values land in draw order (as normal648_wave does);
//   * the four fading recurrences f = (1-a) f + a n are independent serial chains: all lanes form the driving terms
//     a (ns n) of a tile, lanes 0..3 then walk it with two dependent operations per sample (16-byte LDS loads / stores);
//   * mixing, the delayed tap and the AWGN add are one lane per sample; the signal power that scales the noise is a
//     left-to-right sum over the whole frame, i.e. a chain of 18 432 dependent additions: channel_power_kernel runs it
//     ahead of this kernel with one LANE per frame (64 frames per wave, tiles transposed through LDS), so that the chain
//     costs one wave instruction per 64 frames instead of per frame.
// Output is bit-identical to the reference channel for the same (preset, SNR, seed).
#ifndef RIA_CHAN_TILE
#define RIA_CHAN_TILE 64    // samples per tile: 9.5 KB of LDS per wave, 16 waves per CU; per 32 768 faded frames 512: 48 ms (5 waves per CU), 256: 29, 128: 24, 64: 20
#endif
constexpr int kChanTile = RIA_CHAN_TILE, kChanNbuf = 5 * kChanTile + 704;
struct ChanExactArgs {
    float* samples; long long stride; int frame_samples; int n_frames;
    int fading, multipath, delay;
    float alpha, one_minus_alpha, ns, g1, g2, noise_gain;
    uint32_t seed; uint64_t first_frame;
    const uint32_t* seeds;   // nullable: per-frame mt19937 seeds (then seed / first_frame are not used)
    const float* nstd;       // [n_frames] noise sigma of every frame (channel_power_kernel)
    // CFO impairment of the channel object (hf_channel.hpp:47-51 Config::cfo_hz / random_cfo_max_hz, :97-102, :172-241)
    const float* cfo_hz;     // nullable: per-frame Config::cfo_hz (else cfo_all)
    float cfo_all;
    float random_cfo_max;    // > 0: the frame's CFO is the constructor's uniform draw from the frame's own generator
    float* actual_cfo_out;   // nullable [n_frames]: getActualCFO()
};
__host__ __device__ inline int chan_exact_lds_bytes() { return 624 * 4 + kChanNbuf * 4 + 4 * kChanTile * 4 + kChanTile * 4 + 128 * 4 + 256 * 4 + 64; }

// process() :115-131: signal power over the samples with |x| > 1e-6 (left to right), rms, noise sigma = rms * 10^(-snr/20).
// One lane per frame: a wave takes 64 frames, 64 samples of each at a time, transposed through LDS (row stride 65).
constexpr int kPowRow = 65;
__global__ __launch_bounds__(64) void channel_power_kernel(const float* __restrict__ samples, long long stride, int n, int n_frames,
                                                           float noise_gain, float* __restrict__ nstd) {
    __shared__ float T[64 * kPowRow];
    const int lane = threadIdx.x, f0 = blockIdx.x * 64;
    const int rows = (n_frames - f0 < 64) ? n_frames - f0 : 64;
    float power = 0.0f; int cnt = 0;
    float xv[64];
    auto fetch = [&](int base) {
        const int i = base + lane;
#pragma unroll
        for (int r = 0; r < 64; ++r) xv[r] = (r < rows && i < n) ? samples[static_cast<long long>(f0 + r) * stride + i] : 0.0f;
    };
    fetch(0);
    for (int base = 0; base < n; base += 64) {
#pragma unroll
        for (int r = 0; r < 64; ++r) T[r * kPowRow + lane] = xv[r];
        wave_sync();
        if (base + 64 < n) fetch(base + 64);                // in flight during the chain below
#pragma unroll 16
        for (int i = 0; i < 64; ++i) {
            const float v = T[lane * kPowRow + i];
            const bool c = fabs_(v) > 1e-6f;                  // zero padding never passes
            power += c ? v * v : 0.0f;                        // + (+0.0f) is exact: power is never -0.0
            cnt += c ? 1 : 0;
        }
        wave_sync();
    }
    if (lane < rows) {
        const float rms = cnt ? fsqrt(fdiv(power, static_cast<float>(cnt))) : 0.1f;
        nstd[f0 + lane] = rms * noise_gain;
    }
}

__global__ __launch_bounds__(64) void channel_exact_kernel(ChanExactArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* st = reinterpret_cast<uint32_t*>(smem);
    float* nbuf = reinterpret_cast<float*>(st + 624);
    float* fr = nbuf + kChanNbuf;              // [4][tile]
    float* xt = fr + 4 * kChanTile;            // original samples of the tile
    float* hist = xt + kChanTile;              // last delay+1 original samples of the previous tile
    float* tmp = hist + 128;                   // [256]
    const int lane = threadIdx.x, n = A.frame_samples;
    float* x = A.samples + static_cast<long long>(blockIdx.x) * A.stride;
    const float nstd = A.nstd[blockIdx.x];
    mt_seed_wave(st, A.seeds ? A.seeds[blockIdx.x] : A.seed + static_cast<uint32_t>(A.first_frame + blockIdx.x), lane);
    for (int i = lane; i < 128; i += 64) hist[i] = 0.0f;
    // The constructor's CFO draw (hf_channel.hpp:97-102) takes the generator's FIRST word, so the normal
    // distribution's pairs start at word 1 and straddle the twists: (623 of one twist, 0 of the next).
    const bool odd = A.random_cfo_max > 0.0f;
    bool first_twist = true;
    uint32_t carry = 0u;
    float actual_cfo = A.cfo_hz ? A.cfo_hz[blockIdx.x] : A.cfo_all;
    // applyCFO state (hf_channel.hpp:182-241): lanes 0/1 own the running sums of I/Q, lane 2 the CFO phase
    float* ibt = xt;                           // tile of I_bb (xt is free once the delayed-tap history is saved)
    float* qbt = fr + 3 * kChanTile;           // tile of Q_bb; fr[0..3*tile) = running sums I, Q and the phase per sample
    float* ibh = tmp + 128;                    // last 48 I_bb / Q_bb of the previous tile
    float* qbh = tmp + 176;
    float run = 0.0f;
    int have = 0;
    const int per = A.fading ? 5 : 1;
    float fc = (lane == 0 || lane == 2) ? 1.0f : 0.0f;       // lanes 0..3: f1.re, f1.im, f2.re, f2.im
    for (int base = 0; base < n; base += kChanTile) {
        const int tn = (n - base < kChanTile) ? n - base : kChanTile;
        const int need = per * tn;
        while (have < need) {
            mt_twist_wave(st, lane);
            if (odd && first_twist)   // uniform_real_distribution<float>(-max, max): canonical * (b - a) + a
                actual_cfo = mt_canonical(st[0]) * (A.random_cfo_max - (-A.random_cfo_max)) + (-A.random_cfo_max);
            for (int pb = 0; pb < 312; pb += 64) {
                const int p = pb + lane;
                bool acc = false;
                float u = 0.f, v = 0.f, r2 = 1.f;
                if (p < 312 && !(odd && first_twist && p == 0)) {
                    const uint32_t wu = !odd ? st[2 * p] : (p == 0) ? carry : st[2 * p - 1];
                    const uint32_t wv = !odd ? st[2 * p + 1] : st[2 * p];
                    u = 2.0f * mt_canonical(wu) - 1.0f;
                    v = 2.0f * mt_canonical(wv) - 1.0f;
                    r2 = u * u + v * v;
                    acc = !(r2 > 1.0f || r2 == 0.0f);
                }
                const unsigned long long mask = __ballot(acc);
                if (acc) {
                    const int q = have + 2 * __popcll(mask & ((1ull << lane) - 1ull));
                    const float mult = fsqrt(fdiv(-2.0f * logf_glibc(r2), r2));
                    nbuf[q] = v * mult;          // returned first
                    nbuf[q + 1] = u * mult;      // saved, returned by the next call
                }
                have += 2 * __popcll(mask);
            }
            if (odd) carry = st[623];
            first_twist = false;
            wave_sync();
        }
        for (int i = lane; i < tn; i += 64) xt[i] = x[base + i];
        wave_sync();
        if (A.fading) {
            // driving terms alpha * (ns * g) of the four recurrences by all lanes, then lanes 0..3 walk their row in place
            static_assert(kChanTile % 64 == 0 && kChanTile % 4 == 0, "tile = whole waves of samples");
#pragma unroll
            for (int q = 0; q < kChanTile / 64; ++q) {
                const int i = 64 * q + lane;
                if (i < tn) {
                    const float g0 = nbuf[5 * i + 1], g1 = nbuf[5 * i], g2 = nbuf[5 * i + 3], g3 = nbuf[5 * i + 2];   // imag is drawn before real
                    fr[i] = A.alpha * (A.ns * g0); fr[kChanTile + i] = A.alpha * (A.ns * g1);
                    fr[2 * kChanTile + i] = A.alpha * (A.ns * g2); fr[3 * kChanTile + i] = A.alpha * (A.ns * g3);
                }
            }
            wave_sync();
            if (lane < 4) {
                float* row = fr + lane * kChanTile;
                const float oma = A.one_minus_alpha;
                int i = 0;
                for (; i + 4 <= tn; i += 4) {
                    float4 b = *reinterpret_cast<const float4*>(row + i);
                    fc = oma * fc + b.x; b.x = fc;
                    fc = oma * fc + b.y; b.y = fc;
                    fc = oma * fc + b.z; b.z = fc;
                    fc = oma * fc + b.w; b.w = fc;
                    *reinterpret_cast<float4*>(row + i) = b;
                }
                for (; i < tn; ++i) { fc = oma * fc + row[i]; row[i] = fc; }
            }
            wave_sync();
        }
        for (int i = lane; i < tn; i += 64) {
            const float sv = xt[i];
            float h1 = 1.0f, h2 = 1.0f;
            if (A.fading) { h1 = hypotf_glibc(fr[i], fr[kChanTile + i]); h2 = hypotf_glibc(fr[2 * kChanTile + i], fr[3 * kChanTile + i]); }
            float o = 0.0f;
            if (A.multipath && A.delay > 0) {
                o += sv * A.g1 * h1;
                const int j = i - A.delay - 1;
                const float delayed = (j >= 0) ? xt[j] : hist[128 + j];     // j in [-(delay+1), -1]
                o += delayed * A.g2 * h2;
            } else {
                o = sv * h1;
            }
            o += nstd * nbuf[per * i + (per - 1)];
            x[base + i] = o;
        }
        wave_sync();
        // history for the next tile's delayed tap, leftover normals to the front
        for (int i = lane; i < 128; i += 64) { const int j = tn - 128 + i; tmp[i] = (j >= 0) ? xt[j] : hist[(i + tn < 128) ? i + tn : 127]; }
        wave_sync();
        for (int i = lane; i < 128; i += 64) hist[i] = tmp[i];
        const int left = have - need;
        for (int b0 = 0; b0 < left; b0 += 64) {
            const int i = b0 + lane;
            const float t = (i < left) ? nbuf[need + i] : 0.0f;
            wave_sync();
            if (i < left) nbuf[i] = t;
            wave_sync();
        }
        have = left;
        wave_sync();
        // ---- applyCFO on this tile (process() :172-174: |actual CFO| > 0.001 Hz; applyCFO returns early below 256 samples)
        if (fabs_(actual_cfo) > 0.001f && n >= 256) {
            const float inc = static_cast<float>(static_cast<double>(2.0f) * 3.14159265358979323846 * static_cast<double>(actual_cfo) / static_cast<double>(48000u));
            float cm[kChanTile / 64], sm[kChanTile / 64];
#pragma unroll
            for (int q = 0; q < kChanTile / 64; ++q) {           // mix to baseband at fc = 1500 Hz (:195-200)
                const int i = 64 * q + lane;
                cm[q] = 0.0f; sm[q] = 0.0f;
                if (i < tn) {
                    const float t = fdiv(static_cast<float>(base + i), 48000.0f);
                    const float mp = static_cast<float>(static_cast<double>(2.0f) * 3.14159265358979323846 * static_cast<double>(1500.0f) * static_cast<double>(t));
                    cm[q] = cosf_glibc(mp); sm[q] = sinf_glibc(mp);
                    const float o = x[base + i];                 // this lane's own store of the stage above
                    ibt[i] = o * cm[q]; qbt[i] = o * sm[q];
                }
            }
            wave_sync();
            if (lane < 2) {                                      // 48-tap moving average as running sums (:204-217): serial by definition
                const float* cur = lane ? qbt : ibt;
                const float* old = lane ? qbh : ibh;
                float* dst = fr + lane * kChanTile;
                for (int i = 0; i < tn; ++i) {
                    run += cur[i];
                    if (base + i >= 48) run -= (i >= 48) ? cur[i - 48] : old[i];
                    dst[i] = run;
                }
            } else if (lane == 2) {                              // CFO phase walk (:223-238): wraps only downwards
                float* dst = fr + 2 * kChanTile;
                for (int i = 0; i < tn; ++i) {
                    dst[i] = run;
                    run += inc;
                    if (static_cast<double>(run) > static_cast<double>(2.0f) * 3.14159265358979323846)
                        run = static_cast<float>(static_cast<double>(run) - static_cast<double>(2.0f) * 3.14159265358979323846);
                }
            }
            wave_sync();
#pragma unroll
            for (int q = 0; q < kChanTile / 64; ++q) {
                const int i = 64 * q + lane;
                if (i < tn) {
                    const int m = (base + i + 1 < 48) ? base + i + 1 : 48;
                    const float If = fdiv(fr[i], static_cast<float>(m)), Qf = fdiv(fr[kChanTile + i], static_cast<float>(m));
                    const float ph = fr[2 * kChanTile + i];
                    const float cc = cosf_glibc(ph), cs = sinf_glibc(ph);
                    const float Ic = If * cc - Qf * cs, Qc = If * cs + Qf * cc;
                    x[base + i] = 2.0f * (Ic * cm[q] - Qc * sm[q]);
                }
            }
            wave_sync();
            if (lane < 48 && tn == kChanTile) { const float a = ibt[kChanTile - 48 + lane], b = qbt[kChanTile - 48 + lane]; ibh[lane] = a; qbh[lane] = b; }
            wave_sync();
        }
    }
    if (A.actual_cfo_out && lane == 0) A.actual_cfo_out[blockIdx.x] = actual_cfo;
}

inline void launch_channel_exact(int kind, float snr_db, uint32_t seed, uint64_t first_frame, float* samples, long long stride,
                                 int frame_samples, int n_frames, hipStream_t s, float* nstd_ws, const uint32_t* seeds = nullptr,
                                 const float* cfo_hz = nullptr, float cfo_all = 0.0f, float random_cfo_max = 0.0f, float* actual_cfo_out = nullptr) {
    float delay_ms = 0, doppler = 0, g1 = 1.0f, g2 = 0.0f;  // presets hf_channel.hpp:411-488
    int fading = 1, multipath = 1;
    switch (kind) {
        case 0: fading = 0; multipath = 0; break;
        case 1: delay_ms = 0.5f; doppler = 0.1f; g1 = g2 = 0.707f; break;
        case 2: delay_ms = 1.0f; doppler = 0.5f; g1 = g2 = 0.707f; break;
        case 3: delay_ms = 2.0f; doppler = 1.0f; g1 = g2 = 0.707f; break;
        default: delay_ms = 0.5f; doppler = 10.0f; g1 = g2 = 0.707f; break;
    }
    ChanExactArgs A{};
    A.samples = samples; A.stride = stride; A.frame_samples = frame_samples; A.n_frames = n_frames;
    A.fading = fading; A.multipath = multipath;
    A.delay = static_cast<int>(delay_ms * 48000 / 1000.0f);
    const float norm_dopp = doppler / 48000;
    A.alpha = static_cast<float>(1.0f - std::exp(-2.0f * 3.14159265358979323846 * static_cast<double>(norm_dopp)));   // hf_channel.hpp:84-90
    A.one_minus_alpha = 1.0f - A.alpha;
    A.ns = fading ? std::sqrt(1.0f / A.alpha) : 0.0f;
    A.g1 = g1; A.g2 = g2;
    A.noise_gain = powf(10.0f, -snr_db / 20.0f);
    A.seed = seed; A.first_frame = first_frame; A.seeds = seeds;
    A.cfo_hz = cfo_hz; A.cfo_all = cfo_all; A.random_cfo_max = random_cfo_max; A.actual_cfo_out = actual_cfo_out;
    A.nstd = nstd_ws;
    hipLaunchKernelGGL(channel_power_kernel, dim3((n_frames + 63) / 64), dim3(64), 0, s, samples, stride, frame_samples, n_frames, A.noise_gain, nstd_ws);
    hipLaunchKernelGGL(channel_exact_kernel, dim3(n_frames), dim3(64), chan_exact_lds_bytes(), s, A);
}

}  // namespace ria
