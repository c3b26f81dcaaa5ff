// ria_amd/csrc/ldpc_kernels.hip.h — flooding normalised min-sum LDPC(648,k) decoder for gfx950.
//
// Replaces LDPCDecoder::decodeSoft/decodeBP (src/fec/ldpc_decoder.cpp:154-260) and the per-codeword
// part of v2::decodeFixedFrame (src/protocol/frame_v2.cpp:1335-1563), bit-exactly:
//   * one 64-lane wavefront owns one codeword; a 256-thread workgroup owns the 4 codewords of a frame;
//   * all messages live in LDS in a [slot][check] layout (slot s of check i at s*m+i) so the
//     check-node pass reads/writes consecutive addresses across lanes (bank-conflict free);
//   * the check-node pass computes sign-product / min1 / min2 once per check (O(deg)) instead of the
//     reference's O(deg^2) loop; the result per edge is the same float
//     (sign * min_abs) * factor, one rounding;
//   * the variable-node pass accumulates check messages in ASCENDING CHECK ORDER, the order the
//     reference's row-major loop produces (ldpc_decoder.cpp:206-215) — float addition order matters;
//   * messages are updated in place: after the check pass a slot holds c2v, after the variable
//     pass it holds v2c = clamp(total - c2v, +-50).
// MFMA is not used: there is no dense contraction in min-sum.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ria_gpu.h"
#include "devmath.h"

namespace ria {

constexpr int kMaxRowDegDev = 7;  // 6 information columns at most + the identity column

struct LdpcDev {
    int k, m, n, max_col_deg, max_iter, bytes_per_cw;
    const uint8_t* row_deg;    // [m]
    const uint16_t* row_var;   // [7][m]
    const uint8_t* col_deg;    // [n]
    const uint16_t* col_slot;  // [max_col_deg][n]
};

// LDS bytes one wave needs for a codeword of this code
// (the message area doubles as mt19937 state + 648 normals during the retry cascade: >= 1296 words)
__host__ __device__ inline int ldpc_msg_floats(int m) { return (7 * m > 1296) ? 7 * m : 1296; }
__host__ __device__ inline int ldpc_wave_lds_bytes(int m) { return (ldpc_msg_floats(m) + 648) * 4 + 656; }
constexpr int kFrameSharedBytes = 640;  // decoded bytes [4][68] + result words + flat frame copy

__device__ __forceinline__ void wave_sync() {
    // One wave's DS operations execute in issue order, so cross-lane visibility inside the wave
    // needs no hardware wait; the fence keeps the compiler from moving LDS accesses across it.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Decodes the 648 LLRs in `llr` (LDS, decoder order).  Returns the iteration index at which all
// parities were satisfied (LDPCDecoder::lastIterations(): 0-based), or max_iter; *ok says which.
// After return hard[j] = (total[j] < 0) of the last iteration.
__device__ inline int ldpc_decode_wave(const LdpcDev& c, float* __restrict__ msg, const float* __restrict__ llr,
                                       uint8_t* __restrict__ hard, float factor, int max_iter, int lane, bool* ok) {
    const int m = c.m, n = c.n;
    for (int i = lane; i < m; i += 64) {
        int deg = c.row_deg[i];
        for (int s = 0; s < deg; ++s) msg[s * m + i] = llr[c.row_var[s * m + i]];
    }
    wave_sync();
    int it = 0;
    bool success = false;
    for (; it < max_iter; ++it) {
        // ---- check nodes: c2v[e] = (prod_{e'!=e} sgn v[e']) * min_{e'!=e} |v[e']| * factor
        for (int i = lane; i < m; i += 64) {
            int deg = c.row_deg[i];
            float v[kMaxRowDegDev];
            float min1 = 3.402823466e+38f, min2 = 3.402823466e+38f;
            int arg = -1, neg = 0;
#pragma unroll
            for (int s = 0; s < kMaxRowDegDev; ++s) {
                if (s < deg) {
                    float x = msg[s * m + i];
                    v[s] = x;
                    neg ^= (x < 0.0f) ? 1 : 0;
                    float a = fabs_(x);
                    if (a < min1) { min2 = min1; min1 = a; arg = s; }
                    else if (a < min2) { min2 = a; }
                }
            }
#pragma unroll
            for (int s = 0; s < kMaxRowDegDev; ++s) {
                if (s < deg) {
                    int sg = neg ^ ((v[s] < 0.0f) ? 1 : 0);
                    float mn = (s == arg) ? min2 : min1;
                    float sm = sg ? -mn : mn;  // sign * min_abs (exact)
                    msg[s * m + i] = sm * factor;
                }
            }
        }
        wave_sync();
        // ---- variable nodes: total = llr + sum c2v (ascending check order); v2c = clamp(total - c2v)
        for (int j = lane; j < n; j += 64) {
            int deg = c.col_deg[j];
            float tot = llr[j];
            for (int d = 0; d < deg; ++d) tot += msg[c.col_slot[d * n + j]];
            for (int d = 0; d < deg; ++d) {
                int a = c.col_slot[d * n + j];
                float x = tot - msg[a];
                x = (x < 50.0f) ? x : 50.0f;    // std::min(50.0f, x)
                x = (-50.0f < x) ? x : -50.0f;  // std::max(-50.0f, .)
                msg[a] = x;
            }
            hard[j] = (tot < 0.0f) ? 1 : 0;
        }
        wave_sync();
        // ---- syndrome
        int syn = 0;
        for (int i = lane; i < m; i += 64) {
            int deg = c.row_deg[i], p = 0;
            for (int s = 0; s < deg; ++s) p ^= hard[c.row_var[s * m + i]];
            syn |= p;
        }
        if (__ballot(syn != 0) == 0ull) { success = true; break; }
    }
    *ok = success;
    return it;
}

// Packs the first k hard bits MSB-first into out (ceil(k/8) bytes), as decodeBP does.
__device__ inline void ldpc_pack_info(const LdpcDev& c, const uint8_t* hard, uint8_t* out, int nbytes, int lane) {
    for (int b = lane; b < nbytes; b += 64) {
        int v = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            int j = 8 * b + t;
            int bit = (j < c.k) ? hard[j] : 0;
            v |= bit << (7 - t);
        }
        out[b] = static_cast<uint8_t>(v);
    }
}

// ------------------------------------------------------------------------------------------------
// Kernel: raw codeword decode (LDPCDecoder::decodeSoft on rows of 648 LLRs). One wave per codeword.
__global__ __launch_bounds__(256) void ldpc_decode_rows_kernel(LdpcDev c, const float* __restrict__ llr, int n_cw,
                                                               int max_iter, float factor, uint8_t* __restrict__ out,
                                                               uint8_t* __restrict__ ok_out,
                                                               uint16_t* __restrict__ iters_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cw = blockIdx.x * 4 + wave;
    if (cw >= n_cw) return;
    unsigned char* base = smem + static_cast<size_t>(wave) * ldpc_wave_lds_bytes(c.m);
    float* msg = reinterpret_cast<float*>(base);
    float* l = msg + ldpc_msg_floats(c.m);
    uint8_t* hard = reinterpret_cast<uint8_t*>(l + 648);
    for (int i = lane; i < 648; i += 64) l[i] = (i < c.n) ? llr[static_cast<size_t>(cw) * 648 + i] : 0.0f;
    wave_sync();
    bool ok;
    int it = ldpc_decode_wave(c, msg, l, hard, factor, max_iter, lane, &ok);
    int nb = (c.k + 7) / 8;
    ldpc_pack_info(c, hard, out + static_cast<size_t>(cw) * nb, nb, lane);
    if (lane == 0) { ok_out[cw] = ok ? 1 : 0; iters_out[cw] = static_cast<uint16_t>(it); }
}


// ------------------------------------------------------------------------------------------------
// std::mt19937 + libstdc++ std::normal_distribution<float> on one wavefront.
// The retry cascade perturbs LLRs with noise drawn exactly as the reference draws it
// (frame_v2.cpp:1431-1436): a fresh mt19937(seed) and normal_distribution per attempt, 648 values.
// state: 624 words in LDS.  Seeding is a serial recurrence (lane 0); the twist runs in three
// data-independent chunks across the lanes; the Marsaglia polar rejection consumes draws in aligned
// pairs, so acceptance is decided per pair in parallel and ranked with a ballot prefix count.
__device__ inline void mt_seed_wave(uint32_t* st, uint32_t seed, int lane) {
    if (lane == 0) {
        uint32_t x = seed;
        st[0] = x;
        for (int i = 1; i < 624; ++i) {
            x = 1812433253u * (x ^ (x >> 30)) + static_cast<uint32_t>(i);
            st[i] = x;
        }
    }
    wave_sync();
}
__device__ inline void mt_twist_wave(uint32_t* st, int lane) {
    auto step = [&](int i, int a, int b) {
        uint32_t y = (st[i] & 0x80000000u) | (st[a] & 0x7fffffffu);
        uint32_t v = st[b] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        return v;
    };
    for (int base = 0; base < 227; base += 64) {
        int i = base + lane;
        uint32_t v = 0;
        if (i < 227) v = step(i, i + 1, i + 397);
        wave_sync();
        if (i < 227) st[i] = v;
        wave_sync();
    }
    for (int base = 227; base < 623; base += 64) {
        int i = base + lane;
        uint32_t v = 0;
        if (i < 623) v = step(i, i + 1, i - 227);
        wave_sync();
        if (i < 623) st[i] = v;
        wave_sync();
    }
    if (lane == 0) st[623] = step(623, 0, 396);
    wave_sync();
}
__device__ __forceinline__ float mt_canonical(uint32_t y) {  // tempering + generate_canonical<float,24>
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    float r = __uint2float_rn(y) * 2.3283064365386963e-10f;  // / 2^32, exact scaling
    return (r >= 1.0f) ? u2f(0x3f7fffffu) : r;
}
// Fills normal[0..647] with the first 648 values of normal_distribution<float>(0,1) on mt19937(seed).
__device__ inline void normal648_wave(uint32_t* st, float* normal, uint32_t seed, int lane) {
    mt_seed_wave(st, seed, lane);
    int accepted = 0;  // accepted pairs so far (wave-uniform)
    while (accepted < 324) {
        mt_twist_wave(st, lane);
        for (int base = 0; base < 312 && accepted < 324; base += 64) {
            int p = base + lane;
            bool acc = false;
            float x = 0.f, y = 0.f, r2 = 1.f;
            if (p < 312) {
                x = 2.0f * mt_canonical(st[2 * p]) - 1.0f;
                y = 2.0f * mt_canonical(st[2 * p + 1]) - 1.0f;
                r2 = x * x + y * y;
                acc = !(r2 > 1.0f || r2 == 0.0f);
            }
            unsigned long long mask = __ballot(acc);
            int q = accepted + __popcll(mask & ((1ull << lane) - 1ull));
            if (acc && q < 324) {
                float mult = fsqrt(fdiv(-2.0f * logf_glibc(r2), r2));
                normal[2 * q] = y * mult;      // returned first
                normal[2 * q + 1] = x * mult;  // the saved value, returned by the next call
            }
            accepted += __popcll(mask);
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// Kernel: v2::decodeFixedFrame for a batch of frames.  One workgroup (4 waves) per frame.
struct FrameDecodeArgs {
    LdpcDev c;
    const uint16_t* gather;    // [4*648] both de-interleavers folded
    const float* llr;          // frame f at llr + f*llr_stride
    int llr_stride;
    int n_frames;
    uint32_t flags;
    uint8_t* info_out;         // [n_frames][4*bytes_per_cw]
    ria_decode_status* status; // [n_frames]
    const uint16_t* crc_bit;   // [1280+16] CRC-16 delta of the bit at distance q from the end
    const uint16_t* crc_init;  // [161+] CRC-16 (init 0xFFFF) of L zero bytes
};

__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o);
    return v;
}
// CRC-16/CCITT-FALSE (ControlFrame::calculateCRC, frame_v2.cpp:115-128) of d[0..L) by one wave.
__device__ inline uint32_t crc16_wave(const uint8_t* d, int L, const uint16_t* crc_bit, const uint16_t* crc_init,
                                      int lane) {
    uint32_t acc = 0;
    for (int i = lane; i < L; i += 64) {
        int b = d[i], q0 = (L - 1 - i) * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (b & (1 << t)) acc ^= crc_bit[q0 + t];
    }
    return (wave_xor(acc) ^ crc_init[L]) & 0xffffu;
}

__device__ inline void retry_transform_params(int a, uint32_t h, uint32_t* seed, float* sigma, float* factor, int* kind) {
    const float s1[15] = {0.3f, 0.7f, 0.3f, 1.0f, 0.5f, 1.5f, 0.3f, 2.0f, 0.5f, 0.7f, 1.0f, 2.5f, 0.3f, 1.5f, 0.5f};
    const float f1[15] = {0.75f, 0.625f, 0.875f, 0.75f, 0.625f, 0.75f, 0.5f, 0.625f, 0.875f, 0.75f, 0.625f, 0.875f, 0.75f, 0.5f, 0.625f};
    const float s2[5] = {0.3f, 0.8f, 1.5f, 2.5f, 4.0f};
    const float s3[3] = {0.5f, 1.5f, 3.0f};
    const float s5[5] = {0.0f, 0.2f, 0.5f, 1.0f, 1.5f};
    const float s6[3] = {0.3f, 1.0f, 2.0f};
    if (a < 15) { *kind = 0; *sigma = s1[a]; *factor = f1[a]; *seed = h + static_cast<uint32_t>(a * 997 + a * 31); }
    else if (a < 20) { int t = a - 15; *kind = 1; *sigma = s2[t]; *factor = (t % 2 == 0) ? 0.625f : 0.875f; *seed = h + static_cast<uint32_t>(a * 997 + 12345); }
    else if (a < 23) { *kind = 2; *sigma = s3[a - 20]; *factor = 0.875f; *seed = h + static_cast<uint32_t>(a * 997 + 54321); }
    else if (a < 26) { *kind = 3; *sigma = s3[a - 23]; *factor = 0.875f; *seed = h + static_cast<uint32_t>(a * 997 + 99999); }
    else if (a < 31) { *kind = 4; *sigma = s5[a - 26]; *factor = 0.875f; *seed = h + static_cast<uint32_t>(a * 997 + 33333); }
    else { *kind = 5; *sigma = s6[a - 31]; *factor = 0.875f; *seed = h + static_cast<uint32_t>(a * 997 + 77777); }
}

__global__ __launch_bounds__(256) void decode_frames_kernel(FrameDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdpcDev& c = A.c;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frame = blockIdx.x;
    const int wave_bytes = ldpc_wave_lds_bytes(c.m);
    unsigned char* base = smem + static_cast<size_t>(wave) * wave_bytes;
    float* msg = reinterpret_cast<float*>(base);
    float* cur = msg + ldpc_msg_floats(c.m);  // decoder input of the current attempt
    uint8_t* hard = reinterpret_cast<uint8_t*>(cur + 648);
    unsigned char* shared = smem + 4 * static_cast<size_t>(wave_bytes);
    uint8_t* cwd = shared;                                      // [4][68] decoded info bytes
    volatile int* res = reinterpret_cast<volatile int*>(shared + 272);  // [4][4]: primary_ok, phase0_ok, final_ok, iters
    const int bpc = c.bytes_per_cw;

    // decoder-order LLRs of this codeword stay in registers for all attempts
    float mine[11];
    const float* fl = A.llr + static_cast<size_t>(frame) * A.llr_stride;
#pragma unroll
    for (int r = 0; r < 11; ++r) {
        int i = r * 64 + lane;
        mine[r] = (i < 648) ? fl[A.gather[wave * 648 + i]] : 0.0f;
    }
    auto load_cur = [&]() {
#pragma unroll
        for (int r = 0; r < 11; ++r) { int i = r * 64 + lane; if (i < 648) cur[i] = mine[r]; }
        wave_sync();
    };

    bool ok = false, p_ok = false, p0_ok = false;
    int iters = 0, attempts = 0;
    auto primary_and_phase0 = [&](float f_before) {
        load_cur();
        attempts = 1;
        iters = ldpc_decode_wave(c, msg, cur, hard, f_before, c.max_iter, lane, &ok);
        p_ok = ok; p0_ok = false;
        if (!ok && (A.flags & RIA_DECODE_PHASE0)) {
            const float f0[4] = {0.875f, 0.75f, 0.625f, 0.5f};
            for (int t = 0; t < 4 && !ok; ++t) {
                bool k2; attempts++;
                int it = ldpc_decode_wave(c, msg, cur, hard, f0[t], c.max_iter, lane, &k2);
                if (k2) { ok = true; iters = it; }
            }
            p0_ok = ok;
        }
        if (lane == 0) { res[wave * 4 + 0] = p_ok; res[wave * 4 + 1] = p0_ok; }
    };
    // Speculative pass: every codeword assumes the decoder still has its default factor.
    primary_and_phase0(0.9375f);
    __syncthreads();
    // The reference decodes the 4 codewords in order on ONE decoder object whose min-sum factor is
    // restored to 0.9375 only after phase 0 and left at 0.875 by phases 1-2
    // (frame_v2.cpp:1359-1361,1412,1447,1470): a codeword that follows one that needed phase >= 1
    // starts at 0.875.  Resolve that chain in order; redo the (rare) mis-speculated codewords.
    float f = 0.9375f;
    for (int cw = 0; cw < 4; ++cw) {
        if (cw > 0) {
            if (wave == cw && f != 0.9375f) primary_and_phase0(f);
            __syncthreads();
        }
        bool pk = res[cw * 4 + 0] != 0, p0k = res[cw * 4 + 1] != 0;
        if (!pk) {
            if (A.flags & RIA_DECODE_PHASE0) f = 0.9375f;
            if (!p0k && (A.flags & RIA_DECODE_PERTURB)) f = 0.875f;
        }
    }
    // Stochastic retry phases 1-6 (frame_v2.cpp:1415-1546)
    if (!ok && (A.flags & RIA_DECODE_PERTURB)) {
        uint32_t h = 0;
        {   // data_hash over the first 16 decoder-order LLR bit patterns (frame_v2.cpp:1391-1396)
            for (int j = 0; j < 16; ++j) {
                uint32_t u = f2u(__shfl(mine[0], j));
                h ^= u + 0x9e3779b9u + (h << 6) + (h >> 2);
            }
        }
        uint32_t* mt = reinterpret_cast<uint32_t*>(msg);          // RNG state: scratch in the message area
        float* normal = reinterpret_cast<float*>(msg) + 640;      // 648 normals after it
        for (int a = 0; a < 34 && !ok; ++a) {
            uint32_t seed; float sigma, factor; int kind;
            retry_transform_params(a, h, &seed, &sigma, &factor, &kind);
            normal648_wave(mt, normal, seed, lane);
#pragma unroll
            for (int r = 0; r < 11; ++r) {
                int i = r * 64 + lane;
                if (i < 648) {
                    float v = mine[r];
                    if (kind == 1) { v = (v < 10.0f) ? v : 10.0f; v = (-10.0f < v) ? v : -10.0f; }
                    else if (kind == 2) v = v * 0.5f;
                    else if (kind == 3) { v = (v < 6.0f) ? v : 6.0f; v = (-6.0f < v) ? v : -6.0f; }
                    else if (kind == 4) v = (v >= 0.0f) ? 1.0f : -1.0f;
                    else if (kind == 5) v = v * 0.25f;
                    float nz = normal[i] * sigma + 0.0f;
                    cur[i] = v + nz;
                }
            }
            wave_sync();
            bool k2; attempts++;
            int it = ldpc_decode_wave(c, msg, cur, hard, factor, c.max_iter, lane, &k2);
            if (k2) { ok = true; iters = it; }
        }
    }
    if (ok) ldpc_pack_info(c, hard, cwd + wave * 68, bpc, lane);
    else for (int b = lane; b < 68; b += 64) cwd[wave * 68 + b] = 0;
    if (lane == 0) { res[wave * 4 + 2] = ok; res[wave * 4 + 3] = iters; res[16 + wave] = attempts; }
    __syncthreads();
    // ---- outputs
    uint8_t* out = A.info_out + static_cast<size_t>(frame) * 4 * bpc;
    for (int b = threadIdx.x; b < 4 * bpc; b += 256) {
        int cw = b / bpc;
        out[b] = res[cw * 4 + 2] ? cwd[cw * 68 + (b - cw * bpc)] : 0;
    }
    if (wave == 0) {
        bool all_ok = res[2] && res[6] && res[10] && res[14];
        int valid = 0, quirk = 0;
        if (all_ok) {
            // parseHeader + DataFrame::deserialize on the straight concatenation (frame_v2.cpp:1195-1252, :556-600).
            // A CW1+ starting with 0xD5 takes the reference's marker-stripping path in
            // reassembleCodewords (:959-989): leave those frames to the host restatement.
            uint8_t* flat = shared + 352;
            for (int b = lane; b < 4 * bpc; b += 64) flat[b] = cwd[(b / bpc) * 68 + (b % bpc)];
            wave_sync();
            bool magic = flat[0] == 0x55 && flat[1] == 0x4C;
            int t = flat[2];
            bool ctl = (t == 0x10 || t == 0x11 || t == 0x16 || t == 0x17 || t == 0x20 || t == 0x21 || t == 0x15 || t == 0x40);
            int plen = (flat[13] << 8) | flat[14];
            int expected = ctl ? 20 : 17 + plen + 2;
            for (int cw = 1; cw < 4; ++cw) if (cw * bpc < expected && flat[cw * bpc] == 0xD5) quirk = 1;
            if (magic && !quirk) {
                if (ctl) {
                    uint32_t crc = crc16_wave(flat, 18, A.crc_bit, A.crc_init, lane);
                    valid = crc == static_cast<uint32_t>((flat[18] << 8) | flat[19]);
                } else {
                    uint32_t hc = crc16_wave(flat, 15, A.crc_bit, A.crc_init, lane);
                    if (hc == static_cast<uint32_t>((flat[15] << 8) | flat[16]) && expected <= 4 * bpc) {
                        uint32_t fc = crc16_wave(flat, expected - 2, A.crc_bit, A.crc_init, lane);
                        valid = fc == static_cast<uint32_t>((flat[expected - 2] << 8) | flat[expected - 1]);
                    }
                }
            }
        }
        if (lane == 0) {
            ria_decode_status s;
            for (int cw = 0; cw < 4; ++cw) {
                s.cw_ok[cw] = res[cw * 4 + 2] ? 1 : 0;
                s.iterations[cw] = static_cast<uint16_t>(res[cw * 4 + 3]);
                s.attempts[cw] = static_cast<uint8_t>(res[16 + cw]);
            }
            s.frame_valid = static_cast<uint8_t>(valid);
            s.needs_recovery = static_cast<uint8_t>(all_ok && !valid);
            s.reserved[0] = static_cast<uint8_t>(quirk);
            s.reserved[1] = 0;
            A.status[frame] = s;
        }
    }
}

}  // namespace ria
