// ria_amd/csrc/ldpc_kernels.hip.h — wave-level helpers shared by the LDPC decode kernels (ldpc_fast.hip.h):
// the wave fence, std::mt19937 + std::normal_distribution<float> on one wavefront (the retry cascade
// perturbs LLRs with noise drawn exactly as frame_v2.cpp:1431-1436 draws it), the wave-parallel CRC-16
// and the cascade's attempt schedule (frame_v2.cpp:1415-1546).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ria_gpu.h"
#include "devmath.h"

namespace ria {

__device__ __forceinline__ void wave_sync() {
    // One wave's DS operations execute in issue order, so cross-lane visibility inside the wave
    // needs no hardware wait; the fence keeps the compiler from moving LDS accesses across it.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// std::mt19937 + libstdc++ std::normal_distribution<float> on one wavefront.
// The retry cascade perturbs LLRs with noise drawn exactly as the reference draws it
// (frame_v2.cpp:1431-1436): a fresh mt19937(seed) and normal_distribution per attempt, 648 values.
// state: 624 words in LDS.  Seeding is a serial recurrence (lane 0); the twist runs in three
// data-independent chunks across the lanes; the Marsaglia polar rejection consumes draws in aligned
// pairs, so acceptance is decided per pair in parallel and ranked with a ballot prefix count.
// W: word stride of the state / output arrays (2 = every other dword: one component of the dual decoder's
// interleaved LDS image, ldpc_dual.hip.h)
template <int W = 1>
__device__ inline void mt_seed_wave(uint32_t* st, uint32_t seed, int lane) {
    // The seeding recurrence is serial, so it runs on the SCALAR unit: the value is wave-uniform (readfirstlane
    // tells the compiler), the 4 ops per step are s_lshr/s_xor/s_mul/s_add and issue beside other waves'
    // vector work; v_writelane parks word i in lane i%64 of a register, ten stores publish the state.
    uint32_t x = __builtin_amdgcn_readfirstlane(seed);
    uint32_t keep = x;   // word 0
    (void)lane;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t reg = 0;
        const int j0 = (r == 0) ? 1 : 0;
        if (r == 0) reg = keep;   // all lanes hold word 0; lane 0 keeps it
        const int jn = (r == 9) ? 624 - 576 : 64;
        for (int j = j0; j < jn; ++j) {
            x = 1812433253u * (x ^ (x >> 30)) + static_cast<uint32_t>(64 * r + j);
            // lane select through M0: a second SGPR operand would exceed the constant-bus limit of one
            asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(reg) : "s"(x), "s"(j) : "m0");
        }
        if (r < 9 || lane < 624 - 576) st[(64 * r + lane) * W] = reg;
    }
    wave_sync();
}
template <int W = 1>
__device__ inline void mt_twist_wave(uint32_t* st, int lane) {
    auto step = [&](int i, int a, int b) {
        uint32_t y = (st[i * W] & 0x80000000u) | (st[a * W] & 0x7fffffffu);
        uint32_t v = st[b * W] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        return v;
    };
    for (int base = 0; base < 227; base += 64) {
        int i = base + lane;
        uint32_t v = 0;
        if (i < 227) v = step(i, i + 1, i + 397);
        wave_sync();
        if (i < 227) st[i * W] = v;
        wave_sync();
    }
    for (int base = 227; base < 623; base += 64) {
        int i = base + lane;
        uint32_t v = 0;
        if (i < 623) v = step(i, i + 1, i - 227);
        wave_sync();
        if (i < 623) st[i * W] = v;
        wave_sync();
    }
    if (lane == 0) st[623 * W] = step(623, 0, 396);
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
template <int W = 1>
__device__ inline void normal648_wave(uint32_t* st, float* normal, uint32_t seed, int lane) {
    mt_seed_wave<W>(st, seed, lane);
    int accepted = 0;  // accepted pairs so far (wave-uniform)
    while (accepted < 324) {
        mt_twist_wave<W>(st, lane);
        for (int base = 0; base < 312 && accepted < 324; base += 64) {
            int p = base + lane;
            bool acc = false;
            float x = 0.f, y = 0.f, r2 = 1.f;
            if (p < 312) {
                x = 2.0f * mt_canonical(st[(2 * p) * W]) - 1.0f;
                y = 2.0f * mt_canonical(st[(2 * p + 1) * W]) - 1.0f;
                r2 = x * x + y * y;
                acc = !(r2 > 1.0f || r2 == 0.0f);
            }
            unsigned long long mask = __ballot(acc);
            int q = accepted + __popcll(mask & ((1ull << lane) - 1ull));
            if (acc && q < 324) {
                float mult = fsqrt(fdiv(-2.0f * logf_glibc(r2), r2));
                normal[(2 * q) * W] = y * mult;      // returned first
                normal[(2 * q + 1) * W] = x * mult;  // the saved value, returned by the next call
            }
            accepted += __popcll(mask);
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// Kernel: v2::decodeFixedFrame for a batch of frames.  One workgroup (4 waves) per frame.
__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o);
    return v;
}
// CRC-16/CCITT-FALSE (ControlFrame::calculateCRC, frame_v2.cpp:115-128) of d[0..L) by one wave.
__device__ inline uint32_t crc16_wave(const uint8_t* d, int L, const uint16_t* crc_bit, const uint16_t* crc_init,
                                      int lane) {
    // the eight per-bit syndromes of a byte are one aligned 16-byte load (the table is indexed by distance from the
    // end of the message, bit-major inside a byte); selected by the byte's bits without a branch
    uint32_t acc = 0;
    for (int i = lane; i < L; i += 64) {
        const uint32_t b = d[i];
        const uint4 w = *reinterpret_cast<const uint4*>(crc_bit + (L - 1 - i) * 8);
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const uint32_t e = (t & 1) ? (ws[t >> 1] >> 16) : (ws[t >> 1] & 0xffffu);
            acc ^= (b & (1u << t)) ? e : 0u;
        }
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

}  // namespace ria
