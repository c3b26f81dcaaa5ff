// ria_amd/csrc/ldpc_dual.hip.h — the min-sum decoder of ldpc_fast.hip.h run on TWO codewords per wavefront.
//
// Same arithmetic, per codeword, as fast_decode (= LDPCDecoder::decodeBP, ldpc_decoder.cpp:154-260), bit for bit.
// What changes is how the work is fed to a CDNA4 compute unit:
//   * the LDS image interleaves the two codewords dword by dword (word w of stream c at byte 8*w + 4*c).  Every code
//     has ONE parity-check matrix, so both streams gather from the same word: one ds_read_b64 per gather serves two
//     codewords, and on gfx950 a 64-bit LDS read costs the LDS array the same two cycles as a 32-bit one.  Gather
//     cycles per codeword halve; the lane-linear stores become ds_write_b64 (6 store-path cycles for two codewords
//     instead of 2 x 4).  The bank pattern is the single decoder's (8-byte words on 64 banks = 4-byte words on 32),
//     so the annealed lane/slot layouts (core_layouts.inc) apply unchanged.
//   * the two streams sit in the halves of 64-bit register pairs: tot - c2v, (sign * min) * factor and the column sums
//     are v_pk_add_f32 / v_pk_mul_f32 (IEEE per half, so the results are the single decoder's); the selections
//     (min3 / med3 / xor / bfi) stay per half.
//   * two independent dependency chains per wave hide each other's LDS latency; the kernels run 2 waves per SIMD with a
//     256-VGPR budget, which also keeps every own-c2v word in registers (no LDS re-reads of the row's own messages).
//   * the streams are independent work-queue consumers: when one codeword finishes (converged, or out of iterations)
//     its result is published and the stream takes the next unit while the other one carries on mid-decode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ldpc_fast.hip.h"

namespace ria {

using lds_v2f_ptr = __attribute__((address_space(3))) v2f*;
__device__ __forceinline__ v2f lds_f2(uint32_t a) { return *(lds_v2f_ptr)(uintptr_t)a; }
__device__ __forceinline__ void lds_sf2(uint32_t a, v2f v) { *(lds_v2f_ptr)(uintptr_t)a = v; }

#ifndef RIA_DUAL_PREFETCH
#define RIA_DUAL_PREFETCH 8
#endif
constexpr int kDualPrefetch = RIA_DUAL_PREFETCH;   // rounds of total gathers in flight ahead of the round being computed (>= NR: all up front)

template <class S>
struct DualInfo {
    using I = ShapeInfo<S>;
    // the wave's LDS image: I::words 8-byte words.  One stream's dwords (stride 2) double as the mt19937 state + 648
    // normals of that stream's next perturbation while the other stream is parked mid-decode (>= 1288 dwords)
    static_assert(I::words >= 640 + 648, "a stream's words must hold the perturbation scratch");
    static constexpr int lds_bytes = (I::words * 8 + 15) & ~15;
};

template <class S>
struct DualState {
    using I = ShapeInfo<S>;
    uint32_t rv[I::TS];                    // gather addresses of the check pass (8-byte words)
    uint32_t cs[I::TD > 0 ? I::TD : 1];    // gather addresses of the column pass
    v2f cv[I::TS];                         // c2v of the row's own edges, both streams
    v2f li[S::NC], lp[S::NR], pv[S::NR], pt[S::NR];
};

template <class S>
__device__ inline void dual_load_tables(DualState<S>& st, const FastCode& c, const unsigned char* lds, int lane) {
    using I = ShapeInfo<S>;
    const uint32_t base = lds_addr(lds);
#pragma unroll
    for (int i = 0; i < I::TS; ++i) { uint32_t a = base + 2u * c.row_addr[i * 64 + lane]; asm volatile("" : "+v"(a)); st.rv[i] = a; }
#pragma unroll
    for (int i = 0; i < I::TD; ++i) { uint32_t a = base + 2u * c.col_addr[i * 64 + lane]; asm volatile("" : "+v"(a)); st.cs[i] = a; }
}

// stream C starts a codeword: its LLRs are in st.li[.][C] / st.lp[.][C]
template <class S, int C>
__device__ inline void dual_reset_stream(DualState<S>& st, unsigned char* lds, int lane) {
    using I = ShapeInfo<S>;
    const uint32_t a0 = lds_addr(lds) + static_cast<uint32_t>(lane) * 8u + 4u * C;
#pragma unroll
    for (int i = 0; i < I::TS; ++i) { st.cv[i][C] = 0.0f; lds_sf(a0 + 512u * i, 0.0f); }
#pragma unroll
    for (int r = 0; r < S::NC; ++r) lds_sf(a0 + 8u * (I::tot_word + 64 * r), st.li[r][C]);
    lds_sf(a0 + 8u * I::zero_word, 0.0f);
    lds_sf(a0 + 8u * I::big_word, kPadTotal);
#pragma unroll
    for (int r = 0; r < S::NR; ++r) { st.pv[r][C] = st.lp[r][C]; st.pt[r][C] = 0.0f; }
}

// check pass of both streams (see fast_decode for the message flow); syn[c] = per-lane syndrome word of stream c
template <class S>
__device__ __forceinline__ void dual_check_pass(DualState<S>& st, uint32_t lane8, v2f hi, v2f factor, uint32_t& syn0, uint32_t& syn1) {
    using I = ShapeInfo<S>;
    const uint32_t kAbs = 0x7fffffffu;
    syn0 = 0; syn1 = 0;
    // The gathers of round r + 2 are issued before round r is computed: LDS operations keep their program order (the
    // compiler cannot tell the c2v stores from the total gathers), so without this every round would start by waiting
    // out a full LDS round trip with nothing to compute.
    v2f tt[I::TS];
    auto gather_round = [&](auto R_) __attribute__((always_inline)) {
        constexpr int r = decltype(R_)::value;
        if constexpr (r < S::NR) {
#pragma unroll
            for (int s = 0; s < S::ne(r); ++s) tt[I::row_off(r) + s] = lds_f2(st.rv[I::row_off(r) + s]);
        }
    };
    static_for<0, kDualPrefetch>([&](auto R_) __attribute__((always_inline)) { gather_round(R_); });
    static_for<0, S::NR>([&](auto R_) __attribute__((always_inline)) {
        constexpr int r = decltype(R_)::value;
        constexpr int NE = S::ne(r);
        constexpr int off = I::row_off(r);
        gather_round(std::integral_constant<int, r + kDualPrefetch>{});
        v2f t[NE], v[NE];
#pragma unroll
        for (int s = 0; s < NE; ++s) t[s] = tt[off + s];
#pragma unroll
        for (int s = 0; s < NE; ++s) v[s] = t[s] - st.cv[off + s];          // v_pk_add_f32 (neg): both streams
        const v2f pvr = st.pv[r];
        float vx[NE + 1], vy[NE + 1];
#pragma unroll
        for (int s = 0; s < NE; ++s) { vx[s] = v[s].x; vy[s] = v[s].y; }
        vx[NE] = pvr.x; vy[NE] = pvr.y;
        float m1x, m2x, m1y, m2y;
        two_smallest_abs<NE + 1>(vx, m1x, m2x);
        two_smallest_abs<NE + 1>(vy, m1y, m2y);
        m1x = min_raw(m1x, hi.x); m2x = min_raw(m2x, hi.x);
        m1y = min_raw(m1y, hi.y); m2y = min_raw(m2y, hi.y);
        uint32_t px = f2u(st.pt[r].x), py = f2u(st.pt[r].y), sx = f2u(pvr.x), sy = f2u(pvr.y);
#pragma unroll
        for (int s = 0; s < NE; s += 2) {
            if (s + 1 < NE) {
                px = __builtin_amdgcn_bitop3_b32(px, f2u(t[s].x), f2u(t[s + 1].x), 0x96);
                py = __builtin_amdgcn_bitop3_b32(py, f2u(t[s].y), f2u(t[s + 1].y), 0x96);
                sx = __builtin_amdgcn_bitop3_b32(sx, f2u(vx[s]), f2u(vx[s + 1]), 0x96);
                sy = __builtin_amdgcn_bitop3_b32(sy, f2u(vy[s]), f2u(vy[s + 1]), 0x96);
            } else {
                px ^= f2u(t[s].x); py ^= f2u(t[s].y);
                sx ^= f2u(vx[s]); sy ^= f2u(vy[s]);
            }
        }
        syn0 |= px; syn1 |= py;
        const uint32_t dx = bfi(kAbs, f2u(m1x) ^ f2u(m2x), sx), dy = bfi(kAbs, f2u(m1y) ^ f2u(m2y), sy);
        static_for<0, NE>([&](auto S_) __attribute__((always_inline)) {
            constexpr int s = decltype(S_)::value;
            v2f o;
            o.x = u2f(dx ^ f2u(__builtin_amdgcn_fmed3f(vx[s], m2x, -m2x)));
            o.y = u2f(dy ^ f2u(__builtin_amdgcn_fmed3f(vy[s], m2y, -m2y)));
            o = o * factor;                                                  // v_pk_mul_f32
            lds_sf2(lane8 + 512u * (off + s), o);
            st.cv[off + s] = o;
        });
        {   // identity column (degree 1): total = llr + c2v, v2c = total - c2v (clamped where it is used)
            v2f c2v;
            c2v.x = u2f(dx ^ f2u(__builtin_amdgcn_fmed3f(pvr.x, m2x, -m2x)));
            c2v.y = u2f(dy ^ f2u(__builtin_amdgcn_fmed3f(pvr.y, m2y, -m2y)));
            c2v = c2v * factor;
            const v2f tot = st.lp[r] + c2v;
            st.pv[r] = tot - c2v;
            st.pt[r] = tot;
        }
    });
}

template <class S>
__device__ __forceinline__ void dual_column_pass(DualState<S>& st, uint32_t lane8) {
    using I = ShapeInfo<S>;
    static_for<0, S::NC>([&](auto R_) __attribute__((always_inline)) {
        constexpr int r = decltype(R_)::value;
        constexpr int DV = S::dv(r);
        constexpr int off = I::col_off(r);
        if constexpr (DV > 0) {
            v2f cv[DV];
#pragma unroll
            for (int d = 0; d < DV; ++d) cv[d] = lds_f2(st.cs[off + d]);
            v2f tot = st.li[r];
#pragma unroll
            for (int d = 0; d < DV; ++d) tot = tot + cv[d];                  // ascending check order, v_pk_add_f32
            lds_sf2(lane8 + 8u * (I::tot_word + 64 * r), tot);
        }
    });
}

// information hard bits of stream C -> bytes (MSB first); scratch = the image's dump words (never read by the decoder)
template <class S, int C>
__device__ inline void dual_pack(const FastCode& c, unsigned char* lds, uint8_t* out, int nbytes, int lane) {
    using I = ShapeInfo<S>;
    unsigned char* scratch = lds + 8 * I::dump_word;
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(scratch);
    const int nr = (c.k + 63) / 64;
    lane = opaque_lane(lane);
    wave_sync();
    for (int r = 0; r < nr; ++r) {
        const int j = lane + 64 * r;
        bool bit = false;
        if (j < c.k) bit = (f2u(lds_f(lds_addr(lds) + 8u * (I::tot_word + c.col_pos[j]) + 4u * C)) >> 31) != 0u;
        unsigned long long mk = __ballot(bit);
        if (lane == 0) masks[r] = mk;
    }
    wave_sync();
    for (int b = lane; b < nbytes; b += 64) out[b] = static_cast<uint8_t>(__brev(static_cast<unsigned>(scratch[b])) >> 24);
    wave_sync();
}

// The two-stream decode loop.  Ops supplies the work:
//   template <int C> bool fetch(DualState&, float* factor, int* max_iter)   next unit of stream C into st.li/lp[.][C]; false = queue empty
//   template <int C> void finish(bool ok, int iterations)                   publish stream C's result (may call dual_pack<S, C>)
// Per stream, with k = column passes made so far, a check pass evaluates the syndrome of those k iterations
// (fast_decode): k > 0 and syndrome 0 -> converged, lastIterations = k - 1; k == max_iter -> failed, max_iter.
template <class S, class Ops>
__device__ inline void dual_decode_loop(DualState<S>& st, const FastCode& c, unsigned char* lds, int lane, Ops& ops) {
    const uint32_t lane8 = lds_addr(lds) + static_cast<uint32_t>(lane) * 8u;
    v2f factor = {0.0f, 0.0f}, hi = {__builtin_inff(), __builtin_inff()};
    int k0 = 0, k1 = 0, max0 = 0, max1 = 0;
    float f = 0.0f;
    bool act0 = ops.template fetch<0>(st, &f, &max0);
    if (act0) { factor.x = f; dual_reset_stream<S, 0>(st, lds, lane); }
    bool act1 = ops.template fetch<1>(st, &f, &max1);
    if (act1) { factor.y = f; dual_reset_stream<S, 1>(st, lds, lane); }
    wave_sync();
    while (act0 || act1) {
        uint32_t syn0, syn1;
        dual_check_pass<S>(st, lane8, hi, factor, syn0, syn1);
        hi.x = 50.0f; hi.y = 50.0f;
        const bool z0 = __ballot(static_cast<int>(syn0) < 0) == 0ull, z1 = __ballot(static_cast<int>(syn1) < 0) == 0ull;
        bool fresh0 = false, fresh1 = false;
        if (act0 && ((k0 > 0 && z0) || k0 == max0)) {
            const bool ok = k0 > 0 && z0;
            ops.template finish<0>(ok, ok ? k0 - 1 : max0);
            act0 = ops.template fetch<0>(st, &f, &max0);
            if (act0) { factor.x = f; dual_reset_stream<S, 0>(st, lds, lane); hi.x = __builtin_inff(); }
            k0 = 0; fresh0 = true;
        }
        if (act1 && ((k1 > 0 && z1) || k1 == max1)) {
            const bool ok = k1 > 0 && z1;
            ops.template finish<1>(ok, ok ? k1 - 1 : max1);
            act1 = ops.template fetch<1>(st, &f, &max1);
            if (act1) { factor.y = f; dual_reset_stream<S, 1>(st, lds, lane); hi.y = __builtin_inff(); }
            k1 = 0; fresh1 = true;
        }
        if (!(act0 || act1)) break;
        wave_sync();
        // a stream that has just been reset holds c2v = 0 everywhere: its column pass rewrites tot = llr (x + 0.0 = x)
        dual_column_pass<S>(st, lane8);
        if (!fresh0) ++k0;
        if (!fresh1) ++k1;
        wave_sync();
    }
}

template <class S, int C>
__device__ inline void dual_load_staged(DualState<S>& st, const float* __restrict__ src, int lane) {
    lane = opaque_lane(lane);
#pragma unroll
    for (int r = 0; r < S::NC; ++r) st.li[r][C] = src[r * 64 + lane];
#pragma unroll
    for (int r = 0; r < S::NR; ++r) st.lp[r][C] = src[(S::NC + r) * 64 + lane];
}

// ------------------------------------------------------------------------------------------------ phase 0, dual
template <class S>
struct Phase0Ops {
    const FastDecodeArgs& A;
    unsigned char* lds;
    int lane;
    unsigned total;
    unsigned fc[2];
    int fi[2];
    unsigned guard = 0;
    template <int C>
    __device__ bool fetch(DualState<S>& st, float* factor, int* max_iter) {
        if (++guard > total + 2u) { if (lane == 0) atomicExch(&A.ctl->queue_fault, 1u); return false; }   // RIA_QUEUE_GUARD
        unsigned u = atomicAdd(&A.ctl->next_z, lane == 0 ? 1u : 0u);   // all-lane atomic form
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= total) return false;
        fc[C] = A.list1[u >> 2];
        fi[C] = 1 + static_cast<int>(u & 3u);
        dual_load_staged<S, C>(st, A.staged + static_cast<size_t>(u >> 2) * kStageFloats, lane);
        *factor = kFactors[fi[C]];
        *max_iter = A.c.max_iter;
        return true;
    }
    template <int C>
    __device__ void finish(bool ok, int it) {
        const FastCode& c = A.c;
        if (ok) dual_pack<S, C>(c, lds, A.res_bytes + (static_cast<size_t>(fc[C]) * kNumFactors + fi[C]) * c.bytes_per_cw, c.bytes_per_cw, lane);
        if (lane == 0) { A.res[fc[C]].state[fi[C]] = ok ? 2 : 1; A.res[fc[C]].iters[fi[C]] = static_cast<uint16_t>(it); }
    }
};

template <class S>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void dual_phase0_kernel(FastDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const unsigned total = A.ctl->n_list1 * 4u;
    if (2u * blockIdx.x >= total) return;
    DualState<S> st;
    dual_load_tables(st, A.c, smem, lane);
    Phase0Ops<S> ops{A, smem, lane, total, {0u, 0u}, {0, 0}, 0u};
    dual_decode_loop<S>(st, A.c, smem, lane, ops);
}

// ------------------------------------------------------------------------------------------------ cascade, dual
template <class S>
struct CascadeOps {
    const FastDecodeArgs& A;
    unsigned char* lds;
    int lane;
    unsigned n_entries, total;
    unsigned e[2], a[2];
    unsigned guard = 0;
    template <int C>
    __device__ bool fetch(DualState<S>& st, float* factor, int* max_iter) {
        const FastCode& c = A.c;
        for (;;) {
            if (++guard > total + 2u) { if (lane == 0) atomicExch(&A.ctl->queue_fault, 1u); return false; }   // RIA_QUEUE_GUARD
            unsigned u = atomicAdd(&A.ctl->next_unit, lane == 0 ? 1u : 0u);
            u = __builtin_amdgcn_readfirstlane(u);
            if (u >= total) return false;
            const unsigned aa = u / n_entries, ee = u - aa * n_entries;
            unsigned b = __hip_atomic_load(&A.best[ee], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b = __builtin_amdgcn_readfirstlane(b);
            if (b < aa) continue;                       // an earlier attempt already succeeded
            e[C] = ee; a[C] = aa;
            break;
        }
        const unsigned fcw = A.entries[e[C]];
        const unsigned li = A.l1idx[fcw];
        const float* src = A.staged + static_cast<size_t>(li) * kStageFloats;
        const int lane = opaque_lane(this->lane);   // per-unit indices and addresses are recomputed, not hoisted (ldpc_fast.hip.h)
        float bi[S::NC], bp[S::NR];
#pragma unroll
        for (int r = 0; r < S::NC; ++r) bi[r] = src[r * 64 + lane];
#pragma unroll
        for (int r = 0; r < S::NR; ++r) bp[r] = src[(S::NC + r) * 64 + lane];
        uint32_t seed; float sigma, fac; int kind;
        retry_transform_params(static_cast<int>(a[C]), A.l1hash[li], &seed, &sigma, &fac, &kind);
        // this stream's own dwords of the image are free until dual_reset_stream: mt19937 state + 648 normals, stride 2
        uint32_t* mt = reinterpret_cast<uint32_t*>(lds) + C;
        float* normal = reinterpret_cast<float*>(lds) + C + 2 * 640;
        wave_sync();
        normal648_wave<2>(mt, normal, seed, lane);
        auto tf = [&](float v, float nz) {
            if (kind == 1) { v = (v < 10.0f) ? v : 10.0f; v = (-10.0f < v) ? v : -10.0f; }
            else if (kind == 2) v = v * 0.5f;
            else if (kind == 3) { v = (v < 6.0f) ? v : 6.0f; v = (-6.0f < v) ? v : -6.0f; }
            else if (kind == 4) v = (v >= 0.0f) ? 1.0f : -1.0f;
            else if (kind == 5) v = v * 0.25f;
            return v + (nz * sigma + 0.0f);
        };
#pragma unroll
        for (int r = 0; r < S::NC; ++r) {
            const uint32_t j = c.col_at[lane + 64 * r];
            st.li[r][C] = (j != 0xFFFFu) ? llr_canon(tf(bi[r], normal[2 * j])) : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < S::NR; ++r) {
            const uint32_t i = c.check_at[lane + 64 * r];
            st.lp[r][C] = (i != 0xFFFFu) ? llr_canon(tf(bp[r], normal[2 * (c.k + i)])) : kIdleRowLlr;
        }
        wave_sync();
        *factor = fac;
        *max_iter = c.max_iter;
        return true;
    }
    template <int C>
    __device__ void finish(bool ok, int it) {
        if (!ok) return;
        const FastCode& c = A.c;
        const unsigned ee = e[C], aa = a[C];
        unsigned int prev = atomicMin(&A.best[ee], lane == 0 ? aa : 0xFFFFFFFFu);   // all-lane form: only lane 0's operand can lower it
        prev = __builtin_amdgcn_readfirstlane(prev);
        if (aa < prev) {   // best so far: publish under the entry's lock (held for one 40..68-byte store)
            CascadeWin* w = A.win + ee;
            for (;;) {     // wave-uniform spin: every lane tries, lane 0's result decides
                unsigned got = atomicCAS(&w->lock, 0u, lane == 0 ? 1u : 0u);
                got = __builtin_amdgcn_readfirstlane(got);
                if (got == 0u) break;
                __builtin_amdgcn_s_sleep(1);
            }
            __threadfence();
            unsigned int cur = __hip_atomic_load(&A.best[ee], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cur = __builtin_amdgcn_readfirstlane(cur);
            if (cur == aa) {
                dual_pack<S, C>(c, lds, w->bytes, c.bytes_per_cw, lane);
                if (lane == 0) w->iters = static_cast<unsigned int>(it);
            }
            __threadfence();
            if (lane == 0) atomicExch(&w->lock, 0u);
        }
    }
};

template <class S>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void dual_cascade_kernel(FastDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const unsigned int n_entries = A.ctl->n_entries;
    const unsigned int total = n_entries * 34u;
    if (total == 0) return;
    DualState<S> st;
    dual_load_tables(st, A.c, smem, lane);
    CascadeOps<S> ops{A, smem, lane, n_entries, total, {0u, 0u}, {0u, 0u}, 0u};
    dual_decode_loop<S>(st, A.c, smem, lane, ops);
}

}  // namespace ria
