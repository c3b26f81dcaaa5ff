// ria_amd/csrc/ldpc_fast.hip.h — register-resident flooding min-sum LDPC(648,k) decoder, v2.
//
// Same arithmetic as ldpc_kernels.hip.h (bit-exact LDPCDecoder::decodeBP, ldpc_decoder.cpp:154-260);
// what changes is where things live and how work is scheduled:
//   * one wavefront per codeword, NO index loads in the iteration loop: each lane owns rows
//     i = lane + 64r and information columns j = lane + 64r; their degrees and message-slot addresses
//     sit in VGPRs (packed u16), loaded once per decode;
//   * the identity (parity) column k+i has a single edge, to row i, owned by the SAME lane: its
//     message never touches LDS and its variable update is fused into the check pass;
//   * the syndrome of iteration t is evaluated inside the check pass of iteration t+1 from per-edge
//     hard-bit bytes the variable pass leaves next to the messages (no adjacency walk);
//   * LDS per wave: [6][m] floats + [6][m] bytes (9.7 KB at R1/2), so >= 12 waves per CU;
//   * the retry cascade (frame_v2.cpp:1415-1546) is a second, persistent kernel over a device-side
//     work list of (codeword, attempt) units, so one hopeless codeword no longer pins a whole
//     workgroup for 39 x 80 iterations while its three sibling waves idle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ria_gpu.h"
#include "devmath.h"
#include "ldpc_kernels.hip.h"

namespace ria {

constexpr int kInfoSlots = 6;  // information edges per check (max_check_degree, ldpc_decoder.cpp:87)

// Compile-time shape of a code: RR row rounds (ceil(m/64)), IR information-column rounds (ceil(k/64)),
// DV max information-column degree.
template <int RR_, int IR_, int DV_>
struct CodeShape { static constexpr int RR = RR_, IR = IR_, DV = DV_; };
using ShapeR12 = CodeShape<6, 6, 5>;    // R1/2: m = 324, k = 324, dv <= 5
using ShapeR13 = CodeShape<6, 6, 6>;    // R1/3 (same k, m; H seeded differently): dv <= 6
using ShapeR14 = CodeShape<8, 3, 13>;   // m = 486, k = 162
using ShapeR23 = CodeShape<4, 7, 3>;    // m = 216, k = 432
using ShapeR34 = CodeShape<3, 8, 3>;    // m = 162, k = 486
using ShapeR56 = CodeShape<2, 9, 3>;    // m = 108, k = 540

// Device view of host_tables.hpp FastTables (rows sorted by degree: position p = 64*round + lane).
struct FastCode {
    int k, m, max_iter, bytes_per_cw;
    const uint16_t* perm;      // [m] position -> check index
    const uint8_t* row_ne;     // [m] information edges at position p
    const uint16_t* row_var;   // [6][m] information variable of slot s at position p
    const uint8_t* col_deg;    // [k]
    const uint16_t* col_slot;  // [dv][k] slot word index s*m + p, ascending check order
    uint8_t round_ne[8];       // wave-uniform loop bounds
    uint8_t round_cd[16];
};

// LDS per wave (words): c2v messages [6][m] | total LLR of the information columns [k] | 64 dummy words
// (stores of inactive edges are redirected there, one word per lane, to keep the loop branch-free).
// The same region doubles as mt19937 state + 648 normals during the retry cascade (>= 1296 words).
__host__ __device__ inline int fast_tot_word(int m) { return kInfoSlots * m; }
__host__ __device__ inline int fast_dummy_word(int m, int k) { return kInfoSlots * m + k; }
__host__ __device__ inline int fast_wave_lds_bytes(int m, int k = 324) {
    int w = kInfoSlots * m + k + 64;
    if (w < 1296) w = 1296;
    return (w * 4 + 15) & ~15;
}

template <class S>
struct FastState {
    // tables (loaded once per wave): BYTE addresses inside the wave's LDS region, packed 2 x u16
    uint32_t rne;                                // 4 bits per row round: information edges (0 beyond m)
    uint32_t cdeg[(S::IR + 7) / 8];              // 4 bits per column round
    uint32_t rvar[(S::RR * kInfoSlots + 1) / 2]; // byte address of tot[var] for (round, slot)
    uint32_t cslot[(S::IR * S::DV + 1) / 2];     // byte address of the c2v slot for (round, d)
    // decoder input of the current attempt
    float li[S::IR];               // information-column LLRs (column j = lane + 64 r)
    float lp[S::RR];               // identity-column LLRs   (column k + perm[lane + 64 r])
    // running state
    float pv[S::RR];               // v2c of the identity edges
    uint32_t phard;                // hard bits of the identity columns (bit r)
    uint32_t ihard;                // hard bits of the information columns (bit r)
};

template <int PER, class T>
__device__ __forceinline__ uint32_t unpack16(const T& arr, int idx) {
    uint32_t w = arr[idx >> 1];
    return (idx & 1) ? (w >> 16) : (w & 0xffffu);
}

template <class S>
__device__ inline void fast_load_tables(FastState<S>& st, const FastCode& c, int lane) {
    st.rne = 0;
#pragma unroll
    for (int w = 0; w < (S::RR * kInfoSlots + 1) / 2; ++w) st.rvar[w] = 0;
#pragma unroll
    for (int w = 0; w < (S::IR + 7) / 8; ++w) st.cdeg[w] = 0;
#pragma unroll
    for (int w = 0; w < (S::IR * S::DV + 1) / 2; ++w) st.cslot[w] = 0;
    const uint32_t tot_base = static_cast<uint32_t>(fast_tot_word(c.m)) * 4u;
#pragma unroll
    for (int r = 0; r < S::RR; ++r) {
        int p = lane + 64 * r;
        uint32_t ne = (p < c.m) ? c.row_ne[p] : 0u;
        st.rne |= ne << (4 * r);
#pragma unroll
        for (int s = 0; s < kInfoSlots; ++s) {
            uint32_t a = (s < static_cast<int>(ne)) ? tot_base + 4u * c.row_var[s * c.m + p] : tot_base;
            st.rvar[(r * kInfoSlots + s) >> 1] |= a << (((r * kInfoSlots + s) & 1) * 16);
        }
    }
#pragma unroll
    for (int r = 0; r < S::IR; ++r) {
        int j = lane + 64 * r;
        uint32_t d = (j < c.k) ? c.col_deg[j] : 0u;
        st.cdeg[r >> 3] |= d << (4 * (r & 7));
#pragma unroll
        for (int e = 0; e < S::DV; ++e) {
            uint32_t a = (e < static_cast<int>(d)) ? 4u * c.col_slot[e * c.k + j] : 0u;
            st.cslot[(r * S::DV + e) >> 1] |= a << (((r * S::DV + e) & 1) * 16);
        }
    }
}

__device__ __forceinline__ float lds_f(const unsigned char* base, uint32_t off) { return *reinterpret_cast<const float*>(base + off); }
__device__ __forceinline__ void lds_sf(unsigned char* base, uint32_t off, float v) { *reinterpret_cast<float*>(base + off) = v; }

// Runs the decoder on the LLRs in st.li / st.lp.  Returns LDPCDecoder::lastIterations(); *ok = converged.
// On return st.ihard holds the information hard bits of the accepted (or last) iteration.
//
// Message flow per iteration (same arithmetic as ldpc_decoder.cpp:176-236, different bookkeeping):
//   check pass   for every edge: v2c = clamp(tot[var] - c2v_old) (iteration 0: v2c = llr), hard bit =
//                tot[var] < 0 (-> syndrome of the PREVIOUS iteration for free), then the min-sum update
//                written back to the edge's slot as c2v;
//   column pass  tot[j] = llr[j] + sum of its c2v slots in ascending check order -> ONE store per column.
// Integer tricks keep it exact: |x| of finite floats orders like the unsigned bit pattern, so min1/min2
// use integer min/med3; "x < 0" is taken from a float compare so that -0.0 counts as positive exactly
// like the reference's `if (msg < 0)`.
template <class S>
__device__ inline int fast_decode(FastState<S>& st, const FastCode& c, unsigned char* __restrict__ lds,
                                  float factor, int max_iter, int lane, bool* ok) {
    const int m = c.m;
    const uint32_t tot_base = static_cast<uint32_t>(fast_tot_word(m)) * 4u;
    const uint32_t dummy = static_cast<uint32_t>(fast_dummy_word(m, c.k) + lane) * 4u;
    const uint32_t kInfBits = 0x7f7fffffu;  // FLT_MAX, the reference's initial min_abs
    // tot := channel LLR of the information columns
#pragma unroll
    for (int r = 0; r < S::IR; ++r) {
        int j = lane + 64 * r;
        lds_sf(lds, (j < c.k) ? tot_base + 4u * j : dummy, st.li[r]);
    }
#pragma unroll
    for (int r = 0; r < S::RR; ++r) st.pv[r] = st.lp[r];
    st.phard = 0; st.ihard = 0;
    wave_sync();
    int it = 0;
    bool success = false;
    for (;; ++it) {
        const bool last_check_only = (it == max_iter);  // trailing pass: syndrome of the final iteration
        const bool first = (it == 0);
        // Keep the per-lane tables opaque inside the loop: otherwise LICM hoists all 36 unpacked
        // addresses and 36 lane masks (64-bit SGPR pairs) out of it and the kernel drowns in spills.
        uint32_t rne = st.rne;
        asm volatile("" : "+v"(rne));
        uint32_t syn = 0;
#pragma unroll
        for (int r = 0; r < S::RR; ++r) {
            const int p = lane + 64 * r;
            // 6-bit activity pattern of this row's slots (ones for s < ne), rows beyond m: none
            uint32_t abits = (1u << ((rne >> (4 * r)) & 15u)) - 1u;
            const uint32_t vmask = static_cast<uint32_t>((p - m) >> 31); // all ones for p < m
            const uint32_t srow = 4u * static_cast<uint32_t>(p) & vmask;
            // (opaque per round: the unpacked addresses must not be hoisted out of their round)
#pragma unroll
            for (int w = (r * kInfoSlots) / 2; w <= (r * kInfoSlots + kInfoSlots - 1) / 2; ++w) asm volatile("" : "+v"(st.rvar[w]));
            // issue all 12 LDS reads of the round back to back (inactive slots read valid dummy addresses)
            float t[kInfoSlots], cold[kInfoSlots];
#pragma unroll
            for (int s = 0; s < kInfoSlots; ++s) {
                t[s] = lds_f(lds, unpack16<0>(st.rvar, r * kInfoSlots + s));
                cold[s] = lds_f(lds, srow + 4u * s * m);
            }
            uint32_t ab[kInfoSlots], sb[kInfoSlots], act[kInfoSlots];
            uint32_t min1 = kInfBits, min2 = kInfBits, sgn = 0, par = (st.phard >> r) & 1u;
#pragma unroll
            for (int s = 0; s < kInfoSlots; ++s) {
                // sign-extended bit s of abits: all ones when the slot is active (kept opaque so that the
                // compiler does not turn the masks back into 64-bit SGPR lane masks)
                int am;
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(am) : "v"(abits), "n"(s));
                act[s] = static_cast<uint32_t>(am);
                float v = t[s] - cold[s];
                v = (v < 50.0f) ? v : 50.0f;
                v = (-50.0f < v) ? v : -50.0f;
                v = first ? t[s] : v;
                par ^= ((t[s] < 0.0f) ? 1u : 0u) & act[s];
                sb[s] = ((v < 0.0f) ? 0x80000000u : 0u) & act[s];
                sgn ^= sb[s];
                const uint32_t a = f2u(v) & 0x7fffffffu;
                ab[s] = (a & act[s]) | (kInfBits & ~act[s]);
                const uint32_t lo = min(ab[s], min1);
                const uint32_t hi = max(ab[s], min1);
                min2 = min(hi, min2);
                min1 = lo;
            }
            syn |= par & vmask;
            if (!last_check_only) {   // wave-uniform
                const float xp = st.pv[r];                // identity edge: the last edge of the row
                const uint32_t sp = (xp < 0.0f) ? 0x80000000u : 0u;
                sgn ^= sp;
                const uint32_t ap = f2u(xp) & 0x7fffffffu;
                {
                    const uint32_t lo = min(ap, min1);
                    const uint32_t hi = max(ap, min1);
                    min2 = min(hi, min2);
                    min1 = lo;
                }
#pragma unroll
                for (int s = 0; s < kInfoSlots; ++s) {
                    // min over the OTHER edges: min2 if this edge holds the minimum (ties: min2 == min1)
                    const uint32_t mn = (ab[s] == min1) ? min2 : min1;
                    const float cv = u2f(mn | (sgn ^ sb[s])) * factor;   // (sign * min_abs) * factor
                    const uint32_t am = act[s] & vmask;
                    const uint32_t addr = ((srow + 4u * s * m) & am) | (dummy & ~am);
                    lds_sf(lds, addr, cv);
                }
                {   // identity column: degree 1, total = llr + c2v, v2c = clamp(total - c2v)
                    const uint32_t mn = (ap == min1) ? min2 : min1;
                    const float c2v = u2f(mn | (sgn ^ sp)) * factor;
                    const float tot = st.lp[r] + c2v;
                    float y = tot - c2v;
                    y = (y < 50.0f) ? y : 50.0f;
                    y = (-50.0f < y) ? y : -50.0f;
                    st.pv[r] = u2f(f2u(y) & vmask);
                    st.phard = (st.phard & ~(1u << r)) | (((tot < 0.0f) ? (1u << r) : 0u) & vmask);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // one round at a time: keeps the live set (VGPRs) small
        }
        if (it > 0 && __ballot(syn != 0) == 0ull) { success = true; --it; break; }
        if (last_check_only) break;
        wave_sync();
        // information columns: tot = llr + sum of c2v in ascending check order
        uint32_t cd0 = st.cdeg[0];
        asm volatile("" : "+v"(cd0));
#pragma unroll
        for (int r = 0; r < S::IR; ++r) {
            const int j = lane + 64 * r;
            const int deg = static_cast<int>((((r < 8) ? cd0 : st.cdeg[r >> 3]) >> (4 * (r & 7))) & 15u);
            float tot = st.li[r];
#pragma unroll
            for (int w = (r * S::DV) / 2; w <= (r * S::DV + S::DV - 1) / 2; ++w) asm volatile("" : "+v"(st.cslot[w]));
            float cv[S::DV];
#pragma unroll
            for (int d = 0; d < S::DV; ++d) cv[d] = lds_f(lds, unpack16<0>(st.cslot, r * S::DV + d));
#pragma unroll
            for (int d = 0; d < S::DV; ++d) tot = (d < deg) ? tot + cv[d] : tot;
            const uint32_t jm = static_cast<uint32_t>((j - c.k) >> 31);
            lds_sf(lds, ((tot_base + 4u * j) & jm) | (dummy & ~jm), tot);
            st.ihard = (st.ihard & ~(1u << r)) | ((tot < 0.0f) ? (1u << r) : 0u);
            if (r & 1) __builtin_amdgcn_sched_barrier(0);
        }
        wave_sync();
    }
    *ok = success;
    return success ? it : max_iter;
}

// information hard bits -> bytes, MSB first (decodeBP's packing); scratch: >= 8*IR bytes of LDS
template <class S>
__device__ inline void fast_pack(const FastState<S>& st, const FastCode& c, uint8_t* scratch, uint8_t* out, int nbytes,
                                 int lane) {
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(scratch);
#pragma unroll
    for (int r = 0; r < S::IR; ++r) {
        bool bit = ((st.ihard >> r) & 1u) && (lane + 64 * r < c.k);
        unsigned long long mk = __ballot(bit);
        if (lane == 0) masks[r] = mk;
    }
    wave_sync();
    for (int b = lane; b < nbytes; b += 64) out[b] = static_cast<uint8_t>(__brev(static_cast<unsigned>(scratch[b])) >> 24);
    wave_sync();
}

// ------------------------------------------------------------------------------------------------ args
struct DecodeCtl {          // zeroed by hipMemsetAsync before every decode call
    unsigned int n_entries; // codewords that need the cascade
    unsigned int next_unit; // cascade work queue head
    unsigned int n_list1;   // codewords that need the other four min-sum factors
    unsigned int next_z;    // phase-0 work queue head
};

// result of decoding one codeword's UNMODIFIED LLRs with factor kFactors[f]
constexpr int kNumFactors = 5;
__device__ __constant__ float kFactors[kNumFactors] = {0.9375f, 0.875f, 0.75f, 0.625f, 0.5f};
struct CwResult {
    uint8_t state[kNumFactors];      // 0 not computed (table is zeroed per call), 1 failed, 2 converged
    uint8_t pad[kNumFactors - 2];
    uint16_t iters[kNumFactors];
};  // 18 bytes

struct FastDecodeArgs {
    FastCode c;
    const uint16_t* gather;
    const float* llr;
    int llr_stride;
    int n_frames;
    uint32_t flags;
    uint8_t* info_out;
    ria_decode_status* status;
    const uint16_t* crc_bit;
    const uint16_t* crc_init;
    DecodeCtl* ctl;
    unsigned int* entries;   // [4*n_frames]  frame*4 + cw
    unsigned int* best;      // [4*n_frames]  first successful cascade attempt (0..33) or 0xFFFFFFFF
    unsigned int* list1;     // [4*n_frames]  frame*4 + cw needing factors 1..4
    CwResult* res;           // [4*n_frames]
    uint8_t* res_bytes;      // [4*n_frames][5][bytes_per_cw]
};

template <class S>
__device__ inline void fast_gather_llr(FastState<S>& st, const FastCode& c, const float* fl, const uint16_t* gather,
                                       int cw, int lane) {
#pragma unroll
    for (int r = 0; r < S::IR; ++r) { int j = lane + 64 * r; st.li[r] = (j < c.k) ? fl[gather[cw * 648 + j]] : 0.0f; }
#pragma unroll
    for (int r = 0; r < S::RR; ++r) { int p = lane + 64 * r; st.lp[r] = (p < c.m) ? fl[gather[cw * 648 + c.k + c.perm[p]]] : 0.0f; }
}

// decode codeword `fc` (= frame*4 + cw) with factor index f and record the result
template <class S>
__device__ inline void fast_unit(FastState<S>& st, const FastDecodeArgs& A, unsigned char* lds, unsigned fc, int f, int lane) {
    const FastCode& c = A.c;
    fast_gather_llr(st, c, A.llr + static_cast<size_t>(fc >> 2) * A.llr_stride, A.gather, fc & 3, lane);
    bool ok;
    int it = fast_decode(st, c, lds, kFactors[f], c.max_iter, lane, &ok);
    if (ok) fast_pack(st, c, lds, A.res_bytes + (static_cast<size_t>(fc) * kNumFactors + f) * c.bytes_per_cw,
                      c.bytes_per_cw, lane);
    if (lane == 0) { A.res[fc].state[f] = ok ? 2 : 1; A.res[fc].iters[f] = static_cast<uint16_t>(it); }
}

// ------------------------------------------------------------------------------------------------ kernel P
// first decode of every codeword (factor 0.9375): one single-wave workgroup per codeword, so the
// hardware dispatcher balances converging (few iterations) and hopeless (80 iterations) codewords.
template <class S>
__global__ __launch_bounds__(64) void fast_primary_kernel(FastDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    FastState<S> st;
    fast_load_tables(st, A.c, lane);
    fast_unit(st, A, smem, blockIdx.x, 0, lane);
}

// per frame: every codeword at or after the first one whose first decode failed may need the other
// four factors (phase 0 itself, or a first decode that inherits factor 0.875 through the chain)
__global__ void fast_mark_kernel(FastDecodeArgs A) {
    int frame = blockIdx.x * blockDim.x + threadIdx.x;
    if (frame >= A.n_frames) return;
    if (!(A.flags & (RIA_DECODE_PHASE0 | RIA_DECODE_PERTURB))) return;
    bool failed = false;
    for (int cw = 0; cw < 4; ++cw) {
        unsigned fc = static_cast<unsigned>(frame) * 4u + cw;
        failed = failed || A.res[fc].state[0] != 2;
        if (failed) A.list1[atomicAdd(&A.ctl->n_list1, 1u)] = fc;
    }
}

// ------------------------------------------------------------------------------------------------ kernel Z
// one single-wave workgroup per (list1 entry, factor 1..4); the grid is sized for the worst case and
// surplus workgroups exit at once (the list length is only known on the device)
template <class S>
__global__ __launch_bounds__(64) void fast_phase0_kernel(FastDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const unsigned u = blockIdx.x;
    if (u >= A.ctl->n_list1 * 4u) return;
    FastState<S> st;
    fast_load_tables(st, A.c, lane);
    fast_unit(st, A, smem, A.list1[u >> 2], 1 + static_cast<int>(u & 3u), lane);
}

// ------------------------------------------------------------------------------------------------ chain
// One thread per frame replays the reference's sequential bookkeeping over the result table:
// ONE decoder object serves the 4 codewords in order; its factor is restored to 0.9375 only after
// phase 0 and left at 0.875 by phases 1-2 (frame_v2.cpp:1359-1361,1412,1447,1470), so a codeword that
// follows one that needed phase >= 1 makes its FIRST decode at 0.875.
__global__ void fast_chain_kernel(FastDecodeArgs A) {
    int frame = blockIdx.x * blockDim.x + threadIdx.x;
    if (frame >= A.n_frames) return;
    const int bpc = A.c.bytes_per_cw;
    int f = 0;  // index into kFactors of the decoder's current factor (0 -> 0.9375, 1 -> 0.875)
    ria_decode_status* s = A.status + frame;
    for (int cw = 0; cw < 4; ++cw) {
        const unsigned fc = static_cast<unsigned>(frame) * 4u + cw;
        const CwResult r = A.res[fc];
        int accept = -1, attempts = 1, iters = r.iters[f];
        if (r.state[f] == 2) accept = f;
        else {
            if (A.flags & RIA_DECODE_PHASE0) {
                for (int t = 1; t <= 4 && accept < 0; ++t) { attempts++; if (r.state[t] == 2) { accept = t; iters = r.iters[t]; } }
                f = 0;
            }
            if (accept < 0 && (A.flags & RIA_DECODE_PERTURB)) {
                f = 1;
                unsigned e = atomicAdd(&A.ctl->n_entries, 1u);
                A.entries[e] = fc;
                A.best[e] = 0xFFFFFFFFu;
            }
        }
        uint8_t* out = A.info_out + static_cast<size_t>(fc) * bpc;
        if (accept >= 0) {
            const uint8_t* src = A.res_bytes + (static_cast<size_t>(fc) * kNumFactors + accept) * bpc;
            for (int b = 0; b < bpc; ++b) out[b] = src[b];
        } else {
            for (int b = 0; b < bpc; ++b) out[b] = 0;
        }
        s->cw_ok[cw] = accept >= 0 ? 1 : 0;
        s->iterations[cw] = static_cast<uint16_t>(iters);
        s->attempts[cw] = static_cast<uint8_t>(attempts);
    }
}

// perturbed decoder input of cascade attempt a (frame_v2.cpp:1415-1546) into st.li/st.lp
template <class S>
__device__ inline float fast_perturb(FastState<S>& st, const FastCode& c, const float* base_i, const float* base_p,
                                     uint32_t* mt, float* normal, int a, uint32_t h, int lane) {
    uint32_t seed; float sigma, factor; int kind;
    retry_transform_params(a, h, &seed, &sigma, &factor, &kind);
    normal648_wave(mt, normal, seed, lane);
    auto tf = [&](float v, float nz) {
        if (kind == 1) { v = (v < 10.0f) ? v : 10.0f; v = (-10.0f < v) ? v : -10.0f; }
        else if (kind == 2) v = v * 0.5f;
        else if (kind == 3) { v = (v < 6.0f) ? v : 6.0f; v = (-6.0f < v) ? v : -6.0f; }
        else if (kind == 4) v = (v >= 0.0f) ? 1.0f : -1.0f;
        else if (kind == 5) v = v * 0.25f;
        return v + (nz * sigma + 0.0f);
    };
#pragma unroll
    for (int r = 0; r < S::IR; ++r) { int j = lane + 64 * r; st.li[r] = (j < c.k) ? tf(base_i[r], normal[j]) : 0.0f; }
#pragma unroll
    for (int r = 0; r < S::RR; ++r) { int p = lane + 64 * r; st.lp[r] = (p < c.m) ? tf(base_p[r], normal[c.k + c.perm[p]]) : 0.0f; }
    wave_sync();
    return factor;
}

template <class S>
__device__ inline uint32_t fast_hash16(const FastState<S>& st) {  // frame_v2.cpp:1391-1396
    uint32_t h = 0;
    for (int j = 0; j < 16; ++j) {
        uint32_t u = f2u(__shfl(st.li[0], j));
        h ^= u + 0x9e3779b9u + (h << 6) + (h >> 2);
    }
    return h;
}

// ------------------------------------------------------------------------------------------------ kernel D2
// persistent single-wave workgroups; unit u = attempt-major (a = u / n_entries, e = u % n_entries)
template <class S>
__global__ __launch_bounds__(64) void fast_cascade_kernel(FastDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FastCode& c = A.c;
    const int lane = threadIdx.x;
    float* msg = reinterpret_cast<float*>(smem);
    const unsigned int n_entries = A.ctl->n_entries;
    const unsigned int total = n_entries * 34u;
    if (total == 0) return;
    FastState<S> st;
    fast_load_tables(st, c, lane);
    for (;;) {
        unsigned int u = 0;
        if (lane == 0) u = atomicAdd(&A.ctl->next_unit, 1u);
        u = __shfl(u, 0);
        if (u >= total) break;
        const unsigned int a = u / n_entries, e = u - a * n_entries;
        unsigned int b = __hip_atomic_load(&A.best[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b = __shfl(b, 0);
        if (b < a) continue;  // an earlier attempt already succeeded: this one can never be chosen
        const unsigned int fc = A.entries[e];
        fast_gather_llr(st, c, A.llr + static_cast<size_t>(fc >> 2) * A.llr_stride, A.gather, fc & 3, lane);
        float bi[S::IR], bp[S::RR];
#pragma unroll
        for (int r = 0; r < S::IR; ++r) bi[r] = st.li[r];
#pragma unroll
        for (int r = 0; r < S::RR; ++r) bp[r] = st.lp[r];
        uint32_t h = fast_hash16(st);
        float factor = fast_perturb(st, c, bi, bp, reinterpret_cast<uint32_t*>(msg), msg + 640, static_cast<int>(a), h, lane);
        bool ok;
        (void)fast_decode(st, c, smem, factor, c.max_iter, lane, &ok);
        if (ok && lane == 0) atomicMin(&A.best[e], a);
    }
}

// ------------------------------------------------------------------------------------------------ kernel D3
// one wave per cascade entry: re-run the winning attempt (deterministic) and publish its bytes
template <class S>
__global__ __launch_bounds__(64) void fast_finalize_kernel(FastDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FastCode& c = A.c;
    const int lane = threadIdx.x;
    const unsigned int n_entries = A.ctl->n_entries;
    float* msg = reinterpret_cast<float*>(smem);
    for (unsigned int e = blockIdx.x; e < n_entries; e += gridDim.x) {
        const unsigned int fc = A.entries[e], frame = fc >> 2, cw = fc & 3;
        const unsigned int a = A.best[e];
        ria_decode_status* s = A.status + frame;
        if (a >= 34u) {
            if (lane == 0) s->attempts[cw] = static_cast<uint8_t>(s->attempts[cw] + 34);
            continue;
        }
        FastState<S> st;
        fast_load_tables(st, c, lane);
        fast_gather_llr(st, c, A.llr + static_cast<size_t>(frame) * A.llr_stride, A.gather, cw, lane);
        float bi[S::IR], bp[S::RR];
#pragma unroll
        for (int r = 0; r < S::IR; ++r) bi[r] = st.li[r];
#pragma unroll
        for (int r = 0; r < S::RR; ++r) bp[r] = st.lp[r];
        uint32_t h = fast_hash16(st);
        float factor = fast_perturb(st, c, bi, bp, reinterpret_cast<uint32_t*>(msg), msg + 640, static_cast<int>(a), h, lane);
        bool ok;
        int it = fast_decode(st, c, smem, factor, c.max_iter, lane, &ok);
        fast_pack(st, c, smem, A.info_out + (static_cast<size_t>(frame) * 4 + cw) * c.bytes_per_cw,
                  c.bytes_per_cw, lane);
        if (lane == 0) {
            s->cw_ok[cw] = ok ? 1 : 0;  // ok is true by construction
            s->iterations[cw] = static_cast<uint16_t>(it);
            s->attempts[cw] = static_cast<uint8_t>(s->attempts[cw] + a + 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------ kernel D4
// frame validity: parseHeader + DataFrame::deserialize on the straight concatenation
// (frame_v2.cpp:1195-1252, :556-600); CW1..3 starting with 0xD5 go to the host restatement.
__global__ __launch_bounds__(256) void frame_validate_kernel(const uint8_t* __restrict__ info, int bpc, int n_frames,
                                                             const uint16_t* __restrict__ crc_bit,
                                                             const uint16_t* __restrict__ crc_init,
                                                             ria_decode_status* __restrict__ status) {
    __shared__ uint8_t flat_all[4][4 * 68];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frame = blockIdx.x * 4 + wave;
    if (frame >= n_frames) return;
    uint8_t* flat = flat_all[wave];
    ria_decode_status* s = status + frame;
    const bool all_ok = s->cw_ok[0] && s->cw_ok[1] && s->cw_ok[2] && s->cw_ok[3];
    int valid = 0, quirk = 0;
    if (all_ok) {
        for (int b = lane; b < 4 * bpc; b += 64) flat[b] = info[static_cast<size_t>(frame) * 4 * bpc + b];
        wave_sync();
        bool magic = flat[0] == 0x55 && flat[1] == 0x4C;
        int t = flat[2];
        bool ctl = (t == 0x10 || t == 0x11 || t == 0x16 || t == 0x17 || t == 0x20 || t == 0x21 || t == 0x15 || t == 0x40);
        int plen = (flat[13] << 8) | flat[14];
        int expected = ctl ? 20 : 17 + plen + 2;
        for (int cw = 1; cw < 4; ++cw) if (cw * bpc < expected && flat[cw * bpc] == 0xD5) quirk = 1;
        if (magic && !quirk) {
            if (ctl) {
                uint32_t crc = crc16_wave(flat, 18, crc_bit, crc_init, lane);
                valid = crc == static_cast<uint32_t>((flat[18] << 8) | flat[19]);
            } else {
                uint32_t hc = crc16_wave(flat, 15, crc_bit, crc_init, lane);
                if (hc == static_cast<uint32_t>((flat[15] << 8) | flat[16]) && expected <= 4 * bpc) {
                    uint32_t fc = crc16_wave(flat, expected - 2, crc_bit, crc_init, lane);
                    valid = fc == static_cast<uint32_t>((flat[expected - 2] << 8) | flat[expected - 1]);
                }
            }
        }
    }
    if (lane == 0) {
        s->frame_valid = static_cast<uint8_t>(valid);
        s->needs_recovery = static_cast<uint8_t>(all_ok && !valid);
        s->reserved[0] = static_cast<uint8_t>(quirk);
        s->reserved[1] = 0;
    }
}

// ------------------------------------------------------------------------------------------------ CRC recovery prep
// (frame_v2.cpp:1564-1880 runs on the host, ria_amd/csrc/frame_recovery.hpp; these kernels stage its inputs)
struct RecoveryArgs {
    FastDecodeArgs d;
    unsigned int* n_flagged;     // counter
    unsigned int* flagged;       // [n_frames] frame indices needing recovery
    unsigned int* n_list2;       // counter
    unsigned int* list2;         // [16*n_frames] (fc << 3) | factor index
    uint8_t* info_c;             // [n_flagged][4*bpc]
    float* rows_c;               // [n_flagged][4][648] decoder-order LLRs
    uint8_t* redec_ok;           // [n_flagged][4 factors][4 cw]   (factor order of the reference: 0.75, 0.625, 0.5, 0.875)
    uint8_t* redec_bytes;        // [n_flagged][4][4][bpc]
};

// one thread per frame: list the frames that need recovery and the (codeword, factor) decodes the
// fallback stage (frame_v2.cpp:1836-1866) will want that the result table does not hold yet
__global__ void recovery_list_kernel(RecoveryArgs R) {
    int frame = blockIdx.x * blockDim.x + threadIdx.x;
    if (frame >= R.d.n_frames || !R.d.status[frame].needs_recovery) return;
    R.flagged[atomicAdd(R.n_flagged, 1u)] = static_cast<unsigned>(frame);
    for (int cw = 0; cw < 4; ++cw) {
        unsigned fc = static_cast<unsigned>(frame) * 4u + cw;
        for (int f = 1; f <= 4; ++f)
            if (R.d.res[fc].state[f] == 0) R.list2[atomicAdd(R.n_list2, 1u)] = (fc << 3) | static_cast<unsigned>(f);
    }
}
template <class S>
__global__ __launch_bounds__(64) void recovery_fill_kernel(RecoveryArgs R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned u = blockIdx.x;
    if (u >= *R.n_list2) return;
    FastState<S> st;
    fast_load_tables(st, R.d.c, threadIdx.x);
    const unsigned e = R.list2[u];
    fast_unit(st, R.d, smem, e >> 3, static_cast<int>(e & 7u), threadIdx.x);
}
// one workgroup per flagged frame: compact copies of everything the host stage reads
__global__ __launch_bounds__(256) void recovery_gather_kernel(RecoveryArgs R) {
    const unsigned q = blockIdx.x;
    if (q >= *R.n_flagged) return;
    const unsigned frame = R.flagged[q];
    const int bpc = R.d.c.bytes_per_cw, ib = 4 * bpc;
    for (int i = threadIdx.x; i < ib; i += 256) R.info_c[static_cast<size_t>(q) * ib + i] = R.d.info_out[static_cast<size_t>(frame) * ib + i];
    const float* fl = R.d.llr + static_cast<size_t>(frame) * R.d.llr_stride;
    for (int i = threadIdx.x; i < 4 * 648; i += 256) R.rows_c[static_cast<size_t>(q) * 4 * 648 + i] = fl[R.d.gather[i]];
    const int forder[4] = {2, 3, 4, 1};   // reference tries 0.75, 0.625, 0.5, 0.875 (frame_v2.cpp:1837)
    if (threadIdx.x < 16) {
        int at = threadIdx.x >> 2, cw = threadIdx.x & 3;
        R.redec_ok[static_cast<size_t>(q) * 16 + threadIdx.x] = R.d.res[frame * 4u + cw].state[forder[at]] == 2 ? 1 : 0;
    }
    for (int i = threadIdx.x; i < 16 * bpc; i += 256) {
        int slot = i / bpc, b = i - slot * bpc, at = slot >> 2, cw = slot & 3;
        R.redec_bytes[static_cast<size_t>(q) * 16 * bpc + i] =
            R.d.res_bytes[(static_cast<size_t>(frame * 4u + cw) * kNumFactors + forder[at]) * bpc + b];
    }
}

// ------------------------------------------------------------------------------------------------ raw rows
template <class S>
__global__ __launch_bounds__(64) void fast_rows_kernel(FastCode c, const float* __restrict__ llr, int n_cw, int max_iter,
                                                       float factor, uint8_t* __restrict__ out, uint8_t* __restrict__ ok_out,
                                                       uint16_t* __restrict__ iters_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    FastState<S> st;
    fast_load_tables(st, c, lane);
    const int nb = (c.k + 7) / 8;
    for (int cw = blockIdx.x; cw < n_cw; cw += gridDim.x) {
        const float* l = llr + static_cast<size_t>(cw) * 648;
#pragma unroll
        for (int r = 0; r < S::IR; ++r) { int j = lane + 64 * r; st.li[r] = (j < c.k) ? l[j] : 0.0f; }
#pragma unroll
        for (int r = 0; r < S::RR; ++r) { int p = lane + 64 * r; st.lp[r] = (p < c.m) ? l[c.k + c.perm[p]] : 0.0f; }
        bool ok;
        int it = fast_decode(st, c, smem, factor, max_iter, lane, &ok);
        fast_pack(st, c, smem, out + static_cast<size_t>(cw) * nb, nb, lane);
        if (lane == 0) { ok_out[cw] = ok ? 1 : 0; iters_out[cw] = static_cast<uint16_t>(it); }
    }
}

}  // namespace ria
