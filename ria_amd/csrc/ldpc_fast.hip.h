// ria_amd/csrc/ldpc_fast.hip.h — register-resident flooding min-sum LDPC(648,k) decoder, core v5.
//
// Arithmetic: bit-exact LDPCDecoder::decodeBP (ldpc_decoder.cpp:154-260) for finite LLRs (|x| <= 1e30,
// see include/ria_gpu.h).  What is MI355X-specific is where things live and how work is scheduled:
//   * one wavefront per codeword, NO index loads and NO per-edge predication in the iteration loop.
//     Rows are grouped by degree into "rounds" of 64 rows with the SAME number of information edges
//     (the round structure of each code is a compile-time shape, checked against the generated H at
//     create time), so the unrolled loop body contains exactly the edges that exist;
//   * every LDS address is either lane*4 + immediate (the row's own c2v words, the column totals) or one
//     VGPR loaded once per decode (the gather addresses);
//   * the identity (parity) column k+i has a single edge, to row i, owned by the SAME lane: its
//     message never touches LDS and its variable update is fused into the check pass;
//   * the syndrome of iteration t is evaluated inside the check pass of iteration t+1 from the sign
//     bits of the column totals it reads anyway;
//   * information columns are ordered by degree too; a padded column slot reads a word that holds
//     +0.0f (x + 0.0f == x), idle lanes own private padding words, so nothing is masked;
//   * sign/magnitude work is integer work on the float bit patterns: zeros are kept canonical (+0.0)
//     so that "x < 0" IS the sign bit; magnitudes ride on the float source modifiers of v_min3 / v_med3 / v_min;
//   * the retry cascade (frame_v2.cpp:1415-1546) is a second, persistent kernel over a device-side
//     work list of (codeword, attempt) units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/ria_gpu.h"
#include "devmath.h"
#include "ldpc_kernels.hip.h"

namespace ria {

// ---- compile-time shape of a code ------------------------------------------------------------------
// NR row rounds of 64 rows (rows sorted by decreasing degree): round r is unrolled for ne(r) = its largest
// number of information edges, nm(r) = its smallest; slots s >= nm(r) are "mixed" (some lanes padded).
// NC information-column rounds with dv(r) = largest degree in the round (columns sorted by decreasing
// degree).  host_tables.hpp derives the same structure from H and ria_gpu_create refuses a code whose
// structure differs.
struct ShapeR12 {   // R1/2: m = 324, k = 324
    static constexpr int NR = 6, NC = 6;
    static constexpr int kCascadeWaves = 3;   // waves/SIMD the cascade kernel is held to (its register budget)
    static constexpr int kCv = 24;   // the row's own c2v words of slot groups < kCv stay in VGPRs between iterations (their LDS re-reads saved) as the register budget allows
    static constexpr int ne(int r) { constexpr int t[NR] = {6, 6, 5, 4, 2, 1}; return t[r]; }
    static constexpr int nm(int r) { constexpr int t[NR] = {6, 5, 4, 2, 1, 1}; return t[r]; }
    static constexpr int dv(int r) { constexpr int t[NC] = {5, 4, 4, 4, 4, 4}; return t[r]; }
};
struct ShapeR13 {   // R1/3 entry of the rate table: same (324,324) parameters, H seeded differently
    static constexpr int NR = 6, NC = 6;
    static constexpr int kCascadeWaves = 3;   // waves/SIMD the cascade kernel is held to (its register budget)
    static constexpr int kCv = 24;   // the row's own c2v words of slot groups < kCv stay in VGPRs between iterations (their LDS re-reads saved) as the register budget allows
    static constexpr int ne(int r) { constexpr int t[NR] = {6, 6, 5, 3, 3, 1}; return t[r]; }
    static constexpr int nm(int r) { constexpr int t[NR] = {6, 5, 3, 3, 1, 1}; return t[r]; }
    static constexpr int dv(int r) { constexpr int t[NC] = {6, 4, 4, 4, 4, 4}; return t[r]; }
};
struct ShapeR14 {   // m = 486, k = 162
    static constexpr int NR = 8, NC = 3;
    static constexpr int kCascadeWaves = 3;   // waves/SIMD the cascade kernel is held to (its register budget)
    static constexpr int kCv = 0;   // the row's own c2v words of slot groups < kCv stay in VGPRs between iterations (their LDS re-reads saved) as the register budget allows
    static constexpr int ne(int r) { constexpr int t[NR] = {6, 6, 5, 5, 4, 3, 2, 2}; return t[r]; }
    static constexpr int nm(int r) { constexpr int t[NR] = {6, 6, 5, 4, 3, 2, 2, 1}; return t[r]; }
    static constexpr int dv(int r) { constexpr int t[NC] = {13, 12, 12}; return t[r]; }
};
struct ShapeR23 {   // m = 216, k = 432
    static constexpr int NR = 4, NC = 7;
    static constexpr int kCascadeWaves = 3;   // waves/SIMD the cascade kernel is held to (its register budget)
    static constexpr int kCv = 24;   // the row's own c2v words of slot groups < kCv stay in VGPRs between iterations (their LDS re-reads saved) as the register budget allows
    static constexpr int ne(int r) { constexpr int t[NR] = {6, 6, 6, 6}; return t[r]; }
    static constexpr int nm(int r) { constexpr int t[NR] = {6, 6, 6, 4}; return t[r]; }
    static constexpr int dv(int r) { constexpr int t[NC] = {3, 3, 3, 3, 3, 3, 3}; return t[r]; }
};
struct ShapeR34 {   // m = 162, k = 486 (161 information columns have no edge at all)
    static constexpr int NR = 3, NC = 8;
    static constexpr int kCascadeWaves = 4;   // waves/SIMD the cascade kernel is held to (its register budget)
    static constexpr int kCv = 0;   // the row's own c2v words of slot groups < kCv stay in VGPRs between iterations (their LDS re-reads saved) as the register budget allows
    static constexpr int ne(int r) { constexpr int t[NR] = {6, 6, 6}; return t[r]; }
    static constexpr int nm(int r) { constexpr int t[NR] = {6, 6, 6}; return t[r]; }
    static constexpr int dv(int r) { constexpr int t[NC] = {3, 3, 3, 3, 3, 3, 0, 0}; return t[r]; }
};
struct ShapeR56 {   // m = 108, k = 540
    static constexpr int NR = 2, NC = 9;
    static constexpr int kCascadeWaves = 5;   // waves/SIMD the cascade kernel is held to (its register budget)
    static constexpr int kCv = 0;   // the row's own c2v words of slot groups < kCv stay in VGPRs between iterations (their LDS re-reads saved) as the register budget allows
    static constexpr int ne(int r) { constexpr int t[NR] = {6, 6}; return t[r]; }
    static constexpr int nm(int r) { constexpr int t[NR] = {6, 6}; return t[r]; }
    static constexpr int dv(int r) { constexpr int t[NC] = {3, 3, 3, 3, 0, 0, 0, 0, 0}; return t[r]; }
};

// A padded slot of a mixed-degree round (a lane whose row has no edge there) gathers its "column total" from the lane's
// big word: tot - c2v is then a positive magnitude no real message reaches (|LLR| <= 1e30 on entry, 50 afterwards), i.e.
// the neutral element of the row's sign product and of its two smallest magnitudes, whatever the slot's own (unused)
// c2v word holds.  Nothing is masked and nothing special is kept per iteration.
constexpr float kPadTotal = 1.7e38f;

template <class S>
struct ShapeInfo {
    static constexpr int row_off(int r) { int t = 0; for (int i = 0; i < r; ++i) t += S::ne(i); return t; }   // x 64 words
    static constexpr int col_off(int r) { int t = 0; for (int i = 0; i < r; ++i) t += S::dv(i); return t; }
    static constexpr int mix_off(int r) { int t = 0; for (int i = 0; i < r; ++i) t += S::ne(i) - S::nm(i); return t; }
    static constexpr int TS = row_off(S::NR);          // c2v slot groups (64 words each)
    static constexpr int TD = col_off(S::NC);          // column gather addresses per lane
    static constexpr int TM = mix_off(S::NR);          // mixed slots (per-lane store addresses)
    static constexpr int tot_word = 64 * TS;           // column totals [64*NC]
    static constexpr int zero_word = tot_word + 64 * S::NC;   // 64 words of +0.0f (one per lane)
    static constexpr int dump_word = zero_word + 64;          // 64 write-only words (one per lane)
    static constexpr int big_word = dump_word + 64;           // 64 words of kPadTotal (one per lane): "column total" of a padded row slot
    static constexpr int words = big_word + 64;
    // the same region doubles as mt19937 state + 648 normals during the retry cascade (>= 1296 words)
    static constexpr int lds_bytes = ((words < 1296 ? 1296 : words) * 4 + 15) & ~15;
};

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// Device view of host_tables.hpp CoreTables.
struct FastCode {
    int k, m, max_iter, bytes_per_cw;
    const uint16_t* row_addr;   // [TS][64]  byte address of tot[column] read by (slot group, lane); zero word when idle
    const uint16_t* col_addr;   // [TD][64]  byte address of the c2v word of the column's d-th edge (ascending check order)
    const uint16_t* check_at;   // [64*NR]   check index handled at position 64*r + lane, 0xFFFF idle
    const uint16_t* col_at;     // [64*NC]   information column at sorted position q, 0xFFFF idle
    const uint16_t* col_pos;    // [k]       sorted position of information column j
};

// Build-time knobs:
//   RIA_ADDTID   the lane-linear stores (c2v words, column totals) as ds_write_addtid_b32 (address = M0 + offset + 4*lane: no
//                address VGPR, 2 store-path cycles per dword against 3 for the pairs of ds_write2st64_b32).  Slower in
//                round 1 (when the loop was bound elsewhere), faster since the own-c2v words live in registers and the DS
//                issue rate is what binds: 278 -> 273 CU-cycles per codeword-iteration, step -1 %.  On.
//   RIA_CV_REGS  force every own-c2v word into registers whatever Shape::kCv says (experiments).
#ifndef RIA_CV_REGS
#define RIA_CV_REGS 0
#endif
#ifndef RIA_ADDTID
#define RIA_ADDTID 1
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr bool kCvRegs = RIA_CV_REGS != 0;
constexpr bool kAddTid = RIA_ADDTID != 0;

template <class S>
struct FastState {
    using I = ShapeInfo<S>;
    uint32_t rv[I::TS];            // gather addresses of the check pass
    uint32_t cs[I::TD > 0 ? I::TD : 1];   // gather addresses of the column pass
    float cv[I::TS];               // c2v of the row's own edges as written in the previous iteration (slots < NCV of fast_decode only)
    float li[S::NC];               // information-column LLRs (sorted position q = lane + 64 r)
    float lp[S::NR];               // identity-column LLRs (row position p = lane + 64 r)
    float pv[S::NR];               // v2c of the identity edges
    float pt[S::NR];               // totals of the identity columns
};

// A lane index the optimizer cannot see through: what a persistent loop derives from it (table indices, addresses of
// the unit's input) is recomputed per unit instead of being hoisted out of the loop into VGPRs that would then stay
// live across the decode (measured: 18-30 registers, i.e. a wave per SIMD or the c2v words that could live there).
__device__ __forceinline__ int opaque_lane(int lane) { asm volatile("" : "+v"(lane)); return lane; }
// LDS accesses by 32-bit LDS byte address (ds_read_b32 / ds_write_b32 vaddr [+ immediate offset])
using lds_float_ptr = __attribute__((address_space(3))) float*;
__device__ __forceinline__ uint32_t lds_addr(const unsigned char* p) {
    return static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const unsigned char*)p));
}
__device__ __forceinline__ float lds_f(uint32_t a) { return *(lds_float_ptr)(uintptr_t)a; }
__device__ __forceinline__ void lds_sf(uint32_t a, float v) { *(lds_float_ptr)(uintptr_t)a = v; }
// (the wave's LDS region starts at `lds`; absolute LDS addresses, one full VGPR each — kept opaque so
// that the compiler does not re-pack them into 16-bit halves and pay an unpack per access)
template <class S>
__device__ inline void fast_load_tables(FastState<S>& st, const FastCode& c, const unsigned char* lds, int lane) {
    using I = ShapeInfo<S>;
    const uint32_t base = lds_addr(lds);
#pragma unroll
    for (int i = 0; i < I::TS; ++i) { uint32_t a = base + c.row_addr[i * 64 + lane]; asm volatile("" : "+v"(a)); st.rv[i] = a; }
#pragma unroll
    for (int i = 0; i < I::TD; ++i) { uint32_t a = base + c.col_addr[i * 64 + lane]; asm volatile("" : "+v"(a)); st.cs[i] = a; }
}

// store to LDS word (base/4 + OFF/4 + lane): ds_write_addtid_b32 takes its address from M0 + offset + 4*lane, needs
// no address VGPR and moves half the dwords of ds_write_b32 to the LDS (2 instead of 4 cycles per wave-store).
// M0 is (re)loaded at every use: the compiler treats it as a scratch register of its own.
template <int OFF>
__device__ __forceinline__ void lds_store_tid(uint32_t base, float v) {
    static_assert(OFF >= 0 && OFF < 65536, "16-bit DS offset");
    if constexpr (!kAddTid) {
        *(lds_float_ptr)(uintptr_t)(base + static_cast<uint32_t>(threadIdx.x) * 4u + OFF) = v;
        return;
    }
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%2" : : "v"(v), "s"(base), "n"(OFF) : "memory", "m0");
}
// N stores STEP bytes apart behind ONE load of M0 (a row round's c2v words, the column totals)
template <int N, int OFF, int STEP>
__device__ __forceinline__ void lds_store_tid_n(uint32_t base, const float (&v)[N]) {
    static_assert(N >= 1 && N <= 6 && OFF >= 0 && OFF + (N - 1) * STEP < 65536, "up to six 16-bit DS offsets");
    if constexpr (!kAddTid) {
#pragma unroll
        for (int i = 0; i < N; ++i) *(lds_float_ptr)(uintptr_t)(base + static_cast<uint32_t>(threadIdx.x) * 4u + OFF + i * STEP) = v[i];
        return;
    }
    if constexpr (N == 1)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%2" : : "v"(v[0]), "s"(base), "n"(OFF) : "memory", "m0");
    else if constexpr (N == 2)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%3\n\tds_write_addtid_b32 %1 offset:%4"
                     : : "v"(v[0]), "v"(v[1]), "s"(base), "n"(OFF), "n"(OFF + STEP) : "memory", "m0");
    else if constexpr (N == 3)
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%4\n\tds_write_addtid_b32 %1 offset:%5\n\tds_write_addtid_b32 %2 offset:%6"
                     : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "s"(base), "n"(OFF), "n"(OFF + STEP), "n"(OFF + 2 * STEP) : "memory", "m0");
    else if constexpr (N == 4)
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%5\n\tds_write_addtid_b32 %1 offset:%6\n\tds_write_addtid_b32 %2 offset:%7\n\tds_write_addtid_b32 %3 offset:%8"
                     : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(base), "n"(OFF), "n"(OFF + STEP), "n"(OFF + 2 * STEP), "n"(OFF + 3 * STEP) : "memory", "m0");
    else if constexpr (N == 5)
        asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%6\n\tds_write_addtid_b32 %1 offset:%7\n\tds_write_addtid_b32 %2 offset:%8\n\tds_write_addtid_b32 %3 offset:%9\n\tds_write_addtid_b32 %4 offset:%10"
                     : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "s"(base), "n"(OFF), "n"(OFF + STEP), "n"(OFF + 2 * STEP), "n"(OFF + 3 * STEP), "n"(OFF + 4 * STEP) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %6\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%7\n\tds_write_addtid_b32 %1 offset:%8\n\tds_write_addtid_b32 %2 offset:%9\n\tds_write_addtid_b32 %3 offset:%10\n\tds_write_addtid_b32 %4 offset:%11\n\tds_write_addtid_b32 %5 offset:%12"
                     : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "s"(base), "n"(OFF), "n"(OFF + STEP), "n"(OFF + 2 * STEP), "n"(OFF + 3 * STEP), "n"(OFF + 4 * STEP), "n"(OFF + 5 * STEP) : "memory", "m0");
}
__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// Selections on magnitudes.  Written with builtins so that the compiler folds |x| into the VOP3 source modifiers and
// schedules them freely (inline asm costs an s_nop before every dependent instruction: the hazard recogniser has to
// assume a transcendental).  Operands are results of float arithmetic or of these selections, i.e. canonical, so the
// IEEE-mode quieting of v_min_f32 never needs a separate instruction.
__device__ __forceinline__ float med3_abs(float x, float b, float c) { return __builtin_amdgcn_fmed3f(__builtin_fabsf(x), b, c); }
__device__ __forceinline__ float min_abs(float x, float b) { return __builtin_fminf(__builtin_fabsf(x), b); }
__device__ __forceinline__ float min3_abs(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c)); }
__device__ __forceinline__ float med3_abs3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(__builtin_fabsf(a), __builtin_fabsf(b), __builtin_fabsf(c)); }
__device__ __forceinline__ float min2_abs(float a, float b) { return __builtin_fminf(__builtin_fabsf(a), __builtin_fabsf(b)); }
__device__ __forceinline__ float max2_abs(float a, float b) { return __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)); }
__device__ __forceinline__ float min_raw(float a, float b) { return __builtin_fminf(a, b); }
// The two smallest magnitudes of w[0..N-1] (with multiplicity: a repeated minimum gives mid == lo) as a small
// tournament of 3-input selections instead of the 2-ops-per-element running update: triples give (min3, med3),
// two (lo, mid) pairs merge as lo = min(l1, l2), mid = med3(l1, l2, min(m1, m2)), a single element joins as
// mid = med3(|e|, lo, mid), lo = min(|e|, lo).  Selections only: the results are the same values bit for bit.
template <int N>
__device__ __forceinline__ void two_smallest_abs(const float (&w)[N], float& lo, float& mid) {
    static_assert(N >= 2 && N <= 7, "row degree");
    if constexpr (N == 2) { lo = min2_abs(w[0], w[1]); mid = max2_abs(w[0], w[1]); }
    else {
        lo = min3_abs(w[0], w[1], w[2]); mid = med3_abs3(w[0], w[1], w[2]);
        if constexpr (N >= 6) {
            const float l2 = min3_abs(w[3], w[4], w[5]), m2 = med3_abs3(w[3], w[4], w[5]);
            const float x = min_raw(mid, m2);
            mid = __builtin_amdgcn_fmed3f(lo, l2, x);
            lo = min_raw(lo, l2);
            if constexpr (N == 7) { mid = med3_abs(w[6], lo, mid); lo = min_abs(w[6], lo); }
        } else {
            if constexpr (N >= 4) { mid = med3_abs(w[3], lo, mid); lo = min_abs(w[3], lo); }
            if constexpr (N >= 5) { mid = med3_abs(w[4], lo, mid); lo = min_abs(w[4], lo); }
        }
    }
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (mask & a) | (~mask & b); }   // v_bfi_b32

// decoder input domain (include/ria_gpu.h): NaN -> +1e30 exactly as the reference demapper's own
// std::min/std::max clip would map it, |x| capped at 1e30, zeros made +0.0 (the reference only ever
// tests `x < 0` and |x|, for which -0.0 and +0.0 are the same number).
__device__ __forceinline__ float llr_canon(float x) {
    x = (x < 1e30f) ? x : 1e30f;
    x = (-1e30f < x) ? x : -1e30f;
    return x + 0.0f;
}

// Runs the decoder on the LLRs in st.li / st.lp.  Returns LDPCDecoder::lastIterations(); *ok = converged.
// On return the information-column totals of the accepted (or last) iteration are in LDS (fast_pack).
//
// Message flow per iteration (same arithmetic as ldpc_decoder.cpp:176-236, different bookkeeping):
//   check pass   for every edge: v2c = clamp(tot[var] - c2v_old) (iteration 0: c2v_old = 0 and the clamp
//                bounds are +-inf, so v2c = llr), parity ^= sign(tot[var]) (-> syndrome of the PREVIOUS
//                iteration for free), then the min-sum update written back to the edge's slot as c2v;
//   column pass  tot[j] = llr[j] + sum of its c2v slots in ascending check order -> ONE store per column.
// Zeros stay canonical: x - y and x + y only give -0.0 from (-0.0, +-0.0) operands, the LLRs are
// canonical and a c2v of -0.0 can only be added to or subtracted from a canonical value.
#ifndef RIA_SINGLE_PREFETCH
#define RIA_SINGLE_PREFETCH 2
#endif
template <class S, int NCV = (kCvRegs ? 1024 : S::kCv), int PF = RIA_SINGLE_PREFETCH>
__device__ inline int fast_decode(FastState<S>& st, const FastCode& c, unsigned char* __restrict__ lds,
                                  float factor, int max_iter, int lane, bool* ok) {
    using I = ShapeInfo<S>;
    const uint32_t lane4 = lds_addr(lds) + static_cast<uint32_t>(lane) * 4u;
    const uint32_t kAbs = 0x7fffffffu;
    const uint32_t m0base = lds_addr(lds);
    // previous c2v := 0 (-FLT_MAX on the padded lanes of mixed slots), tot := channel LLR of the
    // information columns, zero words
    static_for<0, S::NR>([&](auto R_) __attribute__((always_inline)) {
        constexpr int r = decltype(R_)::value;
#pragma unroll
        for (int s = 0; s < S::ne(r); ++s) {
            const float v0 = 0.0f;
            if (I::row_off(r) + s < NCV) st.cv[I::row_off(r) + s] = v0;
            else lds_sf(lane4 + 256u * (I::row_off(r) + s), v0);
        }
    });
#pragma unroll
    for (int r = 0; r < S::NC; ++r) lds_sf(lane4 + 4u * (I::tot_word + 64 * r), st.li[r]);
    lds_sf(lane4 + 4u * I::zero_word, 0.0f);
    lds_sf(lane4 + 4u * I::big_word, kPadTotal);
#pragma unroll
    for (int r = 0; r < S::NR; ++r) { st.pv[r] = st.lp[r]; st.pt[r] = 0.0f; }
    wave_sync();
    float hi = __builtin_inff();
    int it = 0;
    bool success = false;
    for (; it < max_iter; ++it) {
        uint32_t syn = 0;
        // the total gathers of round r + PF are issued before round r is computed (LDS operations keep their program
        // order: without this every round starts by waiting out an LDS round trip)
        float tt[I::TS];
        auto gather_round = [&](auto G_) __attribute__((always_inline)) {
            constexpr int g = decltype(G_)::value;
            if constexpr (g < S::NR) {
#pragma unroll
                for (int s = 0; s < S::ne(g); ++s) tt[I::row_off(g) + s] = lds_f(st.rv[I::row_off(g) + s]);
            }
        };
        static_for<0, PF>([&](auto G_) __attribute__((always_inline)) { gather_round(G_); });
        static_for<0, S::NR>([&](auto R_) __attribute__((always_inline)) {
            constexpr int r = decltype(R_)::value;
            constexpr int NE = S::ne(r);
            constexpr int off = I::row_off(r);
            gather_round(std::integral_constant<int, r + PF>{});
            float t[NE], cold[NE];
#pragma unroll
            for (int s = 0; s < NE; ++s) {
                t[s] = tt[off + s];
                if (off + s < NCV) cold[s] = st.cv[off + s];
                else cold[s] = lds_f(lane4 + 256u * (off + s));
            }
            // |x| rides on the float source modifiers (free), min1/min2 are v_med3_f32: med3(a, b, -inf) =
            // min(a, b) without the sNaN quieting v_min_f32 would need.  Everything here is finite.
            const float pvr = st.pv[r];                   // identity edge: the last edge of the row
            uint32_t par = f2u(st.pt[r]);
            uint32_t sgn = f2u(pvr);
            float min1, min2;
            float v[NE + 1];                               // v[NE] = the identity edge
#pragma unroll
            // v2c = clamp(tot - c2v, -50, 50) (no clamp in the first iteration: hi = inf).  The clamp only ever acts
            // on magnitudes, and the row needs just the two smallest of them and the signs: min_k min(|x_k|, hi) =
            // min(min_k |x_k|, hi), so the tournament runs on the unclamped differences and its two results are
            // clamped (2 operations per row instead of one per edge); min(|x|, min2) below is unchanged by the
            // clamp of x because min2 <= hi.
            for (int s = 0; s < NE; ++s) v[s] = t[s] - cold[s];
            v[NE] = pvr;
            two_smallest_abs<NE + 1>(v, min1, min2);
            min1 = min_raw(min1, hi);
            min2 = min_raw(min2, hi);
#pragma unroll
            for (int s = 0; s < NE; s += 2) {             // parity of the hard bits / product of the signs: xor3
                if (s + 1 < NE) {
                    par = __builtin_amdgcn_bitop3_b32(par, f2u(t[s]), f2u(t[s + 1]), 0x96);
                    sgn = __builtin_amdgcn_bitop3_b32(sgn, f2u(v[s]), f2u(v[s + 1]), 0x96);
                } else {
                    par ^= f2u(t[s]);
                    sgn ^= f2u(v[s]);
                }
            }
            syn |= par;
            // min over the OTHER edges of edge e is min2 if |v_e| == min1 (ties: min2 == min1), else min1, with the sign
            // product of the others.  Selection without a compare: x = sign(v_e) * min(|v_e|, min2) is one v_med3
            // (v, min2, -min2) and carries the bits of min1 in the first case, of min2 in the second; xor with
            // bits(min1) ^ bits(min2) ^ row-sign swaps the magnitude to the other candidate and turns the edge's own
            // sign into the product of the others.  (sign * min_abs) * factor is then the reference's own order.
            const uint32_t dS = bfi(kAbs, f2u(min1) ^ f2u(min2), sgn);
            float o[NE];
#pragma unroll
            for (int s = 0; s < NE; ++s) o[s] = u2f(dS ^ f2u(__builtin_amdgcn_fmed3f(v[s], min2, -min2))) * factor;
            lds_store_tid_n<NE, 256 * off, 256>(m0base, o);
            static_for<0, NE>([&](auto S_) __attribute__((always_inline)) {
                constexpr int s = decltype(S_)::value;
                if constexpr (off + s < NCV) st.cv[off + s] = o[s];
            });
            {   // identity column: degree 1, total = llr + c2v, v2c = clamp(total - c2v)
                const float c2v = u2f(dS ^ f2u(__builtin_amdgcn_fmed3f(pvr, min2, -min2))) * factor;
                const float tot = st.lp[r] + c2v;
                st.pv[r] = tot - c2v;                    // clamped where it is used (see above)
                st.pt[r] = tot;
            }
        });
        if (it > 0 && __ballot(static_cast<int>(syn) < 0) == 0ull) { success = true; --it; break; }
        hi = 50.0f;
        wave_sync();
        // information columns: tot = llr + sum of c2v in ascending check order
        // (packed FP32 for the differences, the factor products and these sums was measured: fewer instructions, same time:
        // a v_pk_* costs the issue time of the two scalar operations it replaces)
        // (columns of degree 0 only exist in the last rounds: their totals stay at the channel value written at the start)
        constexpr int NCE = []() constexpr { int n = 0; for (int r = 0; r < S::NC; ++r) if (S::dv(r) > 0) n = r + 1; return n; }();
        float totv[NCE > 0 ? NCE : 1];
        static_for<0, NCE>([&](auto R_) __attribute__((always_inline)) {
            constexpr int r = decltype(R_)::value;
            constexpr int DV = S::dv(r);
            constexpr int off = I::col_off(r);
            float cv[DV > 0 ? DV : 1];
#pragma unroll
            for (int d = 0; d < DV; ++d) cv[d] = lds_f(st.cs[off + d]);
            float tot = st.li[r];
#pragma unroll
            for (int d = 0; d < DV; ++d) tot = tot + cv[d];
            totv[r] = tot;
        });
        static_for<0, (NCE + 5) / 6>([&](auto G_) __attribute__((always_inline)) {     // groups of up to six stores behind one M0 load
            constexpr int g0 = 6 * decltype(G_)::value, gn = (NCE - g0 < 6) ? NCE - g0 : 6;
            float tv[gn];
#pragma unroll
            for (int i = 0; i < gn; ++i) tv[i] = totv[g0 + i];
            lds_store_tid_n<gn, 4 * (I::tot_word + 64 * g0), 256>(m0base, tv);
        });
        wave_sync();
    }
    if (!success) {   // syndrome of the final iteration
        uint32_t syn = 0;
        static_for<0, S::NR>([&](auto R_) __attribute__((always_inline)) {
            constexpr int r = decltype(R_)::value;
            constexpr int off = I::row_off(r);
            uint32_t par = f2u(st.pt[r]);
#pragma unroll
            for (int s = 0; s < S::ne(r); ++s) par ^= f2u(lds_f(st.rv[off + s]));
            syn |= par;
        });
        if (max_iter > 0 && __ballot(static_cast<int>(syn) < 0) == 0ull) { success = true; it = max_iter - 1; }
    }
    *ok = success;
    return success ? it : max_iter;
}

// information hard bits -> bytes, MSB first (decodeBP's packing), from the column totals in LDS;
// scratch: the c2v area of the wave's LDS region (>= 8*ceil(k/64) bytes, free once the decode is over)
template <class S>
__device__ inline void fast_pack(const FastState<S>& st, const FastCode& c, unsigned char* lds, uint8_t* out, int nbytes,
                                 int lane) {
    using I = ShapeInfo<S>;
    (void)st;
    lane = opaque_lane(lane);
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(lds);
    const int nr = (c.k + 63) / 64;
    for (int r = 0; r < nr; ++r) {
        const int j = lane + 64 * r;
        bool bit = false;
        if (j < c.k) bit = (f2u(lds_f(lds_addr(lds) + 4u * (I::tot_word + c.col_pos[j]))) >> 31) != 0u;
        unsigned long long mk = __ballot(bit);
        if (lane == 0) masks[r] = mk;
    }
    wave_sync();
    for (int b = lane; b < nbytes; b += 64) out[b] = static_cast<uint8_t>(__brev(static_cast<unsigned>(lds[b])) >> 24);
    wave_sync();
}

// ------------------------------------------------------------------------------------------------ args
// Cascade result slot of one entry.  A wave whose attempt succeeded AND lowered best[e] takes the lock, checks that it
// still is the best, and stores its bytes; the overall first successful attempt always passes both tests, and every
// store happens under the lock, so after the kernel the slot holds exactly that attempt's result.
struct CascadeWin { unsigned int lock; unsigned int iters; uint8_t bytes[72]; };   // 80 bytes

struct DecodeCtl {          // zeroed by hipMemsetAsync before every decode call
    unsigned int n_entries; // codewords that need the cascade
    unsigned int next_unit; // cascade work queue head
    unsigned int n_list1;   // codewords that need the other four min-sum factors
    unsigned int next_z;    // phase-0 work queue head
    unsigned int queue_fault;  // set by a persistent wave whose queue loop ran past its bound (see kQueueGuard)
    unsigned int pad_[3];
};
// Queue pops are written in the all-lane form (lane 0 adds 1, the others 0, then readfirstlane).  The round-1 hang had this
// cause, read off the ISA of tools/probes/queue_pop_probe.hip: with the pop written `u = 0; if (lane == 0) u = atomicAdd(..);
// u = readfirstlane(u)` (or __shfl(u, 0)), hipcc 7.2 threads the lane != 0 edge of that branch THROUGH the convergent
// broadcast, folding the constant 0 those lanes carry: the single loop becomes a nest in which lane 0 alone leaves the inner
// loop to pop the queue, while lanes 1..63 re-enter the loop body with EXEC = ~lane0 and `v_mov_b32 u, 0` +
// `v_readfirstlane` of the first ACTIVE lane - unit 0 again, for ever, since 0 < total.  The all-lane form has no
// divergent branch in front of the broadcast, so there is nothing to thread.  Every persistent loop is bounded as well: a
// wave can pop at most `total` live units plus one terminating index; a loop that goes round more often records it in
// DecodeCtl::queue_fault and leaves, and the call's last kernel turns a recorded fault into "every frame failed"
// (decode_fault_mark) - a broken queue loop is reported through the status records, it neither hangs the GPU nor returns
// undecoded frames as if they had been tried.
#define RIA_QUEUE_GUARD(guard, total, ctl) \
    if (++(guard) > (total) + 2u) { if (threadIdx.x == 0) atomicExch(&(ctl)->queue_fault, 1u); break; }

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
    CascadeWin* win;         // [4*n_frames]  result of the best successful cascade attempt so far, per entry
    float* staged;           // [4*n_frames][kStageFloats]  de-interleaved decoder input of list1 entry i, in register order
    unsigned int* l1idx;     // [4*n_frames]  codeword -> its list1 index (valid for listed codewords only)
    unsigned int* l1hash;    // [4*n_frames]  list1 entry -> hash of its first 16 soft bits (seed of the perturbation RNG)
};

// A codeword that needs more than its first decode is decoded up to 4 + 34 more times.  Its soft bits are gathered
// from the frame's interleaved stream once (fast_stage_kernel) into the order the decoder's registers want them
// (value r of lane l at [r*64 + l], information part then parity part), so every later unit is (NC+NR) coalesced
// loads instead of a walk over the whole 10 KB frame.
constexpr int kStageFloats = 12 * 64;
constexpr int kStageFrameFloats = 2688;   // largest interleaved frame: 2592 coded bits rounded up to whole symbols (D8PSK: 2650)

constexpr float kIdleRowLlr = 1e30f;   // idle row lanes: a parity bit that is certainly 0 keeps their syndrome term 0

// gathers one codeword's decoder input straight from the frame's interleaved soft bits
template <class S>
__device__ inline void fast_gather_llr(FastState<S>& st, const FastCode& c, const float* fl, const uint16_t* gather,
                                       int cw, int lane) {
    lane = opaque_lane(lane);
#pragma unroll
    for (int r = 0; r < S::NC; ++r) {
        const uint32_t j = c.col_at[lane + 64 * r];
        st.li[r] = (j != 0xFFFFu) ? llr_canon(fl[gather[cw * 648 + j]]) : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < S::NR; ++r) {
        const uint32_t i = c.check_at[lane + 64 * r];
        st.lp[r] = (i != 0xFFFFu) ? llr_canon(fl[gather[cw * 648 + c.k + i]]) : kIdleRowLlr;
    }
}

template <class S>
__device__ inline void fast_load_staged(FastState<S>& st, const float* __restrict__ src, int lane) {
    static_assert((S::NC + S::NR) * 64 <= kStageFloats, "stage slot too small");
    lane = opaque_lane(lane);
#pragma unroll
    for (int r = 0; r < S::NC; ++r) st.li[r] = src[r * 64 + lane];
#pragma unroll
    for (int r = 0; r < S::NR; ++r) st.lp[r] = src[(S::NC + r) * 64 + lane];
}

// decode codeword `fc` (= frame*4 + cw) with factor index f and record the result; staged != nullptr: the
// codeword's de-interleaved input (fast_stage_kernel) instead of the gather from the frame
template <class S, int NCV = S::kCv>
__device__ inline void fast_unit(FastState<S>& st, const FastDecodeArgs& A, unsigned char* lds, unsigned fc, int f, int lane,
                                 const float* staged = nullptr) {
    const FastCode& c = A.c;
    if (staged) fast_load_staged(st, staged, lane);
    else fast_gather_llr(st, c, A.llr + static_cast<size_t>(fc >> 2) * A.llr_stride, A.gather, fc & 3, lane);
    bool ok;
    int it = fast_decode<S, NCV>(st, c, lds, kFactors[f], c.max_iter, lane, &ok);
    if (ok) fast_pack(st, c, lds, A.res_bytes + (static_cast<size_t>(fc) * kNumFactors + f) * c.bytes_per_cw,
                      c.bytes_per_cw, lane);
    if (lane == 0) { A.res[fc].state[f] = ok ? 2 : 1; A.res[fc].iters[f] = static_cast<uint16_t>(it); }
}

// ------------------------------------------------------------------------------------------------ kernel P
// first decode of every codeword (factor 0.9375): one single-wave workgroup per codeword, so the
// hardware dispatcher balances converging (few iterations) and hopeless (80 iterations) codewords.
// XCD-aware order: workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8) and each XCD has its own
// L2, while the four codewords of a frame gather from the SAME 10 KB of interleaved soft bits; the map
// below gives the four codewords of frame 8g+x the block ids 32g + 8cw + x, i.e. one XCD per frame.
// The first decodes mostly converge within a few iterations: their time is input gathers and set-up latency, and a fourth
// wave per SIMD (no own-c2v words in registers: 122 VGPRs at R1/2) hides more of it than the saved LDS re-reads are worth
// (measured: decode stage -0.12 ms per 25 000 frames); the 80-iteration retry kernels are the other way round (Shape::kCv).
#ifndef RIA_PRIMARY_NCV
#define RIA_PRIMARY_NCV 0
#endif
template <class S>
__global__ __launch_bounds__(64) void fast_primary_kernel(FastDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const unsigned b = blockIdx.x, g = b >> 5, r = b & 31u, cw = r >> 3, x = r & 7u;
    const unsigned frame = 8u * g + x;
    if (frame >= static_cast<unsigned>(A.n_frames)) return;
    FastState<S> st;
    fast_load_tables(st, A.c, smem, lane);
    fast_unit<S, RIA_PRIMARY_NCV>(st, A, smem, 4u * frame + cw, 0, lane);
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
        if (failed) {
            const unsigned i = atomicAdd(&A.ctl->n_list1, 1u);
            A.list1[i] = fc;
            A.l1idx[fc] = i;
        }
    }
}

// hash of the bit patterns of the codeword's first 16 LLRs (frame_v2.cpp:1391-1396)
__device__ inline uint32_t fast_hash16(const float* fl, const uint16_t* gather, int cw, int lane) {
    const uint32_t mine = f2u(fl[gather[cw * 648 + (lane & 15)]]);
    uint32_t h = 0;
    for (int j = 0; j < 16; ++j) {
        uint32_t u = __shfl(mine, j);
        h ^= u + 0x9e3779b9u + (h << 6) + (h >> 2);
    }
    return h;
}

// one workgroup per list1 entry: the frame's soft bits come in as coalesced rows into LDS (one pass over the 10 KB
// frame: scattered 4-byte misses to the same line are not merged on their way to HBM), the codeword is gathered from
// LDS in register order (see kStageFloats), and the perturbation seed hash of its first 16 soft bits is kept
template <class S>
__global__ __launch_bounds__(256) void fast_stage_kernel(FastDecodeArgs A) {
    __shared__ float fr[kStageFrameFloats];
    const FastCode& c = A.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned n = A.ctl->n_list1;
    const int nfl = A.llr_stride < kStageFrameFloats ? A.llr_stride : kStageFrameFloats;
    for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
        const unsigned fc = A.list1[i], cw = fc & 3u;
        const float* fl = A.llr + static_cast<size_t>(fc >> 2) * A.llr_stride;
        __syncthreads();
        for (int q = tid; q < nfl; q += 256) fr[q] = fl[q];
        __syncthreads();
        float* dst = A.staged + static_cast<size_t>(i) * kStageFloats;
        for (int r = wave; r < S::NC + S::NR; r += 4) {
            float v;
            if (r < S::NC) {
                const uint32_t j = c.col_at[lane + 64 * r];
                v = (j != 0xFFFFu) ? llr_canon(fr[A.gather[cw * 648 + j]]) : 0.0f;
            } else {
                const uint32_t k = c.check_at[lane + 64 * (r - S::NC)];
                v = (k != 0xFFFFu) ? llr_canon(fr[A.gather[cw * 648 + c.k + k]]) : kIdleRowLlr;
            }
            dst[r * 64 + lane] = v;
        }
        if (wave == 0) {
            const uint32_t h = fast_hash16(fr, A.gather, cw, lane);
            if (lane == 0) A.l1hash[i] = h;
        }
    }
}

// ------------------------------------------------------------------------------------------------ kernel Z
// persistent single-wave workgroups pull (list1 entry, factor 1..4) units from an atomic queue (ctl->next_z): the
// list length is only known on the device, so the grid is a fixed size; 80-iteration units and 5-iteration units mix
template <class S>
__global__ __launch_bounds__(64) void fast_phase0_kernel(FastDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const unsigned total = A.ctl->n_list1 * 4u;
    if (blockIdx.x >= total) return;
    FastState<S> st;
    fast_load_tables(st, A.c, smem, lane);
    unsigned guard = 0;
    for (;;) {   // persistent waves over an atomic queue (ctl->next_z), all-lane pop (see RIA_QUEUE_GUARD)
        RIA_QUEUE_GUARD(guard, total, A.ctl)
        unsigned u = atomicAdd(&A.ctl->next_z, lane == 0 ? 1u : 0u);
        u = __builtin_amdgcn_readfirstlane(u);   // scalar: the loop control and every address derived from u stay uniform
        if (u >= total) break;
        fast_unit<S>(st, A, smem, A.list1[u >> 2], 1 + static_cast<int>(u & 3u), lane, A.staged + static_cast<size_t>(u >> 2) * kStageFloats);
    }
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
                A.win[e].lock = 0u;
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
    lane = opaque_lane(lane);
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
    for (int r = 0; r < S::NC; ++r) {
        const uint32_t j = c.col_at[lane + 64 * r];
        st.li[r] = (j != 0xFFFFu) ? llr_canon(tf(base_i[r], normal[j])) : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < S::NR; ++r) {
        const uint32_t i = c.check_at[lane + 64 * r];
        st.lp[r] = (i != 0xFFFFu) ? llr_canon(tf(base_p[r], normal[c.k + i])) : kIdleRowLlr;
    }
    wave_sync();
    return factor;
}

// ------------------------------------------------------------------------------------------------ kernel D2
// persistent single-wave workgroups; unit u = attempt-major (a = u / n_entries, e = u % n_entries)
template <class S>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(S::kCascadeWaves))) void fast_cascade_kernel(FastDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FastCode& c = A.c;
    const int lane = threadIdx.x;
    float* msg = reinterpret_cast<float*>(smem);
    const unsigned int n_entries = A.ctl->n_entries;
    const unsigned int total = n_entries * 34u;
    if (total == 0) return;
    FastState<S> st;
    fast_load_tables(st, c, smem, lane);
    unsigned guard = 0;
    for (;;) {
        RIA_QUEUE_GUARD(guard, total, A.ctl)
        unsigned int u = atomicAdd(&A.ctl->next_unit, lane == 0 ? 1u : 0u);   // all lanes take part: see RIA_QUEUE_GUARD
        u = __builtin_amdgcn_readfirstlane(u);   // scalar: the loop control and every address derived from u stay uniform
        if (u >= total) break;
        const unsigned int a = u / n_entries, e = u - a * n_entries;
        unsigned int b = __hip_atomic_load(&A.best[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < a) continue;  // an earlier attempt already succeeded: this one can never be chosen
        const unsigned int fc = A.entries[e];
        const unsigned int li = A.l1idx[fc];
        fast_load_staged(st, A.staged + static_cast<size_t>(li) * kStageFloats, lane);
        float bi[S::NC], bp[S::NR];
#pragma unroll
        for (int r = 0; r < S::NC; ++r) bi[r] = st.li[r];
#pragma unroll
        for (int r = 0; r < S::NR; ++r) bp[r] = st.lp[r];
        const uint32_t h = A.l1hash[li];
        float factor = fast_perturb(st, c, bi, bp, reinterpret_cast<uint32_t*>(msg), msg + 640, static_cast<int>(a), h, lane);
        bool ok;
        const int it = fast_decode<S>(st, c, smem, factor, c.max_iter, lane, &ok);
        if (ok) {
            // all-lane forms here too (no lane-0-only atomic inside the loop): only lane 0's operand can change the word
            unsigned int prev = atomicMin(&A.best[e], lane == 0 ? a : 0xFFFFFFFFu);
            prev = __builtin_amdgcn_readfirstlane(prev);
            if (a < prev) {   // best so far: publish under the entry's lock (held for one 40..68-byte store)
                CascadeWin* w = A.win + e;
                // the holder is a resident wave inside a 40..68-byte store, so the lock frees within microseconds; the spin
                // is bounded all the same (2^20 x 64 cycles = 28 ms): past that the wave records a fault and moves on
                bool mine = false;
                for (unsigned spin = 0; spin < (1u << 20); ++spin) {   // wave-uniform spin: lane 0's compare-and-swap decides
                    unsigned got = atomicCAS(&w->lock, 0u, lane == 0 ? 1u : 0u);
                    got = __builtin_amdgcn_readfirstlane(got);
                    if (got == 0u) { mine = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!mine) { if (lane == 0) atomicExch(&A.ctl->queue_fault, 1u); continue; }
                __threadfence();
                unsigned int cur = __hip_atomic_load(&A.best[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cur = __builtin_amdgcn_readfirstlane(cur);
                if (cur == a) {
                    fast_pack(st, c, smem, w->bytes, c.bytes_per_cw, lane);
                    if (lane == 0) w->iters = static_cast<unsigned int>(it);
                }
                __threadfence();
                if (lane == 0) atomicExch(&w->lock, 0u);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ kernel D3
// one thread per cascade entry: publish the winning attempt's bytes (stored by the cascade under the entry's lock)
__global__ __launch_bounds__(256) void fast_finalize_kernel(FastDecodeArgs A) {
    const unsigned int n_entries = A.ctl->n_entries;
    const int bpc = A.c.bytes_per_cw;
    for (unsigned int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += gridDim.x * blockDim.x) {
        const unsigned int fc = A.entries[e], frame = fc >> 2, cw = fc & 3;
        const unsigned int a = A.best[e];
        ria_decode_status* s = A.status + frame;
        if (a >= 34u) { s->attempts[cw] = static_cast<uint8_t>(s->attempts[cw] + 34); continue; }
        const CascadeWin* w = A.win + e;
        uint8_t* out = A.info_out + (static_cast<size_t>(frame) * 4 + cw) * bpc;
        for (int b = 0; b < bpc; ++b) out[b] = w->bytes[b];
        s->cw_ok[cw] = 1;
        s->iterations[cw] = static_cast<uint16_t>(w->iters);
        s->attempts[cw] = static_cast<uint8_t>(s->attempts[cw] + a + 1);
    }
}

// ------------------------------------------------------------------------------------------------ kernel D4
// frame validity: parseHeader + DataFrame::deserialize on the straight concatenation
// (frame_v2.cpp:1195-1252, :556-600); CW1..3 starting with 0xD5 go to the host restatement.
// A work-queue fault recorded during this call (DecodeCtl::queue_fault, see RIA_QUEUE_GUARD): some units were never decoded,
// so no result of the call can be trusted - every frame is reported failed, with the marker 0xEE in reserved[1].
__device__ inline bool decode_fault_mark(const DecodeCtl* ctl, ria_decode_status* s, uint8_t* info, int info_bytes, int lane) {
    if (__hip_atomic_load(&ctl->queue_fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return false;
    for (int b = lane; b < info_bytes; b += 64) info[b] = 0;
    if (lane == 0) {
        for (int cw = 0; cw < 4; ++cw) s->cw_ok[cw] = 0;
        s->frame_valid = 0; s->needs_recovery = 0; s->reserved[0] = 0; s->reserved[1] = 0xEE;
    }
    return true;
}
constexpr uint8_t kDecodeFaultMarker = 0xEE;

__global__ __launch_bounds__(256) void frame_validate_kernel(uint8_t* __restrict__ info, int bpc, int n_frames,
                                                             const uint16_t* __restrict__ crc_bit,
                                                             const uint16_t* __restrict__ crc_init,
                                                             ria_decode_status* __restrict__ status, const DecodeCtl* ctl) {
    __shared__ uint8_t flat_all[4][4 * 68];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frame = blockIdx.x * 4 + wave;
    if (frame >= n_frames) return;
    uint8_t* flat = flat_all[wave];
    ria_decode_status* s = status + frame;
    if (decode_fault_mark(ctl, s, info + static_cast<size_t>(frame) * 4 * bpc, 4 * bpc, lane)) return;
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

// last kernel of a call whose CRC recovery ran after frame_validate_kernel: the same fault check over every frame
__global__ __launch_bounds__(256) void decode_fault_kernel(const DecodeCtl* ctl, ria_decode_status* __restrict__ status, uint8_t* __restrict__ info,
                                                           int bpc, int n_frames) {
    const int lane = threadIdx.x & 63, frame = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= n_frames) return;
    (void)decode_fault_mark(ctl, status + frame, info + static_cast<size_t>(frame) * 4 * bpc, 4 * bpc, lane);
}

// ------------------------------------------------------------------------------------------------ raw rows
// RIA_ROWS_NCV / RIA_ROWS_WAVES: build-time knobs of the core microbenchmark (tools/bench_core.py drives this kernel):
// own-c2v words kept in registers and waves per SIMD of the raw-rows kernel
#ifndef RIA_ROWS_NCV
#define RIA_ROWS_NCV (kCvRegs ? 1024 : S::kCv)
#endif
#ifdef RIA_ROWS_WAVES
#define RIA_ROWS_ATTR __attribute__((amdgpu_waves_per_eu(RIA_ROWS_WAVES, RIA_ROWS_WAVES)))
#else
#define RIA_ROWS_ATTR
#endif
template <class S>
__global__ __launch_bounds__(64) RIA_ROWS_ATTR void fast_rows_kernel(FastCode c, const float* __restrict__ llr, int n_cw, int max_iter,
                                                       float factor, uint8_t* __restrict__ out, uint8_t* __restrict__ ok_out,
                                                       uint16_t* __restrict__ iters_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    FastState<S> st;
    fast_load_tables(st, c, smem, lane);
    const int nb = (c.k + 7) / 8;
    for (int cw = blockIdx.x; cw < n_cw; cw += gridDim.x) {
        const float* l = llr + static_cast<size_t>(cw) * 648;
        const int ln = opaque_lane(lane);
#pragma unroll
        for (int r = 0; r < S::NC; ++r) { const uint32_t j = c.col_at[ln + 64 * r]; st.li[r] = (j != 0xFFFFu) ? llr_canon(l[j]) : 0.0f; }
#pragma unroll
        for (int r = 0; r < S::NR; ++r) { const uint32_t i = c.check_at[ln + 64 * r]; st.lp[r] = (i != 0xFFFFu) ? llr_canon(l[c.k + i]) : kIdleRowLlr; }
        bool ok;
        int it = fast_decode<S, RIA_ROWS_NCV>(st, c, smem, factor, max_iter, lane, &ok);
        fast_pack(st, c, smem, out + static_cast<size_t>(cw) * nb, nb, lane);
        if (lane == 0) { ok_out[cw] = ok ? 1 : 0; iters_out[cw] = static_cast<uint16_t>(it); }
    }
}

// ------------------------------------------------------------------------------------------------ robust single CW
// robustDecodeSingleCW (streaming_decoder.cpp:1028-1058): a fresh decoder at the recommended iteration count, first
// decode at factor 0.9375, then 0.875 / 0.75 / 0.625 / 0.5 until one converges.  One wave per codeword walks the
// factors itself (the LLRs stay in its registers).  tries = decodes made (1..5); iters = lastIterations() of the last.
template <class S>
__global__ __launch_bounds__(64) void fast_robust_kernel(FastCode c, const float* __restrict__ llr, int n_cw,
                                                         uint8_t* __restrict__ out, uint8_t* __restrict__ ok_out,
                                                         uint16_t* __restrict__ iters_out, uint8_t* __restrict__ tries_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    FastState<S> st;
    fast_load_tables(st, c, smem, lane);
    const int nb = (c.k + 7) / 8;
    for (int cw = blockIdx.x; cw < n_cw; cw += gridDim.x) {
        const float* l = llr + static_cast<size_t>(cw) * 648;
#pragma unroll
        for (int r = 0; r < S::NC; ++r) { const uint32_t j = c.col_at[lane + 64 * r]; st.li[r] = (j != 0xFFFFu) ? llr_canon(l[j]) : 0.0f; }
#pragma unroll
        for (int r = 0; r < S::NR; ++r) { const uint32_t i = c.check_at[lane + 64 * r]; st.lp[r] = (i != 0xFFFFu) ? llr_canon(l[c.k + i]) : kIdleRowLlr; }
        bool ok = false;
        int it = 0, tries = 0;
#pragma unroll 1
        for (int f = 0; f < kNumFactors && !ok; ++f) {
            it = fast_decode(st, c, smem, kFactors[f], c.max_iter, lane, &ok);
            ++tries;
        }
        fast_pack(st, c, smem, out + static_cast<size_t>(cw) * nb, nb, lane);
        if (lane == 0) {
            ok_out[cw] = ok ? 1 : 0; iters_out[cw] = static_cast<uint16_t>(it);
            if (tries_out) tries_out[cw] = static_cast<uint8_t>(tries);
        }
    }
}

}  // namespace ria
