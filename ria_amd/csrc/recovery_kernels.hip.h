// ria_amd/csrc/recovery_kernels.hip.h — the "LDPC false positive" recovery of v2::decodeFixedFrame
// (src/protocol/frame_v2.cpp:1564-1880) on the GPU: one wavefront per flagged frame.
//
// A frame is flagged when all four codewords converged but the frame CRC fails.  The reference then
//   stage 1  searches CRC-guided bit flips (header bits; single bit; CRC bits; pairs, triples and
//            quadruples of the 30 / 15 least reliable "suspect" bits, std::sort order), and
//   stage 2  re-decodes each codeword with min-sum factors {0.75, 0.625, 0.5, 0.875} and accepts the
//            first replacement that makes the frame verify.
// Everything is a search with a first-hit-wins order, so the wave evaluates the cheap linear filters
// (CRC syndromes are linear in the flipped bits) for 64 candidates at a time, takes the hits in the
// reference's loop order, and runs the full reassemble+verify only for those.  The one sequential
// piece, libstdc++'s introsort (its tie order decides the suspects), runs on lane 0 over LDS and only
// as far as the first 30 positions need (sort_exact.hpp).  The stage-2 decodes are the rows of the
// factor result table that phase 0 fills (ldpc_fast.hip.h), completed by recovery_fill_kernel.
// frame_recovery.hpp holds the same logic for the host; tests run both and compare.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ldpc_fast.hip.h"
#include "sort_exact.hpp"

namespace ria {

struct RecCtx {            // wave-uniform view of one frame's working set (all pointers into LDS)
    uint8_t* cw;           // [4][68] decoded codeword payloads, modified in place when recovered
    uint8_t* fd;           // [272] reassembled frame
    uint8_t* trial;        // [272]
    Suspect* sus;          // [4 * bpc * 8]
    int* stack;            // [192]
    uint16_t* sd;          // [32]
    uint16_t* lists;       // [2][sus_cap / 2 + 32] position lists of the partition passes
    int sus_cap;           // suspects the LDS array holds (rec_search returns kRecOverflow if a frame has more)
    unsigned long long* dbg;   // nullable: 8 cycle stamps of this frame's search (developer diagnostics)
    int bpc;
    const uint16_t* crc_bit;
    const uint16_t* crc_init;
    int lane;
};

__device__ inline bool rec_parse_header(const RecCtx& x, const uint8_t* d, int len, bool* ctl, int* plen) {  // frame_v2.cpp:1195-1252
    if (len < 20) return false;
    if (d[0] != 0x55 || d[1] != 0x4C) return false;
    const int t = d[2];
    *ctl = (t == 0x10 || t == 0x11 || t == 0x16 || t == 0x17 || t == 0x20 || t == 0x21 || t == 0x15 || t == 0x40);
    if (*ctl) {
        if (crc16_wave(d, 18, x.crc_bit, x.crc_init, x.lane) != static_cast<uint32_t>((d[18] << 8) | d[19])) return false;
        *plen = 0;
    } else {
        *plen = (d[13] << 8) | d[14];
        if (crc16_wave(d, 15, x.crc_bit, x.crc_init, x.lane) != static_cast<uint32_t>((d[15] << 8) | d[16])) return false;
    }
    return true;
}
// CodewordStatus::reassemble + reassembleCodewords (frame_v2.cpp:1030-1063, :959-989), all CWs decoded
__device__ inline int rec_reassemble(const RecCtx& x, uint8_t* out) {
    bool ctl; int plen;
    wave_sync();
    if (!rec_parse_header(x, x.cw, x.bpc, &ctl, &plen)) return 0;
    const int expected = ctl ? 20 : 17 + plen + 2;
    int n = 0;
    for (int i = 0; i < 4; ++i) {
        const int remaining = expected - n;
        if (remaining == 0) break;
        const uint8_t* src = x.cw + i * 68;
        int avail = x.bpc;
        if (i != 0 && src[0] == 0xD5) { src += 2; avail -= 2; }
        const int c = remaining < avail ? remaining : avail;
        for (int b = x.lane; b < c; b += 64) out[n + b] = src[b];
        n += c;
    }
    wave_sync();
    return n;
}
__device__ inline bool rec_verify(const RecCtx& x, const uint8_t* d, int len) {  // verifyFrame lambda, frame_v2.cpp:1583-1589
    bool ctl; int plen;
    if (len == 0 || !rec_parse_header(x, d, len, &ctl, &plen)) return false;
    if (ctl) return true;
    const int sz = 17 + plen + 2;
    if (len < sz) return false;
    return crc16_wave(d, sz - 2, x.crc_bit, x.crc_init, x.lane) == static_cast<uint32_t>((d[sz - 2] << 8) | d[sz - 1]);
}
__device__ inline bool rec_try(const RecCtx& x) {
    const int tl = rec_reassemble(x, x.trial);
    return rec_verify(x, x.trial, tl);
}
__device__ inline void rec_flip(const RecCtx& x, int byte_index, int bit) {   // wave-uniform arguments
    wave_sync();
    if (x.lane == 0) x.cw[byte_index] ^= static_cast<uint8_t>(1u << bit);
    wave_sync();
}

// ---- std::sort of the suspects, first 30 positions, by the whole wave (sort_exact.hpp: sort_exact_prefix_lists) ------
// One __unguarded_partition_pivot pass over v[first, last): median of three to v[first], then the two-list rule - the
// positions whose key is not below the pivot (ascending) and those whose key is not above it (descending) are ranked
// with ballot prefix counts, swap k exchanges the rank-k members of the two lists while they have not crossed.
// scratch: chunk masks in x.fd / x.trial (free during the sort), the two position lists in x.lists.
__device__ inline int rec_partition_wave(const RecCtx& x, int first, int last) {
    Suspect* v = x.sus;
    const int lane = x.lane;
    const int mid = first + (last - first) / 2;
    const float ka = v[first + 1].abs_llr, kb = v[mid].abs_llr, kc = v[last - 1].abs_llr;
    int pick;
    if (ka < kb) { pick = (kb < kc) ? mid : (ka < kc) ? last - 1 : first + 1; }
    else { pick = (ka < kc) ? first + 1 : (kb < kc) ? last - 1 : mid; }
    wave_sync();
    if (lane == 0) { const Suspect t = v[first]; v[first] = v[pick]; v[pick] = t; }
    wave_sync();
    const float pk = v[first].abs_llr;
    const int d0 = first + 1, m = last - first - 1, nch = (m + 63) >> 6;
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(x.fd);   // [nch][2]: not-below / not-above the pivot
    uint16_t* alist = x.lists;
    uint16_t* blist = x.lists + x.sus_cap / 2 + 32;
    const unsigned long long lt_mask = (1ull << lane) - 1ull, le_mask = lt_mask | (1ull << lane);
    int ns_total = 0;
    for (int c = 0; c < nch; ++c) {
        const int t = 64 * c + lane;
        const float key = (t < m) ? v[d0 + t].abs_llr : 0.0f;
        const unsigned long long mb = __ballot(t < m && !(key < pk)), ms = __ballot(t < m && !(pk < key));
        if (lane == 0) { masks[2 * c] = mb; masks[2 * c + 1] = ms; }
        ns_total += __popcll(ms);
    }
    wave_sync();
    // number of swaps: a not-below element of left rank a is swapped iff more than a not-above elements lie to its right
    int kk = 0, base_b = 0, base_s = 0;
    for (int c = 0; c < nch; ++c) {
        const unsigned long long mb = masks[2 * c], ms = masks[2 * c + 1];
        const int a = base_b + __popcll(mb & lt_mask), s_le = base_s + __popcll(ms & le_mask);
        kk += __popcll(__ballot(((mb >> lane) & 1ull) != 0ull && ns_total - s_le > a));
        base_b += __popcll(mb); base_s += __popcll(ms);
    }
    // the rank lists of the swapped elements, and the first not-below element that is not swapped (if any)
    const int kBig = 0x7fffffff;
    int t_a = kBig;
    base_b = 0; base_s = 0;
    for (int c = 0; c < nch; ++c) {
        const unsigned long long mb = masks[2 * c], ms = masks[2 * c + 1];
        const int t = 64 * c + lane;
        const bool boe = ((mb >> lane) & 1ull) != 0ull, soe = ((ms >> lane) & 1ull) != 0ull;
        const int a = base_b + __popcll(mb & lt_mask), b = ns_total - (base_s + __popcll(ms & le_mask));
        if (boe && a < kk) alist[a] = static_cast<uint16_t>(t);
        if (soe && b < kk) blist[b] = static_cast<uint16_t>(t);
        const unsigned long long here = __ballot(boe && a == kk);
        if (here) t_a = 64 * c + __builtin_ctzll(here);
        base_b += __popcll(mb); base_s += __popcll(ms);
    }
    wave_sync();
    for (int k = lane; k < kk; k += 64) {
        const int i = d0 + alist[k], j = d0 + blist[k];
        const Suspect t = v[i]; v[i] = v[j]; v[j] = t;
    }
    const int t_b = (kk > 0) ? static_cast<int>(blist[kk - 1]) : kBig;
    wave_sync();
    return d0 + (t_a < t_b ? t_a : t_b);
}
__device__ inline void rec_sort_suspects_wave(const RecCtx& x, int n, int want) {
    if (n <= 0) return;
    Suspect* v = x.sus;
    const int lane = x.lane;
    const int limit = (want >= n) ? n : want + 16;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    int sp = 1;
    if (lane == 0) { x.stack[0] = 0; x.stack[1] = n; x.stack[2] = 2 * lg; }
    wave_sync();
    while (sp > 0) {
        --sp;
        int first = x.stack[3 * sp], last = x.stack[3 * sp + 1], depth = x.stack[3 * sp + 2];
        while (last - first > 16) {
            if (first >= limit) break;
            if (depth == 0) {   // libstdc++ falls back to heapsort (rare): serial, as before
                wave_sync();
                if (lane == 0) sortx::heap_sort(v + first, v + last, suspect_lt);
                wave_sync();
                break;
            }
            --depth;
            const int cut = rec_partition_wave(x, first, last);
            if (cut < limit) {
                if (lane == 0) { x.stack[3 * sp] = cut; x.stack[3 * sp + 1] = last; x.stack[3 * sp + 2] = depth; }
                ++sp;
                wave_sync();
            }
            last = cut;
        }
    }
    // __final_insertion_sort over the first `fin` <= 46 positions = a stable sort of them: one lane per element
    const int fin = (limit < n) ? limit : n;
    wave_sync();
    Suspect mine = {0, 0.0f};
    if (lane < fin) mine = v[lane];
    int rank = 0;
    for (int j = 0; j < fin; ++j) {
        const float kj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.abs_llr), j));
        rank += (kj < mine.abs_llr || (j < lane && !(mine.abs_llr < kj))) ? 1 : 0;
    }
    wave_sync();
    if (lane < fin) v[rank] = mine;
    wave_sync();
}

// Stage 1 (frame_v2.cpp:1579-1834).  llr(c, i): decoder-order LLR i of codeword c.
// Returns 1 (recovered), 0 (not), or kRecOverflow: more suspects than the LDS array holds - nothing has been modified,
// the caller hands the frame to the big-LDS instance of the kernel.
constexpr int kRecOverflow = 2;
constexpr int kRecSmallCap = 320;     // suspects in the common instance's LDS (2.5 KB instead of 4*bpc*8 entries)
template <class LlrFn>
__device__ inline int rec_search(const RecCtx& x, LlrFn llr) {
    const int bpc = x.bpc, lane = x.lane;
    auto stamp = [&](int k) { if (x.dbg && lane == 0) x.dbg[k] = __builtin_readcyclecounter(); };
    stamp(0);
    const int flen = rec_reassemble(x, x.fd);
    if (flen == 0) {
        // case 1: header CRC error in CW0.  hdr_ok() = magic bytes match && CRC(bytes 0..14) == bytes 15..16;
        // both are linear in the flipped bits: per-bit deltas dh (header CRC syndrome) and dm (magic).
        const int tb = bpc * 8;
        const uint8_t* c0 = x.cw;
        const uint32_t hsyn = crc16_wave(c0, 15, x.crc_bit, x.crc_init, lane) ^ static_cast<uint32_t>((c0[15] << 8) | c0[16]);
        const uint32_t msyn = static_cast<uint32_t>((c0[0] << 8) | c0[1]) ^ 0x554Cu;
        auto dh = [&](int b) -> uint32_t {
            const int by = b >> 3, bit = b & 7;
            if (by < 15) return x.crc_bit[(14 - by) * 8 + bit];
            if (by == 15) return 1u << (8 + bit);
            if (by == 16) return 1u << bit;
            return 0u;
        };
        auto dm = [&](int b) -> uint32_t {
            const int by = b >> 3, bit = b & 7;
            if (by == 0) return 1u << (8 + bit);
            if (by == 1) return 1u << bit;
            return 0u;
        };
        // Both syndromes of a flipped bit b packed in one word, D(b) = dh(b) << 16 | dm(b): nonzero only inside the
        // first 17 bytes, and the same for every frame.  Lane l keeps D(l), D(l + 64), D(l + 128) in registers (bits
        // 136.. have D = 0), so the searches below are register compares: no table load inside the loops.
        const uint32_t d0 = (dh(lane) << 16) | dm(lane), d1 = (dh(lane + 64) << 16) | dm(lane + 64), d2 = (dh(lane + 128) << 16) | dm(lane + 128);
        const uint32_t syn2 = (hsyn << 16) | msyn;
        auto d_of = [&](int b) -> uint32_t {                 // uniform b
            if (b >= 192) return 0u;
            const uint32_t r = (b < 64) ? d0 : (b < 128) ? d1 : d2;
            return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(r), b & 63));
        };
        // candidates b2 in [from, tb) with D(b2) == want, ascending: f(b2) is called for each until it returns true
        auto for_matches = [&](uint32_t want, int from, auto f) -> bool {
            for (int base = 0; base < 192 && base < tb; base += 64) {
                const int b = base + lane;
                const uint32_t dv = (base == 0) ? d0 : (base == 64) ? d1 : d2;
                unsigned long long hits = __ballot(b >= from && b < tb && dv == want);
                while (hits) {
                    const int bb = base + __builtin_ctzll(hits);
                    hits &= hits - 1;
                    if (f(bb)) return true;
                }
            }
            if (want == 0u)                                   // bits beyond the first 24 bytes move neither syndrome
                for (int bb = (from > 192 ? from : 192); bb < tb; ++bb) if (f(bb)) return true;
            return false;
        };
        if (for_matches(syn2, 0, [&](int bb) {               // single bit, ascending bit index
                rec_flip(x, bb >> 3, bb & 7);
                if (rec_try(x)) return true;
                rec_flip(x, bb >> 3, bb & 7);
                return false; })) return true;
        for (int b1 = 0; b1 < tb; ++b1) {                    // pairs b1 < b2, b2 ascending inside b1
            const uint32_t want = syn2 ^ d_of(b1);
            if (want == 0u) {
                // Flipping b1 alone repairs the header (and did not verify above), so EVERY b2 outside the first 17
                // bytes passes the header filter: the reference tries them all, 184 full verifications.  With the header
                // fixed the verdict on (b1, b2) is the frame CRC alone, which is linear in b2 as well: one evaluation
                // of the syndrome after b1, then only the b2 whose delta cancels it are really tried (in order).
                rec_flip(x, b1 >> 3, b1 & 7);
                const int tl = rec_reassemble(x, x.trial);
                bool ctl2 = false; int plen2 = 0;
                const bool hdr = tl > 0 && rec_parse_header(x, x.trial, tl, &ctl2, &plen2);
                const int exp2 = 17 + plen2 + 2;
                bool done = false;
                if (hdr && !ctl2 && tl >= exp2) {
                    const uint32_t s2 = static_cast<uint32_t>((x.trial[exp2 - 2] << 8) | x.trial[exp2 - 1]) ^
                                        crc16_wave(x.trial, exp2 - 2, x.crc_bit, x.crc_init, lane);
                    const int from = (b1 + 1 > 136) ? b1 + 1 : 136;
                    for (int base = (from & ~63); base < tb && !done; base += 64) {
                        const int b2 = base + lane, by = b2 >> 3, bit = b2 & 7;
                        uint32_t dl = 0u;                      // change of the frame-CRC syndrome when bit b2 of codeword 0 flips
                        if (b2 < tb) {
                            if (by < exp2 - 2) dl = x.crc_bit[(exp2 - 3 - by) * 8 + bit];
                            else if (by < exp2) dl = 1u << (bit + 8 * (exp2 - 1 - by));
                        }
                        unsigned long long hits = __ballot(b2 >= from && b2 < tb && dl == s2);
                        while (hits && !done) {
                            const int bb = base + __builtin_ctzll(hits);
                            hits &= hits - 1;
                            rec_flip(x, bb >> 3, bb & 7);
                            if (rec_try(x)) done = true; else rec_flip(x, bb >> 3, bb & 7);
                        }
                    }
                } else if (hdr && ctl2) {
                    done = true;                              // a control frame verifies on its header alone (not reachable: the single-bit pass would have taken it)
                } else if (!hdr) {
                    // b1 repairs magic and header CRC, yet the frame does not parse: a control-type byte whose own CRC over
                    // bytes 0..17 (stored in 18..19) fails.  The reference still verifies every b2 > b1 that passes the
                    // filter (frame_v2.cpp:1617-1640), and a b2 in bytes 17..19 can complete such a control frame: take
                    // the exhaustive walk here (bits below 136 would break the filter again; rare, so no shortcut)
                    for (int bb = (b1 + 1 > 136) ? b1 + 1 : 136; bb < tb && !done; ++bb) {
                        rec_flip(x, bb >> 3, bb & 7);
                        if (rec_try(x)) done = true; else rec_flip(x, bb >> 3, bb & 7);
                    }
                }
                if (done) return true;
                rec_flip(x, b1 >> 3, b1 & 7);
                continue;
            }
            if (for_matches(want, b1 + 1, [&](int bb) {
                    rec_flip(x, b1 >> 3, b1 & 7);
                    rec_flip(x, bb >> 3, bb & 7);
                    if (rec_try(x)) return true;
                    rec_flip(x, bb >> 3, bb & 7);
                    rec_flip(x, b1 >> 3, b1 & 7);
                    return false; })) return true;
        }
        return false;
    }
    // case 2: frame CRC error
    bool ctl; int plen;
    if (!rec_parse_header(x, x.fd, flen, &ctl, &plen) || ctl) return false;
    const int expected = 17 + plen + 2;
    if (flen < expected) return false;
    const uint32_t stored = static_cast<uint32_t>((x.fd[expected - 2] << 8) | x.fd[expected - 1]);
    const int data_bytes = expected - 2, data_bits = data_bytes * 8;
    const uint32_t syn = stored ^ crc16_wave(x.fd, data_bytes, x.crc_bit, x.crc_init, lane);
    auto delta = [&](int p) -> uint32_t { return x.crc_bit[(data_bytes - 1 - (p >> 3)) * 8 + (p & 7)]; };
    auto fix = [&](int p) {   // frame bit -> codeword byte, the reference's own (un-stripped) mapping
        const int fb = p >> 3, c = fb / bpc;
        if (c < 4) rec_flip(x, c * 68 + fb % bpc, p & 7);
    };
    {   // single data bit: accepted without re-verification.  All table loads first, then the ordered test
        constexpr int kMaxRounds = (4 * 68 * 8 + 63) / 64;
        uint32_t dl[kMaxRounds];
#pragma unroll
        for (int q = 0; q < kMaxRounds; ++q) { const int p = 64 * q + lane; dl[q] = (p < data_bits) ? delta(p) : 0xFFFFFFFFu; }
#pragma unroll
        for (int q = 0; q < kMaxRounds; ++q) {
            const int base = 64 * q, p = base + lane;
            if (base >= data_bits) break;
            unsigned long long hits = __ballot(p < data_bits && dl[q] == syn && (p >> 3) / bpc < 4);
            if (hits) { fix(base + __builtin_ctzll(hits)); return true; }
        }
    }
    for (int bit = 0; bit < 16; ++bit)                        // a bit of the stored CRC itself
        if (syn == (1u << bit)) {
            const int fb = (bit >= 8) ? expected - 2 : expected - 1, c = fb / bpc;
            if (c < 4) { rec_flip(x, c * 68 + fb % bpc, bit & 7); return true; }
        }
    // suspects: decoded bit (LSB-first index, as the reference reads it) differs from the channel's hard decision
    stamp(1);
    int ns_all = 0;
    const int cw_bits = (bpc * 8 < 648) ? bpc * 8 : 648;
    for (int c = 0; c < 4; ++c) {
        // the codeword's soft bits first, all loads in flight together (up to 9 per lane), then the ordered compaction
        constexpr int kMaxRounds = 9;   // ceil(68 * 8 / 64)
        float lv[kMaxRounds];
#pragma unroll
        for (int q = 0; q < kMaxRounds; ++q) {
            const int i = 64 * q + lane;
            lv[q] = (i < cw_bits && ((c * bpc * 8 + i) >> 3) < data_bytes) ? llr(c, i) : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < kMaxRounds; ++q) {
            const int base = 64 * q;
            if (base >= cw_bits) break;
            const int i = base + lane;
            bool is = false;
            const float l = lv[q];
            int fbit = 0;
            if (i < cw_bits) {
                fbit = c * bpc * 8 + i;
                if ((fbit >> 3) < data_bytes) {
                    const int chb = (l < 0.0f) ? 1 : 0, db = (x.cw[c * 68 + (i >> 3)] >> (i & 7)) & 1;
                    is = chb != db;
                }
            }
            const unsigned long long mk = __ballot(is);
            if (is) {
                const int at = ns_all + __popcll(mk & ((1ull << lane) - 1ull));
                if (at < x.sus_cap) {
                    x.sus[at].frame_bit = fbit;
                    x.sus[at].abs_llr = fabs_(l);
                }
            }
            ns_all += __popcll(mk);
        }
    }
    if (ns_all > x.sus_cap) return kRecOverflow;
    wave_sync();
    stamp(2);
    rec_sort_suspects_wave(x, ns_all, 30);
    stamp(3);
    if (x.dbg && lane == 0) x.dbg[7] = static_cast<unsigned long long>(ns_all);
    const int ns = ns_all < 30 ? ns_all : 30;
    // CRC deltas of the (at most 30) suspects live in lane a of one register: the search loops below read them with
    // v_readlane instead of going through LDS
    const uint32_t my = (lane < ns) ? delta(x.sus[lane].frame_bit) : 0u;
    auto sd_at = [&](int a) -> uint32_t { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(my), a)); };   // uniform a
    auto sd_lane = [&](int a) -> uint32_t { return static_cast<uint32_t>(__shfl(static_cast<int>(my), a)); };                          // per-lane a
    auto sd_at_lane = [&](uint32_t mine, int p, int ln) -> uint32_t { (void)p; (void)ln; return mine; };                                // lane p's own delta
    auto attempt = [&](int n_idx, int a, int b, int c, int d) -> bool {
        const int idx[4] = {a, b, c, d};
        for (int q = 0; q < n_idx; ++q) fix(x.sus[idx[q]].frame_bit);
        if (rec_try(x)) return true;
        for (int q = 0; q < n_idx; ++q) fix(x.sus[idx[q]].frame_bit);
        return false;
    };
    // The reference walks pairs, then triples, then quadruples of suspects in lexicographic order and tries every
    // combination whose CRC deltas xor to the syndrome, first success wins.  Per class, one lane per PREFIX (the
    // combination without its last member, enumerated in lexicographic order) looks for the smallest last member that
    // completes it; hits are rare (2^-16 per combination), so the common case is a handful of select instructions
    // per prefix and no ballot at all.  Hits are then tried in the reference's order.
    auto try_class = [&](int n_idx, int n_members, int n_prefixes, auto prefix_of) -> bool {
        for (int base = 0; base < n_prefixes; base += 64) {
            const bool valid = base + lane < n_prefixes;
            const int p = valid ? base + lane : n_prefixes - 1;   // every lane decodes a prefix: the shuffles inside need all lanes active
            int ia = 0, ib = 0, ic = 0;
            uint32_t want = 0u;
            int last_from = n_members;
            prefix_of(p, ia, ib, ic, want, last_from);
            // every completion of this lane's prefix, ascending: a bit mask of the members d >= last_from with delta == want
            uint32_t found = 0u;
            for (int d = 1; d < n_members; ++d) {
                const uint32_t sdv = sd_at(d);
                if (valid && d >= last_from && sdv == want) found |= 1u << d;
            }
            unsigned long long lanes = __ballot(found != 0u);
            while (lanes) {                                   // prefixes in lexicographic order
                const int l = __builtin_ctzll(lanes);
                lanes &= lanes - 1;
                uint32_t f = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(found), l));
                const int pa = __builtin_amdgcn_readlane(ia, l), pb = __builtin_amdgcn_readlane(ib, l), pc = __builtin_amdgcn_readlane(ic, l);
                while (f) {                                   // completions in ascending order
                    const int d = __builtin_ctz(f);
                    f &= f - 1;
                    const bool ok = (n_idx == 2) ? attempt(2, pa, d, 0, 0) : (n_idx == 3) ? attempt(3, pa, pb, d, 0) : attempt(4, pa, pb, pc, d);
                    if (ok) return true;
                }
            }
        }
        return false;
    };
    // pairs (a, b): prefix = a
    if (try_class(2, ns, ns, [&](int p, int& ia, int& ib, int& ic, uint32_t& want, int& from) {
            (void)ib; (void)ic; ia = p; want = syn ^ sd_at_lane(my, p, lane); from = p + 1; })) return true;
    stamp(4);
    // triples (a, b, c): prefix = (a, b), a < b, lexicographic
    {
        const int np = ns * (ns - 1) / 2;
        if (try_class(3, ns, np, [&](int p, int& ia, int& ib, int& ic, uint32_t& want, int& from) {
                (void)ic;
                int a = 0, rem = p;                           // p -> (a, b): row a holds ns - 1 - a pairs
                while (rem >= ns - 1 - a) { rem -= ns - 1 - a; ++a; }
                ia = a; ib = a + 1 + rem;
                want = syn ^ sd_lane(ia) ^ sd_lane(ib); from = ib + 1; })) return true;
    }
    // quadruples of the first 15 (a, b, c, d): prefix = (a, b, c)
    {
        const int n4 = ns < 15 ? ns : 15;
        const int np = n4 * (n4 - 1) * (n4 - 2) / 6;
        if (try_class(4, n4, np, [&](int p, int& ia, int& ib, int& ic, uint32_t& want, int& from) {
                int a = 0, rem = p;                           // p -> (a, b, c), lexicographic
                for (;;) { const int m = n4 - 1 - a, cnt = m * (m - 1) / 2; if (rem < cnt) break; rem -= cnt; ++a; }
                int b = a + 1;
                for (;;) { const int cnt = n4 - 1 - b; if (rem < cnt) break; rem -= cnt; ++b; }
                ia = a; ib = b; ic = b + 1 + rem;
                want = syn ^ sd_lane(ia) ^ sd_lane(ib) ^ sd_lane(ic); from = ic + 1; })) return true;
    }
    return false;
}

struct RecoveryArgs {
    FastDecodeArgs d;
    unsigned int* n_flagged;     // counter
    unsigned int* flagged;       // [n_frames] frame indices needing recovery
    unsigned int* n_list2;       // counter
    unsigned int* list2;         // [4*n_frames used of 16*n_frames] (fc << 4) | mask of the factor indices 1..4 still to decode (bit f-1)
    unsigned int* n_stage2;      // counter
    unsigned int* stage2;        // [n_frames] frames whose stage 1 failed
    unsigned int* next_fill;     // counter: work queue head of recovery_fill_kernel
    unsigned long long* dbg;     // nullable: [n_frames][8] stage-1 cycle stamps (RIA_DEBUG_REC_STAMPS)
    unsigned int* n_overflow;    // counter
    unsigned int* overflow;      // [n_frames] frames with more suspects than the small stage-1 instance holds
    int list_units_now;          // recovery_list_kernel also lists the missing re-decodes (host-search path)
    // host-search staging (RIA_RECOVERY_HOST=1 only)
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
    if (!R.list_units_now) return;     // device path: only frames whose stage 1 fails ask for the re-decodes
    for (int cw = 0; cw < 4; ++cw) {
        unsigned fc = static_cast<unsigned>(frame) * 4u + cw, mask = 0;
        for (int f = 1; f <= 4; ++f)
            if (R.d.res[fc].state[f] == 0) mask |= 1u << (f - 1);
        if (mask) R.list2[atomicAdd(R.n_list2, 1u)] = (fc << 4) | mask;
    }
}
template <class S>
__global__ __launch_bounds__(64) void recovery_fill_kernel(RecoveryArgs R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned total = *R.n_list2 * 4u;
    if (blockIdx.x >= total) return;
    FastState<S> st;
    fast_load_tables(st, R.d.c, smem, threadIdx.x);
    unsigned guard = 0;
    for (;;) {   // persistent waves over an atomic queue (all-lane pop, bounded: see RIA_QUEUE_GUARD)
        RIA_QUEUE_GUARD(guard, total, R.d.ctl)
        unsigned u = atomicAdd(R.next_fill, threadIdx.x == 0 ? 1u : 0u);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= total) break;
        // one unit = one (codeword, factor) decode the result table lacks: units of at most max_iter iterations balance
        // the kernel's tail (a codeword that fails at every other factor used to be one unit of 4 x 80 iterations)
        const unsigned e = R.list2[u >> 2], fc = e >> 4;
        const int f = 1 + static_cast<int>(u & 3u);
        if (!((e >> (f - 1)) & 1u)) continue;
        const FastCode& c = R.d.c;
        const int lane = threadIdx.x;
        fast_gather_llr(st, c, R.d.llr + static_cast<size_t>(fc >> 2) * R.d.llr_stride, R.d.gather, fc & 3, lane);
        bool ok;
        const int it = fast_decode<S>(st, c, smem, kFactors[f], c.max_iter, lane, &ok);
        if (ok) fast_pack(st, c, smem, R.d.res_bytes + (static_cast<size_t>(fc) * kNumFactors + f) * c.bytes_per_cw, c.bytes_per_cw, lane);
        if (lane == 0) { R.d.res[fc].state[f] = ok ? 2 : 1; R.d.res[fc].iters[f] = static_cast<uint16_t>(it); }
    }
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


// LDS of one frame's wave: codewords / frame / trial / deltas (1 KB), sort stack, suspects.  BIG: room for every bit of
// the frame (4 * bpc * 8 suspects); the common instance holds kRecSmallCap of them
__host__ __device__ inline int recovery_sus_cap(int bpc, bool big) { return big ? 4 * bpc * 8 : (kRecSmallCap < 4 * bpc * 8 ? kRecSmallCap : 4 * bpc * 8); }
__host__ __device__ inline int recovery_lds_bytes(int bpc, bool big = true) {
    const int cap = recovery_sus_cap(bpc, big);
    return 1024 + cap * 8 + 192 * 4 + 2 * (cap / 2 + 32) * 2 + 64;
}

// ---- device path: stage 1 for every flagged frame; the frames it cannot recover queue the re-decodes the
// fallback needs; stage 2 runs for those only, after recovery_fill_kernel
__device__ inline void rec_setup(RecCtx& x, unsigned char* smem, const RecoveryArgs& R, int lane, bool big = true) {
    x.sus_cap = recovery_sus_cap(R.d.c.bytes_per_cw, big);
    x.cw = smem; x.fd = smem + 272; x.trial = smem + 544;
    x.sd = reinterpret_cast<uint16_t*>(smem + 816);
    x.stack = reinterpret_cast<int*>(smem + 1024);
    x.sus = reinterpret_cast<Suspect*>(smem + 1024 + 192 * 4);
    x.lists = reinterpret_cast<uint16_t*>(smem + 1024 + 192 * 4 + x.sus_cap * 8);
    x.bpc = R.d.c.bytes_per_cw; x.crc_bit = R.d.crc_bit; x.crc_init = R.d.crc_init; x.lane = lane;
    x.dbg = nullptr;
}
__device__ inline void rec_load_cw(const RecCtx& x, const uint8_t* info, int lane) {
    for (int i = lane; i < 4 * 68; i += 64) { const int c = i / 68, b = i - c * 68; x.cw[i] = (b < x.bpc) ? info[c * x.bpc + b] : 0; }
    wave_sync();
}
__device__ inline void rec_publish(const RecCtx& x, const RecoveryArgs& R, unsigned frame, uint8_t* info, bool good, int lane) {
    wave_sync();
    for (int i = lane; i < 4 * x.bpc; i += 64) { const int c = i / x.bpc, b = i - c * x.bpc; info[i] = good ? x.cw[c * 68 + b] : 0; }
    if (lane == 0) {
        ria_decode_status* s = R.d.status + frame;
        s->needs_recovery = 0;
        s->frame_valid = good ? 1 : 0;
        for (int c = 0; c < 4; ++c) s->cw_ok[c] = good ? 1 : 0;
    }
}

// The reference compares the decoded bits read LSB-first with the hard decisions of MSB-first soft bits
// (frame_v2.cpp:1700-1716), so about HALF of a frame's bits are "suspects" (640 of 1280 on the bench workload): the
// suspect array always needs room for every bit.
__global__ __launch_bounds__(64) void recovery_stage1_kernel(RecoveryArgs R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned q = blockIdx.x;
    if (q >= *R.n_flagged) return;
    const unsigned frame = R.flagged[q];
    const int lane = threadIdx.x;
    RecCtx x;
    rec_setup(x, smem, R, lane, true);
    if (R.dbg) x.dbg = R.dbg + static_cast<size_t>(frame) * 8;
    uint8_t* info = R.d.info_out + static_cast<size_t>(frame) * 4 * x.bpc;
    rec_load_cw(x, info, lane);
    const float* fl = R.d.llr + static_cast<size_t>(frame) * R.d.llr_stride;
    const uint16_t* gather = R.d.gather;
    const int good = rec_search(x, [&](int c, int i) { return fl[gather[c * 648 + i]]; });
    if (x.dbg && lane == 0) { x.dbg[6] = __builtin_readcyclecounter(); x.dbg[5] = static_cast<unsigned long long>(good); }
    if (good == kRecOverflow) return;   // cannot happen: the array holds every bit of the frame
    if (good) { rec_publish(x, R, frame, info, true, lane); return; }
    // queue the decodes the fallback still misses, one entry per codeword (mask of its missing factors): lanes 0..15 =
    // (codeword, factor) pairs, ONE reservation per frame on the shared counter (a counter bumped once per entry by
    // thousands of waves serialises)
    const unsigned fc = frame * 4u + static_cast<unsigned>((lane >> 2) & 3);
    const int f = 1 + (lane & 3);
    const bool need = lane < 16 && R.d.res[fc].state[f] == 0;
    const unsigned long long mk = __ballot(need);
    const unsigned mine = static_cast<unsigned>(mk >> (lane & 12)) & 15u;          // the four factor bits of this lane's codeword
    const bool writer = lane < 16 && (lane & 3) == 0 && mine != 0u;
    const unsigned long long wk = __ballot(writer);
    unsigned base = 0;
    if (lane == 0) {
        R.stage2[atomicAdd(R.n_stage2, 1u)] = frame;
        if (wk) base = atomicAdd(R.n_list2, static_cast<unsigned>(__popcll(wk)));
    }
    base = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(base)));
    if (writer) R.list2[base + __popcll(wk & ((1ull << lane) - 1ull))] = (fc << 4) | mine;
}

// Stage 2 (frame_v2.cpp:1836-1866): factors 0.75, 0.625, 0.5, 0.875 = kFactors[2, 3, 4, 1]; stage 1 left the
// codeword bytes untouched (every failed trial is reverted), so they are re-read from the output buffer
__global__ __launch_bounds__(64) void recovery_stage2_kernel(RecoveryArgs R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned q = blockIdx.x;
    if (q >= *R.n_stage2) return;
    const unsigned frame = R.stage2[q];
    const int lane = threadIdx.x;
    RecCtx x;
    rec_setup(x, smem, R, lane);
    const int bpc = x.bpc;
    uint8_t* info = R.d.info_out + static_cast<size_t>(frame) * 4 * bpc;
    rec_load_cw(x, info, lane);
    bool good = false;
    const int forder[4] = {2, 3, 4, 1};
    for (int at = 0; at < 4 && !good; ++at)
        for (int c = 0; c < 4 && !good; ++c) {
            const unsigned fc = frame * 4u + c;
            if (R.d.res[fc].state[forder[at]] != 2) continue;
            const uint8_t* rd = R.d.res_bytes + (static_cast<size_t>(fc) * kNumFactors + forder[at]) * bpc;
            bool diff = false;
            for (int b = lane; b < bpc; b += 64) diff = diff || rd[b] != x.cw[c * 68 + b];
            if (__ballot(diff) == 0ull) continue;
            uint8_t keep0 = 0, keep1 = 0;   // bpc <= 68 < 128: two bytes per lane
            wave_sync();
            if (lane < bpc) { keep0 = x.cw[c * 68 + lane]; x.cw[c * 68 + lane] = rd[lane]; }
            if (lane + 64 < bpc) { keep1 = x.cw[c * 68 + lane + 64]; x.cw[c * 68 + lane + 64] = rd[lane + 64]; }
            wave_sync();
            if (rec_try(x)) { good = true; break; }
            if (lane < bpc) x.cw[c * 68 + lane] = keep0;
            if (lane + 64 < bpc) x.cw[c * 68 + lane + 64] = keep1;
            wave_sync();
        }
    rec_publish(x, R, frame, info, good, lane);
}

}  // namespace ria
