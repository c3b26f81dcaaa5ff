// ria_amd/csrc/sync_kernels.hip.h — acquisition correlators.
//
// zc_detect_kernel: sync::ZCSync::detect (src/sync/zc_sync.hpp:192-391, correlate :485-626,
// computeCorrelationMag :441-482) for a batch of capture buffers, bit-exact.
//   * one workgroup (4 waves) per buffer: the buffer is mixed to baseband ONCE into LDS (the reference
//     mixes it once per root and again inside every computeCorrelationMag call — same values);
//   * one wave per ZC root (1, 3, 5, 7 = PING, PONG, DATA, CONTROL): the four roots run concurrently;
//   * the correlation at one lag is a left-to-right float sum over 1016 samples — that order is the
//     result, so a lag is ONE lane's serial loop and the 64 lanes of the wave take 64 lags at a time
//     (coarse grid of step 31, then the +-31 fine window, then the three lags the rep1/rep2 logic and the
//     CFO estimate can ask for: peak-1016, peak, peak+1016);
//   * the interpolated reference sample i is wave-uniform -> scalar loads; baseband reads are ds_read_b64
//     with a lane stride of 31 complex samples (62 words: conflict-free over 64 banks);
//   * "first maximum" scans become wave arg-max reductions with the index as tie-break.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ria_gpu.h"
#include "devmath.h"

namespace ria {

constexpr int kZcN = 127, kZcUp = 8, kZcRep = kZcN * kZcUp, kZcGap = 480, kZcPreamble = 2 * kZcRep + kZcGap;
constexpr int kZcLdsBuf = 16384, kZcMaxBuf = 1 << 20;   // longest buffer mixed down into LDS / longest buffer at all

struct ZcArgs {
    const float* samples;      // [n_buffers][stride]
    long long stride;
    int buf_len;
    int n_buffers;
    float threshold;
    uint32_t root_mask;
    const float* known_cfo;    // [n_buffers] or null
    const float2* ref;         // [4][1016] interpolated ZC reference per root (host_tables.hpp build_zc_reference)
    ria_zc_result* out;
    float2* bb_ws;             // [n_buffers][buf_len] baseband workspace of the long-buffer form (kLds = false)
};

struct ZcRootOut { float combined; int timing; int has_cfo; float cfo; };

// (value, index) arg-max over the wave: larger value wins, equal values -> smaller index (= first maximum
// of a left-to-right scan with a strict '>')
__device__ inline void wave_argmax_first(float& v, int& idx) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float ov = __shfl_xor(v, off);
        int oi = __shfl_xor(idx, off);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

struct ZcSum { float sr, si, e; };
// sum over one repetition of bb * conj(ref) and the received energy at `lag` (zc_sync.hpp:539-556)
__device__ inline ZcSum zc_corr_at(const float2* __restrict__ bb, const float2* __restrict__ ref, int lag) {
    float sr = 0.0f, si = 0.0f, e = 0.0f;
    const float2* b = bb + lag;
#pragma unroll 4
    for (int i = 0; i < kZcRep; ++i) {
        const float2 z = ref[i];
        const float2 x = b[i];
        const float c = z.x, d = -z.y;
        sr += x.x * c - x.y * d;
        si += x.x * d + x.y * c;
        e += x.x * x.x + x.y * x.y;
    }
    return {sr, si, e};
}

// kLds: the mixed-down buffer lives in LDS (buffers up to ~19 000 samples: the batched acquisition sweeps); otherwise in a
// global workspace served by L1 / L2 (the host's connected-mode search windows: 31 000 - 48 000 samples,
// streaming_decoder.cpp:424-431).  Same arithmetic, same order.
template <bool kLds>
__global__ __launch_bounds__(256) void zc_detect_kernel(ZcArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* bb = kLds ? reinterpret_cast<float2*>(smem) : A.bb_ws + static_cast<size_t>(blockIdx.x) * A.buf_len;
    ZcRootOut* ro = reinterpret_cast<ZcRootOut*>(smem + (kLds ? static_cast<size_t>(A.buf_len) * sizeof(float2) : 0));
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = A.buf_len;
    const float* rx = A.samples + static_cast<long long>(blockIdx.x) * A.stride;
    ria_zc_result* out = A.out + blockIdx.x;
    if (n < kZcRep) {
        if (tid == 0) { out->detected = 0; out->frame_type = 255; out->start_sample = -1; out->root_detected = -1; out->correlation = 0.f; out->cfo_hz = 0.f; out->snr_estimate = 0.f; out->reserved = 0.f; }
        return;
    }
    const float known = A.known_cfo ? A.known_cfo[blockIdx.x] : 0.0f;
    const float f = 1500.0f + known;
    // baseband[i] = rx[i] * (cos(phase), sin(phase)), phase = -2*pi*f*t in double, rounded to float (:503-508)
    for (int i = tid; i < n; i += 256) {
        const float t = fdiv(static_cast<float>(i), 48000.0f);
        const float phase = static_cast<float>(static_cast<double>(-2.0f) * 3.14159265358979323846 * static_cast<double>(f) * static_cast<double>(t));
        const float x = rx[i];
        bb[i] = make_float2(x * cosf_glibc(phase), x * sinf_glibc(phase));
    }
    __syncthreads();
    const float ref_energy = static_cast<float>(kZcRep);
    if (A.root_mask & (1u << w)) {
        const float2* ref = A.ref + w * kZcRep;
        const int corr_len = n - kZcRep + 1, step = kZcRep / 32;
        const int n_coarse = (corr_len + step - 1) / step;
        // ---- coarse grid: running first-maximum of the coarse criterion and of |normalised correlation|
        float cbest = 0.0f; int cpos = 0;      // coarse_best_mag / coarse_best_pos
        float pbest = 0.0f; int ppos = 0;      // peak over the stored (normalised) correlation entries
        for (int base = 0; base < n_coarse; base += 64) {
            const int k = base + lane;
            const bool valid = k < n_coarse;
            const int lag = valid ? k * step : 0;
            const ZcSum s = zc_corr_at(bb, ref, lag);
            const float denom = fsqrt(s.e * ref_energy);
            float mag = 0.0f, pm = 0.0f;
            if (valid && denom > 1e-10f) {
                mag = fdiv(hypotf_glibc(s.sr, s.si), denom);
                pm = hypotf_glibc(fdiv(s.sr, denom), fdiv(s.si, denom));
            }
            float v = valid ? mag : -1.0f; int idx = lag;
            wave_argmax_first(v, idx);
            if (v > cbest) { cbest = v; cpos = idx; }
            v = valid ? pm : -1.0f; idx = lag;
            wave_argmax_first(v, idx);
            if (v > pbest) { pbest = v; ppos = idx; }
        }
        // ---- fine window around the coarse peak
        const int fine_start = (cpos - step < 0) ? 0 : cpos - step;
        const int fine_end = (cpos + step + 1 > corr_len) ? corr_len : cpos + step + 1;
        {
            const bool valid = fine_start + lane < fine_end;
            const int lag = valid ? fine_start + lane : 0;
            const ZcSum s = zc_corr_at(bb, ref, lag);
            const float denom = fsqrt(s.e * ref_energy);
            float pm = 0.0f;
            if (valid && denom > 1e-10f) pm = hypotf_glibc(fdiv(s.sr, denom), fdiv(s.si, denom));
            float v = valid ? pm : -1.0f; int idx = lag;
            wave_argmax_first(v, idx);
            // merge with the coarse entries in index order: larger wins, equal -> smaller index
            if (v > pbest || (v == pbest && v > 0.0f && idx < ppos)) { pbest = v; ppos = idx; }
        }
        const float peak_mag = pbest;
        const int peak_pos = (pbest > 0.0f) ? ppos : 0;
        // ---- the three lags the repetition logic and the CFO estimate can ask for
        const int l3 = peak_pos + (lane - 1) * kZcRep;           // lanes 0,1,2: peak-1016, peak, peak+1016
        const bool v3 = lane < 3 && l3 >= 0 && l3 + kZcRep <= n;
        const ZcSum s3 = zc_corr_at(bb, ref, v3 ? l3 : 0);
        float cm = 0.0f;                                          // computeCorrelationMag
        {
            const float denom = fsqrt(s3.e * ref_energy);
            if (v3 && denom > 1e-10f) cm = fdiv(hypotf_glibc(s3.sr, s3.si), denom);
        }
        const float cm_e = __shfl(cm, 0), cm_p = __shfl(cm, 1), cm_l = __shfl(cm, 2);
        const float sr_e = __shfl(s3.sr, 0), si_e = __shfl(s3.si, 0), sr_p = __shfl(s3.sr, 1), si_p = __shfl(s3.si, 1);
        const float sr_l = __shfl(s3.sr, 2), si_l = __shfl(s3.si, 2);
        int timing = peak_pos;
        bool at_earlier = false;
        if (peak_mag > A.threshold && peak_pos >= kZcRep) {
            if (cm_e > peak_mag * 0.4f) { timing = peak_pos - kZcRep; at_earlier = true; }
        }
        float combined = peak_mag;
        const int rep2 = timing + kZcRep;
        if (peak_mag > 0.0f && peak_mag < 0.25f && rep2 + kZcRep <= n) {
            const float m1 = at_earlier ? cm_e : cm_p, m2 = at_earlier ? cm_p : cm_l;
            combined = fdiv(fsqrt(m1 * m1 + m2 * m2), fsqrt(2.0f));
            if (!(combined > peak_mag)) combined = peak_mag;
        }
        int has_cfo = 0; float cfo = 0.0f;
        if (rep2 + kZcRep <= n) {
            const float r1 = at_earlier ? sr_e : sr_p, i1 = at_earlier ? si_e : si_p;
            const float r2 = at_earlier ? sr_p : sr_l, i2 = at_earlier ? si_p : si_l;
            const float m1 = fdiv(hypotf_glibc(r1, i1), static_cast<float>(kZcRep)), m2 = fdiv(hypotf_glibc(r2, i2), static_cast<float>(kZcRep));
            if (m1 > 0.1f && m2 > 0.1f) {
                const float c = r1, d = -i1;
                const float pr = r2 * c - i2 * d, pi_ = r2 * d + i2 * c;   // corr2 * conj(corr1)
                const float phase_diff = atan2f_glibc(pi_, pr);
                const float rep_duration = fdiv(static_cast<float>(kZcRep), 48000.0f);
                cfo = static_cast<float>(static_cast<double>(phase_diff) / (static_cast<double>(2.0f) * 3.14159265358979323846 * static_cast<double>(rep_duration)));
                has_cfo = 1;
            }
        }
        if (lane == 0) { ro[w].combined = combined; ro[w].timing = timing; ro[w].has_cfo = has_cfo; ro[w].cfo = cfo; }
    }
    __syncthreads();
    if (tid == 0) {
        float best_corr = 0.0f, best_cfo = 0.0f;
        int best_root = -1, best_pos = -1;
        for (int r = 0; r < 4; ++r) {
            if (!(A.root_mask & (1u << r))) continue;
            if (ro[r].combined > best_corr) {
                best_corr = ro[r].combined; best_root = 2 * r + 1; best_pos = ro[r].timing;
                if (ro[r].has_cfo) best_cfo = ro[r].cfo;
            }
        }
        ria_zc_result o;
        o.detected = 0; o.frame_type = (best_root >= 0) ? (best_root - 1) / 2 : 255; o.start_sample = -1; o.root_detected = best_root;
        o.correlation = best_corr; o.cfo_hz = 0.0f; o.snr_estimate = 0.0f; o.reserved = 0.0f;
        if (best_corr > A.threshold && best_root >= 0) {
            o.detected = 1; o.cfo_hz = best_cfo; o.start_sample = best_pos + kZcPreamble;
            float snr;
            if (best_corr <= 0.01f) snr = -10.0f;
            else if (best_corr >= 0.99f) snr = 30.0f;
            else {
                snr = 20.0f * log10f_glibc(fdiv(best_corr, 1.0f - best_corr + 0.01f));
                snr = (snr < -10.0f) ? -10.0f : snr;
                snr = (snr > 30.0f) ? 30.0f : snr;
            }
            o.snr_estimate = snr;
        }
        *out = o;
    }
}


// =====================================================================================================
// Dual-chirp acquisition: sync::ChirpSync::detectDualChirp (src/sync/chirp_sync.hpp:352-512) for a batch
// of capture buffers, bit-exact.
//
// The reference correlates with a 24 000-sample complex chirp through a 131 072-point FFT
// (detectChirpTemplateFFT :627-712: FFT(signal) * conj(FFT(template)) -> IFFT -> |.| normalised by a
// sliding energy -> first maximum) for the up-chirp over the whole buffer and again for the down-chirp
// over a window placed after the up-chirp.  Its FFT is the in-tree radix-2 DIT (src/dsp/fft.cpp:96-128).
// A float result does not depend on the ORDER in which independent butterflies are evaluated, only on
// each butterfly's own arithmetic, so the 17 stages are regrouped here into four register-resident
// passes (4+4+4+5 stages; every thread runs a complete 2^G-point butterfly network on 2^G strided
// elements) with the twiddles of fft.cpp:83-87:
//   pass 1  gathers the bit-reversed input with COALESCED reads (thread t takes inputs t + m*N/16; they
//           all land in output block bitrev(t)) — real samples, zero padding and window offset folded in;
//   pass 4  of the forward transform multiplies by the stored conj(FFT(template)) on the way out;
//   pass 4  of the inverse transform scales by 1/N and writes |corr| only.
// The working set of one buffer (2 x 1 MiB complex + 0.5 MiB) is meant to stay in L2 / Infinity Cache:
// the host loops over chunks of buffers small enough for that.
// The sliding-energy normalisation uses a float running sum over the window (cumsum_energy :668-672):
// serial by definition, so it is one LANE per buffer; the peak search is one workgroup per buffer.
// Short down-chirp windows (< 48 000 samples) take the reference's time-domain path (:759-817).
constexpr int kChLen = 24000, kChGap = 4800, kChFft = 131072, kChLog = 17;

struct ChirpBufState {      // per buffer, device memory
    int active;             // 1: this stage runs the FFT path, 2: time-domain path, 0: nothing to do
    int win_start, win_len; // window of the current stage (up: whole buffer)
    int pos;                // detection of the current stage (-1: none)
    float corr;
    int up_pos; float up_corr;
};
struct ChirpArgs {
    const float* samples; long long stride; int buf_len; int n_buffers; int first;   // outer chunk = buffers [first, first+n_buffers): state + cumsum arrays
    int sub, n_sub;            // inner chunk [sub, sub+n_sub) of the outer chunk: owns the FFT workspace slots 0..n_sub-1
    float threshold;
    const float2* tw;          // [65536]
    const float2* tmpl_fft;    // [2][131072] conj(FFT(template)): up, down
    const float* tmpl;         // [4][24000] up sin, up cos, down sin, down cos
    float tmpl_energy[2];
    float2* w1; float2* w2;    // [chunk][131072]
    float* mag;                // [chunk][131072] (unused since the peak search moved into the last inverse pass)
    unsigned long long* best;  // [chunk] packed (normalised correlation bits << 32) | ~position: atomicMax = first maximum
    float* cum;                // [outer chunk][131073]
    ChirpBufState* st;         // [outer chunk]
    ria_chirp_result* out;     // [all buffers]
    int down;                  // stage: 0 up, 1 down
};

__device__ __forceinline__ void wave_lds_fence() {   // single-wave workgroup: DS ops execute in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float2 ch_cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ void ch_bfly(float2& a, float2& b, float2 w) {  // fft.cpp:113-117
    const float2 t = ch_cmul(w, b);
    b = make_float2(a.x - t.x, a.y - t.y);
    a = make_float2(a.x + t.x, a.y + t.y);
}
__device__ __forceinline__ int ch_bitrev(int v, int bits) { return static_cast<int>(__brev(static_cast<unsigned>(v)) >> (32 - bits)); }

// stage setup: up = whole buffer; down = window after the detected up-chirp (chirp_sync.hpp:430-452)
__global__ void chirp_window_kernel(ChirpArgs A) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.n_buffers) return;
    ChirpBufState& s = A.st[b];
    const int n = A.buf_len;
    if (!A.down) {
        s.up_pos = -1; s.up_corr = 0.0f; s.pos = -1; s.corr = 0.0f;
        s.win_start = 0; s.win_len = n;
        s.active = (n >= 2 * kChLen + kChGap) ? 1 : 0;   // :372-376 (the up search always takes the FFT path)
        ria_chirp_result o; o.success = 0; o.up_chirp_start = -1; o.down_chirp_start = -1; o.cfo_hz = 0.f; o.up_correlation = 0.f;
        o.down_correlation = 0.f; o.reserved[0] = 0; o.reserved[1] = 0;
        A.out[A.first + b] = o;
        return;
    }
    s.up_pos = s.pos; s.up_corr = s.corr;
    const bool had = s.active != 0;
    s.active = 0;
    if (had) A.out[A.first + b].up_correlation = s.up_corr;
    if (!had || s.up_pos < 0) return;
    const long long up = s.up_pos;
    const long long start = up + kChLen / 2, expected = up + kChLen + kChGap, min_len = 2 * kChLen + 1000;
    long long end = (expected + 10000 + kChLen > start + min_len) ? expected + 10000 + kChLen : start + min_len;
    if (end > n) end = n;
    if (start >= n) return;
    if (end <= start + kChLen) { end = start + 2 * kChLen; if (end > n) end = n; }
    s.win_start = static_cast<int>(start); s.win_len = static_cast<int>(end - start);
    s.pos = -1; s.corr = 0.0f;
    if (s.win_len < kChLen) { s.active = 0; return; }            // detectChirpTemplate :722-725 -> {-1, 0}
    s.active = (s.win_len >= 2 * kChLen) ? 1 : 2;
}

// cumsum_energy[i+1] = cumsum_energy[i] + s[i]^2 (chirp_sync.hpp:668-672).  The running float sum is serial by
// definition (no re-association), and a serial chain occupies one LANE: a wavefront therefore carries kCumB buffers at once -
// all lanes load and square a 256-sample tile of each (coalesced), lanes 0..kCumB-1 each walk their own buffer's tile through
// LDS with ONE instruction stream (the dependent adds are the whole cost, and an instruction issued for one lane costs the
// same as for eight), all lanes store the tiles (coalesced).  Rows are kCumRow floats apart so that the walkers' 16-byte
// reads fall on different banks.
constexpr int kCumB = 8, kCumRow = 256 + 8;
__global__ __launch_bounds__(64) void chirp_cumsum_kernel(ChirpArgs A) {
    __shared__ __attribute__((aligned(16))) float sq[kCumB * kCumRow];
    __shared__ __attribute__((aligned(16))) float cs[kCumB * kCumRow];
    const int lane = threadIdx.x, b0 = blockIdx.x * kCumB;
    int fft_in[kCumB], len_max = 0;
    const float* x[kCumB];
    float* cum[kCumB];
#pragma unroll
    for (int q = 0; q < kCumB; ++q) {
        const int b = b0 + q;
        fft_in[q] = 0; x[q] = nullptr; cum[q] = nullptr;
        if (b < A.n_buffers) {
            const ChirpBufState s = A.st[b];
            if (s.active == 1) {
                fft_in[q] = s.win_len < kChFft ? s.win_len : kChFft;
                x[q] = A.samples + static_cast<long long>(A.first + b) * A.stride + s.win_start;
                cum[q] = A.cum + static_cast<size_t>(b) * (kChFft + 1);
                if (lane == 0) cum[q][0] = 0.0f;
            }
        }
        len_max = fft_in[q] > len_max ? fft_in[q] : len_max;
    }
    if (len_max == 0) return;
    float c = 0.0f;                                           // lane q < kCumB: running sum of buffer b0 + q
    int walk_len = 0;
#pragma unroll
    for (int q = 0; q < kCumB; ++q) if (lane == q) walk_len = fft_in[q];
    float nx[kCumB][4];   // next tile of every buffer, loaded one iteration ahead so that the HBM latency hides behind the serial walk
#pragma unroll
    for (int q = 0; q < kCumB; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int i = 64 * r + lane; nx[q][r] = (i < fft_in[q]) ? x[q][i] : 0.0f; }
    for (int base = 0; base < len_max; base += 256) {
#pragma unroll
        for (int q = 0; q < kCumB; ++q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = nx[q][r];
                sq[q * kCumRow + 64 * r + lane] = v * v;
                const int i = base + 256 + 64 * r + lane;
                nx[q][r] = (i < fft_in[q]) ? x[q][i] : 0.0f;
            }
        }
        wave_lds_fence();
        if (lane < kCumB && base < walk_len) {
            // register blocks of 32 samples: the loads of a block do not wait for the stores of the previous one (separate
            // input / output arrays), so only the dependent adds are on the critical path
            const float4* in = reinterpret_cast<const float4*>(sq + lane * kCumRow);
            float4* outp = reinterpret_cast<float4*>(cs + lane * kCumRow);
#pragma unroll 2
            for (int blk = 0; blk < 8; ++blk) {
                float4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = in[8 * blk + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    c = c + v[q].x; v[q].x = c;
                    c = c + v[q].y; v[q].y = c;
                    c = c + v[q].z; v[q].z = c;
                    c = c + v[q].w; v[q].w = c;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) outp[8 * blk + q] = v[q];
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < kCumB; ++q) {
            if (base < fft_in[q]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = base + 64 * r + lane;
                    if (i < fft_in[q]) cum[q][i + 1] = cs[q * kCumRow + 64 * r + lane];
                }
            }
        }
        wave_lds_fence();
    }
}

// One pass of the 131072-point radix-2 DIT FFT: stages S0+1 .. S0+G on 2^G elements per thread.
// MODE 0: in place on w.   MODE 1: first pass, real input window (zero padded) -> dst.
// MODE 2: first pass, complex src -> dst.   MODE 3: in place + multiply by tmpl on the way out (last forward).
// MODE 4: last inverse pass: scale 1/N, write |.| to mag.
template <int G, int S0, int MODE, bool INV>
__global__ __launch_bounds__(256) void chirp_fft_pass(ChirpArgs A, const float2* __restrict__ src_all, float2* __restrict__ dst_all) {
    constexpr int R = 1 << G;
    const int b = blockIdx.y;                       // workspace slot; buffer A.first + A.sub + b
    const ChirpBufState s = A.st[A.sub + b];
    if (s.active != 1) return;
    const int t = blockIdx.x * 256 + threadIdx.x;    // < N / R
    float2* dst = dst_all + static_cast<size_t>(b) * kChFft;
    float2 x[R];
    int idx0, stridej;     // element j lives at idx0 + j*stridej in the DIT-ordered array
    int lo = 0;
    if constexpr (S0 == 0) {
        constexpr int TB = kChLog - G;
        const int hi = ch_bitrev(t, TB);
        idx0 = hi * R; stridej = 1;
        if constexpr (MODE == 1) {
            const float* in = A.samples + static_cast<long long>(A.first + A.sub + b) * A.stride + s.win_start;
            const int fft_in = s.win_len < kChFft ? s.win_len : kChFft;
#pragma unroll
            for (int m = 0; m < R; ++m) {
                const int n_in = m * (kChFft / R) + t;
                x[ch_bitrev(m, G)] = make_float2(n_in < fft_in ? in[n_in] : 0.0f, 0.0f);
            }
        } else {
            const float2* in = src_all + static_cast<size_t>(b) * kChFft;
#pragma unroll
            for (int m = 0; m < R; ++m) x[ch_bitrev(m, G)] = in[m * (kChFft / R) + t];
        }
    } else {
        lo = t & ((1 << S0) - 1);
        const int hi = t >> S0;
        idx0 = lo + (hi << (S0 + G)); stridej = 1 << S0;
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = dst[idx0 + j * stridej];
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const int half = 1 << u, sidx = S0 + u + 1;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if ((j & half) == 0) {
                const int k = lo + ((j & (half - 1)) << S0);
                float2 w = A.tw[k << (kChLog - sidx)];
                if (INV) w.y = -w.y;
                ch_bfly(x[j], x[j + half], w);
            }
        }
    }
    if constexpr (MODE == 3) {
        const float2* tf = A.tmpl_fft + static_cast<size_t>(A.down) * kChFft;
#pragma unroll
        for (int j = 0; j < R; ++j) { const int i = idx0 + j * stridej; dst[i] = ch_cmul(x[j], tf[i]); }
    } else if constexpr (MODE == 4) {
        // scale by 1/N, |.|, normalise by the sliding energy and keep the FIRST maximum over pos < search_len
        // (chirp_sync.hpp:677-693): larger value wins, equal values -> smaller position, which is exactly an
        // unsigned max of (value bits, ~position) because the values are non-negative floats
        const float* cum = A.cum + static_cast<size_t>(A.sub + b) * (kChFft + 1);
        const int fft_in = s.win_len < kChFft ? s.win_len : kChFft;
        const int search_len = fft_in - kChLen;
        const float te = A.tmpl_energy[A.down];
        const float scale = 1.0f / static_cast<float>(kChFft);
        unsigned long long key = 0ull;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int pos = idx0 + j * stridej;
            if (pos < search_len) {
                const float mag = hypotf_glibc(x[j].x * scale, x[j].y * scale);
                const float se = cum[pos + kChLen] - cum[pos];
                const float denom = fsqrt(se * te);
                const float nc = (denom > 1e-10f) ? fdiv(mag, denom) : 0.0f;
                if (nc > 0.0f) {
                    const unsigned long long k = (static_cast<unsigned long long>(f2u(nc)) << 32) | (0xFFFFFFFFu - static_cast<unsigned>(pos));
                    key = k > key ? k : key;
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off);
            key = o > key ? o : key;
        }
        if ((threadIdx.x & 63) == 0 && key) atomicMax(A.best + b, key);
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) dst[idx0 + j * stridej] = x[j];
    }
}

// unpack the first maximum found by the last inverse pass, apply the threshold (:707-711)
__global__ void chirp_peak_kernel(ChirpArgs A) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.n_sub) return;
    ChirpBufState& s = A.st[A.sub + b];
    if (s.active != 1) return;
    const unsigned long long key = A.best[b];
    const float corr = key ? u2f(static_cast<uint32_t>(key >> 32)) : 0.0f;
    const int pos = key ? static_cast<int>(0xFFFFFFFFu - static_cast<uint32_t>(key & 0xFFFFFFFFull)) : -1;
    s.corr = corr;
    s.pos = (pos >= 0 && !(corr < A.threshold)) ? pos : -1;
}

// time-domain path of detectChirpTemplate (:759-817) for short windows: one workgroup per buffer,
// one lane per candidate position (every correlation is a left-to-right sum over 24 000 samples)
__device__ inline float chirp_td_corr(const float* x, int n, int offset, const float* tsin, const float* tcos, float te) {  // :829-851
    if (offset < 0 || offset + kChLen > n) return 0.0f;
    float ci = 0.0f, cq = 0.0f, e = 0.0f;
    const float* p = x + offset;
#pragma unroll 8
    for (int i = 0; i < kChLen; ++i) {
        const float v = p[i];
        ci += v * tcos[i];
        cq += v * tsin[i];
        e += v * v;
    }
    const float denom = fsqrt(e * te);
    if (denom < 1e-10f) return 0.0f;
    return fdiv(fsqrt(ci * ci + cq * cq), denom);
}
__global__ __launch_bounds__(256) void chirp_td_kernel(ChirpArgs A) {
    __shared__ float sv[4]; __shared__ int si[4]; __shared__ int spos;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    ChirpBufState& s = A.st[b];
    if (s.active != 2) return;
    const float* x = A.samples + static_cast<long long>(A.first + b) * A.stride + s.win_start;
    const int n = s.win_len, search_len = n - kChLen;
    const float* tsin = A.tmpl + static_cast<size_t>(2 * A.down) * kChLen;
    const float* tcos = tsin + kChLen;
    const float te = A.tmpl_energy[A.down];
    // block-wide "first maximum" of (value, index) pairs; -1 marks lanes without a candidate
    auto block_argmax = [&](float v, int idx, float& ov, int& oi) {
        wave_argmax_first(v, idx);
        __syncthreads();
        if (lane == 0) { sv[wv] = v; si[wv] = idx; }
        __syncthreads();
        ov = sv[0]; oi = si[0];
        for (int w = 1; w < 4; ++w) if (sv[w] > ov || (sv[w] == ov && si[w] < oi)) { ov = sv[w]; oi = si[w]; }
    };
    float best = 0.0f; int best_pos = -1;
    const int n_coarse = (search_len + 47) / 48;
    for (int base = 0; base < n_coarse; base += 256) {       // coarse grid, step 48 (:763-773)
        const int k = base + tid;
        const bool valid = k < n_coarse;
        const float c = chirp_td_corr(x, n, valid ? k * 48 : 0, tsin, tcos, te);
        float v; int idx;
        block_argmax(valid ? c : -1.0f, k * 48, v, idx);
        if (v > best) { best = v; best_pos = idx; }
    }
    int pos_out = -1;
    if (!(best_pos < 0 || best < A.threshold * 0.3f)) {
        const int fine_start = best_pos - 48 < 0 ? 0 : best_pos - 48;
        const int fine_end = best_pos + 48 > search_len ? search_len : best_pos + 48;
        {   // fine search, inclusive range of at most 97 positions (:788-795)
            const int p = fine_start + tid;
            const bool valid = p <= fine_end;
            const float c = (tid < 128) ? chirp_td_corr(x, n, valid ? p : 0, tsin, tcos, te) : 0.0f;
            float v; int idx;
            block_argmax(valid ? c : -1.0f, p, v, idx);
            if (v > best) { best = v; best_pos = idx; }
        }
        if (best_pos > 0 && best_pos < search_len - 1) {     // parabolic interpolation (:798-809)
            float cc = 0.0f;
            if (tid < 2) cc = chirp_td_corr(x, n, best_pos + (tid == 0 ? -1 : 1), tsin, tcos, te);
            const float c0 = __shfl(cc, 0), c2 = __shfl(cc, 1);
            if (tid == 0) {
                const float c1 = best;
                const float denom = 2.0f * (c0 - 2.0f * c1 + c2);
                int bp = best_pos;
                if (fabs_(denom) > 1e-10f) {
                    float delta = fdiv(c0 - c2, denom);
                    const float lo = (1.0f < delta) ? 1.0f : delta;
                    delta = (-1.0f < lo) ? lo : -1.0f;
                    bp = static_cast<int>(__builtin_roundf(static_cast<float>(best_pos) + delta));
                }
                spos = bp;
            }
            __syncthreads();
            best_pos = spos;
        }
        pos_out = (best >= A.threshold) ? best_pos : -1;
    }
    if (tid == 0) { s.corr = best; s.pos = pos_out; }
}

// CFO and position correction from the two detections (:454-509)
__global__ void chirp_finish_kernel(ChirpArgs A) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.n_buffers) return;
    const ChirpBufState s = A.st[b];
    if (s.active == 0 || s.pos < 0) return;
    ria_chirp_result o = A.out[A.first + b];
    const int up_pos = s.up_pos, down_pos = s.pos + s.win_start;
    o.down_correlation = s.corr;
    const float T = fdiv(500.0f, 1000.0f);
    const float chirp_rate = fdiv(2700.0f - 300.0f, T);
    const float cfo_to_samples = fdiv(48000.0f, chirp_rate);
    const int expected_gap = kChLen + kChGap, actual_gap = down_pos - up_pos;
    const float gap_error = static_cast<float>(actual_gap - expected_gap);
    const float cfo = fdiv(gap_error, 2.0f * cfo_to_samples);
    o.cfo_hz = cfo;
    if (!(fabs_(cfo) > 100.0f)) {
        const float up_corr = cfo * cfo_to_samples, down_corr = -cfo * cfo_to_samples;
        o.up_chirp_start = static_cast<int>(__builtin_roundf(static_cast<float>(up_pos) + up_corr));
        o.down_chirp_start = static_cast<int>(__builtin_roundf(static_cast<float>(down_pos) + down_corr));
        o.success = 1;
    }
    A.out[A.first + b] = o;
}

// =====================================================================================================
// LTS light sync: OFDMChirpWaveform::detectDataSync (src/waveform/ofdm_chirp_waveform.cpp:207-384), bit-exact.
// One workgroup per capture buffer: energy gate (first 64-sample window above 3x the noise floor),
// Hilbert-65 analytic signal of the search span into LDS, one LANE per candidate offset for the
// one-symbol-delay autocorrelation (a 1152-term left-to-right complex sum), then the reference's sequential
// rules replayed over the per-offset results: stop at the first correlation above 0.95, first maximum up
// to there, +-4 refinement, burst-interleave marker from the sign of the CFO-compensated peak.
constexpr int kLtsSym = 1152, kLtsTaps = 65, kLtsMaxSpan = 10 * kLtsSym, kLtsMaxOffsets = kLtsSym + 8;

struct LtsArgs {
    const float* samples; long long stride; int buf_len; int n_buffers;
    const float* known_cfo; float threshold;
    const float* hilbert;       // [65]
    ria_lts_result* out;
};

// The analytic signal sits in LDS with one empty slot after every 8 samples (sample i at slot i + (i >> 3)): the coarse
// search gives lane l the offset 8 l, so without the padding the 32 lanes of a half-wave read slots 64 bytes apart - four
// bank groups, an 8-way conflict on every one of the 2 x 1152 reads of a correlation; padded they are 72 bytes apart and
// cover all 64 banks.
__host__ __device__ inline int lts_slot(int i) { return i + (i >> 3); }
__host__ __device__ inline int lts_lds_bytes() { return (lts_slot(kLtsMaxSpan) + 8) * 8 + kLtsMaxOffsets * 12 + 64; }

struct LtsCorr { float corr, pr, pi; };
__device__ inline LtsCorr lts_corr_at(const float2* __restrict__ an, int rel, int offset, int n) {
    float pr = 0.0f, pi = 0.0f, e1 = 0.0f, e2 = 0.0f;
    int kmax = n - kLtsSym - offset;                 // the reference leaves the loop at the first k with offset + k + 1152 >= n
    kmax = kmax < 0 ? 0 : (kmax > kLtsSym ? kLtsSym : kmax);
#pragma unroll 4
    for (int k = 0; k < kmax; ++k) {
        const float2 s1 = an[lts_slot(rel + k)], s2 = an[lts_slot(rel + k + kLtsSym)];
        const float a = s1.x, b = -s1.y, c = s2.x, d = s2.y;     // conj(s1) * s2
        pr += a * c - b * d;
        pi += a * d + b * c;
        e1 += s1.x * s1.x + s1.y * s1.y;
        e2 += c * c + d * d;
    }
    const float den = fsqrt(e1 * e2) + 1e-10f;
    return {fdiv(hypotf_glibc(pr, pi), den), pr, pi};
}

// kLtsThreads: the noise-only search (4 symbols) has 576 candidate offsets, the connected one (8 symbols) 1152: 640 lanes take
// them in one / two rounds, and ten waves per CU (the span's 118 KB of LDS allow one workgroup) hide the LDS latency of
// the 2 x 1152 dependent reads per candidate that four waves could not
constexpr int kLtsThreads = 640;
__global__ __launch_bounds__(kLtsThreads) void lts_sync_kernel(LtsArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* an = reinterpret_cast<float2*>(smem);                         // analytic signal of [base, base + span)
    float* ccorr = reinterpret_cast<float*>(an + lts_slot(kLtsMaxSpan) + 8);   // per coarse offset
    float* cpr = ccorr + kLtsMaxOffsets;
    float* cpi = cpr + kLtsMaxOffsets;
    __shared__ float sh_noise; __shared__ int sh_start;
    const int tid = threadIdx.x, n = A.buf_len, L = kLtsSym;
    const float* x = A.samples + static_cast<long long>(blockIdx.x) * A.stride;
    ria_lts_result* out = A.out + blockIdx.x;
    const float known = A.known_cfo ? A.known_cfo[blockIdx.x] : 0.0f;
    if (n < 3 * L) {
        if (tid == 0) { out->detected = 0; out->start_sample = 0; out->correlation = 0.f; out->cfo_hz = known; out->burst_interleaved = 0; out->reserved[0] = out->reserved[1] = out->reserved[2] = 0; }
        return;
    }
    if (tid == 0) {   // noise floor over the first min(n/4, 4800) samples (:232-237)
        const int ns = n / 4 < 4800 ? n / 4 : 4800;
        float acc = 0.0f;
        for (int i = 0; i < ns; ++i) acc += x[i] * x[i];
        sh_noise = fsqrt(fdiv(acc, static_cast<float>(ns)));
        sh_start = 0x7fffffff;
    }
    __syncthreads();
    const float noise_floor = sh_noise;
    const float energy_threshold = noise_floor * 3.0f + 0.01f;
    const bool in_noise = noise_floor < 0.05f;
    int signal_start = 0;
    if (in_noise) {   // first window of 64 samples whose rms exceeds the threshold (:243-258)
        const int lim = n - 2 * L;
        for (int base = 0; base < lim; base += kLtsThreads) {
            const int i = base + tid;
            bool hit = false;
            if (i < lim) {
                float e = 0.0f;
                for (int j = 0; j < 64; ++j) if (i + j < n) e += x[i + j] * x[i + j];
                hit = fsqrt(fdiv(e, 64.0f)) > energy_threshold;
            }
            if (hit) atomicMin(&sh_start, i);
            __syncthreads();
            const int cur = sh_start;
            __syncthreads();
            if (cur != 0x7fffffff) break;
        }
        __syncthreads();
        signal_start = (sh_start == 0x7fffffff) ? 0 : sh_start;
    }
    const int search_window = 4 * L, max_connected = 8 * L;
    const int actual = in_noise ? search_window : max_connected;
    const int search_end = (signal_start + actual < n - 2 * L) ? signal_start + actual : n - 2 * L;
    // analytic signal (HilbertTransform(65).process, filters.cpp:293-317) of the span the search touches
    const int base = signal_start;
    int span = search_end + 2 * L - base;
    if (base + span > n) span = n - base;
    if (span < 0) span = 0;
    for (int r = tid; r < span; r += kLtsThreads) {
        const int i = base + r;
        float q = 0.0f;
        for (int k = 0; k < kLtsTaps; ++k) q += A.hilbert[k] * ((i - k >= 0) ? x[i - k] : 0.0f);
        an[lts_slot(r)] = make_float2((i - 32 >= 0) ? x[i - 32] : 0.0f, q);
    }
    __syncthreads();
    // coarse grid, step 8 (:279-316)
    const int K = (search_end > signal_start) ? (search_end - signal_start + 7) / 8 : 0;
    for (int k = tid; k < K; k += kLtsThreads) {
        const LtsCorr c = lts_corr_at(an, 8 * k, signal_start + 8 * k, n);
        ccorr[k] = c.corr; cpr[k] = c.pr; cpi[k] = c.pi;
    }
    __syncthreads();
    __shared__ float sh_best; __shared__ int sh_off; __shared__ float sh_pr, sh_pi;
    if (tid == 0) {
        float best = 0.0f, bpr = 0.0f, bpi = 0.0f; int boff = 0;
        for (int k = 0; k < K; ++k) {
            if (ccorr[k] > best) { best = ccorr[k]; boff = signal_start + 8 * k; bpr = cpr[k]; bpi = cpi[k]; }
            if (ccorr[k] > 0.95f) break;
        }
        sh_best = best; sh_off = boff; sh_pr = bpr; sh_pi = bpi;
    }
    __syncthreads();
    float best = sh_best; int boff = sh_off;
    // +-4 refinement (:322-353)
    if (best > A.threshold) {
        const int rs = (signal_start > boff - 4) ? signal_start : boff - 4;
        const int re = (search_end < boff + 5) ? search_end : boff + 5;
        if (tid < 9) {
            const int off = rs + tid;
            if (off < re && off != boff) {
                const LtsCorr c = lts_corr_at(an, off - base, off, n);
                ccorr[tid] = c.corr; cpr[tid] = c.pr; cpi[tid] = c.pi;
            } else ccorr[tid] = -1.0f;
        }
        __syncthreads();
        if (tid == 0) {
            float b = best, bpr = sh_pr, bpi = sh_pi; int bo = boff;
            for (int q = 0; q < 9; ++q) if (ccorr[q] > b) { b = ccorr[q]; bo = rs + q; bpr = cpr[q]; bpi = cpi[q]; }
            sh_best = b; sh_off = bo; sh_pr = bpr; sh_pi = bpi;
        }
        __syncthreads();
        best = sh_best; boff = sh_off;
    }
    if (tid == 0) {
        ria_lts_result o;
        o.detected = 0; o.start_sample = 0; o.correlation = best; o.cfo_hz = known; o.burst_interleaved = 0;
        o.reserved[0] = o.reserved[1] = o.reserved[2] = 0;
        if (best > A.threshold) {
            o.detected = 1; o.start_sample = boff;
            const float cfo_phase = static_cast<float>(static_cast<double>(2.0f) * 3.14159265358979323846 * static_cast<double>(known) * static_cast<double>(L) / static_cast<double>(48000.0f));
            const float cr = cosf_glibc(-cfo_phase), ci = sinf_glibc(-cfo_phase);
            o.burst_interleaved = (sh_pr * cr - sh_pi * ci < 0.0f) ? 1 : 0;
        }
        *out = o;
    }
}

}  // namespace ria
