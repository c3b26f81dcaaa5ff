// Schmidl-Cox acquisition of the OFDM-COX waveform on gfx950, bit-exact.
//
// Reference: OFDMDemodulator::searchForSync (src/ofdm/demodulator.cpp:1450-1542) with
//   Impl::hasMinimumEnergy             src/ofdm/ofdm_sync.cpp:20-50
//   Impl::toAnalytic                   src/ofdm/ofdm_sync.cpp:56-86
//   Impl::measureSchmidlCoxCorrelation src/ofdm/ofdm_sync.cpp:118-163
//   Impl::estimateCoarseCFO            src/ofdm/ofdm_sync.cpp:230-261
//   Impl::refineLTSTiming              src/ofdm/ofdm_sync.cpp:386-484
//
// The reference walks the buffer sequentially (64-sample grid, early exit, a noise-floor tracker that is
// updated only at the offsets it visits) and pays an FFT-1024 pair per offset it looks at.  Here the
// data-independent part is computed for EVERY offset the walk could touch, in parallel, and the walk itself is
// replayed over the tables by one lane:
//   cox_prepare_kernel  one LANE per 8-sample offset: the 1024-term left-to-right DC sum, and on the 64-sample
//                       grid the 144-term energy of hasMinimumEnergy.
//   cox_metric_kernel   (two passes: the 64-sample grid first, then only the plateau offsets around grid points above the
//                       threshold) one WAVE per 8-sample offset: DC removal, forward FFT-1024 (the reference's radix-2 DIT
//                       order, as in the demodulator kernel), Hilbert mask, inverse FFT, the 512 conjugate
//                       products by all lanes and the four 512-term ordered sums (P.re, P.im, R1, R2) by four
//                       lanes, normalised metric.
//   cox_scan_kernel     one WORKGROUP (16 waves) per buffer: lane 0 replays the search over the tables; every
//                       candidate that passes the plateau rule gets its 4033 passband LTS correlations computed
//                       one lane per offset (three 1152-term ordered sums each), first-maximum, earlier-LTS
//                       preference, confirmation threshold; on success wave 0 computes the coarse CFO.
#pragma once

namespace ria {

constexpr int kCoxL = 1152, kCoxCp = 128, kCoxTotal = 6 * kCoxL, kCoxWindow = 2 * kCoxL;
constexpr int kCoxBack = 3 * kCoxL, kCoxFwd = kCoxL / 2, kCoxLtsOffsets = kCoxBack + kCoxFwd + 1;   // 4033
constexpr int kCoxMinSearch = 4000, kCoxMaxBuf = 240000;   // demodulator_constants.hpp:45-46

struct CoxArgs {
    const float* samples; long long stride; int buf_len; int n_buffers;
    float threshold; const float* noise_in;
    const float2* twiddle;      // [512]
    const float* tI; const float* tQ; float energy_ref;   // LTS passband templates [1152]
    float* dc; float* energy; float* metric;   // [n_buffers][nM], [n_buffers][nE], [n_buffers][nM]
    int nM, nE;
    ria_cox_result* out;
};

// metric offsets o = 8k with o + 6*1152 < n (plateau bound, demodulator.cpp:1499); search grid i = 64k < n - 8*1152
__host__ __device__ inline int cox_n_metric(int n) { return n > kCoxTotal ? (n - kCoxTotal - 1) / 8 + 1 : 0; }
__host__ __device__ inline int cox_n_energy(int n) { return n > kCoxTotal + kCoxWindow ? (n - kCoxTotal - kCoxWindow + 63) / 64 : 0; }

__global__ __launch_bounds__(256) void cox_prepare_kernel(CoxArgs A) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= A.nM) return;
    const int b = blockIdx.y;
    const float* x = A.samples + static_cast<long long>(b) * A.stride;
    const int o = 8 * k;
    {   // measureSchmidlCoxCorrelation: dc_sum over the FFT part (ofdm_sync.cpp:131-135)
        const float* d = x + o + kCoxCp;
        float acc = 0.0f;
#pragma unroll 8
        for (int i = 0; i < 1024; ++i) acc += d[i];
        A.dc[static_cast<long long>(b) * A.nM + k] = fdiv(acc, 1024.0f);
    }
    if ((k & 7) == 0 && (k >> 3) < A.nE) {   // hasMinimumEnergy: every 16th sample of two symbols (ofdm_sync.cpp:23-33)
        float acc = 0.0f;
        for (int i = 0; i < kCoxWindow; i += 16) { const float s = x[o + i]; acc += s * s; }
        A.energy[static_cast<long long>(b) * A.nE + (k >> 3)] = fdiv(acc, static_cast<float>(kCoxWindow / 16));
    }
}

// forward FFT-1024 by one wavefront, all bins (fft.cpp:96-128).  buf: plain natural-order input; result in
// registers, y[4*t + q] = bin (lane + 64 t) + 256 q.
__device__ inline void fft1024_full_wave(float2* buf, const float2* __restrict__ tw, int lane, float2 (&y)[16]) {
    float2 x[16];
    {
        const int rl = __brev(static_cast<unsigned>(lane)) >> 26;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int r4 = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
            x[r] = buf[rl + 64 * r4];
        }
    }
    wave_sync();
#pragma unroll
    for (int s = 1; s <= 4; ++s) {
        const int half = 1 << (s - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if ((r & half) == 0) bfly(x[r], x[r + half], tw[(r & (half - 1)) << (10 - s)]);
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
                if ((r & hr) == 0) bfly(x[r], x[r + hr], tw[(a + 16 * (r & (hr - 1))) << (10 - s)]);
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[base + 17 * r] = x[r];
    }
    wave_sync();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int b = lane + 64 * t, pa = b + (b >> 4);
        float2 y0 = buf[pa], y1 = buf[pa + 272], y2 = buf[pa + 544], y3 = buf[pa + 816];
        const float2 w9 = tw[b << 1];
        bfly(y0, y1, w9); bfly(y2, y3, w9);
        bfly(y0, y2, tw[b]); bfly(y1, y3, tw[b + 256]);
        y[4 * t] = y0; y[4 * t + 1] = y1; y[4 * t + 2] = y2; y[4 * t + 3] = y3;
    }
    wave_sync();
}

// toAnalytic for len == fft_len == 1024 (ofdm_sync.cpp:56-86): buf holds the real samples (plain layout, zero
// imaginary parts) on entry and the analytic signal at buf[i + (i >> 4)] on exit.
__device__ inline void cox_analytic_wave(float2* buf, const float2* __restrict__ tw, int lane) {
    float2 y[16];
    fft1024_full_wave(buf, tw, lane, y);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int b = lane + 64 * t;
        float2 v0 = y[4 * t], v1 = y[4 * t + 1], v2 = y[4 * t + 2];
        if (b != 0) { v0 = make_float2(v0.x * 2.0f, v0.y * 2.0f); v2 = make_float2(0.0f, 0.0f); }   // bin 0 and bin 512 kept
        v1 = make_float2(v1.x * 2.0f, v1.y * 2.0f);
        buf[b] = v0; buf[b + 256] = v1; buf[b + 512] = v2; buf[b + 768] = make_float2(0.0f, 0.0f);
    }
    wave_sync();
    ifft1024_wave(buf, tw, lane);
}

__device__ __forceinline__ float cox_chain512(const float* f) {   // left-to-right sum of 512 floats
    const float4* q = reinterpret_cast<const float4*>(f);
    float acc = 0.0f;
#pragma unroll 4
    for (int k = 0; k < 128; ++k) { const float4 v = q[k]; acc += v.x; acc += v.y; acc += v.z; acc += v.w; }
    return acc;
}

// PHASE 0: the offsets of the 64-sample search grid (every 8th table entry).  PHASE 1: the other offsets, but only
// those a plateau scan can read: o in [i, i + 300] for a grid point i whose metric exceeds the threshold
// (demodulator.cpp:1494-1509) - a superset of what the sequential walk touches, since the energy gate is ignored.
template <int PHASE>
__global__ __launch_bounds__(256) void cox_metric_kernel(CoxArgs A) {
    __shared__ __attribute__((aligned(16))) float2 smem[4 * kFftBufFloats2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kk = blockIdx.x * 4 + wave;
    const int k = PHASE == 0 ? 8 * kk : kk;
    if (k >= A.nM) return;
    if (PHASE == 1) {
        if ((k & 7) == 0) return;
        const float* Mg = A.metric + static_cast<long long>(blockIdx.y) * A.nM;
        const int o = 8 * k, search_end = A.buf_len - kCoxTotal - kCoxWindow;
        bool needed = false;
        for (int i = (o >> 6) << 6; i >= 0 && o - i <= 300; i -= 64)
            if (i < search_end && Mg[i >> 3] > A.threshold) needed = true;
        if (!needed) return;
    }
    const int b = blockIdx.y;
    float2* buf = smem + wave * kFftBufFloats2;
    const float* d = A.samples + static_cast<long long>(b) * A.stride + 8 * k + kCoxCp;
    const float dc = A.dc[static_cast<long long>(b) * A.nM + k];
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[lane + 64 * r] = make_float2(d[lane + 64 * r] - dc, 0.0f);
    wave_sync();
    cox_analytic_wave(buf, A.twiddle, lane);
    float pr[8], pi[8], n1[8], n2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {   // conj(a[i]) * a[i + 512], |a[i]|^2, |a[i + 512]|^2 (ofdm_sync.cpp:147-151)
        const int i = lane + 64 * t, j = i + 512;
        const float2 s1 = buf[i + (i >> 4)], s2 = buf[j + (j >> 4)];
        const float a = s1.x, bb = -s1.y, c = s2.x, dd = s2.y;
        pr[t] = a * c - bb * dd;
        pi[t] = a * dd + bb * c;
        n1[t] = s1.x * s1.x + s1.y * s1.y;
        n2[t] = c * c + dd * dd;
    }
    wave_sync();
    float* f = reinterpret_cast<float*>(buf);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int i = lane + 64 * t;
        f[i] = pr[t]; f[512 + i] = pi[t]; f[1024 + i] = n1[t]; f[1536 + i] = n2[t];
    }
    wave_sync();
    float acc = 0.0f;
    if (lane < 4) acc = cox_chain512(f + 512 * lane);
    const float Pr = __shfl(acc, 0), Pi = __shfl(acc, 1), R1 = __shfl(acc, 2), R2 = __shfl(acc, 3);
    if (lane == 0) {
        const float norm = fsqrt(R1 * R2);
        A.metric[static_cast<long long>(b) * A.nM + k] = (norm < 1e-10f) ? 0.0f : fdiv(hypotf_glibc(Pr, Pi), norm);
    }
}

__global__ __launch_bounds__(1024) void cox_scan_kernel(CoxArgs A) {
    __shared__ float s_tI[kCoxL], s_tQ[kCoxL];
    __shared__ float s_corr[4096];
    __shared__ __attribute__((aligned(16))) float2 s_buf[kFftBufFloats2];
    __shared__ float s_wv[16];
    __shared__ int s_wi[16];
    __shared__ int sh_state, sh_peak, sh_i, sh_ok, sh_lts;
    __shared__ float sh_nf;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, n = A.buf_len;
    const float* x = A.samples + static_cast<long long>(b) * A.stride;
    ria_cox_result* out = A.out + b;
    const float nf0 = A.noise_in ? A.noise_in[b] : 0.0f;
    if (n < kCoxMinSearch || n < kCoxTotal + kCoxWindow) {   // demodulator.cpp:1454, :1466
        if (tid == 0) { ria_cox_result o{}; o.noise_floor = nf0; *out = o; }
        return;
    }
    for (int i = tid; i < kCoxL; i += 1024) { s_tI[i] = A.tI[i]; s_tQ[i] = A.tQ[i]; }
    if (tid == 0) { sh_i = 0; sh_nf = nf0; }
    __syncthreads();
    const float* E = A.energy + static_cast<long long>(b) * A.nE;
    const float* M = A.metric + static_cast<long long>(b) * A.nM;
    const int search_end = n - kCoxTotal - kCoxWindow;
    for (;;) {
        if (tid == 0) {   // the sequential walk (demodulator.cpp:1486-1527) over the precomputed tables
            int i = sh_i, state = 0, peak_pos = 0;
            float nf = sh_nf;
            for (; i < search_end; i += 64) {
                const float energy = E[i >> 6];
                if (nf < 1e-20f) nf = energy * 0.1f;
                if (energy < nf) nf = energy;
                else if (energy < nf * 3.0f) nf = (1.0f - 0.01f) * nf + 0.01f * energy;
                if (!(energy >= nf * 4.0f)) { i += kCoxWindow / 2 - 64; continue; }
                const float corr = M[i >> 3];
                if (corr > A.threshold) {
                    int plateau = 0; float peak = corr; peak_pos = i;
                    for (int j = 0; j <= 300 && i + j + kCoxTotal < n; j += 8) {
                        const float c = M[(i + j) >> 3];
                        if (c >= 0.90f) ++plateau;
                        if (c > peak) { peak = c; peak_pos = i + j; }
                    }
                    if (plateau >= 15) { state = 1; i += 64; break; }
                }
            }
            sh_state = state; sh_peak = peak_pos; sh_i = i; sh_nf = nf;
        }
        __syncthreads();
        if (sh_state == 0) break;
        // ---- refineLTSTiming(peak) (ofdm_sync.cpp:386-484)
        const int peak = sh_peak;
        const int coarse = peak + 4 * kCoxL;
        if (coarse + kCoxFwd + kCoxL > n) {                 // not enough data: the coarse position is accepted
            if (tid == 0) { sh_ok = 1; sh_lts = coarse; }
        } else {
            const int base = coarse - kCoxBack;
            for (int idx = tid; idx < kCoxLtsOffsets; idx += 1024) {
                const float* xp = x + base + idx;
                float cI = 0.0f, cQ = 0.0f, e = 0.0f;
#pragma unroll 4
                for (int i = 0; i < kCoxL; ++i) {
                    const float s = xp[i];
                    cI += s * s_tI[i];
                    cQ += s * s_tQ[i];
                    e += s * s;
                }
                const float mag = fsqrt(cI * cI + cQ * cQ), norm = fsqrt(e * A.energy_ref);
                s_corr[idx] = (norm > 1e-6f) ? fdiv(mag, norm) : 0.0f;
            }
            __syncthreads();
            float v = -1.0f; int vi = 0x7fffffff;
            for (int idx = tid; idx < kCoxLtsOffsets; idx += 1024) { const float c = s_corr[idx]; if (c > v) { v = c; vi = idx; } }
            wave_argmax_first(v, vi);
            if (lane == 0) { s_wv[wave] = v; s_wi[wave] = vi; }
            __syncthreads();
            if (tid == 0) {
                float bv = s_wv[0]; int bi = s_wi[0];
                for (int w = 1; w < 16; ++w) if (s_wv[w] > bv || (s_wv[w] == bv && s_wi[w] < bi)) { bv = s_wv[w]; bi = s_wi[w]; }
                float best = 0.0f; int best_off = coarse;        // strict '>' from 0: first maximum, if positive
                if (bv > 0.0f) { best = bv; best_off = base + bi; }
                if (best_off >= kCoxL) {                         // prefer the earlier LTS copy when close (:455-467)
                    const int prev = best_off - kCoxL;
                    if (prev >= base) {
                        const float pc = s_corr[prev - base];
                        if (pc >= best * 0.92f) { best_off = prev; best = pc; }
                    }
                }
                sh_ok = (best < 0.05f) ? 0 : 1;
                sh_lts = best_off;
            }
        }
        __syncthreads();
        if (sh_ok) {
            if (wave == 0) {   // estimateCoarseCFO(peak) (ofdm_sync.cpp:230-261): no DC removal here
                const float* d = x + peak + kCoxCp;
#pragma unroll
                for (int r = 0; r < 16; ++r) s_buf[lane + 64 * r] = make_float2(d[lane + 64 * r], 0.0f);
                wave_sync();
                cox_analytic_wave(s_buf, A.twiddle, lane);
                float pr[8], pi[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int i = lane + 64 * t, j = i + 512;
                    const float2 s1 = s_buf[i + (i >> 4)], s2 = s_buf[j + (j >> 4)];
                    const float a = s1.x, bb = -s1.y, c = s2.x, dd = s2.y;
                    pr[t] = a * c - bb * dd;
                    pi[t] = a * dd + bb * c;
                }
                wave_sync();
                float* f = reinterpret_cast<float*>(s_buf);
#pragma unroll
                for (int t = 0; t < 8; ++t) { f[lane + 64 * t] = pr[t]; f[512 + lane + 64 * t] = pi[t]; }
                wave_sync();
                float acc = 0.0f;
                if (lane < 2) acc = cox_chain512(f + 512 * lane);
                const float Pr = __shfl(acc, 0), Pi = __shfl(acc, 1);
                if (lane == 0) {
                    const float phase = atan2f_glibc(Pi, Pr);
                    const float cfo = static_cast<float>(static_cast<double>(phase * 48000.0f) / (3.14159265358979323846 * 1024.0));
                    const float max_cfo = 46.0f;             // uint32 48000 / size_t 1024: integer division (:253)
                    ria_cox_result o{};
                    o.found = 1; o.start_sample = sh_lts; o.cfo_hz = maxf_(-max_cfo, minf_(max_cfo, cfo));
                    o.noise_floor = sh_nf; o.sts_position = peak;
                    *out = o;
                }
            }
            return;
        }
        __syncthreads();   // sh_* are rewritten by lane 0 at the top of the next round
    }
    if (tid == 0) { ria_cox_result o{}; o.noise_floor = sh_nf; *out = o; }
}

}  // namespace ria
