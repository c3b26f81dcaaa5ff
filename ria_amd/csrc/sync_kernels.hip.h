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

__global__ __launch_bounds__(256) void zc_detect_kernel(ZcArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* bb = reinterpret_cast<float2*>(smem);
    ZcRootOut* ro = reinterpret_cast<ZcRootOut*>(smem + static_cast<size_t>(A.buf_len) * sizeof(float2));
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

}  // namespace ria
