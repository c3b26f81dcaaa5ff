/* oracle/ria_oracle_sync.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the reference's acquisition correlators:
 *   sync::ZCSync     src/sync/zc_sync.hpp   generateZC :420-436, generatePreambleForRoot :133-190,
 *                                           correlate :485-626, computeCorrelationMag :441-482, detect :192-391
 * Plain C, float arithmetic written operation by operation in the reference's order (left-to-right sums,
 * double where `M_PI` promotes an expression).  Pinned bit-for-bit against oracle/_ref (the unmodified
 * reference) by oracle/check_against_ref.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ria_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------------ ZC */
#define ZC_N 127
#define ZC_UP 8
#define ZC_REP (ZC_N * ZC_UP)              /* 1016 samples per repetition */
#define ZC_GAP 480                         /* 10 ms at 48 kHz */
#define ZC_PREAMBLE (2 * ZC_REP + ZC_GAP)  /* 2512 */
static const float kZcFs = 48000.0f, kZcFc = 1500.0f;

static void zc_sequence(int root, float* re, float* im) { /* zc_sync.hpp:420-436 (N odd) */
    for (int n = 0; n < ZC_N; ++n) {
        float phase = (float)(-M_PI * root * n * (n + 1) / ZC_N);
        re[n] = cosf(phase);
        im[n] = sinf(phase);
    }
}
/* interpolated reference sample i (0..1015): zc[c]*(1-frac) + zc[c+1]*frac (zc_sync.hpp:147-157) */
static void zc_interp_table(int root, float* tre, float* tim) {
    float re[ZC_N], im[ZC_N];
    zc_sequence(root, re, im);
    for (int i = 0; i < ZC_REP; ++i) {
        float chip_pos = (float)i / ZC_UP;
        int c = (int)chip_pos;
        float frac = chip_pos - c;
        if (c < ZC_N - 1) {
            float a = 1.0f - frac;
            tre[i] = re[c] * a + re[c + 1] * frac;
            tim[i] = im[c] * a + im[c + 1] * frac;
        } else {
            tre[i] = re[c];
            tim[i] = im[c];
        }
    }
}

int ro_zc_preamble_samples(void) { return ZC_PREAMBLE; }

int ro_zc_generate(int root, float* out, int max_n) { /* zc_sync.hpp:133-190 */
    if (max_n < ZC_PREAMBLE) return -ZC_PREAMBLE;
    float tre[ZC_REP], tim[ZC_REP];
    zc_interp_table(root, tre, tim);
    for (int rep = 0; rep < 2; ++rep)
        for (int i = 0; i < ZC_REP; ++i) {
            int g = rep * ZC_REP + i;
            float t = (float)g / kZcFs;
            float ph = (float)(2.0f * M_PI * kZcFc * t);
            out[g] = tre[i] * cosf(ph) - tim[i] * sinf(ph);
        }
    float max_amp = 0.0f;
    for (int i = 0; i < 2 * ZC_REP; ++i) { float a = fabsf(out[i]); if (a > max_amp) max_amp = a; }
    if (max_amp > 0.0f) {
        float scale = 0.8f / max_amp;
        for (int i = 0; i < 2 * ZC_REP; ++i) out[i] *= scale;
    }
    for (int i = 0; i < ZC_GAP; ++i) out[2 * ZC_REP + i] = 0.0f;
    return ZC_PREAMBLE;
}

/* baseband sample i: rx[i] * (cos(phase), sin(phase)), phase = -2*pi*f*t (zc_sync.hpp:503-508) */
static void zc_mix(const float* rx, int i, float f, float* bre, float* bim) {
    float t = (float)i / kZcFs;
    float phase = (float)(-2.0f * M_PI * f * t);
    *bre = rx[i] * cosf(phase);
    *bim = rx[i] * sinf(phase);
}
/* sum over one repetition of bb * conj(ref), plus the received energy */
static void zc_corr_at(const float* rx, int lag, float f, const float* tre, const float* tim, float* sre, float* sim_,
                       float* energy) {
    float sr = 0.0f, si = 0.0f, e = 0.0f;
    for (int i = 0; i < ZC_REP; ++i) {
        float br, bi;
        zc_mix(rx, lag + i, f, &br, &bi);
        /* bb * conj(z) with std::complex<float> operator*: (a+bi)(c+di'), d' = -d */
        float c = tre[i], d = -tim[i];
        sr += br * c - bi * d;
        si += br * d + bi * c;
        e += br * br + bi * bi;
    }
    *sre = sr; *sim_ = si; *energy = e;
}
static float zc_corr_mag(const float* rx, int n, int lag, float f, const float* tre, const float* tim) { /* :441-482 */
    if (lag < 0 || lag + ZC_REP > n) return 0.0f;
    float sr, si, e;
    zc_corr_at(rx, lag, f, tre, tim, &sr, &si, &e);
    float denom = sqrtf(e * (float)ZC_REP);
    return (denom > 1e-10f) ? hypotf(sr, si) / denom : 0.0f;
}
static float zc_corr_to_snr(float corr) { /* :628-633 */
    if (corr <= 0.01f) return -10.0f;
    if (corr >= 0.99f) return 30.0f;
    float snr = 20.0f * log10f(corr / (1.0f - corr + 0.01f));
    if (snr < -10.0f) snr = -10.0f;
    if (snr > 30.0f) snr = 30.0f;
    return snr;
}

/* out7: detected, frame_type, start_sample, correlation, cfo_hz, snr_estimate, root_detected */
int ro_zc_detect(const float* rx, int n, float threshold, int root_mask, float known_cfo_hz, float* out7) { /* :192-391 */
    static const int roots[4] = {1, 3, 5, 7};
    out7[0] = 0.f; out7[1] = 255.f; out7[2] = -1.f; out7[3] = 0.f; out7[4] = 0.f; out7[5] = 0.f; out7[6] = -1.f;
    if (n < ZC_REP) return 0;
    const float f = kZcFc + known_cfo_hz;
    const int corr_len = n - ZC_REP + 1, step = ZC_REP / 32; /* 31 */
    float* cre = (float*)malloc(sizeof(float) * (size_t)corr_len);
    float* cim = (float*)malloc(sizeof(float) * (size_t)corr_len);
    float best_corr = 0.0f, best_cfo = 0.0f;
    int best_root = -1, best_pos = -1;
    for (int ri = 0; ri < 4; ++ri) {
        if (!(root_mask & (1 << ri))) continue;
        float tre[ZC_REP], tim[ZC_REP];
        zc_interp_table(roots[ri], tre, tim);
        memset(cre, 0, sizeof(float) * (size_t)corr_len);
        memset(cim, 0, sizeof(float) * (size_t)corr_len);
        /* correlate(): coarse search, fine search around the coarse peak, coarse values kept */
        int coarse_pos = 0;
        float coarse_mag = 0.0f;
        for (int lag = 0; lag < corr_len; lag += step) {
            float sr, si, e;
            zc_corr_at(rx, lag, f, tre, tim, &sr, &si, &e);
            float denom = sqrtf(e * (float)ZC_REP);
            float mag = (denom > 1e-10f) ? hypotf(sr, si) / denom : 0.0f;
            if (mag > coarse_mag) { coarse_mag = mag; coarse_pos = lag; }
            if (denom > 1e-10f) { cre[lag] = sr / denom; cim[lag] = si / denom; }
        }
        int fine_start = coarse_pos - step < 0 ? 0 : coarse_pos - step;
        int fine_end = coarse_pos + step + 1 > corr_len ? corr_len : coarse_pos + step + 1;
        for (int lag = fine_start; lag < fine_end; ++lag) {
            float sr, si, e;
            zc_corr_at(rx, lag, f, tre, tim, &sr, &si, &e);
            float denom = sqrtf(e * (float)ZC_REP);
            if (denom > 1e-10f) { cre[lag] = sr / denom; cim[lag] = si / denom; } else { cre[lag] = 0.f; cim[lag] = 0.f; }
        }
        /* detect(): earliest strongest peak */
        float peak_mag = 0.0f;
        int peak_pos = 0;
        for (int i = 0; i < corr_len; ++i) {
            float mag = hypotf(cre[i], cim[i]);
            if (mag > peak_mag) { peak_mag = mag; peak_pos = i; }
        }
        int timing_pos = peak_pos;
        if (peak_mag > threshold && peak_pos >= ZC_REP) {
            int earlier = peak_pos - ZC_REP;
            float em = zc_corr_mag(rx, n, earlier, f, tre, tim);
            if (em > peak_mag * 0.4f) timing_pos = earlier;
        }
        float combined = peak_mag;
        if (peak_mag > 0.0f && peak_mag < 0.25f) {
            int rep2 = timing_pos + ZC_REP;
            if (rep2 + ZC_REP <= n) {
                float m1 = zc_corr_mag(rx, n, timing_pos, f, tre, tim);
                float m2 = zc_corr_mag(rx, n, rep2, f, tre, tim);
                combined = sqrtf(m1 * m1 + m2 * m2) / sqrtf(2.0f);
                if (!(combined > peak_mag)) combined = peak_mag; /* std::max(combined, peak) */
            }
        }
        if (combined > best_corr) {
            best_corr = combined;
            best_root = roots[ri];
            best_pos = timing_pos;
            int rep2 = timing_pos + ZC_REP;
            if (rep2 + ZC_REP <= n) {
                float r1, i1, e1, r2, i2, e2;
                zc_corr_at(rx, timing_pos, f, tre, tim, &r1, &i1, &e1);
                zc_corr_at(rx, rep2, f, tre, tim, &r2, &i2, &e2);
                float m1 = hypotf(r1, i1) / ZC_REP, m2 = hypotf(r2, i2) / ZC_REP;
                if (m1 > 0.1f && m2 > 0.1f) {
                    /* corr2 * conj(corr1) */
                    float c = r1, d = -i1;
                    float pr = r2 * c - i2 * d, pi_ = r2 * d + i2 * c;
                    float phase_diff = atan2f(pi_, pr);
                    float rep_duration = (float)ZC_REP / kZcFs;
                    best_cfo = (float)(phase_diff / (2.0f * M_PI * rep_duration));
                }
            }
        }
    }
    free(cre); free(cim);
    out7[3] = best_corr;
    out7[6] = (float)best_root;
    if (best_root >= 0) out7[1] = (float)((best_root - 1) / 2); /* roots 1,3,5,7 -> PING, PONG, DATA, CONTROL */
    if (best_corr > threshold && best_root >= 0) {
        out7[0] = 1.f;
        out7[4] = best_cfo;
        out7[2] = (float)(best_pos + ZC_PREAMBLE);
        out7[5] = zc_corr_to_snr(best_corr);
        return 1;
    }
    return 0;
}
