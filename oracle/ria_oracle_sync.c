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

/* ------------------------------------------------------------------------------------------------ chirp
 * sync::ChirpSync   src/sync/chirp_sync.hpp   generateTemplate :874-900, generate :61-108, initFFT :573-623,
 *                   detectChirpTemplateFFT :627-712, detectChirpTemplate :717-818,
 *                   computeComplexTemplateCorrelation :829-851, detectDualChirp :352-512
 * FFT: src/dsp/fft.cpp:83-128 (radix-2 DIT, twiddles cosf/sinf(-2*pi*k/N), inverse = conjugated twiddles + 1/N).
 */
#define CH_LEN 24000      /* 500 ms */
#define CH_GAP 4800       /* 100 ms */
#define CH_FFT 131072
typedef struct { float re, im; } scf;

static float* g_ch_up_sin, *g_ch_up_cos, *g_ch_dn_sin, *g_ch_dn_cos;
static float g_ch_up_energy, g_ch_dn_energy;
static scf* g_ch_tw;          /* [CH_FFT/2] */
static scf* g_ch_up_fft, *g_ch_dn_fft;   /* conj(FFT(template)) */

static float ch_up_phase(float t) {
    float T = 500.0f / 1000.0f, k = (2700.0f - 300.0f) / T;
    return (float)(2.0f * M_PI * (300.0f * t + 0.5f * k * t * t));
}
static float ch_dn_phase(float t) {
    float T = 500.0f / 1000.0f, k = (2700.0f - 300.0f) / T;
    return (float)(2.0f * M_PI * (2700.0f * t - 0.5f * k * t * t));
}
static void ch_fft(scf* data, int inverse) {
    int size = CH_FFT, j = 0;
    for (int i = 0; i < size - 1; ++i) {
        if (i < j) { scf t = data[i]; data[i] = data[j]; data[j] = t; }
        int k = size / 2;
        while (k <= j) { j -= k; k /= 2; }
        j += k;
    }
    for (int len = 2; len <= size; len *= 2) {
        int half = len / 2, step = size / len;
        for (int i = 0; i < size; i += len)
            for (int k = 0; k < half; ++k) {
                scf w = g_ch_tw[k * step];
                if (inverse) w.im = -w.im;
                scf d = data[i + k + half], a = data[i + k], t;
                t.re = w.re * d.re - w.im * d.im;
                t.im = w.re * d.im + w.im * d.re;
                data[i + k + half].re = a.re - t.re; data[i + k + half].im = a.im - t.im;
                data[i + k].re = a.re + t.re; data[i + k].im = a.im + t.im;
            }
    }
    if (inverse) {
        float scale = 1.0f / (float)size;
        for (int i = 0; i < size; ++i) { data[i].re *= scale; data[i].im *= scale; }
    }
}
static void ch_init(void) {
    if (g_ch_tw) return;
    g_ch_up_sin = (float*)malloc(sizeof(float) * CH_LEN); g_ch_up_cos = (float*)malloc(sizeof(float) * CH_LEN);
    g_ch_dn_sin = (float*)malloc(sizeof(float) * CH_LEN); g_ch_dn_cos = (float*)malloc(sizeof(float) * CH_LEN);
    g_ch_up_energy = 0.0f; g_ch_dn_energy = 0.0f;
    for (int i = 0; i < CH_LEN; ++i) {
        float t = (float)i / 48000.0f;
        float p = ch_up_phase(t);
        g_ch_up_sin[i] = sinf(p); g_ch_up_cos[i] = cosf(p);
        g_ch_up_energy += g_ch_up_sin[i] * g_ch_up_sin[i];
    }
    for (int i = 0; i < CH_LEN; ++i) {
        float t = (float)i / 48000.0f;
        float p = ch_dn_phase(t);
        g_ch_dn_sin[i] = sinf(p); g_ch_dn_cos[i] = cosf(p);
        g_ch_dn_energy += g_ch_dn_sin[i] * g_ch_dn_sin[i];
    }
    scf* tw = (scf*)malloc(sizeof(scf) * CH_FFT / 2);
    for (int k = 0; k < CH_FFT / 2; ++k) {
        float angle = (float)(-2.0f * M_PI * (double)k / (double)CH_FFT);
        tw[k].re = cosf(angle); tw[k].im = sinf(angle);
    }
    g_ch_tw = tw;
    g_ch_up_fft = (scf*)calloc(CH_FFT, sizeof(scf));
    g_ch_dn_fft = (scf*)calloc(CH_FFT, sizeof(scf));
    for (int i = 0; i < CH_LEN; ++i) {
        g_ch_up_fft[i].re = g_ch_up_cos[i]; g_ch_up_fft[i].im = g_ch_up_sin[i];
        g_ch_dn_fft[i].re = g_ch_dn_cos[i]; g_ch_dn_fft[i].im = g_ch_dn_sin[i];
    }
    ch_fft(g_ch_up_fft, 0); ch_fft(g_ch_dn_fft, 0);
    for (int i = 0; i < CH_FFT; ++i) { g_ch_up_fft[i].im = -g_ch_up_fft[i].im; g_ch_dn_fft[i].im = -g_ch_dn_fft[i].im; }
}

int ro_chirp_generate(float* out, int max_n) { /* :61-108, dual chirp, tx_cfo 0 */
    const int total = 2 * CH_LEN + 2 * CH_GAP;
    if (max_n < total) return -total;
    memset(out, 0, sizeof(float) * (size_t)total);
    float T = 500.0f / 1000.0f, k = (2700.0f - 300.0f) / T, cfo = 0.0f;
    float f_start_up = 300.0f + cfo, f_start_down = 2700.0f + cfo;
    for (int i = 0; i < CH_LEN; ++i) {
        float t = (float)i / 48000.0f;
        float phase = (float)(2.0f * M_PI * (f_start_up * t + 0.5f * k * t * t));
        out[i] = 0.5f * sinf(phase);
    }
    for (int i = 0; i < CH_LEN; ++i) {
        float t = (float)i / 48000.0f;
        float phase = (float)(2.0f * M_PI * (f_start_down * t - 0.5f * k * t * t));
        out[CH_LEN + CH_GAP + i] = 0.5f * sinf(phase);
    }
    return total;
}

static float ch_td_corr(const float* s, int n, int offset, const float* tsin, const float* tcos, float tmpl_energy) { /* :829-851 */
    if (offset + CH_LEN > n) return 0.0f;
    float ci = 0.0f, cq = 0.0f, e = 0.0f;
    for (int i = 0; i < CH_LEN; ++i) {
        float x = s[offset + i];
        ci += x * tcos[i];
        cq += x * tsin[i];
        e += x * x;
    }
    float denom = sqrtf(e * tmpl_energy);
    if (denom < 1e-10f) return 0.0f;
    return sqrtf(ci * ci + cq * cq) / denom;
}

/* detectChirpTemplate: returns position (or -1) and the correlation */
static int ch_detect_template(const float* s, int n, int down, float threshold, float* corr_out) {
    const float* tsin = down ? g_ch_dn_sin : g_ch_up_sin;
    const float* tcos = down ? g_ch_dn_cos : g_ch_up_cos;
    const float tmpl_energy = down ? g_ch_dn_energy : g_ch_up_energy;
    *corr_out = 0.0f;
    if (n < CH_LEN) return -1;
    if (n >= 2 * CH_LEN) { /* detectChirpTemplateFFT :627-712 */
        const scf* tf = down ? g_ch_dn_fft : g_ch_up_fft;
        const int fft_in = n < CH_FFT ? n : CH_FFT;
        const int search_len = fft_in - CH_LEN;
        scf* buf = (scf*)calloc(CH_FFT, sizeof(scf));
        for (int i = 0; i < fft_in; ++i) buf[i].re = s[i];
        ch_fft(buf, 0);
        for (int i = 0; i < CH_FFT; ++i) { /* signal_fft * tmpl_fft (std::complex operator*) */
            scf a = buf[i], b = tf[i];
            buf[i].re = a.re * b.re - a.im * b.im;
            buf[i].im = a.re * b.im + a.im * b.re;
        }
        ch_fft(buf, 1);
        float* cum = (float*)malloc(sizeof(float) * (size_t)(fft_in + 1));
        cum[0] = 0.0f;
        for (int i = 0; i < fft_in; ++i) cum[i + 1] = cum[i] + s[i] * s[i];
        float best = 0.0f;
        int best_pos = -1;
        for (int pos = 0; pos < search_len; ++pos) {
            float mag = hypotf(buf[pos].re, buf[pos].im);
            float se = cum[pos + CH_LEN] - cum[pos];
            float denom = sqrtf(se * tmpl_energy);
            float nc = (denom > 1e-10f) ? mag / denom : 0.0f;
            if (nc > best) { best = nc; best_pos = pos; }
        }
        free(buf); free(cum);
        *corr_out = best;
        return (best < threshold) ? -1 : best_pos;
    }
    /* time-domain fallback :759-817 */
    const int search_len = n - CH_LEN;
    float best = 0.0f;
    int best_pos = -1;
    for (int pos = 0; pos < search_len; pos += 48) {
        float c = ch_td_corr(s, n, pos, tsin, tcos, tmpl_energy);
        if (c > best) { best = c; best_pos = pos; }
    }
    *corr_out = best;
    if (best_pos < 0 || best < threshold * 0.3f) return -1;
    int fine_start = best_pos - 48 < 0 ? 0 : best_pos - 48;
    int fine_end = best_pos + 48 > search_len ? search_len : best_pos + 48;
    for (int pos = fine_start; pos <= fine_end; ++pos) {
        float c = ch_td_corr(s, n, pos, tsin, tcos, tmpl_energy);
        if (c > best) { best = c; best_pos = pos; }
    }
    if (best_pos > 0 && best_pos < search_len - 1) {
        float c0 = ch_td_corr(s, n, best_pos - 1, tsin, tcos, tmpl_energy);
        float c1 = best;
        float c2 = ch_td_corr(s, n, best_pos + 1, tsin, tcos, tmpl_energy);
        float denom = 2.0f * (c0 - 2.0f * c1 + c2);
        if (fabsf(denom) > 1e-10f) {
            float delta = (c0 - c2) / denom;
            float lo = (1.0f < delta) ? 1.0f : delta;   /* std::min(1.0f, delta) */
            delta = (-1.0f < lo) ? lo : -1.0f;          /* std::max(-1.0f, .) */
            best_pos = (int)roundf((float)best_pos + delta);
        }
    }
    *corr_out = best;
    return (best >= threshold) ? best_pos : -1;
}

/* out6 = {success, up_chirp_start, down_chirp_start, cfo_hz, up_correlation, down_correlation} */
int ro_chirp_detect(const float* s, int n, float threshold, float* out6) { /* detectDualChirp :352-512 */
    ch_init();
    out6[0] = 0.f; out6[1] = -1.f; out6[2] = -1.f; out6[3] = 0.f; out6[4] = 0.f; out6[5] = 0.f;
    if (n < 2 * CH_LEN + CH_GAP) return 0;
    float up_corr, down_corr;
    int up_pos = ch_detect_template(s, n, 0, threshold, &up_corr);
    out6[4] = up_corr;
    if (up_pos < 0) return 0;
    long long down_search_start = (long long)up_pos + CH_LEN / 2;
    long long expected_down_pos = (long long)up_pos + CH_LEN + CH_GAP;
    long long min_search_len = 2 * CH_LEN + 1000;
    long long a = expected_down_pos + 10000 + CH_LEN, b = down_search_start + min_search_len;
    long long down_search_end = (a > b) ? a : b;
    if (down_search_end > n) down_search_end = n;
    if (down_search_start >= n) return 0;
    if (down_search_end <= down_search_start + CH_LEN) {
        down_search_end = down_search_start + 2 * CH_LEN;
        if (down_search_end > n) down_search_end = n;
    }
    int down_len = (int)(down_search_end - down_search_start);
    int down_rel = ch_detect_template(s + down_search_start, down_len, 1, threshold, &down_corr);
    if (down_rel < 0) return 0;
    int down_pos = down_rel + (int)down_search_start;
    out6[5] = down_corr;
    float T = 500.0f / 1000.0f;
    float chirp_rate = (2700.0f - 300.0f) / T;
    float cfo_to_samples = 48000.0f / chirp_rate;
    int expected_gap = CH_LEN + CH_GAP;
    int actual_gap = down_pos - up_pos;
    float gap_error = (float)(actual_gap - expected_gap);
    float cfo = gap_error / (2.0f * cfo_to_samples);
    out6[3] = cfo;
    if (fabsf(cfo) > 100.0f) return 0;
    float up_correction = cfo * cfo_to_samples;
    float down_correction = -cfo * cfo_to_samples;
    out6[1] = (float)(int)roundf((float)up_pos + up_correction);
    out6[2] = (float)(int)roundf((float)down_pos + down_correction);
    out6[0] = 1.f;
    return 1;
}

/* ------------------------------------------------------------------------------------------------ MC-DPSK
 * src/psk/multi_carrier_dpsk.hpp: modulator :126-323 (generateTrainingSequence, generateReferenceSymbol,
 * modulate), demodulator: applyCFOCorrection :901-926 (HilbertTransform, src/dsp/filters.cpp:266-317),
 * demodulateOneSymbol :931-946, processTraining :473-505, setReference :507-518, demodulateSoft :520-736,
 * fading indices :404-437; driven as processGotChirp :797-896 with an external chirp detection.
 * HARQ chase combine: src/fec/chase_cache.cpp:27-88 (store).
 */
#define MC_SPS 512
#define MC_TRAIN 8
#define MC_MAXC 20
static const float kMcFs = 48000.0f;
static const float kPiF = (float)M_PI;

static void mc_freqs(int nc, float* f) { /* getCarrierFreqs :66-78 */
    if (nc == 1) { f[0] = (500.0f + 2500.0f) / 2.0f; return; }
    float spacing = (2500.0f - 500.0f) / (nc - 1);
    for (int i = 0; i < nc; ++i) f[i] = 500.0f + i * spacing;
}
typedef struct { float re, im; } mcf;
static mcf mc_polar(float rho, float theta) { mcf r = {rho * cosf(theta), rho * sinf(theta)}; return r; }
static mcf mc_mul(mcf a, mcf b) { mcf r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static mcf mc_conj(mcf a) { mcf r = {a.re, -a.im}; return r; }

int ro_mcdpsk_modulate(int nc, int bps, int spreading, const uint8_t* data, int n_bytes, float* out, int max_n) {
    float freqs[MC_MAXC];
    mcf prev[MC_MAXC];
    mc_freqs(nc, freqs);
    int bits_per_sym = nc * bps;
    int n_bits = n_bytes * 8;
    int n_data_sym = (n_bits + bits_per_sym - 1) / bits_per_sym;
    int total = (MC_TRAIN + 1 + n_data_sym * spreading) * MC_SPS;
    if (total > max_n) return -total;
    memset(out, 0, sizeof(float) * (size_t)total);
    /* training :141-175 */
    for (int sym = 0; sym < MC_TRAIN; ++sym)
        for (int c = 0; c < nc; ++c) {
            float phase_offset = (float)((c * sym) * M_PI / 2.0f);
            mcf ts = mc_polar(1.0f, phase_offset);
            float phase_inc = (float)(2.0f * M_PI * freqs[c] / kMcFs);
            for (int i = 0; i < MC_SPS; ++i) {
                float t = i * phase_inc;
                mcf m = mc_mul(ts, mc_polar(1.0f, t));
                out[sym * MC_SPS + i] += m.re / nc;
            }
        }
    /* reference :178-199 */
    float* ref = out + MC_TRAIN * MC_SPS;
    for (int c = 0; c < nc; ++c) {
        float phase_inc = (float)(2.0f * M_PI * freqs[c] / kMcFs);
        mcf rs = {1.0f, 0.0f};
        prev[c] = rs;
        for (int i = 0; i < MC_SPS; ++i) {
            float t = i * phase_inc;
            mcf m = mc_mul(rs, mc_polar(1.0f, t));
            ref[i] += m.re / nc;
        }
    }
    /* data :202-281 */
    float* dat = ref + MC_SPS;
    float* one = (float*)malloc(sizeof(float) * MC_SPS);
    int bit_idx = 0;
    for (int ds = 0; ds < n_data_sym; ++ds) {
        memset(one, 0, sizeof(float) * MC_SPS);
        for (int c = 0; c < nc; ++c) {
            int sb = 0;
            for (int b = 0; b < bps; ++b) {
                int bit = (bit_idx < n_bits) ? (data[bit_idx >> 3] >> (7 - (bit_idx & 7))) & 1 : 0;
                ++bit_idx;
                sb = (sb << 1) | bit;
            }
            float phase_change;
            if (bps == 2) {
                const float ph[4] = {(float)(M_PI / 4), (float)(3 * M_PI / 4), (float)(-3 * M_PI / 4), (float)(-M_PI / 4)};
                phase_change = ph[sb];
            } else {
                phase_change = sb ? (float)M_PI : 0.0f;
            }
            mcf cur = mc_mul(prev[c], mc_polar(1.0f, phase_change));
            float a = hypotf(cur.re, cur.im);
            cur.re /= a; cur.im /= a;
            prev[c] = cur;
            float phase_inc = (float)(2.0f * M_PI * freqs[c] / kMcFs);
            for (int i = 0; i < MC_SPS; ++i) {
                float t = i * phase_inc;
                mcf m = mc_mul(cur, mc_polar(1.0f, t));
                one[i] += m.re / nc;
            }
        }
        for (int rep = 0; rep < spreading; ++rep) memcpy(dat + (size_t)(ds * spreading + rep) * MC_SPS, one, sizeof(float) * MC_SPS);
    }
    free(one);
    return total;
}

static mcf mc_demod_symbol(const float* s, float freq) { /* :931-946 */
    float phase_inc = (float)(2.0f * M_PI * freq / kMcFs);
    float sr = 0.0f, si = 0.0f, phase = 0.0f;
    for (int i = 0; i < MC_SPS; ++i) {
        mcf m = mc_polar(1.0f, -phase);
        sr += s[i] * m.re;
        si += s[i] * m.im;
        phase += phase_inc;
    }
    mcf r = {sr / (float)MC_SPS, si / (float)MC_SPS};
    return r;
}

static void mc_apply_cfo(float* x, int n, float cfo_hz, float phase0) { /* :901-926 + filters.cpp:266-317 */
    if (fabsf(cfo_hz) < 0.01f || n < 128) return;
    enum { TAPS = 127, M = 63 };
    float coeff[TAPS];
    for (int k = 0; k < TAPS; ++k) {
        int kk = k - M;
        if (kk == 0) coeff[k] = 0;
        else if (kk % 2 != 0) coeff[k] = (float)(2.0f / (M_PI * kk));
        else coeff[k] = 0;
        float w = (float)(2.0f * M_PI * k / (TAPS - 1));
        coeff[k] *= 0.42f - 0.5f * cosf(w) + 0.08f * cosf(2.0f * w);
    }
    float* ar = (float*)malloc(sizeof(float) * (size_t)n);
    float* ai = (float*)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        float q = 0;
        for (int k = 0; k < TAPS; ++k) {
            float v = (i - k >= 0) ? x[i - k] : 0.0f;
            q += coeff[k] * v;
        }
        ar[i] = (i - M >= 0) ? x[i - M] : 0.0f;
        ai[i] = q;
    }
    float phase_inc = (float)(-2.0f * M_PI * cfo_hz / kMcFs);
    float phase = phase0;
    for (int i = 0; i < n; ++i) {
        float rc = cosf(phase), rs = sinf(phase);
        x[i] = ar[i] * rc - ai[i] * rs;
        phase += phase_inc;
        if (phase > M_PI) phase = (float)(phase - 2.0f * M_PI);
        if (phase < -M_PI) phase = (float)(phase + 2.0f * M_PI);
    }
    free(ar); free(ai);
}

/* aux4 = {cfo after the frame, fading index, frequency fading index, temporal fading index}; returns #LLRs */
int ro_mcdpsk_demod(int nc, int bps, int spreading, const float* samples, int n, float cfo_hz, float phase0,
                    float* llr_out, int max_llr, float* aux4) {
    float freqs[MC_MAXC];
    mcf prev[MC_MAXC];
    mc_freqs(nc, freqs);
    const int local_preamble = (MC_TRAIN + 1) * MC_SPS;
    int data_samples;
    if (n > local_preamble) data_samples = n - local_preamble;
    else { int bpsym = nc * bps; data_samples = ((648 + bpsym - 1) / bpsym) * MC_SPS; }
    if (n < local_preamble + data_samples) return -1;
    float* x = (float*)malloc(sizeof(float) * (size_t)n);
    memcpy(x, samples, sizeof(float) * (size_t)n);
    float cfo = cfo_hz;
    if (fabsf(cfo) > 0.1f) { mc_apply_cfo(x, n, cfo, phase0); cfo = 0.0f; }
    /* processTraining (its estimate is discarded after an external chirp detection, :857-861) */
    float saved_cfo = cfo;
    {
        float sum = 0.0f;
        for (int c = 0; c < nc; ++c) {
            mcf s0 = mc_demod_symbol(x, freqs[c]), s1 = mc_demod_symbol(x + MC_SPS, freqs[c]);
            float expected_phase = (float)((c * 1 - c * 0) * M_PI / 2.0f);
            mcf ed = mc_polar(1.0f, expected_phase);
            mcf ad = mc_mul(s1, mc_conj(s0));
            mcf er = mc_mul(ad, mc_conj(ed));
            sum += atan2f(er.im, er.re);
        }
        float avg = sum / nc;
        float symbol_duration = MC_SPS / kMcFs;
        float residual = (float)(avg / (2.0f * M_PI * symbol_duration));
        cfo += residual;
        float lo = (50.0f < cfo) ? 50.0f : cfo;
        cfo = (-50.0f < lo) ? lo : -50.0f;
    }
    cfo = saved_cfo;
    /* setReference */
    const float* rsym = x + MC_TRAIN * MC_SPS;
    for (int c = 0; c < nc; ++c) {
        mcf p = mc_demod_symbol(rsym, freqs[c]);
        float a = hypotf(p.re, p.im);
        if (a > 0.001f) { float a2 = hypotf(p.re, p.im); p.re /= a2; p.im /= a2; } else { p.re = 1.0f; p.im = 0.0f; }
        prev[c] = p;
    }
    /* demodulateSoft */
    const float* data = x + local_preamble;
    int num_rx = data_samples / MC_SPS;
    int nds = num_rx / spreading;
    if (nds < 1) nds = 1;
    int n_llr = nds * nc * bps;
    if (n_llr > max_llr) { free(x); return -n_llr; }
    float mag_sum[MC_MAXC] = {0}, mag_sq[MC_MAXC] = {0};
    float* sym_total = (float*)calloc((size_t)nds, sizeof(float));
    float* cph = (float*)malloc(sizeof(float) * (size_t)nds * nc);
    float* cmag = (float*)calloc((size_t)nds * nc, sizeof(float));
    float noise_sum = 0.0f;
    int noise_count = 0;
    for (int ds = 0; ds < nds; ++ds)
        for (int c = 0; c < nc; ++c) {
            mcf comb = {0.0f, 0.0f};
            for (int rep = 0; rep < spreading; ++rep) {
                int rs = ds * spreading + rep;
                if (rs >= num_rx) break;
                mcf cur = mc_demod_symbol(data + (size_t)rs * MC_SPS, freqs[c]);
                comb.re += cur.re; comb.im += cur.im;
            }
            comb.re /= (float)spreading; comb.im /= (float)spreading;
            float mag = hypotf(comb.re, comb.im);
            cmag[ds * nc + c] = mag;
            sym_total[ds] += mag;
            mag_sum[c] += mag;
            mag_sq[c] += mag * mag;
            mcf nrm;
            if (mag > 0.0001f) { nrm.re = comb.re / mag; nrm.im = comb.im / mag; } else { nrm.re = 1.0f; nrm.im = 0.0f; }
            mcf diff = mc_mul(nrm, mc_conj(prev[c]));
            prev[c] = nrm;
            float phase = atan2f(diff.im, diff.re);
            cph[ds * nc + c] = phase;
            float pe;
            if (bps == 2) {
                float shifted = phase - kPiF / 4.0f;
                float nearest = roundf(shifted / (kPiF / 2.0f));
                float ideal = nearest * kPiF / 2.0f + kPiF / 4.0f;
                pe = phase - ideal;
            } else {
                float nearest = roundf(phase / kPiF);
                float ideal = nearest * kPiF;
                pe = phase - ideal;
            }
            while (pe > kPiF) pe -= 2.0f * kPiF;
            while (pe < -kPiF) pe += 2.0f * kPiF;
            noise_sum += pe * pe;
            noise_count++;
        }
    int valid = nds;
    if (nds >= 4) {
        float ref_mag = 0.0f;
        for (int s = 0; s < 4; ++s) ref_mag += sym_total[s];
        ref_mag /= 4.0f;
        if (ref_mag > 0.001f) {
            float thr = ref_mag * 0.2f;
            while (valid > 4 && sym_total[valid - 1] < thr) valid--;
            if (valid < nds) {
                for (int c = 0; c < nc; ++c) { mag_sum[c] = 0.0f; mag_sq[c] = 0.0f; }
                for (int s = 0; s < valid; ++s)
                    for (int c = 0; c < nc; ++c) { float m = cmag[s * nc + c]; mag_sum[c] += m; mag_sq[c] += m * m; }
            }
        }
    }
    float pnv = (noise_count > 0) ? noise_sum / noise_count : 0.5f;
    pnv = (0.01f < pnv) ? pnv : 0.01f;                  /* std::max(0.01f, pnv) */
    float scale = 2.0f * sqrtf(1.0f / pnv);
    scale = (20.0f < scale) ? 20.0f : scale;            /* std::min(scale, 20.0f) */
    float rel[MC_MAXC];
    for (int c = 0; c < nc; ++c) rel[c] = 1.0f;
    if (bps == 1 && valid > 0) {
        float mean_mag[MC_MAXC];
        float gsum = 0.0f;
        int gcount = 0;
        for (int c = 0; c < nc; ++c) {
            float mean = mag_sum[c] / valid;
            mean_mag[c] = mean;
            if (mean > 1e-4f) { gsum += mean; gcount++; }
        }
        float gmean = (gcount > 0) ? gsum / gcount : 0.0f;
        for (int c = 0; c < nc; ++c) {
            float mean = mean_mag[c];
            if (mean <= 1e-4f || gmean <= 1e-4f) { rel[c] = 0.12f; continue; }
            float mean_sq = mag_sq[c] / valid;
            float var = mean_sq - mean * mean;
            var = (0.0f < var) ? var : 0.0f;            /* std::max(0.0f, var) */
            float cv = sqrtf(var) / (mean + 1e-6f);
            float ratio = mean / gmean;
            float t1 = (ratio < 1.25f) ? ratio : 1.25f; /* std::min(1.25f, ratio) */
            float mw = (0.10f < t1) ? t1 : 0.10f;
            float sw = 1.0f / (1.0f + 1.5f * cv);
            float wd = 1.0f;
            if (ratio < 0.20f) wd = 0.25f; else if (ratio < 0.35f) wd = 0.50f;
            float w = mw * sw * wd;
            float t2 = (w < 1.25f) ? w : 1.25f;
            rel[c] = (0.12f < t2) ? t2 : 0.12f;
        }
    }
    int o = 0;
    for (int ds = 0; ds < nds; ++ds)
        for (int c = 0; c < nc; ++c) {
            float phase = cph[ds * nc + c];
            float cs = scale * rel[c];
            if (bps == 2) {
                float sb0 = cs * sinf(phase), sb1 = cs * sinf(2.0f * phase);
                float a = (sb0 < 20.0f) ? sb0 : 20.0f; llr_out[o++] = (-20.0f < a) ? a : -20.0f;
                float b = (sb1 < 20.0f) ? sb1 : 20.0f; llr_out[o++] = (-20.0f < b) ? b : -20.0f;
            } else {
                float sb = cs * cosf(phase);
                float a = (sb < 20.0f) ? sb : 20.0f; llr_out[o++] = (-20.0f < a) ? a : -20.0f;
            }
        }
    /* fading indices :404-437, :705-733 */
    float cmagn[MC_MAXC];
    for (int c = 0; c < nc; ++c) cmagn[c] = (valid > 0) ? mag_sum[c] / valid : 0.0f;
    float tfi = 0.0f;
    if (valid >= 4) {
        float cvsum = 0.0f;
        int vc = 0;
        for (int c = 0; c < nc; ++c) {
            float mean = mag_sum[c] / valid;
            if (mean < 0.001f) continue;
            float mean_sq = mag_sq[c] / valid;
            float var = mean_sq - mean * mean;
            var = (0.0f < var) ? var : 0.0f;
            float cv = sqrtf(var) / mean;
            cvsum += cv;
            vc++;
        }
        tfi = (vc > 0) ? cvsum / vc : 0.0f;
    }
    float ffi = 0.0f;
    {
        float sum = 0.0f;
        for (int c = 0; c < nc; ++c) sum += cmagn[c];
        float mean = sum / nc;
        if (!(mean < 0.001f)) {
            float vs = 0.0f;
            for (int c = 0; c < nc; ++c) { float d = cmagn[c] - mean; vs += d * d; }
            ffi = sqrtf(vs / nc) / mean;
        }
    }
    aux4[0] = cfo; aux4[1] = ffi + 1.0f * tfi; aux4[2] = ffi; aux4[3] = tfi;
    free(x); free(sym_total); free(cph); free(cmag);
    return n_llr;
}

/* ChaseCacheEntry combine (chase_cache.cpp:27-88): count/decoded bookkeeping of ONE codeword slot */
int ro_chase_store(float* existing, int* combine_count, int decoded, const float* soft) {
    if (decoded) return 0;
    if (*combine_count >= 4) return 0;
    if (*combine_count == 0) { memcpy(existing, soft, sizeof(float) * 648); *combine_count = 1; return 1; }
    for (int i = 0; i < 648; ++i) existing[i] += soft[i];
    (*combine_count)++;
    return 1;
}

/* ------------------------------------------------------------------------------------------------ LTS light sync
 * OFDMChirpWaveform::detectDataSync (src/waveform/ofdm_chirp_waveform.cpp:207-384): energy gate, Hilbert-65
 * analytic signal (src/dsp/filters.cpp:266-317), one-symbol-delay autocorrelation on a step-8 grid with an
 * early exit above 0.95, +-4 refinement, burst-interleave marker from the sign of the CFO-compensated peak.
 * out4 = {detected, start_sample, correlation, burst_interleaved}
 */
int ro_detect_data_sync(const float* x, int n, float known_cfo_hz, float threshold, float* out4) {
    const int L = RO_SYM; /* 1152 */
    out4[0] = 0.f; out4[1] = 0.f; out4[2] = 0.f; out4[3] = 0.f;
    const int search_window = L * 4;
    if (n < L * 3) return 0;
    float noise_floor = 0.0f;
    int noise_samples = n / 4 < 4800 ? n / 4 : 4800;
    for (int i = 0; i < noise_samples; ++i) noise_floor += x[i] * x[i];
    noise_floor = sqrtf(noise_floor / noise_samples);
    float energy_threshold = noise_floor * 3.0f + 0.01f;
    int signal_start = 0;
    int signal_in_noise = noise_floor < 0.05f;
    if (signal_in_noise) {
        for (int i = 0; i < n - L * 2; ++i) {
            float energy = 0.0f;
            for (int j = 0; j < 64; ++j) if (i + j < n) energy += x[i + j] * x[i + j];
            energy = sqrtf(energy / 64);
            if (energy > energy_threshold) { signal_start = i; break; }
        }
    }
    enum { TAPS = 65, M = 32 };
    float coeff[TAPS];
    for (int k = 0; k < TAPS; ++k) {
        int kk = k - M;
        if (kk == 0) coeff[k] = 0;
        else if (kk % 2 != 0) coeff[k] = (float)(2.0f / (M_PI * kk));
        else coeff[k] = 0;
        float w = (float)(2.0f * M_PI * k / (TAPS - 1));
        coeff[k] *= 0.42f - 0.5f * cosf(w) + 0.08f * cosf(2.0f * w);
    }
    float* ar = (float*)malloc(sizeof(float) * (size_t)n);
    float* ai = (float*)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        float q = 0;
        for (int k = 0; k < TAPS; ++k) q += coeff[k] * ((i - k >= 0) ? x[i - k] : 0.0f);
        ar[i] = (i - M >= 0) ? x[i - M] : 0.0f;
        ai[i] = q;
    }
    float best_corr = 0.0f, bpr = 0.0f, bpi = 0.0f;
    int best_offset = 0;
    int max_connected = search_window > L * 8 ? search_window : L * 8;
    int actual = signal_in_noise ? search_window : max_connected;
    int search_end = signal_start + actual < n - L * 2 ? signal_start + actual : n - L * 2;
#define RO_LTS_CORR(offset, corr_out, pr_out, pi_out)                                              \
    do {                                                                                           \
        float pr_ = 0.0f, pi_ = 0.0f, e1_ = 0.0f, e2_ = 0.0f;                                      \
        for (int nn = 0; nn < L; ++nn) {                                                           \
            int i1 = (offset) + nn, i2 = (offset) + nn + L;                                        \
            if (i2 >= n) break;                                                                    \
            /* conj(s1) * s2 */                                                                    \
            float a = ar[i1], b = -ai[i1], c = ar[i2], d = ai[i2];                                 \
            pr_ += a * c - b * d;                                                                  \
            pi_ += a * d + b * c;                                                                  \
            e1_ += ar[i1] * ar[i1] + ai[i1] * ai[i1];                                              \
            e2_ += c * c + d * d;                                                                  \
        }                                                                                          \
        float den_ = sqrtf(e1_ * e2_) + 1e-10f;                                                    \
        corr_out = hypotf(pr_, pi_) / den_; pr_out = pr_; pi_out = pi_;                            \
    } while (0)
    for (int offset = signal_start; offset < search_end; offset += 8) {
        float corr, pr, pi;
        RO_LTS_CORR(offset, corr, pr, pi);
        if (corr > best_corr) { best_corr = corr; best_offset = offset; bpr = pr; bpi = pi; }
        if (corr > 0.95f) break;
    }
    if (best_corr > threshold) {
        int rs = signal_start > best_offset - 4 ? signal_start : best_offset - 4;
        int re = search_end < best_offset + 5 ? search_end : best_offset + 5;
        int center = best_offset;
        for (int offset = rs; offset < re; ++offset) {
            if (offset == center) continue;
            float corr, pr, pi;
            RO_LTS_CORR(offset, corr, pr, pi);
            if (corr > best_corr) { best_corr = corr; best_offset = offset; bpr = pr; bpi = pi; }
        }
    }
    free(ar); free(ai);
    out4[2] = best_corr;
    if (best_corr > threshold) {
        out4[0] = 1.f;
        out4[1] = (float)best_offset;
        float cfo_phase = (float)(2.0f * M_PI * known_cfo_hz * L / 48000.0f);
        float cr = cosf(-cfo_phase), ci = sinf(-cfo_phase);
        float mr = bpr * cr - bpi * ci;
        out4[3] = (mr < 0.0f) ? 1.f : 0.f;
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * SimulatedChannel::applyTxCFO (tools/cli_simulator.cpp:298-341): analytic-signal frequency shift of one transmission.
 * FFT = ultra::FFT of size 2^ceil(log2 n) (src/dsp/fft.cpp:83-128: twiddle k = (cosf, sinf)((float)(-2*pi*k/N)), radix-2
 * DIT, inverse = conjugated twiddles then * (1/N)). */
static void tx_cfo_fft(scf* data, int size, const scf* tw, int inverse) {
    int j = 0;
    for (int i = 0; i < size - 1; ++i) {
        if (i < j) { scf t = data[i]; data[i] = data[j]; data[j] = t; }
        int k = size / 2;
        while (k <= j) { j -= k; k /= 2; }
        j += k;
    }
    for (int len = 2; len <= size; len *= 2) {
        int half = len / 2, step = size / len;
        for (int i = 0; i < size; i += len)
            for (int k = 0; k < half; ++k) {
                scf w = tw[k * step];
                if (inverse) w.im = -w.im;
                scf d = data[i + k + half], a = data[i + k], t;
                t.re = w.re * d.re - w.im * d.im;
                t.im = w.re * d.im + w.im * d.re;
                data[i + k + half].re = a.re - t.re; data[i + k + half].im = a.im - t.im;
                data[i + k].re = a.re + t.re; data[i + k].im = a.im + t.im;
            }
    }
    if (inverse) {
        float scale = 1.0f / (float)size;
        for (int i = 0; i < size; ++i) { data[i].re *= scale; data[i].im *= scale; }
    }
}
int ro_apply_tx_cfo(const float* in, int n, float cfo_hz, float* phase_inout, float* out) {
    if (fabsf(cfo_hz) < 0.001f || n <= 0) {                              /* :299-301 */
        if (n > 0 && out != in) memcpy(out, in, sizeof(float) * (size_t)n);
        return n;
    }
    int size = 1;
    while (size < n) size <<= 1;                                         /* :304-305 */
    scf* tw = (scf*)malloc(sizeof(scf) * (size_t)(size / 2 + 1));
    for (int k = 0; k < size / 2; ++k) {
        float angle = (float)(-2.0f * M_PI * (double)k / (double)size);
        tw[k].re = cosf(angle); tw[k].im = sinf(angle);
    }
    scf* f = (scf*)calloc((size_t)size, sizeof(scf));
    for (int i = 0; i < n; ++i) f[i].re = in[i];
    tx_cfo_fft(f, size, tw, 0);
    if (size >= 2) {                                                     /* :318-325 */
        for (int i = 1; i < size / 2; ++i) { f[i].re *= 2.0f; f[i].im *= 2.0f; }
        for (int i = size / 2 + 1; i < size; ++i) { f[i].re = 0.0f; f[i].im = 0.0f; }
    }
    tx_cfo_fft(f, size, tw, 1);
    const float pi_f = (float)M_PI;
    const float phase_inc = 2.0f * pi_f * cfo_hz / 48000.0f;             /* all float (:330) */
    float phase = phase_inout ? *phase_inout : 0.0f;
    for (int i = 0; i < n; ++i) {
        float c = cosf(phase), s = sinf(phase);
        out[i] = f[i].re * c - f[i].im * s;                              /* real(analytic * rot) */
        phase += phase_inc;
        if (phase > pi_f) phase -= 2.0f * pi_f;
        else if (phase < -pi_f) phase += 2.0f * pi_f;
    }
    if (phase_inout) *phase_inout = phase;
    free(f); free(tw);
    return n;
}

/* ------------------------------------------------------------------ helper arithmetic of the reference's test programs */
void ro_tool_add_noise(float* x, int n, float snr_db, ro_mt* rng) { /* tools/test_zc_sync.cpp:22-39 */
    float sig_power = 0.0f;
    for (int i = 0; i < n; ++i) sig_power += x[i] * x[i];
    sig_power /= (float)n;                                  /* size_t -> float */
    float snr_linear = powf(10.0f, snr_db / 10.0f);
    float noise_power = sig_power / snr_linear;
    float noise_std = sqrtf(noise_power);
    ro_normal nd = {0.0f, 0};
    for (int i = 0; i < n; ++i) x[i] += ro_normal_draw(&nd, rng, 0.0f, noise_std);
}

void ro_tool_apply_cfo(float* x, int n, float cfo_hz, float sample_rate) { /* tools/test_zc_sync.cpp:43-63 */
    if (fabsf(cfo_hz) < 0.01f || n < 128) return;
    enum { TAPS = 127, M = 63 };
    float coeff[TAPS];
    for (int k = 0; k < TAPS; ++k) {                        /* filters.cpp:266-291 */
        int kk = k - M;
        if (kk == 0) coeff[k] = 0;
        else if (kk % 2 != 0) coeff[k] = (float)(2.0f / (M_PI * kk));
        else coeff[k] = 0;
        float w = (float)(2.0f * M_PI * k / (TAPS - 1));
        coeff[k] *= 0.42f - 0.5f * cosf(w) + 0.08f * cosf(2.0f * w);
    }
    float* ar = (float*)malloc(sizeof(float) * (size_t)n);
    float* ai = (float*)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) {                           /* filters.cpp:293-317: newest sample first */
        float q = 0;
        for (int k = 0; k < TAPS; ++k) q += coeff[k] * ((i - k >= 0) ? x[i - k] : 0.0f);
        ar[i] = (i - M >= 0) ? x[i - M] : 0.0f;
        ai[i] = q;
    }
    float phase = 0.0f;
    float phase_inc = (float)(2.0f * M_PI * cfo_hz / sample_rate);
    for (int i = 0; i < n; ++i) {
        float rc = cosf(phase), rs = sinf(phase);
        x[i] = ar[i] * rc - ai[i] * rs;                     /* real part of analytic * rotation */
        phase += phase_inc;
        while (phase > M_PI) phase = (float)(phase - 2.0f * M_PI);
        while (phase < -M_PI) phase = (float)(phase + 2.0f * M_PI);
    }
    free(ar); free(ai);
}

void ro_tool_chase_reception(const uint8_t* coded81, float snr_db, ro_mt* rng, float* llr648) { /* tools/test_chase_cache.cpp:21-62 */
    for (int i = 0; i < 648; ++i) llr648[i] = ((coded81[i >> 3] >> (7 - (i & 7))) & 1) ? -4.0f : 4.0f;
    float snr_linear = powf(10.0f, snr_db / 10.0f);
    float noise_std = 1.0f / sqrtf(snr_linear);
    ro_normal nd = {0.0f, 0};
    for (int i = 0; i < 648; ++i) {
        float sign = (llr648[i] > 0) ? 1.0f : -1.0f;
        float received = sign + ro_normal_draw(&nd, rng, 0.0f, noise_std);
        llr648[i] = 2.0f * received * snr_linear;
    }
}
