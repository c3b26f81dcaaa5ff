/* oracle/ria_oracle.c — CPU restatement of the RIA RX hot path (plain C11, scalar, one thread).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path, never the product (see ria_oracle.h).
 * Pinned bit-for-bit against the compiled unmodified reference (oracle/_ref, build container only)
 * by oracle/check_against_ref.py; its outputs for seeded inputs are committed under tests/golden/.
 *
 * Arithmetic follows the reference toolchain semantics measured in SURVEY.md §A.6:
 *   complex*complex  naive float formula, no FMA          (libstdc++ operator*, -O3, x86-64)
 *   complex/complex  libgcc __divsc3: straight formula in double, one rounding to float
 *   abs(complex)     hypotf;  arg() atan2f;  polar/exp(j*phi)  (cosf, sinf)
 * Build with -ffp-contract=off (oracle/Makefile).
 */
#include "ria_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct { float re, im; } cf;

static inline cf cf_mk(float re, float im) { cf z = { re, im }; return z; }
static inline cf cf_add(cf a, cf b) { return cf_mk(a.re + b.re, a.im + b.im); }
static inline cf cf_sub(cf a, cf b) { return cf_mk(a.re - b.re, a.im - b.im); }
static inline cf cf_conj(cf a) { return cf_mk(a.re, -a.im); }
static inline cf cf_mul(cf a, cf b) {
    return cf_mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
static inline cf cf_scale(float s, cf a) { return cf_mk(s * a.re, s * a.im); }
static inline cf cf_divf(cf a, float s) { return cf_mk(a.re / s, a.im / s); }
static inline cf cf_div(cf x, cf y) { /* libgcc __divsc3, wide-type path */
    double a = x.re, b = x.im, c = y.re, d = y.im;
    double denom = c * c + d * d;
    return cf_mk((float)((a * c + b * d) / denom), (float)((b * c - a * d) / denom));
}
static inline float cf_abs(cf a) { return hypotf(a.re, a.im); }
static inline float cf_norm(cf a) { return a.re * a.re + a.im * a.im; }
static inline float cf_arg(cf a) { return atan2f(a.im, a.re); }
static inline cf cf_expj(float phi) { return cf_mk(cosf(phi), sinf(phi)); }
static inline float fmaxf_(float a, float b) { return (a < b) ? b : a; } /* std::max */
static inline float fminf_(float a, float b) { return (b < a) ? b : a; } /* std::min */

/* ------------------------------------------------------------------ mt19937 / normal */
void ro_mt_seed(ro_mt* m, uint32_t seed) {
    m->s[0] = seed;
    for (int i = 1; i < 624; ++i)
        m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}
uint32_t ro_mt_next(ro_mt* m) {
    if (m->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (m->s[i] & 0x80000000u) | (m->s[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = m->s[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            m->s[i] = v;
        }
        m->idx = 0;
    }
    uint32_t y = m->s[m->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
/* libstdc++ generate_canonical<float,24>(mt19937): one draw, float(u32)/2^32, clamp below 1 */
static float ro_canonical(ro_mt* m) {
    float sum = (float)ro_mt_next(m);
    float ret = sum / 4294967296.0f;
    if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
    return ret;
}
/* libstdc++ normal_distribution<float>::operator(): Marsaglia polar, caches the second value */
float ro_normal_draw(ro_normal* n, ro_mt* m, float mean, float stddev) {
    float ret;
    if (n->has_saved) {
        n->has_saved = 0;
        ret = n->saved;
    } else {
        float x, y, r2;
        do {
            x = 2.0f * ro_canonical(m) - 1.0f;
            y = 2.0f * ro_canonical(m) - 1.0f;
            r2 = x * x + y * y;
        } while (r2 > 1.0f || r2 == 0.0f);
        float mult = sqrtf(-2.0f * logf(r2) / r2);
        n->saved = x * mult;
        n->has_saved = 1;
        ret = y * mult;
    }
    return ret * stddev + mean;
}

/* ------------------------------------------------------------------ geometry */
static int ro_bits_per_carrier(int mod) { /* types.hpp:42-56 */
    switch (mod) {
        case RO_DBPSK: case RO_BPSK: return 1;
        case RO_DQPSK: case RO_QPSK: return 2;
        case RO_D8PSK: case RO_QAM8: return 3;
        case RO_QAM16: return 4;
        case RO_QAM32: return 5;
        case RO_QAM64: return 6;
        case RO_QAM256: return 8;
        default: return 1;
    }
}
static int ro_is_coherent(int mod) {
    return !(mod == RO_DBPSK || mod == RO_DQPSK || mod == RO_D8PSK);
}
static int ro_pilot_spacing(int mod, int rate) { /* ofdm_link_adaptation.hpp:26-70 */
    if (ro_is_coherent(mod)) {
        switch (rate) {
            case RO_R5_6: case RO_R7_8: return 6;
            case RO_R3_4: return 8;
            default: return 5;
        }
    }
    if (mod == RO_D8PSK) {
        switch (rate) {
            case RO_R3_4: case RO_R2_3: case RO_R1_2: return 8;
            default: return 10;
        }
    }
    return (rate == RO_R3_4) ? 15 : 10;
}
static int ro_info_bits(int rate) { /* frame_v2.hpp:671-681 */
    switch (rate) {
        case RO_R1_4: return 162;
        case RO_R1_3: return 216;
        case RO_R1_2: return 324;
        case RO_R2_3: return 432;
        case RO_R3_4: return 486;
        case RO_R5_6: return 540;
        default: return 162;
    }
}
static int ro_recommended_iters(int rate) { /* ldpc_codec.hpp:86-95 */
    switch (rate) {
        case RO_R3_4: return 60;
        case RO_R2_3: return 70;
        case RO_R1_2: return 80;
        case RO_R1_3: return 60;
        default: return 50;
    }
}

int ro_geom_init(ro_geom* g, int mod, int rate) {
    memset(g, 0, sizeof(*g));
    g->mod = mod;
    g->rate = rate;
    g->pilot_spacing = ro_pilot_spacing(mod, rate);
    /* demodulator.cpp:45-76 setupCarriers */
    int neg = RO_NCAR / 2, pos = (RO_NCAR + 1) / 2, l = 0;
    for (int i = -neg; i <= pos; ++i) {
        if (i == 0) continue;
        int bin = (i + RO_FFT) % RO_FFT;
        int isp = (l % g->pilot_spacing) == 0;
        g->all_idx[l] = bin;
        g->is_pilot[l] = isp;
        if (isp) { g->pilot_idx[g->n_pilot] = bin; g->pilot_logical[g->n_pilot++] = l; }
        else     { g->data_idx[g->n_data] = bin;   g->data_logical[g->n_data++] = l; }
        ++l;
    }
    /* demodulator.cpp:78-93 generateSequences */
    for (int n = 0; n < RO_NCAR; ++n) {
        float phase = (float)(-M_PI * 1.0 * (double)n * (double)(n + 1) / (double)RO_NCAR);
        g->sync_re[n] = cosf(phase);
        g->sync_im[n] = sinf(phase);
    }
    ro_mt rng;
    ro_mt_seed(&rng, 0x50494C54u);
    for (int p = 0; p < g->n_pilot; ++p) g->pilot_seq[p] = (ro_mt_next(&rng) & 1u) ? 1.0f : -1.0f;
    /* demodulator.cpp:145-202 buildInterpTable */
    int d = 0;
    for (int ci = 0; ci < RO_NCAR; ++ci) {
        if (g->is_pilot[ci]) continue;
        int lo = -1, hi = -1;
        for (int j = ci - 1; j >= 0; --j) if (g->is_pilot[j]) { lo = j; break; }
        for (int j = ci + 1; j < RO_NCAR; ++j) if (g->is_pilot[j]) { hi = j; break; }
        float alpha = 0.5f;
        if (lo >= 0 && hi >= 0) {
            float total = (float)(hi - lo);
            alpha = (total > 0) ? (float)(ci - lo) / total : 0.5f;
        }
        g->interp_lo[d] = (lo >= 0) ? lo / g->pilot_spacing : -1; /* pilot ordinal */
        g->interp_hi[d] = (hi >= 0) ? hi / g->pilot_spacing : -1;
        g->interp_alpha[d] = alpha;
        ++d;
    }
    g->bits_per_carrier = ro_bits_per_carrier(mod);
    g->bits_per_symbol = g->n_data * g->bits_per_carrier;
    g->n_data_symbols = (RO_FRAME_BITS + g->bits_per_symbol - 1) / g->bits_per_symbol;
    g->frame_samples = (2 + g->n_data_symbols) * RO_SYM;
    g->n_llr = g->n_data_symbols * g->bits_per_symbol;
    g->info_bits = ro_info_bits(rate);
    g->bytes_per_cw = g->info_bits / 8;
    g->max_iter = ro_recommended_iters(rate);
    return 0;
}

/* ------------------------------------------------------------------ FFT (src/dsp/fft.cpp:83-128) */
static cf g_tw[RO_FFT / 2];
static int g_tw_ready = 0;
static void ro_fft_init(void) {
    if (g_tw_ready) return;
    for (int k = 0; k < RO_FFT / 2; ++k) {
        float angle = (float)(-2.0f * M_PI * (double)k / (double)RO_FFT);
        g_tw[k] = cf_mk(cosf(angle), sinf(angle));
    }
    g_tw_ready = 1;
}
static void ro_fft(cf* data, int inverse) {
    ro_fft_init();
    int size = RO_FFT, j = 0;
    for (int i = 0; i < size - 1; ++i) {
        if (i < j) { cf t = data[i]; data[i] = data[j]; data[j] = t; }
        int k = size / 2;
        while (k <= j) { j -= k; k /= 2; }
        j += k;
    }
    for (int len = 2; len <= size; len *= 2) {
        int half = len / 2, step = size / len;
        for (int i = 0; i < size; i += len)
            for (int k = 0; k < half; ++k) {
                cf w = g_tw[k * step];
                if (inverse) w = cf_conj(w);
                cf t = cf_mul(w, data[i + k + half]);
                data[i + k + half] = cf_sub(data[i + k], t);
                data[i + k] = cf_add(data[i + k], t);
            }
    }
    if (inverse) {
        float scale = 1.0f / (float)size;
        for (int i = 0; i < size; ++i) data[i] = cf_mk(data[i].re * scale, data[i].im * scale);
    }
}

/* ------------------------------------------------------------------ NCO (src/dsp/filters.cpp:228-238) */
typedef struct { float phase, inc; } ro_nco;
static void ro_nco_init(ro_nco* n, float freq, float fs) {
    n->phase = 0.0f;
    n->inc = (float)(2.0f * M_PI * (double)freq / (double)fs);
}
static cf ro_nco_next(ro_nco* n) {
    cf out = cf_mk(cosf(n->phase), sinf(n->phase));
    n->phase += n->inc;
    if ((double)n->phase > 2.0f * M_PI) n->phase = (float)((double)n->phase - 2.0f * M_PI);
    if (n->phase < 0) n->phase = (float)((double)n->phase + 2.0f * M_PI);
    return out;
}

/* ------------------------------------------------------------------ LDPC */
static void ro_code_params(int rate, int* k, int* m) { /* ldpc_decoder.cpp:21-36 */
    switch (rate) {
        case RO_R1_4: *k = 162; *m = 486; break;
        case RO_R1_2: *k = 324; *m = 324; break;
        case RO_R2_3: *k = 432; *m = 216; break;
        case RO_R3_4: *k = 486; *m = 162; break;
        case RO_R5_6: *k = 540; *m = 108; break;
        default: *k = 324; *m = 324; break;
    }
}

int ro_ldpc_build(ro_ldpc* c, int rate) { /* ldpc_decoder.cpp:65-138 */
    int k, m;
    ro_code_params(rate, &k, &m);
    c->rate = rate; c->k = k; c->m = m; c->n = k + m;
    ro_mt rng;
    ro_mt_seed(&rng, (uint32_t)(0x12345678 + rate));
    int target_var = (4 * m) / k;
    if (target_var < 3) target_var = 3;
    if (target_var > m / 2) target_var = m / 2;
    const int max_check = 6;
    static int rows[RO_CW_BITS][16];
    int deg[RO_CW_BITS];
    int avail[RO_CW_BITS];
    memset(deg, 0, sizeof(deg));
    for (int j = 0; j < k; ++j) {
        int na = 0;
        for (int i = 0; i < m; ++i) if (deg[i] < max_check) avail[na++] = i;
        for (int i = na; i > 1; --i) {
            int r = (int)(ro_mt_next(&rng) % (uint32_t)i);
            int t = avail[i - 1]; avail[i - 1] = avail[r]; avail[r] = t;
        }
        int conn = target_var < na ? target_var : na;
        for (int d = 0; d < conn; ++d) { int chk = avail[d]; rows[chk][deg[chk]++] = j; }
    }
    for (int i = 0; i < m; ++i)
        if (deg[i] == 0) rows[i][deg[i]++] = (int)(ro_mt_next(&rng) % (uint32_t)k);
    int e = 0;
    for (int i = 0; i < m; ++i) {
        c->row_ptr[i] = e;
        for (int d = 0; d < deg[i]; ++d) c->edge_var[e++] = rows[i][d];
        c->edge_var[e++] = k + i;
    }
    c->row_ptr[m] = e;
    c->n_edges = e;
    return e;
}

int ro_ldpc_encode(const ro_ldpc* c, const uint8_t* info, int n_info_bytes, uint8_t* coded81) {
    /* ldpc_encoder.cpp:193-257, single block (n_info_bytes*8 <= k + 7) */
    uint8_t bits[RO_CW_BITS];
    memset(bits, 0, sizeof(bits));
    for (int j = 0; j < c->k && j < n_info_bytes * 8; ++j) bits[j] = (info[j / 8] >> (7 - (j % 8))) & 1;
    for (int i = 0; i < c->m; ++i) {
        uint8_t s = 0;
        for (int e = c->row_ptr[i]; e < c->row_ptr[i + 1] - 1; ++e) s ^= bits[c->edge_var[e]];
        bits[c->k + i] = s;
    }
    memset(coded81, 0, 81);
    for (int j = 0; j < c->n; ++j) if (bits[j]) coded81[j / 8] |= (uint8_t)(1u << (7 - (j % 8)));
    return 81;
}

int ro_ldpc_decode(const ro_ldpc* c, const float* llr, int n_llr, int max_iter, float factor,
                   uint8_t* out, int* iters) { /* ldpc_decoder.cpp:154-260 */
    int n = c->n, k = c->k, m = c->m;
    float llr_in[RO_CW_BITS], tot[RO_CW_BITS];
    static _Thread_local float v2c[RO_MAX_EDGES], c2v[RO_MAX_EDGES];
    for (int j = 0; j < n; ++j) { llr_in[j] = (j < n_llr) ? llr[j] : 0.0f; tot[j] = llr_in[j]; }
    for (int e = 0; e < c->n_edges; ++e) { v2c[e] = llr_in[c->edge_var[e]]; c2v[e] = 0.0f; }
    int success = 0, it;
    for (it = 0; it < max_iter; ++it) {
        for (int i = 0; i < m; ++i) {
            int e0 = c->row_ptr[i], e1 = c->row_ptr[i + 1];
            for (int e = e0; e < e1; ++e) {
                float sign = 1.0f, min_abs = 3.402823466e+38f;
                for (int e2 = e0; e2 < e1; ++e2) {
                    if (e2 == e) continue;
                    float msg = v2c[e2];
                    if (msg < 0) sign = -sign;
                    float a = fabsf(msg);
                    if (a < min_abs) min_abs = a;
                }
                c2v[e] = sign * min_abs * factor;
            }
        }
        for (int j = 0; j < n; ++j) tot[j] = llr_in[j];
        for (int e = 0; e < c->n_edges; ++e) tot[c->edge_var[e]] += c2v[e];
        for (int e = 0; e < c->n_edges; ++e) {
            float v = tot[c->edge_var[e]] - c2v[e];
            v2c[e] = fmaxf_(-50.0f, fminf_(50.0f, v));
        }
        int ok = 1;
        for (int i = 0; i < m && ok; ++i) {
            int s = 0;
            for (int e = c->row_ptr[i]; e < c->row_ptr[i + 1]; ++e) s ^= (tot[c->edge_var[e]] < 0);
            if (s) ok = 0;
        }
        if (ok) { success = 1; break; }
    }
    if (iters) *iters = it;
    int nb = (k + 7) / 8;
    memset(out, 0, (size_t)nb);
    for (int j = 0; j < k; ++j) if (tot[j] < 0) out[j / 8] |= (uint8_t)(1u << (7 - (j % 8)));
    return success;
}

/* ------------------------------------------------------------------ interleavers */
static int gcd_i(int a, int b) { while (b) { int t = b; b = a % b; a = t; } return a; }
int ro_channel_interleaver_step(int n, int total) { /* ldpc_decoder.cpp:552-577 */
    int target = n * 3;
    if (target >= total) target = total / 2;
    for (int s = target; s < total; ++s) if (gcd_i(s, total) == 1) return s;
    for (int s = n + 1; s < total; ++s) if (gcd_i(s, total) == 1) return s;
    return n + 1;
}
/* table[cw*648 + i] = index into the 2592 interleaved soft bits feeding decoder input i of cw
 * (frame_interleaver.cpp:37-45 inverted, then ldpc_decoder.cpp:598-625). */
void ro_rx_gather_table(int bps, int use_channel, int* table) {
    int step = use_channel ? ro_channel_interleaver_step(bps, RO_CW_BITS) : 1;
    for (int cw = 0; cw < 4; ++cw)
        for (int i = 0; i < RO_CW_BITS; ++i) {
            int bit = use_channel ? (int)(((long)i * step) % RO_CW_BITS) : i;
            table[cw * RO_CW_BITS + i] = bit * 4 + (cw + bit) % 4;
        }
}

/* ------------------------------------------------------------------ frame build */
uint16_t ro_crc16(const uint8_t* d, int n) { /* frame_v2.cpp:115-128 */
    uint16_t crc = 0xFFFF;
    for (int i = 0; i < n; ++i) {
        crc ^= (uint16_t)((uint16_t)d[i] << 8);
        for (int j = 0; j < 8; ++j) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
    }
    return crc;
}
static uint32_t ro_hash_callsign(const char* s) { /* frame_v2.cpp:78-84 */
    uint32_t h = 5381;
    for (; *s; ++s) {
        int c = *s;
        if (c >= 'a' && c <= 'z') c -= 32;
        h = ((h << 5) + h) ^ (uint8_t)c;
    }
    return h & 0xFFFFFF;
}
/* makeFixedDataFrame("TEST","RX",seq,payload,rate).serialize(), zero padded to 4*bytes_per_cw
 * (frame_v2.cpp:1890-1912, :502-554, :1291-1297) */
int ro_make_frame(const uint8_t* payload, int payload_len, int seq, int rate, uint8_t* out) {
    int total = 4 * (ro_info_bits(rate) / 8);
    int cap = total - 19;
    if (payload_len > cap) payload_len = cap;
    memset(out, 0, (size_t)total);
    uint32_t src = ro_hash_callsign("TEST"), dst = ro_hash_callsign("RX");
    out[0] = 0x55; out[1] = 0x4C; out[2] = 0x30; out[3] = 0x01;
    out[4] = (uint8_t)(seq >> 8); out[5] = (uint8_t)seq;
    out[6] = (uint8_t)(src >> 16); out[7] = (uint8_t)(src >> 8); out[8] = (uint8_t)src;
    out[9] = (uint8_t)(dst >> 16); out[10] = (uint8_t)(dst >> 8); out[11] = (uint8_t)dst;
    out[12] = 4;
    out[13] = (uint8_t)(payload_len >> 8); out[14] = (uint8_t)payload_len;
    uint16_t h = ro_crc16(out, 15);
    out[15] = (uint8_t)(h >> 8); out[16] = (uint8_t)h;
    memcpy(out + 17, payload, (size_t)payload_len);
    int sz = 17 + payload_len + 2;
    uint16_t f = ro_crc16(out, sz - 2);
    out[sz - 2] = (uint8_t)(f >> 8); out[sz - 1] = (uint8_t)f;
    return total;
}
int ro_encode_fixed_frame(const uint8_t* info, int n_info, int rate, int ch_il, int bps,
                          uint8_t* coded324) { /* frame_v2.cpp:1285-1328 */
    static _Thread_local ro_ldpc code;
    if (code.rate != rate || code.n == 0) ro_ldpc_build(&code, rate);
    int bpc = ro_info_bits(rate) / 8, total = 4 * bpc;
    uint8_t padded[4 * 68];
    memset(padded, 0, sizeof(padded));
    memcpy(padded, info, (size_t)(n_info < total ? n_info : total));
    int step = ch_il ? ro_channel_interleaver_step(bps, RO_CW_BITS) : 1;
    uint8_t bits[RO_FRAME_BITS];
    for (int cw = 0; cw < 4; ++cw) {
        uint8_t coded[81];
        ro_ldpc_encode(&code, padded + cw * bpc, bpc, coded);
        for (int i = 0; i < RO_CW_BITS; ++i) {
            int b = (coded[i / 8] >> (7 - (i % 8))) & 1;
            int pos = ch_il ? (int)(((long)i * step) % RO_CW_BITS) : i; /* out[perm[i]] = in[i] */
            bits[pos * 4 + (cw + pos) % 4] = (uint8_t)b;
        }
    }
    memset(coded324, 0, 324);
    for (int i = 0; i < RO_FRAME_BITS; ++i) if (bits[i]) coded324[i / 8] |= (uint8_t)(1u << (7 - (i % 8)));
    return 324;
}

/* ------------------------------------------------------------------ TX modulator */
static cf ro_map_bits(uint32_t bits, int mod) { /* modulator.cpp:27-116 */
    switch (mod) {
        case RO_BPSK: return cf_mk((bits & 1) ? 1.0f : -1.0f, 0.0f);
        case RO_QAM16: {
            static const float lv[] = { -3, -1, 3, 1 };
            const float s = 0.3162277660168379f;
            return cf_mk(lv[(bits >> 2) & 3] * s, lv[bits & 3] * s);
        }
        case RO_QAM32: {
            const float s = 0.1961161351381840f;
            static const float IL[4] = { -3, -1, 1, 3 };
            static const int IG[4] = { 0, 1, 3, 2 };
            static const float QL[8] = { -7, -5, -3, -1, 1, 3, 5, 7 };
            static const int QG[8] = { 0, 1, 3, 2, 6, 7, 5, 4 };
            int qb = (bits >> 2) & 7, ib = bits & 3, qi = 0, ii = 0;
            for (int i = 0; i < 4; ++i) if (IG[i] == ib) { ii = i; break; }
            for (int i = 0; i < 8; ++i) if (QG[i] == qb) { qi = i; break; }
            return cf_mk(IL[ii] * s, QL[qi] * s);
        }
        case RO_QAM64: {
            static const float lv[] = { -7, -5, -1, -3, 7, 5, 1, 3 };
            const float s = 0.1543033499620919f;
            return cf_mk(lv[(bits >> 3) & 7] * s, lv[bits & 7] * s);
        }
        case RO_QAM256: {
            static const float lv[] = { -15, -13, -9, -11, -1, -3, -7, -5, 15, 13, 9, 11, 1, 3, 7, 5 };
            const float s = 0.0645497224367903f;
            return cf_mk(lv[(bits >> 4) & 15] * s, lv[bits & 15] * s);
        }
        case RO_QPSK:
        default: {
            const float s = 0.7071067811865476f;
            return cf_mk((bits & 2) ? s : -s, (bits & 1) ? s : -s);
        }
    }
}

static void ro_emit_symbol(const ro_geom* g, const cf* data_syms, ro_nco* mixer, float* out) {
    /* modulator.cpp:217-283 createOFDMSymbol + complexToReal (output_scale 40) */
    cf fd[RO_FFT];
    memset(fd, 0, sizeof(fd));
    for (int i = 0; i < g->n_data; ++i) fd[g->data_idx[i]] = data_syms[i];
    for (int p = 0; p < g->n_pilot; ++p) fd[g->pilot_idx[p]] = cf_mk(g->pilot_seq[p], 0.0f);
    ro_fft(fd, 1);
    for (int i = 0; i < RO_SYM; ++i) {
        cf s = (i < RO_CP) ? fd[RO_FFT - RO_CP + i] : fd[i - RO_CP];
        cf mixed = cf_mul(s, ro_nco_next(mixer));
        out[i] = mixed.re * 40.0f;
    }
}

int ro_modulate(const ro_geom* g, const uint8_t* coded, int n_coded, float* samples, int max_samples) {
    /* generateTrainingSymbols(2) ++ modulate(coded): modulator.cpp:534-583, :348-477 */
    ro_nco mixer;
    ro_nco_init(&mixer, 1500.0f, 48000.0f);
    cf syms[RO_NCAR], prev[RO_NCAR];
    int n_out = 0;
    for (int i = 0; i < g->n_data; ++i) { syms[i] = cf_mk(g->sync_re[i % RO_NCAR], g->sync_im[i % RO_NCAR]); prev[i] = cf_mk(1, 0); }
    for (int t = 0; t < 2; ++t) {
        if (n_out + RO_SYM > max_samples) return -1;
        ro_emit_symbol(g, syms, &mixer, samples + n_out);
        n_out += RO_SYM;
    }
    int data_idx = 0, bit_idx = 0;
    while (data_idx < n_coded) {
        int c;
        for (c = 0; c < g->n_data && data_idx < n_coded; ++c) {
            uint32_t bits = 0;
            for (int b = 0; b < g->bits_per_carrier; ++b) {
                bits <<= 1;
                if (data_idx < n_coded) {
                    bits |= (coded[data_idx] >> (7 - bit_idx)) & 1u;
                    if (++bit_idx >= 8) { bit_idx = 0; ++data_idx; }
                }
            }
            if (g->mod == RO_DBPSK) {
                cf pc = (bits & 1) ? cf_mk(-1, 0) : cf_mk(1, 0);
                prev[c] = cf_mul(prev[c], pc);
                syms[c] = prev[c];
            } else if (g->mod == RO_DQPSK) {
                static const float pr[4] = { 1, 0, -1, 0 }, pi[4] = { 0, 1, 0, -1 };
                prev[c] = cf_mul(prev[c], cf_mk(pr[bits & 3], pi[bits & 3]));
                syms[c] = prev[c];
            } else if (g->mod == RO_D8PSK) {
                const float pi_f = 3.14159265358979f;
                float angle = (float)(bits & 7) * (pi_f / 4.0f) + pi_f / 8.0f;
                prev[c] = cf_mul(prev[c], cf_mk(cosf(angle), sinf(angle)));
                syms[c] = prev[c];
            } else {
                syms[c] = ro_map_bits(bits, g->mod);
            }
        }
        for (; c < g->n_data; ++c) syms[c] = cf_mk(0, 0);
        if (n_out + RO_SYM > max_samples) return -1;
        ro_emit_symbol(g, syms, &mixer, samples + n_out);
        n_out += RO_SYM;
    }
    return n_out;
}

/* ------------------------------------------------------------------ channel */
/* WattersonChannel::applyCFO (hf_channel.hpp:182-241): mix to baseband at 1500 Hz, 48-tap running-sum moving average,
 * rotate by the CFO phase, mix back.  The mixer phase is 2*pi*fc*t evaluated in double and rounded to float; the running
 * sums add the new and subtract the old sample in float; the CFO phase only wraps downwards (positive offsets). */
static void ro_apply_cfo(float* s, int n, float phase_inc) {
    if (n < 256) return;
    const float fc = 1500.0f, fs = (float)48000u;
    float* Ib = (float*)malloc(sizeof(float) * (size_t)n); float* Qb = (float*)malloc(sizeof(float) * (size_t)n);
    float* If = (float*)malloc(sizeof(float) * (size_t)n); float* Qf = (float*)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        float t = (float)i / fs;
        float mp = (float)(2.0f * M_PI * (double)fc * (double)t);
        Ib[i] = s[i] * cosf(mp);
        Qb[i] = s[i] * sinf(mp);
    }
    const int win = 48;
    float Is = 0, Qs = 0;
    for (int i = 0; i < n; ++i) {
        Is += Ib[i]; Qs += Qb[i];
        if (i >= win) { Is -= Ib[i - win]; Qs -= Qb[i - win]; }
        int m = (i + 1 < win) ? i + 1 : win;
        If[i] = Is / (float)m;      /* float / size_t: the count is converted to float */
        Qf[i] = Qs / (float)m;
    }
    float phase = 0.0f;             /* cfo_phase_ of a fresh channel object */
    for (int i = 0; i < n; ++i) {
        float t = (float)i / fs;
        float mp = (float)(2.0f * M_PI * (double)fc * (double)t);
        float cc = cosf(phase), cs = sinf(phase);
        float Ic = If[i] * cc - Qf[i] * cs;
        float Qc = If[i] * cs + Qf[i] * cc;
        s[i] = 2.0f * (Ic * cosf(mp) - Qc * sinf(mp));
        phase += phase_inc;
        if ((double)phase > 2.0f * M_PI) phase = (float)((double)phase - 2.0f * M_PI);
    }
    free(Ib); free(Qb); free(If); free(Qf);
}

int ro_channel(int kind, float snr_db, uint32_t seed, const float* in, int n, float* out) {
    return ro_channel_cfo(kind, snr_db, seed, 0.0f, 0.0f, in, n, out, NULL);
}

int ro_channel_cfo(int kind, float snr_db, uint32_t seed, float cfo_hz, float random_cfo_max_hz, const float* in, int n, float* out,
                   float* actual_cfo_out) {
    /* hf_channel.hpp:68-103 ctor, :107-177 process, :267-284 updateFading, :182-241 applyCFO, presets :411-488 */
    float delay_ms = 0, doppler = 0, g1 = 1.0f, g2 = 0.0f;
    int fading = 1, multipath = 1;
    switch (kind) {
        case 0: fading = 0; multipath = 0; break;
        case 1: delay_ms = 0.5f; doppler = 0.1f; g1 = g2 = 0.707f; break;
        case 2: delay_ms = 1.0f; doppler = 0.5f; g1 = g2 = 0.707f; break;
        case 3: delay_ms = 2.0f; doppler = 1.0f; g1 = g2 = 0.707f; break;
        default: delay_ms = 0.5f; doppler = 10.0f; g1 = g2 = 0.707f; break;
    }
    ro_mt rng;
    ro_mt_seed(&rng, seed);
    ro_normal gs = { 0, 0 };
    /* CFO of this channel object (:97-102): a uniform_real_distribution<float>(-max, max) draw from the SAME generator,
     * before any noise is drawn (libstdc++: (b - a) * generate_canonical<float,24>(rng) + a) */
    float actual_cfo = cfo_hz;
    if (random_cfo_max_hz > 0.0f) actual_cfo = (random_cfo_max_hz - (-random_cfo_max_hz)) * ro_canonical(&rng) + (-random_cfo_max_hz);
    float cfo_inc = (float)(2.0f * M_PI * (double)actual_cfo / (double)48000u);
    if (actual_cfo_out) *actual_cfo_out = actual_cfo;
    int delay = (int)(delay_ms * 48000 / 1000.0f);
    float norm_dopp = doppler / 48000;
    float alpha = (float)(1.0f - exp(-2.0f * M_PI * (double)norm_dopp));
    cf f1 = cf_mk(1, 0), f2 = cf_mk(1, 0);
    float* dl = (float*)calloc((size_t)delay + 1, sizeof(float));
    int dl_head = 0, dl_len = delay + 1;

    float power = 0.0f;
    size_t cnt = 0;
    for (int i = 0; i < n; ++i) if (fabsf(in[i]) > 1e-6f) { power += in[i] * in[i]; cnt++; }
    float rms = cnt ? sqrtf(power / (float)cnt) : 0.1f;
    float nstd = rms * powf(10.0f, -snr_db / 20.0f);

    for (int i = 0; i < n; ++i) {
        float s = in[i];
        if (fading) {
            float ns = sqrtf(1.0f / alpha);
            /* complex<float> noise1(ns*g(), ns*g()): gcc evaluates constructor arguments right to
             * left, so the imaginary part takes the first draw (checked against oracle/_ref). */
            float n1i = ns * ro_normal_draw(&gs, &rng, 0.0f, 1.0f);
            float n1r = ns * ro_normal_draw(&gs, &rng, 0.0f, 1.0f);
            float n2i = ns * ro_normal_draw(&gs, &rng, 0.0f, 1.0f);
            float n2r = ns * ro_normal_draw(&gs, &rng, 0.0f, 1.0f);
            f1 = cf_add(cf_scale(1.0f - alpha, f1), cf_scale(alpha, cf_mk(n1r, n1i)));
            f2 = cf_add(cf_scale(1.0f - alpha, f2), cf_scale(alpha, cf_mk(n2r, n2i)));
        }
        float o = 0.0f;
        if (multipath && delay > 0) {
            float h1 = fading ? cf_abs(f1) : 1.0f, h2 = fading ? cf_abs(f2) : 1.0f;
            o += s * g1 * h1;
            float delayed = dl[dl_head];
            dl[dl_head] = s; /* pop_front + push_back on a deque of delay+1 entries */
            dl_head = (dl_head + 1) % dl_len;
            o += delayed * g2 * h2;
        } else {
            float h = fading ? cf_abs(f1) : 1.0f;
            o = s * h;
        }
        o += nstd * ro_normal_draw(&gs, &rng, 0.0f, 1.0f);
        out[i] = o;
    }
    free(dl);
    if (fabsf(actual_cfo) > 0.001f) ro_apply_cfo(out, n, cfo_inc);   /* :172-174 (cfo_enabled defaults to true) */
    return n;
}

/* ------------------------------------------------------------------ RX demodulator */
typedef struct {
    const ro_geom* g;
    ro_nco mixer;
    float cfo_hz, corr_phase;
    cf H[RO_FFT];
    float noise_var, snr_lin, fading_index, slope;
    int snr_count;
    cf prev_pilot[RO_NCAR];
    int have_prev_pilot;
    float dd[RO_NCAR];
    int have_dd;
    float cnv[RO_NCAR], ema[RO_NCAR], var[RO_NCAR];
    int have_ema;
    cf carrier_phase_corr;
    int carrier_phase_init;
    cf dprev[RO_NCAR];
    int have_dprev;
    int n_soft;
} ro_rx;

static void ro_to_baseband(ro_rx* r, const float* x, cf* bb) { /* channel_equalizer.cpp:99-173 */
    float inc = (float)(-2.0f * M_PI * (double)r->cfo_hz / (double)48000u);
    for (int i = 0; i < RO_SYM; ++i) {
        cf osc = ro_nco_next(&r->mixer);
        cf mixed = cf_mk(x[i] * osc.re, x[i] * -osc.im);
        if (fabsf(r->cfo_hz) > 0.01f) {
            mixed = cf_mul(mixed, cf_mk(cosf(r->corr_phase), sinf(r->corr_phase)));
            r->corr_phase += inc;
            if ((double)r->corr_phase > M_PI) r->corr_phase = (float)((double)r->corr_phase - 2.0f * M_PI);
            else if ((double)r->corr_phase < -M_PI) r->corr_phase = (float)((double)r->corr_phase + 2.0f * M_PI);
        }
        bb[i] = mixed;
    }
}
static void ro_symbol_fft(ro_rx* r, const float* x, cf* fd) { /* toBaseband + extractSymbol :175-187 */
    cf bb[RO_SYM];
    ro_to_baseband(r, x, bb);
    memcpy(fd, bb + RO_CP, sizeof(cf) * RO_FFT);
    ro_fft(fd, 0);
}

static void ro_lts_pass(ro_rx* r, const float* x, cf hd[2][RO_NCAR], cf* hp_last) {
    const ro_geom* g = r->g;
    cf fd[RO_FFT];
    for (int s = 0; s < 2; ++s) {
        ro_symbol_fft(r, x + s * RO_SYM, fd);
        for (int i = 0; i < g->n_data; ++i) {
            cf tx = cf_mk(g->sync_re[i % RO_NCAR], g->sync_im[i % RO_NCAR]);
            if (cf_abs(tx) > 0.01f) hd[s][i] = cf_div(fd[g->data_idx[i]], tx);
        }
        for (int p = 0; p < g->n_pilot; ++p) {
            cf tx = cf_mk(g->pilot_seq[p], 0.0f);
            if (cf_abs(tx) > 0.01f) hp_last[p] = cf_div(fd[g->pilot_idx[p]], tx);
        }
    }
}

static void ro_estimate_lts(ro_rx* r, const float* x) { /* channel_equalizer.cpp:193-643 */
    const ro_geom* g = r->g;
    cf hd[2][RO_NCAR], hp[RO_NCAR];
    memset(hd, 0, sizeof(hd));
    memset(hp, 0, sizeof(hp));
    float phase0 = r->corr_phase;
    ro_lts_pass(r, x, hd, hp);

    cf sum = cf_mk(0, 0);
    int valid = 0;
    for (int i = 0; i < g->n_data; ++i) {
        cf h0 = hd[0][i], h1 = hd[1][i];
        if (cf_abs(h0) > 0.01f && cf_abs(h1) > 0.01f) {
            cf diff = cf_mul(h1, cf_conj(h0));
            float mag = cf_abs(diff);
            if (mag > 1e-6f) { sum = cf_add(sum, cf_divf(diff, mag)); valid++; }
        }
    }
    if (valid > 10) {
        float avg = atan2f(sum.im, sum.re);
        float dur = (float)RO_SYM / (float)48000u;
        float res = (float)((double)avg / (2.0f * M_PI * (double)dur));
        if (fabsf(res) > 0.3f && fabsf(res) < 5.0f) {
            r->cfo_hz += res;
            r->mixer.phase = 0.0f;
            r->corr_phase = phase0;
            memset(hd, 0, sizeof(hd));
            ro_lts_pass(r, x, hd, hp);
        }
    }
    for (int i = 0; i < g->n_data; ++i) r->H[g->data_idx[i]] = hd[1][i];
    for (int p = 0; p < g->n_pilot; ++p) r->H[g->pilot_idx[p]] = hp[p];

    cf ss = cf_mk(0, 0);
    int sc = 0;
    for (int i = 0; i + 1 < RO_NCAR; ++i) {
        cf h0 = r->H[g->all_idx[i]], h1 = r->H[g->all_idx[i + 1]];
        if (cf_abs(h0) > 0.01f && cf_abs(h1) > 0.01f) {
            cf diff = cf_mul(h1, cf_conj(h0));
            float mag = cf_abs(diff);
            if (mag > 1e-6f) { ss = cf_add(ss, cf_divf(diff, mag)); sc++; }
        }
    }
    if (sc > 0) r->slope = cf_arg(cf_divf(ss, (float)sc));

    float noise_sum = 0, signal_sum = 0;
    int count = 0;
    for (int i = 0; i < g->n_data; ++i) {
        cf h0 = hd[0][i], h1 = hd[1][i];
        if (cf_abs(h0) > 1e-6f && cf_abs(h1) > 1e-6f) {
            noise_sum += cf_norm(cf_sub(h1, h0));
            signal_sum += (cf_norm(h0) + cf_norm(h1)) / 2.0f;
            count++;
        }
    }
    if (count > 0) {
        float nv = noise_sum / (4.0f * (float)count);
        float sp = signal_sum / (float)count;
        float snr = sp / fmaxf_(nv, 1e-10f);
        snr = fmaxf_(3.16f, fminf_(10000.0f, snr));
        r->noise_var = nv;
        r->snr_lin = snr;
    }
    float mean = 0;
    for (int i = 0; i < g->n_data; ++i) mean += cf_abs(r->H[g->data_idx[i]]);
    mean /= (float)g->n_data;
    float var = 0;
    for (int i = 0; i < g->n_data; ++i) { float d = cf_abs(r->H[g->data_idx[i]]) - mean; var += d * d; }
    var /= (float)g->n_data;
    r->fading_index = (mean > 0.01f) ? sqrtf(var) / mean : 0.0f;
    r->snr_count = 2;
}

static void ro_update_channel(ro_rx* r, const cf* fd) { /* channel_equalizer.cpp:645-1043 */
    const ro_geom* g = r->g;
    int diff_mode = !ro_is_coherent(g->mod);
    int first = (r->n_soft == 0);
    float alpha = first ? 1.0f : (diff_mode ? 0.5f : 0.9f);
    int np = g->n_pilot;
    cf hls[RO_NCAR];
    cf hsum = cf_mk(0, 0);
    for (int p = 0; p < np; ++p) {
        hls[p] = cf_div(fd[g->pilot_idx[p]], cf_mk(g->pilot_seq[p], 0.0f));
        hsum = cf_add(hsum, hls[p]);
    }
    if (!diff_mode && !r->carrier_phase_init) {
        r->carrier_phase_init = 1;
    } else if (diff_mode && !r->carrier_phase_init) {
        cf havg = cf_divf(hsum, (float)np);
        float am = cf_abs(havg);
        if (am > 0.01f) { r->carrier_phase_corr = cf_divf(cf_conj(havg), am); r->carrier_phase_init = 1; }
    }
    for (int p = 0; p < np; ++p) hls[p] = cf_mul(hls[p], r->carrier_phase_corr);

    if (!diff_mode) { /* CPE */
        cf cs = cf_mk(0, 0);
        float ws = 0.0f;
        for (int p = 0; p < np; ++p) {
            cf hold = r->H[g->pilot_idx[p]];
            float hm = cf_abs(hold);
            if (hm > 0.01f) {
                cf ratio = cf_mul(hls[p], cf_conj(hold));
                float mag = cf_abs(ratio);
                if (mag > 1e-6f) { cs = cf_add(cs, cf_scale(hm, cf_divf(ratio, mag))); ws += hm; }
            }
        }
        if (ws > 0.01f) {
            float ph = cf_arg(cs);
            if (fabsf(ph) > 0.001f) {
                cf c = cf_expj(ph);
                for (int i = 0; i < g->n_data; ++i) r->H[g->data_idx[i]] = cf_mul(r->H[g->data_idx[i]], c);
                for (int p = 0; p < np; ++p) r->H[g->pilot_idx[p]] = cf_mul(r->H[g->pilot_idx[p]], c);
            }
        }
    }
    float sp_sum = 0.0f;
    for (int p = 0; p < np; ++p) sp_sum += cf_norm(hls[p]);
    float signal_power = sp_sum / (float)np;
    int noise_count = 0;
    float noise_power_sum = 0.0f;
    for (int p = 0; p < np; ++p) {
        int idx = g->pilot_idx[p];
        if (r->have_prev_pilot)
            if (cf_norm(r->prev_pilot[p]) > 1e-6f && cf_norm(hls[p]) > 1e-6f) {
                noise_power_sum += cf_norm(cf_sub(hls[p], r->prev_pilot[p]));
                noise_count++;
            }
        cf hold = r->H[idx];
        if (diff_mode) {
            float nm = alpha * cf_abs(hls[p]) + (1.0f - alpha) * cf_abs(hold);
            float ph = cf_arg(hold);
            r->H[idx] = cf_mk(nm * cosf(ph), nm * sinf(ph));
        } else {
            r->H[idx] = cf_add(cf_scale(alpha, hls[p]), cf_scale(1.0f - alpha, hold));
        }
    }
    if (noise_count == 0) { noise_power_sum = signal_power / 31.6f; noise_count = 1; }
    memcpy(r->prev_pilot, hls, sizeof(cf) * (size_t)np);
    r->have_prev_pilot = 1;

    if (!diff_mode) {
        cf des[RO_NCAR];
        for (int p = 0; p < np; ++p) {
            int bin = g->pilot_idx[p];
            int k = (bin <= RO_FFT / 2) ? bin : bin - RO_FFT;
            float ph = -r->slope * (float)k;
            des[p] = cf_mul(r->H[bin], cf_mk(cosf(ph), sinf(ph)));
        }
        for (int d = 0; d < g->n_data; ++d) {
            int lo = g->interp_lo[d], hi = g->interp_hi[d];
            cf hl = (lo >= 0) ? des[lo] : cf_mk(0, 0), hu = (hi >= 0) ? des[hi] : cf_mk(0, 0), ih;
            float a = g->interp_alpha[d];
            if (lo >= 0 && hi >= 0) ih = cf_add(cf_scale(1.0f - a, hl), cf_scale(a, hu));
            else if (lo >= 0) ih = hl;
            else ih = hu;
            int bin = g->data_idx[d];
            int k = (bin <= RO_FFT / 2) ? bin : bin - RO_FFT;
            float ph = r->slope * (float)k;
            r->H[bin] = cf_mul(ih, cf_mk(cosf(ph), sinf(ph)));
        }
    } else {
        for (int d = 0; d < g->n_data; ++d) {
            int lo = g->interp_lo[d], hi = g->interp_hi[d];
            float a = g->interp_alpha[d], im = 0.0f;
            if (lo >= 0 && hi >= 0) {
                float m1 = cf_abs(r->H[g->pilot_idx[lo]]), m2 = cf_abs(r->H[g->pilot_idx[hi]]);
                im = (1.0f - a) * m1 + a * m2;
            } else if (lo >= 0) im = cf_abs(r->H[g->pilot_idx[lo]]);
            else if (hi >= 0) im = cf_abs(r->H[g->pilot_idx[hi]]);
            float ph = cf_arg(r->H[g->data_idx[d]]);
            r->H[g->data_idx[d]] = cf_mk(im * cosf(ph), im * sinf(ph));
        }
    }
    if (!diff_mode && r->have_dd && r->snr_count >= 3) {
        for (int i = 0; i < g->n_data; ++i) {
            float c = r->dd[i];
            if (fabsf(c) > 0.001f) r->H[g->data_idx[i]] = cf_mul(r->H[g->data_idx[i]], cf_expj(c * 0.3f));
        }
    }
    float mm = 0.0f;
    for (int p = 0; p < np; ++p) mm += cf_abs(hls[p]);
    mm /= (float)np;
    float mv = 0.0f;
    for (int p = 0; p < np; ++p) { float d = cf_abs(hls[p]) - mm; mv += d * d; }
    mv /= (float)np;
    r->fading_index = (mm > 0.01f) ? sqrtf(mv) / mm : 0.0f;
    if (noise_count > 0 && noise_power_sum > 0.0f && !diff_mode && noise_count > 1) {
        float inst = signal_power / fmaxf_(r->noise_var, 1e-6f);
        inst = fmaxf_(0.1f, fminf_(10000.0f, inst));
        r->snr_lin = 0.3f * inst + (1.0f - 0.3f) * r->snr_lin;
    }
    r->snr_count++;
}

static cf ro_hard_decision(cf s, int mod) { /* channel_equalizer.cpp:1168-1230 */
    switch (mod) {
        case RO_BPSK: return cf_mk(s.re > 0 ? 1.0f : -1.0f, 0);
        case RO_QAM16: {
            float v[2] = { s.re, s.im }, o[2];
            for (int i = 0; i < 2; ++i) {
                float x = v[i];
                o[i] = (x < -0.4f) ? -0.9487f : (x < 0.0f) ? -0.3162f : (x < 0.4f) ? 0.3162f : 0.9487f;
            }
            return cf_mk(o[0], o[1]);
        }
        case RO_QAM32: {
            const float d = 0.1961161351381840f;
            float x = s.re, y = s.im, I, Q;
            I = (x < -2 * d) ? -3 * d : (x < 0) ? -d : (x < 2 * d) ? d : 3 * d;
            Q = (y < -6 * d) ? -7 * d : (y < -4 * d) ? -5 * d : (y < -2 * d) ? -3 * d : (y < 0) ? -d
              : (y < 2 * d) ? d : (y < 4 * d) ? 3 * d : (y < 6 * d) ? 5 * d : 7 * d;
            return cf_mk(I, Q);
        }
        case RO_QAM64: {
            const float d = 0.1543f;
            float v[2] = { s.re, s.im }, o[2];
            for (int i = 0; i < 2; ++i) {
                float y = v[i];
                o[i] = (y < -6 * d) ? -7 * d : (y < -4 * d) ? -5 * d : (y < -2 * d) ? -3 * d : (y < 0) ? -d
                     : (y < 2 * d) ? d : (y < 4 * d) ? 3 * d : (y < 6 * d) ? 5 * d : 7 * d;
            }
            return cf_mk(o[0], o[1]);
        }
        default: return cf_mk(s.re > 0 ? 0.7071f : -0.7071f, s.im > 0 ? 0.7071f : -0.7071f);
    }
}

static void ro_equalize(ro_rx* r, const cf* fd, cf* eq) { /* channel_equalizer.cpp:1259-1451 */
    const ro_geom* g = r->g;
    int nd = g->n_data, mod = g->mod;
    if (!ro_is_coherent(mod)) {
        float avg = 0.0f;
        for (int i = 0; i < nd; ++i) avg += cf_norm(r->H[g->data_idx[i]]);
        avg /= (float)nd;
        float thr = 0.25f * avg;
        float snv = r->noise_var;
        if (snv < 1e-6f) snv = avg / 31.6f;
        for (int i = 0; i < nd; ++i) {
            cf h = r->H[g->data_idx[i]];
            float hp = cf_norm(h), den = hp + snv;
            if (den < 1e-10f) { eq[i] = cf_mk(0, 0); r->cnv[i] = 100.0f; }
            else { eq[i] = cf_divf(cf_mul(fd[g->data_idx[i]], cf_conj(h)), den); r->cnv[i] = snv / (hp + snv); }
            if (hp < thr) r->cnv[i] = 100.0f;
            r->cnv[i] = fmaxf_(1e-6f, fminf_(100.0f, r->cnv[i]));
        }
        return;
    }
    for (int i = 0; i < nd; ++i) {
        cf h = r->H[g->data_idx[i]];
        float hp = cf_norm(h), den = hp + r->noise_var;
        if (den < 1e-10f) { eq[i] = cf_mk(0, 0); r->cnv[i] = 100.0f; }
        else {
            eq[i] = cf_divf(cf_mul(cf_conj(h), fd[g->data_idx[i]]), den);
            r->cnv[i] = fmaxf_(1e-6f, fminf_(100.0f, r->noise_var / den));
        }
    }
    float avg = 0.0f;
    for (int i = 0; i < nd; ++i) avg += cf_norm(r->H[g->data_idx[i]]);
    avg /= (float)nd;
    float thr = 0.25f * avg;
    for (int i = 0; i < nd; ++i) if (cf_norm(r->H[g->data_idx[i]]) < thr) r->cnv[i] = 100.0f;

    int dd_ok = (mod == RO_QPSK || mod == RO_BPSK || mod == RO_QAM16 || mod == RO_QAM32 || mod == RO_QAM64);
    if (dd_ok && r->snr_count >= 2) {
        if (!r->have_dd) { memset(r->dd, 0, sizeof(r->dd)); r->have_dd = 1; }
        float mt = 0.3f, pt = 0.61f;
        if (mod == RO_QAM16) { mt = 0.25f; pt = 0.44f; }
        else if (mod == RO_QAM32 || mod == RO_QAM64) { mt = 0.20f; pt = 0.35f; }
        for (int i = 0; i < nd; ++i) {
            if (cf_abs(eq[i]) < mt) { r->dd[i] = 0.0f; continue; }
            cf dec = ro_hard_decision(eq[i], mod);
            float pe = cf_arg(cf_mul(eq[i], cf_conj(dec)));
            r->dd[i] = (fabsf(pe) < pt) ? -pe : 0.0f;
        }
    }
}

static float ro_clip(float llr) { /* soft_demap.hpp:22-29 */
    float c = fmaxf_(-20.0f, fminf_(20.0f, llr));
    if (fabsf(c) < 0.01f) c = (c >= 0) ? 0.01f : -0.01f;
    return c;
}
static float ro_ce_margin(int mod) { /* soft_demap.hpp:309-332 */
    switch (mod) {
        case RO_D8PSK: case RO_QAM8: return 1.1f;
        case RO_QAM16: return 1.2f;
        case RO_QAM32: return 1.5f;
        case RO_QAM64: return 1.8f;
        case RO_QAM256: return 2.5f;
        default: return 1.0f;
    }
}

static int ro_demap(int mod, cf sym, cf prev, float nv, float* o) { /* soft_demap.hpp:36-263 */
    float I = sym.re, Q = sym.im;
    switch (mod) {
        case RO_BPSK: o[0] = ro_clip(-2.0f * I / nv); return 1;
        case RO_QPSK: {
            float sc = -2.0f * 0.7071067811865476f / nv;
            o[0] = ro_clip(I * sc); o[1] = ro_clip(Q * sc); return 2;
        }
        case RO_QAM16: {
            float sc = 2.0f / nv;
            const float T = 0.6324555320336759f;
            o[0] = ro_clip(-sc * I); o[1] = ro_clip(sc * (fabsf(I) - T));
            o[2] = ro_clip(-sc * Q); o[3] = ro_clip(sc * (fabsf(Q) - T));
            return 4;
        }
        case RO_QAM32: {
            static const float IL[4] = { -3, -1, 1, 3 }, QL[8] = { -7, -5, -3, -1, 1, 3, 5, 7 };
            static const int IG[4] = { 0, 1, 3, 2 }, QG[8] = { 0, 1, 3, 2, 6, 7, 5, 4 };
            const float S = 0.1961161351381840f;
            float sf = 2.0f / nv;
            for (int b = 0; b < 5; ++b) {
                int mask = 1 << (4 - b);
                float m0 = 1e10f, m1 = 1e10f;
                for (int qi = 0; qi < 8; ++qi)
                    for (int ii = 0; ii < 4; ++ii) {
                        float dr = I - IL[ii] * S, di = Q - QL[qi] * S;
                        float d2 = dr * dr + di * di;
                        int bits = (QG[qi] << 2) | IG[ii];
                        if (bits & mask) { if (d2 < m1) m1 = d2; } else { if (d2 < m0) m0 = d2; }
                    }
                o[b] = ro_clip(sf * (m1 - m0));
            }
            return 5;
        }
        case RO_QAM64: {
            float sc = 2.0f / nv;
            const float D2 = 0.3086067f, D4 = 0.6172134f;
            o[0] = ro_clip(-sc * I); o[1] = ro_clip(sc * (fabsf(I) - D4)); o[2] = ro_clip(sc * (fabsf(fabsf(I) - D4) - D2));
            o[3] = ro_clip(-sc * Q); o[4] = ro_clip(sc * (fabsf(Q) - D4)); o[5] = ro_clip(sc * (fabsf(fabsf(Q) - D4) - D2));
            return 6;
        }
        case RO_QAM256: {
            float sc = 2.0f / nv;
            const float D2 = 0.1290994f, D4 = 0.2581989f, D8 = 0.5163978f;
            float v[2] = { I, Q };
            for (int a = 0; a < 2; ++a) {
                float x = v[a];
                o[4 * a + 0] = ro_clip(-sc * x);
                o[4 * a + 1] = ro_clip(sc * (fabsf(x) - D8));
                o[4 * a + 2] = ro_clip(sc * (fabsf(fabsf(x) - D8) - D4));
                o[4 * a + 3] = ro_clip(sc * (fabsf(fabsf(fabsf(x) - D8) - D4) - D2));
            }
            return 8;
        }
        case RO_DBPSK: {
            cf diff = cf_mul(sym, cf_conj(prev));
            float pd = atan2f(diff.im, diff.re);
            float sp = cf_abs(sym) * cf_abs(prev);
            if (sp < 1e-6f) { o[0] = 0.0f; return 1; }
            float dnv = 2.0f * nv;
            float conf = 2.0f * sp / dnv;
            o[0] = ro_clip(conf * cosf(pd));
            return 1;
        }
        case RO_DQPSK: {
            cf diff = cf_mul(sym, cf_conj(prev));
            float dI = diff.re, dQ = diff.im, dm = cf_abs(diff);
            if (dm < 1e-6f) { o[0] = o[1] = 0.0f; return 2; }
            float dnv = 2.0f * nv;
            float sp = cf_abs(sym) * cf_abs(prev);
            float snr = sp / dnv;
            float sc = 2.0f * sqrtf(snr);
            const float pi_f = 3.14159265358979f;
            float ph = atan2f(dQ, dI);
            o[0] = ro_clip(sc * sinf(ph + pi_f / 4));
            o[1] = ro_clip(sc * (fabsf(dI) - fabsf(dQ)) / dm);
            return 2;
        }
        case RO_D8PSK: { /* soft_demap.hpp:239-263 */
            cf diff = cf_mul(sym, cf_conj(prev));
            float pd = atan2f(diff.im, diff.re);
            float sp = cf_abs(sym) * cf_abs(prev);
            if (sp < 1e-6f) { o[0] = o[1] = o[2] = 0.0f; return 3; }
            float dnv = 2.0f * nv;
            float conf = sp / dnv;
            o[0] = ro_clip(conf * sinf(pd));
            o[1] = ro_clip(conf * sinf(2.0f * pd));
            o[2] = ro_clip(conf * sinf(4.0f * pd));
            return 3;
        }
        default: return 0;
    }
}

static int ro_demod_symbol(ro_rx* r, const cf* eq, float* soft) { /* demodulator.cpp:208-508 */
    const ro_geom* g = r->g;
    int nd = g->n_data, mod = g->mod, n = 0;
    float margin = ro_ce_margin(mod);
    if (!r->have_ema) {
        for (int i = 0; i < nd; ++i) { r->ema[i] = cf_abs(eq[i]); r->var[i] = 0.0f; }
        r->have_ema = 1;
    } else {
        for (int i = 0; i < nd; ++i) {
            float mag = cf_abs(eq[i]);
            float delta = mag - r->ema[i];
            r->ema[i] += 0.3f * delta;
            r->var[i] += 0.3f * (delta * delta - r->var[i]);
        }
    }
    if ((mod == RO_DQPSK || mod == RO_DBPSK || mod == RO_D8PSK) && !r->have_dprev) {
        for (int i = 0; i < nd; ++i) r->dprev[i] = cf_mk(1, 0);
        r->have_dprev = 1;
    }
    if (mod == RO_D8PSK && r->fading_index > 0.30f) {
        /* demodulateD8PSKTwoPass (demodulator.cpp:533-620): common phase error from the embedded DQPSK grid,
         * half of it removed before the D8PSK demap; the corrected symbol becomes the next reference */
        float sin_sum = 0.0f, cos_sum = 0.0f, weight_sum = 0.0f;
        for (int i = 0; i < nd; ++i) {
            cf prev = r->dprev[i];
            float sp = cf_abs(eq[i]) * cf_abs(prev);
            if (sp > 0.1f) {
                cf diff = cf_mul(eq[i], cf_conj(prev));
                float phase = atan2f(diff.im, diff.re);
                float pmo = (float)(phase - M_PI / 4.0f);
                int quadrant = (int)round((double)(pmo * 2.0f) / M_PI);
                quadrant = ((quadrant % 4) + 4) % 4;
                float expected = (float)(quadrant * M_PI / 2.0f + M_PI / 4.0f);
                float error = phase - expected;
                while (error > M_PI) error = (float)(error - 2 * M_PI);
                while (error < -M_PI) error = (float)(error + 2 * M_PI);
                sin_sum += sp * sinf(error);
                cos_sum += sp * cosf(error);
                weight_sum += sp;
            }
        }
        float mean_error = (weight_sum > 0.1f) ? atan2f(sin_sum, cos_sum) : 0.0f;
        cf pc = cf_mk(1.0f, 0.0f);
        if (fabsf(mean_error) > 0.05f && fabsf(mean_error) < 0.26f) {
            float ce = mean_error * 0.5f;
            pc = cf_mk(cosf(-ce), sinf(-ce));
        }
        for (int i = 0; i < nd; ++i) {
            float nv = r->cnv[i] * margin;
            float msq = r->ema[i] * r->ema[i] + 1e-6f;
            float nvar = r->var[i] / msq;
            nv *= (1.0f + 10.0f * nvar);
            cf cs = cf_mul(eq[i], pc);
            n += ro_demap(mod, cs, r->dprev[i], nv, soft + n);
            r->dprev[i] = cs;
        }
        r->snr_count++;   /* demodulator.cpp:292 */
        return n;
    }
    for (int i = 0; i < nd; ++i) {
        float nv = r->cnv[i] * margin;
        float msq = r->ema[i] * r->ema[i] + 1e-6f;
        float nvar = r->var[i] / msq;
        nv *= (1.0f + 10.0f * nvar);
        int nb = ro_demap(mod, eq[i], r->dprev[i], nv, soft + n);
        if (!ro_is_coherent(mod)) r->dprev[i] = eq[i];
        n += nb;
    }
    /* Differential decision-directed tracking (demodulator.cpp:418-493) reads dbpsk_prev_equalized
     * after it was overwritten with the current symbol, so diff = |eq|^2 (phase +0) and every
     * channel_estimate rotation is by exp(-j0): values unchanged.  pilot_phase_correction is
     * never read by equalize().  Nothing to restate. */
    return n;
}

int ro_rx_process(const ro_geom* g, const float* samples, int n, float cfo_hz, long long abs_pos,
                  float* llr_out, int max_llr, ro_rx_aux* aux) {
    return ro_rx_process_flags(g, samples, n, cfo_hz, abs_pos, 0, llr_out, max_llr, aux);
}

int ro_rx_process_flags(const ro_geom* g, const float* samples_in, int n, float cfo_hz, long long abs_pos, int flags,
                        float* llr_out, int max_llr, ro_rx_aux* aux) {
    /* OFDMChirpWaveform::process (ofdm_chirp_waveform.cpp:391-468) +
     * OFDMDemodulator::processPresynced (demodulator.cpp:1250-1414) */
    if (n < RO_SYM) return 0;
    /* flags bit0 = burst_interleaved_detected_: the first LTS symbol (getSamplesPerSymbol() samples, cyclic prefix
     * included) is negated on a copy before the demodulator sees it (ofdm_chirp_waveform.cpp:421-440) */
    const float* samples = samples_in;
    float* modified = NULL;
    if (flags & 1) {
        modified = (float*)malloc((size_t)n * sizeof(float));
        if (!modified) return 0;
        memcpy(modified, samples_in, (size_t)n * sizeof(float));
        for (int i = 0; i < RO_SYM && i < n; ++i) modified[i] = -modified[i];
        samples = modified;
    }
    static _Thread_local ro_rx r;
    memset(&r, 0, sizeof(r));
    r.g = g;
    float init = (float)(-2.0f * M_PI * (double)cfo_hz * (double)(unsigned long long)abs_pos / (double)48000u);
    while ((double)init > M_PI) init = (float)((double)init - 2.0f * M_PI);
    while ((double)init < -M_PI) init = (float)((double)init + 2.0f * M_PI);
    r.cfo_hz = cfo_hz;
    r.corr_phase = init;
    ro_nco_init(&r.mixer, 1500.0f, 48000.0f);
    for (int i = 0; i < RO_FFT; ++i) r.H[i] = cf_mk(1, 0);
    r.snr_lin = 1.0f;
    r.noise_var = 0.1f;
    r.carrier_phase_corr = cf_mk(1, 0);

    ro_estimate_lts(&r, samples);
    int remaining = n - 2 * RO_SYM;
    const float* p = samples + 2 * RO_SYM;
    cf fd[RO_FFT], eq[RO_NCAR];
    float soft[8 * RO_NCAR];
    while (remaining >= RO_SYM) {
        ro_symbol_fft(&r, p, fd);
        if (g->n_pilot > 0) ro_update_channel(&r, fd);
        ro_equalize(&r, fd, eq);
        int nb = ro_demod_symbol(&r, eq, soft);
        for (int b = 0; b < nb; ++b) if (r.n_soft + b < max_llr) llr_out[r.n_soft + b] = soft[b];
        r.n_soft += nb;
        p += RO_SYM;
        remaining -= RO_SYM;
    }
    if (aux) {
        aux->snr_db = 10.0f * log10f(r.snr_lin);
        aux->cfo_hz = r.cfo_hz;
        aux->fading_index = r.fading_index;
        aux->noise_variance = r.noise_var;
        aux->lts_phase_slope = r.slope;
        aux->snr_linear = r.snr_lin;
        aux->corr_phase = r.corr_phase;
        aux->snr_symbol_count = (float)r.snr_count;
        for (int l = 0; l < RO_NCAR; ++l) { aux->h[2 * l] = r.H[g->all_idx[l]].re; aux->h[2 * l + 1] = r.H[g->all_idx[l]].im; }
    }
    free(modified);
    return r.n_soft;
}

/* fec::BurstInterleaver (src/fec/burst_interleaver.cpp:8-78): N frames of 324 coded bytes / 2592 soft bits.
 * TX: physical[(N*b+f)/324][(N*b+f)%324] = logical[f][b]; RX: the 8 soft bits of every byte travel together.
 * N < 2 copies (burst_interleaver.cpp:12,45). */
void ro_burst_interleave(int n_frames, const uint8_t* logical, uint8_t* physical) {
    const int B = 324;
    if (n_frames < 2) { memcpy(physical, logical, (size_t)(n_frames > 0 ? n_frames : 0) * B); return; }
    for (int f = 0; f < n_frames; ++f)
        for (int b = 0; b < B; ++b) {
            const int flat = n_frames * b + f;
            physical[(flat / B) * B + flat % B] = logical[f * B + b];
        }
}
void ro_burst_deinterleave(int n_frames, const float* physical, int stride, float* logical /* [n_frames][2592] */) {
    const int B = 324, BITS = 2592;
    for (int f = 0; f < n_frames; ++f)
        for (int b = 0; b < B; ++b) {
            const int flat = n_frames * b + f;
            const int pf = n_frames < 2 ? f : flat / B, pb = n_frames < 2 ? b : flat % B;
            for (int bit = 0; bit < 8; ++bit) logical[f * BITS + 8 * b + bit] = physical[(size_t)pf * stride + 8 * pb + bit];
        }
}

/* ------------------------------------------------------------------ decodeFixedFrame */

/* CodewordStatus::reassemble (frame_v2.cpp:1030-1063) + reassembleCodewords (:959-989), including the
 * reference quirk that a CW1+ whose first byte equals 0xD5 loses its first two bytes. All four
 * codewords decoded. Returns assembled length (0 = "empty"). */
static int ro_parse_header(const uint8_t* d, int len, int* is_control, int* payload_len) {
    /* frame_v2.cpp:1195-1252 */
    if (len < 20) return 0;
    if (d[0] != 0x55 || d[1] != 0x4C) return 0;
    int t = d[2];
    int ctl = (t == 0x10 || t == 0x11 || t == 0x16 || t == 0x17 || t == 0x20 || t == 0x21 || t == 0x15 || t == 0x40);
    *is_control = ctl;
    if (ctl) {
        if (ro_crc16(d, 18) != (uint16_t)((d[18] << 8) | d[19])) return 0;
        *payload_len = 0;
    } else {
        *payload_len = (d[13] << 8) | d[14];
        if (ro_crc16(d, 15) != (uint16_t)((d[15] << 8) | d[16])) return 0;
    }
    return 1;
}
static int ro_reassemble(uint8_t cw[4][68], int bpc, uint8_t* out /* >= 4*68 */) {
    int ctl, plen;
    if (!ro_parse_header(cw[0], bpc, &ctl, &plen)) return 0;
    int expected = ctl ? 20 : 17 + plen + 2, n = 0;
    for (int i = 0; i < 4; ++i) {
        int remaining = expected - n;
        if (remaining == 0) break;
        if (i == 0 || cw[i][0] != 0xD5) {
            int c = remaining < bpc ? remaining : bpc;
            memcpy(out + n, cw[i], (size_t)c); n += c;
        } else {
            int c = remaining < bpc - 2 ? remaining : bpc - 2;
            memcpy(out + n, cw[i] + 2, (size_t)c); n += c;
        }
    }
    return n;
}
/* verifyFrame lambda (frame_v2.cpp:1583-1589) == the frame_valid test (:1564-1577) */
static int ro_verify_frame(const uint8_t* d, int len) {
    int ctl, plen;
    if (len == 0) return 0;
    if (!ro_parse_header(d, len, &ctl, &plen)) return 0;
    if (ctl) return 1; /* ControlFrame::deserialize re-checks magic + the same CRC */
    int sz = 17 + plen + 2;
    if (len < sz) return 0;
    return ro_crc16(d, sz - 2) == (uint16_t)((d[sz - 2] << 8) | d[sz - 1]);
}

typedef struct { int frame_bit; float abs_llr; } ro_suspect;
/* libstdc++ std::sort (introsort, threshold 16, median-of-3 to first), comparator a.abs_llr < b.abs_llr.
 * Restated because the order of equal keys decides which suspects are searched first. */
static int sus_lt(const ro_suspect* a, const ro_suspect* b) { return a->abs_llr < b->abs_llr; }
static void sus_swap(ro_suspect* a, ro_suspect* b) { ro_suspect t = *a; *a = *b; *b = t; }
static void sus_linear_insert(ro_suspect* last) {
    ro_suspect val = *last;
    ro_suspect* next = last - 1;
    while (sus_lt(&val, next)) { *last = *next; last = next; --next; }
    *last = val;
}
static void sus_insertion(ro_suspect* first, ro_suspect* last) {
    if (first == last) return;
    for (ro_suspect* i = first + 1; i != last; ++i) {
        if (sus_lt(i, first)) { ro_suspect v = *i; memmove(first + 1, first, (size_t)(i - first) * sizeof(*i)); *first = v; }
        else sus_linear_insert(i);
    }
}
static int g_sort_unpinned = 0;
static void sus_introsort(ro_suspect* first, ro_suspect* last, int depth) {
    while (last - first > 16) {
        if (depth == 0) { g_sort_unpinned = 1; sus_insertion(first, last); return; } /* heap fallback: not restated */
        --depth;
        ro_suspect *mid = first + (last - first) / 2, *a = first + 1, *b = mid, *c = last - 1;
        if (sus_lt(a, b)) { if (sus_lt(b, c)) sus_swap(first, b); else if (sus_lt(a, c)) sus_swap(first, c); else sus_swap(first, a); }
        else { if (sus_lt(a, c)) sus_swap(first, a); else if (sus_lt(b, c)) sus_swap(first, c); else sus_swap(first, b); }
        ro_suspect *lo = first + 1, *hi = last;
        for (;;) {
            while (sus_lt(lo, first)) ++lo;
            --hi;
            while (sus_lt(first, hi)) --hi;
            if (!(lo < hi)) break;
            sus_swap(lo, hi);
            ++lo;
        }
        sus_introsort(lo, last, depth);
        last = lo;
    }
}
static void sus_sort(ro_suspect* a, int n) {
    if (n == 0) return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    sus_introsort(a, a + n, 2 * lg);
    if (n > 16) { sus_insertion(a, a + 16); for (ro_suspect* i = a + 16; i != a + n; ++i) sus_linear_insert(i); }
    else sus_insertion(a, a + n);
}

int ro_decode_fixed_frame(const float* llr, int n, int rate, int ch_deint, int bps, int flags,
                          uint8_t* data_out, uint8_t* ok_out, int* iters_out, int* attempts_out) {
    /* frame_v2.cpp:1335-1883 */
    static _Thread_local ro_ldpc code;
    static _Thread_local int table[4 * RO_CW_BITS];
    static _Thread_local int table_key = -1;
    if (code.rate != rate || code.n == 0) ro_ldpc_build(&code, rate);
    int key = (bps << 1) | (ch_deint ? 1 : 0);
    if (table_key != key) { ro_rx_gather_table(bps, ch_deint, table); table_key = key; }
    int bpc = ro_info_bits(rate) / 8, max_iter = ro_recommended_iters(rate);
    memset(data_out, 0, (size_t)(4 * bpc));
    memset(ok_out, 0, 4);
    if (n < RO_FRAME_BITS) return 0;

    static const float f0[4] = { 0.875f, 0.75f, 0.625f, 0.5f };
    static const float s1[15] = { 0.3f, 0.7f, 0.3f, 1.0f, 0.5f, 1.5f, 0.3f, 2.0f, 0.5f, 0.7f, 1.0f, 2.5f, 0.3f, 1.5f, 0.5f };
    static const float f1[15] = { 0.75f, 0.625f, 0.875f, 0.75f, 0.625f, 0.75f, 0.5f, 0.625f, 0.875f, 0.75f, 0.625f, 0.875f, 0.75f, 0.5f, 0.625f };
    static const float s2[5] = { 0.3f, 0.8f, 1.5f, 2.5f, 4.0f };
    static const float s3[3] = { 0.5f, 1.5f, 3.0f };
    static const float s4[3] = { 0.5f, 1.5f, 3.0f };
    static const float s5[5] = { 0.0f, 0.2f, 0.5f, 1.0f, 1.5f };
    static const float s6[3] = { 0.3f, 1.0f, 2.0f };

    int good = 0;
    float cwllr[4][RO_CW_BITS];
    uint8_t cwd[4][68];
    memset(cwd, 0, sizeof(cwd));
    /* One LDPCDecoder object serves the whole frame and its min-sum factor is only restored to
     * 0.9375 after phase 0; phases 1 and 2 leave it at 0.875, which then also applies to the
     * FIRST decode of the following codewords (frame_v2.cpp:1359-1361, :1416, :1447, :1470). */
    float dec_factor = 0.9375f;
    for (int cw = 0; cw < 4; ++cw) {
        float* b = cwllr[cw];
        for (int i = 0; i < RO_CW_BITS; ++i) b[i] = llr[table[cw * RO_CW_BITS + i]];
        uint8_t dec[81];
        int iters = 0, attempts = 1;
        int ok = ro_ldpc_decode(&code, b, RO_CW_BITS, max_iter, dec_factor, dec, &iters);
        if (!ok && (flags & 3)) {
            uint32_t h = 0;
            for (int j = 0; j < 16; ++j) {
                uint32_t u;
                memcpy(&u, &b[j], 4);
                h ^= u + 0x9e3779b9u + (h << 6) + (h >> 2);
            }
            if (flags & 1) {
                for (int t = 0; t < 4 && !ok; ++t) {
                    int it;
                    uint8_t d2[81];
                    attempts++;
                    if (ro_ldpc_decode(&code, b, RO_CW_BITS, max_iter, f0[t], d2, &it)) { ok = 1; iters = it; }
                    memcpy(dec, d2, 81);
                }
                dec_factor = 0.9375f;
            }
            if (!ok && (flags & 2)) {
                float pert[RO_CW_BITS];
                ro_mt rng;
                for (int phase = 1; phase <= 6 && !ok; ++phase) {
                    int cnt = (phase == 1) ? 15 : (phase == 2) ? 5 : (phase == 5) ? 5 : 3;
                    for (int t = 0; t < cnt && !ok; ++t) {
                        float sigma;
                        uint32_t seed;
                        switch (phase) {
                            case 1: dec_factor = f1[t]; sigma = s1[t]; seed = h + (uint32_t)(t * 997 + t * 31); break;
                            case 2: dec_factor = (t % 2 == 0) ? 0.625f : 0.875f; sigma = s2[t]; seed = h + (uint32_t)((t + 15) * 997 + 12345); break;
                            case 3: sigma = s3[t]; seed = h + (uint32_t)((t + 20) * 997 + 54321); break;
                            case 4: sigma = s4[t]; seed = h + (uint32_t)((t + 23) * 997 + 99999); break;
                            case 5: sigma = s5[t]; seed = h + (uint32_t)((t + 26) * 997 + 33333); break;
                            default: sigma = s6[t]; seed = h + (uint32_t)((t + 31) * 997 + 77777); break;
                        }
                        ro_mt_seed(&rng, seed);
                        ro_normal nd = { 0, 0 };
                        for (int i = 0; i < RO_CW_BITS; ++i) {
                            float v = b[i];
                            switch (phase) {
                                case 1: v += ro_normal_draw(&nd, &rng, 0.0f, sigma); break;
                                case 2: v = fmaxf_(-10.0f, fminf_(10.0f, v)); v += ro_normal_draw(&nd, &rng, 0.0f, sigma); break;
                                case 3: v = v * 0.5f + ro_normal_draw(&nd, &rng, 0.0f, sigma); break;
                                case 4: v = fmaxf_(-6.0f, fminf_(6.0f, v)); v += ro_normal_draw(&nd, &rng, 0.0f, sigma); break;
                                case 5: v = (v >= 0) ? 1.0f : -1.0f; v += ro_normal_draw(&nd, &rng, 0.0f, sigma); break;
                                default: v = v * 0.25f + ro_normal_draw(&nd, &rng, 0.0f, sigma); break;
                            }
                            pert[i] = v;
                        }
                        int it;
                        attempts++;
                        if (ro_ldpc_decode(&code, pert, RO_CW_BITS, max_iter, dec_factor, dec, &it)) { ok = 1; iters = it; }
                    }
                    if (phase == 1 || phase == 2) dec_factor = 0.875f;
                }
            }
        }
        ok_out[cw] = (uint8_t)ok;
        if (ok) { memcpy(cwd[cw], dec, (size_t)bpc); good++; }
        if (iters_out) iters_out[cw] = iters;
        if (attempts_out) attempts_out[cw] = attempts;
    }

    /* ---- LDPC false-positive recovery (frame_v2.cpp:1564-1880) */
    if (good == 4 && (flags & 4)) {
        uint8_t fd[4 * 68], trial[4 * 68];
        int flen = ro_reassemble(cwd, bpc, fd);
        if (!ro_verify_frame(fd, flen)) {
            int recovered = 0;
            if (flen == 0) {
                for (int by = 0; by < bpc && !recovered; ++by)
                    for (int bit = 0; bit < 8 && !recovered; ++bit) {
                        cwd[0][by] ^= (uint8_t)(1 << bit);
                        if (cwd[0][0] == 0x55 && cwd[0][1] == 0x4C &&
                            ro_crc16(cwd[0], 15) == (uint16_t)((cwd[0][15] << 8) | cwd[0][16])) {
                            int tl = ro_reassemble(cwd, bpc, trial);
                            if (ro_verify_frame(trial, tl)) recovered = 1;
                        }
                        if (!recovered) cwd[0][by] ^= (uint8_t)(1 << bit);
                    }
                if (!recovered) {
                    int tb = bpc * 8;
                    for (int b1 = 0; b1 < tb && !recovered; ++b1) {
                        cwd[0][b1 / 8] ^= (uint8_t)(1 << (b1 % 8));
                        for (int b2 = b1 + 1; b2 < tb && !recovered; ++b2) {
                            cwd[0][b2 / 8] ^= (uint8_t)(1 << (b2 % 8));
                            if (cwd[0][0] == 0x55 && cwd[0][1] == 0x4C &&
                                ro_crc16(cwd[0], 15) == (uint16_t)((cwd[0][15] << 8) | cwd[0][16])) {
                                int tl = ro_reassemble(cwd, bpc, trial);
                                if (ro_verify_frame(trial, tl)) recovered = 1;
                            }
                            if (!recovered) cwd[0][b2 / 8] ^= (uint8_t)(1 << (b2 % 8));
                        }
                        if (!recovered) cwd[0][b1 / 8] ^= (uint8_t)(1 << (b1 % 8));
                    }
                }
            } else {
                int ctl, plen;
                if (ro_parse_header(fd, flen, &ctl, &plen) && !ctl) {
                    int expected = 17 + plen + 2;
                    if (flen >= expected) {
                        uint16_t stored = (uint16_t)((fd[expected - 2] << 8) | fd[expected - 1]);
                        int data_bytes = expected - 2, data_bits = data_bytes * 8;
                        uint16_t orig = ro_crc16(fd, data_bytes), syn = stored ^ orig;
                        static _Thread_local uint16_t deltas[4 * 68 * 8];
                        for (int p = 0; p < data_bits; ++p) {
                            fd[p / 8] ^= (uint8_t)(1 << (p % 8));
                            deltas[p] = orig ^ ro_crc16(fd, data_bytes);
                            fd[p / 8] ^= (uint8_t)(1 << (p % 8));
                        }
                        for (int p = 0; p < data_bits && !recovered; ++p)
                            if (deltas[p] == syn) {
                                int fb = p / 8, ci = fb / bpc;
                                if (ci < 4) { cwd[ci][fb % bpc] ^= (uint8_t)(1 << (p % 8)); recovered = 1; }
                            }
                        if (!recovered)
                            for (int bit = 0; bit < 16 && !recovered; ++bit)
                                if (syn == (1u << bit)) {
                                    int fb = (bit >= 8) ? expected - 2 : expected - 1, ci = fb / bpc;
                                    if (ci < 4) { cwd[ci][fb % bpc] ^= (uint8_t)(1 << (bit % 8)); recovered = 1; }
                                }
                        static _Thread_local ro_suspect sus[4 * 68 * 8];
                        int nsus = 0;
                        for (int c = 0; c < 4; ++c)
                            for (int i = 0; i < bpc * 8 && i < RO_CW_BITS; ++i) {
                                int fbit = c * bpc * 8 + i;
                                if (fbit / 8 >= data_bytes) continue;
                                int chb = cwllr[c][i] < 0, db = (cwd[c][i / 8] >> (i % 8)) & 1;
                                if (chb != db) { sus[nsus].frame_bit = fbit; sus[nsus].abs_llr = fabsf(cwllr[c][i]); nsus++; }
                            }
                        sus_sort(sus, nsus);
                        int ns = nsus < 30 ? nsus : 30;
                        uint16_t sd[30];
                        for (int i = 0; i < ns; ++i) sd[i] = deltas[sus[i].frame_bit];
#define RO_FIX(p) do { int fb_ = (p) / 8, c_ = fb_ / bpc; if (c_ < 4) cwd[c_][fb_ % bpc] ^= (uint8_t)(1 << ((p) % 8)); } while (0)
                        if (!recovered)
                            for (int a = 0; a < ns && !recovered; ++a)
                                for (int b2 = a + 1; b2 < ns && !recovered; ++b2)
                                    if ((uint16_t)(sd[a] ^ sd[b2]) == syn) {
                                        RO_FIX(sus[a].frame_bit); RO_FIX(sus[b2].frame_bit);
                                        int tl = ro_reassemble(cwd, bpc, trial);
                                        if (ro_verify_frame(trial, tl)) recovered = 1;
                                        else { RO_FIX(sus[a].frame_bit); RO_FIX(sus[b2].frame_bit); }
                                    }
                        if (!recovered)
                            for (int a = 0; a < ns && !recovered; ++a)
                                for (int b2 = a + 1; b2 < ns && !recovered; ++b2)
                                    for (int c2 = b2 + 1; c2 < ns && !recovered; ++c2)
                                        if ((uint16_t)(sd[a] ^ sd[b2] ^ sd[c2]) == syn) {
                                            RO_FIX(sus[a].frame_bit); RO_FIX(sus[b2].frame_bit); RO_FIX(sus[c2].frame_bit);
                                            int tl = ro_reassemble(cwd, bpc, trial);
                                            if (ro_verify_frame(trial, tl)) recovered = 1;
                                            else { RO_FIX(sus[a].frame_bit); RO_FIX(sus[b2].frame_bit); RO_FIX(sus[c2].frame_bit); }
                                        }
                        if (!recovered) {
                            int ns4 = ns < 15 ? ns : 15;
                            for (int a = 0; a < ns4 && !recovered; ++a)
                                for (int b2 = a + 1; b2 < ns4 && !recovered; ++b2)
                                    for (int c2 = b2 + 1; c2 < ns4 && !recovered; ++c2)
                                        for (int d2 = c2 + 1; d2 < ns4 && !recovered; ++d2)
                                            if ((uint16_t)(sd[a] ^ sd[b2] ^ sd[c2] ^ sd[d2]) == syn) {
                                                RO_FIX(sus[a].frame_bit); RO_FIX(sus[b2].frame_bit); RO_FIX(sus[c2].frame_bit); RO_FIX(sus[d2].frame_bit);
                                                int tl = ro_reassemble(cwd, bpc, trial);
                                                if (ro_verify_frame(trial, tl)) recovered = 1;
                                                else { RO_FIX(sus[a].frame_bit); RO_FIX(sus[b2].frame_bit); RO_FIX(sus[c2].frame_bit); RO_FIX(sus[d2].frame_bit); }
                                            }
                        }
#undef RO_FIX
                    }
                }
            }
            if (!recovered) {
                static const float rf[4] = { 0.75f, 0.625f, 0.5f, 0.875f };
                for (int at = 0; at < 4 && !recovered; ++at)
                    for (int cw = 0; cw < 4 && !recovered; ++cw) {
                        uint8_t orig[68], dec[81];
                        int it;
                        memcpy(orig, cwd[cw], (size_t)bpc);
                        if (ro_ldpc_decode(&code, cwllr[cw], RO_CW_BITS, max_iter, rf[at], dec, &it) &&
                            memcmp(dec, orig, (size_t)bpc) != 0) {
                            memcpy(cwd[cw], dec, (size_t)bpc);
                            int tl = ro_reassemble(cwd, bpc, trial);
                            if (ro_verify_frame(trial, tl)) recovered = 1;
                            else memcpy(cwd[cw], orig, (size_t)bpc);
                        }
                    }
            }
            if (!recovered) { memset(ok_out, 0, 4); good = 0; }
        }
    }
    for (int cw = 0; cw < 4; ++cw) if (ok_out[cw]) memcpy(data_out + cw * bpc, cwd[cw], (size_t)bpc);
    return g_sort_unpinned ? -1 - good : good;
}

/* ------------------------------------------------------------------ Schmidl-Cox acquisition
 * OFDMDemodulator::searchForSync          src/ofdm/demodulator.cpp:1450-1542
 * Impl::hasMinimumEnergy                  src/ofdm/ofdm_sync.cpp:20-50
 * Impl::toAnalytic                        src/ofdm/ofdm_sync.cpp:56-86   (FFT Hilbert, fft_len == len == 1024)
 * Impl::measureSchmidlCoxCorrelation      src/ofdm/ofdm_sync.cpp:118-163
 * Impl::estimateCoarseCFO                 src/ofdm/ofdm_sync.cpp:230-261
 * Impl::refineLTSTiming                   src/ofdm/ofdm_sync.cpp:386-484
 * LTS passband templates                  src/ofdm/demodulator.cpp:108-141
 * Pinned bit-for-bit against oracle/_ref by oracle/check_against_ref.py. */
static void cox_analytic(const float* s, cf* out) { /* toAnalytic, len 1024 */
    for (int i = 0; i < RO_FFT; ++i) out[i] = cf_mk(s[i], 0.0f);
    ro_fft(out, 0);
    for (int i = 1; i < RO_FFT / 2; ++i) out[i] = cf_mk(out[i].re * 2.0f, out[i].im * 2.0f);
    for (int i = RO_FFT / 2 + 1; i < RO_FFT; ++i) out[i] = cf_mk(0.0f, 0.0f);
    ro_fft(out, 1);
}
static float cox_metric(const float* x, int n, int offset) { /* measureSchmidlCoxCorrelation */
    const int cp = RO_SYM - RO_FFT, half = RO_FFT / 2;
    if (offset + cp + RO_FFT > n) return 0.0f;
    const float* d = x + offset + cp;
    float dc_sum = 0.0f;
    for (int i = 0; i < RO_FFT; ++i) dc_sum += d[i];
    float dc = dc_sum / (float)RO_FFT;
    static float tmp[RO_FFT];
    static cf an[RO_FFT];
    for (int i = 0; i < RO_FFT; ++i) tmp[i] = d[i] - dc;
    cox_analytic(tmp, an);
    cf P = cf_mk(0.0f, 0.0f);
    float R1 = 0.0f, R2 = 0.0f;
    for (int i = 0; i < half; ++i) {
        P = cf_add(P, cf_mul(cf_conj(an[i]), an[i + half]));
        R1 += cf_norm(an[i]);
        R2 += cf_norm(an[i + half]);
    }
    float norm = sqrtf(R1 * R2);
    if (norm < 1e-10f) return 0.0f;
    return cf_abs(P) / norm;
}
static int cox_has_energy(const float* x, int n, int offset, int window, float* nf) { /* hasMinimumEnergy */
    if (offset + window > n) return 0;
    float sum_sq = 0.0f;
    int count = 0;
    for (int i = 0; i < window; i += 16) { float s = x[offset + i]; sum_sq += s * s; ++count; }
    float energy = sum_sq / (float)count;
    if (*nf < 1e-20f) *nf = energy * 0.1f;
    if (energy < *nf) *nf = energy;
    else if (energy < *nf * 3.0f) *nf = (1.0f - 0.01f) * *nf + 0.01f * energy;
    return energy >= *nf * 4.0f;
}
static float cox_coarse_cfo(const float* x, int n, int sync_offset) { /* estimateCoarseCFO */
    const int cp = RO_SYM - RO_FFT, half = RO_FFT / 2;
    int ds = sync_offset + cp;
    if (ds + RO_FFT > n) return 0.0f;
    static cf an[RO_FFT];
    cox_analytic(x + ds, an);
    cf P = cf_mk(0.0f, 0.0f);
    for (int i = 0; i < half; ++i) P = cf_add(P, cf_mul(cf_conj(an[i]), an[i + half]));
    float phase = atan2f(P.im, P.re);
    float cfo = (float)((double)(phase * 48000.0f) / (M_PI * (double)RO_FFT));
    float max_cfo = (float)(48000 / RO_FFT); /* integer division in the reference: uint32_t / size_t */
    return fmaxf_(-max_cfo, fminf_(max_cfo, cfo));
}
void ro_cox_lts_template(const ro_geom* g, float* tI, float* tQ) { /* demodulator.cpp:108-141 */
    static cf f[RO_FFT];
    for (int i = 0; i < RO_FFT; ++i) f[i] = cf_mk(0.0f, 0.0f);
    for (int i = 0; i < g->n_data; ++i) f[g->data_idx[i]] = cf_mk(g->sync_re[i % RO_NCAR], g->sync_im[i % RO_NCAR]);
    for (int i = 0; i < g->n_pilot; ++i) f[g->pilot_idx[i]] = cf_mk(g->pilot_seq[i], 0.0f);
    ro_fft(f, 1);
    const int cp = RO_SYM - RO_FFT;
    ro_nco nco;
    ro_nco_init(&nco, 1500.0f, 48000.0f);
    for (int i = 0; i < RO_SYM; ++i) {
        cf b = (i < cp) ? f[RO_FFT - cp + i] : f[i - cp];
        cf m = cf_mul(b, ro_nco_next(&nco));
        tI[i] = m.re; tQ[i] = m.im;
    }
}
static float cox_lts_corr(const float* x, int n, long long offset, const float* tI, const float* tQ, float eref) {
    if (offset + RO_SYM > n) return 0.0f;
    float cI = 0.0f, cQ = 0.0f, erx = 0.0f;
    for (int i = 0; i < RO_SYM; ++i) {
        float s = x[offset + i];
        cI += s * tI[i];
        cQ += s * tQ[i];
        erx += s * s;
    }
    float mag = sqrtf(cI * cI + cQ * cQ), norm = sqrtf(erx * eref);
    return (norm > 1e-6f) ? mag / norm : 0.0f;
}
/* returns refined LTS start, or -1 for the reference's SIZE_MAX */
static long long cox_refine_lts(const float* x, int n, int coarse_sts, const float* tI, const float* tQ) {
    const int L = RO_SYM, BACK = 3 * L, FWD = L / 2;
    long long coarse = (long long)coarse_sts + 4 * L;
    if (coarse < BACK || coarse + FWD + L > n) return coarse;
    float eref = 0.0f;
    for (int i = 0; i < L; ++i) { eref += tI[i] * tI[i]; eref += tQ[i] * tQ[i]; }
    eref *= 0.5f;
    float best = 0.0f;
    long long best_off = coarse;
    for (int d = -BACK; d <= FWD; ++d) {
        float c = cox_lts_corr(x, n, coarse + d, tI, tQ, eref);
        if (c > best) { best = c; best_off = coarse + d; }
    }
    if (best_off >= L) {
        long long prev = best_off - L;
        if (prev >= coarse - BACK) {
            float pc = cox_lts_corr(x, n, prev, tI, tQ, eref);
            if (pc >= best * 0.92f) { best_off = prev; best = pc; }
        }
    }
    if (best < 0.05f) return -1;
    return best_off;
}
/* out3 = {found, position (first LTS symbol), cfo_hz}; *noise_floor is Impl::noise_floor_energy (0 for a fresh demodulator)
 * and is updated as the reference's member would be. */
int ro_cox_search(const ro_geom* g, const float* x, int n, float threshold, float* noise_floor, float* out3) {
    out3[0] = out3[1] = out3[2] = 0.0f;
    const int L = RO_SYM, total = 6 * L, window = 2 * L;
    if (n < 4000) return 0;
    if (n < total + window) return 0;
    static float tI[RO_SYM], tQ[RO_SYM];
    ro_cox_lts_template(g, tI, tQ);
    float nf = noise_floor ? *noise_floor : 0.0f;
    int found = 0;
    const int search_end = n - total - window;
    for (int i = 0; i < search_end; i += 64) {
        if (!cox_has_energy(x, n, i, window, &nf)) { i += window / 2 - 64; continue; }
        float corr = cox_metric(x, n, i);
        if (corr > threshold) {
            int plateau = 0, peak_pos = i;
            float peak = corr;
            for (int j = 0; j <= 300 && i + j + total < n; j += 8) {
                float c = cox_metric(x, n, i + j);
                if (c >= 0.90f) ++plateau;
                if (c > peak) { peak = c; peak_pos = i + j; }
            }
            if (plateau >= 15) {
                long long lts = cox_refine_lts(x, n, peak_pos, tI, tQ);
                if (lts >= 0) {
                    found = 1;
                    out3[0] = 1.0f; out3[1] = (float)lts; out3[2] = cox_coarse_cfo(x, n, peak_pos);
                    break;
                }
            }
        }
    }
    if (noise_floor) *noise_floor = nf;
    return found;
}
