// ria_amd/csrc/mcdpsk_kernels.hip.h — multi-carrier DPSK demodulator and the HARQ
// chase combine (a19).
//
// mcdpsk_corr / chain / stats / llr kernels: MultiCarrierDPSKDemodulator driven as MCDPSKWaveform::process drives it after an
// external chirp detection (mc_dpsk_waveform.cpp:322-332 -> processGotChirp, multi_carrier_dpsk.hpp:797-896):
// applyCFOCorrection (:901-926, Hilbert FIR src/dsp/filters.cpp:266-317) -> processTraining (:473-505, its
// estimate is reported but not applied, :857-861) -> setReference (:507-518) -> demodulateSoft (:520-736),
// four kernels over a chunk of frames, bit-exact:
//   * every (symbol, carrier) correlation is a left-to-right float sum over 512 samples against the
//     carrier's mixer e^{-j*phase_i}; phase_i is a float recurrence that restarts at every symbol, so the
//     mixer is a per-carrier table built once on the host; one lane per (carrier, 4 consecutive symbols), samples and
//     mixer staged through LDS a chunk of the symbol at a time (mcdpsk_corr_kernel, one workgroup per frame);
//   * the differential chain, the noise/magnitude statistics and the reliability weights are short ordered
//     loops: kernels of their own with one LANE per (carrier, frame) / per frame over item-major tables
//     (mcdpsk_chain_kernel, mcdpsk_stats_kernel) - inside the per-frame workgroup they ran on 1-10 of its 256
//     lanes and were half of the time;
//   * LLRs: one lane per (frame, data symbol, carrier) (mcdpsk_llr_kernel);
//   * CFO correction: the 127-tap Hilbert FIR is one lane per output sample (ordered 127-term sum); the
//     rotation phase is a wrapped float recurrence over the whole frame -> one lane walks it while the
//     other waves filter.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ria_gpu.h"
#include "devmath.h"

namespace ria {

constexpr int kMcSps = 512, kMcTrain = 8, kMcMaxCarriers = 20, kMcHilbertTaps = 127;

struct McArgs {
    const float* samples; long long stride; int frame_samples; int n_frames; int first;   // n_frames: this chunk, from frame `first`
    int nc, bps, spreading;
    const float* cfo; const float* phase0;
    const float2* mixer;       // [nc][512]
    const float* hilbert;      // [127]
    int chunk;                 // samples of a symbol staged per pass of the correlations (mcdpsk_corr_chunk)
    float* ws;                 // [n_frames][2][frame_samples] corrected samples | rotation phases (frames with a CFO)
    // per-chunk tables between the kernels, ITEM-major ([item][n_frames]) so that one-lane-per-frame kernels read them coalesced
    float2* Yg;                // [n_sym * nc][n_frames]  correlations / 512 (training 0, training 1, reference, data ...)
    float* cph; float* cmag; float* pe2;   // [nds * nc][n_frames]
    float* rel;                // [kMcMaxCarriers][n_frames]
    float* scale;              // [n_frames]
    float* llr; int llr_stride;
    ria_mcdpsk_status* status;
};

__device__ __forceinline__ float2 mc_cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

struct McShape { int n, num_rx, nds, n_sym; };
// floats per staged sample row: the frame's symbols rounded up to whole groups of four, never a multiple of 32 (LDS banks)
__host__ __device__ inline int mc_sym_row(int n_sym) { int r = (n_sym + 3) & ~3; if ((r & 31) == 0) r += 4; return r; }
__device__ __forceinline__ McShape mc_shape(const McArgs& A) {
    McShape g;
    g.n = A.frame_samples;
    g.num_rx = (g.n - (kMcTrain + 1) * kMcSps) / kMcSps;
    g.nds = g.num_rx / A.spreading;
    if (g.nds < 1) g.nds = 1;
    g.n_sym = 3 + g.num_rx;                       // training 0, training 1, reference, data...
    return g;
}
// the CFO the demodulator keeps after its correction (:901-926): corrected frames continue with 0
__device__ __forceinline__ bool mc_corrects(float cfo, int n) { return fabs_(cfo) > 0.1f && !(fabs_(cfo) < 0.01f) && n >= 128; }

// ---- kernel 1: one workgroup per frame: CFO correction (:901-926) and the (symbol, carrier) correlations (:931-946)
__global__ __launch_bounds__(256) void mcdpsk_corr_kernel(McArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, f = blockIdx.x;
    const McShape g = mc_shape(A);
    const int nc = A.nc, n = g.n, n_sym = g.n_sym;
    // staging of one chunk of every symbol: SAMPLE-major rows of S4 symbols (a lane reads the same sample of its 4 consecutive
    // symbols as one 16-byte word), and the mixer carrier-major in padded rows (two consecutive values per 16-byte word)
    const int CH = A.chunk, chs = 31 - __builtin_clz(static_cast<unsigned>(CH)), S4 = mc_sym_row(n_sym), mrow = CH + 2;
    float2* Y = reinterpret_cast<float2*>(smem);                           // [n_sym][nc]  running sums
    float* xs_l = reinterpret_cast<float*>(smem + ((n_sym * nc * 8 + 15) & ~15));   // [CH][S4]
    float2* mx_l = reinterpret_cast<float2*>(xs_l + CH * S4);              // [nc][mrow]
    const float* x = A.samples + static_cast<long long>(A.first + f) * A.stride;
    const float cfo = A.cfo ? A.cfo[A.first + f] : 0.0f;
    if (mc_corrects(cfo, n)) {
        float* xc = A.ws + static_cast<size_t>(f) * 2 * n;
        float* ph = xc + n;
        if (tid == 0) {
            const float phase_inc = static_cast<float>(static_cast<double>(-2.0f) * 3.14159265358979323846 * static_cast<double>(cfo) / static_cast<double>(48000.0f));
            float phase = A.phase0 ? A.phase0[A.first + f] : 0.0f;
            for (int i = 0; i < n; ++i) {
                ph[i] = phase;
                phase += phase_inc;
                if (static_cast<double>(phase) > 3.14159265358979323846) phase = static_cast<float>(static_cast<double>(phase) - static_cast<double>(2.0f) * 3.14159265358979323846);
                if (static_cast<double>(phase) < -3.14159265358979323846) phase = static_cast<float>(static_cast<double>(phase) + static_cast<double>(2.0f) * 3.14159265358979323846);
            }
        }
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            float q = 0.0f;
            for (int k = 0; k < kMcHilbertTaps; ++k) {
                const float v = (i - k >= 0) ? x[i - k] : 0.0f;
                q += A.hilbert[k] * v;
            }
            const float re = (i - 63 >= 0) ? x[i - 63] : 0.0f;
            const float p = ph[i];
            xc[i] = re * cosf_glibc(p) - q * sinf_glibc(p);
        }
        __syncthreads();
        x = xc;
    }
    // task (symbol s, carrier c); s = 0,1 training, 2 reference, 3.. data.  A lane owns one carrier and FOUR consecutive
    // symbols: per sample it reads the mixer value once and the four symbols' samples as one word, and advances four
    // independent pairs of left-to-right sums, each pair as one packed multiply and one packed add (the IEEE operations of
    // the halves).  The running sums stay in Y between the chunks, so the order of every sum is the reference's.
    typedef float pk2 __attribute__((ext_vector_type(2)));
    const int T = n_sym * nc, n_grp = (n_sym + 3) / 4, G = n_grp * nc;
    for (int t = tid; t < T; t += 256) Y[t] = make_float2(0.0f, 0.0f);
    for (int i0 = 0; i0 < kMcSps; i0 += CH) {
        __syncthreads();
        for (int idx = tid; idx < n_sym * CH; idx += 256) {                 // CH is a power of two
            const int s = idx >> chs, i = idx & (CH - 1);
            const int sym_index = (s < 2) ? s : (s == 2 ? kMcTrain : kMcTrain + 1 + (s - 3));
            xs_l[i * S4 + s] = x[sym_index * kMcSps + i0 + i];
        }
        for (int idx = tid; idx < (S4 - n_sym) * CH; idx += 256) {          // symbols beyond the frame in the last group: zeros
            const int s = n_sym + (idx >> chs), i = idx & (CH - 1);
            xs_l[i * S4 + s] = 0.0f;
        }
        for (int idx = tid; idx < nc * CH; idx += 256) {
            const int c = idx >> chs, i = idx & (CH - 1);
            mx_l[c * mrow + i] = A.mixer[c * kMcSps + i0 + i];
        }
        __syncthreads();
        for (int gi = tid; gi < G; gi += 256) {
            const int sg = gi / nc, c = gi - sg * nc, s0 = 4 * sg;
            pk2 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 a = (s0 + j < n_sym) ? Y[(s0 + j) * nc + c] : make_float2(0.0f, 0.0f);
                acc[j] = pk2{a.x, a.y};
            }
            const float* xp = xs_l + s0;
            const float2* mp = mx_l + c * mrow;
#pragma unroll 2
            for (int i = 0; i < CH; i += 2) {
                const float4 w = *reinterpret_cast<const float4*>(mp + i);                 // mixer values i, i + 1
                const float4 xa = *reinterpret_cast<const float4*>(xp + i * S4), xb = *reinterpret_cast<const float4*>(xp + (i + 1) * S4);
                acc[0] = acc[0] + pk2{xa.x, xa.x} * pk2{w.x, w.y};
                acc[1] = acc[1] + pk2{xa.y, xa.y} * pk2{w.x, w.y};
                acc[2] = acc[2] + pk2{xa.z, xa.z} * pk2{w.x, w.y};
                acc[3] = acc[3] + pk2{xa.w, xa.w} * pk2{w.x, w.y};
                acc[0] = acc[0] + pk2{xb.x, xb.x} * pk2{w.z, w.w};
                acc[1] = acc[1] + pk2{xb.y, xb.y} * pk2{w.z, w.w};
                acc[2] = acc[2] + pk2{xb.z, xb.z} * pk2{w.z, w.w};
                acc[3] = acc[3] + pk2{xb.w, xb.w} * pk2{w.z, w.w};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (s0 + j < n_sym) Y[(s0 + j) * nc + c] = make_float2(acc[j].x, acc[j].y);
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
        const float2 a = Y[t];
        A.Yg[static_cast<size_t>(t) * A.n_frames + f] = make_float2(fdiv(a.x, static_cast<float>(kMcSps)), fdiv(a.y, static_cast<float>(kMcSps)));
    }
}

// ---- kernel 2: the per-carrier differential chain (setReference :507-518 + demodulateSoft pass 1 :520-560): serial over the
// data symbols, so one LANE per (carrier, frame), frames adjacent
__global__ __launch_bounds__(256) void mcdpsk_chain_kernel(McArgs A) {
    const McShape g = mc_shape(A);
    const int nc = A.nc, bps = A.bps, sp = A.spreading, num_rx = g.num_rx, nds = g.nds;
    const size_t F = static_cast<size_t>(A.n_frames);
    const size_t idx = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= F * nc) return;
    const int c = static_cast<int>(idx / F);
    const size_t f = idx - static_cast<size_t>(c) * F;
    auto Y = [&](int item) { return A.Yg[static_cast<size_t>(item) * F + f]; };
    {
        float2 prev = Y(2 * nc + c);
        const float a = hypotf_glibc(prev.x, prev.y);
        if (a > 0.001f) { const float a2 = hypotf_glibc(prev.x, prev.y); prev = make_float2(fdiv(prev.x, a2), fdiv(prev.y, a2)); }
        else prev = make_float2(1.0f, 0.0f);
        const float kPi = 3.14159274101257324f;   // (float)M_PI
        for (int ds = 0; ds < nds; ++ds) {
            float2 comb = make_float2(0.0f, 0.0f);
            for (int rep = 0; rep < sp; ++rep) {
                const int rs = ds * sp + rep;
                if (rs >= num_rx) break;
                const float2 cur = Y((3 + rs) * nc + c);
                comb.x += cur.x; comb.y += cur.y;
            }
            comb.x = fdiv(comb.x, static_cast<float>(sp)); comb.y = fdiv(comb.y, static_cast<float>(sp));
            const float mag = hypotf_glibc(comb.x, comb.y);
            float2 nrm = (mag > 0.0001f) ? make_float2(fdiv(comb.x, mag), fdiv(comb.y, mag)) : make_float2(1.0f, 0.0f);
            const float2 diff = mc_cmul(nrm, make_float2(prev.x, -prev.y));
            prev = nrm;
            const float phase = atan2f_glibc(diff.y, diff.x);
            float pe;
            if (bps == 2) {
                const float shifted = phase - fdiv(kPi, 4.0f);
                const float nearest = __builtin_roundf(fdiv(shifted, fdiv(kPi, 2.0f)));
                const float ideal = fdiv(nearest * kPi, 2.0f) + fdiv(kPi, 4.0f);
                pe = phase - ideal;
            } else {
                const float nearest = __builtin_roundf(fdiv(phase, kPi));
                pe = phase - nearest * kPi;
            }
            while (pe > kPi) pe -= 2.0f * kPi;
            while (pe < -kPi) pe += 2.0f * kPi;
            const size_t o = static_cast<size_t>(ds * nc + c) * F + f;
            A.cph[o] = phase; A.cmag[o] = mag; A.pe2[o] = pe * pe;
        }
    }
}

// ---- kernel 3: processTraining (:473-505), the ordered statistics of demodulateSoft (:560-640), the DBPSK reliability weights
// and the fading indices (:404-437, :705-733): short serial loops over a frame's symbols, so one LANE per frame
__global__ __launch_bounds__(64) void mcdpsk_stats_kernel(McArgs A) {
    const McShape g = mc_shape(A);
    const int nc = A.nc, bps = A.bps, nds = g.nds;
    const size_t F = static_cast<size_t>(A.n_frames);
    const size_t f = static_cast<size_t>(blockIdx.x) * 64 + threadIdx.x;
    if (f >= F) return;
    auto Y = [&](int item) { return A.Yg[static_cast<size_t>(item) * F + f]; };
    auto cmag = [&](int item) { return A.cmag[static_cast<size_t>(item) * F + f]; };
    const float cfo_in = A.cfo ? A.cfo[A.first + f] : 0.0f;
    const float cfo_after = mc_corrects(cfo_in, g.n) ? 0.0f : cfo_in;
    {
        // processTraining (:473-505)
        float psum = 0.0f;
        for (int c = 0; c < nc; ++c) {
            const float2 s0 = Y(c), s1 = Y(nc + c);
            const float expected_phase = static_cast<float>(static_cast<double>(c) * 3.14159265358979323846 / static_cast<double>(2.0f));
            const float2 ed = make_float2(1.0f * cosf_glibc(expected_phase), 1.0f * sinf_glibc(expected_phase));
            const float2 ad = mc_cmul(s1, make_float2(s0.x, -s0.y));
            const float2 er = mc_cmul(ad, make_float2(ed.x, -ed.y));
            psum += atan2f_glibc(er.y, er.x);
        }
        const float avg = fdiv(psum, static_cast<float>(nc));
        const float symbol_duration = fdiv(static_cast<float>(kMcSps), 48000.0f);
        const float residual = static_cast<float>(static_cast<double>(avg) / (static_cast<double>(2.0f) * 3.14159265358979323846 * static_cast<double>(symbol_duration)));
        // statistics of demodulateSoft (:560-640)
        float noise_sum = 0.0f;
        for (int i = 0; i < nds * nc; ++i) noise_sum += A.pe2[static_cast<size_t>(i) * F + f];
        const int noise_count = nds * nc;
        float mag_sum[kMcMaxCarriers], mag_sq[kMcMaxCarriers];
        for (int c = 0; c < nc; ++c) { mag_sum[c] = 0.0f; mag_sq[c] = 0.0f; }
        int valid = nds;
        {
            for (int ds = 0; ds < nds; ++ds)
                for (int c = 0; c < nc; ++c) { const float m = cmag(ds * nc + c); mag_sum[c] += m; mag_sq[c] += m * m; }
            if (nds >= 4) {
                auto sym_total = [&](int ds) { float t = 0.0f; for (int c = 0; c < nc; ++c) t += cmag(ds * nc + c); return t; };
                float ref_mag = 0.0f;
                for (int s = 0; s < 4; ++s) ref_mag += sym_total(s);
                ref_mag = fdiv(ref_mag, 4.0f);
                if (ref_mag > 0.001f) {
                    const float thr = ref_mag * 0.2f;
                    while (valid > 4 && sym_total(valid - 1) < thr) valid--;
                    if (valid < nds) {
                        for (int c = 0; c < nc; ++c) { mag_sum[c] = 0.0f; mag_sq[c] = 0.0f; }
                        for (int s = 0; s < valid; ++s)
                            for (int c = 0; c < nc; ++c) { const float m = cmag(s * nc + c); mag_sum[c] += m; mag_sq[c] += m * m; }
                    }
                }
            }
        }
        float pnv = (noise_count > 0) ? fdiv(noise_sum, static_cast<float>(noise_count)) : 0.5f;
        pnv = (0.01f < pnv) ? pnv : 0.01f;
        float scale = 2.0f * fsqrt(fdiv(1.0f, pnv));
        scale = (20.0f < scale) ? 20.0f : scale;
        float rel[kMcMaxCarriers];
        for (int c = 0; c < nc; ++c) rel[c] = 1.0f;
        if (bps == 1 && valid > 0) {
            float mean_mag[kMcMaxCarriers];
            float gsum = 0.0f; int gcount = 0;
            for (int c = 0; c < nc; ++c) {
                const float mean = fdiv(mag_sum[c], static_cast<float>(valid));
                mean_mag[c] = mean;
                if (mean > 1e-4f) { gsum += mean; gcount++; }
            }
            const float gmean = (gcount > 0) ? fdiv(gsum, static_cast<float>(gcount)) : 0.0f;
            for (int c = 0; c < nc; ++c) {
                const float mean = mean_mag[c];
                if (mean <= 1e-4f || gmean <= 1e-4f) { rel[c] = 0.12f; continue; }
                const float mean_sq = fdiv(mag_sq[c], static_cast<float>(valid));
                float var = mean_sq - mean * mean;
                var = (0.0f < var) ? var : 0.0f;
                const float cv = fdiv(fsqrt(var), mean + 1e-6f);
                const float ratio = fdiv(mean, gmean);
                const float t1 = (ratio < 1.25f) ? ratio : 1.25f;
                const float mw = (0.10f < t1) ? t1 : 0.10f;
                const float sw = fdiv(1.0f, 1.0f + 1.5f * cv);
                float wd = 1.0f;
                if (ratio < 0.20f) wd = 0.25f; else if (ratio < 0.35f) wd = 0.50f;
                const float w = mw * sw * wd;
                const float t2 = (w < 1.25f) ? w : 1.25f;
                rel[c] = (0.12f < t2) ? t2 : 0.12f;
            }
        }
        A.scale[f] = scale;
        for (int c = 0; c < nc; ++c) A.rel[static_cast<size_t>(c) * F + f] = rel[c];
        // fading indices (:404-437, :705-733)
        float tfi = 0.0f;
        if (valid >= 4) {
            float cvsum = 0.0f; int vc = 0;
            for (int c = 0; c < nc; ++c) {
                const float mean = fdiv(mag_sum[c], static_cast<float>(valid));
                if (mean < 0.001f) continue;
                const float mean_sq = fdiv(mag_sq[c], static_cast<float>(valid));
                float var = mean_sq - mean * mean;
                var = (0.0f < var) ? var : 0.0f;
                cvsum += fdiv(fsqrt(var), mean);
                vc++;
            }
            tfi = (vc > 0) ? fdiv(cvsum, static_cast<float>(vc)) : 0.0f;
        }
        float ffi = 0.0f;
        {
            float cm[kMcMaxCarriers];
            float sum = 0.0f;
            for (int c = 0; c < nc; ++c) { cm[c] = (valid > 0) ? fdiv(mag_sum[c], static_cast<float>(valid)) : 0.0f; sum += cm[c]; }
            const float mean = fdiv(sum, static_cast<float>(nc));
            if (!(mean < 0.001f)) {
                float vs = 0.0f;
                for (int c = 0; c < nc; ++c) { const float d = cm[c] - mean; vs += d * d; }
                ffi = fdiv(fsqrt(fdiv(vs, static_cast<float>(nc))), mean);
            }
        }
        ria_mcdpsk_status st;
        st.cfo_hz = cfo_after; st.fading_index = ffi + 1.0f * tfi; st.freq_fading_index = ffi; st.temporal_fading_index = tfi;
        st.training_cfo_residual = residual; st.n_llr = nds * nc * bps; st.valid_symbols = valid; st.reserved = 0;
        A.status[A.first + f] = st;
    }
}

// ---- kernel 4: LLRs (:650-667), one lane per (frame, data symbol, carrier)
__global__ __launch_bounds__(256) void mcdpsk_llr_kernel(McArgs A) {
    const McShape g = mc_shape(A);
    const int nc = A.nc, bps = A.bps, per = g.nds * nc;
    const size_t F = static_cast<size_t>(A.n_frames);
    const size_t idx = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= F * per) return;
    const size_t f = idx / per;
    const int i = static_cast<int>(idx - f * per), c = i % nc;
    const float phase = A.cph[static_cast<size_t>(i) * F + f];
    const float cs = A.scale[f] * A.rel[static_cast<size_t>(c) * F + f];
    float* out = A.llr + static_cast<size_t>(A.first + f) * A.llr_stride;
    if (bps == 2) {
        const float sb0 = cs * sinf_glibc(phase), sb1 = cs * sinf_glibc(2.0f * phase);
        const float a = (sb0 < 20.0f) ? sb0 : 20.0f, b = (sb1 < 20.0f) ? sb1 : 20.0f;
        out[2 * i] = (-20.0f < a) ? a : -20.0f;
        out[2 * i + 1] = (-20.0f < b) ? b : -20.0f;
    } else {
        const float sb = cs * cosf_glibc(phase);
        const float a = (sb < 20.0f) ? sb : 20.0f;
        out[i] = (-20.0f < a) ? a : -20.0f;
    }
}

// samples of a symbol staged per pass of the correlations: the largest power of two whose padded rows (every symbol of the
// frame + every carrier's mixer) stay within 48 KB, so that three workgroups share a CU
__host__ __device__ inline int mcdpsk_corr_chunk(int nc, int frame_samples) {
    const int n_sym = 3 + (frame_samples - (kMcTrain + 1) * kMcSps) / kMcSps;
    int ch = 128;
    while (ch > 8 && mc_sym_row(n_sym) * ch * 4 + nc * (ch + 2) * 8 > 48 * 1024) ch >>= 1;
    return ch;
}
__host__ __device__ inline int mcdpsk_lds_bytes(int nc, int frame_samples) {
    const int n_sym = 3 + (frame_samples - (kMcTrain + 1) * kMcSps) / kMcSps;
    const int ch = mcdpsk_corr_chunk(nc, frame_samples);
    return ((n_sym * nc * 8 + 15) & ~15) + mc_sym_row(n_sym) * ch * 4 + nc * (ch + 2) * 8 + 64;
}
// floats of per-chunk workspace per frame: the CFO-corrected samples and phases (with_cfo), Yg, cph / cmag / pe2, rel, scale
__host__ __device__ inline size_t mcdpsk_ws_floats_per_frame(int nc, int frame_samples, int spreading, bool with_cfo) {
    const int num_rx = (frame_samples - (kMcTrain + 1) * kMcSps) / kMcSps;
    int nds = num_rx / spreading; if (nds < 1) nds = 1;
    return (with_cfo ? 2 * static_cast<size_t>(frame_samples) : 0) + 2 * static_cast<size_t>(3 + num_rx) * nc + 3 * static_cast<size_t>(nds) * nc + kMcMaxCarriers + 1;
}

// MultiCarrierDPSKModulator on the device (multi_carrier_dpsk.hpp:141-281): training (8 symbols) + reference + data audio
// of a batch of frames, bit-identical to the host builder (host_tables.hpp build_mcdpsk_frame).  One workgroup per frame:
//   * the differential chain prev <- normalise(prev * phase(bits)) is serial over the data symbols but independent per
//     carrier: lane c walks carrier c and parks the transmitted symbols in LDS;
//   * a sample is the left-to-right sum over the carriers of real(symbol * carrier_table[c][i]) / nc: one lane per sample;
//   * the carrier table (cos, sin)(i * phase_inc_c) and the training rotations are per-handle tables built once on the
//     host with the bit-exact libm restatement (the same values for every frame).
struct McModArgs {
    const uint8_t* data; int n_bytes; int n_frames;   // [n_frames][n_bytes] coded bytes (MSB first)
    int nc, bps, spreading, n_data_sym;
    const float2* carrier;    // [nc][512]  (cos, sin)(i * phase_inc_c)
    const float2* train;      // [8][nc]    (cos, sin)(c * sym * pi / 2)
    float* out; long long stride;
};
__global__ __launch_bounds__(256) void mcdpsk_modulate_kernel(McModArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* symb = reinterpret_cast<float2*>(smem);                        // [n_data_sym][nc]
    const int tid = threadIdx.x, f = blockIdx.x, nc = A.nc, bps = A.bps;
    const uint8_t* data = A.data + static_cast<size_t>(f) * A.n_bytes;
    const int n_bits = A.n_bytes * 8;
    if (tid < nc) {
        float pr = 1.0f, pi = 0.0f;                                        // the reference symbol is +1 on every carrier
        for (int ds = 0; ds < A.n_data_sym; ++ds) {
            int sb = 0;
            for (int b = 0; b < bps; ++b) {
                const int bit_idx = (ds * nc + tid) * bps + b;
                const int bit = (bit_idx < n_bits) ? (data[bit_idx >> 3] >> (7 - (bit_idx & 7))) & 1 : 0;
                sb = (sb << 1) | bit;
            }
            float pc;
            if (bps == 2) {
                const double kPi = 3.14159265358979323846;
                pc = (sb == 0) ? static_cast<float>(kPi / 4) : (sb == 1) ? static_cast<float>(3 * kPi / 4) : (sb == 2) ? static_cast<float>(-3 * kPi / 4) : static_cast<float>(-kPi / 4);
            } else {
                pc = sb ? static_cast<float>(3.14159265358979323846) : 0.0f;
            }
            const float dr = 1.0f * cosf_glibc(pc), di = 1.0f * sinf_glibc(pc);
            float cr = pr * dr - pi * di, ci = pr * di + pi * dr;
            const float a = hypotf_glibc(cr, ci);
            cr = fdiv(cr, a); ci = fdiv(ci, a);
            pr = cr; pi = ci;
            symb[ds * nc + tid] = make_float2(cr, ci);
        }
    }
    __syncthreads();
    float* out = A.out + static_cast<long long>(f) * A.stride;
    const float ncf = static_cast<float>(nc);
    const int n_unique = kMcTrain + 1 + A.n_data_sym;                      // symbols before spreading
    for (int idx = tid; idx < n_unique * kMcSps; idx += 256) {
        const int sym = idx / kMcSps, i = idx - sym * kMcSps;
        float acc = 0.0f;
        for (int c = 0; c < nc; ++c) {
            float2 sv;
            if (sym < kMcTrain) sv = A.train[sym * nc + c];
            else if (sym == kMcTrain) sv = make_float2(1.0f, 0.0f);
            else sv = symb[(sym - kMcTrain - 1) * nc + c];
            const float2 cw = A.carrier[c * kMcSps + i];
            acc += fdiv(sv.x * cw.x - sv.y * cw.y, ncf);
        }
        if (sym <= kMcTrain) out[idx] = acc;
        else {
            const int ds = sym - kMcTrain - 1;
            for (int rep = 0; rep < A.spreading; ++rep) out[static_cast<size_t>(kMcTrain + 1 + ds * A.spreading + rep) * kMcSps + i] = acc;
        }
    }
}

// fec::ChaseCache::store for a batch of codeword slots (src/fec/chase_cache.cpp:27-88): first reception
// copies, later ones add (LLR sum), at most 4 combines, decoded slots are left alone
__global__ __launch_bounds__(256) void chase_combine_kernel(float* __restrict__ acc, int32_t* __restrict__ count,
                                                            const uint8_t* __restrict__ decoded, const float* __restrict__ soft,
                                                            int n_cw, uint8_t* __restrict__ stored) {
    const int cw = blockIdx.x;
    if (cw >= n_cw) return;
    const int cnt = count[cw];
    const bool skip = (decoded && decoded[cw]) || cnt >= 4;
    __syncthreads();
    if (!skip) {
        float* a = acc + static_cast<size_t>(cw) * 648;
        const float* s = soft + static_cast<size_t>(cw) * 648;
        for (int i = threadIdx.x; i < 648; i += 256) a[i] = (cnt == 0) ? s[i] : a[i] + s[i];
    }
    if (threadIdx.x == 0) {
        if (!skip) count[cw] = cnt + 1;
        if (stored) stored[cw] = skip ? 0 : 1;
    }
}

// fec::BurstInterleaver (src/fec/burst_interleaver.cpp:8-78): pure permutations, one thread per byte position
__global__ __launch_bounds__(256) void burst_deinterleave_kernel(const float* __restrict__ phys, int stride, int N, int n_groups,
                                                                 float* __restrict__ logical) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // (group, frame f, byte b)
    if (i >= n_groups * N * 324) return;
    const int b = i % 324, f = (i / 324) % N, g = i / (324 * N);
    int pf = f, pb = b;
    if (N >= 2) { const int flat = N * b + f; pf = flat / 324; pb = flat % 324; }
    const float* src = phys + static_cast<size_t>(g * N + pf) * stride + pb * 8;
    float* dst = logical + static_cast<size_t>(g * N + f) * stride + b * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = src[k];
}
__global__ __launch_bounds__(256) void burst_interleave_kernel(const uint8_t* __restrict__ logical, int N, int n_groups,
                                                               uint8_t* __restrict__ phys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_groups * N * 324) return;
    const int b = i % 324, f = (i / 324) % N, g = i / (324 * N);
    int pf = f, pb = b;
    if (N >= 2) { const int flat = N * b + f; pf = flat / 324; pb = flat % 324; }
    phys[static_cast<size_t>(g * N + pf) * 324 + pb] = logical[static_cast<size_t>(g * N + f) * 324 + b];
}

}  // namespace ria
