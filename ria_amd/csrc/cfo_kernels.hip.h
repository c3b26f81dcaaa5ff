// ria_amd/csrc/cfo_kernels.hip.h — the simulator's transmitter frequency offset on the device.
//
// SimulatedChannel::applyTxCFO (tools/cli_simulator.cpp:298-341), the CFO impairment BASELINE.json's config 4
// names ("CFO applied by analytic-signal rotation"): one transmission of n samples is
//   zero padded to N = 2^ceil(log2 n)  ->  ultra::FFT forward (src/dsp/fft.cpp:96-128, radix-2 DIT)
//   -> bins 1..N/2-1 doubled, bins N/2+1..N-1 cleared  ->  inverse FFT (conjugated twiddles, * 1/N)
//   -> out[i] = real(analytic[i] * (cos p_i, sin p_i)),  p_{i+1} = p_i + inc wrapped to (-pi, pi] in float.
// Bit-exact: a float butterfly does not depend on the order in which independent butterflies run, so the
// log2 N stages are regrouped into register-resident passes of up to 6 stages per thread (as the dual-chirp
// correlator does, sync_kernels.hip.h); the twiddle of stage s, index k is the N-point table entry
// (cosf, sinf)((float)(-2 pi k / len)), which equals entry k * 131072 / len of the 131072-point table the chirp
// correlator already keeps (scaling numerator and denominator by a power of two is exact in double), so one
// table serves every size.  The phase recurrence is serial: one thread per buffer walks it into a table.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "devmath.h"

namespace ria {

constexpr int kTxCfoMaxLog = 17;   // the shared twiddle table is the 131072-point one

struct TxCfoArgs {
    const float* in; long long in_stride;
    float* out; long long out_stride;
    int n, log2n, n_buffers;   // n_buffers: this chunk (blockIdx.y)
    const float* cfo_hz;       // [n_buffers]
    float* phase;              // [n_buffers] accumulator in/out (nullable: starts at 0, not returned)
    const float2* tw;          // [65536]
    float2* w1; float2* w2;    // [n_buffers][N]
    float* ph;                 // [n_buffers][n] rotation phase per sample
};

__device__ __forceinline__ bool txcfo_active(const TxCfoArgs& A, int b) { return !(fabs_(A.cfo_hz[b]) < 0.001f); }   // :299
__device__ __forceinline__ int txcfo_bitrev(int v, int bits) { return bits ? static_cast<int>(__brev(static_cast<unsigned>(v)) >> (32 - bits)) : 0; }

// phase walk (:330-339), one thread per buffer
__global__ void txcfo_phase_kernel(TxCfoArgs A) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.n_buffers) return;
    if (!txcfo_active(A, b)) return;
    const float pi_f = 3.14159274101257324f;                       // static_cast<float>(M_PI)
    const float inc = fdiv(2.0f * pi_f * A.cfo_hz[b], 48000.0f);    // all float
    float p = A.phase ? A.phase[b] : 0.0f;
    float* ph = A.ph + static_cast<size_t>(b) * A.n;
    for (int i = 0; i < A.n; ++i) {
        ph[i] = p;
        p += inc;
        if (p > pi_f) p -= 2.0f * pi_f;
        else if (p < -pi_f) p += 2.0f * pi_f;
    }
    if (A.phase) A.phase[b] = p;
}

// One pass = stages S0+1 .. S0+G of the N-point radix-2 DIT transform on 2^G elements per thread.
// mode 0: in place on dst.  mode 1: first forward pass, real zero-padded input (bit-reversed gather) -> dst.
// mode 2: first inverse pass: src with the Hilbert mask (:318-325) -> dst (bit-reversed gather).
template <int G>
__global__ __launch_bounds__(256) void txcfo_fft_pass(TxCfoArgs A, const float2* __restrict__ src_all, float2* __restrict__ dst_all, int S0, int mode, int inv) {
    constexpr int R = 1 << G;
    const int b = blockIdx.y;
    if (!txcfo_active(A, b)) return;
    const int L = A.log2n, N = 1 << L;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (N >> G)) return;
    float2* dst = dst_all + (static_cast<size_t>(b) << L);
    float2 x[R];
    int idx0, stridej, lo = 0;
    if (S0 == 0) {
        idx0 = txcfo_bitrev(t, L - G) * R; stridej = 1;
        if (mode == 1) {
            const float* in = A.in + static_cast<long long>(b) * A.in_stride;
#pragma unroll
            for (int m = 0; m < R; ++m) {
                const int i = m * (N >> G) + t;
                x[txcfo_bitrev(m, G)] = make_float2(i < A.n ? in[i] : 0.0f, 0.0f);
            }
        } else {
            const float2* in = src_all + (static_cast<size_t>(b) << L);
#pragma unroll
            for (int m = 0; m < R; ++m) {
                const int i = m * (N >> G) + t;
                float2 v = in[i];
                if (mode == 2) {
                    if (i >= 1 && i < N / 2) { v.x *= 2.0f; v.y *= 2.0f; }
                    else if (i > N / 2) { v.x = 0.0f; v.y = 0.0f; }
                }
                x[txcfo_bitrev(m, G)] = v;
            }
        }
    } else {
        lo = t & ((1 << S0) - 1);
        idx0 = lo + ((t >> S0) << (S0 + G)); stridej = 1 << S0;
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
                float2 w = A.tw[k << (kTxCfoMaxLog - sidx)];
                if (inv) w.y = -w.y;
                const float2 a = x[j], d = x[j + half];
                const float2 tt = make_float2(w.x * d.x - w.y * d.y, w.x * d.y + w.y * d.x);   // fft.cpp:113-117
                x[j + half] = make_float2(a.x - tt.x, a.y - tt.y);
                x[j] = make_float2(a.x + tt.x, a.y + tt.y);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) dst[idx0 + j * stridej] = x[j];
}

// inverse scale 1/N (fft.cpp:121-126), rotation, real part (:333-335); buffers below the 0.001 Hz gate are copied
__global__ __launch_bounds__(256) void txcfo_rotate_kernel(TxCfoArgs A) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float* in = A.in + static_cast<long long>(b) * A.in_stride;
    float* out = A.out + static_cast<long long>(b) * A.out_stride;
    if (!txcfo_active(A, b)) { out[i] = in[i]; return; }
    const int N = 1 << A.log2n;
    float2 a = A.log2n ? A.w2[(static_cast<size_t>(b) << A.log2n) + i] : make_float2(in[i], 0.0f);
    if (A.log2n) { const float scale = fdiv(1.0f, static_cast<float>(N)); a.x *= scale; a.y *= scale; }
    const float p = A.ph[static_cast<size_t>(b) * A.n + i];
    out[i] = a.x * cosf_glibc(p) - a.y * sinf_glibc(p);
}

}  // namespace ria
