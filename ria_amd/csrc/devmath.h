// ria_amd/csrc/devmath.h — float32 elementary functions whose results are bit-identical to
// the host libm the reference links against (glibc 2.35, x86-64), for use inside HIP kernels.
//
// Why: the reference estimator branches on thresholds of atan2f/hypotf/sinf results
// (SURVEY.md §7 hard part 1, channel_equalizer.cpp:327, :739, :1432-1446).  OCML's device
// functions differ from glibc in the last ulp, which flips those branches and breaks bit-exact
// decoded payloads.  These restate the *published algorithms* glibc 2.35 uses:
//   sinf/cosf/logf  ARM optimized-routines (double-precision polynomials; the FMA multiarch
//                   variants glibc selects on FMA-capable CPUs, hence the explicit fma()),
//   atanf/atan2f    fdlibm float (s_atanf.c / e_atan2f.c),
//   hypotf          sqrt((double)x*x + (double)y*y) rounded once.
// tests/test_devmath_host.py compiles this header with g++ and compares it against libm over
// hundreds of millions of arguments; tests/test_gpu_math.py does the same for the device build.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RIA_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#include <string.h>
#define RIA_HD static inline
#endif

namespace ria {

RIA_HD uint32_t f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
RIA_HD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
RIA_HD double dfma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fma_rn(a, b, c);
#else
    return fma(a, b, c);
#endif
}
RIA_HD double dsqrt(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_sqrt(a);
#else
    return sqrt(a);
#endif
}
RIA_HD float fsqrt(float a) {
#if defined(__HIP_DEVICE_COMPILE__)
    // NOT __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS that is the 1-ulp native sqrt.
    // __builtin_sqrtf is IEEE-correct under -fhip-fp32-correctly-rounded-divide-sqrt (build flag).
    return __builtin_sqrtf(a);
#else
    return sqrtf(a);
#endif
}
RIA_HD float fdiv(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return a / b;  // IEEE-correct under -fhip-fp32-correctly-rounded-divide-sqrt
#else
    return a / b;
#endif
}
RIA_HD float fabs_(float a) { return u2f(f2u(a) & 0x7fffffffu); }

// ---------------------------------------------------------------- sinf / cosf
// glibc sysdeps/ieee754/flt-32/{s_sinf.c,s_cosf.c,sincosf.h,sincosf_data.c}
struct sincos_poly { double c0, c1, c2, c3, c4, s1, s2, s3; };

RIA_HD float sincos_eval(double x, double x2, bool neg_table, int n) {
    // coefficients of __sincosf_table[0]; table[1] negates the cosine polynomial
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5,
                 c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double t1 = dfma(x2, s3, s2);
        double x7 = x3 * x2;
        double s = dfma(x3, s1, x);
        return (float)dfma(x7, t1, s);
    } else {
        double sg = neg_table ? -1.0 : 1.0;
        double x4 = x2 * x2;
        double t2 = dfma(x2, sg * c4, sg * c3);
        double t1 = dfma(x2, sg * c1, sg * c0);
        double x6 = x4 * x2;
        double c = dfma(x4, sg * c2, t1);
        return (float)dfma(x6, t2, c);
    }
}

RIA_HD uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ff; }

// |y| >= 120: glibc's reduce_large (sincosf.h): 3 x 32-bit words of 4/pi picked by the exponent, a 96-bit
// product, n = the two top bits; the sync correlators mix at 2*pi*f*t with t up to seconds, so they get here.
RIA_HD double sincos_reduce_large(uint32_t xi, int* np) {
    const uint32_t inv_pio4[24] = {0xa2u, 0xa2f9u, 0xa2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
                                   0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu,
                                   0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u,
                                   0x993c4390u, 0x3c439041u};
    const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = static_cast<uint64_t>(static_cast<uint32_t>(xi * arr[0]));
    const uint64_t res1 = static_cast<uint64_t>(xi) * arr[4];
    const uint64_t res2 = static_cast<uint64_t>(xi) * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    *np = static_cast<int>(n);
    return static_cast<double>(static_cast<int64_t>(res0)) * 0x1.921FB54442D18p-62;
}

RIA_HD float sinf_glibc(float y) {
    double x = y;
    if (abstop12(y) < 0x3f4) {  // |y| < pi/4 : abstop12(0x1.921FB6p-1f) = 0x3f4
        double s = x * x;
        if (abstop12(y) < 0x398) return y;  // |y| < 2^-12 : abstop12(0x1p-12f) = 0x398
        return sincos_eval(x, s, false, 0);
    }
    if (abstop12(y) >= 0x42f) {  // |y| >= 120 (finite)
        const uint32_t xi = f2u(y);
        int n;
        x = sincos_reduce_large(xi, &n);
        n += static_cast<int>(xi >> 31);
        const double sgl = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return sincos_eval(x * sgl, x * x, (n & 2) != 0, n - static_cast<int>(xi >> 31));
    }
    const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p0;
    double r = x * hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = dfma(-(double)n, hpi, x);
    double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return sincos_eval(x * sg, x * x, (n & 2) != 0, n);
}
RIA_HD float cosf_glibc(float y) {
    double x = y;
    if (abstop12(y) < 0x3f4) {
        double x2 = x * x;
        if (abstop12(y) < 0x398) return 1.0f;
        return sincos_eval(x, x2, false, 1);
    }
    if (abstop12(y) >= 0x42f) {  // |y| >= 120 (finite)
        const uint32_t xi = f2u(y);
        int n;
        x = sincos_reduce_large(xi, &n);
        const int ns = n + static_cast<int>(xi >> 31);
        const double sgl = ((ns & 3) == 1 || (ns & 3) == 2) ? -1.0 : 1.0;
        return sincos_eval(x * sgl, x * x, (ns & 2) != 0, n ^ 1);
    }
    const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p0;
    double r = x * hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = dfma(-(double)n, hpi, x);
    double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return sincos_eval(x * sg, x * x, (n & 2) != 0, n ^ 1);
}

// ---------------------------------------------------------------- logf (positive normal x)
// glibc sysdeps/ieee754/flt-32/e_logf.c + e_logf_data.c
RIA_HD float logf_glibc(float x) {
    const double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
        {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2}, {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
        {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
        {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},
        {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    const double Ln2 = 0x1.62e42fefa39efp-1;
    uint32_t ix = f2u(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {  // subnormal (zero/inf/nan/negative: not on our path)
        ix = f2u(x * 0x1p23f);
        ix -= 23u << 23;
    }
    uint32_t tmp = ix - 0x3f330000u;
    int i = (tmp >> 19) % 16;
    int k = (int32_t)tmp >> 23;
    uint32_t iz = ix - (tmp & 0xff800000u);
    double invc = T[i][0], logc = T[i][1];
    double z = (double)u2f(iz);
    double r = dfma(z, invc, -1.0);
    double y0 = dfma((double)k, Ln2, logc);
    double r2 = r * r;
    double y = dfma(A1, r, A2);
    y = dfma(A0, r2, y);
    y = dfma(y, r2, y0 + r);
    return (float)y;
}

// ---------------------------------------------------------------- log10f (positive normal x)
// glibc 2.35 sysdeps/ieee754/flt-32/e_log10f.c (fdlibm form on top of logf)
RIA_HD float log10f_glibc(float x) {
    const float ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f;
    int32_t hx = (int32_t)f2u(x);
    int32_t k = (hx >> 23) - 127;
    int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
    float y = (float)(k + i);
    float z = y * log10_2lo + ivln10 * logf_glibc(u2f((uint32_t)hx));
    return z + y * log10_2hi;
}

// ---------------------------------------------------------------- atanf / atan2f (fdlibm float)
RIA_HD float atanf_glibc(float x) {
    // decimal literals as in fdlibm/glibc s_atanf.c (the hex comments there are not all exact)
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                          9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                          4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    int32_t hx = (int32_t)f2u(x), ix = hx & 0x7fffffff, id;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return (hx > 0) ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {  // |x| < 0.4375
        if (ix < 0x31000000) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = fabs_(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = fdiv(2.0f * x - 1.0f, 2.0f + x); }
            else { id = 1; x = fdiv(x - 1.0f, x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = fdiv(x - 1.5f, 1.0f + 1.5f * x); }
            else { id = 3; x = fdiv(-1.0f, x); }
        }
    }
    float z = x * x, w = z * z;
    float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}

RIA_HD float atan2f_glibc(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return atanf_glibc(y);
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
            case 0: case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = atanf_glibc(fabs_(fdiv(y, x)));
    switch (m) {
        case 0: return z;
        case 1: return u2f(f2u(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// ---------------------------------------------------------------- hypotf
RIA_HD float hypotf_glibc(float x, float y) {
    double dx = x, dy = y;
    return (float)dsqrt(dx * dx + dy * dy);
}

}  // namespace ria
