// Compiles ria_amd/csrc/devmath.h for the host and compares it bit-for-bit with the libm the
// reference links (glibc).  argv[1] = stride over float bit patterns (1 = exhaustive).
#include "../../ria_amd/csrc/devmath.h"
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace ria;
int main(int argc, char** argv) {
    uint32_t stride = argc > 1 ? static_cast<uint32_t>(atoi(argv[1])) : 97;
    long bad = 0, n = 0;
    uint32_t lim = 0x7f800000u;   // every finite float: the |x| >= 120 branch (reduce_large) serves the sync mixers
    for (uint64_t u = 0; u < lim; u += stride)
        for (int sg = 0; sg < 2; ++sg) {
            float x = u2f(static_cast<uint32_t>(u) | (sg ? 0x80000000u : 0u));
            bad += f2u(sinf_glibc(x)) != f2u(sinf(x));
            bad += f2u(cosf_glibc(x)) != f2u(cosf(x));
            n += 2;
        }
    for (uint64_t u = 1; u < 0x7f800000u; u += stride) {
        float x = u2f(static_cast<uint32_t>(u));
        bad += f2u(logf_glibc(x)) != f2u(logf(x));
        if (u >= 0x00800000u) { bad += f2u(log10f_glibc(x)) != f2u(log10f(x)); ++n; }
        ++n;
    }
    std::mt19937_64 rng(99);
    for (long i = 0; i < 20000000L; ++i) {
        uint64_t r = rng();
        uint32_t ea = 107 + (r % 41), eb = 107 + ((r >> 8) % 41);
        float a = u2f((static_cast<uint32_t>(r >> 16) & 0x807fffffu) | (ea << 23));
        float b = u2f((static_cast<uint32_t>(r >> 40) & 0x7fffffu) | (eb << 23) | (static_cast<uint32_t>(r >> 63) << 31));
        bad += f2u(atan2f_glibc(a, b)) != f2u(atan2f(a, b));
        bad += f2u(hypotf_glibc(a, b)) != f2u(hypotf(a, b));
        n += 2;
    }
    printf("%ld %ld\n", bad, n);
    return bad != 0;
}
