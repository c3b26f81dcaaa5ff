// Developer probe: issue cost of single VALU opcodes on gfx950 at 1..4 waves per SIMD, and the clock the chip holds.
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/valu_rate_probe tools/probes/valu_rate_probe.hip
// One 256*W-thread workgroup per CU (a 100 KB LDS request keeps it alone there): W waves on every SIMD run a loop of 64
// independent instructions of one kind; cycles = s_memtime around the loop; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(1024) void probe(unsigned long long* out, int iters, float seed) {
    extern __shared__ float lds[];
    float a[16]; v2f p[16];
    const float b = seed + threadIdx.x, c = seed * 0.5f;
    const v2f pb = {b, c};
    for (int i = 0; i < 16; ++i) { a[i] = seed * i + threadIdx.x; p[i] = v2f{a[i], a[i] + 1.0f}; }
    if (threadIdx.x == 0) lds[0] = seed;
    __syncthreads();
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define MIN3(i) asm volatile("v_min3_f32 %0, |%0|, |%1|, |%2|" : "+v"(a[i]) : "v"(b), "v"(c));
#define BITOP(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
#define XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pb));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if constexpr (OP == 0) { REP16(ADD) REP16(ADD) REP16(ADD) REP16(ADD) }
        if constexpr (OP == 1) { REP16(MUL) REP16(MUL) REP16(MUL) REP16(MUL) }
        if constexpr (OP == 2) { REP16(PKADD) REP16(PKADD) REP16(PKADD) REP16(PKADD) }
        if constexpr (OP == 3) { REP16(PKMUL) REP16(PKMUL) REP16(PKMUL) REP16(PKMUL) }
        if constexpr (OP == 4) { REP16(MED3) REP16(MED3) REP16(MED3) REP16(MED3) }
        if constexpr (OP == 5) { REP16(MIN3) REP16(MIN3) REP16(MIN3) REP16(MIN3) }
        if constexpr (OP == 6) { REP16(BITOP) REP16(BITOP) REP16(BITOP) REP16(BITOP) }
        if constexpr (OP == 7) { REP16(XOR) REP16(XOR) REP16(XOR) REP16(XOR) }
        if constexpr (OP == 8) { REP16(BFI) REP16(BFI) REP16(BFI) REP16(BFI) }
        if constexpr (OP == 9) { REP16(PKFMA) REP16(PKFMA) REP16(PKFMA) REP16(PKFMA) }
        if constexpr (OP == 10) { REP16(FMA) REP16(FMA) REP16(FMA) REP16(FMA) }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
    float acc = 0.0f;
    for (int i = 0; i < 16; ++i) acc += a[i] + p[i].x + p[i].y;
    if (acc == 12345.678f) lds[1] = acc;   // keep the chains alive
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (static_cast<size_t>(blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64) * 2;
        out[w] = t1 - t0; out[w + 1] = r1 - r0;
    }
}

template <int OP>
void run(const char* name) {
    const int iters = 4000;
    for (int W = 1; W <= 4; ++W) {
        const int threads = 256 * W, blocks = 256;
        unsigned long long* d; hipMalloc(&d, sizeof(unsigned long long) * blocks * (threads / 64) * 2);
        hipFuncSetAttribute(reinterpret_cast<const void*>(probe<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 100 * 1024, 0, d, iters, 1.25f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * (threads / 64) * 2);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, clk;
        for (size_t i = 0; i < h.size(); i += 2) { cyc.push_back(double(h[i])); clk.push_back(double(h[i]) / double(h[i + 1]) * 100.0); }
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        const double c = cyc[cyc.size() / 2];
        // a SIMD issued W waves x iters x 64 instructions in c cycles
        printf("%-12s W=%d  %.2f cycles per wave-instruction per SIMD  (clock %.0f MHz)\n", name, W, c / (double(W) * iters * 64), clk[clk.size() / 2]);
        hipFree(d);
    }
}

int main() {
    run<0>("v_add_f32"); run<1>("v_mul_f32"); run<2>("v_pk_add_f32"); run<3>("v_pk_mul_f32"); run<4>("v_med3_f32"); run<5>("v_min3_f32|abs|");
    run<6>("v_bitop3_b32"); run<7>("v_xor_b32"); run<8>("v_bfi_b32"); run<9>("v_pk_fma_f32"); run<10>("v_fma_f32");
    return 0;
}
