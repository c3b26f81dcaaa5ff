// Developer probe: which compute units does a CU-masked stream use on this MI355X (SPX mode, 8 XCDs x 32 CUs)?
// Each workgroup records (XCC_ID, SE_ID, CU_ID); the host prints the distinct places per mask.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where(unsigned* out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the CU busy a little so that workgroups spread
    for (int i = 0; i < 2000; ++i) asm volatile("s_nop 7");
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
    const int n = 8192;
    unsigned* d; hipMalloc(&d, n * 8);
    std::vector<unsigned> h(2 * n);
    struct M { const char* name; std::vector<uint32_t> m; };
    std::vector<M> masks = {{"all", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
                            {"low8", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
                            {"bits0-31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
                            {"all-but-low8", {0xffffff00u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
                            {"word7", {0, 0, 0, 0, 0, 0, 0, 0xffffffffu}}};
    for (auto& mk : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mk.m.size(), mk.m.data());
        if (e != hipSuccess) { printf("%s: create failed: %s\n", mk.name, hipGetErrorString(e)); continue; }
        hipMemsetAsync(d, 0xff, n * 8, s);
        hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
        std::set<unsigned> places; std::set<unsigned> xccs;
        for (int i = 0; i < n; ++i) { unsigned hw = h[2 * i], x = h[2 * i + 1] & 0xf; unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; places.insert((x << 16) | (se << 8) | (sh << 4) | cu); xccs.insert(x); }
        printf("%s: %zu distinct (xcc,se,sh,cu) places on %zu XCDs:", mk.name, places.size(), xccs.size());
        int k = 0; for (unsigned p : places) { if (k++ < 12) printf(" x%u.se%u.sh%u.cu%u", p >> 16, (p >> 8) & 0xff, (p >> 4) & 0xf, p & 0xf); }
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
