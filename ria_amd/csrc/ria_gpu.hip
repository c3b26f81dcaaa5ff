// ria_amd/csrc/ria_gpu.hip — C-ABI implementation of libria_gpu.so (include/ria_gpu.h).
// Host-side orchestration only: table upload, workspace, kernel launches.  No torch types, no CPU
// fallback: every entry point fails loudly if the HIP device or a launch is unavailable.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ria_gpu.h"
#include "host_tables.hpp"
#include "../host/link_adaptation.hpp"
#include "frame_recovery.hpp"
#include "ldpc_kernels.hip.h"
#include "ldpc_fast.hip.h"
#ifdef RIA_WITH_DUAL_DECODER   // experiment record (two codewords per wave; measured slower): not in the default library
#include "ldpc_dual.hip.h"
#endif
#include "recovery_kernels.hip.h"
#include "demod_kernels.hip.h"
#include "tx_kernels.hip.h"
#include "sync_kernels.hip.h"
#include "mcdpsk_kernels.hip.h"
#include "cox_kernels.hip.h"
#include "cfo_kernels.hip.h"

using namespace ria;

constexpr int kMaxParts = 4;   // ria_gpu_rx_batch overlaps up to this many parts of a batch on internal streams
struct ria_gpu {
    ria_gpu_config cfg{};
    ria_gpu_geometry geo{};
    CarrierPlan plan{};
    LdpcCode code;
    std::string err;
    int device = 0;
    // device tables
    void* d_row_deg = nullptr; void* d_row_var = nullptr; void* d_col_deg = nullptr; void* d_col_slot = nullptr;
    void* d_gather = nullptr; void* d_gather_nochan = nullptr;
    void* d_crc_bit = nullptr; void* d_crc_init = nullptr;
    void* d_zc_ref = nullptr;
    // dual-chirp acquisition: tables (built at first use) and the per-chunk workspace
    void* d_ch_tw = nullptr; void* d_ch_tmpl = nullptr; void* d_ch_tmpl_fft = nullptr; float ch_energy[2] = {0, 0};
    void* d_ch_w1 = nullptr; void* d_ch_w2 = nullptr; void* d_ch_mag = nullptr; void* d_ch_cum = nullptr; void* d_ch_st = nullptr;
    int ch_chunk = 0, ch_outer = 0;
    hipStream_t ch_side = nullptr; hipEvent_t ch_ev[2] = {nullptr, nullptr};   // the time-domain fallback runs beside the FFT path
    // transmitter-CFO impairment (cfo_kernels.hip.h): two complex arrays + the phase table, grown on demand
    void* d_txcfo_ws = nullptr; size_t txcfo_ws_bytes = 0;
    void* d_zc_ws = nullptr; size_t zc_ws_bytes = 0;   // baseband workspace of the long-buffer ZC search
    float* d_chan_nstd = nullptr; int chan_nstd_frames = 0;   // per-frame noise sigma of the reference-identical channel
    // MC-DPSK: mixer tables per carrier count, Hilbert taps, CFO workspace
    std::map<int, void*> d_mc_carrier, d_mc_train;   // device modulator tables per carrier count
    std::map<int, void*> d_mc_mixer; void* d_mc_hilbert = nullptr; void* d_hilbert65 = nullptr; void* d_sync_host = nullptr; size_t sync_host_bytes = 0; void* d_mc_ws = nullptr; size_t mc_ws_floats = 0;
    void* d_twiddle = nullptr; void* d_nco = nullptr;
    // Schmidl-Cox acquisition: LTS passband templates (built at first use) and the metric-table workspace
    void* d_cox_tI = nullptr; void* d_cox_tQ = nullptr; float cox_energy_ref = 0.0f; void* d_cox_ws = nullptr; size_t cox_ws_floats = 0;
    void* d_demod_const = nullptr;
    void* d_demod_ws[kMaxParts] = {nullptr, nullptr, nullptr, nullptr};   // split demodulator: bins / CFO / phase markers of one chunk, per stream slot
    void* d_tx_const = nullptr;
    // workspace
    float* d_llr_ws = nullptr;            // max_batch * llrs_per_frame (fused path)
    FastCode fast{};
    CoreTables ftab;
    void* d_f_row_addr = nullptr; void* d_f_col_addr = nullptr; void* d_f_check_at = nullptr; void* d_f_col_at = nullptr; void* d_f_col_pos = nullptr;
    int wave_lds = 0;
    hipStream_t aux_stream[4] = {nullptr, nullptr, nullptr, nullptr};   // ria_gpu_rx_batch: the parts of a large batch overlap here
    hipEvent_t aux_event[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    DecodeCtl* d_ctl = nullptr;           // cascade work-list control block
    unsigned int* d_entries = nullptr;    // [4 * ws_frames]
    unsigned int* d_best = nullptr;       // [4 * ws_frames]
    unsigned int* d_list1 = nullptr;      // [4 * ws_frames]
    CwResult* d_res = nullptr;            // [4 * ws_frames]
    uint8_t* d_res_bytes = nullptr;       // [4 * ws_frames][5][bytes_per_cw]
    CascadeWin* d_win = nullptr;          // [4 * ws_frames]
    float* d_staged = nullptr;            // [4 * ws_frames][kStageFloats]
    unsigned int* d_l1idx = nullptr;      // [4 * ws_frames]
    unsigned int* d_l1hash = nullptr;     // [4 * ws_frames]
    int ws_frames = 0;
    int split_parts = 0;                  // RIA_OPT_SPLIT_PARTS (0 = library default)
    int dual_decoder = 0;                 // RIA_OPT_DUAL_DECODER: 0 = default (environment RIA_DUAL, else off), 1 = on, -1 = off
    // host-buffer entry points (the single-frame IWaveform adaptor path): one device + one pinned staging block and a
    // stream, kept for the life of the handle, grown on demand - no allocation and no device-wide sync per call
    unsigned char* d_hstage = nullptr; unsigned char* p_hstage = nullptr; size_t hstage_bytes = 0; hipStream_t hstream = nullptr;
    int zc_lds_opted = 0, mc_lds_opted = 0, lts_lds_opted = 0;   // dynamic-LDS opt-ins made on this handle's device
    // CRC-recovery staging (device + pinned host mirrors), sized for ws_frames
    unsigned int* d_rctl = nullptr; unsigned int* d_flagged = nullptr; unsigned int* d_list2 = nullptr; unsigned int* d_stage2 = nullptr; unsigned int* d_overflow = nullptr;
    uint8_t* d_info_c = nullptr; float* d_rows_c = nullptr; uint8_t* d_redec_ok = nullptr; uint8_t* d_redec_bytes = nullptr;
    ria_decode_status* d_st_c = nullptr;
    unsigned int* p_rctl = nullptr; unsigned int* p_flagged = nullptr; uint8_t* p_info_c = nullptr; float* p_rows_c = nullptr;
    uint8_t* p_redec_ok = nullptr; uint8_t* p_redec_bytes = nullptr; ria_decode_status* p_st_c = nullptr;
    int rec_frames = 0, rec_host_frames = 0;
    Crc16Tables crc;
};

namespace {

int fail(ria_gpu_handle h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    // a failed HIP call leaves its code in the runtime's per-thread "last error"; the caller has been told through the
    // return value, so do not let it surface again in whoever calls hipGetLastError next (e.g. the host framework)
    if (code == RIA_ERR_HIP) (void)hipGetLastError();
    return code;
}

#define HIP_TRY(h, expr)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(h, RIA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
hipError_t upload(void** dst, const std::vector<T>& v) {
    hipError_t e = hipMalloc(dst, v.size() * sizeof(T) + 16);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

uint16_t crc16_host(const uint8_t* d, int n, uint16_t init) {  // frame_v2.cpp:115-128
    uint16_t crc = init;
    for (int i = 0; i < n; ++i) {
        crc ^= static_cast<uint16_t>(d[i]) << 8;
        for (int j = 0; j < 8; ++j) crc = (crc & 0x8000) ? static_cast<uint16_t>((crc << 1) ^ 0x1021) : static_cast<uint16_t>(crc << 1);
    }
    return crc;
}

}  // namespace

// ---- rate -> compiled code shape ------------------------------------------------------------------
template <class F>
static void dispatch_shape(int rate, F&& f) {
    switch (rate) {
        case RIA_RATE_1_4: f(ShapeR14{}); break;
        case RIA_RATE_2_3: f(ShapeR23{}); break;
        case RIA_RATE_3_4: f(ShapeR34{}); break;
        case RIA_RATE_5_6: f(ShapeR56{}); break;
        case RIA_RATE_1_3: f(ShapeR13{}); break;
        default: f(ShapeR12{}); break;
    }
}
// the compiled round structure must be the one host_tables.hpp derives from the generated H
static bool shape_fits(int rate, const CoreTables& t, int* wave_lds) {
    bool ok = false;
    dispatch_shape(rate, [&](auto s) {
        using S = decltype(s);
        using I = ShapeInfo<S>;
        ok = static_cast<int>(t.ne.size()) == S::NR && static_cast<int>(t.dv.size()) == S::NC;
        for (int r = 0; ok && r < S::NR; ++r) ok = t.ne[r] == S::ne(r) && t.nm[r] == S::nm(r);
        for (int r = 0; ok && r < S::NC; ++r) ok = t.dv[r] == S::dv(r);
        ok = ok && t.ts == I::TS && t.td == I::TD && t.tot_word == I::tot_word && t.zero_word == I::zero_word && t.dump_word == I::dump_word && t.big_word == I::big_word && t.n_mixed == I::TM && 4 * I::words <= 65536;
        *wave_lds = I::lds_bytes;
    });
    return ok;
}
static hipError_t ensure_decode_ws(ria_gpu_handle h, int n_frames) {
    if (n_frames <= h->ws_frames && h->d_ctl && h->d_l1hash) return hipSuccess;
    h->ws_frames = 0;   // nothing is valid until every allocation below has succeeded
    for (void* p_ : {(void*)h->d_entries, (void*)h->d_best, (void*)h->d_list1, (void*)h->d_res, (void*)h->d_res_bytes, (void*)h->d_win, (void*)h->d_staged, (void*)h->d_l1idx, (void*)h->d_l1hash})
        if (p_) (void)hipFree(p_);
    h->d_entries = h->d_best = h->d_list1 = nullptr; h->d_res = nullptr; h->d_res_bytes = nullptr; h->d_win = nullptr; h->d_staged = nullptr; h->d_l1idx = nullptr; h->d_l1hash = nullptr;
    hipError_t e;
    if (!h->d_ctl && (e = hipMalloc(reinterpret_cast<void**>(&h->d_ctl), kMaxParts * sizeof(DecodeCtl))) != hipSuccess) return e;   // one per stream slot
    if ((e = hipMalloc(reinterpret_cast<void**>(&h->d_entries), static_cast<size_t>(n_frames) * 4 * sizeof(unsigned))) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&h->d_best), static_cast<size_t>(n_frames) * 4 * sizeof(unsigned))) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&h->d_list1), static_cast<size_t>(n_frames) * 4 * sizeof(unsigned))) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&h->d_res), static_cast<size_t>(n_frames) * 4 * sizeof(CwResult))) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&h->d_res_bytes), static_cast<size_t>(n_frames) * 4 * kNumFactors * h->geo.bytes_per_codeword)) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&h->d_win), static_cast<size_t>(n_frames) * 4 * sizeof(CascadeWin))) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&h->d_staged), static_cast<size_t>(n_frames) * 4 * kStageFloats * sizeof(float))) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&h->d_l1idx), static_cast<size_t>(n_frames) * 4 * sizeof(unsigned))) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&h->d_l1hash), static_cast<size_t>(n_frames) * 4 * sizeof(unsigned))) != hipSuccess) return e;
    h->ws_frames = n_frames;
    return hipSuccess;
}
static void set_fast_attributes(int rate, int wb) {
    dispatch_shape(rate, [&](auto s) {
        using S = decltype(s);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fast_primary_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, wb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fast_phase0_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, wb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fast_cascade_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, wb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fast_rows_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, wb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fast_robust_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, wb);
#ifdef RIA_WITH_DUAL_DECODER
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dual_phase0_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, DualInfo<S>::lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dual_cascade_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, DualInfo<S>::lds_bytes);
#endif
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(recovery_fill_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, wb);
    });
}

// ---- CRC-guided false-positive recovery glue (host logic in frame_recovery.hpp) -----------------
template <class F>
static void parallel_for(int n, F&& f) {
    int nt = static_cast<int>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
    if (n < 64 || nt == 1) { for (int i = 0; i < n; ++i) f(i); return; }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&]() { for (;;) { int i0 = next.fetch_add(32); if (i0 >= n) return; for (int i = i0; i < std::min(n, i0 + 32); ++i) f(i); } });
    for (auto& t : th) t.join();
}

static hipError_t ensure_recovery_ws(ria_gpu_handle h, int n_frames, bool host_staging) {
    const size_t n = static_cast<size_t>(n_frames), ib = h->geo.info_bytes_per_frame, bpc = h->geo.bytes_per_codeword;
    hipError_t e;
#define A_TRY(expr) if ((e = (expr)) != hipSuccess) return e
    if (n_frames > h->rec_frames) {
        for (void** p : {(void**)&h->d_rctl, (void**)&h->d_flagged, (void**)&h->d_list2, (void**)&h->d_stage2, (void**)&h->d_overflow}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        h->rec_frames = 0;
        A_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_rctl), 32 * kMaxParts));   // one 32-byte counter block per stream slot
        A_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_overflow), n * 4));
        A_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_flagged), n * 4));
        A_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_list2), n * 16 * 4));
        A_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_stage2), n * 4));
        h->rec_frames = n_frames;
    }
    if (host_staging && n_frames > h->rec_host_frames) {
        for (void** p : {(void**)&h->d_info_c, (void**)&h->d_rows_c, (void**)&h->d_redec_ok, (void**)&h->d_redec_bytes, (void**)&h->d_st_c}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        for (void** p : {(void**)&h->p_rctl, (void**)&h->p_flagged, (void**)&h->p_info_c, (void**)&h->p_rows_c, (void**)&h->p_redec_ok,
                         (void**)&h->p_redec_bytes, (void**)&h->p_st_c}) { if (*p) (void)hipHostFree(*p); *p = nullptr; }
        A_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_info_c), n * ib));
        A_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_rows_c), n * 4 * 648 * 4));
        A_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_redec_ok), n * 16));
        A_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_redec_bytes), n * 16 * bpc));
        A_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_st_c), n * sizeof(ria_decode_status)));
        A_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->p_rctl), 16, hipHostMallocDefault));
        A_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->p_flagged), n * 4, hipHostMallocDefault));
        A_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->p_info_c), n * ib, hipHostMallocDefault));
        A_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->p_rows_c), n * 4 * 648 * 4, hipHostMallocDefault));
        A_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->p_redec_ok), n * 16, hipHostMallocDefault));
        A_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->p_redec_bytes), n * 16 * bpc, hipHostMallocDefault));
        A_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->p_st_c), n * sizeof(ria_decode_status), hipHostMallocDefault));
        h->rec_host_frames = n_frames;
    }
#undef A_TRY
    return hipSuccess;
}

__global__ void recovery_status_gather_kernel(const unsigned int* n_flagged, const unsigned int* flagged,
                                              const ria_decode_status* st, ria_decode_status* st_c) {
    unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < *n_flagged) st_c[q] = st[flagged[q]];
}
__global__ void recovery_scatter_kernel(int nf, const unsigned int* __restrict__ flagged, const uint8_t* __restrict__ info_c,
                                        const ria_decode_status* __restrict__ st_c, int info_bytes,
                                        uint8_t* __restrict__ info_out, ria_decode_status* __restrict__ st_out) {
    int f = blockIdx.x;
    if (f >= nf) return;
    unsigned dst = flagged[f];
    for (int i = threadIdx.x; i < info_bytes; i += blockDim.x)
        info_out[static_cast<size_t>(dst) * info_bytes + i] = info_c[static_cast<size_t>(f) * info_bytes + i];
    if (threadIdx.x == 0) st_out[dst] = st_c[f];
}

// Runs after the decode kernels when RIA_DECODE_CRC_RECOVER is set (frame_v2.cpp:1564-1880).
// The GPU lists the flagged frames and completes the min-sum-factor result table for them (the fallback
// stage re-decodes with 0.75/0.625/0.5/0.875: the same decodes phase 0 makes); then one wave per flagged
// frame runs the CRC-guided searches (recovery_kernels.hip.h).  Nothing leaves the device and nothing
// synchronises.  RIA_RECOVERY_HOST=1 selects the host restatement of the same searches
// (frame_recovery.hpp) instead, which the tests use to cross-check the two implementations.
static int run_crc_recovery_host(ria_gpu_handle h, const FastDecodeArgs& D, hipStream_t s);
static int run_crc_recovery(ria_gpu_handle h, const FastDecodeArgs& D, hipStream_t s, int slot = 0, int ws_off = 0) {
    const char* env = getenv("RIA_RECOVERY_HOST");   // read per call: tests flip it
    const bool on_host = env != nullptr && env[0] == '1';
    if (on_host) return run_crc_recovery_host(h, D, s);
    const int n_frames = D.n_frames;
    hipError_t e0 = ensure_recovery_ws(h, std::max(ws_off + n_frames, h->cfg.max_batch), false);
    if (e0 != hipSuccess) return fail(h, RIA_ERR_HIP, "recovery workspace: %s", hipGetErrorString(e0));
    RecoveryArgs R{};
    R.d = D;
    unsigned int* rctl = h->d_rctl + 8 * slot;
    R.n_flagged = rctl; R.n_list2 = rctl + 1; R.n_stage2 = rctl + 2; R.next_fill = rctl + 3; R.n_overflow = rctl + 4;
    R.flagged = h->d_flagged + ws_off; R.list2 = h->d_list2 + static_cast<size_t>(ws_off) * 16; R.stage2 = h->d_stage2 + ws_off;
    R.overflow = h->d_overflow + ws_off;
#ifdef RIA_DEBUG_STAMPS   // diagnostic builds only (tools/build_variant.sh ... -DRIA_DEBUG_STAMPS): a raw device pointer from the environment
    if (const char* e = getenv("RIA_DEBUG_REC_STAMPS")) R.dbg = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    R.list_units_now = 0;
    if (hipMemsetAsync(rctl, 0, 32, s) != hipSuccess) return fail(h, RIA_ERR_HIP, "hipMemsetAsync failed");
    const int rl = recovery_lds_bytes(h->geo.bytes_per_codeword);
    hipLaunchKernelGGL(recovery_list_kernel, dim3((n_frames + 255) / 256), dim3(256), 0, s, R);
    hipLaunchKernelGGL(recovery_stage1_kernel, dim3(n_frames), dim3(64), rl, s, R);
    dispatch_shape(h->cfg.code_rate, [&](auto sh) {
        using S = decltype(sh);
        hipLaunchKernelGGL(recovery_fill_kernel<S>, dim3(std::min(n_frames * 16, 3072)), dim3(64), h->wave_lds, s, R);
    });
    hipLaunchKernelGGL(recovery_stage2_kernel, dim3(n_frames), dim3(64), rl, s, R);
    // a work-queue fault recorded anywhere in this call (cascade, phase 0, recovery fill) turns every frame into a failure
    hipLaunchKernelGGL(decode_fault_kernel, dim3((n_frames + 3) / 4), dim3(256), 0, s, static_cast<const DecodeCtl*>(D.ctl), D.status, D.info_out,
                       h->geo.bytes_per_codeword, n_frames);
    if (hipGetLastError() != hipSuccess) return fail(h, RIA_ERR_HIP, "recovery kernel launch failed");
    return RIA_OK;
}

static int run_crc_recovery_host(ria_gpu_handle h, const FastDecodeArgs& D, hipStream_t s) {
    static const bool tdbg = getenv("RIA_DEBUG_RECOVERY") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const int n_frames = D.n_frames;
    double tA = now();
    hipError_t e0 = ensure_recovery_ws(h, std::max(n_frames, h->cfg.max_batch), true);
    if (e0 != hipSuccess) return fail(h, RIA_ERR_HIP, "recovery workspace: %s", hipGetErrorString(e0));
    const int bpc = h->geo.bytes_per_codeword, ib = h->geo.info_bytes_per_frame;
    RecoveryArgs R{};
    R.d = D;
    R.n_flagged = h->d_rctl; R.n_list2 = h->d_rctl + 1; R.next_fill = h->d_rctl + 3;
    R.flagged = h->d_flagged; R.list2 = h->d_list2; R.list_units_now = 1;
    R.info_c = h->d_info_c; R.rows_c = h->d_rows_c; R.redec_ok = h->d_redec_ok; R.redec_bytes = h->d_redec_bytes;
#define R_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(h, RIA_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
    R_TRY(hipMemsetAsync(h->d_rctl, 0, 16, s));
    hipLaunchKernelGGL(recovery_list_kernel, dim3((n_frames + 255) / 256), dim3(256), 0, s, R);
    const int wb = h->wave_lds;
    dispatch_shape(h->cfg.code_rate, [&](auto sh) {
        using S = decltype(sh);
        hipLaunchKernelGGL(recovery_fill_kernel<S>, dim3(std::min(n_frames * 16, 3072)), dim3(64), wb, s, R);
    });
    hipLaunchKernelGGL(recovery_gather_kernel, dim3(n_frames), dim3(256), 0, s, R);
    hipLaunchKernelGGL(recovery_status_gather_kernel, dim3((n_frames + 255) / 256), dim3(256), 0, s, h->d_rctl, h->d_flagged,
                       D.status, h->d_st_c);
    R_TRY(hipGetLastError());
    R_TRY(hipMemcpyAsync(h->p_rctl, h->d_rctl, 16, hipMemcpyDeviceToHost, s));
    R_TRY(hipStreamSynchronize(s));
    double tB = now();
    const int nf = static_cast<int>(h->p_rctl[0]);
    if (nf == 0) return RIA_OK;
    R_TRY(hipMemcpyAsync(h->p_flagged, h->d_flagged, static_cast<size_t>(nf) * 4, hipMemcpyDeviceToHost, s));
    R_TRY(hipMemcpyAsync(h->p_info_c, h->d_info_c, static_cast<size_t>(nf) * ib, hipMemcpyDeviceToHost, s));
    R_TRY(hipMemcpyAsync(h->p_rows_c, h->d_rows_c, static_cast<size_t>(nf) * 4 * 648 * 4, hipMemcpyDeviceToHost, s));
    R_TRY(hipMemcpyAsync(h->p_redec_ok, h->d_redec_ok, static_cast<size_t>(nf) * 16, hipMemcpyDeviceToHost, s));
    R_TRY(hipMemcpyAsync(h->p_redec_bytes, h->d_redec_bytes, static_cast<size_t>(nf) * 16 * bpc, hipMemcpyDeviceToHost, s));
    R_TRY(hipMemcpyAsync(h->p_st_c, h->d_st_c, static_cast<size_t>(nf) * sizeof(ria_decode_status), hipMemcpyDeviceToHost, s));
    R_TRY(hipStreamSynchronize(s));
    double tC = now();
    FrameRecovery rec(h->crc, bpc);
    std::atomic<int> n_stage2{0};
    parallel_for(nf, [&](int i) {
        uint8_t cw[4][68];
        std::memset(cw, 0, sizeof(cw));
        uint8_t* inf = h->p_info_c + static_cast<size_t>(i) * ib;
        for (int c = 0; c < 4; ++c) std::memcpy(cw[c], inf + c * bpc, bpc);
        bool good = rec.recover_search(cw, h->p_rows_c + static_cast<size_t>(i) * 4 * 648);
        if (!good) {
            n_stage2++;
            uint8_t rd[4][4][68], rok[4][4];
            for (int at = 0; at < 4; ++at)
                for (int c = 0; c < 4; ++c) {
                    rok[at][c] = h->p_redec_ok[static_cast<size_t>(i) * 16 + at * 4 + c];
                    std::memcpy(rd[at][c], h->p_redec_bytes + (static_cast<size_t>(i) * 16 + at * 4 + c) * bpc, bpc);
                }
            good = rec.recover_fallback(cw, rok, rd);
        }
        ria_decode_status& sn = h->p_st_c[i];
        sn.needs_recovery = 0;
        sn.frame_valid = good ? 1 : 0;
        for (int c = 0; c < 4; ++c) {
            sn.cw_ok[c] = good ? 1 : 0;
            if (good) std::memcpy(inf + c * bpc, cw[c], bpc); else std::memset(inf + c * bpc, 0, bpc);
        }
    });
    double tD = now();
    R_TRY(hipMemcpyAsync(h->d_info_c, h->p_info_c, static_cast<size_t>(nf) * ib, hipMemcpyHostToDevice, s));
    R_TRY(hipMemcpyAsync(h->d_st_c, h->p_st_c, static_cast<size_t>(nf) * sizeof(ria_decode_status), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(recovery_scatter_kernel, dim3(nf), dim3(64), 0, s, nf, h->d_flagged, h->d_info_c, h->d_st_c, ib, D.info_out,
                       D.status);
    R_TRY(hipGetLastError());
    R_TRY(hipStreamSynchronize(s));
#undef R_TRY
    if (tdbg) fprintf(stderr, "[ria_gpu] recovery: frames %d flagged %d stage2 %d | gpu(decode+prep) %.2f ms, D2H %.2f ms, host search %.2f ms, H2D+scatter %.2f ms\n",
                      n_frames, nf, n_stage2.load(), tB - tA, tC - tB, tD - tC, now() - tD);
    return RIA_OK;
}

extern "C" {

int ria_gpu_abi_version(void) { return RIA_GPU_ABI_VERSION; }

void ria_gpu_default_config(ria_gpu_config* cfg) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->abi_version = RIA_GPU_ABI_VERSION;
    cfg->device = 0;
    cfg->modulation = RIA_MOD_QAM16;
    cfg->code_rate = RIA_RATE_1_2;
    cfg->fft_size = 1024;
    cfg->num_carriers = 59;
    cfg->cyclic_prefix = 128;
    cfg->sample_rate = 48000;
    cfg->center_freq = 1500;
    cfg->max_batch = 4096;
}

const char* ria_gpu_last_error(ria_gpu_handle h) { return h ? h->err.c_str() : "null handle"; }

void ria_gpu_destroy(ria_gpu_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    void* ptrs[] = {h->d_row_deg, h->d_row_var, h->d_col_deg, h->d_col_slot, h->d_gather, h->d_gather_nochan,
                    h->d_crc_bit, h->d_crc_init, h->d_zc_ref, h->d_ch_tw, h->d_ch_tmpl, h->d_ch_tmpl_fft, h->d_ch_w1, h->d_ch_w2, h->d_ch_mag, h->d_ch_cum, h->d_ch_st, h->d_twiddle, h->d_nco, h->d_demod_const, h->d_tx_const, h->d_llr_ws,
                    h->d_ctl, h->d_entries, h->d_best, h->d_list1, h->d_res, h->d_res_bytes, h->d_win, h->d_staged, h->d_l1idx, h->d_l1hash,
                    h->d_f_row_addr, h->d_f_col_addr, h->d_f_check_at, h->d_f_col_at, h->d_f_col_pos};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& st_ : h->aux_stream) if (st_) (void)hipStreamDestroy(st_);
    if (h->hstream) (void)hipStreamDestroy(h->hstream);
    if (h->ch_side) (void)hipStreamDestroy(h->ch_side);
    for (auto& ev_ : h->ch_ev) if (ev_) (void)hipEventDestroy(ev_);
    if (h->d_hstage) (void)hipFree(h->d_hstage);
    if (h->p_hstage) (void)hipHostFree(h->p_hstage);
    for (auto& ev_ : h->aux_event) if (ev_) (void)hipEventDestroy(ev_);
    for (auto& kv : h->d_mc_mixer) if (kv.second) (void)hipFree(kv.second);
    for (auto& kv : h->d_mc_carrier) if (kv.second) (void)hipFree(kv.second);
    for (auto& kv : h->d_mc_train) if (kv.second) (void)hipFree(kv.second);
    if (h->d_mc_hilbert) (void)hipFree(h->d_mc_hilbert);
    if (h->d_hilbert65) (void)hipFree(h->d_hilbert65);
    if (h->d_sync_host) (void)hipFree(h->d_sync_host);
    if (h->d_mc_ws) (void)hipFree(h->d_mc_ws);
    for (void* p : {h->d_cox_tI, h->d_cox_tQ, h->d_cox_ws, h->d_txcfo_ws, h->d_zc_ws, static_cast<void*>(h->d_chan_nstd)}) if (p) (void)hipFree(p);
    for (void* p : h->d_demod_ws) if (p) (void)hipFree(p);
    for (void* p : {(void*)h->d_rctl, (void*)h->d_flagged, (void*)h->d_list2, (void*)h->d_stage2, (void*)h->d_info_c, (void*)h->d_rows_c,
                    (void*)h->d_redec_ok, (void*)h->d_redec_bytes, (void*)h->d_st_c, (void*)h->d_overflow}) if (p) (void)hipFree(p);
    for (void* p : {(void*)h->p_rctl, (void*)h->p_flagged, (void*)h->p_info_c, (void*)h->p_rows_c, (void*)h->p_redec_ok,
                    (void*)h->p_redec_bytes, (void*)h->p_st_c}) if (p) (void)hipHostFree(p);
    delete h;
}

int ria_gpu_create(const ria_gpu_config* cfg, ria_gpu_handle* out) {
    if (!cfg || !out) return RIA_ERR_INVALID;
    *out = nullptr;
    if (cfg->abi_version != RIA_GPU_ABI_VERSION) return RIA_ERR_INVALID;
    if (cfg->fft_size != 1024 || cfg->num_carriers != 59 || cfg->cyclic_prefix != 128 ||
        cfg->sample_rate != 48000 || cfg->center_freq != 1500)
        return RIA_ERR_UNSUPPORTED;  // the production OFDM-CHIRP shape (types.hpp:252-268)
    if (bits_per_carrier(cfg->modulation) == 0) return RIA_ERR_UNSUPPORTED;
    if (cfg->code_rate < RIA_RATE_1_4 || cfg->code_rate > RIA_RATE_5_6) return RIA_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device) return RIA_ERR_NO_DEVICE;
    ria_gpu_handle h = new ria_gpu();
    h->cfg = *cfg;
    h->device = cfg->device;
    if (h->cfg.max_batch <= 0) h->cfg.max_batch = 4096;
    if (hipSetDevice(h->device) != hipSuccess) { delete h; return RIA_ERR_NO_DEVICE; }

    h->plan = build_carrier_plan(cfg->modulation, cfg->code_rate);
    h->code = build_ldpc(cfg->code_rate);
    ria_gpu_geometry& g = h->geo;
    g.pilot_spacing = h->plan.spacing;
    g.n_pilots = h->plan.n_pilot;
    g.n_data_carriers = h->plan.n_data;
    g.bits_per_carrier = bits_per_carrier(cfg->modulation);
    g.bits_per_symbol = g.n_data_carriers * g.bits_per_carrier;
    g.n_data_symbols = (kFrameBits + g.bits_per_symbol - 1) / g.bits_per_symbol;
    g.samples_per_symbol = kSym;
    g.frame_samples = (2 + g.n_data_symbols) * kSym;
    g.llrs_per_frame = g.n_data_symbols * g.bits_per_symbol;
    g.info_bits = info_bits_for(cfg->code_rate);
    g.bytes_per_codeword = g.info_bits / 8;
    g.info_bytes_per_frame = 4 * g.bytes_per_codeword;
    g.ldpc_max_iterations = recommended_iterations(cfg->code_rate);
    g.ldpc_edges = h->code.n_edges;
    g.ldpc_k = h->code.k;

#define CREATE_TRY(expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "ria_gpu_create: %s failed: %s\n", #expr, hipGetErrorString(e_));        \
            ria_gpu_destroy(h);                                                                       \
            return RIA_ERR_HIP;                                                                       \
        }                                                                                             \
    } while (0)

    CREATE_TRY(upload(&h->d_gather, build_rx_gather(g.bits_per_symbol, true)));
    CREATE_TRY(upload(&h->d_gather_nochan, build_rx_gather(g.bits_per_symbol, false)));
    {
        // CRC-16 linear decomposition: crc(M) = crc_init[L] ^ XOR_{set bits} crc_bit[distance from end]
        std::vector<uint16_t> bit(4 * 68 * 8 + 16), init(4 * 68 + 2);
        std::vector<uint8_t> z(4 * 68 + 2, 0);
        for (size_t q = 0; q < bit.size(); ++q) {
            std::vector<uint8_t> msg(q / 8 + 1, 0);
            msg[0] = static_cast<uint8_t>(1u << (q % 8));
            bit[q] = crc16_host(msg.data(), static_cast<int>(msg.size()), 0);
        }
        for (size_t L = 0; L < init.size(); ++L) init[L] = crc16_host(z.data(), static_cast<int>(L), 0xFFFF);
        CREATE_TRY(upload(&h->d_crc_bit, bit));
        CREATE_TRY(upload(&h->d_crc_init, init));
    }
    h->crc.build(4 * 68);
    {
        std::vector<float> zc(static_cast<size_t>(4) * kZcRepSamples * 2), re, im;
        for (int r = 0; r < 4; ++r) {
            build_zc_reference(2 * r + 1, re, im);
            for (int i = 0; i < kZcRepSamples; ++i) { zc[(static_cast<size_t>(r) * kZcRepSamples + i) * 2] = re[i]; zc[(static_cast<size_t>(r) * kZcRepSamples + i) * 2 + 1] = im[i]; }
        }
        CREATE_TRY(upload(&h->d_zc_ref, zc));
    }
    CREATE_TRY(upload(&h->d_twiddle, build_twiddles()));
    CREATE_TRY(upload(&h->d_nco, build_nco_table(g.frame_samples)));
    {
        DemodConst dc = build_demod_const(h->plan, cfg->modulation, g);
        std::vector<DemodConst> v(1, dc);
        CREATE_TRY(upload(&h->d_demod_const, v));
        TxConst tc = build_tx_const(h->plan, h->code, cfg->modulation, g);
        std::vector<TxConst> tv(1, tc);
        CREATE_TRY(upload(&h->d_tx_const, tv));
    }
    CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_llr_ws),
                         static_cast<size_t>(h->cfg.max_batch) * g.llrs_per_frame * sizeof(float)));

    {   // the lane/slot assignment against LDS bank conflicts (host_tables.hpp): the layout annealed offline and shipped
        // with the library (validated against this build's H), or - RIA_BANKOPT_MOVES set, or no valid shipped layout -
        // annealed here, once per rate and process
        static std::mutex mu;
        static std::map<std::pair<int, int>, CoreTables> cache;
        const char* mv = getenv("RIA_BANKOPT_MOVES");
        const int moves = mv ? atoi(mv) : -1;              // -1: shipped layout, annealing (1 M moves) as the fallback
        std::lock_guard<std::mutex> lock(mu);
        auto key = std::make_pair(static_cast<int>(cfg->code_rate), moves);
        auto it = cache.find(key);
        if (it == cache.end()) {
            CoreTables t;
            if (moves >= 0 || !load_saved_core_tables(h->code, t)) t = build_core_tables(h->code, moves >= 0 ? moves : 1000000);
            it = cache.emplace(key, std::move(t)).first;
        }
        h->ftab = it->second;
    }
    CREATE_TRY(upload(&h->d_f_row_addr, h->ftab.row_addr));
    CREATE_TRY(upload(&h->d_f_col_addr, h->ftab.col_addr));
    CREATE_TRY(upload(&h->d_f_check_at, h->ftab.check_at));
    CREATE_TRY(upload(&h->d_f_col_at, h->ftab.col_at));
    CREATE_TRY(upload(&h->d_f_col_pos, h->ftab.col_pos));
    h->fast.k = h->code.k; h->fast.m = h->code.m; h->fast.max_iter = g.ldpc_max_iterations; h->fast.bytes_per_cw = g.bytes_per_codeword;
    h->fast.row_addr = static_cast<const uint16_t*>(h->d_f_row_addr);
    h->fast.col_addr = static_cast<const uint16_t*>(h->d_f_col_addr);
    h->fast.check_at = static_cast<const uint16_t*>(h->d_f_check_at);
    h->fast.col_at = static_cast<const uint16_t*>(h->d_f_col_at);
    h->fast.col_pos = static_cast<const uint16_t*>(h->d_f_col_pos);
    if (!shape_fits(cfg->code_rate, h->ftab, &h->wave_lds)) { ria_gpu_destroy(h); return RIA_ERR_UNSUPPORTED; }
    set_fast_attributes(cfg->code_rate, h->wave_lds);
    CREATE_TRY(ensure_decode_ws(h, h->cfg.max_batch));
    CREATE_TRY(demod_set_attributes());
#undef CREATE_TRY
    *out = h;
    return RIA_OK;
}

int ria_gpu_get_geometry(ria_gpu_handle h, ria_gpu_geometry* out) {
    if (!h || !out) return RIA_ERR_INVALID;
    *out = h->geo;
    return RIA_OK;
}

int ria_gpu_set_option(ria_gpu_handle h, int option, int value) {
    if (!h) return RIA_ERR_INVALID;
    if (option == RIA_OPT_SPLIT_PARTS && value >= 0 && value <= kMaxParts) { h->split_parts = value; return RIA_OK; }
    if (option == RIA_OPT_DUAL_DECODER && value >= -1 && value <= 1) {
#ifndef RIA_WITH_DUAL_DECODER
        if (value > 0) return fail(h, RIA_ERR_UNSUPPORTED, "ria_gpu_set_option: this build does not contain the two-codewords-per-wave kernels (-DRIA_WITH_DUAL_DECODER)");
#endif
        h->dual_decoder = value; return RIA_OK;
    }
    return fail(h, RIA_ERR_INVALID, "ria_gpu_set_option: unknown option %d or value %d out of range", option, value);
}

// ------------------------------------------------------------------------------------------------ decode
int ria_gpu_ldpc_decode_batch(ria_gpu_handle h, const float* llr_dev, int n_cw, int max_iterations,
                              float min_sum_factor, uint8_t* out_dev, uint8_t* ok_dev, uint16_t* iters_dev,
                              void* stream) {
    if (!h || !llr_dev || !out_dev || !ok_dev || !iters_dev || n_cw < 0 || max_iterations < 0)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_ldpc_decode_batch: bad argument");
    if (n_cw == 0) return RIA_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    dispatch_shape(h->cfg.code_rate, [&](auto sh) {
        using S = decltype(sh);
        hipLaunchKernelGGL(fast_rows_kernel<S>, dim3(std::min(n_cw, 16384)), dim3(64), h->wave_lds,
                           static_cast<hipStream_t>(stream), h->fast, llr_dev, n_cw, max_iterations, min_sum_factor,
                           out_dev, ok_dev, iters_dev);
    });
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

int ria_gpu_ldpc_decode_robust_batch(ria_gpu_handle h, const float* llr_dev, int n_cw, uint8_t* out_dev, uint8_t* ok_dev,
                                     uint16_t* iters_dev, uint8_t* tries_dev, void* stream) {
    if (!h || !llr_dev || !out_dev || !ok_dev || !iters_dev || n_cw < 0)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_ldpc_decode_robust_batch: bad argument");
    if (n_cw == 0) return RIA_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    dispatch_shape(h->cfg.code_rate, [&](auto sh) {
        using S = decltype(sh);
        hipLaunchKernelGGL(fast_robust_kernel<S>, dim3(std::min(n_cw, 16384)), dim3(64), h->wave_lds,
                           static_cast<hipStream_t>(stream), h->fast, llr_dev, n_cw, out_dev, ok_dev, iters_dev, tries_dev);
    });
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

// slot / ws_off: which control block and which frame range of the per-handle workspace this launch owns
// (ria_gpu_rx_batch runs the two halves of a large batch on two streams: slot 0 at offset 0, slot 1 behind it)
static int launch_decode(ria_gpu_handle h, const float* llr_dev, int llr_stride, int n_frames, uint32_t flags,
                         uint8_t* info_out_dev, ria_decode_status* status_dev, hipStream_t s, int slot = 0, int ws_off = 0) {
    hipError_t e = ensure_decode_ws(h, ws_off + n_frames);
    if (e != hipSuccess) return fail(h, RIA_ERR_HIP, "decode workspace: %s", hipGetErrorString(e));
    FastDecodeArgs A;
    const size_t o4 = static_cast<size_t>(ws_off) * 4;
    if ((e = hipMemsetAsync(h->d_res + o4, 0, static_cast<size_t>(n_frames) * 4 * sizeof(CwResult), s)) != hipSuccess)
        return fail(h, RIA_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    A.c = h->fast;
    A.gather = static_cast<const uint16_t*>((flags & RIA_DECODE_NO_CHANNEL_DEINTERLEAVE) ? h->d_gather_nochan : h->d_gather);
    A.llr = llr_dev;
    A.llr_stride = llr_stride;
    A.n_frames = n_frames;
    A.flags = flags;
    A.info_out = info_out_dev;
    A.status = status_dev;
    A.crc_bit = static_cast<const uint16_t*>(h->d_crc_bit);
    A.crc_init = static_cast<const uint16_t*>(h->d_crc_init);
    A.ctl = h->d_ctl + slot;
    A.entries = h->d_entries + o4;
    A.best = h->d_best + o4;
    A.list1 = h->d_list1 + o4;
    A.res = h->d_res + o4;
    A.res_bytes = h->d_res_bytes + o4 * kNumFactors * h->geo.bytes_per_codeword;
    A.win = h->d_win + o4;
    A.staged = h->d_staged + o4 * kStageFloats;
    A.l1idx = h->d_l1idx + o4;
    A.l1hash = h->d_l1hash + o4;
    if ((e = hipMemsetAsync(A.ctl, 0, sizeof(DecodeCtl), s)) != hipSuccess)
        return fail(h, RIA_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    const int wb = h->wave_lds;
    // RIA_OPT_DUAL_DECODER / RIA_DUAL=1: the retry kernels decode two codewords per wave (ldpc_dual.hip.h) on 2 waves per
    // SIMD instead of one codeword per wave on 3.  Same results (tested); measured slower on the bench workload
    // (DESIGN.md section 4), so it is not the default.
#ifdef RIA_WITH_DUAL_DECODER
    static const bool dual_env = getenv("RIA_DUAL") && getenv("RIA_DUAL")[0] == '1';
    const bool dual = h->dual_decoder > 0 || (h->dual_decoder == 0 && dual_env);
#else
    const bool dual = false;
#endif
    static const int grid_env = getenv("RIA_PERSIST_GRID") ? std::max(64, atoi(getenv("RIA_PERSIST_GRID"))) : 0;
    const int persist_grid = grid_env ? grid_env : (dual ? 2048 : 3072);
    static const bool dbg = getenv("RIA_DEBUG_SYNC") != nullptr;   // stage-by-stage sync + trace on stderr
    auto stage = [&](const char* name) {
        if (!dbg) return;
        hipError_t e2 = hipStreamSynchronize(s);
        fprintf(stderr, "[ria_gpu] %s: %s\n", name, hipGetErrorString(e2));
        fflush(stderr);
    };
    dispatch_shape(h->cfg.code_rate, [&](auto sh) {
        using S = decltype(sh);
        hipLaunchKernelGGL(fast_primary_kernel<S>, dim3(32 * ((n_frames + 7) / 8)), dim3(64), wb, s, A);
        stage("primary");
        hipLaunchKernelGGL(fast_mark_kernel, dim3((n_frames + 255) / 256), dim3(256), 0, s, A);
        stage("mark");
        if (flags & (RIA_DECODE_PHASE0 | RIA_DECODE_PERTURB)) {
            hipLaunchKernelGGL(fast_stage_kernel<S>, dim3(std::min(4 * n_frames, 8192)), dim3(256), 0, s, A);
            stage("stage");
        }
        if (flags & (RIA_DECODE_PHASE0 | RIA_DECODE_PERTURB)) {
#ifdef RIA_WITH_DUAL_DECODER
            if (dual) hipLaunchKernelGGL(dual_phase0_kernel<S>, dim3(std::min(n_frames * 8, persist_grid)), dim3(64), DualInfo<S>::lds_bytes, s, A);
            else
#endif
            hipLaunchKernelGGL(fast_phase0_kernel<S>, dim3(std::min(n_frames * 16, persist_grid)), dim3(64), wb, s, A);
        }
        stage("phase0");
        hipLaunchKernelGGL(fast_chain_kernel, dim3((n_frames + 255) / 256), dim3(256), 0, s, A);
        stage("chain");
        if (flags & RIA_DECODE_PERTURB) {
            // persistent waves over the device-side work list; sized to fill the chip (256 CUs x 12)
#ifdef RIA_WITH_DUAL_DECODER
            if (dual) hipLaunchKernelGGL(dual_cascade_kernel<S>, dim3(persist_grid), dim3(64), DualInfo<S>::lds_bytes, s, A);
            else
#endif
            hipLaunchKernelGGL(fast_cascade_kernel<S>, dim3(persist_grid), dim3(64), wb, s, A);
            stage("cascade");
            hipLaunchKernelGGL(fast_finalize_kernel, dim3(std::min((4 * n_frames + 255) / 256, 1024)), dim3(256), 0, s, A);
            stage("finalize");
        }
    });
    hipLaunchKernelGGL(frame_validate_kernel, dim3((n_frames + 3) / 4), dim3(256), 0, s, info_out_dev,
                       h->geo.bytes_per_codeword, n_frames, A.crc_bit, A.crc_init, status_dev, static_cast<const DecodeCtl*>(A.ctl));
    stage("validate");
    if (hipGetLastError() != hipSuccess) return fail(h, RIA_ERR_HIP, "decode kernel launch failed");
    if (flags & RIA_DECODE_CRC_RECOVER) return run_crc_recovery(h, A, s, slot, ws_off);
    return RIA_OK;
}

int ria_gpu_decode_batch(ria_gpu_handle h, const float* llr_dev, int llr_stride, int n_frames, uint32_t flags,
                         uint8_t* info_out_dev, ria_decode_status* status_dev, void* stream) {
    if (h && n_frames == 0) return RIA_OK;
    if (!h || !llr_dev || !info_out_dev || !status_dev || n_frames < 0 || llr_stride < kFrameBits)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_decode_batch: bad argument (llr_stride must be >= 2592)");
    HIP_TRY(h, hipSetDevice(h->device));
    return launch_decode(h, llr_dev, llr_stride, n_frames, flags, info_out_dev, status_dev, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------------ demod
static int demod_batch_slot(ria_gpu_handle h, const float* samples_dev, const uint64_t* frame_offsets_dev, const ria_frame_meta* meta_dev,
                            int n_frames, float* llr_out_dev, ria_frame_status* status_dev, hipStream_t stream, int slot) {
    DemodArgs A;
    A.k = static_cast<const DemodConst*>(h->d_demod_const);
    A.twiddle = static_cast<const float2*>(h->d_twiddle);
    A.nco = static_cast<const float2*>(h->d_nco);
    A.samples = samples_dev;
    A.offsets = frame_offsets_dev;
    A.meta = meta_dev;
    A.n_frames = n_frames;
    A.llr_out = llr_out_dev;
    A.llr_stride = h->geo.llrs_per_frame;
    A.status = status_dev;
    A.dbg = nullptr;
#ifdef RIA_DEBUG_STAMPS
    if (const char* e = getenv("RIA_DEBUG_DEMOD_STAMPS")) A.dbg = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    static const bool fused = getenv("RIA_DEMOD_FUSED") && getenv("RIA_DEMOD_FUSED")[0] == '1';
    if (fused) launch_demod_fused(A, stream);
    else {
        if (!h->d_demod_ws[slot]) HIP_TRY(h, hipMalloc(&h->d_demod_ws[slot], demod_ws_bytes(2 + h->geo.n_data_symbols)));
        launch_demod(A, h->geo, h->cfg.modulation, h->d_demod_ws[slot], stream);
    }
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

int ria_gpu_demod_batch(ria_gpu_handle h, const float* samples_dev, const uint64_t* frame_offsets_dev,
                        const ria_frame_meta* meta_dev, int n_frames, float* llr_out_dev,
                        ria_frame_status* status_dev, void* stream) {
    if (h && n_frames == 0) return RIA_OK;
    if (!h || !samples_dev || !llr_out_dev || n_frames < 0)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_demod_batch: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    return demod_batch_slot(h, samples_dev, frame_offsets_dev, meta_dev, n_frames, llr_out_dev, status_dev, static_cast<hipStream_t>(stream), 0);
}

int ria_gpu_rx_batch(ria_gpu_handle h, const float* samples_dev, const uint64_t* frame_offsets_dev,
                     const ria_frame_meta* meta_dev, int n_frames, uint32_t flags, uint8_t* info_out_dev,
                     ria_decode_status* decode_status_dev, float* llr_out_dev, ria_frame_status* demod_status_dev,
                     void* stream) {
    if (h && n_frames == 0) return RIA_OK;
    if (!h || !samples_dev || !info_out_dev || !decode_status_dev || n_frames < 0)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_rx_batch: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Demodulate (LLRs stay in the HBM workspace / L2), then decode.  A large chunk is cut in parts (default 3) that run
    // on internal streams: the low-occupancy phases of one part (first decodes, phase 0, finalise, CRC recovery)
    // overlap with the cascade of another.  Results do not depend on the split
    // (tests/test_gpu_modes.py::test_rx_batch_split_modes_are_bit_identical_and_match_the_reference).
    const char* rh = getenv("RIA_RECOVERY_HOST");
    const bool single_stream_only = (rh && rh[0] == '1') || getenv("RIA_DEBUG_SYNC") != nullptr;   // host recovery / stage tracing own slot 0
    int want_parts = h->split_parts;              // per handle (ria_gpu_set_option); 0: environment, else the default 3
    if (want_parts == 0) {
        const char* sp = getenv("RIA_SPLIT_PARTS");
        want_parts = getenv("RIA_NO_SPLIT") ? 1 : sp ? std::max(1, std::min(kMaxParts, atoi(sp))) : 3;
    }
    if (single_stream_only) want_parts = 1;
    for (int done = 0; done < n_frames;) {
        int nb = n_frames - done;
        if (!llr_out_dev && nb > h->cfg.max_batch) nb = h->cfg.max_batch;
        float* llr = llr_out_dev ? llr_out_dev + static_cast<size_t>(done) * h->geo.llrs_per_frame : h->d_llr_ws;
        const int n_parts = (nb >= 4096) ? want_parts : 1;
        {   // grow the workspaces BEFORE anything is in flight: the parts share them
            hipError_t e = ensure_decode_ws(h, nb);
            if (e == hipSuccess && (flags & RIA_DECODE_CRC_RECOVER)) e = ensure_recovery_ws(h, std::max(nb, h->cfg.max_batch), false);
            if (e != hipSuccess) return fail(h, RIA_ERR_HIP, "rx workspace: %s", hipGetErrorString(e));
        }
        if (n_parts > 1 && !h->aux_stream[0]) {
            for (auto& st_ : h->aux_stream) HIP_TRY(h, hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
            for (auto& ev_ : h->aux_event) HIP_TRY(h, hipEventCreateWithFlags(&ev_, hipEventDisableTiming));
        }
        if (n_parts > 1) {
            HIP_TRY(h, hipEventRecord(h->aux_event[0], s));
            for (int part = 0; part < n_parts; ++part) HIP_TRY(h, hipStreamWaitEvent(h->aux_stream[part], h->aux_event[0], 0));
        }
        const int share = (n_parts > 1) ? (((nb + n_parts - 1) / n_parts + 7) & ~7) : nb;
        for (int part = 0; part < n_parts; ++part) {
            const int p0 = part * share, pn = std::min(share, nb - p0);
            if (pn <= 0) break;
            hipStream_t ps = (n_parts > 1) ? h->aux_stream[part] : s;
            const int g0 = done + p0;
            const uint64_t* offs = frame_offsets_dev ? frame_offsets_dev + g0 : nullptr;
            const float* smp = frame_offsets_dev ? samples_dev : samples_dev + static_cast<size_t>(g0) * h->geo.frame_samples;
            float* pl = llr + static_cast<size_t>(p0) * h->geo.llrs_per_frame;
            int rc = demod_batch_slot(h, smp, offs, meta_dev ? meta_dev + g0 : nullptr, pn, pl,
                                      demod_status_dev ? demod_status_dev + g0 : nullptr, ps, part);
            if (rc != RIA_OK) return rc;
            rc = launch_decode(h, pl, h->geo.llrs_per_frame, pn, flags, info_out_dev + static_cast<size_t>(g0) * h->geo.info_bytes_per_frame,
                               decode_status_dev + g0, ps, part, p0);
            if (rc != RIA_OK) return rc;
        }
        if (n_parts > 1) {
            for (int part = 0; part < n_parts; ++part) {
                HIP_TRY(h, hipEventRecord(h->aux_event[1 + part], h->aux_stream[part]));
                HIP_TRY(h, hipStreamWaitEvent(s, h->aux_event[1 + part], 0));
            }
        }
        done += nb;
    }
    return RIA_OK;
}

// staging block of the host-buffer entry points: `bytes` of device memory and as many of pinned host memory
static int ensure_host_stage(ria_gpu_handle h, size_t bytes) {
    if (!h->hstream) HIP_TRY(h, hipStreamCreateWithFlags(&h->hstream, hipStreamNonBlocking));
    if (bytes <= h->hstage_bytes) return RIA_OK;
    HIP_TRY(h, hipStreamSynchronize(h->hstream));
    if (h->d_hstage) (void)hipFree(h->d_hstage);
    if (h->p_hstage) (void)hipHostFree(h->p_hstage);
    h->d_hstage = nullptr; h->p_hstage = nullptr; h->hstage_bytes = 0;
    bytes = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_hstage), bytes));
    HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->p_hstage), bytes, hipHostMallocDefault));
    h->hstage_bytes = bytes;
    return RIA_OK;
}
static inline size_t up256(size_t v) { return (v + 255) & ~size_t(255); }

int ria_gpu_rx_frames_host(ria_gpu_handle h, const float* samples_host, const ria_frame_meta* meta_host, int n_frames,
                           uint32_t flags, uint8_t* info_out_host, ria_decode_status* decode_status_host,
                           float* llr_out_host, ria_frame_status* demod_status_host) {
    const bool demod_only = (flags & RIA_RX_DEMOD_ONLY) != 0;
    if (!h || !samples_host || n_frames <= 0 || (!demod_only && (!info_out_host || !decode_status_host)) || (demod_only && !llr_out_host))
        return fail(h, RIA_ERR_INVALID, "ria_gpu_rx_frames_host: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    const ria_gpu_geometry& g = h->geo;
    const size_t n = static_cast<size_t>(n_frames);
    // layout of the staging block (same offsets on the device and in pinned memory): inputs first, then outputs
    const size_t o_s = 0, b_s = n * g.frame_samples * sizeof(float);
    const size_t o_m = up256(o_s + b_s), b_m = n * sizeof(ria_frame_meta);
    const size_t o_llr = up256(o_m + b_m), b_llr = n * g.llrs_per_frame * sizeof(float);
    const size_t o_info = up256(o_llr + b_llr), b_info = n * g.info_bytes_per_frame;
    const size_t o_ds = up256(o_info + b_info), b_ds = n * sizeof(ria_decode_status);
    const size_t o_fs = up256(o_ds + b_ds), b_fs = n * sizeof(ria_frame_status);
    const size_t total = up256(o_fs + b_fs);
    int rc = ensure_host_stage(h, total);
    if (rc != RIA_OK) return rc;
    unsigned char *D = h->d_hstage, *P = h->p_hstage;
    hipStream_t s = h->hstream;
    std::memcpy(P + o_s, samples_host, b_s);
    if (meta_host) std::memcpy(P + o_m, meta_host, b_m);
    HIP_TRY(h, hipMemcpyAsync(D, P, meta_host ? o_m + b_m : b_s, hipMemcpyHostToDevice, s));   // samples (+ meta) in one copy
    const ria_frame_meta* d_meta = meta_host ? reinterpret_cast<const ria_frame_meta*>(D + o_m) : nullptr;
    if (demod_only)
        rc = ria_gpu_demod_batch(h, reinterpret_cast<const float*>(D + o_s), nullptr, d_meta, n_frames, reinterpret_cast<float*>(D + o_llr),
                                 reinterpret_cast<ria_frame_status*>(D + o_fs), s);
    else
        rc = ria_gpu_rx_batch(h, reinterpret_cast<const float*>(D + o_s), nullptr, d_meta, n_frames, flags, D + o_info,
                              reinterpret_cast<ria_decode_status*>(D + o_ds), reinterpret_cast<float*>(D + o_llr),
                              reinterpret_cast<ria_frame_status*>(D + o_fs), s);
    if (rc != RIA_OK) return rc;
    // outputs back in one copy (the regions wanted are contiguous in the block), one stream sync
    const size_t out0 = (llr_out_host ? o_llr : demod_only ? o_fs : o_info);
    HIP_TRY(h, hipMemcpyAsync(P + out0, D + out0, total - out0, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    if (llr_out_host) std::memcpy(llr_out_host, P + o_llr, b_llr);
    if (!demod_only) {
        std::memcpy(info_out_host, P + o_info, b_info); std::memcpy(decode_status_host, P + o_ds, b_ds);
        if (decode_status_host[0].reserved[1] == kDecodeFaultMarker) return fail(h, RIA_ERR_HIP, "decode work-queue fault: no frame of this call was decoded");
    }
    if (demod_status_host) std::memcpy(demod_status_host, P + o_fs, b_fs);
    return RIA_OK;
}

int ria_gpu_decode_frames_host(ria_gpu_handle h, const float* llr_host, int llr_stride, int n_frames, uint32_t flags,
                               uint8_t* info_out_host, ria_decode_status* status_host) {
    if (!h || !llr_host || !info_out_host || !status_host || n_frames <= 0 || llr_stride < kFrameBits)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_decode_frames_host: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = static_cast<size_t>(n_frames);
    const size_t b_llr = n * llr_stride * sizeof(float), o_info = up256(b_llr), b_info = n * h->geo.info_bytes_per_frame;
    const size_t o_ds = up256(o_info + b_info), b_ds = n * sizeof(ria_decode_status), total = up256(o_ds + b_ds);
    int rc = ensure_host_stage(h, total);
    if (rc != RIA_OK) return rc;
    unsigned char *D = h->d_hstage, *P = h->p_hstage;
    hipStream_t s = h->hstream;
    std::memcpy(P, llr_host, b_llr);
    HIP_TRY(h, hipMemcpyAsync(D, P, b_llr, hipMemcpyHostToDevice, s));
    rc = ria_gpu_decode_batch(h, reinterpret_cast<const float*>(D), llr_stride, n_frames, flags, D + o_info,
                              reinterpret_cast<ria_decode_status*>(D + o_ds), s);
    if (rc != RIA_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(P + o_info, D + o_info, total - o_info, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    std::memcpy(info_out_host, P + o_info, b_info);
    std::memcpy(status_host, P + o_ds, b_ds);
    if (status_host[0].reserved[1] == kDecodeFaultMarker) return fail(h, RIA_ERR_HIP, "decode work-queue fault: no frame of this call was decoded");
    return RIA_OK;
}

// ------------------------------------------------------------------------------------------------ TX / channel
int ria_gpu_make_frames(ria_gpu_handle h, uint64_t seed, int first_seq, int n_frames, uint8_t* info_out_dev,
                        void* stream) {
    if (!h || !info_out_dev || n_frames < 0) return fail(h, RIA_ERR_INVALID, "ria_gpu_make_frames: bad argument");
    if (n_frames == 0) return RIA_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    launch_make_frames(static_cast<const TxConst*>(h->d_tx_const), static_cast<const uint16_t*>(h->d_crc_bit),
                       static_cast<const uint16_t*>(h->d_crc_init), seed, first_seq, n_frames, h->geo, info_out_dev,
                       static_cast<hipStream_t>(stream));
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

int ria_gpu_tx_batch(ria_gpu_handle h, const uint8_t* info_dev, int n_frames, float peak_normalize,
                     float* samples_out_dev, void* stream) {
    if (!h || !info_dev || !samples_out_dev || n_frames < 0) return fail(h, RIA_ERR_INVALID, "ria_gpu_tx_batch: bad argument");
    if (n_frames == 0) return RIA_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    launch_tx(static_cast<const TxConst*>(h->d_tx_const), static_cast<const float2*>(h->d_twiddle),
              static_cast<const float2*>(h->d_nco), info_dev, n_frames, peak_normalize, h->geo, samples_out_dev,
              static_cast<hipStream_t>(stream));
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

int ria_gpu_channel_batch(ria_gpu_handle h, int kind, float snr_db, uint64_t seed, uint64_t first_frame,
                          float* samples_dev, int n_frames, void* stream) {
    if (!h || !samples_dev || n_frames < 0 || kind < 0 || kind > 4)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_channel_batch: bad argument");
    if (n_frames == 0) return RIA_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    launch_channel(kind, snr_db, seed, first_frame, samples_dev, n_frames, h->geo.frame_samples,
                   static_cast<hipStream_t>(stream));
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

// per-frame workspace of channel_power_kernel, grown on demand (a growth waits for the stream's earlier users of the old block)
static int chan_ws(ria_gpu_handle h, int n_frames, hipStream_t s) {
    if (n_frames <= h->chan_nstd_frames) return RIA_OK;
    if (h->d_chan_nstd) { HIP_TRY(h, hipStreamSynchronize(s)); (void)hipFree(h->d_chan_nstd); }
    h->d_chan_nstd = nullptr; h->chan_nstd_frames = 0;
    const int cap = std::max(n_frames, 4096);
    HIP_TRY(h, hipMalloc(&h->d_chan_nstd, static_cast<size_t>(cap) * sizeof(float)));
    h->chan_nstd_frames = cap;
    return RIA_OK;
}

int ria_gpu_channel_exact_batch(ria_gpu_handle h, int kind, float snr_db, uint32_t seed, uint64_t first_frame,
                                float* samples_dev, int64_t stride, int frame_samples, int n_frames, void* stream) {
    if (!h || n_frames < 0 || kind < 0 || kind > 4 || frame_samples < 0 || stride < frame_samples)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_channel_exact_batch: bad argument");
    if (n_frames == 0 || frame_samples == 0) return RIA_OK;
    if (!samples_dev) return fail(h, RIA_ERR_INVALID, "ria_gpu_channel_exact_batch: null samples");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = chan_ws(h, n_frames, static_cast<hipStream_t>(stream))) return rc;
    launch_channel_exact(kind, snr_db, seed, first_frame, samples_dev, stride, frame_samples, n_frames, static_cast<hipStream_t>(stream), h->d_chan_nstd);
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

int ria_gpu_channel_exact_seeded_batch(ria_gpu_handle h, int kind, float snr_db, const uint32_t* seeds_dev, float* samples_dev,
                                       int64_t stride, int frame_samples, int n_frames, void* stream) {
    if (!h || n_frames < 0 || kind < 0 || kind > 4 || frame_samples < 0 || stride < frame_samples)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_channel_exact_seeded_batch: bad argument");
    if (n_frames == 0 || frame_samples == 0) return RIA_OK;
    if (!samples_dev || !seeds_dev) return fail(h, RIA_ERR_INVALID, "ria_gpu_channel_exact_seeded_batch: null pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = chan_ws(h, n_frames, static_cast<hipStream_t>(stream))) return rc;
    launch_channel_exact(kind, snr_db, 0u, 0u, samples_dev, stride, frame_samples, n_frames, static_cast<hipStream_t>(stream), h->d_chan_nstd, seeds_dev);
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

int ria_gpu_channel_exact_cfo_batch(ria_gpu_handle h, int kind, float snr_db, const uint32_t* seeds_dev, const float* cfo_hz_dev,
                                    float random_cfo_max_hz, float* actual_cfo_out_dev, float* samples_dev, int64_t stride,
                                    int frame_samples, int n_frames, void* stream) {
    if (!h || n_frames < 0 || kind < 0 || kind > 4 || frame_samples < 0 || stride < frame_samples || !(random_cfo_max_hz >= 0.0f))
        return fail(h, RIA_ERR_INVALID, "ria_gpu_channel_exact_cfo_batch: bad argument");
    if (n_frames == 0 || frame_samples == 0) return RIA_OK;
    if (!samples_dev || !seeds_dev) return fail(h, RIA_ERR_INVALID, "ria_gpu_channel_exact_cfo_batch: null pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = chan_ws(h, n_frames, static_cast<hipStream_t>(stream))) return rc;
    launch_channel_exact(kind, snr_db, 0u, 0u, samples_dev, stride, frame_samples, n_frames, static_cast<hipStream_t>(stream), h->d_chan_nstd, seeds_dev,
                         cfo_hz_dev, 0.0f, random_cfo_max_hz, actual_cfo_out_dev);
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

// ------------------------------------------------------------------------------------------------ debug
__global__ void debug_math_kernel(int op, const float* a, const float* b, int n, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, r;
    switch (op) {
        case 0: r = sinf_glibc(x); break;
        case 1: r = cosf_glibc(x); break;
        case 2: r = logf_glibc(x); break;
        case 3: r = atan2f_glibc(x, y); break;
        case 4: r = hypotf_glibc(x, y); break;
        case 5: r = fdiv(x, y); break;
        default: r = fsqrt(x); break;
    }
    out[i] = r;
}

int ria_gpu_sync_zc_batch(ria_gpu_handle h, const float* samples_dev, int64_t stride, int buf_len, int n_buffers,
                          float threshold, uint32_t root_mask, const float* known_cfo_dev, ria_zc_result* out_dev,
                          void* stream) {
    if (!h) return RIA_ERR_INVALID;
    if (n_buffers == 0) return RIA_OK;
    if (!samples_dev || !out_dev || n_buffers < 0 || buf_len < 0 || buf_len > kZcMaxBuf || stride < buf_len)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_sync_zc_batch: bad arguments");
    ZcArgs A{};
    A.samples = samples_dev; A.stride = stride; A.buf_len = buf_len; A.n_buffers = n_buffers; A.threshold = threshold;
    A.root_mask = root_mask & 15u; A.known_cfo = known_cfo_dev; A.ref = static_cast<const float2*>(h->d_zc_ref); A.out = out_dev;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (buf_len <= kZcLdsBuf) {   // the mixed-down buffer fits one workgroup's LDS
        const int lds = buf_len * static_cast<int>(sizeof(float2)) + 4 * static_cast<int>(sizeof(ZcRootOut));
        if (lds > h->zc_lds_opted) {   // the opt-in is a per-device attribute: kept per handle (= per device), not per process
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(zc_detect_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            h->zc_lds_opted = lds;
        }
        hipLaunchKernelGGL(zc_detect_kernel<true>, dim3(n_buffers), dim3(256), lds, s, A);
    } else {                      // long search windows: baseband in a global workspace, chunks of buffers under 256 MiB
        const size_t per = static_cast<size_t>(buf_len) * sizeof(float2);
        const int chunk = static_cast<int>(std::min<size_t>(n_buffers, std::max<size_t>(1, (size_t(256) << 20) / per)));
        if (per * chunk > h->zc_ws_bytes) {
            if (h->d_zc_ws) { HIP_TRY(h, hipStreamSynchronize(s)); (void)hipFree(h->d_zc_ws); }
            h->d_zc_ws = nullptr; h->zc_ws_bytes = 0;
            HIP_TRY(h, hipMalloc(&h->d_zc_ws, per * chunk));
            h->zc_ws_bytes = per * chunk;
        }
        A.bb_ws = static_cast<float2*>(h->d_zc_ws);
        for (int first = 0; first < n_buffers; first += chunk) {
            A.samples = samples_dev + static_cast<int64_t>(first) * stride; A.n_buffers = std::min(chunk, n_buffers - first);
            A.known_cfo = known_cfo_dev ? known_cfo_dev + first : nullptr; A.out = out_dev + first;
            hipLaunchKernelGGL(zc_detect_kernel<false>, dim3(A.n_buffers), dim3(256), 4 * static_cast<int>(sizeof(ZcRootOut)), s, A);
        }
    }
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

int ria_gpu_zc_preamble(ria_gpu_handle h, int root, float* out_host, int max_n) {
    if (!h || !out_host) return RIA_ERR_INVALID;
    std::vector<float> p = build_zc_preamble(root);
    if (static_cast<int>(p.size()) > max_n) return -static_cast<int>(p.size());
    std::memcpy(out_host, p.data(), p.size() * sizeof(float));
    return static_cast<int>(p.size());
}

// forward or inverse 131072-point FFT of the active buffers of a chunk (see sync_kernels.hip.h)
static void chirp_fft_forward(const ChirpArgs& A, int chunk, hipStream_t s, bool real_input, bool product) {
    // 17 radix-2 stages as 6 + 6 + 5 register-resident butterfly networks (64 / 64 / 32 points per thread): three trips over the
    // 1 MiB per buffer instead of the four of 4 + 4 + 4 + 5 (measured: 83 k -> 96 k preambles/s)
    const dim3 g64(kChFft / 64 / 256, chunk), g32(kChFft / 32 / 256, chunk), blk(256);
    if (real_input) hipLaunchKernelGGL((chirp_fft_pass<6, 0, 1, false>), g64, blk, 0, s, A, static_cast<const float2*>(nullptr), A.w1);
    else hipLaunchKernelGGL((chirp_fft_pass<6, 0, 2, false>), g64, blk, 0, s, A, static_cast<const float2*>(A.w2), A.w1);
    hipLaunchKernelGGL((chirp_fft_pass<6, 6, 0, false>), g64, blk, 0, s, A, static_cast<const float2*>(nullptr), A.w1);
    if (product) hipLaunchKernelGGL((chirp_fft_pass<5, 12, 3, false>), g32, blk, 0, s, A, static_cast<const float2*>(nullptr), A.w1);
    else hipLaunchKernelGGL((chirp_fft_pass<5, 12, 0, false>), g32, blk, 0, s, A, static_cast<const float2*>(nullptr), A.w1);
}
static void chirp_fft_inverse_mag(const ChirpArgs& A, int chunk, hipStream_t s) {
    const dim3 g64(kChFft / 64 / 256, chunk), g32(kChFft / 32 / 256, chunk), blk(256);
    (void)hipMemsetAsync(A.best, 0, static_cast<size_t>(chunk) * sizeof(unsigned long long), s);
    hipLaunchKernelGGL((chirp_fft_pass<6, 0, 2, true>), g64, blk, 0, s, A, static_cast<const float2*>(A.w1), A.w2);
    hipLaunchKernelGGL((chirp_fft_pass<6, 6, 0, true>), g64, blk, 0, s, A, static_cast<const float2*>(nullptr), A.w2);
    hipLaunchKernelGGL((chirp_fft_pass<5, 12, 4, true>), g32, blk, 0, s, A, static_cast<const float2*>(nullptr), A.w2);
}
__global__ void chirp_template_stage_kernel(const float* tmpl, int down, float2* dst, ChirpBufState* st) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { st[0].active = 1; st[0].win_start = 0; st[0].win_len = kChFft; }
    if (i >= kChFft) return;
    // complex template cos + j*sin, zero padded (chirp_sync.hpp:589-594)
    dst[i] = (i < kChLen) ? make_float2(tmpl[(2 * down + 1) * kChLen + i], tmpl[(2 * down) * kChLen + i]) : make_float2(0.f, 0.f);
}
__global__ void chirp_template_conj_kernel(const float2* src, float2* dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kChFft) dst[i] = make_float2(src[i].x, -src[i].y);
}

static int chirp_prepare(ria_gpu_handle h, int chunk, int outer, hipStream_t s) {
    hipError_t e;
#define C_TRY(expr) if ((e = (expr)) != hipSuccess) return fail(h, RIA_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e))
    if (chunk > h->ch_chunk) {
        for (void** p : {&h->d_ch_w1, &h->d_ch_w2, &h->d_ch_mag}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        const size_t c = static_cast<size_t>(chunk);
        C_TRY(hipMalloc(&h->d_ch_w1, c * kChFft * sizeof(float2)));
        C_TRY(hipMalloc(&h->d_ch_w2, c * kChFft * sizeof(float2)));
        C_TRY(hipMalloc(&h->d_ch_mag, c * sizeof(unsigned long long)));   // packed first-maximum keys
        h->ch_chunk = chunk;
    }
    if (outer > h->ch_outer) {
        for (void** p : {&h->d_ch_cum, &h->d_ch_st}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        C_TRY(hipMalloc(&h->d_ch_cum, static_cast<size_t>(outer) * (kChFft + 1) * sizeof(float)));
        C_TRY(hipMalloc(&h->d_ch_st, static_cast<size_t>(outer) * sizeof(ChirpBufState)));
        h->ch_outer = outer;
    }
    if (!h->d_ch_tmpl_fft) {
        ChirpTables t = build_chirp_tables();
        if (!h->d_ch_tw) C_TRY(upload(&h->d_ch_tw, t.tw));
        C_TRY(upload(&h->d_ch_tmpl, t.tmpl));
        h->ch_energy[0] = t.energy[0]; h->ch_energy[1] = t.energy[1];
        C_TRY(hipMalloc(&h->d_ch_tmpl_fft, static_cast<size_t>(2) * kChFft * sizeof(float2)));
        // conj(FFT(template)) with the same butterflies the signal goes through (chirp_sync.hpp:573-623)
        ChirpArgs A{};
        A.n_buffers = 1; A.tw = static_cast<const float2*>(h->d_ch_tw); A.w1 = static_cast<float2*>(h->d_ch_w1);
        A.w2 = static_cast<float2*>(h->d_ch_w2); A.st = static_cast<ChirpBufState*>(h->d_ch_st);
        for (int d = 0; d < 2; ++d) {
            hipLaunchKernelGGL(chirp_template_stage_kernel, dim3(kChFft / 256), dim3(256), 0, s, static_cast<const float*>(h->d_ch_tmpl), d, A.w2, A.st);
            chirp_fft_forward(A, 1, s, false, false);
            hipLaunchKernelGGL(chirp_template_conj_kernel, dim3(kChFft / 256), dim3(256), 0, s, static_cast<const float2*>(A.w1),
                               static_cast<float2*>(h->d_ch_tmpl_fft) + static_cast<size_t>(d) * kChFft);
        }
        C_TRY(hipGetLastError());
    }
#undef C_TRY
    return RIA_OK;
}

int ria_gpu_sync_chirp_batch(ria_gpu_handle h, const float* samples_dev, int64_t stride, int buf_len, int n_buffers,
                             float threshold, ria_chirp_result* out_dev, void* stream) {
    if (!h) return RIA_ERR_INVALID;
    if (n_buffers == 0) return RIA_OK;
    if (!samples_dev || !out_dev || n_buffers < 0 || buf_len < 0 || stride < buf_len)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_sync_chirp_batch: bad arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Two levels of chunking.  The FFT workspace (2.5 MiB per buffer) is sized for 64 buffers so that it stays
    // within the 256 MiB Infinity Cache across the eight passes of a transform pair.  The serial pieces (energy
    // cumsum: one wave per buffer; time-domain fallback) are latency-bound and want as many buffers in flight as
    // possible, so they run once per OUTER chunk of up to 2048 buffers (0.5 MiB of running sums each).
    const int chunk = std::min(n_buffers, 64), outer = std::min(n_buffers, 2048);
    int rc = chirp_prepare(h, chunk, outer, s);
    if (rc != RIA_OK) return rc;
    if (!h->ch_side) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->ch_side, hipStreamNonBlocking));
        for (auto& ev_ : h->ch_ev) HIP_TRY(h, hipEventCreateWithFlags(&ev_, hipEventDisableTiming));
    }
    ChirpArgs A{};
    A.samples = samples_dev; A.stride = stride; A.buf_len = buf_len; A.threshold = threshold;
    A.tw = static_cast<const float2*>(h->d_ch_tw); A.tmpl_fft = static_cast<const float2*>(h->d_ch_tmpl_fft);
    A.tmpl = static_cast<const float*>(h->d_ch_tmpl); A.tmpl_energy[0] = h->ch_energy[0]; A.tmpl_energy[1] = h->ch_energy[1];
    A.w1 = static_cast<float2*>(h->d_ch_w1); A.w2 = static_cast<float2*>(h->d_ch_w2); A.best = static_cast<unsigned long long*>(h->d_ch_mag);
    A.cum = static_cast<float*>(h->d_ch_cum); A.st = static_cast<ChirpBufState*>(h->d_ch_st); A.out = out_dev;
    for (int first = 0; first < n_buffers; first += outer) {
        const int nb = std::min(outer, n_buffers - first);
        A.first = first; A.n_buffers = nb;
        for (int down = 0; down < 2; ++down) {
            A.down = down; A.sub = 0; A.n_sub = nb;
            hipLaunchKernelGGL(chirp_window_kernel, dim3((nb + 63) / 64), dim3(64), 0, s, A);
            hipLaunchKernelGGL(chirp_cumsum_kernel, dim3((nb + kCumB - 1) / kCumB), dim3(64), 0, s, A);
            if (down) {
                // the time-domain fallback (short down windows: a few buffers, 24 000-term sums per candidate, one workgroup
                // per buffer) touches other buffers than the FFT path and is latency-bound: it runs beside it on a side stream
                HIP_TRY(h, hipEventRecord(h->ch_ev[0], s));
                HIP_TRY(h, hipStreamWaitEvent(h->ch_side, h->ch_ev[0], 0));
                hipLaunchKernelGGL(chirp_td_kernel, dim3(nb), dim3(256), 0, h->ch_side, A);
                HIP_TRY(h, hipEventRecord(h->ch_ev[1], h->ch_side));
            }
            for (int sub = 0; sub < nb; sub += chunk) {
                A.sub = sub; A.n_sub = std::min(chunk, nb - sub);
                chirp_fft_forward(A, A.n_sub, s, true, true);
                chirp_fft_inverse_mag(A, A.n_sub, s);
                hipLaunchKernelGGL(chirp_peak_kernel, dim3((A.n_sub + 63) / 64), dim3(64), 0, s, A);
            }
            if (down) HIP_TRY(h, hipStreamWaitEvent(s, h->ch_ev[1], 0));
        }
        A.sub = 0; A.n_sub = nb;
        hipLaunchKernelGGL(chirp_finish_kernel, dim3((nb + 63) / 64), dim3(64), 0, s, A);
    }
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

// SimulatedChannel::applyTxCFO for a batch of transmissions (cfo_kernels.hip.h)
int ria_gpu_tx_cfo_batch(ria_gpu_handle h, const float* samples_dev, int64_t stride, int n_samples, int n_buffers,
                         const float* cfo_hz_dev, float* phase_inout_dev, float* out_dev, int64_t out_stride, void* stream) {
    if (!h) return RIA_ERR_INVALID;
    if (n_buffers == 0 || n_samples == 0) return RIA_OK;
    if (!samples_dev || !out_dev || !cfo_hz_dev || n_buffers < 0 || n_samples < 0 || stride < n_samples || out_stride < n_samples || samples_dev == out_dev)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_tx_cfo_batch: bad arguments");
    if (n_samples > (1 << kTxCfoMaxLog)) return fail(h, RIA_ERR_UNSUPPORTED, "ria_gpu_tx_cfo_batch: at most 131072 samples per transmission");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!h->d_ch_tw) { ChirpTables t = build_chirp_tables(); HIP_TRY(h, upload(&h->d_ch_tw, t.tw)); }
    int L = 0;
    while ((1 << L) < n_samples) ++L;
    const size_t N = size_t(1) << L, per = 2 * N * sizeof(float2) + static_cast<size_t>(n_samples) * sizeof(float);
    // buffers per pass: the workspace of one pass stays under 256 MiB (and the grid's y extent under 65536)
    const int chunk = static_cast<int>(std::min<size_t>(std::min(n_buffers, 32768), std::max<size_t>(1, (size_t(256) << 20) / per)));
    const size_t need = per * static_cast<size_t>(chunk);
    if (need > h->txcfo_ws_bytes) {
        if (h->d_txcfo_ws) { HIP_TRY(h, hipStreamSynchronize(s)); (void)hipFree(h->d_txcfo_ws); }
        h->d_txcfo_ws = nullptr; h->txcfo_ws_bytes = 0;
        HIP_TRY(h, hipMalloc(&h->d_txcfo_ws, need));
        h->txcfo_ws_bytes = need;
    }
    for (int first = 0; first < n_buffers; first += chunk) {
        const int nb = std::min(chunk, n_buffers - first);
        TxCfoArgs A{};
        A.in = samples_dev + static_cast<int64_t>(first) * stride; A.in_stride = stride;
        A.out = out_dev + static_cast<int64_t>(first) * out_stride; A.out_stride = out_stride;
        A.n = n_samples; A.log2n = L; A.n_buffers = nb; A.cfo_hz = cfo_hz_dev + first; A.phase = phase_inout_dev ? phase_inout_dev + first : nullptr;
        A.tw = static_cast<const float2*>(h->d_ch_tw);
        A.w1 = static_cast<float2*>(h->d_txcfo_ws); A.w2 = A.w1 + static_cast<size_t>(nb) * N; A.ph = reinterpret_cast<float*>(A.w2 + static_cast<size_t>(nb) * N);
        hipLaunchKernelGGL(txcfo_phase_kernel, dim3((nb + 63) / 64), dim3(64), 0, s, A);
        for (int inv = 0; inv < 2; ++inv) {
            float2* dst = inv ? A.w2 : A.w1;
            for (int S0 = 0; S0 < L;) {
                const int G = std::min(6, L - S0);
                const int mode = (S0 == 0) ? (inv ? 2 : 1) : 0;
                const dim3 grid(static_cast<unsigned>(((N >> G) + 255) / 256), nb), blk(256);
                const float2* src = A.w1;
                switch (G) {
                    case 1: hipLaunchKernelGGL(txcfo_fft_pass<1>, grid, blk, 0, s, A, src, dst, S0, mode, inv); break;
                    case 2: hipLaunchKernelGGL(txcfo_fft_pass<2>, grid, blk, 0, s, A, src, dst, S0, mode, inv); break;
                    case 3: hipLaunchKernelGGL(txcfo_fft_pass<3>, grid, blk, 0, s, A, src, dst, S0, mode, inv); break;
                    case 4: hipLaunchKernelGGL(txcfo_fft_pass<4>, grid, blk, 0, s, A, src, dst, S0, mode, inv); break;
                    case 5: hipLaunchKernelGGL(txcfo_fft_pass<5>, grid, blk, 0, s, A, src, dst, S0, mode, inv); break;
                    default: hipLaunchKernelGGL(txcfo_fft_pass<6>, grid, blk, 0, s, A, src, dst, S0, mode, inv); break;
                }
                S0 += G;
            }
        }
        hipLaunchKernelGGL(txcfo_rotate_kernel, dim3((n_samples + 255) / 256, nb), dim3(256), 0, s, A);
    }
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

int ria_gpu_chirp_preamble(ria_gpu_handle h, float* out_host, int max_n) {
    if (!h || !out_host) return RIA_ERR_INVALID;
    std::vector<float> p = build_chirp_preamble();
    if (static_cast<int>(p.size()) > max_n) return -static_cast<int>(p.size());
    std::memcpy(out_host, p.data(), p.size() * sizeof(float));
    return static_cast<int>(p.size());
}

int ria_gpu_sync_lts_batch(ria_gpu_handle h, const float* samples_dev, int64_t stride, int buf_len, int n_buffers,
                           const float* known_cfo_dev, float threshold, ria_lts_result* out_dev, void* stream) {
    if (!h) return RIA_ERR_INVALID;
    if (n_buffers == 0) return RIA_OK;
    if (!samples_dev || !out_dev || n_buffers < 0 || buf_len < 0 || stride < buf_len)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_sync_lts_batch: bad arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->d_hilbert65) { std::vector<float> hc = build_hilbert(65); HIP_TRY(h, upload(&h->d_hilbert65, hc)); }
    if (!h->lts_lds_opted) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(lts_sync_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lts_lds_bytes()));
        h->lts_lds_opted = 1;
    }
    LtsArgs A{};
    A.samples = samples_dev; A.stride = stride; A.buf_len = buf_len; A.n_buffers = n_buffers; A.known_cfo = known_cfo_dev;
    A.threshold = threshold; A.hilbert = static_cast<const float*>(h->d_hilbert65); A.out = out_dev;
    hipLaunchKernelGGL(lts_sync_kernel, dim3(n_buffers), dim3(kLtsThreads), lts_lds_bytes(), static_cast<hipStream_t>(stream), A);
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

int ria_gpu_sync_cox_batch(ria_gpu_handle h, const float* samples_dev, int64_t stride, int buf_len, int n_buffers,
                           float threshold, const float* noise_floor_dev, ria_cox_result* out_dev, void* stream) {
    if (!h) return RIA_ERR_INVALID;
    if (n_buffers == 0) return RIA_OK;
    if (!samples_dev || !out_dev || n_buffers < 0 || buf_len < 0 || buf_len > kCoxMaxBuf || stride < buf_len)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_sync_cox_batch: bad arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!h->d_cox_tI) {
        const CoxTemplate t = build_cox_template(h->plan);
        HIP_TRY(h, upload(&h->d_cox_tI, t.tI));
        HIP_TRY(h, upload(&h->d_cox_tQ, t.tQ));
        h->cox_energy_ref = t.energy_ref;
    }
    const bool searched = buf_len >= kCoxMinSearch && buf_len >= kCoxTotal + kCoxWindow;
    const int nM = searched ? cox_n_metric(buf_len) : 0, nE = searched ? cox_n_energy(buf_len) : 0;
    const size_t per = 2 * static_cast<size_t>(nM) + static_cast<size_t>(nE);
    // buffers per pass: the tables of one pass stay under 256 MiB and the grid's y extent under 65536
    int chunk = n_buffers;
    if (per) chunk = static_cast<int>(std::min<size_t>(chunk, std::max<size_t>(1, (size_t(64) << 20) / per)));
    chunk = std::min(chunk, 32768);
    const size_t need = per * static_cast<size_t>(chunk) + 16;
    if (need > h->cox_ws_floats) {
        if (h->d_cox_ws) { HIP_TRY(h, hipStreamSynchronize(s)); (void)hipFree(h->d_cox_ws); }
        h->d_cox_ws = nullptr; h->cox_ws_floats = 0;
        HIP_TRY(h, hipMalloc(&h->d_cox_ws, need * sizeof(float)));
        h->cox_ws_floats = need;
    }
    for (int first = 0; first < n_buffers; first += chunk) {
        const int nb = std::min(chunk, n_buffers - first);
        CoxArgs A{};
        A.samples = samples_dev + static_cast<int64_t>(first) * stride; A.stride = stride; A.buf_len = buf_len; A.n_buffers = nb;
        A.threshold = threshold; A.noise_in = noise_floor_dev ? noise_floor_dev + first : nullptr;
        A.twiddle = static_cast<const float2*>(h->d_twiddle);
        A.tI = static_cast<const float*>(h->d_cox_tI); A.tQ = static_cast<const float*>(h->d_cox_tQ); A.energy_ref = h->cox_energy_ref;
        float* ws = static_cast<float*>(h->d_cox_ws);
        A.dc = ws; A.metric = ws + static_cast<size_t>(nM) * nb; A.energy = ws + 2 * static_cast<size_t>(nM) * nb;
        A.nM = nM; A.nE = nE; A.out = out_dev + first;
        if (nM > 0) {
            hipLaunchKernelGGL(cox_prepare_kernel, dim3((nM + 255) / 256, nb), dim3(256), 0, s, A);
            hipLaunchKernelGGL(cox_metric_kernel<0>, dim3(((nM + 7) / 8 + 3) / 4, nb), dim3(256), 0, s, A);
            hipLaunchKernelGGL(cox_metric_kernel<1>, dim3((nM + 3) / 4, nb), dim3(256), 0, s, A);
        }
        hipLaunchKernelGGL(cox_scan_kernel, dim3(nb), dim3(1024), 0, s, A);
    }
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

int ria_gpu_cox_preamble(ria_gpu_handle h, float* out_host, int max_n) {
    if (!h || !out_host) return RIA_ERR_INVALID;
    std::vector<float> p = build_cox_preamble(h->plan);
    if (static_cast<int>(p.size()) > max_n) return -static_cast<int>(p.size());
    std::memcpy(out_host, p.data(), p.size() * sizeof(float));
    return static_cast<int>(p.size());
}

int ria_gpu_sync_host(ria_gpu_handle h, int kind, const float* samples_host, int n_samples, float threshold, float param,
                      uint32_t root_mask, void* result_out) {
    if (!h || !samples_host || !result_out || n_samples < 0 || kind < 0 || kind > 3) return RIA_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t need = static_cast<size_t>(n_samples) * sizeof(float) + 64;
    if (need > h->sync_host_bytes) {
        if (h->d_sync_host) (void)hipFree(h->d_sync_host);
        h->d_sync_host = nullptr; h->sync_host_bytes = 0;
        HIP_TRY(h, hipMalloc(&h->d_sync_host, need));
        h->sync_host_bytes = need;
    }
    unsigned char* base = static_cast<unsigned char*>(h->d_sync_host);
    float* d_param = reinterpret_cast<float*>(base);
    void* d_res = base + 16;
    float* d_x = reinterpret_cast<float*>(base + 64);
    HIP_TRY(h, hipMemcpy(d_x, samples_host, static_cast<size_t>(n_samples) * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(d_param, &param, sizeof(float), hipMemcpyHostToDevice));
    int rc;
    if (kind == 0) rc = ria_gpu_sync_chirp_batch(h, d_x, n_samples, n_samples, 1, threshold, static_cast<ria_chirp_result*>(d_res), nullptr);
    else if (kind == 1) rc = ria_gpu_sync_lts_batch(h, d_x, n_samples, n_samples, 1, d_param, threshold, static_cast<ria_lts_result*>(d_res), nullptr);
    else if (kind == 3) rc = ria_gpu_sync_cox_batch(h, d_x, n_samples, n_samples, 1, threshold, d_param, static_cast<ria_cox_result*>(d_res), nullptr);
    else rc = ria_gpu_sync_zc_batch(h, d_x, n_samples, n_samples, 1, threshold, root_mask, d_param, static_cast<ria_zc_result*>(d_res), nullptr);
    if (rc != RIA_OK) return rc;
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(result_out, d_res, 32, hipMemcpyDeviceToHost));
    return RIA_OK;
}

static bool mcdpsk_config_ok(const ria_mcdpsk_config* c) {
    return c && c->num_carriers >= 1 && c->num_carriers <= kMcMaxCarriers && (c->bits_per_symbol == 1 || c->bits_per_symbol == 2) &&
           (c->spreading == 1 || c->spreading == 2 || c->spreading == 4);
}

int ria_gpu_mcdpsk_demod_batch(ria_gpu_handle h, const ria_mcdpsk_config* cfg, const float* samples_dev, int64_t stride,
                               int frame_samples, int n_frames, const float* cfo_hz_dev, const float* phase0_dev,
                               float* llr_out_dev, int llr_stride, ria_mcdpsk_status* status_dev, void* stream) {
    if (!h) return RIA_ERR_INVALID;
    if (n_frames == 0) return RIA_OK;
    if (!mcdpsk_config_ok(cfg) || !samples_dev || !llr_out_dev || !status_dev || n_frames < 0 || stride < frame_samples ||
        frame_samples < (kMcTrain + 2) * kMcSps)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_mcdpsk_demod_batch: bad arguments");
    const int nc = cfg->num_carriers, num_rx = (frame_samples - (kMcTrain + 1) * kMcSps) / kMcSps;
    const int nds = std::max(1, num_rx / cfg->spreading);
    if (llr_stride < nds * nc * cfg->bits_per_symbol) return fail(h, RIA_ERR_INVALID, "ria_gpu_mcdpsk_demod_batch: llr_stride too small");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!h->d_mc_mixer.count(nc)) {
        void* p = nullptr;
        std::vector<float> m = build_mcdpsk_mixer(nc);
        HIP_TRY(h, upload(&p, m));
        h->d_mc_mixer[nc] = p;
    }
    if (!h->d_mc_hilbert) { std::vector<float> hc = build_hilbert127(); HIP_TRY(h, upload(&h->d_mc_hilbert, hc)); }
    const int lds = mcdpsk_lds_bytes(nc, frame_samples);
    if (lds > 160 * 1024) return fail(h, RIA_ERR_UNSUPPORTED, "ria_gpu_mcdpsk_demod_batch: frame too long for one workgroup's LDS");
    if (lds > h->mc_lds_opted) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(mcdpsk_corr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        h->mc_lds_opted = lds;
    }
    // the tables handed from kernel to kernel (and the CFO-corrected samples when a CFO array is given) live in a per-handle
    // workspace; a batch is walked in chunks of frames that fit it
    const size_t per_frame = mcdpsk_ws_floats_per_frame(nc, frame_samples, cfg->spreading, cfo_hz_dev != nullptr);
    const int chunk = static_cast<int>(std::min<size_t>(static_cast<size_t>(n_frames), std::max<size_t>(1, (size_t(256) << 20) / (per_frame * sizeof(float)))));
    {
        const size_t need = static_cast<size_t>(chunk) * per_frame + 64;
        if (need > h->mc_ws_floats) {
            if (h->d_mc_ws) { HIP_TRY(h, hipStreamSynchronize(s)); (void)hipFree(h->d_mc_ws); }
            h->d_mc_ws = nullptr; h->mc_ws_floats = 0;
            HIP_TRY(h, hipMalloc(&h->d_mc_ws, need * sizeof(float)));
            h->mc_ws_floats = need;
        }
    }
    McArgs A{};
    A.samples = samples_dev; A.stride = stride; A.frame_samples = frame_samples; A.nc = nc; A.bps = cfg->bits_per_symbol;
    A.spreading = cfg->spreading; A.cfo = cfo_hz_dev; A.phase0 = phase0_dev; A.mixer = static_cast<const float2*>(h->d_mc_mixer[nc]);
    A.hilbert = static_cast<const float*>(h->d_mc_hilbert); A.llr = llr_out_dev;
    A.llr_stride = llr_stride; A.status = status_dev; A.chunk = mcdpsk_corr_chunk(nc, frame_samples);
    const size_t n_sym = static_cast<size_t>(3 + num_rx);
    for (int first = 0; first < n_frames; first += chunk) {
        A.first = first; A.n_frames = std::min(chunk, n_frames - first);
        const size_t F = static_cast<size_t>(A.n_frames);
        float* p = static_cast<float*>(h->d_mc_ws);
        A.ws = p; if (cfo_hz_dev) p += F * 2 * frame_samples;
        p = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(p) + 15) & ~uintptr_t(15));
        A.Yg = reinterpret_cast<float2*>(p); p += 2 * n_sym * nc * F;
        A.cph = p; p += static_cast<size_t>(nds) * nc * F;
        A.cmag = p; p += static_cast<size_t>(nds) * nc * F;
        A.pe2 = p; p += static_cast<size_t>(nds) * nc * F;
        A.rel = p; p += static_cast<size_t>(kMcMaxCarriers) * F;
        A.scale = p;
        hipLaunchKernelGGL(mcdpsk_corr_kernel, dim3(A.n_frames), dim3(256), lds, s, A);
        hipLaunchKernelGGL(mcdpsk_chain_kernel, dim3(static_cast<unsigned>((F * nc + 255) / 256)), dim3(256), 0, s, A);
        hipLaunchKernelGGL(mcdpsk_stats_kernel, dim3(static_cast<unsigned>((F + 63) / 64)), dim3(64), 0, s, A);
        hipLaunchKernelGGL(mcdpsk_llr_kernel, dim3(static_cast<unsigned>((F * nds * nc + 255) / 256)), dim3(256), 0, s, A);
    }
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

// Host-buffer forms for the single-frame MC-DPSK plug-in adaptor (GpuMcDpskWaveform): staged through the handle's pinned +
// device block on the handle's own stream, like ria_gpu_rx_frames_host.
int ria_gpu_mcdpsk_demod_host(ria_gpu_handle h, const ria_mcdpsk_config* cfg, const float* samples_host, int n_samples, float cfo_hz,
                              float phase0, float* llr_out_host, int max_llr, ria_mcdpsk_status* status_out) {
    if (!h || !mcdpsk_config_ok(cfg) || !samples_host || !llr_out_host || !status_out || n_samples < (kMcTrain + 2) * kMcSps)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_mcdpsk_demod_host: bad arguments");
    const int num_rx = (n_samples - (kMcTrain + 1) * kMcSps) / kMcSps;
    const int n_llr = std::max(1, num_rx / cfg->spreading) * cfg->num_carriers * cfg->bits_per_symbol;
    if (max_llr < n_llr) return fail(h, RIA_ERR_INVALID, "ria_gpu_mcdpsk_demod_host: llr buffer too small (%d needed)", n_llr);
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t b_s = static_cast<size_t>(n_samples) * sizeof(float), o_par = up256(b_s), o_llr = up256(o_par + 8);
    const size_t o_st = up256(o_llr + static_cast<size_t>(n_llr) * sizeof(float)), total = up256(o_st + sizeof(ria_mcdpsk_status));
    int rc = ensure_host_stage(h, total);
    if (rc != RIA_OK) return rc;
    unsigned char *D = h->d_hstage, *P = h->p_hstage;
    hipStream_t s = h->hstream;
    std::memcpy(P, samples_host, b_s);
    const float par[2] = {cfo_hz, phase0};
    std::memcpy(P + o_par, par, sizeof(par));
    HIP_TRY(h, hipMemcpyAsync(D, P, o_par + 8, hipMemcpyHostToDevice, s));
    rc = ria_gpu_mcdpsk_demod_batch(h, cfg, reinterpret_cast<const float*>(D), n_samples, n_samples, 1, reinterpret_cast<const float*>(D + o_par),
                                    reinterpret_cast<const float*>(D + o_par) + 1, reinterpret_cast<float*>(D + o_llr), n_llr,
                                    reinterpret_cast<ria_mcdpsk_status*>(D + o_st), s);
    if (rc != RIA_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(P + o_llr, D + o_llr, total - o_llr, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    std::memcpy(llr_out_host, P + o_llr, static_cast<size_t>(n_llr) * sizeof(float));
    std::memcpy(status_out, P + o_st, sizeof(ria_mcdpsk_status));
    return RIA_OK;
}

int ria_gpu_ldpc_decode_robust_host(ria_gpu_handle h, const float* llr_host, int n_cw, uint8_t* out_host, uint8_t* ok_host,
                                    uint16_t* iters_host, uint8_t* tries_host) {
    if (!h || !llr_host || !out_host || !ok_host || n_cw <= 0) return fail(h, RIA_ERR_INVALID, "ria_gpu_ldpc_decode_robust_host: bad arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = static_cast<size_t>(n_cw), nb = static_cast<size_t>((h->code.k + 7) / 8);
    const size_t b_llr = n * 648 * sizeof(float), o_out = up256(b_llr), o_ok = up256(o_out + n * nb), o_it = up256(o_ok + n), o_tr = up256(o_it + 2 * n);
    const size_t total = up256(o_tr + n);
    int rc = ensure_host_stage(h, total);
    if (rc != RIA_OK) return rc;
    unsigned char *D = h->d_hstage, *P = h->p_hstage;
    hipStream_t s = h->hstream;
    std::memcpy(P, llr_host, b_llr);
    HIP_TRY(h, hipMemcpyAsync(D, P, b_llr, hipMemcpyHostToDevice, s));
    rc = ria_gpu_ldpc_decode_robust_batch(h, reinterpret_cast<const float*>(D), n_cw, D + o_out, D + o_ok, reinterpret_cast<uint16_t*>(D + o_it), D + o_tr, s);
    if (rc != RIA_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(P + o_out, D + o_out, total - o_out, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    std::memcpy(out_host, P + o_out, n * nb);
    std::memcpy(ok_host, P + o_ok, n);
    if (iters_host) std::memcpy(iters_host, P + o_it, 2 * n);
    if (tries_host) std::memcpy(tries_host, P + o_tr, n);
    return RIA_OK;
}

int ria_gpu_mcdpsk_modulate_batch(ria_gpu_handle h, const ria_mcdpsk_config* cfg, const uint8_t* data_dev, int n_bytes, int n_frames,
                                  float* out_dev, int64_t out_stride, void* stream) {
    if (!h) return RIA_ERR_INVALID;
    if (n_frames == 0) return RIA_OK;
    if (!mcdpsk_config_ok(cfg) || !data_dev || !out_dev || n_bytes < 0 || n_frames < 0)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_mcdpsk_modulate_batch: bad arguments");
    const int nc = cfg->num_carriers, bits_per_sym = nc * cfg->bits_per_symbol;
    const int n_data_sym = (n_bytes * 8 + bits_per_sym - 1) / bits_per_sym;
    const int64_t frame_samples = static_cast<int64_t>(kMcTrain + 1 + n_data_sym * cfg->spreading) * kMcSps;
    if (out_stride < frame_samples) return fail(h, RIA_ERR_INVALID, "ria_gpu_mcdpsk_modulate_batch: out_stride too small (%lld samples per frame)", static_cast<long long>(frame_samples));
    const int lds = n_data_sym * nc * static_cast<int>(sizeof(float2)) + 16;
    if (lds > 64 * 1024) return fail(h, RIA_ERR_UNSUPPORTED, "ria_gpu_mcdpsk_modulate_batch: too many data symbols for one workgroup");
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->d_mc_carrier.count(nc)) {
        std::vector<float> car, tr;
        build_mcdpsk_mod_tables(nc, car, tr);
        void *pc = nullptr, *pt = nullptr;
        HIP_TRY(h, upload(&pc, car));
        HIP_TRY(h, upload(&pt, tr));
        h->d_mc_carrier[nc] = pc; h->d_mc_train[nc] = pt;
    }
    McModArgs A{};
    A.data = data_dev; A.n_bytes = n_bytes; A.n_frames = n_frames; A.nc = nc; A.bps = cfg->bits_per_symbol; A.spreading = cfg->spreading;
    A.n_data_sym = n_data_sym; A.carrier = static_cast<const float2*>(h->d_mc_carrier[nc]); A.train = static_cast<const float2*>(h->d_mc_train[nc]);
    A.out = out_dev; A.stride = out_stride;
    hipLaunchKernelGGL(mcdpsk_modulate_kernel, dim3(n_frames), dim3(256), lds, static_cast<hipStream_t>(stream), A);
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

int ria_gpu_mcdpsk_modulate_host(ria_gpu_handle h, const ria_mcdpsk_config* cfg, const uint8_t* data, int n_bytes,
                                 float* out_host, int max_n) {
    if (!h || !mcdpsk_config_ok(cfg) || !data || !out_host || n_bytes < 0) return RIA_ERR_INVALID;
    std::vector<float> f = build_mcdpsk_frame(cfg->num_carriers, cfg->bits_per_symbol, cfg->spreading, data, n_bytes);
    if (static_cast<int>(f.size()) > max_n) return -static_cast<int>(f.size());
    std::memcpy(out_host, f.data(), f.size() * sizeof(float));
    return static_cast<int>(f.size());
}

int ria_gpu_chase_combine_batch(ria_gpu_handle h, float* acc_dev, int32_t* count_dev, const uint8_t* decoded_dev,
                                const float* soft_dev, int n_cw, uint8_t* stored_out_dev, void* stream) {
    if (!h) return RIA_ERR_INVALID;
    if (n_cw == 0) return RIA_OK;
    if (!acc_dev || !count_dev || !soft_dev || n_cw < 0) return fail(h, RIA_ERR_INVALID, "ria_gpu_chase_combine_batch: bad arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(chase_combine_kernel, dim3(n_cw), dim3(256), 0, static_cast<hipStream_t>(stream), acc_dev, count_dev, decoded_dev,
                       soft_dev, n_cw, stored_out_dev);
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

void ria_link_recommend(float snr_db, float fading_index, ria_link_recommendation* out) { if (out) *out = recommend_waveform_and_rate(snr_db, fading_index); }
void ria_link_data_mode(float snr_db, int waveform, float fading_index, ria_link_recommendation* out) {
    if (!out) return;
    *out = recommend_data_mode(snr_db, waveform, fading_index);
    if (waveform != kWaveMcDpsk) out->estimated_throughput_bps = 0.0f;
}
int ria_link_ofdm_code_rate(float snr_db, float fading_index) { return select_ofdm_code_rate(snr_db, fading_index); }
int ria_link_cap_initial_rate(float snr_db, float fading_index, int candidate_rate) { return cap_initial_ofdm_rate(snr_db, fading_index, candidate_rate); }

int ria_gpu_burst_deinterleave_batch(ria_gpu_handle h, const float* physical_llr_dev, int llr_stride, int burst_frames,
                                     int n_groups, float* logical_llr_out_dev, void* stream) {
    if (!h) return RIA_ERR_INVALID;
    if (n_groups == 0 || burst_frames == 0) return RIA_OK;
    if (!physical_llr_dev || !logical_llr_out_dev || physical_llr_dev == logical_llr_out_dev || llr_stride < 2592 || burst_frames < 0 || n_groups < 0)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_burst_deinterleave_batch: bad arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    const int total = n_groups * burst_frames * 324;
    hipLaunchKernelGGL(burst_deinterleave_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), physical_llr_dev,
                       llr_stride, burst_frames, n_groups, logical_llr_out_dev);
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}
int ria_gpu_burst_interleave_batch(ria_gpu_handle h, const uint8_t* logical_bytes_dev, int burst_frames, int n_groups,
                                   uint8_t* physical_bytes_out_dev, void* stream) {
    if (!h) return RIA_ERR_INVALID;
    if (n_groups == 0 || burst_frames == 0) return RIA_OK;
    if (!logical_bytes_dev || !physical_bytes_out_dev || logical_bytes_dev == physical_bytes_out_dev || burst_frames < 0 || n_groups < 0)
        return fail(h, RIA_ERR_INVALID, "ria_gpu_burst_interleave_batch: bad arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    const int total = n_groups * burst_frames * 324;
    hipLaunchKernelGGL(burst_interleave_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), logical_bytes_dev,
                       burst_frames, n_groups, physical_bytes_out_dev);
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

int ria_gpu_ldpc_encode_host(ria_gpu_handle h, const uint8_t* info, int n_cw, uint8_t* coded_out) {
    if (!h || !info || !coded_out || n_cw < 0) return RIA_ERR_INVALID;
    const LdpcCode& c = h->code;
    const int kb = (c.k + 7) / 8;
    std::vector<uint8_t> bits(648);
    for (int w = 0; w < n_cw; ++w) {
        const uint8_t* in = info + static_cast<size_t>(w) * kb;
        for (int j = 0; j < c.k; ++j) bits[j] = (in[j >> 3] >> (7 - (j & 7))) & 1;
        for (int i = 0; i < c.m; ++i) {   // H = [H_data | I]: parity i = XOR of the information bits of row i
            int p = 0;
            const auto& row = c.rows[i];
            for (size_t s2 = 0; s2 + 1 < row.size(); ++s2) p ^= bits[row[s2]];
            bits[c.k + i] = static_cast<uint8_t>(p);
        }
        uint8_t* out = coded_out + static_cast<size_t>(w) * 81;
        std::memset(out, 0, 81);
        for (int j = 0; j < 648; ++j) out[j >> 3] |= static_cast<uint8_t>(bits[j] << (7 - (j & 7)));
    }
    return RIA_OK;
}

int ria_gpu_debug_queue_fault(ria_gpu_handle h) {
    if (!h) return RIA_ERR_INVALID;
    if (!h->d_ctl) return 0;
    HIP_TRY(h, hipSetDevice(h->device));
    DecodeCtl c[kMaxParts];
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(c, h->d_ctl, sizeof(c), hipMemcpyDeviceToHost));
    int bad = 0;
    for (const DecodeCtl& q : c) bad |= q.queue_fault ? 1 : 0;
    return bad;
}

int ria_gpu_debug_math(ria_gpu_handle h, int op, const float* a_dev, const float* b_dev, int n, float* out_dev,
                       void* stream) {
    if (!h || !a_dev || !out_dev || n < 0) return fail(h, RIA_ERR_INVALID, "ria_gpu_debug_math: bad argument");
    if (n == 0) return RIA_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), op,
                       a_dev, b_dev, n, out_dev);
    HIP_TRY(h, hipGetLastError());
    return RIA_OK;
}

}  // extern "C"
