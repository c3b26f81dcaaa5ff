// Reproduces the round-1 work-queue hang at the ISA level (nothing here runs): compile with
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 --cuda-device-only -S tools/probes/queue_pop_probe.hip -o /tmp/q.s
// and compare the loop nests of phase0_probe<ShapeR12, 0 / 1> (lane-0-only pop: the body re-entered with EXEC = ~lane0,
// v_mov_b32 u, 0 and v_readfirstlane of the first active lane) with <ShapeR12, 2> (all-lane pop: one loop).  See the comment
// above RIA_QUEUE_GUARD in ria_amd/csrc/ldpc_fast.hip.h.
#include "../../ria_amd/csrc/ldpc_fast.hip.h"
using namespace ria;
// lane-0-only pop (the round-1 form) vs the all-lane pop, same loop otherwise
template <class S, int FORM>
__global__ __launch_bounds__(64) void phase0_probe(FastDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const unsigned total = A.ctl->n_list1 * 4u;
    if (blockIdx.x >= total) return;
    FastState<S> st;
    fast_load_tables(st, A.c, smem, lane);
    for (;;) {
        unsigned u;
        if (FORM == 0) { u = 0; if (lane == 0) u = atomicAdd(&A.ctl->next_z, 1u); u = __shfl(u, 0); }
        else if (FORM == 1) { u = 0; if (lane == 0) u = atomicAdd(&A.ctl->next_z, 1u); u = __builtin_amdgcn_readfirstlane(u); }
        else { u = atomicAdd(&A.ctl->next_z, lane == 0 ? 1u : 0u); u = __builtin_amdgcn_readfirstlane(u); }
        if (u >= total) break;
        fast_unit<S>(st, A, smem, A.list1[u >> 2], 1 + static_cast<int>(u & 3u), lane, A.staged + static_cast<size_t>(u >> 2) * kStageFloats);
    }
}
template __global__ void phase0_probe<ShapeR12, 0>(FastDecodeArgs);
template __global__ void phase0_probe<ShapeR12, 1>(FastDecodeArgs);
template __global__ void phase0_probe<ShapeR12, 2>(FastDecodeArgs);
