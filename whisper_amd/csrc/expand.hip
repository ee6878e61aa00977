// Helper share expansion kernel, in its own translation unit (see the note in kernels.hip).
#include <hip/hip_runtime.h>
#include "keccak.hpp"
#include "prio3_dev.hpp"

namespace wh {

// -------------------------------------------------------------------------------- expand ----
// Helper share expansion (prio3.rs:2480-2536): P::seed_stream(seed, dst(usage), [agg_id])
// .into_field_vec(n).  Jobs [0, npad) expand the measurement share, jobs [npad, 2 npad) the two
// proof shares, so a wave never mixes the two shapes.  Every sponge of a lane runs through the
// same loop body (one copy of the permutation code, small register footprint).
// sel 0: measurement and proof shares of every Helper (draft-07 path); 1: proof shares only (the
// measurement share comes out of k_meas); 2: measurement shares of the dirty list only.
__device__ __forceinline__ void expand_body(const Prio3Dev &P, const Workspace &W, uint32_t npad, uint32_t sel) {
    uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t which = job >= npad;
    uint32_t slot = which ? job - npad : job;
    if (sel == 1 && !which) return;
    if (sel == 2 && which) return;
    if (slot >= W.counters[sel == 2 ? 2 : 0]) return;
    const RepDesc d = W.desc[(sel == 2 ? W.dirty : W.hlist)[slot]];
    // batched report: meas_seed | proof1_seed | jr_blind | proof2_seed; VdafKey: meas_seed | proof_seed | blind
    const uint8_t *seeds = d.seeds;
    SpongeG s;
    const uint32_t nst = which ? (P.mode ? 1 : 2) : 1;
#pragma unroll 1
    for (uint32_t t = 0; t < nst; t++) {
        const uint8_t *seed = which ? seeds + (t ? 48 : 16) : seeds;
        uint8_t *out = const_cast<uint8_t *>(which ? ((t || P.mode) ? d.proof2 : d.proof1) : d.meas);
        s.init_xof(P.alg_id, which ? DST_PROOF : DST_MEAS, seed);
        s.absorb3(P.mode ? d.base : d.base + 16 + 16 * P.num_aggregators, 1, nullptr, 0, nullptr, 0);   // binder = [agg_id]
        s.finalize();
        xof_emit(s, out, which ? P.proof_len : P.input_len, P.sampler_bound);
    }
}


__global__ void __launch_bounds__(256) k_expand(Prio3Dev P, Workspace W, uint32_t npad, uint32_t sel) { expand_body(P, W, npad, sel); }
__global__ void __launch_bounds__(256) k_expand_redo(Prio3Dev P, Workspace W, uint32_t npad) { expand_body(P, W, npad, 2); }

void launch_expand(const Prio3Dev &P, const Workspace &W, uint32_t npad, uint32_t sel, hipStream_t stream) {
    if (sel == 2) hipLaunchKernelGGL(k_expand_redo, dim3(npad / 256), dim3(256), 0, stream, P, W, npad);
    else hipLaunchKernelGGL(k_expand, dim3(2 * npad / 256), dim3(256), 0, stream, P, W, npad, sel);
}

}  // namespace wh
