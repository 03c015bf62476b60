// attention_train.hip — the training forward's fused attention (head_dim 64): attn64v2 (attn64v2.h) with TRAIN (query fragments
// scaled in the kernel, log-sum-exp written for the backward) and DROP (train-mode dropout on the probabilities, reference
// nn.MultiheadAttention dropout 0.1 in src/components/DiT.py:144-148).  Its own translation unit because it is compiled with
// -fno-slp-vectorize (ditto_tts_amd/build.py): the products these variants add must stay scalar v_mul_f32.
#include <type_traits>

#include "attn_common.h"

namespace ditto {

namespace {
#include "attn64v2.h"
}  // namespace

// 3 waves per SIMD, no V-fragment prefetch (the inference launch's measurement: the kernel is latency-bound, occupancy pays more)
hipError_t launch_attention_train64(const AttnParams& p, bool resid, hipStream_t s) {
    const dim3 grid(p.nqb * p.H * p.B), block(256);
    if (p.drop_thr) {
        if (resid) hipLaunchKernelGGL((attn64v2_kernel<true, 3, 2, true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((attn64v2_kernel<false, 3, 2, true, true>), grid, block, 0, s, p);
    } else {
        if (resid) hipLaunchKernelGGL((attn64v2_kernel<true, 3, 2, true, false>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((attn64v2_kernel<false, 3, 2, true, false>), grid, block, 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace ditto
