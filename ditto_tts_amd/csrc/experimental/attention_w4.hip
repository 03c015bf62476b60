// attention_w4.hip — the four-waves-per-SIMD form of the fused head_dim-64 attention forward (attn64w4_kernel, attn64v2.h), in its
// own translation unit because it is compiled with -fno-slp-vectorize (ditto_tts_amd/build.py): its row sum is 32 fp32 adds
// per tile in two chains, which the SLP vectoriser pairs into v_pk_add_f32 — and packed fp32 instructions do not overlap the
// matrix pipe (+22 cycles per packed instruction next to an MFMA).  Replaces softmax(q k^T / sqrt(d_h)) v of
// reference src/components/DiT.py:126-134 (self) and the torch MHA math behind :144-148 (cross), as attention.hip does.
#include <type_traits>

#include "attn_common.h"

namespace ditto {

namespace {
#include "attn64v2.h"
}

hipError_t launch_attn64w4(const AttnParams& p_in, bool resid, hipStream_t s, bool wide) {
    AttnParams p = p_in;
    if (wide) {                                      // 256 queries per workgroup, eight waves
        p.nqb = (p.Sq + 255) / 256;
        const dim3 grid(p.nqb * p.H * p.B);
        if (resid) hipLaunchKernelGGL((attn64w4_kernel<true, 8>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((attn64w4_kernel<false, 8>), grid, dim3(512), 0, s, p);
        return hipGetLastError();
    }
    const dim3 grid(p.nqb * p.H * p.B);
    if (resid) hipLaunchKernelGGL((attn64w4_kernel<true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn64w4_kernel<false>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace ditto
