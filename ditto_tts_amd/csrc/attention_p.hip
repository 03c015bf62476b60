// attention_p.hip — the translation unit of attn64p / attn64q (attn64p.h, attn64q.h): the head_dim-64 attention forward of the
// inference path at 64 queries per wave (reference src/components/DiT.py:131-139 self-attention, :144-148 cross-attention).  Its own file because it is
// compiled without the SLP vectoriser (build.py EXTRA: -fno-honor-nans -fno-slp-vectorize; packed fp32 adds cost it 40 registers)
// and because co-compiled kernel templates perturb one another's register allocation (guide rule 19).
#include <type_traits>

#include "attn_common.h"

namespace ditto {

namespace {
#include "attn64v2.h"   // the tile constants (the kernel itself is instantiated in attention.hip / attention_train.hip)
#include "attn64p.h"
#include "attn64q.h"
}  // namespace

// p.nqb is set here: blocks of 256 queries
hipError_t launch_attn64p(const AttnParams& p_in, bool resid, hipStream_t s, bool ring3, int no_q) {
    AttnParams p = p_in;
    p.nqb = (p.Sq + 255) / 256;
    const dim3 grid(p.nqb * p.H * p.B), block(256);
    // at least two key tiles (the last may be partial): attn64q (attn64q.h: one software-pipelined stream per wave, optimistic softmax with attn64p's
    // loop as the exact path of a workgroup whose rows left the range).  A rule on the SHAPE only.  attn_flags 1048576 = never.
    if (no_q != 1 && !ring3 && p.Skv > KBLK) {
        // K fragments held for block B (24 LDS fragment reads per tile instead of 32) in the plain form, not in the residual form:
        // in-model A/B at C2, self (residual form) 110.6 against 116.5 us without them, cross (plain) 104.0 against 106.0 with
        // them (profiles/r06_attn64q.txt).  A rule on the instantiation, not on the data.  attn_flags 2097152 swaps the two (A/B).
        const bool hold = (no_q == 2) ? resid : !resid;
        if (p.Skv % KBLK) {     // a partial last tile: the instantiation with the masked copy of the loop body (no fragment held)
            if (resid) hipLaunchKernelGGL((attn64q_kernel<true, 0, Q_QD, true, 0, true>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((attn64q_kernel<false, 0, Q_QD, true, 0, true>), grid, block, 0, s, p);
        } else if (resid) {
            if (hold) hipLaunchKernelGGL((attn64q_kernel<true, 0, Q_QD, true, 1>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((attn64q_kernel<true, 0, Q_QD, true, 0>), grid, block, 0, s, p);
        } else {
            if (hold) hipLaunchKernelGGL((attn64q_kernel<false, 0, Q_QD, true, 1>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((attn64q_kernel<false, 0, Q_QD, true, 0>), grid, block, 0, s, p);
        }
        return hipGetLastError();
    }
    if (ring3) {   // A/B (attn_flags 524288): a ring of 3 tile pairs (48 KiB) instead of 4: same bits, same speed in the model
        if (resid) hipLaunchKernelGGL((attn64p_kernel<true, 3>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((attn64p_kernel<false, 3>), grid, block, 0, s, p);
    } else {
        if (resid) hipLaunchKernelGGL((attn64p_kernel<true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((attn64p_kernel<false>), grid, block, 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace ditto
