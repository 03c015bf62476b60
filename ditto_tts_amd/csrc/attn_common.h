// attn_common.h — what the fused head_dim-64 attention kernels (attention.hip, attention_train.hip) share: the kernel
// parameter block, tile constants and two fragment helpers.
#pragma once
#include "gemm_common.h"

namespace ditto {

constexpr int ATT_DH = 64, ATT_KBLK = 64;
constexpr int ATT_KV_TILE_BYTES = ATT_KBLK * ATT_DH * 2;  // 8 KiB: 64 keys x 64 head columns of bf16

struct AttnParams {
    const bf16* q; int ldq;
    const bf16* k; int ldk;
    const bf16* v; int ldv;
    bf16* out; int ldo;
    float* resid; int ldr;
    const float* resid_in;
    int resid_bf16;                // resid / resid_in point to BF16 rows (ldr in bf16 elements): the bf16 residual stream
    int B, H, Sq, Skv, nqb;
    float scale_log2;  // scale * log2(e)
    // training forward (TRAIN instantiations only)
    float* lse;                    // [B, H, Sq] log2-domain log-sum-exp: m * scale_log2 + log2(l)
    unsigned drop_thr; float keep_scale; unsigned seed_lo, seed_hi; int layer;
};

hipError_t launch_attention_train64(const AttnParams& p, bool resid, hipStream_t s);   // attention_train.hip
hipError_t launch_attn64p(const AttnParams& p, bool resid, hipStream_t s, bool ring3 = false, int no_q = 0);   // attention_p.hip: 64 queries per wave (round 6: attn64p, attn64q)

// the self-attention epilogue's stream update  h[row, col .. col+3] = h_in[...] + o  (src/components/DiT.py:139: no out-proj),
// on an fp32 stream or a bf16 one (wave-uniform branch, outside every loop)
DITTO_DEV void attn_resid_update(const AttnParams& p, size_t grow, int col, f32x4 o) {
    if (p.resid_bf16) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16*>(p.resid_in) + grow * p.ldr + col);
        o[0] += __builtin_bit_cast(float, w[0] << 16); o[1] += __builtin_bit_cast(float, w[0] & 0xFFFF0000u);
        o[2] += __builtin_bit_cast(float, w[1] << 16); o[3] += __builtin_bit_cast(float, w[1] & 0xFFFF0000u);
        u32x2 st;
        st[0] = pack_bf16x2(o[0], o[1]); st[1] = pack_bf16x2(o[2], o[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.resid) + grow * p.ldr + col) = st;
    } else {
        f32x4 r = *reinterpret_cast<const f32x4*>(p.resid_in + grow * p.ldr + col);
        r += o;
        *reinterpret_cast<f32x4*>(p.resid + grow * p.ldr + col) = r;
    }
}

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

DITTO_DEV bf16x8 cat4(bf16x4 a, bf16x4 b) {
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

// online-softmax rescale threshold (guide T13), in log2 units of the exp2 domain: the running max is only raised
// (and O^T / l rescaled) when some row's new maximum exceeds it by more than this, so P stays <= 2^8 — exact in
// bf16's exponent range, accumulated in fp32 — and the 32-register O^T rescale is skipped on most tiles.
constexpr float ATT_RESCALE_THR_LOG2 = 8.0f;

}  // namespace ditto
