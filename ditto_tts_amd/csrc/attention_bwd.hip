// attention_bwd.hip — fused (flash-style) attention BACKWARD for gfx950, head_dim 64, bf16 in / fp32 accumulate.
//
// Backward of  O = dropout(softmax(q k^T * scale)) v  (self-attention, reference src/components/DiT.py:131-134, and
// nn.MultiheadAttention's cross-attention, :144-148, dropout 0.1 in train mode) with the probabilities RECOMPUTED
// from q, k and the forward's log-sum-exp, never stored:
//     P = exp2(c s - L),   dP = mask/(1-p) * (dO v^T),   dS = P * (dP - delta) * scale,   delta = rowsum(dO * O)
//     dV = (mask/(1-p) * P)^T dO,    dQ = dS k,    dK = dS^T q.
// Two kernels, each the forward kernel's structure (attention.hip) with different operands, so that every output
// is owned by exactly one workgroup: NO atomics and no cross-workgroup sum => bit-reproducible gradients.
//   dq kernel    one workgroup = 128 queries; streams 64-key tiles.   S^T = K Q^T and dP^T = V dO^T have the QUERY
//                on the lane (L and delta are per-lane scalars); dS^T, packed to bf16 in registers, is the B operand
//                of dQ^T += K^T dS^T (K^T through ds_read_b64_tr_b16 from a second, transposed-read image of K).
//   dkdv kernel  one workgroup = 128 keys; streams 64-query tiles.   S = Q K^T and dP = dO V^T have the KEY on the
//                lane (K, V fragments live in registers for the whole kernel); P and dS in registers are the B
//                operands of dV^T += dO^T P and dK^T += Q^T dS (Q^T, dO^T by transposed reads).
// This costs 7 products instead of the 5 of a single-kernel backward (S and dP are computed in both), 14 B H Sq Skv
// dh FLOPs per call: 40 % more MFMA work bought for determinism and for not needing fp32 dQ atomics.
// Roofline: MFMA-bound; K/V (dq kernel) or Q/dO (dkdv kernel) of one head are re-read from the XCD's L2.
#include <type_traits>

#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int DH = 64, BLK = 128, TILE = 64;
constexpr int IMG = TILE * DH * 2;   // one 64-row x 128-B tile image: 8 KiB
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

struct BwdParams {
    const bf16* q; int ldq; const bf16* k; int ldk; const bf16* v; int ldv;
    const bf16* dout; int lddo;
    bf16* dq; int lddq; bf16* dk; int lddk; bf16* dv; int lddv;
    const float* lse;     // [B, H, Sq] log2 domain
    const float* delta;   // [B, H, Sq]
    int B, H, Sq, Skv, nblk;
    float scale, scale_log2;
    unsigned drop_thr; float keep_scale; unsigned seed_lo, seed_hi; int layer;
    const float* rope_cos; const float* rope_sin;   // inverse RoPE of dq / dk in the epilogue (null: none); [rows, 32]
};

DITTO_DEV bf16x8 cat4(bf16x4 a, bf16x4 b) {
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

// delta[b,h,q] = sum_c dO[row, h*64 + c] * O[row, h*64 + c].  O = o_bf16 (cross-attention output) or
// h_after - h_before (self-attention: the residual stream before / after the segment, no out-proj).
// One wave per row, 4 columns per lane per pass, a head = 16 lanes.
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ dout, int lddo,
                                                         const bf16* __restrict__ o_bf16, int ldo,
                                                         const float* __restrict__ h_after,
                                                         const float* __restrict__ h_before, int ldh,
                                                         float* __restrict__ delta, int B, int H, int Sq) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * Sq) return;
    const int b = row / Sq, qi = row % Sq;
    const int d = H * DH;
    for (int c0 = 0; c0 < d; c0 += 256) {
        const int col = c0 + lane * 4;
        float acc = 0.f;
        if (col < d) {
            const u32x2 g = *reinterpret_cast<const u32x2*>(dout + (size_t)row * lddo + col);
            float o[4];
            if (o_bf16) {
                const u32x2 ov = *reinterpret_cast<const u32x2*>(o_bf16 + (size_t)row * ldo + col);
                o[0] = bf16_lo(ov[0]); o[1] = bf16_hi(ov[0]); o[2] = bf16_lo(ov[1]); o[3] = bf16_hi(ov[1]);
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(h_after + (size_t)row * ldh + col);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(h_before + (size_t)row * ldh + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = a[e] - bb[e];
            }
            acc = bf16_lo(g[0]) * o[0] + bf16_hi(g[0]) * o[1] + bf16_lo(g[1]) * o[2] + bf16_hi(g[1]) * o[3];
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (col < d && (lane & 15) == 0) delta[((size_t)b * H + col / DH) * Sq + qi] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// Shared tile machinery.  A "row image" keeps 16-B chunk c of row r at c ^ ((r>>1)&7) (conflict-free
// ds_read_b128 of MFMA A fragments); a "tr image" keeps it at c ^ (((r>>1)&1)<<2) (the 4 rows of a transposed
// read block in 4 different 64-B bank quarters).  Both swizzles are applied to the DMA's SOURCE address.
// MODE 0 = dq kernel   (block = queries; tiles = keys:    images K_row, V_row, K_tr)
// MODE 1 = dkdv kernel (block = keys;    tiles = queries: images Q_row, dO_row, Q_tr, dO_tr  + L / delta)
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256, MODE == 0 ? 3 : 2) void attn64_bwd_kernel(BwdParams p) {
    constexpr int NIMG = MODE == 0 ? 3 : 4;
    constexpr int BUF = NIMG * IMG + (MODE == 1 ? 512 : 0);   // + L[64] | delta[64] floats
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * BUF (MODE 1: 65 KiB, above the static limit)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = p.nblk * p.H * p.B;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int blk = id % p.nblk, bh = id / p.nblk;
    const int h = bh % p.H, b = bh / p.H;
    const int ql = lane & 31, hh = lane >> 5;

    // block side ("own" rows: queries in MODE 0, keys in MODE 1) and tile side
    const int own_len = MODE == 0 ? p.Sq : p.Skv;
    const int tile_len = MODE == 0 ? p.Skv : p.Sq;
    int own = blk * BLK + wid * 32 + ql;
    const bool own_valid = own < own_len;
    own = own_valid ? own : own_len - 1;

    // B-operand fragments held for the whole kernel: lane holds X[own row][d = 16*ks + 8*hh + 0..7]
    bf16x8 f0[4], f1[4];   // MODE 0: Q, dO      MODE 1: K, V
    {
        const bf16* s0 = MODE == 0 ? p.q + ((size_t)b * p.Sq + own) * p.ldq : p.k + ((size_t)b * p.Skv + own) * p.ldk;
        const bf16* s1 = MODE == 0 ? p.dout + ((size_t)b * p.Sq + own) * p.lddo : p.v + ((size_t)b * p.Skv + own) * p.ldv;
        s0 += h * DH + 8 * hh;
        s1 += h * DH + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f0[ks] = *reinterpret_cast<const bf16x8*>(s0 + 16 * ks);
            f1[ks] = *reinterpret_cast<const bf16x8*>(s1 + 16 * ks);
        }
    }
    float own_L = 0.f, own_delta = 0.f;   // MODE 0: per-lane (query) scalars
    if constexpr (MODE == 0) {
        own_L = own_valid ? p.lse[(size_t)bh * p.Sq + own] : 1e30f;
        own_delta = own_valid ? p.delta[(size_t)bh * p.Sq + own] : 0.f;
    }

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    // tile sources: MODE 0: (K, V), MODE 1: (Q, dO)
    const bf16* t0 = MODE == 0 ? p.k : p.q;
    const int ld0 = MODE == 0 ? p.ldk : p.ldq;
    const bf16* t1 = MODE == 0 ? p.v : p.dout;
    const int ld1 = MODE == 0 ? p.ldv : p.lddo;
    auto dma_tile = [&](int tt, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;
            const int row = piece * 8 + (lane >> 3), cpos = lane & 7;
            int tr = tt * TILE + row;
            tr = tr < tile_len ? tr : tile_len - 1;
            const int crow = cpos ^ ((row >> 1) & 7), ctr = cpos ^ (((row >> 1) & 1) << 2);
            const bf16* r0 = t0 + ((size_t)b * tile_len + tr) * ld0 + h * DH;
            const bf16* r1 = t1 + ((size_t)b * tile_len + tr) * ld1 + h * DH;
            const unsigned dst = lds_base + (unsigned)(buf * BUF + piece * 1024);
            glds16(r0 + crow * 8, dst);                 // image 0: t0 rows
            glds16(r1 + crow * 8, dst + IMG);           // image 1: t1 rows
            glds16(r0 + ctr * 8, dst + 2 * IMG);        // image 2: t0 transposed-read
            if constexpr (MODE == 1) glds16(r1 + ctr * 8, dst + 3 * IMG);   // image 3: t1 transposed-read
        }
    };
    // MODE 1: the tile's 64 query rows' L and delta -> LDS (threads 0..31: one f32x4 each)
    f32x4 stat_reg = {0.f, 0.f, 0.f, 0.f};
    auto load_stats = [&](int tt) {
        if constexpr (MODE == 1) {
            if (tid < 32) {
                const int j = (tid & 15) * 4;
                const float* src = (tid < 16 ? p.lse : p.delta) + (size_t)bh * p.Sq;
                const float fill = tid < 16 ? 1e30f : 0.f;   // rows past Sq: P = exp2(-inf) = 0
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qi = tt * TILE + j + e;
                    stat_reg[e] = qi < p.Sq ? src[qi] : fill;
                }
            }
        }
    };
    auto write_stats = [&](int buf) {
        if constexpr (MODE == 1) {
            if (tid < 32)
                *reinterpret_cast<f32x4*>(smem + buf * BUF + NIMG * IMG + (tid < 16 ? 0 : 256) + (tid & 15) * 16) = stat_reg;
        }
    };

    const int row_off = ql * 128, row_swz = (ql >> 1) & 7;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int tr_colbyte = (16 * ((lane >> 4) & 1) + 4 * tr_p) * 2;
    const int tr_row0 = 4 * hh + tr_q;
    const int tr_swz = ((tr_q >> 1) & 1) << 6;

    f32x16 acc0[2], acc1[2];   // MODE 0: dQ^T in acc0 (acc1 unused)   MODE 1: dK^T in acc0, dV^T in acc1
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[0][i] = 0.f; acc0[1][i] = 0.f; acc1[0][i] = 0.f; acc1[1][i] = 0.f; }
    const float c = p.scale_log2;
    const unsigned dstream = drop_stream(p.seed_lo, p.seed_hi, p.layer, bh);

    const int ntile = (tile_len + TILE - 1) / TILE;
    dma_tile(0, 0);
    load_stats(0);
    write_stats(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int tt = 0; tt < ntile; ++tt) {
        const char* base = smem + (tt & 1) * BUF;
        if (tt + 1 < ntile) {
            dma_tile(tt + 1, (tt + 1) & 1);
            load_stats(tt + 1);
        }
        // ---- st = T0 * F0^T  (S^T or S),  dp = T1 * F1^T  (dP^T or dP): tile row in registers, own row on the lane ----
        f32x16 st[2], dp[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[0][i] = 0.f; st[1][i] = 0.f; dp[0][i] = 0.f; dp[1][i] = 0.f; }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = rb * 32 * 128 + row_off + (((2 * ks + hh) ^ row_swz) << 4);
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(base + off);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(base + IMG + off);
                st[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, f0[ks], st[rb], 0, 0, 0);
                dp[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, f1[ks], dp[rb], 0, 0, 0);
            }
        // ---- P, dS for the 32 tile rows this lane holds (tile row = rb*32 + (r&3) + 8*(r>>2) + 4*hh) ----
        bf16x8 pf[4], dsf[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            float Lr[8], Dr[8];
            if constexpr (MODE == 1) {
                // rows 16*s2 + 4*hh + {0..3} and + 8: two f32x4 of L and of delta from LDS (half-wave broadcast)
                const char* sp = base + NIMG * IMG + (16 * s2 + 4 * hh) * 4;
                const f32x4 l0 = *reinterpret_cast<const f32x4*>(sp), l1 = *reinterpret_cast<const f32x4*>(sp + 32);
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(sp + 256), d1 = *reinterpret_cast<const f32x4*>(sp + 288);
#pragma unroll
                for (int e = 0; e < 4; ++e) { Lr[e] = l0[e]; Lr[4 + e] = l1[e]; Dr[e] = d0[e]; Dr[4 + e] = d1[e]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int rr = 8 * (s2 & 1) + j;
                const int trow = tt * TILE + (s2 >> 1) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hh;
                const float L = MODE == 0 ? own_L : Lr[j];
                const float dl = MODE == 0 ? own_delta : Dr[j];
                float sv = st[s2 >> 1][rr] * c - L;
                if constexpr (MODE == 0) sv = trow < p.Skv ? sv : -1e30f;   // ragged last key tile
                const float pr = __builtin_amdgcn_exp2f(sv);
                float g = dp[s2 >> 1][rr], pd = pr;
                if (p.drop_thr) {
                    const int qi = MODE == 0 ? own : trow, kj = MODE == 0 ? trow : own;
                    const bool keep = drop_keep(dstream, qi, kj, p.drop_thr);
                    g = keep ? g * p.keep_scale : 0.f;
                    pd = keep ? pr * p.keep_scale : 0.f;
                }
                dsf[s2][j] = (bf16)(pr * (g - dl) * p.scale);
                if constexpr (MODE == 1) pf[s2][j] = (bf16)pd;
            }
        }
        // ---- acc^T[d][own] += T^T[d][tile row] * X[tile row][own]  (transposed reads of the tr images) ----
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int colb = (tr_colbyte + 64 * db) ^ tr_swz;
                const char* a0 = base + 2 * IMG + (16 * s2 + tr_row0) * 128 + colb;
                const bf16x8 t0f = cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0)),
                                        __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 8 * 128)));
                acc0[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t0f, dsf[s2], acc0[db], 0, 0, 0);   // K^T dS^T | Q^T dS
                if constexpr (MODE == 1) {
                    const char* a1 = a0 + IMG;
                    const bf16x8 t1f = cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a1)),
                                            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a1 + 8 * 128)));
                    acc1[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t1f, pf[s2], acc1[db], 0, 0, 0);  // dO^T P
                }
            }
        if (tt + 1 < ntile) write_stats((tt + 1) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane (own row, half hh) owns d = 32*db + 8*g + 4*hh + 0..3 ----
    if (!own_valid) return;
    bf16* o0 = MODE == 0 ? p.dq + ((size_t)b * p.Sq + own) * p.lddq : p.dk + ((size_t)b * p.Skv + own) * p.lddk;
    bf16* o1 = MODE == 0 ? nullptr : p.dv + ((size_t)b * p.Skv + own) * p.lddv;
    if (p.rope_cos) {
        // backward of the half-split RoPE (reference DiT.py:126-129 forward: lo' = lo cos - hi sin, hi' = hi cos + lo sin):
        // d lo = g_lo cos + g_hi sin, d hi = g_hi cos - g_lo sin, on the fp32 accumulators — the lane holds d = j (block 0) and
        // d = j + 32 (block 1) of its row for j = 8 g + 4 hh + e, so the pair never leaves the lane.  (This replaced a separate
        // in-place pass over the bf16 gradients: 68 us per layer, and one bf16 rounding less.)
        const float* ct = p.rope_cos + (size_t)own * 32 + 4 * hh;
        const float* st = p.rope_sin + (size_t)own * 32 + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(ct + 8 * g), s4 = *reinterpret_cast<const f32x4*>(st + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = acc0[0][4 * g + e], hi = acc0[1][4 * g + e];
                acc0[0][4 * g + e] = lo * c4[e] + hi * s4[e];
                acc0[1][4 * g + e] = hi * c4[e] - lo * s4[e];
            }
        }
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = h * DH + 32 * db + 8 * g + 4 * hh;
            u32x2 s0;
            s0[0] = pack_bf16x2(acc0[db][4 * g], acc0[db][4 * g + 1]);
            s0[1] = pack_bf16x2(acc0[db][4 * g + 2], acc0[db][4 * g + 3]);
            *reinterpret_cast<u32x2*>(o0 + col) = s0;
            if constexpr (MODE == 1) {
                u32x2 s1;
                s1[0] = pack_bf16x2(acc1[db][4 * g], acc1[db][4 * g + 1]);
                s1[1] = pack_bf16x2(acc1[db][4 * g + 2], acc1[db][4 * g + 3]);
                *reinterpret_cast<u32x2*>(o1 + col) = s1;
            }
        }
}

}  // namespace

hipError_t launch_attention_delta(const void* dout, int lddo, const void* o_bf16, int ldo, const float* h_after,
                                  const float* h_before, int ldh, float* delta, int B, int H, int Sq, hipStream_t s) {
    if ((lddo % 4) || (o_bf16 && ldo % 4) || (!o_bf16 && (ldh % 4 || !h_after || !h_before))) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((B * Sq + 3) / 4), dim3(256), 0, s, (const bf16*)dout, lddo,
                       (const bf16*)o_bf16, ldo, h_after, h_before, ldh, delta, B, H, Sq);
    return hipGetLastError();
}

// fused backward (head_dim 64): lse (log2 domain, from the TRAIN forward) and delta (launch_attention_delta) given
hipError_t launch_attention_bwd64(const AttnBwdArgs& a, const float* lse, const float* delta, hipStream_t s) {
    if (a.dh != DH || a.B <= 0 || a.H <= 0 || a.Sq <= 0 || a.Skv <= 0 || !lse || !delta) return hipErrorInvalidValue;
    if ((a.ldq | a.ldk | a.ldv | a.lddo) % 8 || (a.lddq | a.lddk | a.lddv) % 4) return hipErrorInvalidValue;
    BwdParams p;
    p.q = (const bf16*)a.q; p.ldq = a.ldq; p.k = (const bf16*)a.k; p.ldk = a.ldk; p.v = (const bf16*)a.v; p.ldv = a.ldv;
    p.dout = (const bf16*)a.dout; p.lddo = a.lddo;
    p.dq = (bf16*)a.dq; p.lddq = a.lddq; p.dk = (bf16*)a.dk; p.lddk = a.lddk; p.dv = (bf16*)a.dv; p.lddv = a.lddv;
    p.lse = lse; p.delta = delta; p.B = a.B; p.H = a.H; p.Sq = a.Sq; p.Skv = a.Skv;
    p.scale = a.scale; p.scale_log2 = a.scale * 1.4426950408889634f;
    p.drop_thr = dropout_threshold(a.dropout_p);
    p.keep_scale = p.drop_thr ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
    p.seed_lo = (unsigned)(a.seed & 0xFFFFFFFFu); p.seed_hi = (unsigned)(a.seed >> 32); p.layer = a.layer;
    p.rope_cos = a.rope_cos; p.rope_sin = a.rope_sin;
    if ((a.rope_cos == nullptr) != (a.rope_sin == nullptr) || (a.rope_cos && a.Sq != a.Skv)) return hipErrorInvalidValue;
    constexpr int LDS0 = 2 * 3 * IMG, LDS1 = 2 * (4 * IMG + 512);
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&attn64_bwd_kernel<1>)}, LDS1)) return e;
    p.nblk = (a.Sq + BLK - 1) / BLK;
    hipLaunchKernelGGL((attn64_bwd_kernel<0>), dim3(p.nblk * a.H * a.B), dim3(256), LDS0, s, p);
    p.nblk = (a.Skv + BLK - 1) / BLK;
    hipLaunchKernelGGL((attn64_bwd_kernel<1>), dim3(p.nblk * a.H * a.B), dim3(256), LDS1, s, p);
    return hipGetLastError();
}

}  // namespace ditto
