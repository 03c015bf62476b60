// attention.hip — fused (flash-style) attention for gfx950, head_dim 64, bf16 in / fp32 accumulate,
// plus a generic-head_dim fallback built from the GEMM family.
//
// Replaces  softmax(q k^T / sqrt(dh)) v  of the self-attention (reference src/components/DiT.py:131-134,
// followed by the head merge + residual :137-139, NO out-proj) and of nn.MultiheadAttention's cross
// attention (src/components/DiT.py:144-148 -> torch F.multi_head_attention_forward: q*sqrt(1/dh), bmm,
// softmax, bmm).  No masks anywhere in the reference.  The [B,H,N,N] score matrix the
// reference materialises is never written.
//
// Roofline: MFMA-bound; algorithmic FLOPs = 4*B*H*Sq*Skv*dh per launch.  K/V of one head at Skv = 4096 is
// 1 MiB, so the Sq/128 workgroups of a (batch, head) — placed on ONE XCD by the block remap — re-read it
// from that XCD's L2, not from HBM.
//
// Structure per workgroup (256 threads = 4 waves, 32 query rows per wave, 64-key tiles):
//   * S^T = K * Q^T with v_mfma_f32_32x32x16_bf16 (K rows as the A operand from LDS, Q^T as the B operand
//     held in 16 registers for the whole kernel): the accumulator has the QUERY on the lane and 16 keys
//     in registers, so the row max / row sum are in-lane plus one exchange with lane^32.
//   * P^T stays in registers: packed to bf16 it IS the B operand of the next product
//     O^T += V^T * P^T (guide §3 "an accumulator tile as the next MFMA's operand"; the k-order
//     inside a step is permuted, and V^T is fetched with the same permutation).
//   * V^T fragments come from the row-major V tile in LDS via ds_read_b64_tr_b16 (hardware transpose).
//   * K/V tiles: global -> registers (issued before the tile's math) -> LDS (written after it), two LDS
//     buffers, one barrier per tile (T14 async-STAGE split).
//   * LDS swizzles: K rows (128 B) chunk ^= (key>>1)&7 -> conflict-free ds_read_b128; V rows chunk ^=
//     ((key>>1)&1)<<2 -> the 4 rows of a transposed-read block fall in 4 different 64-B bank quarters.
//   * online softmax in the exp2 domain with fp32 running (max, sum); O^T rescale is a per-lane scalar.
#include <type_traits>

#include "attn_common.h"

namespace ditto {

// bit 0: K/V tiles by LDS-DMA instead of register staging; bit 1: V fragments prefetched ahead of the softmax
// (needs bit 0); bit 4 (16): ditto_attention_bf16's q is pre-scaled
// (unit tests of attn64v2); bit 5 (32): run the kernels above on pre-scaled q instead of attn64v2 (A/B); bit 6 (64):
// attn64v2 at 2 waves per SIMD with the V prefetch instead of 3 without; bit 7 (128): no deep-prefetch instantiation
// on small grids; bit 8 (256): never attn64v3 (the software-pipelined kernel), bit 9 (512): attn64v3 wherever Skv % 128 == 0; bit 10 (1024): its 8-wave (256 queries per
// workgroup) form always, bit 11 (2048): never; bit 13 (8192): the training forward on the older kernel (attn64_kernel<.., TRAIN>)
// instead of attn64v2's TRAIN instantiations (attention_train.hip) — A/B only; bit 17 (131072): never attn64p (attention_p.hip, the
// 64-queries-per-wave kernel of round 6), bit 18 (262144): attn64p whatever the grid, bit 19 (524288): its ring of 3 tile pairs, bit 20 (1048576): never attn64q (the
// pipelined, optimistic form that runs wherever attn64p would and Skv is whole tiles: attn64q.h), bit 21 (2097152): attn64q's two forms (K fragments held for
// block B or not) swapped between the residual and the plain epilogue (A/B).  (Bits 12, 14, 15, 16 selected the round-3 /
// round-5 experiments attn64v4 / attn64w4 / KPF, measured equal or slower and deleted in round 6: DESIGN.md "tried".)
// ditto_set_option("attn_flags")
int g_attn_flags = 3;
// attn64p from this many of its 256-query workgroups on (by class rows): 768 = three per CU, i.e. B >= 16 at N = 1024, 12 heads (in-model
// A/B, profiles/r06_attn_ab.txt: B = 12 loses 3 % of its attention time, B = 16 gains 9 %).  ditto_set_option("attn64p_min_wgs")
int g_attn64p_min_wgs = [] { const char* e = getenv("DITTO_ATTN64P_MIN_WGS"); const int v = e ? atoi(e) : 768; return v > 0 ? v : 768; }();
// ... and without a residual epilogue (the cross-attention) from this many: in-model at B = 4 / 8 / 12 the plain form gains 8 / 6 / 4 %
// of its launch where the residual form ties or loses (profiles/r06_attn64q.txt).  ditto_set_option("attn64p_min_wgs_plain")
int g_attn64p_min_wgs_plain = [] { const char* e = getenv("DITTO_ATTN64P_MIN_WGS_PLAIN"); const int v = e ? atoi(e) : 192; return v > 0 ? v : 192; }();

namespace {

#include "attn64v2.h"   // constants + attn64v2_kernel

template <int V>
struct IC1 { static constexpr int value = V; };

// __launch_bounds__(256, 2): 2 waves per SIMD => a 256-register budget, so the MFMA accumulators (S^T, O^T: 64
// registers) stay in VGPRs.  With the default budget hipcc parks them in AGPRs and moves all 64 through
// v_accvgpr_read/write around every softmax (127 extra VALU per tile, as much as the softmax itself).
template <bool RESID, bool DMA, bool PFV, bool TRAIN = false, int WPS = 2>
__global__ __launch_bounds__(256, WPS) void attn64_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * KV_TILE_BYTES];  // [buf][K|V]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = p.nqb * p.H * p.B;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int qb = id % p.nqb, bh = id / p.nqb;
    const int h = bh % p.H, b = bh / p.H;

    const int ql = lane & 31, hh = lane >> 5;
    const int q0 = qb * QBLK + wid * 32;
    int qrow = q0 + ql;
    const bool qvalid = qrow < p.Sq;
    qrow = qvalid ? qrow : p.Sq - 1;

    // Q^T B-operand fragments: lane holds Q[query ql][d = 16*ks + 8*hh + 0..7]
    bf16x8 qf[4];
    {
        const bf16* qp = p.q + ((size_t)b * p.Sq + qrow) * p.ldq + h * DH + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }

    // K/V staging: thread owns chunks (key = tid>>3 [+32], c = tid&7)
    const int skey = tid >> 3, sc = tid & 7;
    const bf16* kbase = p.k + (size_t)b * p.Skv * p.ldk + h * DH + sc * 8;
    const bf16* vbase = p.v + (size_t)b * p.Skv * p.ldv + h * DH + sc * 8;
    u32x4 kreg[2], vreg[2];
    // DMA variant: K/V tiles go global -> LDS directly (asm LDS-DMA, gemm_common.h glds16): no staging registers,
    // no ds_write pass, no compiler-visible load to wait on.  One wave moves pieces 2*wid, 2*wid+1 (8 keys x 128 B
    // each) of K and of V; the two LDS swizzles are chunk permutations inside a row, applied to the SOURCE address.
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    auto dma_kv = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;
            const int row = piece * 8 + (lane >> 3), cpos = lane & 7;
            int key = kt * KBLK + row;
            key = key < p.Skv ? key : p.Skv - 1;
            const int ck = cpos ^ ((row >> 1) & 7), cv = cpos ^ (((row >> 1) & 1) << 2);
            const bf16* ks = p.k + ((size_t)b * p.Skv + key) * p.ldk + h * DH + ck * 8;
            const bf16* vs = p.v + ((size_t)b * p.Skv + key) * p.ldv + h * DH + cv * 8;
            glds16(ks, lds_base + (unsigned)(buf * 2 * KV_TILE_BYTES + piece * 1024));
            glds16(vs, lds_base + (unsigned)(buf * 2 * KV_TILE_BYTES + KV_TILE_BYTES + piece * 1024));
        }
    };
    auto load_kv = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int key = kt * KBLK + skey + 32 * i;
            key = key < p.Skv ? key : p.Skv - 1;
            kreg[i] = *reinterpret_cast<const u32x4*>(kbase + (size_t)key * p.ldk);
            vreg[i] = *reinterpret_cast<const u32x4*>(vbase + (size_t)key * p.ldv);
        }
    };
    auto write_kv = [&](int buf) {
        char* kb = smem + buf * 2 * KV_TILE_BYTES;
        char* vb = kb + KV_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = skey + 32 * i;
            *reinterpret_cast<u32x4*>(kb + key * 128 + ((sc ^ ((key >> 1) & 7)) << 4)) = kreg[i];
            *reinterpret_cast<u32x4*>(vb + key * 128 + ((sc ^ (((key >> 1) & 1) << 2)) << 4)) = vreg[i];
        }
    };

    // K A-operand read offsets: lane reads K[key = kb*32 + ql][chunk = 2*ks + hh]
    const int k_row_off = ql * 128, k_swz = (ql >> 1) & 7;  // (kb*32 + ql)>>1 & 7 == (ql>>1)&7
    // V^T transposed-read address pieces: group g = lane>>4 -> d half (g&1), key half hh; i = lane&15
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int tr_colbyte = (16 * ((lane >> 4) & 1) + 4 * tr_p) * 2;  // + 64*dblk
    const int tr_row0 = 4 * hh + tr_q;                               // + 16*s (+8)
    // rows r = 16*s + {0,8} + 4*hh + tr_q  ->  (r>>1)&1 == (tr_q>>1)&1 (16s, 8, 4hh are multiples of 4)
    const int tr_swz = ((tr_q >> 1) & 1) << 6;

    f32x16 ot[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { ot[0][i] = 0.f; ot[1][i] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;
    const float c = p.scale_log2;
    DropStream dstream{};
    if constexpr (TRAIN) dstream = drop_stream(p.seed_lo, p.seed_hi, p.layer, bh);

    const int nkt = (p.Skv + KBLK - 1) / KBLK;
    if constexpr (DMA) {
        dma_kv(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        load_kv(0);
        write_kv(0);
    }
    __syncthreads();

    auto tile_body = [&](int kt, auto MASKED) {
        const char* kb = smem + (kt & 1) * 2 * KV_TILE_BYTES;
        const char* vb = kb + KV_TILE_BYTES;
        if (kt + 1 < nkt) {
            if constexpr (DMA) dma_kv(kt + 1, (kt + 1) & 1);   // lands during this tile's math
            else load_kv(kt + 1);
        }

        // ---- S^T[key][query] for the 64 keys of this tile ----
        f32x16 st[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[0][i] = 0.f; st[1][i] = 0.f; }
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + kb2 * 32 * 128 + k_row_off +
                                                                   (((2 * ks + hh) ^ k_swz) << 4));
                st[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[kb2], 0, 0, 0);
            }
        if constexpr (decltype(MASKED)::value) {  // ragged last tile only: keys >= Skv never contribute
            const int kbase_idx = kt * KBLK + 4 * hh;
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase_idx + kb2 * 32 + (r & 3) + 8 * (r >> 2);
                    if (key >= p.Skv) st[kb2][r] = -1e30f;
                }
        }

        // PFV: the 16 transposed V-fragment reads of this tile are ISSUED here, ahead of the softmax, and
        // consumed after it: their LDS latency hides under the VALU block (32 VGPRs, made available by the
        // DMA staging).  sched_barrier keeps hipcc from sinking them back next to the MFMAs.
        bf16x8 vf[8];
        if constexpr (PFV) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const int colb = (tr_colbyte + 64 * db) ^ tr_swz;
                    const char* a0 = vb + (16 * s2 + tr_row0) * 128 + colb;
                    vf[s2 * 2 + db] = cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0)),
                                           __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 8 * 128)));
                }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- online softmax (this lane = one query; its other 32 keys live in lane^32) ----
        float mloc = fmaxf(st[0][0], st[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, fmaxf(st[0][r], st[1][r]));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        // raise the running max only if some query of the wave needs it (wave-uniform decision, taken BEFORE
        // any P of this tile is exponentiated and with the previous tile's P*V complete: guide T13 safe order)
        if (!__all((mloc - m_run) * c <= RESCALE_THR_LOG2)) {
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { ot[0][i] *= alpha; ot[1][i] *= alpha; }
        }
        const float mc = m_run * c;
        float psum = 0.f;
        bf16x8 pf[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                e[j] = __builtin_amdgcn_exp2f(st[s2 >> 1][8 * (s2 & 1) + j] * c - mc);
                psum += e[j];
            }
            if constexpr (TRAIN) {   // dropout on the probabilities; the row sum l stays that of the full softmax
                if (p.drop_thr) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int rr = 8 * (s2 & 1) + j;
                        const int key = kt * KBLK + (s2 >> 1) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hh;
                        e[j] = drop_keep(dstream, qrow, key, p.drop_thr) ? e[j] * p.keep_scale : 0.f;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[s2][j] = (bf16)e[j];
        }
        l_run += psum;

        // ---- O^T[d][query] += V^T[d][key] * P^T[key][query] ----
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                if constexpr (PFV) {
                    ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2 * 2 + db], pf[s2], ot[db], 0, 0, 0);
                } else {
                    const int colb = (tr_colbyte + 64 * db) ^ tr_swz;
                    const char* a0 = vb + (16 * s2 + tr_row0) * 128 + colb;
                    const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0));
                    const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 8 * 128));
                    ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat4(v0, v1), pf[s2], ot[db], 0, 0, 0);
                }
            }

        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt+1 have landed
        } else {
            if (kt + 1 < nkt) write_kv((kt + 1) & 1);
        }
        __syncthreads();
    };

    const bool ragged = (p.Skv & (KBLK - 1)) != 0;
    const int nfull = ragged ? nkt - 1 : nkt;
    for (int kt = 0; kt < nfull; ++kt) tile_body(kt, std::false_type{});
    if (ragged) tile_body(nkt - 1, std::true_type{});

    // ---- epilogue: normalise; lane (query ql, half hh) owns d = 32*db + 8*g + 4*hh + 0..3 ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (!qvalid) return;
    const size_t grow = (size_t)b * p.Sq + qrow;
    if constexpr (TRAIN) {
        if (p.lse && hh == 0) p.lse[(size_t)bh * p.Sq + qrow] = m_run * c + __builtin_amdgcn_logf(l_tot);
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = h * DH + 32 * db + 8 * g + 4 * hh;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ot[db][4 * g + e] * inv;
            if constexpr (RESID) {
                attn_resid_update(p, grow, col, o);   // x = attn_out + residual (src/components/DiT.py:139)
            } else {
                u32x2 st2;
                st2[0] = pack_bf16x2(o[0], o[1]);
                st2[1] = pack_bf16x2(o[2], o[3]);
                *reinterpret_cast<u32x2*>(p.out + grow * p.ldo + col) = st2;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// attn64v3: attn64v2's arithmetic as a THREE-tile software pipeline whose MFMAs and softmax VALU work are interleaved BY
// HAND inside every wave.  Why: in attn64v2 a wave's instruction stream alternates between an MFMA burst (S = K Q^T), a
// VALU burst (max, 32 exp2, 16 converts) and a second MFMA burst (O += V P), and the cycle count per 64-key tile of a SIMD
// is the SUM of its waves' VALU and MFMA times (measured: 1476 cycles per wave and tile = 776 VALU + 640 MFMA; the SQ
// counters read VALU 53 % + MFMA 43 %): three waves per SIMD do not overlap the two pipes for it.  One wave does, when an
// independent VALU chunk follows every MFMA: measured on the full-row GEMM's loop (tools/build_diag.sh -DDITTO_DIAG_FR_VALU),
// two transcendentals + three plain fp32 operations (44 VALU cycles) behind each 32-cycle 32x32x16 MFMA cost 10 cycles, one
// transcendental 3.5, four plain FMAs 6.5 — while two PACKED fp32 FMAs cost 22 (they block the matrix pipe: none here).
//   iteration t of a wave (20 slots, one MFMA + one VALU chunk each, a scheduling barrier behind every slot):
//     slots 0..7    S'(t+1) = K(t+1) Q^T - m   (two chains of four MFMAs, alternating)  |  exp2 of S'(t) -> P(t), 2 per slot
//     slots 8..19   O += V(t-1) P(t-1), l += 1 P(t-1)   (4 x {2 + 1} MFMAs)             |  the other 16 exp2 of S'(t), then
//                                                                                          the row maximum of S'(t+1)
//   so tile t's probabilities are computed while the matrix pipe works on tile t+1's scores and tile t-1's output.
//   A raise of the running maximum (rare: deferred until a row exceeds it by 2^8, as in attn64v2) happens at the END of an
//   iteration, by a whole number of octaves: S'(t+1), -m, O, l and the not yet accumulated P(t) are rescaled together (P in
//   bf16 by a power of two: exact).
//   K tiles live in a ring of 4, V tiles in a ring of 5 (tile t+3 is in flight while V(t-1) is still read): 72 KiB per
//   workgroup, TWO workgroups per CU (256 registers per wave: S' twice, P twice, O, l, -m, Q).
// Contract: as attn64v2 (pre-scaled q) and Skv % 128 == 0 (an even number of whole tiles: the two register sets of S'
// and P swap roles every iteration and the loop is unrolled by two); other shapes take attn64v2.
// ------------------------------------------------------------------------------------------------
constexpr int V3_KSLOTS = 4, V3_VSLOTS = 5;
constexpr int V3_LDS = (V3_KSLOTS + V3_VSLOTS) * KV_TILE_BYTES;   // 72 KiB

// NW = waves per workgroup = 32-query blocks sharing the K/V tiles: 4 (128 queries, two workgroups per CU) or 8 (256
// queries, one workgroup per CU: half the K/V LDS-DMA bytes and ring writes per FLOP; p.nqb counts blocks of 32 NW).
template <bool RESID, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void attn64v3_kernel(AttnParams p) {
    constexpr int PW = 8 / NW;                                      // 1-KiB DMA pieces per wave per tile image (2 or 1)
    extern __shared__ __attribute__((aligned(16))) char smem[];    // [4 K tiles][5 V tiles]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = p.nqb * p.H * p.B;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int qb = id % p.nqb, bh = id / p.nqb;
    const int h = bh % p.H, b = bh / p.H;
    const int ql = lane & 31, hh = lane >> 5;
    int qrow = qb * (32 * NW) + wid * 32 + ql;
    const bool qvalid = qrow < p.Sq;
    qrow = qvalid ? qrow : p.Sq - 1;

    bf16x8 qf[4];
    {
        const bf16* qp = p.q + ((size_t)b * p.Sq + qrow) * p.ldq + h * DH + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }
    const int nkt = p.Skv / KBLK;                                  // even, >= 2
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    // this lane's two (row, chunk) DMA sources of tile 0; tile kt is + kt * 64 rows (LDS swizzles applied on the source)
    const bf16 *ksrc[2], *vsrc[2];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int row = (wid * PW + i) * 8 + (lane >> 3), cpos = lane & 7;
        ksrc[i] = p.k + ((size_t)b * p.Skv + row) * p.ldk + h * DH + (cpos ^ ((row >> 1) & 7)) * 8;
        vsrc[i] = p.v + ((size_t)b * p.Skv + row) * p.ldv + h * DH + (cpos ^ (((row >> 1) & 1) << 2)) * 8;
    }
    const size_t kstep = (size_t)KBLK * p.ldk, vstep = (size_t)KBLK * p.ldv;
    int issued = 0;                                               // tiles whose DMA has been issued
    unsigned ik = 0, iv = 0;                                      // ring slots of the next tile to issue
    auto dma_next = [&]() {                                       // 2 PW loads per wave: PW K pieces, PW V pieces
        if (issued < nkt) {
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const int piece = wid * PW + i;
#ifndef DITTO_DIAG_ATTN_NODMA   // tools/build_diag.sh: attn64v3 without its K/V tile traffic (timing only)
                glds16(ksrc[i], lds_base + (unsigned)(ik * KV_TILE_BYTES + piece * 1024));
                glds16(vsrc[i], lds_base + (unsigned)((V3_KSLOTS + iv) * KV_TILE_BYTES + piece * 1024));
#endif
                ksrc[i] += kstep; vsrc[i] += vstep;
            }
            ++issued;
            ik = ik + 1 == V3_KSLOTS ? 0 : ik + 1;
            iv = iv + 1 == V3_VSLOTS ? 0 : iv + 1;
        }
    };
    // tile `need` has landed for this wave once only the tiles issued after it are in flight; then for everyone
    auto wait_tile = [&](int need) {
        const int younger = issued - 1 - need;
        if constexpr (PW == 2) {
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#ifndef DITTO_DIAG_ATTN_NOBAR   // racy: timing only
        __syncthreads();
#endif
    };

    const int k_row_off = ql * 128, k_swz = (ql >> 1) & 7;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int tr_colbyte = (16 * ((lane >> 4) & 1) + 4 * tr_p) * 2;
    const int tr_row0 = 4 * hh + tr_q;
    const int tr_swz = ((tr_q >> 1) & 1) << 6;

    f32x16 ot[2], lsum, cneg;    // cneg: every register = -m_run (the score chains' initial accumulator)
    f32x16 sX[2], sY[2];         // S' of two tiles in flight
    u32x4 pP[4], pQ[4];          // P (bf16 pairs) of two tiles in flight
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { ot[0][i] = 0.f; ot[1][i] = 0.f; lsum[i] = 0.f; cneg[i] = 0.f; }
    float dm = 0.f;              // row maximum of the newest S' (relative to the running maximum)

    // One iteration.  QK: scores of tile t+1 into stn; EX: probabilities of tile t from stc into pfc; PV: tile t-1's
    // probabilities pfp into O and l.  kq / vq: ring slots of K(t+1) and V(t-1).
    auto body = [&](auto QK_, auto EX_, auto PV_, unsigned kq, unsigned vq, f32x16 (&stc)[2], f32x16 (&stn)[2],
                    u32x4 (&pfp)[4], u32x4 (&pfc)[4]) {
        constexpr bool QK = decltype(QK_)::value != 0, EX = decltype(EX_)::value != 0, PV = decltype(PV_)::value != 0;
        const char* kb = smem + kq * KV_TILE_BYTES;
        const char* vb = smem + (V3_KSLOTS + vq) * KV_TILE_BYTES;
        auto kfrag = [&](int j) {                                   // operand of score MFMA j: chain j & 1, k-step j >> 1
#ifdef DITTO_DIAG_ATTN_NOKV
            return qf[j & 3];
#endif
            return *reinterpret_cast<const bf16x8*>(kb + (j & 1) * 32 * 128 + k_row_off + (((2 * (j >> 1) + hh) ^ k_swz) << 4));
        };
        auto vfrag = [&](int s2, int db) {
            const int colb = (tr_colbyte + 64 * db) ^ tr_swz;
            const char* a0 = vb + (16 * s2 + tr_row0) * 128 + colb;
#ifdef DITTO_DIAG_ATTN_NOTR     // V fragments by ONE plain 16-byte read (wrong data: timing of the transposed reads)
            return *reinterpret_cast<const bf16x8*>(vb + (16 * s2) * 128 + db * 2048 + k_row_off + (((hh) ^ k_swz) << 4));
#endif
#ifdef DITTO_DIAG_ATTN_NOKV     // no LDS fragment reads at all
            return qf[(s2 + db) & 3];
#endif
            return cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0)),
                        __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 8 * 128)));
        };
        // the exponentials of a pair are ISSUED behind a slot's MFMA and PACKED in front of the next slot's: a convert right
        // behind its own v_exp waits out the transcendental latency, and the in-order wave cannot issue its next MFMA meanwhile
        float eh0 = 0.f, eh1 = 0.f;
        auto exp_issue = [&](int q) {                               // P elements 2q, 2q+1 of the tile
            if constexpr (EX) {
                const int n = 2 * q;
#ifdef DITTO_DIAG_ATTN_NOEXP    // tools/build_diag.sh: attn64v3 without its transcendentals (timing only)
                eh0 = stc[n >> 4][n & 15]; eh1 = stc[n >> 4][(n & 15) + 1];
#else
                eh0 = __builtin_amdgcn_exp2f(stc[n >> 4][n & 15]);
                eh1 = __builtin_amdgcn_exp2f(stc[n >> 4][(n & 15) + 1]);
#endif
            }
        };
        auto exp_pack = [&](int q) {
            if constexpr (EX) {
                if (q >= 0) pfc[q >> 2][q & 3] = pack_bf16x2(eh0, eh1);
            }
        };
        auto max8 = [&](int g) {                                    // fold S'(t+1) elements 8g .. 8g+7 into dm
            if constexpr (QK) {
                float a = g == 0 ? stn[0][0] : dm;
#pragma unroll
                for (int r = 0; r < 8; ++r) a = fmaxf(a, stn[g >> 1][8 * (g & 1) + r]);
                dm = a;
            }
        };
        // operand queue: the iteration's LDS fragments in the order the MFMAs consume them — K fragments 0..7 of the score
        // chains, then V fragments 0..7 (2 s2 + db) — read QD fragments ahead of their MFMA (an LDS read issued two MFMAs
        // ahead stalled every MFMA on its operand: 3 400 cycles per iteration for 1 400 of work)
        constexpr int QD = 5, QN = QD + 1;
        constexpr int NK = QK ? 8 : 0, NF = NK + (PV ? 8 : 0);
        bf16x8 opq[QN];
        auto fetch = [&](int f) {                                   // fragment f of the iteration's list -> its queue slot
            if (f < NF) opq[f % QN] = f < NK ? kfrag(f) : vfrag((f - NK) >> 1, (f - NK) & 1);
        };
#pragma unroll
        for (int f = 0; f < QD; ++f) fetch(f);
        __builtin_amdgcn_sched_barrier(0);
        // ---- slots 0..7: the score chains of tile t+1 | 16 of tile t's exponentials ----
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (QK) {
                fetch(j + QD);
#ifdef DITTO_DIAG_ATTN_NOMFMA   // attn64v3 without its MFMAs (operands still fetched): timing only
                asm volatile("" :: "v"(opq[j % QN]));
                if ((j >> 1) == 0) stn[j & 1] = cneg;
#else
                stn[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opq[j % QN], qf[j >> 1], (j >> 1) == 0 ? cneg : stn[j & 1], 0, 0, 0);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            exp_pack(j - 1);
            exp_issue(j);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- slots 8..19: O += V P, l += 1 P of tile t-1 | the other 16 exponentials, then the row maximum of S'(t+1) ----
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                if constexpr (PV) {
                    if (w < 2) {
                        const int f = NK + 2 * s2 + w;
                        fetch(f + QD);
#ifdef DITTO_DIAG_ATTN_NOMFMA
                        asm volatile("" :: "v"(opq[f % QN]), "v"(pfp[s2]));
#else
                        ot[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opq[f % QN], __builtin_bit_cast(bf16x8, pfp[s2]), ot[w], 0, 0, 0);
#endif
                    } else {
#ifndef DITTO_DIAG_ATTN_NOMFMA
                        lsum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, __builtin_bit_cast(bf16x8, pfp[s2]), lsum, 0, 0, 0);
#endif
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (w == 0) exp_pack(s2 == 0 ? 7 : 8 + 2 * s2 - 1);   // the pair issued in the last exponential slot
                if (w == 1) exp_pack(8 + 2 * s2);
                if (w < 2) exp_issue(8 + 2 * s2 + w);
                else max8(s2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        exp_pack(15);
#ifndef DITTO_DIAG_ATTN_NORAISE
        if constexpr (QK) dm = fmaxf(dm, __shfl_xor(dm, 32, 64));
#endif
    };
    // Raise the running maximum by `up` octaves (a whole number per row): everything not yet summed is rescaled.
    auto raise = [&](auto FIRST_, f32x16 (&stn)[2], u32x4 (&pfc)[4]) {
        constexpr bool FIRST = decltype(FIRST_)::value != 0;
#ifdef DITTO_DIAG_ATTN_NORAISE   // tools/build_diag.sh: no running-maximum check after the first tile (timing only)
        if (FIRST) {
#else
        if (FIRST || !__all(dm <= RESCALE_THR_LOG2)) {
#endif
            const float up = FIRST ? ceilf(dm) : ceilf(fmaxf(dm, 0.f));   // first tile: the running maximum IS this tile's
#pragma unroll
            for (int i = 0; i < 16; ++i) { stn[0][i] -= up; stn[1][i] -= up; cneg[i] -= up; }
            if constexpr (!FIRST) {
                const float alpha = __builtin_amdgcn_exp2f(-up);
#pragma unroll
                for (int i = 0; i < 16; ++i) { ot[0][i] *= alpha; ot[1][i] *= alpha; lsum[i] *= alpha; }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned w = pfc[s2][j];
                        pfc[s2][j] = pack_bf16x2(__uint_as_float(w << 16) * alpha, __uint_as_float(w & 0xFFFF0000u) * alpha);
                    }
            }
        }
    };

    // ---- prologue: tiles 0, 1 in flight; scores of tile 0; its maximum becomes the running maximum ----
    using T = IC1<1>; using F = IC1<0>;
    dma_next(); dma_next();
    wait_tile(0);
    dma_next();                                                     // tile 2
    unsigned kq = 0, vq = 0;                                        // ring slots of K(t+1), V(t-1) for the next iteration
    auto kadv = [&]() { kq = kq + 1 == V3_KSLOTS ? 0 : kq + 1; };
    auto vadv = [&]() { vq = vq + 1 == V3_VSLOTS ? 0 : vq + 1; };
    body(T{}, F{}, F{}, kq, vq, sY, sX, pQ, pP);                     // S'(0) -> sX
    raise(T{}, sX, pP);
    kadv();
    wait_tile(1);
    dma_next();                                                     // tile 3
    // ---- t = 0: scores of tile 1 -> sY, probabilities of tile 0 (sX) -> pP ----
    body(T{}, T{}, F{}, kq, vq, sX, sY, pQ, pP);
    raise(F{}, sY, pP);
    kadv();
    // ---- t = 1 .. nkt-2 in pairs ----
    for (int t = 1; t + 1 < nkt; t += 2) {
        wait_tile(t + 1);
        dma_next();                                                 // tile t + 3
        body(T{}, T{}, T{}, kq, vq, sY, sX, pP, pQ);                 // t odd: S'(t+1) -> sX, P(t) from sY -> pQ, O += V(t-1) pP
        raise(F{}, sX, pQ);
        kadv(); vadv();
        wait_tile(t + 2);
        dma_next();
        body(T{}, T{}, T{}, kq, vq, sX, sY, pQ, pP);                 // t+1 even: S'(t+2) -> sY, P(t+1) from sX -> pP, O += V(t) pQ
        raise(F{}, sY, pP);
        kadv(); vadv();
    }
    // ---- t = nkt-1 (odd): no scores left; probabilities of the last tile from sY -> pQ, O += V(nkt-2) pP ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    body(F{}, T{}, T{}, kq, vq, sY, sX, pP, pQ);
    vadv();
    // ---- O += V(nkt-1) pQ ----
    body(F{}, F{}, T{}, kq, vq, sY, sX, pQ, pP);

    const float inv = 1.0f / lsum[0];
    if (!qvalid) return;
    const size_t grow = (size_t)b * p.Sq + qrow;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = h * DH + 32 * db + 8 * g + 4 * hh;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ot[db][4 * g + e] * inv;
            if constexpr (RESID) {
                attn_resid_update(p, grow, col, o);   // x = attn_out + residual (src/components/DiT.py:139)
            } else {
                u32x2 st2;
                st2[0] = pack_bf16x2(o[0], o[1]);
                st2[1] = pack_bf16x2(o[2], o[3]);
                *reinterpret_cast<u32x2*>(p.out + grow * p.ldo + col) = st2;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Generic head_dim fallback pieces (e.g. the reference's shipped config: ONE head, dh = 768).
// scores fp32 [Sq, ldS] -> P bf16 [Sq, ldS] = softmax(scores * scale) over the first Skv columns,
// zeros in the padding columns.  One wave per row.
// ------------------------------------------------------------------------------------------------
// causal_rows > 0: rows are queries of sequences of causal_rows positions and query i sees keys j <= i + causal_off
// (the boolean tgt_mask of src/model/SpeechLP.py:57-61: masked scores are -inf, their probabilities exactly 0).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, bf16* __restrict__ P, int Sq,
                                                           int Skv, int ld, float scale_log2, int causal_rows,
                                                           int causal_off) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Sq) return;
    int vis = Skv;   // number of visible keys
    if (causal_rows > 0) {
        vis = row % causal_rows + causal_off + 1;
        vis = vis < 0 ? 0 : (vis > Skv ? Skv : vis);
    }
    const float* s = S + (size_t)row * ld;
    float m = -1e30f;
    for (int j = lane; j < vis; j += 64) m = fmaxf(m, s[j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < vis; j += 64) sum += __builtin_amdgcn_exp2f((s[j] - m) * scale_log2);
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    bf16* pr = P + (size_t)row * ld;
    for (int j = lane; j < ld; j += 64)
        pr[j] = j < vis ? (bf16)(__builtin_amdgcn_exp2f((s[j] - m) * scale_log2) * inv) : (bf16)0.f;
}
// Vt[dh, ldT] = V[Skv, dh]^T (zero padded to ldT columns); 32x32 LDS tile transpose.
// blockIdx.z = (batch, head) inside the chunk: V += zb * vs_b + zh * vs_h, Vt += z * dh * ldT
__global__ __launch_bounds__(256) void transpose_pad_kernel(const bf16* __restrict__ V, int ldv, bf16* __restrict__ Vt,
                                                            int ldT, int Skv, int dh, int H, int bh0, long long vs_b,
                                                            long long vs_h) {
    __shared__ bf16 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int k0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    {
        const int bh = bh0 + blockIdx.z;
        V += (long long)(bh / H) * vs_b + (long long)(bh % H) * vs_h;
        Vt += (size_t)blockIdx.z * dh * ldT;
    }
    for (int i = ty; i < 32; i += 8) {
        const int key = k0 + i, dd = d0 + tx;
        tile[i][tx] = (key < Skv && dd < dh) ? V[(size_t)key * ldv + dd] : (bf16)0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int dd = d0 + i, key = k0 + tx;
        if (dd < dh && key < ldT) Vt[(size_t)dd * ldT + key] = tile[tx][i];
    }
}

// in-place half-split RoPE on bf16 rows (generic head_dim): columns [0, ncols) are heads of width dh.
__global__ __launch_bounds__(256) void rope_inplace_kernel(bf16* __restrict__ x, int ld, const float* __restrict__ cs,
                                                           const float* __restrict__ sn, int M, int rpb, int nheads,
                                                           int dh, float sin_sign, int hstride) {
    const int half = dh >> 1;
    const int ppr = nheads * half;                          // rotation pairs per row
    const size_t npairs = (size_t)M * ppr;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npairs; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / ppr), pc = (int)(i % ppr);
        const int head = pc / half, j = pc % half;
        const int pos = row % rpb;
        bf16* pl = x + (size_t)row * ld + head * hstride + j;
        const float lo = (float)pl[0], hi = (float)pl[half];
        const float cc = cs[(size_t)pos * half + j], ss = sin_sign * sn[(size_t)pos * half + j];
        pl[0] = (bf16)(lo * cc - hi * ss);
        pl[half] = (bf16)(hi * cc + lo * ss);
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

static size_t generic_fwd_bytes(int Sq, int Skv, int dh) {   // one (batch, head)
    const size_t ld = (size_t)((Skv + 63) / 64) * 64;
    return align256((size_t)Sq * ld * 4) + align256((size_t)Sq * ld * 2) + align256((size_t)dh * ld * 2);
}
// the generic path multiplies a CHUNK of (batch, head) pairs per launch (batched GEMMs): as many as fit 2 GiB
static int generic_chunk(int B, int H, int Sq, int Skv, int dh) {
    const size_t one = generic_fwd_bytes(Sq, Skv, dh);
    size_t n = ((size_t)2 << 30) / one;
    if (n < 1) n = 1;
    if (n > (size_t)B * H) n = (size_t)B * H;
    return (int)n;
}
size_t attention_generic_workspace_bytes(int B, int H, int Sq, int Skv, int dh) {
    return (size_t)generic_chunk(B, H, Sq, Skv, dh) * generic_fwd_bytes(Sq, Skv, dh);
}
size_t attention_workspace_bytes(int B, int H, int Sq, int Skv, int dh) {
    if (dh == DH) return 0;   // the fused head_dim-64 kernels need no scratch
    return attention_generic_workspace_bytes(B, H, Sq, Skv, dh);
}

hipError_t launch_rope_inplace(void* x, int ld, const float* cs, const float* sn, int M, int rpb, int ncols, int dh,
                               hipStream_t s, float sin_sign, int hstride) {
    if (hstride <= 0) hstride = dh;
    const int nheads = ncols / hstride;
    const size_t npairs = (size_t)M * nheads * (dh / 2);
    size_t g = (npairs + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(rope_inplace_kernel, dim3((unsigned)(g ? g : 1)), dim3(256), 0, s, (bf16*)x, ld, cs, sn, M, rpb,
                       nheads, dh, sin_sign, hstride);
    return hipGetLastError();
}

#ifdef DITTO_DIAG_A2_STAMP
}  // namespace ditto
extern "C" int ditto_diag_a2_stamps(unsigned long long* out) {   // out[8]: sums over the waves of the LAST launches (diagnostic build only)
    static unsigned long long host[ditto::A2_STAMP_WAVES * 8];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ditto::g_a2_stamps), sizeof(host)) != hipSuccess) return 1;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (int w = 0; w < ditto::A2_STAMP_WAVES; ++w)
        for (int i = 0; i < 8; ++i) out[i] += host[(size_t)w * 8 + i];
    return 0;
}
namespace ditto {
#endif

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.H <= 0 || a.Sq <= 0 || a.Skv <= 0) return hipErrorInvalidValue;
    const float LOG2E = 1.4426950408889634f;
    if (a.causal && a.dropout_p > 0.f) return hipErrorInvalidValue;   // no caller: the decoder stack runs in eval mode
    // the bf16 stream exists on the fused head_dim-64 kernels only; attn_flags 8192 (the older TRAINING-forward kernel, an A/B bit)
    // concerns the training forward alone: it must not make the default inference forward fail
    const bool train_fwd = a.lse_out != nullptr || a.dropout_p > 0.f;
    if (a.resid_bf16 && (a.dh != DH || a.force_generic || a.causal || (train_fwd && (g_attn_flags & 8192))))
        return hipErrorInvalidValue;
    if (a.dh == DH && !a.force_generic && !a.causal) {
        if ((a.ldq | a.ldk | a.ldv) % 8) return hipErrorInvalidValue;
        AttnParams p;
        p.q = (const bf16*)a.q; p.ldq = a.ldq; p.k = (const bf16*)a.k; p.ldk = a.ldk;
        p.v = (const bf16*)a.v; p.ldv = a.ldv; p.out = (bf16*)a.out_bf16; p.ldo = a.ldo;
        p.resid = a.resid_f32; p.ldr = a.ldr; p.resid_in = a.resid_in ? a.resid_in : a.resid_f32; p.resid_bf16 = a.resid_bf16 ? 1 : 0; p.B = a.B; p.H = a.H; p.Sq = a.Sq; p.Skv = a.Skv;
        p.nqb = (a.Sq + QBLK - 1) / QBLK;
        p.scale_log2 = a.scale * LOG2E;
        p.lse = a.lse_out; p.drop_thr = dropout_threshold(a.dropout_p);
        p.keep_scale = p.drop_thr ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
        p.seed_lo = (unsigned)(a.seed & 0xFFFFFFFFu); p.seed_hi = (unsigned)(a.seed >> 32); p.layer = a.layer;
        if (a.lse_out || p.drop_thr) {   // training forward: log-sum-exp kept for the backward, dropout
            const dim3 gridt(p.nqb * a.H * a.B);
            if (g_attn_flags & 8192) {   // A/B: the older kernel (per-element scale and row sum in the vector pipe)
                if (a.resid_f32) hipLaunchKernelGGL((attn64_kernel<true, true, false, true, 3>), gridt, dim3(256), 0, s, p);
                else hipLaunchKernelGGL((attn64_kernel<false, true, false, true, 3>), gridt, dim3(256), 0, s, p);
                return hipGetLastError();
            }
            return launch_attention_train64(p, a.resid_f32 != nullptr, s);   // attention_train.hip: attn64v2 <TRAIN [, DROP]>
        }
        if (a.q_prescaled && !(g_attn_flags & 32)) {   // q already carries scale * log2(e): the reduced-VALU kernel
            // attn64p (attention_p.hip, round 6: 64 queries per wave, 256 per workgroup, two workgroups per CU) from three workgroups per
            // CU on — decided on the CLASS rows, like every rule that changes an utterance's bits (its row sum is the fp32 sum of the
            // probabilities, attn64v2's that of their bf16 roundings): C2 B = 32 112 against 127 us in isolation (-12 %), Sq = Skv =
            // 4096 397 against 463 (profiles/r06_attn_probe.txt).  attn_flags 131072 = never, 262144 = wherever the strides allow.
            {
                const long rows_cls = opt_class_rows() > 0 ? opt_class_rows() : (long)a.B * a.Sq;
                const long wgs = (rows_cls / 256) * a.H;
                const bool wide_ok = a.ldo % 8 == 0 && (!a.resid_f32 || a.ldr % 8 == 0);   // 16-byte row pieces in the epilogue
                if (wide_ok && !(g_attn_flags & 131072) && (wgs >= (a.resid_f32 ? g_attn64p_min_wgs : g_attn64p_min_wgs_plain) || (g_attn_flags & 262144)))
                    return launch_attn64p(p, a.resid_f32 != nullptr, s, (g_attn_flags & 524288) != 0,
                                          (g_attn_flags & 1048576) ? 1 : (g_attn_flags & 2097152) ? 2 : 0);
            }
            const dim3 gridv(p.nqb * a.H * a.B);
            // whole pairs of key tiles: the software-pipelined, hand-interleaved kernel where it measures faster — small grids
            // (batch-1 serving, 96 workgroups: 13.6 us against 15.6 for attn64v2's deep-prefetch instantiation) and long key
            // sequences (Sq = Skv = 4096, B = 8: 445 against 457 us).  At C2 (N = T = 1024, B = 32) it ties in isolation
            // (142.1 / 141.7 us) and loses in the model (148.9 / 141.3 us self, 133.9 / 132.1 cross): attn64v2 stays there.
            // attn_flags 256 = never, 512 = wherever the shape allows.
            const bool v3_shape = a.Skv % (2 * KBLK) == 0;
            const bool v3_pays = (int)gridv.x <= 320 || a.Skv >= 2048;
            if (v3_shape && !(g_attn_flags & 256) && (v3_pays || (g_attn_flags & 512))) {
                static DevOnce lds_once;
                if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&attn64v3_kernel<true>),
                                                               reinterpret_cast<const void*>(&attn64v3_kernel<false>)}, V3_LDS))
                    return e;
                // 256 queries per workgroup (8 waves share the K/V tiles) where the key sequence is long: Sq = Skv = 4096, B = 8:
                // 427.6 against 441.2 us; at 1024 keys it ties (140.6 / 141.3).  attn_flags 1024 forces it, 2048 forbids it.
                if (((g_attn_flags & 1024) || (a.Skv >= 2048 && a.Sq >= 256)) && !(g_attn_flags & 2048)) {
                    static DevOnce lds_once8;
                    if (hipError_t e = set_max_lds_once(lds_once8, {reinterpret_cast<const void*>(&attn64v3_kernel<true, 8>),
                                                                    reinterpret_cast<const void*>(&attn64v3_kernel<false, 8>)}, V3_LDS))
                        return e;
                    p.nqb = (a.Sq + 255) / 256;
                    const dim3 grid8(p.nqb * a.H * a.B);
                    if (a.resid_f32) hipLaunchKernelGGL((attn64v3_kernel<true, 8>), grid8, dim3(512), V3_LDS, s, p);
                    else hipLaunchKernelGGL((attn64v3_kernel<false, 8>), grid8, dim3(512), V3_LDS, s, p);
                    return hipGetLastError();
                }
                if (a.resid_f32) hipLaunchKernelGGL((attn64v3_kernel<true>), gridv, dim3(256), V3_LDS, s, p);
                else hipLaunchKernelGGL((attn64v3_kernel<false>), gridv, dim3(256), V3_LDS, s, p);
                return hipGetLastError();
            }
            // 3 waves per SIMD (<= 168 registers, no V-fragment prefetch).  Measured in-model (C2, B=32,
            // tools/step_ab.py): 138 / 127 us (self / cross) against 159 / 141 us at 2 waves per SIMD with the
            // prefetch, and 150 / 140 us for attn64 — the kernel is latency-bound (SQ counters: VALU issue 53 %,
            // MFMA 28 %, both idle 34 % of the time at 2 waves), so occupancy pays more than the prefetch.
            if (!(g_attn_flags & 64)) {
                // small grid (at most ~1 workgroup per CU): the deep-prefetch instantiation (attn_flags 128 disables)
                if ((int)gridv.x <= 320 && !(g_attn_flags & 128)) {
                    if (a.resid_f32) hipLaunchKernelGGL((attn64v2_kernel<true, 2, 4>), gridv, dim3(256), 0, s, p);
                    else hipLaunchKernelGGL((attn64v2_kernel<false, 2, 4>), gridv, dim3(256), 0, s, p);
                    return hipGetLastError();
                }
                if (a.resid_f32) hipLaunchKernelGGL((attn64v2_kernel<true, 3>), gridv, dim3(256), 0, s, p);
                else hipLaunchKernelGGL((attn64v2_kernel<false, 3>), gridv, dim3(256), 0, s, p);
                return hipGetLastError();
            }
            if (a.resid_f32) hipLaunchKernelGGL((attn64v2_kernel<true>), gridv, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((attn64v2_kernel<false>), gridv, dim3(256), 0, s, p);
            return hipGetLastError();
        }
        if (a.q_prescaled) p.scale_log2 = 1.0f;   // attn_flags & 32: the older kernels on pre-scaled q (A/B)
        const dim3 grid(p.nqb * a.H * a.B), block(256);
        const bool r = a.resid_f32 != nullptr;
        switch (g_attn_flags & 3) {
            case 0:
                if (r) hipLaunchKernelGGL((attn64_kernel<true, false, false>), grid, block, 0, s, p);
                else hipLaunchKernelGGL((attn64_kernel<false, false, false>), grid, block, 0, s, p);
                break;
            case 1:
                if (r) hipLaunchKernelGGL((attn64_kernel<true, true, false>), grid, block, 0, s, p);
                else hipLaunchKernelGGL((attn64_kernel<false, true, false>), grid, block, 0, s, p);
                break;
            default:   // 3 (2 alone is not built: the prefetch needs the registers the DMA staging frees)
                if (r) hipLaunchKernelGGL((attn64_kernel<true, true, true>), grid, block, 0, s, p);
                else hipLaunchKernelGGL((attn64_kernel<false, true, true>), grid, block, 0, s, p);
                break;
        }
        return hipGetLastError();
    }
    // ---- generic head_dim: S = Q K^T (GEMM) -> row softmax -> O = P V (GEMM on V^T), a chunk of (batch, head) pairs per
    //      launch (batched GEMMs; the dropout kernels derive the pair's hash stream from the row index) ----
    if (a.dh % 64) return hipErrorInvalidValue;
    const int ld = ((a.Skv + 63) / 64) * 64;
    const int BH = a.B * a.H;
    const size_t one = generic_fwd_bytes(a.Sq, a.Skv, a.dh);
    if (!a.workspace || a.workspace_bytes < one) return hipErrorInvalidValue;
    int chunk = (int)(a.workspace_bytes / one);
    if (chunk > BH) chunk = BH;
    // only batch chunks that are whole runs of heads / batches: either all H heads of some batches, or heads of one batch
    if (chunk >= a.H) chunk = (chunk / a.H) * a.H;
    else while (a.H % chunk) --chunk;
    char* ws = (char*)a.workspace;
    const size_t sS = (size_t)a.Sq * ld, sV = (size_t)a.dh * ld;
    float* S = (float*)ws;
    bf16* P = (bf16*)(ws + align256((size_t)chunk * sS * 4));
    bf16* Vt = (bf16*)((char*)P + align256((size_t)chunk * sS * 2));
    for (int bh0 = 0; bh0 < BH; bh0 += chunk) {
        const int nb = chunk < BH - bh0 ? chunk : BH - bh0;
        const int b0 = bh0 / a.H, h0 = bh0 % a.H;
        const int nz_o = nb >= a.H ? nb / a.H : 1, nz_i = nb >= a.H ? a.H : nb;
        const bf16* q = (const bf16*)a.q + (size_t)b0 * a.Sq * a.ldq + h0 * a.dh;
        const bf16* k = (const bf16*)a.k + (size_t)b0 * a.Skv * a.ldk + h0 * a.dh;
        const bf16* v = (const bf16*)a.v + (size_t)b0 * a.Skv * a.ldv + h0 * a.dh;
        GemmArgs g1{};
        g1.A = q; g1.lda = a.ldq; g1.W = k; g1.ldw = a.ldk; g1.out = S; g1.ldo = ld;
        g1.M = a.Sq; g1.N = ld; g1.K = a.dh; g1.w_rows = a.Skv;
        g1.batch_outer = nz_o; g1.batch_inner = nz_i;
        g1.sA[0] = (long long)a.Sq * a.ldq; g1.sA[1] = a.dh; g1.sW[0] = (long long)a.Skv * a.ldk; g1.sW[1] = a.dh;
        g1.sO[0] = (long long)nz_i * sS; g1.sO[1] = (long long)sS;
        hipError_t e = launch_gemm(g1, EPI_BIAS_F32, s);
        if (e != hipSuccess) return e;
        if (a.dropout_p > 0.f) {
            e = launch_softmax_drop_rows(S, P, nb * a.Sq, a.Sq, a.Skv, ld, a.scale, a.seed, a.layer, bh0, a.dropout_p, s);
            if (e != hipSuccess) return e;
        } else {
            hipLaunchKernelGGL(softmax_rows_kernel, dim3((nb * a.Sq + 3) / 4), dim3(256), 0, s, S, P, nb * a.Sq, a.Skv, ld,
                               a.scale * LOG2E, a.causal ? a.Sq : 0, a.Skv - a.Sq);
        }
        hipLaunchKernelGGL(transpose_pad_kernel, dim3(ld / 32, (a.dh + 31) / 32, nb), dim3(256), 0, s, (const bf16*)a.v,
                           a.ldv, Vt, ld, a.Skv, a.dh, a.H, bh0, (long long)a.Skv * a.ldv, (long long)a.dh);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        (void)v;
        GemmArgs g2{};
        g2.A = P; g2.lda = ld; g2.W = Vt; g2.ldw = ld; g2.M = a.Sq; g2.N = a.dh; g2.K = ld; g2.w_rows = a.dh;
        g2.batch_outer = nz_o; g2.batch_inner = nz_i;
        g2.sA[0] = (long long)nz_i * sS; g2.sA[1] = (long long)sS; g2.sW[0] = (long long)nz_i * sV; g2.sW[1] = (long long)sV;
        if (a.resid_f32) {
            float* r = a.resid_f32 + (size_t)b0 * a.Sq * a.ldr + h0 * a.dh;
            g2.residual = (a.resid_in ? a.resid_in : a.resid_f32) + (size_t)b0 * a.Sq * a.ldr + h0 * a.dh;
            g2.ldr = a.ldr; g2.out = r; g2.ldo = a.ldr;
            g2.sO[0] = (long long)a.Sq * a.ldr; g2.sO[1] = a.dh; g2.sR[0] = g2.sO[0]; g2.sR[1] = g2.sO[1];
            e = launch_gemm(g2, EPI_BIAS_RES_F32, s);
        } else {
            g2.out = (bf16*)a.out_bf16 + (size_t)b0 * a.Sq * a.ldo + h0 * a.dh; g2.ldo = a.ldo;
            g2.sO[0] = (long long)a.Sq * a.ldo; g2.sO[1] = a.dh;
            e = launch_gemm(g2, EPI_BIAS_BF16, s);
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// Attention backward, GEMM-composed (any head_dim % 64 == 0), per (batch, head), probabilities recomputed:
//   S = Q K^T,  P = dropout(softmax(S * scale)),  dV = P^T dO,  dPd = dO V^T,
//   dS = P_nodrop * (mask * dPd / (1-p) - rowsum) * scale,  dQ = dS K,  dK = dS^T Q.
// The "^T" operands of the forward GEMM family (K-contiguous rows) are made by transpose_bf16.
// ------------------------------------------------------------------------------------------------
namespace {
struct BwdWs { size_t S, dP, P, dS, Pt, dOt, Kt, Qt, total; };   // offsets of the per-chunk arrays
BwdWs plan_bwd(int Sq, int Skv, int dh, int chunk) {
    BwdWs w;
    const size_t ld = (size_t)((Skv + 63) / 64) * 64, sqp = (size_t)((Sq + 63) / 64) * 64, n = (size_t)chunk;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += align256(b); return o; };
    w.S = take(n * Sq * ld * 4); w.dP = take(n * Sq * ld * 4); w.P = take(n * Sq * ld * 2);
    w.dS = take(n * Sq * ld * 2); w.Pt = take(n * ld * sqp * 2); w.dOt = take(n * dh * sqp * 2);
    w.Kt = take(n * dh * ld * 2); w.Qt = take(n * dh * sqp * 2);
    w.total = off;
    return w;
}
int bwd_chunk(int B, int H, int Sq, int Skv, int dh, size_t budget) {
    const size_t one = plan_bwd(Sq, Skv, dh, 1).total + 8 * 256;
    size_t n = budget / one;
    if (n < 1) n = 1;
    if (n > (size_t)B * H) n = (size_t)B * H;
    int chunk = (int)n;
    if (chunk >= H) chunk = (chunk / H) * H;   // whole batches, or a divisor of H inside one batch
    else while (H % chunk) --chunk;
    return chunk;
}
}  // namespace

size_t attention_train_workspace_bytes(int B, int H, int Sq, int Skv, int dh) {
    // generic forward / backward: a chunk of (batch, head) pairs per launch, as many as fit 2 GiB
    const int cf = generic_chunk(B, H, Sq, Skv, dh);
    const size_t f = (size_t)cf * generic_fwd_bytes(Sq, Skv, dh);
    const int cb = bwd_chunk(B, H, Sq, Skv, dh, (size_t)2 << 30);
    const size_t b = plan_bwd(Sq, Skv, dh, cb).total;
    const size_t dl = align256(attention_bwd_stats_bytes(B, H, Sq));   // fused backward: the per-tile {L, delta} records
    const size_t g = f > b ? f : b;
    return g > dl ? g : dl;
}

bool attention_bwd_fuses_rope(const AttnBwdArgs& a) { return a.dh == DH && a.lse && a.rope_cos && a.rope_sin && a.Sq == a.Skv; }

hipError_t launch_attention_bwd(const AttnBwdArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.H <= 0 || a.Sq <= 0 || a.Skv <= 0 || a.dh % 64) return hipErrorInvalidValue;
    if ((a.rope_cos || a.rope_sin) && !attention_bwd_fuses_rope(a)) return hipErrorInvalidValue;   // the caller asks first
    if (a.dh == DH && a.lse) {   // fused: the two kernels; the {L, delta = rowsum(dO * O)} records live in the workspace
        if (!a.workspace || a.workspace_bytes < attention_bwd_stats_bytes(a.B, a.H, a.Sq)) return hipErrorInvalidValue;
        return launch_attention_bwd64(a, (float*)a.workspace, s);
    }
    if (!a.workspace || a.workspace_bytes < plan_bwd(a.Sq, a.Skv, a.dh, 1).total) return hipErrorInvalidValue;
    const int BH = a.B * a.H;
    const int chunk = bwd_chunk(a.B, a.H, a.Sq, a.Skv, a.dh, a.workspace_bytes);
    const BwdWs w = plan_bwd(a.Sq, a.Skv, a.dh, chunk);
    const int ld = ((a.Skv + 63) / 64) * 64, sqp = ((a.Sq + 63) / 64) * 64;
    char* ws = (char*)a.workspace;
    float* S = (float*)(ws + w.S);
    float* dP = (float*)(ws + w.dP);
    bf16* P = (bf16*)(ws + w.P);
    bf16* dS = (bf16*)(ws + w.dS);
    bf16* Pt = (bf16*)(ws + w.Pt);
    bf16* dOt = (bf16*)(ws + w.dOt);
    bf16* Kt = (bf16*)(ws + w.Kt);
    bf16* Qt = (bf16*)(ws + w.Qt);
    const long long sS = (long long)a.Sq * ld, sPt = (long long)ld * sqp, sDt = (long long)a.dh * sqp, sKt = (long long)a.dh * ld;
    hipError_t e;
#define ATRY(x) do { if ((e = (x)) != hipSuccess) return e; } while (0)
    for (int bh0 = 0; bh0 < BH; bh0 += chunk) {
        const int nb = chunk < BH - bh0 ? chunk : BH - bh0;
        const int b0 = bh0 / a.H, h0 = bh0 % a.H;
        const int zo = nb >= a.H ? nb / a.H : 1, zi = nb >= a.H ? a.H : nb;
        const bf16* q = (const bf16*)a.q + (size_t)b0 * a.Sq * a.ldq + h0 * a.dh;
        const bf16* k = (const bf16*)a.k + (size_t)b0 * a.Skv * a.ldk + h0 * a.dh;
        const bf16* v = (const bf16*)a.v + (size_t)b0 * a.Skv * a.ldv + h0 * a.dh;
        const bf16* dO = (const bf16*)a.dout + (size_t)b0 * a.Sq * a.lddo + h0 * a.dh;
        bf16* dq = (bf16*)a.dq + (size_t)b0 * a.Sq * a.lddq + h0 * a.dh;
        bf16* dk = (bf16*)a.dk + (size_t)b0 * a.Skv * a.lddk + h0 * a.dh;
        bf16* dv = (bf16*)a.dv + (size_t)b0 * a.Skv * a.lddv + h0 * a.dh;
        auto batched = [&](GemmArgs& g) { g.batch_outer = zo; g.batch_inner = zi; };
        auto stacked = [&](long long (&st)[2], long long per) { st[0] = (long long)zi * per; st[1] = per; };   // [z] arrays
        auto strided = [&](long long (&st)[2], long long rows, int ldx) { st[0] = rows * ldx; st[1] = a.dh; };  // (b, h) views
        GemmArgs g{};
        // S = Q K^T
        g.A = q; g.lda = a.ldq; g.W = k; g.ldw = a.ldk; g.w_rows = a.Skv; g.out = S; g.ldo = ld;
        g.M = a.Sq; g.N = ld; g.K = a.dh;
        batched(g); strided(g.sA, a.Sq, a.ldq); strided(g.sW, a.Skv, a.ldk); stacked(g.sO, sS);
        ATRY(launch_gemm(g, EPI_BIAS_F32, s));
        ATRY(launch_softmax_drop_rows(S, P, nb * a.Sq, a.Sq, a.Skv, ld, a.scale, a.seed, a.layer, bh0, a.dropout_p, s));
        // dV = P^T dO
        ATRY(launch_transpose_bf16(P, ld, a.Sq, ld, Pt, sqp, s, nb, 1, sS, 0, sPt));
        ATRY(launch_transpose_bf16(dO, a.lddo, a.Sq, a.dh, dOt, sqp, s, zo, zi, (long long)a.Sq * a.lddo, a.dh, sDt));
        g = GemmArgs{};
        g.A = Pt; g.lda = sqp; g.W = dOt; g.ldw = sqp; g.w_rows = a.dh; g.out = dv; g.ldo = a.lddv;
        g.M = a.Skv; g.N = a.dh; g.K = sqp;
        batched(g); stacked(g.sA, sPt); stacked(g.sW, sDt); strided(g.sO, a.Skv, a.lddv);
        ATRY(launch_gemm(g, EPI_BIAS_BF16, s));
        // dPd = dO V^T
        g = GemmArgs{};
        g.A = dO; g.lda = a.lddo; g.W = v; g.ldw = a.ldv; g.w_rows = a.Skv; g.out = dP; g.ldo = ld;
        g.M = a.Sq; g.N = ld; g.K = a.dh;
        batched(g); strided(g.sA, a.Sq, a.lddo); strided(g.sW, a.Skv, a.ldv); stacked(g.sO, sS);
        ATRY(launch_gemm(g, EPI_BIAS_F32, s));
        ATRY(launch_softmax_bwd_rows(S, dP, dS, nb * a.Sq, a.Sq, a.Skv, ld, a.scale, a.seed, a.layer, bh0, a.dropout_p, s));
        // dQ = dS K
        ATRY(launch_transpose_bf16(k, a.ldk, a.Skv, a.dh, Kt, ld, s, zo, zi, (long long)a.Skv * a.ldk, a.dh, sKt));
        g = GemmArgs{};
        g.A = dS; g.lda = ld; g.W = Kt; g.ldw = ld; g.w_rows = a.dh; g.out = dq; g.ldo = a.lddq;
        g.M = a.Sq; g.N = a.dh; g.K = ld;
        batched(g); stacked(g.sA, sS); stacked(g.sW, sKt); strided(g.sO, a.Sq, a.lddq);
        ATRY(launch_gemm(g, EPI_BIAS_BF16, s));
        // dK = dS^T Q
        ATRY(launch_transpose_bf16(dS, ld, a.Sq, ld, Pt, sqp, s, nb, 1, sS, 0, sPt));
        ATRY(launch_transpose_bf16(q, a.ldq, a.Sq, a.dh, Qt, sqp, s, zo, zi, (long long)a.Sq * a.ldq, a.dh, sDt));
        g = GemmArgs{};
        g.A = Pt; g.lda = sqp; g.W = Qt; g.ldw = sqp; g.w_rows = a.dh; g.out = dk; g.ldo = a.lddk;
        g.M = a.Skv; g.N = a.dh; g.K = sqp;
        batched(g); stacked(g.sA, sPt); stacked(g.sW, sDt); strided(g.sO, a.Skv, a.lddk);
        ATRY(launch_gemm(g, EPI_BIAS_BF16, s));
    }
#undef ATRY
    return hipSuccess;
}

}  // namespace ditto
