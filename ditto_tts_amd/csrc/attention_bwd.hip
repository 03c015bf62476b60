// attention_bwd.hip — fused (flash-style) attention BACKWARD for gfx950, head_dim 64, bf16 in / fp32 accumulate.
//
// Backward of  O = dropout(softmax(q k^T * scale)) v  (self-attention, reference src/components/DiT.py:131-134, and
// nn.MultiheadAttention's cross-attention, :144-148, dropout 0.1 in train mode) with the probabilities RECOMPUTED
// from q, k and the forward's log-sum-exp, never stored:
//     P = exp2(c s - L),   dP = mask/(1-p) * (dO v^T),   dS = P * (dP - delta) * scale,   delta = rowsum(dO * O)
//     dV = (mask/(1-p) * P)^T dO,    dQ = dS k,    dK = dS^T q.
// Two kernels, so that every output is owned by exactly one workgroup: NO atomics and no cross-workgroup sum =>
// bit-reproducible gradients.
//   dq kernel    one workgroup = 128 queries; streams 64-key tiles.   S^T = K Q^T and dP^T = V dO^T have the QUERY
//                on the lane (L and delta are per-lane scalars; the query fragments carry scale * log2(e) and the score
//                chain starts from -L, as in the training forward); dS^T, packed to bf16 in registers, is the B operand
//                of dQ^T += K^T dS^T (K^T by ds_read_b64_tr_b16 from the same LDS image the row reads use).
//   dkdv kernel  one workgroup = 128 keys; streams 64-query tiles.   S = Q K^T and dP = dO V^T have the KEY on the
//                lane (K, V fragments live in registers for the whole kernel); P and dS in registers are the B
//                operands of dV^T += dO^T P and dK^T += Q^T dS (Q^T, dO^T by transposed reads).
// This costs 7 products instead of the 5 of a single-kernel backward (S and dP are computed in both), 14 B H Sq Skv
// dh FLOPs per call: 40 % more MFMA work bought for determinism and for not needing fp32 dQ atomics.
// Roofline: nominally MFMA (K/V or Q/dO of one head are re-read from the XCD's L2), in practice vector ISSUE: a 64-row
// tile carries as many cycles of P / dS arithmetic as of MFMA (see the comment above the kernel and profiles/r03_attn_bwd.txt).
#include <type_traits>

#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int DH = 64, BLK = 128, TILE = 64;
// Timing-only knock-outs (WRONG results by design; tools/bwd_knockout.sh builds one library per bit and times them):
//   1 = no P / dS vector work, 2 = tiles are DMA'd once (no global->LDS traffic in the loop), 4 = no barrier in the loop,
//   8 = no accumulation MFMAs (second phase), 16 = no S / dP MFMAs (first phase), 32 = no LDS fragment reads in the loop
#ifndef DITTO_DIAG_BWD
#define DITTO_DIAG_BWD 0
#endif
constexpr int IMG = TILE * DH * 2;   // one 64-row x 128-B tile image: 8 KiB
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

struct BwdParams {
    const bf16* q; int ldq; const bf16* k; int ldk; const bf16* v; int ldv;
    const bf16* dout; int lddo;
    bf16* dq; int lddq; bf16* dk; int lddk; bf16* dv; int lddv;
    float* stats;         // [B, H, ceil(Sq / 64), 128]: {L[64] (log2 domain) | delta[64]} per query tile: WRITTEN by the dq kernel
                          // (it holds dO fragments of its queries anyway), read by the dk,dv kernel that follows it on the stream
    const float* lse;     // [B, H, Sq] log2 domain, from the training forward
    const bf16* o; int ldo;                                  // O (cross-attention), or
    const float* h_after; const float* h_before; int ldh;   // O = h_after - h_before (self-attention: never stored)
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

// Per-row statistics of the backward, one record per (batch, head, 64-query tile): stats[((b H + h) nt + tile) 128 + {i, 64 + i}]
// = {L, delta} of query tile * 64 + i, nt = ceil(Sq / 64):  L = the forward's log2-domain log-sum-exp, delta = sum_c dO[row, h*64 + c]
// * O[row, h*64 + c].  Rows past Sq hold L = 1e30 (P = exp2(-inf) = 0) and delta = 0, so the dk,dv kernel needs no row mask,
// and a tile's record is one 512-byte LDS-DMA.  The dq kernel writes them in its prologue: a lane already holds 32 of its
// query's 64 dO columns as MFMA fragments, loads the same columns of O, and one exchange with lane ^ 32 completes the dot
// product.  (A separate row pre-pass did this before: 41 us per call, 2 x 12 calls per training step.)

template <int N>
DITTO_DEV void bwd_vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------------------------------------
// Shared tile machinery.  ONE 64-row x 128-B image per operand tile serves both read patterns: 16-B chunk c of row r sits at
// chunk position c ^ S(r), S(r) = (((r>>1)&1) << 2) | ((r>>2)&3) — a bit permutation of the (r>>1)&7 row swizzle (so the
// ds_read_b128 of MFMA A fragments, 32 rows x one chunk, stay conflict-free), whose bit 2 separates rows r and r+2 (so the four
// rows of a ds_read_b64_tr_b16 block, 4 rows x 64 B, land in four different 64-B bank quarters).  The swizzle is applied to the
// DMA's SOURCE address.  (An earlier version kept a row image and a transposed-read image of every operand: twice the LDS-DMA
// traffic and twice the LDS, which left room for one tile of look-ahead only; with the tile's DMA knocked out the dkdv kernel
// ran 22 % faster — the loads were landing late.)
// MODE 0 = dq kernel   (block = queries; tiles = keys:    images K, V)
// MODE 1 = dkdv kernel (block = keys;    tiles = queries: images Q, dO  + the tile's L / delta record)
// Ring of NBUF = 4 tile buffers and the two-tile software pipeline of the loop: see the comment at the loop.
// Measured issue costs that shape it (tools/probe_mfma_valu.hip, one wave per SIMD): a v_mfma_f32_32x32x16_bf16 gap runs
// max(32, 8 + sum of the vector instructions' costs) cycles with v_fma_f32 5.2 and v_exp_f32 9 (the MFMA's own 8 issue cycles
// never hide; accumulators in AGPRs change nothing) — a tile of the dq kernel carries ~870 cycles of P / dS vector issue against
// 768 of MFMA, so the kernels are bound by vector ISSUE, and everything that is not an MFMA or P / dS arithmetic is overhead.
// ------------------------------------------------------------------------------------------------
constexpr int NBUF = 4;
constexpr int STAT_BYTES = 1024;   // L[64] | delta[64] floats, written twice over by one 64-lane 16-B DMA

DITTO_DEV int img_swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// DROP: train-mode dropout on P (the hash mask of the forward) compiled in; without it no per-element hash, no branch.
// RAG (dq kernel only): Skv is not a multiple of 64: keys past Skv are masked per element (P = 0) in every tile; shapes with whole
// tiles compile the mask out.  The dkdv kernel needs no mask: its tile rows are queries, and rows past Sq carry L = 1e30.
template <int MODE, bool DROP, bool RAG = false>
__global__ __launch_bounds__(256, 2) void attn64_bwd_kernel(BwdParams p) {
    static_assert(MODE == 0 || !RAG, "only the dq kernel masks keys");
    constexpr int BUF = 2 * IMG + (MODE == 1 ? STAT_BYTES : 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NBUF * BUF: 64 / 68 KiB (two workgroups per CU)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = p.nblk * p.H * p.B;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int blk = id % p.nblk, bh = id / p.nblk;
    const int h = bh % p.H, b = bh / p.H;
    const int ql = lane & 31, hh = lane >> 5;
    const int nqt = (p.Sq + TILE - 1) / TILE;   // stats records per (batch, head)

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
    // MODE 0: the query fragments carry scale * log2(e), rounded to bf16 exactly as the training forward rounds them
    // (attn64v2_kernel<.., TRAIN>), so K Q'^T is already in log2 units, and the chain's first MFMA starts from -L: the scores
    // come out of the matrix pipe as S' - L, no per-element scale and subtract
    if constexpr (MODE == 0) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i) f0[ks][i] = (bf16)((float)f0[ks][i] * p.scale_log2);
    }
    float own_L = 1e30f, own_delta = 0.f;   // MODE 0: per-lane (query) scalars
    if constexpr (MODE == 0) {
        // delta = rowsum(dO * O) of the lane's query: its 32 columns (d = 16 ks + 8 hh + 0..7) here, the other 32 in lane ^ 32
        float part = 0.f;
        const size_t grow = (size_t)b * p.Sq + own;
        if (p.o) {
            const bf16* op = p.o + grow * p.ldo + h * DH + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 ov = *reinterpret_cast<const bf16x8*>(op + 16 * ks);
#pragma unroll
                for (int i = 0; i < 8; ++i) part += (float)f1[ks][i] * (float)ov[i];
            }
        } else {
            const float* ap = p.h_after + grow * p.ldh + h * DH + 8 * hh;
            const float* bp = p.h_before + grow * p.ldh + h * DH + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(ap + 16 * ks + 4 * q4);
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + 16 * ks + 4 * q4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) part += (float)f1[ks][4 * q4 + i] * (av[i] - bv[i]);
                }
        }
        part += __shfl_xor(part, 32, 64);
        if (own_valid) {
            own_L = p.lse[(size_t)bh * p.Sq + own];
            own_delta = part;
        }
        const int row_u = blk * BLK + wid * 32 + ql;   // unclamped: rows in [Sq, nqt * 64) get the fill values
        if (hh == 0 && row_u < nqt * TILE) {
            float* rec = p.stats + ((size_t)bh * nqt + row_u / TILE) * 128 + (row_u & (TILE - 1));
            rec[0] = own_L;
            rec[64] = own_delta;
        }
    }

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    // tile sources: MODE 0: (K, V), MODE 1: (Q, dO).  This lane's two (row, chunk) DMA sources of tile 0; tile tt is + tt * 64 rows
    const bf16* t0 = MODE == 0 ? p.k : p.q;
    const int ld0 = MODE == 0 ? p.ldk : p.ldq;
    const bf16* t1 = MODE == 0 ? p.v : p.dout;
    const int ld1 = MODE == 0 ? p.ldv : p.lddo;
    const int ntile = (tile_len + TILE - 1) / TILE;
    const bool ragged_tile = (tile_len & (TILE - 1)) != 0;
    // this lane's two (row, chunk) DMA sources of a tile: wave-uniform 64-bit tile base (scalar registers) + 32-bit byte offset
    unsigned voff0[2], voff1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wid * 2 + i) * 8 + (lane >> 3), c = (lane & 7) ^ img_swz(row);
        voff0[i] = (unsigned)(row * ld0 + c * 8) * 2u;
        voff1[i] = (unsigned)(row * ld1 + c * 8) * 2u;
    }
    const bf16* base0 = t0 + (size_t)b * tile_len * ld0 + h * DH;
    const bf16* base1 = t1 + (size_t)b * tile_len * ld1 + h * DH;
    const size_t step0 = (size_t)TILE * ld0, step1 = (size_t)TILE * ld1;
    const float* stat_base = p.stats + (size_t)bh * nqt * 128;   // wave-uniform; the lane's part is a 32-bit offset
    // 4 DMAs per wave and tile (wave 0 of the dkdv kernel: 5, the tile's L / delta record)
    auto dma_tile = [&](int tt, int buf) {
        const unsigned dst = lds_base + (unsigned)(buf * BUF);
        if (ragged_tile && tt == ntile - 1) {   // rows past the end are clamped (never read out of bounds); their P is 0
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int piece = wid * 2 + i;
                const int row = piece * 8 + (lane >> 3), c = (lane & 7) ^ img_swz(row);
                int tr = tt * TILE + row;
                tr = tr < tile_len ? tr : tile_len - 1;
                glds16(t0 + ((size_t)b * tile_len + tr) * ld0 + h * DH + c * 8, dst + piece * 1024);
                glds16(t1 + ((size_t)b * tile_len + tr) * ld1 + h * DH + c * 8, dst + IMG + piece * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int piece = wid * 2 + i;
                glds16_so(voff0[i], base0 + (size_t)tt * step0, dst + piece * 1024);
                glds16_so(voff1[i], base1 + (size_t)tt * step1, dst + IMG + piece * 1024);
            }
        }
        if constexpr (MODE == 1) {
            if (wid == 0) glds16_so((unsigned)(lane & 31) * 16u, stat_base + (size_t)tt * 128, dst + 2 * IMG);
        }
    };
    // wait until at most `younger` whole tiles requested after the one needed are still in flight (0, 1 or 2)
    auto wait_tile = [&](int younger) {
        if (MODE == 1 && wid == 0) {
            if (younger >= 2) bwd_vm_wait<10>();
            else if (younger == 1) bwd_vm_wait<5>();
            else bwd_vm_wait<0>();
        } else {
            if (younger >= 2) bwd_vm_wait<8>();
            else if (younger == 1) bwd_vm_wait<4>();
            else bwd_vm_wait<0>();
        }
    };

    const int row_off = ql * 128, row_swz = img_swz(ql);
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int tr_colbyte = (16 * ((lane >> 4) & 1) + 4 * tr_p) * 2;
    const int tr_row0 = 4 * hh + tr_q;
    // rows 16 s2 + tr_row0 and + 8: S = ((tr_q >> 1) & 1) << 2 | (hh [+ 2]) & 3
    const int tr_swz0 = (((tr_q >> 1) & 1) << 6) | (hh << 4), tr_swz1 = (((tr_q >> 1) & 1) << 6) | (((hh + 2) & 3) << 4);

    f32x16 acc0[2], acc1[2];   // MODE 0: dQ^T in acc0 (acc1 unused)   MODE 1: dK^T in acc0, dV^T in acc1
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[0][i] = 0.f; acc0[1][i] = 0.f; acc1[0][i] = 0.f; acc1[1][i] = 0.f; }
    const float c = p.scale_log2;
    const DropStream dstream = drop_stream(p.seed_lo, p.seed_hi, p.layer, bh);

    // ---- the tile loop: a software pipeline across TWO tiles, written out by hand ----
    // Per tile: A0..A3 = four steps of 4 MFMAs (S | S^T and dP | dP^T of row block 0: A0, A1; of block 1: A2, A3), V0..V3 = the
    // P / dS vector work of 16 tile rows each (V0, V1 need A1; V2, V3 need A3), B0..B3 = four accumulation slots of 2 | 4 MFMAs
    // (Bs needs Vs).  A wave's vector instructions run in the matrix pipe's shadow only when they FOLLOW an MFMA of the same wave
    // (a lone wave per SIMD showed the one-tile order A A A+V A+V B+V B+V B B as the plain sum MFMA + vector + LDS + DMA), so
    // every region below pairs one MFMA group with half a V, A steps of tile t+1 alternating with B slots of tile t:
    //     r0  A0(t+1) + V2b(t)     r1  B2(t) + V3a(t)       r2  A1(t+1) + V3b(t)     r3  B3(t) + V0a(t+1)
    //     r4  A2(t+1) + V0b(t+1)   r5  B0(t+1) + V1a(t+1)   r6  A3(t+1) + V1b(t+1)   r7  B1(t+1) + V2a(t+1)
    // with fewer registers than one tile at a time needs: block 1's S / dP of tile t die in r2, tile t+1's are born in r4, and
    // every P / dS group is consumed one region after its second half is made (two groups live, not four).
    // The next A step's row fragments are requested BEHIND the current one's MFMAs, into the same registers (consumed two
    // regions later), likewise the transposed fragments of the next B slot; a V half's L / delta go, two regions ahead, into the
    // buffer the half before last has just read.  Ring of NBUF = 4 tile buffers: tiles t, t+1 in use, t+2 landed, t+3 in flight;
    // ONE barrier per iteration (behind r6): every wave has its pieces of tile t+2 (vmcnt(0): it is the youngest request) and
    // is done with tile t-1's buffer, so the request for tile t+3 and the first fragment reads of tile t+2 follow it.
    constexpr int G = 2;                        // k-steps (of 16) per A step: 2 G fragments
    bf16x8 fa[2 * G];                           // the row fragments of one A step
    bf16x8 tf[MODE == 0 ? 2 : 4];               // the transposed fragments of one B slot: [db] of image 0, MODE 1 also [2 + db] of image 1
    f32x16 st[2], dp[2];                        // [row block]: tile row in registers, own row on the lane
    bf16x8 pf[2], dsf[2];                       // [16-row group & 1]: a group is consumed before the one after next is made
    f32x4 sl[2], sd[2];                         // MODE 1: L / delta of the 4 tile rows of a V half, [half]
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 neg_l16;   // MODE 0: every register = -L of the lane's query
#pragma unroll
    for (int i = 0; i < 16; ++i) neg_l16[i] = MODE == 0 ? -own_L : 0.f;
    auto ld_rows = [&](const char* base, int u, bf16x8* dst) {
        if constexpr ((DITTO_DIAG_BWD & 32) != 0) {   // opaque to the compiler: the MFMAs that read dst stay where they are
#pragma unroll
            for (int k = 0; k < 2 * G; ++k) asm volatile("" : "+v"(dst[k]));
            return;
        }
        const int rb = (u * G) >> 2, ks0 = (u * G) & 3;
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
            const int off = rb * 32 * 128 + row_off + (((2 * (ks0 + kk) + hh) ^ row_swz) << 4);
            dst[2 * kk] = *reinterpret_cast<const bf16x8*>(base + off);
            dst[2 * kk + 1] = *reinterpret_cast<const bf16x8*>(base + IMG + off);
        }
    };
    auto ld_tr = [&](const char* base, int s2, bf16x8* dst) {
        if constexpr ((DITTO_DIAG_BWD & 32) != 0) {
#pragma unroll
            for (int k = 0; k < (MODE == 0 ? 2 : 4); ++k) asm volatile("" : "+v"(dst[k]));
            return;
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const char* r0 = base + (16 * s2 + tr_row0) * 128;
            const char* a0 = r0 + ((tr_colbyte + 64 * db) ^ tr_swz0);
            const char* a8 = r0 + 8 * 128 + ((tr_colbyte + 64 * db) ^ tr_swz1);
            dst[db] = cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0)),
                           __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a8)));
            if constexpr (MODE == 1)
                dst[2 + db] = cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + IMG)),
                                   __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a8 + IMG)));
        }
    };
    // MODE 1: L and delta of tile rows 16 s2 + 8 half + 4 hh + {0..3} (half-wave broadcast reads)
    auto ld_stats = [&](const char* base, int s2, int half) {
        if constexpr (MODE == 1) {
            const char* sp = base + 2 * IMG + (16 * s2 + 8 * half + 4 * hh) * 4;
            sl[half] = *reinterpret_cast<const f32x4*>(sp);
            sd[half] = *reinterpret_cast<const f32x4*>(sp + 256);
        }
    };
    // A step u (tile t+1): 4 MFMAs on row unit u
    auto a_step = [&](int u) {
        const int rb = (u * G) >> 2, ks0 = (u * G) & 3;
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
            if constexpr ((DITTO_DIAG_BWD & 16) != 0) {
                asm volatile("" : "+v"(st[rb]), "+v"(dp[rb]));
                continue;
            }
            const bool first = ks0 + kk == 0;   // a chain's first MFMA takes the constant 0 as its accumulator operand
            st[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2 * kk], f0[ks0 + kk], first ? (MODE == 0 ? neg_l16 : zero16) : st[rb], 0, 0, 0);
            dp[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2 * kk + 1], f1[ks0 + kk], first ? zero16 : dp[rb], 0, 0, 0);
        }
    };
    // B slot s2 (tile t): acc^T[d][own] += T^T[d][tile row] * X[tile row][own]  (transposed reads of the same images)
    auto b_slot = [&](int s2) {
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            if constexpr ((DITTO_DIAG_BWD & 8) != 0) {   // the operands stay live (and so does the work that makes them)
                asm volatile("" : "+v"(acc0[db]) : "v"(dsf[s2 & 1]), "v"(tf[db]));
                if constexpr (MODE == 1) asm volatile("" : "+v"(acc1[db]) : "v"(pf[s2 & 1]), "v"(tf[2 + db]));
                continue;
            }
            acc0[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[db], dsf[s2 & 1], acc0[db], 0, 0, 0);   // K^T dS^T | Q^T dS
            if constexpr (MODE == 1)
                acc1[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[2 + db], pf[s2 & 1], acc1[db], 0, 0, 0);  // dO^T P
        }
    };
    // V half: P, dS of tile rows rb*32 + (r&3) + 8*(r>>2) + 4*hh, r = 8 (s2 & 1) + 4 half + {0..3}, of block rb = s2 >> 1.
    // Staged over the 4 elements (exponent arguments, exponentials, ...): independent instructions between a value's producer and
    // its consumer.  (Compiled without SLP packing: packed fp32 instructions occupy the matrix pipe, ditto_tts_amd/build.py.)
    auto v_half = [&](int tile, int s2, int half) {
        const int r0 = 8 * (s2 & 1) + 4 * half;
        if constexpr ((DITTO_DIAG_BWD & 1) != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dsf[s2 & 1][4 * half + j] = (bf16)dp[s2 >> 1][r0 + j];
                if constexpr (MODE == 1) pf[s2 & 1][4 * half + j] = (bf16)st[s2 >> 1][r0 + j];
            }
            return;
        }
        float sv[4], pr[4], gg[4], km[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sv[j] = MODE == 0 ? st[s2 >> 1][r0 + j] : st[s2 >> 1][r0 + j] * c - sl[half][j];
            if constexpr (RAG) {
                const int rr = r0 + j;
                const int trow = tile * TILE + (s2 >> 1) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hh;
                sv[j] = trow < p.Skv ? sv[j] : -1e30f;
            }
        }
        if constexpr (DROP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = r0 + j;
                const int trow = tile * TILE + (s2 >> 1) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hh;
                const int qi = MODE == 0 ? own : trow, kj = MODE == 0 ? trow : own;
                km[j] = drop_keep(dstream, qi, kj, p.drop_thr) ? p.keep_scale : 0.f;   // one select, two products
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) pr[j] = __builtin_amdgcn_exp2f(sv[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dl = MODE == 0 ? own_delta : sd[half][j];
            const float g = dp[s2 >> 1][r0 + j];
            gg[j] = DROP ? g * km[j] - dl : g - dl;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) gg[j] *= pr[j];   // dS / scale: the factor is applied once, to the accumulators (epilogue)
#pragma unroll
        for (int j = 0; j < 4; ++j) dsf[s2 & 1][4 * half + j] = (bf16)gg[j];
        if constexpr (MODE == 1) {
            if constexpr (DROP) {
#pragma unroll
                for (int j = 0; j < 4; ++j) pr[j] *= km[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) pf[s2 & 1][4 * half + j] = (bf16)pr[j];
        }
    };
    // a region = NM MFMAs + NV vector instructions, one MFMA then an equal share of the vector instructions and so on (a
    // scheduling hint), and BEHIND them the region's LDS requests (pinned: they overwrite the fragments those MFMAs read)
    constexpr int NVH = (MODE == 0 ? 18 : 20) + (DROP ? 36 : 0);   // vector instructions of one V half
    constexpr int NMB = MODE == 0 ? 2 : 4;                          // MFMAs of a B slot
    auto interleave = [&](auto NM, auto NV) {
        constexpr int nm = decltype(NM)::value, nv = decltype(NV)::value;
        if constexpr (nm > 0 && nv > 0) {
#pragma unroll
            for (int k = 0; k < nm; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (nv + nm - 1) / nm, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#define DITTO_BWD_MIX(nm, nv) interleave(std::integral_constant<int, (nm)>{}, std::integral_constant<int, (nv)>{})

    dma_tile(0, 0);
    if (ntile > 1) dma_tile(1, 1);
    wait_tile(ntile > 1 ? 1 : 0);
    __syncthreads();
    // the compiler's own wait for the fragment loads above must fall HERE, not at their first use inside the tile loop: it
    // counts only the loads it knows, so its vmcnt(0) in the loop would drain the whole DMA ring once per tile
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(f0[ks]), "+v"(f1[ks]));
    asm volatile("" : "+v"(own_L), "+v"(own_delta));
    ld_rows(smem, 0, fa);

    // one iteration: the second half of tile t (CUR) and the first half of tile t+1 (NXT)
    auto body = [&](int t, auto HAS_CUR, auto HAS_NEXT) {
        constexpr bool CUR = decltype(HAS_CUR)::value, NXT = decltype(HAS_NEXT)::value;
        constexpr int MA = NXT ? 2 * G : 0, VC = CUR ? NVH : 0, VN = NXT ? NVH : 0;
        const char* cb = smem + (t & (NBUF - 1)) * BUF;          // tile t
        const char* nb = smem + ((t + 1) & (NBUF - 1)) * BUF;    // tile t + 1
        constexpr int MBC = CUR ? NMB : 0, MBN = NXT ? NMB : 0;
        // r0: A0(t+1) + V2b(t)
        if constexpr (NXT) a_step(0);
        if constexpr (CUR) v_half(t, 2, 1);
        DITTO_BWD_MIX(MA, VC);
        if constexpr (NXT) ld_rows(nb, 1, fa);
        if constexpr (CUR) ld_stats(cb, 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        // r1: B2(t) + V3a(t)
        if constexpr (CUR) { b_slot(2); v_half(t, 3, 0); }
        DITTO_BWD_MIX(MBC, VC);
        if constexpr (CUR) ld_tr(cb, 3, tf);
        if constexpr (NXT) ld_stats(nb, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // r2: A1(t+1) + V3b(t)
        if constexpr (NXT) a_step(1);
        if constexpr (CUR) v_half(t, 3, 1);
        DITTO_BWD_MIX(MA, VC);
        if constexpr (NXT) { ld_rows(nb, 2, fa); ld_stats(nb, 0, 1); }
        __builtin_amdgcn_sched_barrier(0);
        // r3: B3(t) + V0a(t+1)
        if constexpr (CUR) b_slot(3);
        if constexpr (NXT) v_half(t + 1, 0, 0);
        DITTO_BWD_MIX(MBC, VN);
        if constexpr (NXT) { ld_tr(nb, 0, tf); ld_stats(nb, 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        // r4: A2(t+1) + V0b(t+1)
        if constexpr (NXT) { a_step(2); v_half(t + 1, 0, 1); }
        DITTO_BWD_MIX(MA, VN);
        if constexpr (NXT) { ld_rows(nb, 3, fa); ld_stats(nb, 1, 1); }
        __builtin_amdgcn_sched_barrier(0);
        // r5: B0(t+1) + V1a(t+1)
        if constexpr (NXT) { b_slot(0); v_half(t + 1, 1, 0); }
        DITTO_BWD_MIX(MBN, VN);
        if constexpr (NXT) { ld_tr(nb, 1, tf); ld_stats(nb, 2, 0); }
        __builtin_amdgcn_sched_barrier(0);
        // r6: A3(t+1) + V1b(t+1)
        if constexpr (NXT) { a_step(3); v_half(t + 1, 1, 1); }
        DITTO_BWD_MIX(MA, VN);
        if constexpr (NXT) ld_stats(nb, 2, 1);
        if (t + 2 < ntile) {
            if constexpr ((DITTO_DIAG_BWD & 2) == 0) bwd_vm_wait<0>();   // tile t + 2 is the youngest request
            if constexpr ((DITTO_DIAG_BWD & 4) == 0) __builtin_amdgcn_s_barrier();
            if constexpr ((DITTO_DIAG_BWD & 2) == 0)
                if (t + 3 < ntile) dma_tile(t + 3, (t + 3) & (NBUF - 1));
            ld_rows(smem + ((t + 2) & (NBUF - 1)) * BUF, 0, fa);
        }
        __builtin_amdgcn_sched_barrier(0);
        // r7: B1(t+1) + V2a(t+1)
        if constexpr (NXT) { b_slot(1); v_half(t + 1, 2, 0); }
        DITTO_BWD_MIX(MBN, VN);
        if constexpr (NXT) { ld_tr(nb, 2, tf); ld_stats(nb, 3, 0); }
        __builtin_amdgcn_sched_barrier(0);
    };
#undef DITTO_BWD_MIX
    body(-1, std::false_type{}, std::true_type{});
    for (int t = 0; t + 1 < ntile; ++t) body(t, std::true_type{}, std::true_type{});
    body(ntile - 1, std::true_type{}, std::false_type{});
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[0][i] *= p.scale; acc0[1][i] *= p.scale; }   // dQ = scale dS' k, dK = scale dS'^T q

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

size_t attention_bwd_stats_bytes(int B, int H, int Sq) { return (size_t)B * H * ((Sq + TILE - 1) / TILE) * 128 * sizeof(float); }

// fused backward (head_dim 64): stats = attention_bwd_stats_bytes of scratch for the per-tile {L, delta} records (the dq kernel
// writes them, the dk,dv kernel reads them); a.lse from the training forward; O = a.o_bf16, or a.h_after - a.h_before
hipError_t launch_attention_bwd64(const AttnBwdArgs& a, float* stats, hipStream_t s) {
    if (a.dh != DH || a.B <= 0 || a.H <= 0 || a.Sq <= 0 || a.Skv <= 0 || !stats || !a.lse) return hipErrorInvalidValue;
    if ((a.ldq | a.ldk | a.ldv | a.lddo) % 8 || (a.lddq | a.lddk | a.lddv) % 4) return hipErrorInvalidValue;
    if (a.o_bf16 ? a.ldo % 8 != 0 : (!a.h_after || !a.h_before || a.ldh % 4 != 0)) return hipErrorInvalidValue;
    BwdParams p;
    p.q = (const bf16*)a.q; p.ldq = a.ldq; p.k = (const bf16*)a.k; p.ldk = a.ldk; p.v = (const bf16*)a.v; p.ldv = a.ldv;
    p.dout = (const bf16*)a.dout; p.lddo = a.lddo;
    p.dq = (bf16*)a.dq; p.lddq = a.lddq; p.dk = (bf16*)a.dk; p.lddk = a.lddk; p.dv = (bf16*)a.dv; p.lddv = a.lddv;
    p.stats = stats; p.lse = a.lse; p.o = (const bf16*)a.o_bf16; p.ldo = a.ldo;
    p.h_after = a.h_after; p.h_before = a.h_before; p.ldh = a.ldh;
    p.B = a.B; p.H = a.H; p.Sq = a.Sq; p.Skv = a.Skv;
    p.scale = a.scale; p.scale_log2 = a.scale * 1.4426950408889634f;
    p.drop_thr = dropout_threshold(a.dropout_p);
    p.keep_scale = p.drop_thr ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
    p.seed_lo = (unsigned)(a.seed & 0xFFFFFFFFu); p.seed_hi = (unsigned)(a.seed >> 32); p.layer = a.layer;
    p.rope_cos = a.rope_cos; p.rope_sin = a.rope_sin;
    if ((a.rope_cos == nullptr) != (a.rope_sin == nullptr) || (a.rope_cos && a.Sq != a.Skv)) return hipErrorInvalidValue;
    // diagnostic: DITTO_BWD_LDS_PAD=32768 pads the launch's LDS so that ONE workgroup fits a CU (one wave per SIMD): a lone wave's
    // timeline is serial, so knock-out builds (tools/bwd_knockout.sh) then read as additive shares of a tile's cycles
    static const int lds_pad = [] { const char* e = getenv("DITTO_BWD_LDS_PAD"); return e ? atoi(e) : 0; }();
    const int LDS0 = NBUF * 2 * IMG + lds_pad, LDS1 = NBUF * (2 * IMG + STAT_BYTES) + lds_pad;
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&attn64_bwd_kernel<0, false, false>),
                                                   reinterpret_cast<const void*>(&attn64_bwd_kernel<0, true, false>),
                                                   reinterpret_cast<const void*>(&attn64_bwd_kernel<0, false, true>),
                                                   reinterpret_cast<const void*>(&attn64_bwd_kernel<0, true, true>),
                                                   reinterpret_cast<const void*>(&attn64_bwd_kernel<1, false>),
                                                   reinterpret_cast<const void*>(&attn64_bwd_kernel<1, true>)}, LDS1 + 32768)) return e;
    const bool drop = p.drop_thr != 0, rag = (a.Skv & (TILE - 1)) != 0;
    p.nblk = (a.Sq + BLK - 1) / BLK;
    const dim3 g0(p.nblk * a.H * a.B), blk(256);
    if (drop && rag) hipLaunchKernelGGL((attn64_bwd_kernel<0, true, true>), g0, blk, LDS0, s, p);
    else if (drop) hipLaunchKernelGGL((attn64_bwd_kernel<0, true, false>), g0, blk, LDS0, s, p);
    else if (rag) hipLaunchKernelGGL((attn64_bwd_kernel<0, false, true>), g0, blk, LDS0, s, p);
    else hipLaunchKernelGGL((attn64_bwd_kernel<0, false, false>), g0, blk, LDS0, s, p);
    p.nblk = (a.Skv + BLK - 1) / BLK;
    if (drop) hipLaunchKernelGGL((attn64_bwd_kernel<1, true>), dim3(p.nblk * a.H * a.B), blk, LDS1, s, p);
    else hipLaunchKernelGGL((attn64_bwd_kernel<1, false>), dim3(p.nblk * a.H * a.B), blk, LDS1, s, p);
    return hipGetLastError();
}

}  // namespace ditto
