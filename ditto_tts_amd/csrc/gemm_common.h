// gemm_common.h — parameters and fused epilogues shared by the three GEMM tile structures
// (gemm.hip: 128x128 two-barrier; gemm256.hip: persistent 256x256 eight-phase; gemm_p128.hip: persistent
// 256x128 with a 3-slot LDS ring).
//
// Both structures compute C^T with swapped MFMA operands (weights as the instruction's A operand), so a
// lane holds, for ONE output row, 4 consecutive columns of each of the 4 16-column blocks of its wave's
// 64-column span:  acc[n][e] = C[row][cbase + 16*n + 4*fq + e],  fq = lane >> 4.
#pragma once
#include "common.h"
#include "kernels.h"

namespace ditto {

struct GemmParams {
    const bf16* A; int lda;
    const bf16* W; int ldw; int w_rows;
    const float* bias;
    const float* wscale;   // fp8 GEMMs: per-output-column weight scale (null otherwise)
    const float* residual; int ldr;
    void* out; int ldo;
    bf16* out2; int ldo2;
    const float* rope_cos; const float* rope_sin; int rope_rpb; int rope_cols;
    const float* rope_freq_rev;   // inv_freq / (2 pi) [32]: angles computed in the epilogue (v_sin / v_cos) instead of tables
    int M, N, K;
    int tiles_m, tiles_n;
    int group_n;       // super-column width in column tiles (tile order, see tile_to_mn)
    int tile_stride;   // gemm256: persistent workgroups walk tiles b, b + stride, ...
    int stagger_ticks; // GF_STAGGER_START: s_memrealtime ticks (100 MHz) per quarter tile
    int flags;         // experiment switches (ditto_set_option("gemm_flags")): see GF_* below
    int k_splits; size_t split_stride;   // gemm128 only: see GemmArgs
    int batch_inner;                      // gemm128 only: blockIdx.y = zo * batch_inner + zi
    long long sA[2], sW[2], sO[2], sR[2];
    int out_esz;                          // bytes per output element (batch offset of `out`)
    const bf16* pre; int ldpre;           // EPI_GATED_BWD
    float* colpart;                       // EPI_GATED_BWD
};

enum { GF_RELAXED_WAIT = 1,        // tile-start wait skips over the previous tile's epilogue stores
       GF_DIAG_NO_STORE = 2,       // DIAGNOSTIC (wrong results): epilogue computes but does not store
       GF_DIAG_NO_EPILOGUE = 4,    // DIAGNOSTIC (wrong results): no epilogue at all
       GF_STAGGER_START = 8,       // de-synchronise the persistent workgroups: group g of 4 starts g/4 tile late (off by default)
       GF_DIAG_LINEAR_STORE = 16,  // DIAGNOSTIC (wrong results): bf16 stores go to lane-linear addresses
       GF_STORE_SC1 = 32,          // output stores write-through, line dropped from L2 (sc1)
       GF_STORE_NT = 64,           // fp32 output stores non-temporal (nt)
       GF_ROWMAJOR_TILES = 128,    // A/B switch: plain row-major tile order instead of super-columns
       GF_WIDE_PHASE = 256,        // gemm256: 32 MFMAs per barrier pair, LDS reads retired before the barrier
       GF_DIAG_SMALL_OUT = 512,    // DIAGNOSTIC (wrong results): every tile stores into rows 0..255 (stays in L2)
       GF_SLOW_EPILOGUE = 1024,    // A/B: never take the specialised straight-line epilogue (was: RoPE table loads A/B, retired)
       GF_TN_TWO_BUFFER = 2048,    // gemm_tn.hip A/B: the first (two-buffer) weight-gradient kernel
       GF_TN_NARROW = 4096,        // ditto_train.hip A/B: weight gradients on the 128 x 128 kernel only (no 256 x 256 tiles)
       GF_PP_PARITY = 8192,
       GF_NO_FLAT_K = 16384 };     // gemm256 wide-phase A/B: the round-3 tile switch (prologue DMA between the K loop and the epilogue) instead of the flat K loop

// Tile order.  An XCD (private 4 MiB L2) receives a contiguous range of the linear tile index (xcd_remap); within it
// the tiles run down M inside a SUPER-COLUMN of `G` column tiles, so the tiles an XCD works on at one time are a
// few row panels x G column tiles and the weight rows of the super-column stay L2-resident while the XCD walks
// down M.  With the plain row-major order an XCD sweeps all N/BN column tiles of a row panel: for the gated MLP
// GEMM that working set is the whole 9.4 MB weight matrix, and rocprofv3 FETCH_SIZE showed ~1 GB of L2-miss
// traffic per launch against 60 MB of algorithmic input.
DITTO_DEV void tile_to_mn(int t, int tiles_m, int tiles_n, int G, int& tm, int& tn) {
    const int per_sc = tiles_m * G;
    const int sc = t / per_sc;
    const int r = t - sc * per_sc;
    const int w = (tiles_n - sc * G) < G ? (tiles_n - sc * G) : G;   // last super-column may be narrower
    tm = r / w;
    tn = sc * G + r % w;
}
// host side: split tiles_n into ceil(tiles_n / 8) super-columns of (nearly) equal width
inline int pick_group_n(int tiles_n, int flags) {
    if (g_gemm_group > 0) return g_gemm_group < tiles_n ? g_gemm_group : tiles_n;   // ditto_set_option("gemm_group"): A/B
    // measured in-model: with <= 12 column tiles the weights fit the 4 MiB L2 and row-major is ~5 % faster (QKV)
    if ((flags & GF_ROWMAJOR_TILES) || tiles_n <= 12) return tiles_n;
    const int nsc = (tiles_n + 7) / 8;
    return (tiles_n + nsc - 1) / nsc;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// LDS-DMA (global_load_lds_dwordx4) through inline asm: 64 lanes x 16 B land at LDS byte address
// `lds_dst` (wave-uniform) + lane*16.  Why not the builtin: in the PERSISTENT kernels the next tile's first DMA is
// issued into buffer 0 at the end of the tile loop, and hipcc — which tracks the builtin as a pending LDS write —
// then cannot prove that it does not alias the first ds_reads of the K-loop and drains the whole DMA pipeline
// with s_waitcnt vmcnt(0) at the top of EVERY iteration (seen in the .s).  An asm DMA is invisible to that pass;
// every wait on DMA data in those kernels is hand-counted (s_waitcnt vmcnt(N) + s_barrier before the reader).
// hipcc's own counted waits stay safe: its loads (epilogue bias / residual / RoPE tables) are issued after the
// asm DMAs, and vmcnt retires in order, so extra OLDER operations only make its waits stricter.
// M0 holds the LDS base for the instruction and is compiler-reserved: saved and restored in the same statement.
DITTO_DEV void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// The same with a cache-policy suffix on the load ("nt", "sc1", ...): A/B builds of the operand streams' L2 retention
// (tools/build_diag.sh -DDITTO_G256_A_POLICY='"nt"' / -DDITTO_G256_W_POLICY='"nt"'; round 4: the gated GEMM's A panel is
// re-fetched from the Infinity Cache every tile round because A + W of a round exceed an XCD's 4 MiB L2)
#define DITTO_GLDS16_POLICY(gsrc, lds_dst, POLICY)                                                                        \
    do {                                                                                                                  \
        unsigned keep_;                                                                                                   \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off " POLICY          \
                     "\n\ts_mov_b32 m0, %0"                                                                               \
                     : "=&s"(keep_)                                                                                       \
                     : "v"(gsrc), "s"(lds_dst)                                                                            \
                     : "memory");                                                                                         \
    } while (0)

// LDS-DMA with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: the per-K-tile address update
// is one scalar add on the base instead of a 64-bit vector add per load (the loads of a kernel's main loop differ only
// by the K offset).
DITTO_DEV void glds16_so(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// 16-byte global load the COMPILER DOES NOT TRACK: the caller waits for it with its own counted s_waitcnt (an asm
// statement that names the result as an in/out operand, so no use can be scheduled above the wait).
DITTO_DEV f32x4 gload16_asm(const void* ptr) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}

// Bias of the lane's 4 x 4 columns, loaded ONCE per wave (the same for every row of the tile): keeps the row
// loop free of dependent global loads.
DITTO_DEV void load_bias(const GemmParams& p, int cbase, int fq, f32x4 (&b)[4]) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int c = cbase + n * 16 + fq * 4;
        b[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && c < p.N) b[n] = *reinterpret_cast<const f32x4*>(p.bias + c);
    }
}

// 16-byte global store with a cache policy: plain, sc1 (write-through, no L2 residency) or nt.
// Measured on MI355X (tools/gemm_bench.py, same-process A/B): nt on the fp32 in-place residual epilogue
// (d x d out-proj, fc2) -20..-26 % kernel time; on bf16 outputs nt / sc1 are within +-3 %, so nt applies to
// fp32 outputs only.
template <bool F32_OUT, bool FAST = false>
DITTO_DEV void store16(void* ptr, u32x4 v, int flags) {
    if constexpr (FAST) {   // production flags known at compile time: nt on fp32 outputs, plain on bf16, no diagnostics
        // s_nop 1: a 16-byte store reads its data registers over several cycles; hipcc pads its own stores against a
        // VALU overwrite of them but knows nothing about this one
        if constexpr (F32_OUT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
        else *reinterpret_cast<u32x4*>(ptr) = v;
        return;
    }
    if (flags & GF_STORE_SC1) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
    } else if (F32_OUT && (flags & GF_STORE_NT)) {
        asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
    } else {
        *reinterpret_cast<u32x4*>(ptr) = v;
    }
}

// Widened bf16 row store (guide T21, with v_permlane16_swap): a lane holds 4 packed bf16 (8 B) of two adjacent
// 16-column blocks (pa = block nb, pb = block nb+1).  Lanes l and l^16 (fq and fq^1: same row, neighbouring
// 4-column groups) trade halves, after which the even-fq lane owns 8 CONSECUTIVE columns of block nb and the
// odd-fq lane 8 consecutive columns of block nb+1: one 16-B store per lane instead of two 8-B stores.  The tile
// epilogue is store-ISSUE-bound (32 dwordx2 per lane measured ~10 us per 256x256 tile), so halving the
// instruction count at equal bytes is what matters.  Must be called by both lanes of every (l, l^16) pair.
template <bool FAST = false>
DITTO_DEV void store_bf16_pair(bf16* rowp, int col0 /* column of block nb */, u32x2 pa, u32x2 pb, int fq, int ncols,
                                int flags = 0, int ld = 0) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
    // even fq: [own nb | partner's nb]      odd fq: [partner's nb+1 | own nb+1]
    const int odd = fq & 1;
    int c = col0 + odd * 16 + 4 * (fq - odd);
    if constexpr (FAST) {   // interior tile, production flags: no guard, no diagnostics -> straight-line code
        u32x4 st;
        st[0] = r0[0]; st[1] = r1[0]; st[2] = r0[1]; st[3] = r1[1];
#ifdef DITTO_DIAG_FAST_NOSTORE   // timing experiment (WRONG results): the fast epilogue computes, keeps live, does not store
        asm volatile("" ::"v"(st), "v"(rowp + c));
#elif defined(DITTO_BF16_STORE_NT)   // A/B build: non-temporal bf16 output stores
        asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(rowp + c), "v"(st) : "memory");
#elif defined(DITTO_BF16_STORE_SC1)  // A/B build: write-through bf16 output stores
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(rowp + c), "v"(st) : "memory");
#else
        *reinterpret_cast<u32x4*>(rowp + c) = st;
#endif
        return;
    }
    if (flags & GF_DIAG_LINEAR_STORE) {   // same instruction, same bytes, 1 KiB contiguous per wave-instruction
        rowp -= (size_t)(threadIdx.x & 15) * ld;
        c = (col0 & ~63) + (threadIdx.x & 63) * 8;
        if (c + 8 > ncols) c = ncols - 8;
    }
    if (c < ncols) {
        u32x4 st;
        st[0] = r0[0]; st[1] = r1[0]; st[2] = r0[1]; st[3] = r1[1];
        if (flags & GF_DIAG_NO_STORE) asm volatile("" ::"v"(st));   // computed, kept live, not stored
        else store16<false>(rowp + c, st, flags);
    }
}

#ifdef DITTO_DIAG_GATED_CHEAP   // tools/build_diag.sh: gated epilogue with trivial arithmetic (what is NOT the activation math?)
DITTO_DEV f32x2 cheap_gate2(f32x2 x, f32x2 g) { return x * g; }
#define fast_gelu_sigmoid2 cheap_gate2
#endif
#ifdef DITTO_GATED_H    // A/B build of the packed-fp16 sigmoid form (common.h)
#define fast_gelu_sigmoid2 fast_gelu_sigmoid2_h
#endif
#ifdef DITTO_GATED_AS   // A/B build of the second packed form (A&S erf, 4 transcendentals per output)
#define fast_gelu_sigmoid2 fast_gelu_sigmoid2_as
#endif
#ifdef DITTO_GATED_V1   // A/B build of the first packed form of the gated-MLP math (tools/README.md)
#define fast_gelu_sigmoid2 fast_gelu_sigmoid2_v1
#endif

// One output row x the wave's 64-column span.  `row` < M is checked by the caller.
// FAST: the tile is interior in M and N, there is no bf16 side copy (out2) and the flags are the production set
// (GF_STORE_NT, no diagnostics / sc1): every guard and flag test compiles out, so the eight row blocks of a wave tile
// are ONE basic block and hipcc interleaves their independent dependency chains (as written before, each row block
// ended in 6-8 wave-uniform branches around its store and the transcendental latencies of a block were exposed).
template <int EPI, bool FAST = false>
DITTO_DEV void epilogue_row(const GemmParams& p, int row, int cbase, const f32x4 (&acc)[4], const f32x4 (&bias)[4],
                            int fq) {
    const int c4 = fq * 4;
    if constexpr (EPI == EPI_GATED_FP8) {
        // as EPI_GATED with an fp8 e4m3 result: 4 values = one dword per 16-column block; the lane^16 exchange
        // makes it 8 consecutive bytes per lane
        unsigned pk[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const f32x4 h = acc[2 * pr] + bias[2 * pr], g = acc[2 * pr + 1] + bias[2 * pr + 1];
            const f32x2 o01 = fast_gelu_sigmoid2(f32x2{h[0], h[1]}, f32x2{g[0], g[1]});
            const f32x2 o23 = fast_gelu_sigmoid2(f32x2{h[2], h[3]}, f32x2{g[2], g[3]});
            pk[pr] = pack_fp8x4(o01[0], o01[1], o23[0], o23[1]);
        }
        const auto r = __builtin_amdgcn_permlane16_swap(pk[0], pk[1], false, false);
        const int odd = fq & 1;
        const int c = cbase / 2 + odd * 16 + 4 * (fq - odd);
        if (c < p.N / 2) {
            u32x2 st;
            st[0] = r[0]; st[1] = r[1];
            *reinterpret_cast<u32x2*>((unsigned char*)p.out + (size_t)row * p.ldo + c) = st;
        }
    } else if constexpr (EPI == EPI_GATED || EPI == EPI_GATED_PRE) {
        // packed columns: 16 x fc1 | 16 x gate | 16 x fc1 | 16 x gate  (reference src/components/DiT.py:153-155)
        u32x2 pk[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const f32x4 h = acc[2 * pr] + bias[2 * pr], g = acc[2 * pr + 1] + bias[2 * pr + 1];
            const f32x2 o01 = fast_gelu_sigmoid2(f32x2{h[0], h[1]}, f32x2{g[0], g[1]});
            const f32x2 o23 = fast_gelu_sigmoid2(f32x2{h[2], h[3]}, f32x2{g[2], g[3]});
            pk[pr][0] = pack_bf16x2(o01[0], o01[1]);
            pk[pr][1] = pack_bf16x2(o23[0], o23[1]);
        }
        store_bf16_pair<FAST>((bf16*)p.out + (size_t)row * p.ldo, cbase / 2, pk[0], pk[1], fq, p.N / 2, p.flags, p.ldo);
        if ((EPI == EPI_GATED_PRE) || (!FAST && p.out2)) {   // training forward: the pre-activations (bias added, interleaved packed order) for the backward
            u32x2 pr[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const f32x4 v = acc[n] + bias[n];
                pr[n][0] = pack_bf16x2(v[0], v[1]);
                pr[n][1] = pack_bf16x2(v[2], v[3]);
            }
            bf16* rowp = p.out2 + (size_t)row * p.ldo2;
            store_bf16_pair<FAST>(rowp, cbase, pr[0], pr[1], fq, p.N, p.flags & ~GF_DIAG_LINEAR_STORE, p.ldo2);
            store_bf16_pair<FAST>(rowp, cbase + 32, pr[2], pr[3], fq, p.N, p.flags & ~GF_DIAG_LINEAR_STORE, p.ldo2);
        }
    } else if constexpr (EPI == EPI_QKV_ROPE) {
        float v[4][4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[n][e] = acc[n][e] + bias[n][e];
        }
        if (FAST || cbase < p.rope_cols) {  // a q or k head (FAST: the caller runs v columns through EPI_BIAS_BF16) (width 64): half-split RoPE, pair (j, j+32); DiT.py:52-72
            const int pos = row % p.rope_rpb;
            float r[4][4];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                f32x4 cs, sn;
                if (p.rope_freq_rev) {
                    // cos / sin of pos * inv_freq[j] right here: v_fract + v_sin + v_cos on the angle in revolutions
                    // (the table loads were 8 dependent row-indexed f32x4 loads per row block in a serial epilogue: 27 us of
                    // the 129 us QKV GEMM, measured with the no-epilogue diagnostic).  fp32 fract at <= 4096 positions keeps
                    // 14 fraction bits: 2e-4 rad, 20x below the bf16 rounding of q and k.
                    const f32x4 fr = *reinterpret_cast<const f32x4*>(p.rope_freq_rev + n * 16 + c4);
                    const float posf = (float)pos;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float rev = __builtin_amdgcn_fractf(posf * fr[e]);
                        cs[e] = __builtin_amdgcn_cosf(rev);
                        sn[e] = __builtin_amdgcn_sinf(rev);
                    }
                } else {
                    cs = *reinterpret_cast<const f32x4*>(p.rope_cos + (size_t)pos * 32 + n * 16 + c4);
                    sn = *reinterpret_cast<const f32x4*>(p.rope_sin + (size_t)pos * 32 + n * 16 + c4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = v[n][e], hi = v[n + 2][e];
                    r[n][e] = lo * cs[e] - hi * sn[e];      // t*cos + (-t[j+32])*sin
                    r[n + 2][e] = hi * cs[e] + lo * sn[e];  // t*cos + ( t[j-32])*sin
                }
            }
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[n][e] = r[n][e];
        }
        u32x2 pk[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            pk[n][0] = pack_bf16x2(v[n][0], v[n][1]);
            pk[n][1] = pack_bf16x2(v[n][2], v[n][3]);
        }
        bf16* rowp = (bf16*)p.out + (size_t)row * p.ldo;
        store_bf16_pair<FAST>(rowp, cbase, pk[0], pk[1], fq, p.N, p.flags, p.ldo);
        store_bf16_pair<FAST>(rowp, cbase + 32, pk[2], pk[3], fq, p.N, p.flags, p.ldo);
    } else if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_RELU_BF16) {
        u32x2 pk[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x4 v = acc[n] + bias[n];
            if constexpr (EPI == EPI_BIAS_RELU_BF16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            pk[n][0] = pack_bf16x2(v[0], v[1]);
            pk[n][1] = pack_bf16x2(v[2], v[3]);
        }
        bf16* rowp = (bf16*)p.out + (size_t)row * p.ldo;
        store_bf16_pair<FAST>(rowp, cbase, pk[0], pk[1], fq, p.N, p.flags, p.ldo);
        store_bf16_pair<FAST>(rowp, cbase + 32, pk[2], pk[3], fq, p.N, p.flags, p.ldo);
    } else {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int c = cbase + n * 16 + c4;
            if (!FAST && c >= p.N) continue;
            f32x4 v = acc[n] + bias[n];
            if constexpr (EPI == EPI_BIAS_RES_F32) {
                if (FAST || p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (size_t)row * p.ldr + c);
            }
            if (!FAST && (p.flags & GF_DIAG_NO_STORE)) asm volatile("" ::"v"(v));
            else store16<true, FAST>((float*)p.out + (size_t)row * p.ldo + c, __builtin_bit_cast(u32x4, v), p.flags);
            if constexpr (EPI == EPI_BIAS_RES_F32) {
                if (!FAST && p.out2) {   // bf16 side copy (last layer only): plain 8-B stores, off the hot path
                    u32x2 st;
                    st[0] = pack_bf16x2(v[0], v[1]);
                    st[1] = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(p.out2 + (size_t)row * p.ldo2 + c) = st;
                }
            }
        }
    }
}

// v += the same value of the 15 other lanes of the 16-lane DPP row (the lanes that hold the other rows of a 16-row block):
// four v_add_f32_dpp (xor 1, xor 2 by quad_perm, then the half-row and row mirrors: after the first two steps a quad holds one value)
DITTO_DEV float dpp_row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

// EPI_GATED_BWD, one wave's 128 x 64 block of a 256 x 256 tile (training backward; reference src/components/DiT.py:152-154
// differentiated).  acc[m][n][e] = dact[r0 + 16 m][cb + 16 n + 4 fq + e].  The pre-activations of act columns cb + 16 n .. + 15
// sit at packed columns 2 (cb + 16 n) + {0 .. 15 fc1, 16 .. 31 gate}: the store exchange of store_bf16_pair run backwards — the
// even-fq lane loads 8 consecutive fc1 values (its own 4 and its lane^16 partner's), the odd-fq lane 8 gate values, two
// v_permlane16_swap hand every lane its own 4 + 4 — so a row block costs 4 16-B loads and 4 16-B stores per lane.  Column sums
// are taken over the ROUNDED values (what the weight-gradient GEMM reads next), per lane over its 8 rows, then over the 16
// rows of a block by DPP; lanes 0, 16, 32, 48 write the wave's partial row `part_row`.  GUARD: the tile may pass row M.
template <bool GUARD>
DITTO_DEV void epilogue_gated_bwd(const GemmParams& p, int urow /* wave-uniform: first row of the wave's block */, int cb,
                                  const f32x4 (&acc)[8][4], int fq, int frow, int part_row) {
    // dact rounded to bf16 first, as the two-launch path stored it (same bits as that path up to fp contraction), and so that
    // 64 registers hold what 128 did: the loads of two row blocks can be in flight above the arithmetic of the previous two
    unsigned dyp[8][4][2];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            dyp[m][n][0] = pack_bf16x2(acc[m][n][0], acc[m][n][1]);
            dyp[m][n][1] = pack_bf16x2(acc[m][n][2], acc[m][n][3]);
            asm volatile("" : "+v"(dyp[m][n][0]), "+v"(dyp[m][n][1]));
        }
    float cs[4][2][4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) { cs[n][0][e] = 0.f; cs[n][1][e] = 0.f; }
    const int odd = fq & 1;
    const int lc = 2 * cb + odd * 16 + 4 * (fq - odd);      // this lane's 8-column piece of the first packed 32-column group
    // addresses = wave-uniform row base (scalar registers) + ONE 32-bit lane offset per matrix: the row pointers of eight row
    // blocks as 64-bit vector values were 32 registers this epilogue does not have
    const unsigned pre_lane = (unsigned)(frow * p.ldpre + lc) * 2u, out_lane = (unsigned)(frow * p.ldo + lc) * 2u;
    auto row_ok = [&](int m) { return !GUARD || urow + m * 16 + frow < p.M; };   // the same for a lane and its lane^16 partner (same row)
    auto pre_ptr = [&](int m) {
        if (GUARD && !row_ok(m)) return (const char*)p.pre + (size_t)(p.M - 1) * p.ldpre * 2 + (size_t)lc * 2;
        return (const char*)p.pre + (size_t)(urow + m * 16) * p.ldpre * 2 + pre_lane;
    };
    // (three pairs in flight instead of two — 24 loads per wave — measured no faster: 317 vs 308 us per launch at C2, B = 32;
    // the epilogue is as much vector arithmetic, ~3 000 instructions per wave and tile, as it is traffic)
    u32x4 ld[2][2][4];                                      // [buffer][row block of the pair][n]
    auto load_pair = [&](int buf, int mp) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* prow = pre_ptr(2 * mp + i);
#pragma unroll
            for (int n = 0; n < 4; ++n) ld[buf][i][n] = *reinterpret_cast<const u32x4*>(prow + 64 * n);
        }
    };
    load_pair(0, 0);
#pragma unroll
    for (int mp = 0; mp < 4; ++mp) {
        if (mp < 3) load_pair((mp + 1) & 1, mp + 1);
        asm volatile("" ::: "memory");                      // the next pair's loads stay above this pair's arithmetic and stores
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = 2 * mp + i;
            const bool ok = row_ok(m);
            char* drow = (char*)p.out + (size_t)(urow + m * 16) * p.ldo * 2 + out_lane;   // (not dereferenced past row M)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const u32x4 l4 = ld[mp & 1][i][n];
                const auto t0 = __builtin_amdgcn_permlane16_swap(l4[0], l4[2], false, false);
                const auto t1 = __builtin_amdgcn_permlane16_swap(l4[1], l4[3], false, false);
                f32x2 da01, dg01, da23, dg23;
                gated_bwd2(f32x2{bf16_lo(t0[0]), bf16_hi(t0[0])}, f32x2{bf16_lo(t0[1]), bf16_hi(t0[1])},
                           f32x2{bf16_lo(dyp[m][n][0]), bf16_hi(dyp[m][n][0])}, da01, dg01);
                gated_bwd2(f32x2{bf16_lo(t1[0]), bf16_hi(t1[0])}, f32x2{bf16_lo(t1[1]), bf16_hi(t1[1])},
                           f32x2{bf16_lo(dyp[m][n][1]), bf16_hi(dyp[m][n][1])}, da23, dg23);
                const unsigned pa0 = pack_bf16x2(da01[0], da01[1]), pa1 = pack_bf16x2(da23[0], da23[1]);
                const unsigned pg0 = pack_bf16x2(dg01[0], dg01[1]), pg1 = pack_bf16x2(dg23[0], dg23[1]);
                if (ok) {
                    cs[n][0][0] += bf16_lo(pa0); cs[n][0][1] += bf16_hi(pa0); cs[n][0][2] += bf16_lo(pa1); cs[n][0][3] += bf16_hi(pa1);
                    cs[n][1][0] += bf16_lo(pg0); cs[n][1][1] += bf16_hi(pg0); cs[n][1][2] += bf16_lo(pg1); cs[n][1][3] += bf16_hi(pg1);
                }
                const auto s0 = __builtin_amdgcn_permlane16_swap(pa0, pg0, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(pa1, pg1, false, false);
                u32x4 st;
                st[0] = s0[0]; st[1] = s1[0]; st[2] = s0[1]; st[3] = s1[1];
                if (ok) *reinterpret_cast<u32x4*>(drow + 64 * n) = st;
            }
        }
        asm volatile("" ::: "memory");
    }
    float* prt = p.colpart + (size_t)part_row * (2 * (size_t)p.N) + 2 * cb + 4 * fq;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = dpp_row_sum16(cs[n][t][e]);
            if (frow == 0) *reinterpret_cast<f32x4*>(prt + 32 * n + 16 * t) = v;
        }
}

// May the tile at (m0, n0) of extent (BM_, BN_) take the FAST epilogue?  (wave-uniform)
template <int EPI>
DITTO_DEV bool epilogue_fast_ok(const GemmParams& p, int m0, int n0, int BM_, int BN_) {
    constexpr int must_off = GF_DIAG_NO_STORE | GF_DIAG_NO_EPILOGUE | GF_DIAG_LINEAR_STORE | GF_DIAG_SMALL_OUT | GF_STORE_SC1;
    if (m0 + BM_ > p.M || n0 + BN_ > p.N || (p.out2 && EPI != EPI_GATED_PRE) || (p.flags & must_off) || !(p.flags & GF_STORE_NT)) return false;
    if ((p.flags & GF_SLOW_EPILOGUE) || !p.bias) return false;
    if constexpr (EPI == EPI_BIAS_RES_F32) return p.residual != nullptr;
    if constexpr (EPI == EPI_QKV_ROPE) return p.rope_freq_rev != nullptr;   // table-free angles (the model path)
    return true;
}

// gemm256.hip
hipError_t launch_gemm256(const GemmParams& p, GemmEpilogue epi, hipStream_t s);
hipError_t launch_gemm256_fp8(const GemmParams& p, GemmEpilogue epi, hipStream_t s);
// gemm_p128.hip
hipError_t launch_gemm_p128(const GemmParams& p, GemmEpilogue epi, hipStream_t s);
// gemm192.hip
bool gemm192_supports(GemmEpilogue epi);
hipError_t launch_gemm192(const GemmParams& p, GemmEpilogue epi, hipStream_t s);
// gemm_fr.hip / gemm_fr64.hip: full-row N = 768 GEMM, fp32 residual in place, fused LayerNorm -> u bf16 (gamma/u null: none)
struct FrParams {
    GemmParams g;
    const float* gamma; const float* beta;   // LayerNorm affine of the fused norm (null: no LayerNorm output)
    bf16* u; int ldu;                         // LayerNorm output
    int rot_period;                           // > 0: 128-row tiles t and t + rot_period start their K loop at the same place
    int stagger_ticks;                        // gemm_fr64: start delay (10 ns ticks) of the workgroup holding its CU's second LDS allocation
    bool u_fp8;                               // gemm_fr64, N = 1024: u is fp8 e4m3 bytes ([M, ldu] bytes) for the fp8 linear path
    bool hb = false;                          // gemm_frd only: residual and out are BF16 [M, ldr / ldo] (the bf16 residual stream)
};
bool gemm_fr_supports(int M, int N, int K, size_t lda, size_t ldw);
hipError_t launch_gemm_fr(const GemmParams& p, const float* gamma, const float* beta, void* u_bf16, int ldu, int rot_period,
                          hipStream_t s, bool u_fp8 = false, bool hb = false);   // u_fp8 (N = 1024 only): u is fp8 e4m3 bytes;
                                                                                 // hb: bf16 residual / out (gemm_frd only: else an error)
// The two full-row launches of a DiT block over M rows of width d, each judged on the operand strides IT runs with (the
// kernel builds 32-bit byte offsets from M * lda: fc2 reads A at lda = 4d).  One predicate for the inference forward, the
// LayerNorm chaining decision and the training forward, so that the three cannot disagree.
// gemm_fr64.hip: the same contract and the SAME BITS on 64-row tiles, two workgroups per CU (called by launch_gemm_fr)
hipError_t launch_gemm_fr64(const FrParams& fp, hipStream_t s);
// gemm_frd.hip: the same contract at N = 768 with W fetched straight from L2 into registers (128-row tiles, wave-private W)
hipError_t launch_gemm_frd(const FrParams& fp, hipStream_t s);
bool gemm_fr64_supports(int M, int N, int K, size_t lda, size_t ldw);   // N = 768 or 1024
// d = 768: gemm_fr.hip (or its bit-identical 64-row twin); d = 1024: gemm_fr64.hip only.
inline bool fr_outproj_ok(int M, int d) {
    if (d == 1024) return fr_pays_64(M) && gemm_fr64_supports(M, d, d, (size_t)d, (size_t)d);
    return fr_pays(M) && gemm_fr_supports(M, d, d, (size_t)d, (size_t)d);
}
inline bool fr_fc2_ok(int M, int d) {
    if (d == 1024)   // (not where the tiled fc2 runs whole rounds of 256 x 256 tiles: kernels.h gemm256_whole_rounds; by CLASS rows)
        return fr_pays_64(M) && !gemm256_whole_rounds(opt_class_rows() > 0 ? opt_class_rows() : M, d, 4 * d) &&
               gemm_fr64_supports(M, d, 4 * d, (size_t)4 * d, (size_t)4 * d);
    return fr_pays(M) && gemm_fr_supports(M, d, 4 * d, (size_t)4 * d, (size_t)4 * d);
}
// gemm_pp.hip
bool gemm_pp_supports(const GemmParams& p, GemmEpilogue epi);
hipError_t launch_gemm_pp(const GemmParams& p, GemmEpilogue epi, hipStream_t s);

}  // namespace ditto
