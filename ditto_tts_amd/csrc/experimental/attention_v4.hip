// attention_v4.hip — fused attention forward for head_dim 64 at ONE WAVE PER SIMD with 64 QUERIES PER WAVE (gfx950).
//
// Replaces  softmax(q k^T / sqrt(dh)) v  of the self-attention (reference src/components/DiT.py:131-134, head merge +
// residual :137-139) and of nn.MultiheadAttention's cross-attention (:144-148 -> torch functional.py MHA math), as
// attention.hip does; same contract as attn64v2 (q pre-scaled by scale * log2(e) at pack time) and, by construction,
// the SAME ARITHMETIC IN THE SAME ORDER: results are bitwise those of attn64v2 (tests/test_gpu_kernels.py).
//
// Why another kernel.  attn64v2 / v3 run 2-3 waves per SIMD with one 32-query
// block each.  Their SIMD time per wave and 64-key tile is the SUM of the wave's matrix and vector work (1 476 cycles
// for 640 of MFMA), every wave re-reads every K / V fragment from LDS for its own 32 queries, and the waves meet at a
// barrier per tile.  Here a wave owns the whole 512-entry register file and TWO 32-query blocks, A and B:
//   * the K / V fragments of a tile are read from LDS ONCE and serve both blocks (24 LDS reads per 40 MFMAs instead of
//     24 per 20), the K / V tiles of a workgroup serve 256 queries (half the LDS-DMA bytes per FLOP), one barrier per
//     40 MFMAs instead of one per 20;
//   * the two blocks run HALF A TILE APART in one instruction stream, so that every MFMA has an independent vector
//     chunk of the OTHER block (or of an earlier phase of its own) behind it:
//
//       slots  0.. 7   S'_A(t)   = K(t) Q_A^T - m_A        | exp2 of S'_B(t-1) (second half) -> P_B(t-1)
//       slots  8..19   O_B, l_B += V(t-1) P_B(t-1)          | row max of S'_A(t), raise decision, exp2 -> P_A(t)
//       slots 20..27   S'_B(t)   = K(t) Q_B^T - m_B        | exp2 of S'_A(t) (second half); V(t) fragments from LDS
//       slots 28..39   O_A, l_A += V(t) P_A(t)              | row max of S'_B(t), raise decision, exp2 -> P_B(t);
//                                                             K(t+1) fragments from LDS
//
//     one slot = one v_mfma_f32_32x32x16_bf16 (32 cycles of the matrix pipe, 8 of the wave's issue) + <= ~24 cycles of
//     vector / LDS issue, pinned in that order by scheduling barriers.  Per tile and wave: 40 MFMAs (32 + 8 row-sum
//     MFMAs with an all-ones operand) = 1 280 matrix-pipe cycles for 64 queries x 64 keys, against 2 x 1 476 measured
//     for attn64v2.
//   * MFMAs are asm statements whose operand constraints fix the register file: O, l and the Q fragments live in
//     AGPRs (only MFMAs touch them), S', -m, P, K and V fragments in VGPRs (the vector pipe reads and writes them).
//     Given builtins and 512 registers hipcc puts every accumulator in AGPRs and copies S' out through
//     v_accvgpr_read for the softmax (attention.hip header: +127 VALU per tile).
//   * asm MFMA results have no compiler-inserted wait states: a chain's S' is first READ by the vector pipe two MFMA
//     slots (>= 64 cycles) after its last MFMA was issued — the matrix pipe is busy with those two for as long as the
//     producer needs to retire; O / l are read by the vector pipe only in the (rare) raise path, >= 16 slots after
//     their last MFMA, and in the epilogue behind explicit s_nops.
//
// LDS: K ring of 3 tiles + V ring of 3 tiles (48 KiB).  Group g(t) = { K(t+3), V(t+2) } (4 LDS-DMA pieces of 1 KiB per
// wave) is issued at the start of iteration t into the slots of K(t) / V(t-1), whose last reads precede the barrier that
// ended iteration t-1; it is needed from iteration t+2 on (V(t+2) in slots 20..27, K(t+3) in slots 28..35), so the
// counted wait in front of the barrier that ends iteration t+1 leaves g(t+1) in flight: one barrier per iteration.
#include "attn_common.h"

namespace ditto {

namespace {

// K ring and V ring of NS tiles each (template parameter): NS = 8 -> 128 KiB, the one workgroup a CU holds anyway (512
// registers per wave).  With 3 + 3 slots (the first version) a CU had two tiles = 32 KiB in flight and the kernel ran at the
// LATENCY of its LDS-DMA (3 000 cycles per iteration for 1 500 of issue: 154 us against attn64v2's 133).
constexpr int v4_lds_bytes(int ns) { return 2 * ns * ATT_KV_TILE_BYTES; }
constexpr int V4_QWG = 256;                                            // queries per workgroup (4 waves x 2 blocks x 32)

template <int V>
struct IC4 { static constexpr int value = V; };

// ---- one SLOT = one asm statement: an MFMA with fixed register files ("a" = accumulator half of the unified file, "v" =
//      vector half) followed by a chunk of vector work.  Why one statement: hipcc keeps volatile asm statements in program
//      order (given builtins it sank whole groups of exponentials out of their slots into the join block behind the raise
//      branch, where they ran back to back with the matrix pipe idle), but it pads EVERY asm boundary whose registers
//      look hazardous to it with an s_nop (4 cycles each; 70 of them per iteration with one statement per instruction).
//      Inside a statement nothing is padded: VALU -> VALU dependences are interlocked by the hardware, and the only
//      software hazards here (asm MFMA result -> vector reader, v_exp result -> its consumer) are kept apart by the slot
//      distance (a pair's exponentials are packed in the NEXT slot; S' is first read two MFMA slots after its chain). ----
// A chunk = an exponential part EK (of one block) and / or a row-maximum part XK (of the other block).  Temporaries
// alternate with the slot's parity (two exponential pairs in flight, packed TWO slots after they were issued; two partial
// maxima), so that no two adjacent statements share a register: hipcc pads an asm boundary across which a register is
// written and then read with an s_nop, whatever the statements contain.
enum { EK_NONE, EK_E, EK_PE, EK_P };           // issue a pair | pack the pair of two slots ago + issue a pair | pack only
enum { XK_NONE, XK_M3F, XK_M3, XK_M2 };        // 3 max3 starting a partial maximum | 3 max3 | 2 max3
#define V4_ETXT_0 ""
#define V4_ETXT_1 "\n\tv_exp_f32 %[e0], %[x0]\n\tv_exp_f32 %[e1], %[x1]"
#define V4_ETXT_2 "\n\tv_cvt_pk_bf16_f32 %[fp], %[e0], %[e1]\n\tv_exp_f32 %[e0], %[x0]\n\tv_exp_f32 %[e1], %[x1]"
#define V4_ETXT_3 "\n\tv_cvt_pk_bf16_f32 %[fp], %[e0], %[e1]"
#define V4_XTXT_0 ""
#define V4_XTXT_1 "\n\tv_max3_f32 %[dm], %[y0], %[y0], %[y1]\n\tv_max3_f32 %[dm], %[dm], %[y2], %[y3]\n\tv_max3_f32 %[dm], %[dm], %[y4], %[y5]"
#define V4_XTXT_2 "\n\tv_max3_f32 %[dm], %[dm], %[y0], %[y1]\n\tv_max3_f32 %[dm], %[dm], %[y2], %[y3]\n\tv_max3_f32 %[dm], %[dm], %[y4], %[y5]"
#define V4_XTXT_3 "\n\tv_max3_f32 %[dm], %[dm], %[y0], %[y1]\n\tv_max3_f32 %[dm], %[dm], %[y2], %[y3]"
#define V4_FOUT [e0] "+v"(f.e0), [e1] "+v"(f.e1), [fp] "=&v"(f.pk), [dm] "+v"(f.dm)
#define V4_FIN [x0] "v"(f.x0), [x1] "v"(f.x1), [y0] "v"(f.y0), [y1] "v"(f.y1), [y2] "v"(f.y2), [y3] "v"(f.y3), [y4] "v"(f.y4), [y5] "v"(f.y5)
struct V4Fill {            // the operands of a slot's chunk (unused ones are dummies)
    float& e0; float& e1;  // the exponential pair of this slot's parity
    unsigned& pk;          // the packed pair (written by EK_PE / EK_P only)
    float& dm;             // the partial row maximum of this slot's parity
    float x0, x1;          // S' elements of the pair to issue
    float y0, y1, y2, y3, y4, y5;   // S' elements of the max3 ops
};
#define V4_C ,   // a comma that survives being passed inside a macro argument
// one (EK, XK) combination per asm statement; OUTS / INS are the MFMA's own operand lists (with their trailing comma)
#define V4_ROW(PRE, E, ET, OUTS, INS)                                                                    \
    if constexpr (EK == E && XK == XK_NONE) asm volatile(PRE ET V4_XTXT_0 : OUTS V4_FOUT : INS V4_FIN);   \
    if constexpr (EK == E && XK == XK_M3F) asm volatile(PRE ET V4_XTXT_1 : OUTS V4_FOUT : INS V4_FIN);    \
    if constexpr (EK == E && XK == XK_M3) asm volatile(PRE ET V4_XTXT_2 : OUTS V4_FOUT : INS V4_FIN);     \
    if constexpr (EK == E && XK == XK_M2) asm volatile(PRE ET V4_XTXT_3 : OUTS V4_FOUT : INS V4_FIN);
#define V4_MTXT3 "v_mfma_f32_32x32x16_bf16 %[md], %[ma], %[mb], %[md]"
#define V4_MTXT4 "v_mfma_f32_32x32x16_bf16 %[md], %[ma], %[mb], %[mc]"
// S' chain, first step: D = K Q^T + C with C = the -m block (D and C are different VGPR blocks)
template <int EK, int XK>
DITTO_DEV void slot_s0(f32x16& d, const bf16x8& a, const bf16x8& b, const f32x16& c, V4Fill f) {
    V4_ROW(V4_MTXT4, EK_NONE, V4_ETXT_0, [md] "=&v"(d) V4_C, [ma] "v"(a) V4_C [mb] "a"(b) V4_C [mc] "v"(c) V4_C)
    V4_ROW(V4_MTXT4, EK_E, V4_ETXT_1, [md] "=&v"(d) V4_C, [ma] "v"(a) V4_C [mb] "a"(b) V4_C [mc] "v"(c) V4_C)
    V4_ROW(V4_MTXT4, EK_PE, V4_ETXT_2, [md] "=&v"(d) V4_C, [ma] "v"(a) V4_C [mb] "a"(b) V4_C [mc] "v"(c) V4_C)
    V4_ROW(V4_MTXT4, EK_P, V4_ETXT_3, [md] "=&v"(d) V4_C, [ma] "v"(a) V4_C [mb] "a"(b) V4_C [mc] "v"(c) V4_C)
}
template <int EK, int XK>
DITTO_DEV void slot_s(f32x16& d, const bf16x8& a, const bf16x8& b, V4Fill f) {
    V4_ROW(V4_MTXT3, EK_NONE, V4_ETXT_0, [md] "+v"(d) V4_C, [ma] "v"(a) V4_C [mb] "a"(b) V4_C)
    V4_ROW(V4_MTXT3, EK_E, V4_ETXT_1, [md] "+v"(d) V4_C, [ma] "v"(a) V4_C [mb] "a"(b) V4_C)
    V4_ROW(V4_MTXT3, EK_PE, V4_ETXT_2, [md] "+v"(d) V4_C, [ma] "v"(a) V4_C [mb] "a"(b) V4_C)
    V4_ROW(V4_MTXT3, EK_P, V4_ETXT_3, [md] "+v"(d) V4_C, [ma] "v"(a) V4_C [mb] "a"(b) V4_C)
}
// O^T += V^T P^T  /  l += 1 P^T : accumulators in AGPRs
template <int EK, int XK>
DITTO_DEV void slot_o(f32x16& d, const bf16x8& a, const u32x4& b, V4Fill f) {
    V4_ROW(V4_MTXT3, EK_NONE, V4_ETXT_0, [md] "+a"(d) V4_C, [ma] "v"(a) V4_C [mb] "v"(b) V4_C)
    V4_ROW(V4_MTXT3, EK_E, V4_ETXT_1, [md] "+a"(d) V4_C, [ma] "v"(a) V4_C [mb] "v"(b) V4_C)
    V4_ROW(V4_MTXT3, EK_PE, V4_ETXT_2, [md] "+a"(d) V4_C, [ma] "v"(a) V4_C [mb] "v"(b) V4_C)
    V4_ROW(V4_MTXT3, EK_P, V4_ETXT_3, [md] "+a"(d) V4_C, [ma] "v"(a) V4_C [mb] "v"(b) V4_C)
}
template <int EK, int XK>
DITTO_DEV void slot_l(f32x16& d, const bf16x8& a, const u32x4& b, V4Fill f) {
    V4_ROW(V4_MTXT3, EK_NONE, V4_ETXT_0, [md] "+a"(d) V4_C, [ma] "a"(a) V4_C [mb] "v"(b) V4_C)
    V4_ROW(V4_MTXT3, EK_E, V4_ETXT_1, [md] "+a"(d) V4_C, [ma] "a"(a) V4_C [mb] "v"(b) V4_C)
    V4_ROW(V4_MTXT3, EK_PE, V4_ETXT_2, [md] "+a"(d) V4_C, [ma] "a"(a) V4_C [mb] "v"(b) V4_C)
    V4_ROW(V4_MTXT3, EK_P, V4_ETXT_3, [md] "+a"(d) V4_C, [ma] "a"(a) V4_C [mb] "v"(b) V4_C)
}
// a chunk without an MFMA in front (slots of the first / last iteration whose MFMA does not exist)
template <int EK, int XK>
DITTO_DEV void fill_only(V4Fill f) {
    V4_ROW("; chunk", EK_NONE, V4_ETXT_0, , )
    V4_ROW("; chunk", EK_E, V4_ETXT_1, , )
    V4_ROW("; chunk", EK_PE, V4_ETXT_2, , )
    V4_ROW("; chunk", EK_P, V4_ETXT_3, , )
}
// acc (an element of an AGPR-resident accumulator block) *= f, without the compiler ever seeing a vector-pipe use of the
// block: with `O[i] *= alpha` in the raise branch hipcc made the loop-carried O a VGPR value and copied all of it out of
// and back into the AGPRs in EVERY iteration (48 v_accvgpr_read at the loop head).
DITTO_DEV void scale_acc(float& acc, float f) {
    float t;
    asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 0\n\tv_accvgpr_write_b32 %0, %1" : "+a"(acc), "=&v"(t) : "v"(f));
}
#define V4_FENCE() __builtin_amdgcn_sched_barrier(0)

#ifdef DITTO_DIAG_V4_STAMP   // tools/build_diag.sh: where does an iteration spend its cycles?  (s_memtime stamps; timing build only)
constexpr int V4_STAMP_WAVES = 8192;
__device__ unsigned long long g_v4_stamps[V4_STAMP_WAVES * 16];   // one record per wave: plain stores (atomics on 16 words cost more than the kernel)   // summed over waves: prologue | slots 0..19 | slots 20..39 | wait + barrier | drain | epilogue | iterations | waves
DITTO_DEV unsigned long long v4_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define V4_STAMP(var) const unsigned long long var = v4_now()
#define V4_ACC(i, a, b) st_acc[i] += (b) - (a)
#if DITTO_DIAG_V4_STAMP > 1      // also inside the tile loop (three stamps per iteration: they cost more than they show)
#define V4_STAMP_IN(var) const unsigned long long var = v4_now()
#define V4_ACC_IN(i, a, b) st_acc[i] += (b) - (a)
#else
#define V4_STAMP_IN(var)
#define V4_ACC_IN(i, a, b)
#endif
#else
#define V4_STAMP(var)
#define V4_ACC(i, a, b)
#define V4_STAMP_IN(var)
#define V4_ACC_IN(i, a, b)
#endif

// s_waitcnt vmcnt takes an immediate: the (even) number of this wave's LDS-DMA pieces that may stay in flight
DITTO_DEV void wait_vm_pieces(int y) {
#define V4_VM(n) case n: asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)" ::: "memory"); break;
    switch (y) {
        V4_VM(2) V4_VM(4) V4_VM(6) V4_VM(8) V4_VM(10) V4_VM(12) V4_VM(14) V4_VM(16) V4_VM(18) V4_VM(20) V4_VM(22) V4_VM(24)
        V4_VM(26) V4_VM(28) V4_VM(30) V4_VM(32) V4_VM(34) V4_VM(36) V4_VM(38) V4_VM(40)
        default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    }
#undef V4_VM
}

// LDS-DMA piece with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset; M0 (the LDS destination) is left
// as set: nothing else in this kernel reads it, and with one wave per SIMD every scalar instruction is issue time.
DITTO_DEV void v4_dma(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// NS: tiles per ring (K and V each).  G: iterations per barrier.  In iteration t the tiles K(t + D), V(t + D - 1) are issued,
// D = NS - G, into the ring slots of K(t - G) / V(t - G - 1), which every wave left before the last barrier.
template <bool RESID, int NS, int G>
__global__ __launch_bounds__(256, 1) void attn64v4_kernel(AttnParams p) {
    constexpr int V4_KSLOTS = NS, V4_VSLOTS = NS, D = NS - G;
    constexpr int Y_STEADY = 4 * (D - G - 1);      // pieces that may stay in flight at a barrier while both streams still issue
    static_assert(G >= 1 && D >= G + 1 && Y_STEADY <= 40, "ring depth / barrier period");
    extern __shared__ __attribute__((aligned(16))) char smem[];    // [NS K tiles][NS V tiles]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef DITTO_DIAG_V4_STAMP
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const int nwg = p.nqb * p.H * p.B;
    // gridDim.x == nwg: one item per workgroup.  gridDim.x < nwg (attn_flags bit 18: one workgroup per CU): a workgroup walks
    // the items blockIdx.x, blockIdx.x + gridDim.x, ... one after the other (no re-dispatch between them); round r of the grid
    // covers the ids [r gridDim.x, (r + 1) gridDim.x), XCD-contiguous inside the round.
  for (int item = blockIdx.x; item < nwg; item += gridDim.x) {
    const int round0 = (item / (int)gridDim.x) * (int)gridDim.x;
    const int in_round = nwg - round0 < (int)gridDim.x ? nwg - round0 : (int)gridDim.x;
    const int id = round0 + xcd_remap(item - round0, in_round);
    V4_STAMP(t_begin);
    const int qb = id % p.nqb, bh = id / p.nqb;
    const int h = bh % p.H, b = bh / p.H;
    const int ql = lane & 31, hh = lane >> 5;
    int qrowA = qb * V4_QWG + wid * 64 + ql, qrowB = qrowA + 32;
    const bool validA = qrowA < p.Sq, validB = qrowB < p.Sq;
    qrowA = validA ? qrowA : p.Sq - 1;
    qrowB = validB ? qrowB : p.Sq - 1;

    // ---- Q fragments (B operands of the score MFMAs): 2 blocks x 4 k-steps, AGPR-resident from here on ----
    bf16x8 qA[4], qB[4];
    {
        const bf16* pa = p.q + ((size_t)b * p.Sq + qrowA) * p.ldq + h * ATT_DH + 8 * hh;
        const bf16* pb = p.q + ((size_t)b * p.Sq + qrowB) * p.ldq + h * ATT_DH + 8 * hh;
        // hand-written loads, straight into the AGPRs, the OLDEST vector-memory operations of the wave: the prologue's
        // counted wait for the first K / V tiles retires them too.  (As compiler loads hipcc put its vmcnt waits in front of
        // the loop's first MFMAs, where they drained the LDS-DMA in flight.)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(qA[ks]) : "v"(pa + 16 * ks) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(qB[ks]) : "v"(pb + 16 * ks) : "memory");
        }
    }
    const int nkt = p.Skv / ATT_KBLK;                              // >= 1, whole tiles
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

    // ---- LDS-DMA: a tile image = 8 pieces of 1 KiB (8 rows x 128 B); wave w moves pieces 2w, 2w+1 of K and of V.
    //      LDS swizzles are applied on the SOURCE address (the DMA writes lane-linearly) ----
    //      Source = scalar tile base (advances by 64 rows per tile) + per-lane byte offset (constant for the kernel).
    unsigned kvo[2], vvo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wid * 2 + i) * 8 + (lane >> 3), cpos = lane & 7;
        kvo[i] = (unsigned)((row * p.ldk + (cpos ^ ((row >> 1) & 7)) * 8) * 2);
        vvo[i] = (unsigned)((row * p.ldv + (cpos ^ (((row >> 1) & 1) << 2)) * 8) * 2);
    }
    const char* kbase = reinterpret_cast<const char*>(p.k + (size_t)b * p.Skv * p.ldk + h * ATT_DH);
    const char* vbase = reinterpret_cast<const char*>(p.v + (size_t)b * p.Skv * p.ldv + h * ATT_DH);
    // Tiles are issued in the order K(0) | K(1), V(0) | K(2), V(1) | ...; past the last tile nothing is issued (wave-uniform
    // branches around the asm statements: the slot sequence is asm volatile and keeps its order across them).
    const size_t kstep = (size_t)ATT_KBLK * p.ldk * 2, vstep = (size_t)ATT_KBLK * p.ldv * 2;   // bytes per tile
    int k_issued = 0, v_issued = 0;                               // tiles whose DMA has been issued
    unsigned ik = 0, iv = 0;                                      // ring slots the next K / V tile goes to
    auto dma_k_piece = [&](int i) {
        v4_dma(kvo[i], kbase, lds_base + (unsigned)(ik * ATT_KV_TILE_BYTES + (wid * 2 + i) * 1024));
    };
    auto dma_v_piece = [&](int i) {
        v4_dma(vvo[i], vbase, lds_base + (unsigned)((V4_KSLOTS + iv) * ATT_KV_TILE_BYTES + (wid * 2 + i) * 1024));
    };
    auto k_advance = [&]() { ++k_issued; kbase += kstep; ik = ik + 1 == V4_KSLOTS ? 0 : ik + 1; };
    auto v_advance = [&]() { ++v_issued; vbase += vstep; iv = iv + 1 == V4_VSLOTS ? 0 : iv + 1; };

    // ---- fragment addressing (byte offsets inside a tile image) ----
    // K: lane reads key row ql of a 32-key block, 16-B chunk (2 ks + hh) ^ ((ql >> 1) & 7)
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = ql * 128 + (((2 * ks + hh) ^ ((ql >> 1) & 7)) << 4);
    // V^T by transposed reads (ds_read_b64_tr_b16): two base offsets (head-column halves db), k-step s2 and the second
    // 4-row block are immediates
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int tr_colbyte = (16 * ((lane >> 4) & 1) + 4 * tr_p) * 2;
    const int tr_swz = ((tr_q >> 1) & 1) << 6;
    int voff[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) voff[db] = (4 * hh + tr_q) * 128 + ((tr_colbyte + 64 * db) ^ tr_swz);

    // ---- state ----
    f32x16 oA[2], oB[2], lA, lB;           // AGPR (asm "+a")
    f32x16 sA[2], sB[2], cA, cB;           // VGPR: S' of the tile in flight per block, and -m as a 16-register block
    u32x4 pA[4], pB[4];                    // VGPR: P (bf16 pairs), k-step s2 = registers of one B operand
    bf16x8 kf[8], vf[8];                   // VGPR: K fragments [2 ks + kb2] (the order the score chains take them), V^T fragments [s2 * 2 + db]
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;
    asm volatile("" : "+a"(ones));         // an opaque AGPR resident (as a constant hipcc re-materialised it before every row-sum MFMA)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        oA[0][i] = 0.f; oA[1][i] = 0.f; oB[0][i] = 0.f; oB[1][i] = 0.f; lA[i] = 0.f; lB[i] = 0.f; cA[i] = 0.f; cB[i] = 0.f;
    }
    // The zeros are MATERIALISED HERE, a prologue away from their first MFMA: left to itself hipcc writes an accumulator's
    // initial value right in front of its first use (seen: v_mov_b64 v[50:51] directly followed by the asm MFMA that reads
    // v[50:65] as its C operand), and for an asm MFMA nobody inserts the VALU-write -> MFMA-read wait states: the first two
    // score registers of block A came out as garbage (rows attending to one key only, NaNs).
    asm volatile("s_nop 7" : "+a"(oA[0]), "+a"(oA[1]), "+a"(oB[0]), "+a"(oB[1]), "+a"(lA), "+a"(lB), "+v"(cA), "+v"(cB));
    float dmA[2] = {0.f, 0.f}, dmB[2] = {0.f, 0.f};   // partial row maxima of the newest S' (relative to m), by slot parity
    float eh[2][2] = {{0.f, 0.f}, {0.f, 0.f}};        // the two exponential pairs in flight (issued in slot s, packed in s + 2)

    unsigned kq = 0, vq = 0;               // ring slots of the K tile / V tile the NEXT LDS fragment reads take
    auto read_k = [&](int f) {             // K fragment f in CONSUMPTION order: key block f & 1, k-step f >> 1 (ring slot kq)
        kf[f] = *reinterpret_cast<const bf16x8*>(smem + kq * ATT_KV_TILE_BYTES + (f & 1) * 4096 + koff[f >> 1]);
    };
    auto read_v = [&](int f) {             // V^T fragment f = s2 * 2 + db of the tile in ring slot vq
        const char* a0 = smem + (V4_KSLOTS + vq) * ATT_KV_TILE_BYTES + (f >> 1) * 2048 + voff[f & 1];
        vf[f] = cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0)),
                     __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 8 * 128)));
    };

    // ---- the vector work of one block, addressed by its RELATIVE slot r (r = 0: the slot behind the last MFMA of its
    //      score chains).  r 0, 1: nothing may read S' yet (asm MFMA results: wait states by distance);
    //      r 2..7: row maximum, 16 x v_max3 (3, 3, 3, 3, 2, 2);  r 8: lane exchange + raise decision (compiler code);
    //      r 9..24: exponential pair q = r - 9 issued;  r 10..25: pair q - 1 packed ----
    auto ek_of = [](int r) constexpr { return (r == 9 || r == 10) ? EK_E : (r >= 11 && r <= 24) ? EK_PE : (r == 25 || r == 26) ? EK_P : EK_NONE; };
    auto xk_of = [](int r) constexpr { return (r == 2 || r == 3) ? XK_M3F : (r == 4 || r == 5) ? XK_M3 : (r == 6 || r == 7) ? XK_M2 : XK_NONE; };
    // Operands of a slot's chunk: the exponential part belongs to block (SE, PE) at relative slot RE (-1: none), the
    // maximum part to block SX with partial maxima dmx[2] at relative slot RX (-1: none).  `body(IC4<ek>, IC4<xk>, operands)`.
    auto with_chunk = [&](auto RE_, f32x16 (&SE)[2], u32x4 (&PE)[4], auto RX_, f32x16 (&SX)[2], float (&dmx)[2], auto&& body) {
        constexpr int re = decltype(RE_)::value, rx = decltype(RX_)::value;
        constexpr int ek = ek_of(re), xk = xk_of(rx);
        constexpr int par = (re >= 0 ? re : rx >= 0 ? rx : 0) & 1;
        unsigned t = 0;
        float x0 = eh[par][0], x1 = x0;
        if constexpr (ek == EK_E || ek == EK_PE) {
            constexpr int n = 2 * (re - 9);
            x0 = SE[n >> 4][n & 15]; x1 = SE[(n + 1) >> 4][(n + 1) & 15];
        }
        float y0 = x0, y1 = x0, y2 = x0, y3 = x0, y4 = x0, y5 = x0;
        if constexpr (xk != XK_NONE) {
            constexpr int j = rx <= 5 ? 3 * (rx - 2) : 12 + 2 * (rx - 6);   // first max3 op: op j folds S' elements 2j, 2j+1
            constexpr int e = 2 * j, e4 = xk == XK_M2 ? e : e + 4;          // (XK_M2 has no third op: dummies)
            y0 = SX[e >> 4][e & 15]; y1 = SX[(e + 1) >> 4][(e + 1) & 15]; y2 = SX[(e + 2) >> 4][(e + 2) & 15];
            y3 = SX[(e + 3) >> 4][(e + 3) & 15]; y4 = SX[e4 >> 4][e4 & 15]; y5 = SX[(e4 + 1) >> 4][(e4 + 1) & 15];
        }
        body(IC4<ek>{}, IC4<xk>{}, V4Fill{eh[par][0], eh[par][1], t, dmx[rx >= 0 ? (rx & 1) : 0], x0, x1, y0, y1, y2, y3, y4, y5});
        if constexpr (ek == EK_PE || ek == EK_P) {
            constexpr int q = re - 11;
            PE[q >> 2][q & 3] = t;
        }
    };
    // r == 8: the query's other 16 keys of each block live in lane ^ 32 (v_permlane32_swap: no LDS round trip); raise the
    // running maximum when some row exceeds it by more than the threshold (rare), by whole octaves, as attn64v2
    auto decide = [&](auto FIRSTTILE_, f32x16 (&S)[2], f32x16& C, f32x16 (&O)[2], f32x16& L, float (&dmx)[2]) {
        constexpr bool first_tile = decltype(FIRSTTILE_)::value != 0;
        float dm = fmaxf(dmx[0], dmx[1]);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(dm), __float_as_uint(dm), false, false);
        dm = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        if (first_tile || !__all(dm <= ATT_RESCALE_THR_LOG2)) {
            const float up = ceilf(first_tile ? dm : fmaxf(dm, 0.f));   // first tile: the running maximum IS this tile's
#pragma unroll
            for (int i = 0; i < 16; ++i) { S[0][i] -= up; S[1][i] -= up; C[i] -= up; }
            if constexpr (!first_tile) {
                const float alpha = __builtin_amdgcn_exp2f(-up);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float a0 = O[0][i], a1 = O[1][i], a2 = L[i];
                    scale_acc(a0, alpha); scale_acc(a1, alpha); scale_acc(a2, alpha);
                    O[0][i] = a0; O[1][i] = a1; L[i] = a2;
                }
            }
        }
    };

    // One iteration = tile t of block A and of block B (HAVE_CUR) plus the second half of tile t-1 of block B (HAVE_PREV).
    // FIRST_TILE: t == 0 (the running maxima are this tile's).  Slot s: block A is at r = s - 8, block B at r = s - 28
    // (this iteration's tile) or s + 12 (the previous iteration's).
    auto iteration = [&](auto HAVE_PREV_, auto HAVE_CUR_, auto FIRST_TILE_, auto HAVE_NEXT_K_) {
        constexpr bool have_prev = decltype(HAVE_PREV_)::value != 0, have_cur = decltype(HAVE_CUR_)::value != 0;
        constexpr bool have_next_k = decltype(HAVE_NEXT_K_)::value != 0;   // K(t+1) exists: its fragments are read in slots 28..35
        using FT = decltype(FIRST_TILE_);
        const bool do_k = k_issued < nkt, do_v = v_issued < nkt;           // does tile K(t+NS) / V(t+NS-1) exist?
        // slot s: `mfma(IC4<ek>, IC4<xk>, operands)` issues the slot's MFMA with the slot's chunk behind it in ONE statement
        // (HAS_MFMA_ = 0: the MFMA does not exist in this iteration; the chunk runs alone).  Exponential work: block A in
        // slots 17..34, block B in 37..39 and 0..14; maximum work: A in 10..15, B in 30..35 — at most one of each per slot.
        auto slot = [&](auto S_, auto HAS_MFMA_, auto&& mfma) {
            constexpr int s = decltype(S_)::value;
            constexpr bool has_mfma = decltype(HAS_MFMA_)::value != 0;
            constexpr int rA = (have_cur && s >= 8) ? s - 8 : -1;
            constexpr int rB = (have_cur && s >= 28) ? s - 28 : ((have_prev && s + 12 <= 26) ? s + 12 : -1);
            constexpr bool eA = ek_of(rA) != EK_NONE, eB = ek_of(rB) != EK_NONE, xA = xk_of(rA) != XK_NONE, xB = xk_of(rB) != XK_NONE;
            static_assert(!(eA && eB) && !(xA && xB), "two blocks claim the same part of a slot");
            constexpr int re = eA ? rA : eB ? rB : -1, rx = xA ? rA : xB ? rB : -1;
            auto alone = [&](auto EKc, auto XKc, V4Fill f) { fill_only<decltype(EKc)::value, decltype(XKc)::value>(f); };
            auto go = [&](auto&& body) {
                if constexpr (eB) {
                    if constexpr (xA) with_chunk(IC4<re>{}, sB, pB, IC4<rx>{}, sA, dmA, body); else with_chunk(IC4<re>{}, sB, pB, IC4<rx>{}, sB, dmB, body);
                } else {
                    if constexpr (xA) with_chunk(IC4<re>{}, sA, pA, IC4<rx>{}, sA, dmA, body); else with_chunk(IC4<re>{}, sA, pA, IC4<rx>{}, sB, dmB, body);
                }
            };
            if constexpr (has_mfma) go(mfma);
            else if constexpr (re >= 0 || rx >= 0) go(alone);
            if constexpr (rA == 8) decide(FT{}, sA, cA, oA, lA, dmA);
            if constexpr (rB == 8) {
                if constexpr (s >= 28) decide(FT{}, sB, cB, oB, lB, dmB); else decide(IC4<0>{}, sB, cB, oB, lB, dmB);
            }
            // group g(t) = { K(t+NS), V(t+NS-1) }: one LDS-DMA piece behind each of the first four slots
            if constexpr (have_cur) {
                if constexpr (s == 0) { if (do_k) dma_k_piece(0); }
                if constexpr (s == 1) { if (do_k) { dma_k_piece(1); k_advance(); } }
                if constexpr (s == 2) { if (do_v) dma_v_piece(0); }
                if constexpr (s == 3) { if (do_v) { dma_v_piece(1); v_advance(); } }
            }
            V4_FENCE();
        };
        // ---- slots 0..7: S'_A(t) = K(t) Q_A^T - m_A, the two key blocks' chains alternating ----
        auto sa = [&](auto J_) {
            constexpr int j = decltype(J_)::value;
            slot(IC4<j>{}, IC4<have_cur>{}, [&](auto EKc, auto XKc, V4Fill f) {
                constexpr int ek = decltype(EKc)::value, xk = decltype(XKc)::value;
                if constexpr ((j >> 1) == 0) slot_s0<ek, xk>(sA[j & 1], kf[j], qA[0], cA, f);
                else slot_s<ek, xk>(sA[j & 1], kf[j], qA[j >> 1], f);
            });
        };
        V4_STAMP_IN(t_it0);
        sa(IC4<0>{}); sa(IC4<1>{}); sa(IC4<2>{}); sa(IC4<3>{}); sa(IC4<4>{}); sa(IC4<5>{}); sa(IC4<6>{}); sa(IC4<7>{});
        // ---- slots 8..19: O_B, l_B += V(t-1) P_B(t-1) ----
        // (iteration 0 has no MFMA here: the distance that lets the vector pipe read S'_A is then made of s_nops)
        if constexpr (!have_prev && have_cur) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        auto pvb = [&](auto I_) {
            constexpr int i = decltype(I_)::value, s2 = i / 3, w = i % 3;
            slot(IC4<8 + i>{}, IC4<have_prev>{}, [&](auto EKc, auto XKc, V4Fill f) {
                constexpr int ek = decltype(EKc)::value, xk = decltype(XKc)::value;
                if constexpr (w < 2) slot_o<ek, xk>(oB[w], vf[s2 * 2 + w], pB[s2], f);
                else slot_l<ek, xk>(lB, ones, pB[s2], f);
            });
        };
        pvb(IC4<0>{}); pvb(IC4<1>{}); pvb(IC4<2>{}); pvb(IC4<3>{}); pvb(IC4<4>{}); pvb(IC4<5>{});
        pvb(IC4<6>{}); pvb(IC4<7>{}); pvb(IC4<8>{}); pvb(IC4<9>{}); pvb(IC4<10>{}); pvb(IC4<11>{});
        V4_STAMP_IN(t_it1);
        // ---- slots 20..27: S'_B(t) = K(t) Q_B^T - m_B; V(t) fragments from LDS (their registers were released by slot 18) ----
        auto sb = [&](auto J_) {
            constexpr int j = decltype(J_)::value;
            slot(IC4<20 + j>{}, IC4<have_cur>{}, [&](auto EKc, auto XKc, V4Fill f) {
                constexpr int ek = decltype(EKc)::value, xk = decltype(XKc)::value;
                if constexpr ((j >> 1) == 0) slot_s0<ek, xk>(sB[j & 1], kf[j], qB[0], cB, f);
                else slot_s<ek, xk>(sB[j & 1], kf[j], qB[j >> 1], f);
            });
            if constexpr (have_cur) { read_v(j); V4_FENCE(); }
        };
        sb(IC4<0>{}); sb(IC4<1>{}); sb(IC4<2>{}); sb(IC4<3>{}); sb(IC4<4>{}); sb(IC4<5>{}); sb(IC4<6>{}); sb(IC4<7>{});
        if constexpr (have_cur) { vq = vq + 1 == V4_VSLOTS ? 0 : vq + 1; }
        // ---- slots 28..39: O_A, l_A += V(t) P_A(t); K(t+1) fragments from LDS (released by slot 27) ----
        auto pva = [&](auto I_) {
            constexpr int i = decltype(I_)::value, s2 = i / 3, w = i % 3;
            slot(IC4<28 + i>{}, IC4<have_cur>{}, [&](auto EKc, auto XKc, V4Fill f) {
                constexpr int ek = decltype(EKc)::value, xk = decltype(XKc)::value;
                if constexpr (w < 2) slot_o<ek, xk>(oA[w], vf[s2 * 2 + w], pA[s2], f);
                else slot_l<ek, xk>(lA, ones, pA[s2], f);
            });
            if constexpr (have_cur && have_next_k && i < 8) { read_k(i); V4_FENCE(); }
        };
        pva(IC4<0>{}); pva(IC4<1>{}); pva(IC4<2>{}); pva(IC4<3>{}); pva(IC4<4>{}); pva(IC4<5>{});
        pva(IC4<6>{}); pva(IC4<7>{}); pva(IC4<8>{}); pva(IC4<9>{}); pva(IC4<10>{}); pva(IC4<11>{});
        if constexpr (have_cur && have_next_k) { kq = kq + 1 == V4_KSLOTS ? 0 : kq + 1; }
        V4_STAMP_IN(t_it2);
        if constexpr (have_cur && have_prev) { V4_ACC_IN(1, t_it0, t_it1); V4_ACC_IN(2, t_it1, t_it2); V4_ACC_IN(6, 0ull, 1ull); }
        if constexpr (!have_cur) V4_ACC_IN(4, t_it0, t_it2);
    };
    // A barrier ends every G-th iteration.  At the barrier behind iteration t everything the next G iterations read from LDS —
    // K(<= t+G+1) and V(<= t+G) — has landed for this wave once only the pieces issued after them are in flight (tiles go out
    // in the order ... K(j+1), V(j), K(j+2), V(j+1) ...: 2 pieces per tile); then, by the barrier, for every wave.  The fragment
    // reads of this iteration are retired first: the barrier also releases ring slots to the DMA of the next G iterations.
    auto end_of_iteration = [&](int t) {
        if (G > 1 && (t + 1) % G != 0) return;
        V4_STAMP_IN(t_w0);
        const int yk = k_issued - (t + G + 2), yv = v_issued - (t + G + 1);
        const int y = 2 * (yk > 0 ? yk : 0) + 2 * (yv > 0 ? yv : 0);
        if (y == Y_STEADY) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(Y_STEADY) : "memory");
        else wait_vm_pieces(y);
        asm volatile("s_barrier" ::: "memory");
        V4_STAMP_IN(t_w1);
        if (t > 0) V4_ACC_IN(3, t_w0, t_w1);
    };

    // ---- prologue: K(0) | K(1), V(0) | ... | K(D-1), V(D-2) in flight; K(0) fragments ----
    using T = IC4<1>; using F = IC4<0>;
    V4_STAMP(t_p0);                    // [8] setup + Q loads issued
    V4_ACC(8, t_begin, t_p0);
    dma_k_piece(0); dma_k_piece(1); k_advance();
#pragma unroll
    for (int j = 0; j + 1 < D; ++j) {
        if (k_issued < nkt) { dma_k_piece(0); dma_k_piece(1); k_advance(); }
        if (v_issued < nkt) { dma_v_piece(0); dma_v_piece(1); v_advance(); }
    }
    V4_STAMP(t_p1);                    // [9] prologue DMA issued
    V4_ACC(9, t_p0, t_p1);
    {                                  // Q (the oldest loads), K(<= G), V(<= G-1) have landed: what iterations 0 .. G-1 read
        const int yk = k_issued - (G + 1), yv = v_issued - G;
        wait_vm_pieces(2 * (yk > 0 ? yk : 0) + 2 * (yv > 0 ? yv : 0));
        asm volatile("s_barrier" ::: "memory");
    }
    V4_STAMP(t_p2);                    // [10] first tiles landed + barrier
    V4_ACC(10, t_p1, t_p2);
    asm volatile("" : "+a"(qA[0]), "+a"(qA[1]), "+a"(qA[2]), "+a"(qA[3]), "+a"(qB[0]), "+a"(qB[1]), "+a"(qB[2]), "+a"(qB[3]));
#pragma unroll
    for (int f = 0; f < 8; ++f) read_k(f);     // (the DMA of iteration 0 goes to ring slot D != 0: K(0) may be read at leisure)
    kq = 1;

    V4_STAMP(t_loop);
    V4_ACC(0, t_begin, t_loop);
    // ---- t = 0 ----
    if (nkt > 1) { iteration(F{}, T{}, T{}, T{}); end_of_iteration(0); }
    else iteration(F{}, T{}, T{}, F{});
    // ---- t = 1 .. nkt-2 ----
    for (int t = 1; t + 1 < nkt; ++t) {
        iteration(T{}, T{}, F{}, T{});
        end_of_iteration(t);
    }
    // ---- t = nkt-1 (no K(t+1) to fetch; the drain reads nothing from LDS: no barrier behind it) ----
    if (nkt > 1) iteration(T{}, T{}, F{}, F{});
    // ---- drain: the second half of block B's last tile ----
    iteration(T{}, F{}, F{}, F{});
    // every LDS-DMA piece of this wave (the re-fetched tail pieces included) has landed before the wave may end: a piece
    // landing later would write into LDS that already belongs to the next workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    V4_STAMP(t_epi);
    V4_ACC(12, t_loop, t_epi);        // [12] the whole tile loop
    // ---------------- epilogue ----------------
    // MFMA results (asm producers) -> vector readers: wait states hipcc does not insert
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(oA[0]), "+a"(oA[1]), "+a"(oB[0]), "+a"(oB[1]), "+a"(lA), "+a"(lB));
    auto store_block = [&](f32x16 (&O)[2], f32x16& L, int qrow, bool valid) {
        const float inv = 1.0f / L[0];
        if (!valid) return;
        const size_t grow = (size_t)b * p.Sq + qrow;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = h * ATT_DH + 32 * db + 8 * g + 4 * hh;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = O[db][4 * g + e] * inv;
                if constexpr (RESID) {
                    float* rp = p.resid + grow * p.ldr + col;
                    f32x4 r = *reinterpret_cast<const f32x4*>(p.resid_in + grow * p.ldr + col);
                    r += o;
                    *reinterpret_cast<f32x4*>(rp) = r;
                } else {
                    u32x2 st2;
                    st2[0] = pack_bf16x2(o[0], o[1]);
                    st2[1] = pack_bf16x2(o[2], o[3]);
                    *reinterpret_cast<u32x2*>(p.out + grow * p.ldo + col) = st2;
                }
            }
    };
    store_block(oA, lA, qrowA, validA);
    store_block(oB, lB, qrowB, validB);
    if (item + (int)gridDim.x < nwg) asm volatile("s_barrier" ::: "memory");   // every wave is out of this item's LDS before the next item's DMA
  }
#ifdef DITTO_DIAG_V4_STAMP
    V4_STAMP(t_st);                    // [11] epilogue arithmetic + store issue
    V4_ACC(11, t_epi, t_st);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    V4_STAMP(t_end);
    V4_ACC(5, t_epi, t_end);
    if (lane == 0)
        for (int i = 0; i < 16; ++i) g_v4_stamps[(size_t)((blockIdx.x * 4 + wid) % V4_STAMP_WAVES) * 16 + i] = st_acc[i];
#endif
}

}  // namespace

#ifdef DITTO_DIAG_V4_STAMP
extern "C" int ditto_diag_v4_stamps(unsigned long long* out) {   // sum of the per-wave records of the LAST launches, then clear (diagnostic build only)
    static unsigned long long host[V4_STAMP_WAVES * 16];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_v4_stamps), sizeof(host)) != hipSuccess) return 1;
    for (int i = 0; i < 16; ++i) out[i] = 0;
    for (int w = 0; w < V4_STAMP_WAVES; ++w)
        for (int i = 0; i < 16; ++i) out[i] += host[(size_t)w * 16 + i];
    for (size_t i = 0; i < sizeof(host) / sizeof(host[0]); ++i) host[i] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_v4_stamps), host, sizeof(host)) != hipSuccess;
}
#endif

bool attn64v4_supports(const AttnParams& p) {
    return p.Skv >= ATT_KBLK && p.Skv % ATT_KBLK == 0 && p.Sq >= 1 && !p.lse && !p.drop_thr;
}

namespace {
template <int NS, int G>
hipError_t launch_v4_t(const AttnParams& p, bool resid, hipStream_t s) {
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&attn64v4_kernel<true, NS, G>),
                                                   reinterpret_cast<const void*>(&attn64v4_kernel<false, NS, G>)}, v4_lds_bytes(NS)))
        return e;
    int nwg = p.nqb * p.H * p.B;
    if ((g_attn_flags & (1 << 18)) && nwg > 256) nwg = 256;  // A/B (bit 18): persistent workgroups, one per CU
    const dim3 grid(nwg);
    if (resid) hipLaunchKernelGGL((attn64v4_kernel<true, NS, G>), grid, dim3(256), v4_lds_bytes(NS), s, p);
    else hipLaunchKernelGGL((attn64v4_kernel<false, NS, G>), grid, dim3(256), v4_lds_bytes(NS), s, p);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_attn64v4(const AttnParams& p_in, bool resid, hipStream_t s) {
    AttnParams p = p_in;
    p.nqb = (p.Sq + V4_QWG - 1) / V4_QWG;
    // attn_flags bits 16-17 (A/B; moved off bits 13-14 in round 4: bit 13 = 8192 is attention.hip's "older training forward"
    // switch): ring depth / barrier period: 0 -> 8 tiles, a barrier per iteration; 1 -> 8 tiles, a barrier per 2 iterations;
    // 2 -> 4 tiles, per iteration; 3 -> 10 tiles, per 3 iterations
    switch ((g_attn_flags >> 16) & 3) {
        case 1: return launch_v4_t<8, 2>(p, resid, s);
        case 2: return launch_v4_t<4, 1>(p, resid, s);
        case 3: return launch_v4_t<10, 3>(p, resid, s);
        default: return launch_v4_t<8, 1>(p, resid, s);
    }
}

}  // namespace ditto
