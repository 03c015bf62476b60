// experimental/gemm_fr128.hip (gemm_fr.hip of rounds 2-4) — FULL-ROW bf16 MFMA GEMM for the N = d = 768 projections of a DiT block, with the fp32 residual add
// AND the next LayerNorm fused into its epilogue (gfx950).
//
//     h[M, 768] (fp32, in place) = residual + A[M, K] * W[768, K]^T + bias            (reference DiT.py:148, :155)
//     u[M, 768] (bf16)           = LayerNorm(h) * gamma + beta   (eps 1e-5)           (reference DiT.py:152, :105 of
//                                                                                      the next block)
//
// Why: a workgroup that owns WHOLE rows of the residual stream can normalise them while they are still in registers.
// The separate LayerNorm launch (28 us: 100 MB of fp32 h read back + 50 MB written, at its HBM roofline) disappears for
// the two LayerNorms that follow a GEMM (24 of the 36 per step), and the out-projection stops re-reading its A panel
// once per column tile.
//
//   tile      128 rows x 768 columns (all of N), one workgroup per tile (one per CU at M = 32768).  256 threads = 4
//             waves, ONE wave per SIMD with the whole 512-entry register file (launch_bounds(256, 1)): wave (wm, wn)
//             owns rows [64 wm, +64) x columns [384 wn, +384) = 2 x 12 blocks of v_mfma_f32_32x32x16_bf16 = 384
//             accumulator registers.  hipcc cannot be trusted with that many (given them as values it shuttled
//             accumulators between the two halves of the file: 1 591 v_accvgpr moves and 540 scratch accesses per 96
//             MFMAs), so every MFMA is an asm statement whose tied operand fixes the home: column blocks 0..6 in
//             AGPRs (224), 7..11 in VGPRs (160).
//   operands  W arrives PACKED stage-major, Wp[K/16][768][16] (launch_pack_bf16_stage_major, once per weight load): a
//             K = 16 stage is 24 contiguous KiB, every 1-KiB DMA piece a run of whole cache lines.  A is the previous
//             kernel's row-major output and comes in SLABS of 64 k (one whole 128-B line per row, four stages of work).
//             (Both read as 32-B row pieces from row-major images, the L2 -> CU path moved 4x the payload: 2.6 us per
//             K = 32; W packed: 1.52; A in slabs: 1.31; tiles remapped to XCD-contiguous runs + K-loop rotation: 1.10.
//             The MFMAs need 0.64, the loop without its DMA runs at 0.85.)
//   LDS       W ring of FIVE stages x 24 KiB, LDS-DMA three stages ahead behind a counted vmcnt; two A slabs x 16 KiB,
//             the next slab's pieces issued during the first two stages of the current one; 2 KiB of row statistics;
//             the bias row (3 KiB).  157 KiB.  The K = 16 instruction is what makes five stages fit: with 16x16x32 MFMAs
//             (K = 32) the same LDS holds two stages, one stage of look-ahead, and the first version ran at 1.93 us per
//             K = 32.  W: 16-B chunk h (0 / 1) of 32-B row r sits at h ^ ((r >> 3) & 1); A: chunk c of 128-B row r at
//             c ^ ((r >> 1) & 7): conflict-free ds_read_b128 both.
//   stage s   { nb = 0..5: 2 MFMAs + one W piece of stage s+3 each ; nb = 6, 7: + one piece of the next A slab (first two
//               stages of a slab) ; nb = 8: counted vmcnt + s_barrier (stage s+1 has landed for everyone, and everyone is
//               past stage s-1), prefetch stage s+1's A fragments ; nb = 9..11: prefetch stage s+1's first W fragments }
//             W fragments run 3 ahead in a 4-register ring, A fragments ping-pong between two named sets: one barrier
//             per stage, four stages (one slab) per loop iteration.
//   init      the accumulators START as residual + bias: the fp32 residual tile is loaded straight into registers in the
//             accumulator layout (96 hand-written global_load_dwordx4 per lane, a window of 24 in flight), behind the
//             ring's prologue DMA.  The epilogue then only READS the accumulators.
//   epilogue  no compiler-visible global load (each would make hipcc wait for every DMA in flight): gamma / beta are
//             DMA'd into the idle A ring; row sums -> lane l ^ 32 -> the other column half through LDS; mean; the same
//             for sum (v - mean)^2 (two-pass statistics, like nn.LayerNorm); every output row leaves through a wave-private
//             LDS stage so that the stores are whole 128-B lines: h fp32 (nt), u = LN(h) bf16.
//   measured  (M = 32768, Infinity Cache flushed) K = 768: 97 us against 132 for GEMM + LayerNorm; K = 3072: 178 against
//             240.  Of the 97: main loop 26, residual read 28 (100 MB: HBM alone 13-17), stores 13, LayerNorm 8, the rest
//             launch / prologue / epilogue arithmetic.  All 256 workgroups run in lockstep (one tile each, one wave of
//             workgroups), so the residual read, the MFMA loop and the 150 MB of stores do NOT overlap one another: that
//             is the structural cost of owning whole rows with one wave per SIMD (staggering every other workgroup's
//             start by 4-16 us gained 2 us at best).
#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int FM = 128, FN = 768, FK = 16, NST = 5;
constexpr int F_W_BYTES = FN * FK * 2;             // 24 KiB: one K = 16 stage of W (stage-major packed: contiguous)
constexpr int F_WRING = NST * F_W_BYTES;           // 120 KiB
constexpr int F_ASLAB = FM * 64 * 2;               // 16 KiB: 128 rows x 64 k of A = FOUR stages, one whole 128-B line per row
constexpr int F_ARING = F_WRING;                   // two slabs
constexpr int F_RED = F_ARING + 2 * F_ASLAB;       // row statistics [2 passes][2 column halves][128 rows] fp32
constexpr int F_BIAS = F_RED + 2 * 2 * FM * 4;     // bias row (read while the accumulators are initialised)
constexpr int F_LDS = F_BIAS + FN * 4;             // 157 KiB
constexpr int F_GB = F_ARING;                      // gamma | beta rows during the epilogue (the A ring is idle by then)
constexpr int NA = 7;                              // column blocks whose accumulators live in AGPRs

#ifdef DITTO_DIAG_FR_NOSTORE    // tools/build_diag.sh: epilogue without its global stores (timing only)
#define FR_DIAG_M (p.M - (1 << 30))
#else
#define FR_DIAG_M p.M
#endif
#define FR_BAR() asm volatile("s_barrier" ::: "memory")
#define PIN_A(x) asm volatile("" : "+a"(x))
#define PIN_V(x) asm volatile("" : "+v"(x))

template <int V>
struct IC { static constexpr int value = V; };

DITTO_DEV void mfma_a(f32x16& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
}
DITTO_DEV void mfma_v(f32x16& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(a));
}
// Hazard argument for the asm MFMAs.  hipcc inserts the wait states an MFMA result needs before a NON-MFMA reader only for
// MFMAs it emitted itself; behind an asm statement it pads nothing.  (a) Inside an accumulator chain the only reader of a
// result is the next MFMA of the same chain (same vdst = srcC): back-to-back dependent MFMAs of one shape need no software
// wait states (the hardware interlocks srcC).  (b) Operand registers (wf / a fragments) are only READ by the MFMA and
// rewritten by ds_reads, whose data returns long after the MFMA has fetched its operands.  (c) The LAST MFMA of each chain
// is followed by arbitrary compiler-scheduled readers (v_accvgpr_read, a spill's scratch_store, the LayerNorm adds): an
// 8-pass 32x32x16 MFMA needs 11 wait states before any of them.  A separate `s_nop` STATEMENT is not enough: in
// gemm_fr64.hip hipcc placed a spill of the just-written block BETWEEN the MFMA statement and the s_nop statement that
// followed the pair (wrong lanes in u, caught by the bitwise test).  So the last-stage MFMAs carry `s_nop 15` (16 wait
// states) inside their own asm statement; the nop is hidden under the 32 cycles the matrix pipe is busy anyway.
DITTO_DEV void mfma_a_last(f32x16& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+a"(c) : "v"(w), "v"(a));
}
DITTO_DEV void mfma_v_last(f32x16& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+v"(c) : "v"(w), "v"(a));
}

template <bool LN, bool RES>
__global__ __launch_bounds__(256, 1) void gemm_fr_kernel(FrParams fp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemmParams& p = fp.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int nkt = p.K / FK;                  // a multiple of 4 (K % 64 == 0)
    // Workgroups go to the XCDs round-robin; tile = consecutive runs per XCD, so that the 32 workgroups sharing an L2 hold
    // NEIGHBOURING tiles (all eight K-loop rotations below, and adjacent residual / output rows).
    const int ntile = gridDim.x;
    const int tile = (ntile & 7) == 0 ? (int)(blockIdx.x & 7) * (ntile >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int m0 = tile * FM;
    // K-loop ROTATION.  Every workgroup reads the whole of W, and unrotated all 32 workgroups of an XCD ask their L2 for the
    // SAME 24 KiB at the same time: a handful of L2 channels serve everything while the others idle (measured: 1.31 us per
    // K = 32 where the MFMAs need 0.64).  Tile t starts at slab s0(t) and wraps, so that eight different stretches of W are
    // in demand at any time.  The sum over k is the same set of products in a rotated ORDER: a row's bits depend on its
    // tile's rotation, so the rotation is a function of (t mod rot_period) only and the caller passes the number of tiles
    // per utterance: an utterance's rows are then computed identically wherever it sits in the batch (the sharded sampler
    // relies on that).  rot_period = 0: no rotation.
    const int nslab = nkt >> 2;
    const int s0 = fp.rot_period > 0 ? (((tile % fp.rot_period) & 7) * nslab) >> 3 : 0;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;

    // ---- operand DMA.  W: stage-major packed Wp[K/16][768][16] (launch_pack_bf16_stage_major), a stage = 24 contiguous
    //      pieces of 1 KiB (32 rows x 32 B), 6 per wave, into a five-slot ring.  (Read from the row-major [768][K] image, a
    //      piece touched 32 lines for 32 B each and the L2 -> CU path moved 4x the payload.)  A is row-major [M, K] and cannot
    //      be packed (it is the previous kernel's output), so it comes in SLABS of 64 k: 128 B = one whole line per row,
    //      four stages of work, 16 pieces of 8 rows x 128 B, 4 per wave per slab, two slabs double-buffered; the next
    //      slab's pieces go out during the first two stages of the current one.  Sources are (loop-invariant scalar base) +
    //      (per-lane offset that advances per stage / slab); the destination goes straight into M0, which is not restored:
    //      nothing else in this kernel reads it (with one wave per SIMD every scalar instruction is issue time). ----
    const int prow = lane >> 1, ppos = lane & 1;
    const int pc = ppos ^ ((prow >> 3) & 1);                     // W: source chunk landing at position ppos of row prow
    unsigned vwk = (unsigned)(prow * 32 + pc * 16 + s0 * 4 * F_W_BYTES);   // W pieces: per-lane byte offset (+ 24 KiB per stage)
    int w_left = nkt - 4 * s0, a_left = nslab - s0;               // stages / slabs until the rotated K loop wraps to k = 0
    const char* wbase[6];                                         // wave-uniform
#pragma unroll
    for (int i = 0; i < 6; ++i) wbase[i] = (const char*)p.W + (size_t)(wid * 6 + i) * 1024;
    const int arow = lane >> 3, apos = lane & 7;                  // A piece: 8 rows x 8 chunks of 16 B
    unsigned vak[4];                                              // per-lane byte offsets of this wave's 4 slab pieces
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (wid * 4 + j) + arow;                 // row inside the tile
        int ar = m0 + row;
        ar = ar < p.M ? ar : p.M - 1;
#ifdef DITTO_DIAG_FR_AHOT      // tools/build_diag.sh: every tile reads the A rows of tile 0 (L2-resident, statistically the same data): what does A's HBM / Infinity Cache latency cost?
        ar = row;
#endif
        vak[j] = (unsigned)(((size_t)ar * p.lda + (apos ^ ((row >> 1) & 7)) * 8) * 2) + (unsigned)(s0 * 128);   // chunk c of row r sits at c ^ ((r >> 1) & 7)
    }
    unsigned w_slot = lds_base;                                   // LDS byte address of the W ring slot the next stage goes to
    unsigned a_buf = lds_base + F_ARING;                          // ... and of the A buffer the next slab goes to
    auto dma = [&](unsigned voff, const char* base, unsigned dst) {
#ifndef DITTO_DIAG_FR_NODMA     // tools/build_diag.sh: main loop without its global -> LDS traffic (timing only, wrong results)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(dst) : "memory");
#endif
    };
    auto issue_w_piece = [&](auto I) {                            // piece I (0..5) of the W stage at the issue cursor
        constexpr int i = decltype(I)::value;
        const char* base = wbase[i];
        dma(vwk, base, w_slot + (unsigned)((wid * 6 + i) * 1024));
    };
    auto advance_w = [&]() {
        w_slot = w_slot + F_W_BYTES == lds_base + F_WRING ? lds_base : w_slot + F_W_BYTES;
        --w_left;
        vwk += w_left == 0 ? (unsigned)F_W_BYTES - (unsigned)nkt * F_W_BYTES : (unsigned)F_W_BYTES;
    };
    auto issue_a_piece = [&](auto J) {                            // piece J (0..3) of the A slab at the issue cursor
        constexpr int j = decltype(J)::value;
        const unsigned voff = vak[j];
        dma(voff, (const char*)p.A, a_buf + (unsigned)((wid * 4 + j) * 1024));
    };
    auto advance_a = [&]() {
        a_buf = a_buf == lds_base + F_ARING ? lds_base + F_ARING + F_ASLAB : lds_base + F_ARING;
        --a_left;
        const unsigned inc = a_left == 0 ? 128u - (unsigned)nslab * 128u : 128u;
#pragma unroll
        for (int j = 0; j < 4; ++j) vak[j] += inc;
    };
    auto issue_w_stage = [&]() {
        issue_w_piece(IC<0>{}); issue_w_piece(IC<1>{}); issue_w_piece(IC<2>{});
        issue_w_piece(IC<3>{}); issue_w_piece(IC<4>{}); issue_w_piece(IC<5>{});
        advance_w();
    };

    // bias row -> LDS (3 pieces of 1 KiB = 768 fp32): the oldest loads of the kernel
    if (wid == 0) {
        if (p.bias) {
#pragma unroll
            for (int i = 0; i < 3; ++i) glds16(p.bias + i * 256 + lane * 4, lds_base + (unsigned)(F_BIAS + i * 1024));
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(smem + F_BIAS + i * 1024 + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // ---- fragment addressing: lane reads row (lane & 31) of a 32-row block, 16-B half (lane >> 5) ----
    const int r32 = lane & 31, hh = lane >> 5;
    const int fpos = (hh ^ ((r32 >> 3) & 1)) << 4;
    const int w_off = (wn * 384 + r32) * 32 + fpos;                         // in a W slot: + nb * 1024
    // A slab: row (64 wm + 32 mb + r32) x 128 B; stage j of the slab = 16-B chunks 2j + hh, sitting at (2j + hh) ^ ((row >> 1) & 7)
    const int a_row = (wm * 64 + r32) * 128;                                // + mb * 4096   (32 and 64 drop out of the swizzle)
    const int a_x = (hh ^ ((r32 >> 1) & 7)) << 4;                           // ^ (j << 5)

    // ---- prologue DMA, in this order: A slab 0 (4 pieces per wave), W stages 0, 1, 2 (6 pieces per wave each): in flight
    //      while the accumulators are initialised ----
    issue_a_piece(IC<0>{}); issue_a_piece(IC<1>{}); issue_a_piece(IC<2>{}); issue_a_piece(IC<3>{});
    advance_a();
    issue_w_stage(); issue_w_stage(); issue_w_stage();

    // ---- the accumulators START as bias + residual, so the epilogue only READS them.  (Adding the residual afterwards
    //      meant writing 384 updated values back into their AGPR / VGPR homes, which hipcc turned into a scratch copy of
    //      every block and reloads behind vmcnt(0) in the later passes: a 170 us epilogue.)  The fp32 residual tile is read
    //      straight into registers in the accumulator layout: a lane owns 4 consecutive columns (16 B) at 4 places of a row
    //      per block, so a load instruction touches 32 rows x 32 B and the four loads of a block complete its lines in L2.
    //      96 loads per lane, a window of three column blocks (24 loads = 96 KiB per CU) in flight; no LDS, no barriers.
    //      (The first version staged the tile through LDS in six double-buffered 64 KiB chunks behind two barriers each:
    //      24 us for a 100 MB read whose HBM time is 12.6 us.)  Hand-written loads: hipcc does not count them, the waits
    //      are explicit and tie the registers they cover so that no use moves above them. ----
    const float* lbias = reinterpret_cast<const float*>(smem + F_BIAS);
    const float* lgamma = reinterpret_cast<const float*>(smem + F_GB);     // these two: valid in the epilogue only
    const float* lbeta = lgamma + FN;
    f32x16 acca[NA][2], accv[12 - NA][2];
    {
        const float* rp[2] = {nullptr, nullptr};
        if constexpr (RES) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                int gr = m0 + wm * 64 + mb * 32 + r32;
                gr = gr < p.M ? gr : p.M - 1;
                rp[mb] = p.residual + (size_t)gr * p.ldr + wn * 384 + 4 * hh;
            }
        }
        f32x4 T[3][8];                                               // [window slot][mb * 4 + g]
        auto issue_group = [&](auto NB, f32x4 (&t)[8]) {             // the 8 loads of column block NB
            constexpr int nb = decltype(NB)::value;
            if constexpr (RES) {
#pragma unroll
#ifdef DITTO_DIAG_FR_RESLINE   // tools/build_diag.sh: the residual read as whole 128-B lines per 8 lanes (WRONG mapping: timing)
                for (int k = 0; k < 8; ++k) {
                    int gr = m0 + wm * 64 + k * 8 + (lane >> 3);
                    gr = gr < p.M ? gr : p.M - 1;
                    const float* ptr = p.residual + (size_t)gr * p.ldr + wn * 384 + (lane & 7) * 4;
                    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(t[k]) : "v"(ptr), "n"(nb * 128) : "memory");
                }
                if (false)
#endif
                for (int mb = 0; mb < 2; ++mb) {
                    const float* ptr = rp[mb];
                    asm volatile("global_load_dwordx4 %0, %4, off offset:%5\n\t"
                                 "global_load_dwordx4 %1, %4, off offset:%6\n\t"
                                 "global_load_dwordx4 %2, %4, off offset:%7\n\t"
                                 "global_load_dwordx4 %3, %4, off offset:%8"
                                 : "=&v"(t[mb * 4 + 0]), "=&v"(t[mb * 4 + 1]), "=&v"(t[mb * 4 + 2]), "=&v"(t[mb * 4 + 3])
                                 : "v"(ptr), "n"(nb * 128), "n"(nb * 128 + 32), "n"(nb * 128 + 64), "n"(nb * 128 + 96)
                                 : "memory");
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto finish_group = [&](auto NB, auto INFLIGHT, f32x4 (&t)[8]) {
            constexpr int nb = decltype(NB)::value, inflight = decltype(INFLIGHT)::value;
            if constexpr (RES) {
                if constexpr (inflight == 16)
                    asm volatile("s_waitcnt vmcnt(16)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7])::"memory");
                else if constexpr (inflight == 8)
                    asm volatile("s_waitcnt vmcnt(8)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7])::"memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7])::"memory");
            }
            if constexpr (nb == 0) {
                if constexpr (!RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                FR_BAR();      // every wave is past a wait that covers wave 0's bias row (the oldest load): visible to all
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                f32x16 v;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbias + wn * 384 + nb * 32 + 8 * g + 4 * hh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * g + e] = t[mb * 4 + g][e] + b4[e];
                }
                if constexpr (nb < NA) { acca[nb < NA ? nb : 0][mb] = v; PIN_A(acca[nb < NA ? nb : 0][mb]); }
                else { accv[nb < NA ? 0 : nb - NA][mb] = v; PIN_V(accv[nb < NA ? 0 : nb - NA][mb]); }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        issue_group(IC<0>{}, T[0]); issue_group(IC<1>{}, T[1]); issue_group(IC<2>{}, T[2]);
        finish_group(IC<0>{}, IC<16>{}, T[0]); issue_group(IC<3>{}, T[0]);
        finish_group(IC<1>{}, IC<16>{}, T[1]); issue_group(IC<4>{}, T[1]);
        finish_group(IC<2>{}, IC<16>{}, T[2]); issue_group(IC<5>{}, T[2]);
        finish_group(IC<3>{}, IC<16>{}, T[0]); issue_group(IC<6>{}, T[0]);
        finish_group(IC<4>{}, IC<16>{}, T[1]); issue_group(IC<7>{}, T[1]);
        finish_group(IC<5>{}, IC<16>{}, T[2]); issue_group(IC<8>{}, T[2]);
        finish_group(IC<6>{}, IC<16>{}, T[0]); issue_group(IC<9>{}, T[0]);
        finish_group(IC<7>{}, IC<16>{}, T[1]); issue_group(IC<10>{}, T[1]);
        finish_group(IC<8>{}, IC<16>{}, T[2]); issue_group(IC<11>{}, T[2]);
        finish_group(IC<9>{}, IC<16>{}, T[0]);
        finish_group(IC<10>{}, IC<8>{}, T[1]);
        finish_group(IC<11>{}, IC<0>{}, T[2]);
    }

    // slab 0 and W stages 0..2 landed long ago for this wave (they are older than the residual loads); for everyone:
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FR_BAR();
    unsigned c_off = 0;          // W ring byte offset of the stage being multiplied
    unsigned a_cur = F_ARING;    // byte offset of the A slab being multiplied
    bf16x8 a0[2], a1[2], wf[4];
#ifdef DITTO_DIAG_FR_VALU
    float dz0 = 1.f; f32x2 dz1 = {1.f, 1.f}, dz2 = {1.f, 1.f};
    float dy0 = 1.f; f32x2 dy1 = {1.f, 1.f}, dy2 = {1.f, 1.f};
    float dw0 = 1.f, dw1 = 1.f, dw2 = 1.f, dw3 = 1.f, dw4 = 1.f, dw5 = 1.f;
#endif
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) a0[mb] = *reinterpret_cast<const bf16x8*>(smem + a_cur + a_row + mb * 4096 + a_x);
#pragma unroll
    for (int n = 0; n < 3; ++n) wf[n] = *reinterpret_cast<const bf16x8*>(smem + w_off + n * 1024);

    // One stage = K 16.  J: its position in the A slab.  ACUR: its A fragments (resident), ANXT receives the next stage's.
    // ISSUE_W: stage t+3 exists and its 6 pieces go out behind the first MFMAs; ISSUE_A: a next slab exists and (J < 2) two
    // of its 4 pieces go out behind those; NEXT: a next stage exists.  VM: the loads this wave may leave in flight when it
    // needs stage t+1 (and, at J = 3, the next slab) landed = everything issued during stages t-1 and t — counted vmcnt
    // takes an immediate, hence all of this at compile time.
    auto stage = [&](auto J, auto ISSUE_W, auto ISSUE_A, auto NEXT, auto VM, bf16x8 (&ACUR)[2], bf16x8 (&ANXT)[2]) {
        constexpr int j = decltype(J)::value, vm = decltype(VM)::value;
        constexpr bool do_w = decltype(ISSUE_W)::value != 0, do_a = decltype(ISSUE_A)::value != 0 && j < 2;
        constexpr bool has_next = decltype(NEXT)::value != 0;
        const char* cur = smem + c_off;
        const unsigned n_off = c_off + F_W_BYTES == F_WRING ? 0u : c_off + F_W_BYTES;
        const char* nxt = smem + n_off;
        const unsigned a_nxt = j == 3 ? (unsigned)(2 * F_ARING + F_ASLAB) - a_cur : a_cur;
#pragma unroll
        for (int nb = 0; nb < 12; ++nb) {
            if (nb + 3 < 12) wf[(nb + 3) & 3] = *reinterpret_cast<const bf16x8*>(cur + w_off + (nb + 3) * 1024);
            if (nb == 8 && has_next) {
                // stage t+1 has landed: this wave's pieces by the counted vmcnt (what was issued after them stays in
                // flight), everyone's by the barrier — which also certifies that every wave is past stage t-1, whose W
                // slot (and, at J = 3, whose A slab) the NEXT stage's DMA issue overwrites
                static_assert(vm == 16 || vm == 14 || vm == 12 || vm == 6 || vm == 0, "");
                if constexpr (vm == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if constexpr (vm == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                else if constexpr (vm == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if constexpr (vm == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef DITTO_DIAG_FR_NOBAR     // main loop without its per-stage barrier (timing only, racy)
                FR_BAR();
#endif
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    ANXT[mb] = *reinterpret_cast<const bf16x8*>(smem + a_nxt + a_row + mb * 4096 + (a_x ^ (((j + 1) & 3) << 5)));
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if constexpr (has_next) {
                    if (nb < NA) mfma_a(acca[nb < NA ? nb : 0][mb], wf[nb & 3], ACUR[mb]);
                    else mfma_v(accv[nb < NA ? 0 : nb - NA][mb], wf[nb & 3], ACUR[mb]);
                } else {   // last stage: each MFMA carries its own wait states (hazard argument above)
                    if (nb < NA) mfma_a_last(acca[nb < NA ? nb : 0][mb], wf[nb & 3], ACUR[mb]);
                    else mfma_v_last(accv[nb < NA ? 0 : nb - NA][mb], wf[nb & 3], ACUR[mb]);
                }
#if defined(DITTO_DIAG_FR_VALU) && DITTO_DIAG_FR_VALU == 7   // VALU-bound mix behind every MFMA: 2 transcendentals + 3 plain ops (44 cycles)
                if (mb == 0) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %3, %3, %3, %3\n\tv_max_f32 %4, %4, %3\n\tv_exp_f32 %5, %5\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(dz0), "+v"(dz1), "+v"(dz2), "+v"(dy0), "+v"(dw0), "+v"(dw1));
                else asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %3, %3, %3, %3\n\tv_max_f32 %4, %4, %3\n\tv_exp_f32 %5, %5\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(dw2), "+v"(dy1), "+v"(dy2), "+v"(dw3), "+v"(dw4), "+v"(dw5));
#endif
#if defined(DITTO_DIAG_FR_VALU) && DITTO_DIAG_FR_VALU == 4   // 2 packed fp32 FMAs behind every MFMA
                if (mb == 0) asm volatile("v_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2" : "+v"(dz0), "+v"(dz1), "+v"(dz2));
                else asm volatile("v_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2" : "+v"(dy0), "+v"(dy1), "+v"(dy2));
#endif
#if defined(DITTO_DIAG_FR_VALU) && DITTO_DIAG_FR_VALU == 5   // 1 transcendental behind every MFMA
                if (mb == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(dz0), "+v"(dz1), "+v"(dz2));
                else asm volatile("v_rcp_f32 %0, %0" : "+v"(dy0), "+v"(dy1), "+v"(dy2));
#endif
#if defined(DITTO_DIAG_FR_VALU) && DITTO_DIAG_FR_VALU == 6   // 4 plain fp32 FMAs behind every MFMA
                if (mb == 0) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %3, %3, %3, %3\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(dz0), "+v"(dz1), "+v"(dz2), "+v"(dy0));
                else asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %3, %3, %3, %3\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(dy0), "+v"(dy1), "+v"(dy2), "+v"(dz0));
#endif
#if defined(DITTO_DIAG_FR_VALU) && DITTO_DIAG_FR_VALU == 3   // 24 cycles of independent VALU work behind EVERY MFMA
                if (mb == 0) asm volatile("v_exp_f32 %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2" : "+v"(dz0), "+v"(dz1), "+v"(dz2));
                else asm volatile("v_rcp_f32 %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2" : "+v"(dy0), "+v"(dy1), "+v"(dy2));
#endif
            }
#if defined(DITTO_DIAG_FR_VALU) && DITTO_DIAG_FR_VALU < 3   // tools/build_diag.sh: dummy VALU work behind every MFMA pair (does it hide under the matrix pipe?)
            asm volatile("v_exp_f32 %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2\n\t"
                         "v_rcp_f32 %0, %0\n\tv_pk_mul_f32 %1, %1, %2\n\tv_pk_fma_f32 %2, %2, %1, %1\n\t"
#if DITTO_DIAG_FR_VALU > 1
                         "v_exp_f32 %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2\n\t"
                         "v_rcp_f32 %0, %0\n\tv_pk_mul_f32 %1, %1, %2\n\tv_pk_fma_f32 %2, %2, %1, %1\n\t"
#endif
                         : "+v"(dz0), "+v"(dz1), "+v"(dz2));
#endif
            if constexpr (do_w) {
                if (nb == 0) issue_w_piece(IC<0>{});
                if (nb == 1) issue_w_piece(IC<1>{});
                if (nb == 2) issue_w_piece(IC<2>{});
                if (nb == 3) issue_w_piece(IC<3>{});
                if (nb == 4) issue_w_piece(IC<4>{});
                if (nb == 5) issue_w_piece(IC<5>{});
            }
            if constexpr (do_a) {
                if (nb == 6) issue_a_piece(IC<2 * (j & 1)>{});
                if (nb == 7) issue_a_piece(IC<2 * (j & 1) + 1>{});
            }
            if (nb >= 9 && has_next)   // W fragments 0..2 of the next stage, into ring slots the MFMAs above released
                wf[(nb - 9) & 3] = *reinterpret_cast<const bf16x8*>(nxt + w_off + (nb - 9) * 1024);
        }
        if constexpr (do_w) advance_w();
        if constexpr (do_a && j == 1) advance_a();
        if constexpr (j == 3) a_cur = a_nxt;
        c_off = n_off;
    };
    // all slabs but the last: every stage issues W stage t+3, the first two also half of the next slab each (per stage
    // 8, 8, 6, 6 loads per wave: the waits leave 14, 16, 14, 12 in flight); then the last slab, in which the issue stops
    for (int sl = 0; sl + 1 < nslab; ++sl) {
        stage(IC<0>{}, IC<1>{}, IC<1>{}, IC<1>{}, IC<14>{}, a0, a1);
        stage(IC<1>{}, IC<1>{}, IC<1>{}, IC<1>{}, IC<16>{}, a1, a0);
        stage(IC<2>{}, IC<1>{}, IC<1>{}, IC<1>{}, IC<14>{}, a0, a1);
        stage(IC<3>{}, IC<1>{}, IC<1>{}, IC<1>{}, IC<12>{}, a1, a0);
    }
    stage(IC<0>{}, IC<1>{}, IC<0>{}, IC<1>{}, IC<12>{}, a0, a1);   // stage nkt-4: issues W stage nkt-1
    stage(IC<1>{}, IC<0>{}, IC<0>{}, IC<1>{}, IC<6>{}, a1, a0);    // stage nkt-3
    stage(IC<2>{}, IC<0>{}, IC<0>{}, IC<1>{}, IC<0>{}, a0, a1);    // stage nkt-2
    stage(IC<3>{}, IC<0>{}, IC<0>{}, IC<0>{}, IC<0>{}, a1, a0);    // stage nkt-1

    // ---------------- epilogue: the accumulators hold h = residual + bias + A W^T; they are only READ from here on ----------------
    // MFMA results -> any other reader need wait states hipcc does not insert for asm producers
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s1[2] = {0.f, 0.f};
    if constexpr (LN) {
#pragma unroll
        for (int nb = 0; nb < 12; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const f32x16 v = nb < NA ? acca[nb < NA ? nb : 0][mb] : accv[nb < NA ? 0 : nb - NA][mb];
#pragma unroll
                for (int e = 0; e < 16; ++e) s1[mb] += v[e];
                __builtin_amdgcn_sched_barrier(0);   // one block at a time (hoisted block copies cost registers)
            }
    }
    float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
    if constexpr (LN) {
        float* red = reinterpret_cast<float*>(smem + F_RED);          // [pass][wn][128]
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            float t = s1[mb];
            t += __shfl_xor(t, 32, 64);
            if (hh == 0) red[wn * FM + wm * 64 + mb * 32 + r32] = t;
        }
        __syncthreads();
        // every wave is out of the main loop: gamma and beta rows -> the idle A ring (3 pieces of 1 KiB each), landed by
        // the second exchange below
        if (wid < 2) {
            const float* src = wid == 0 ? fp.gamma : fp.beta;
#pragma unroll
            for (int i = 0; i < 3; ++i) glds16(src + i * 256 + lane * 4, lds_base + (unsigned)(F_GB + wid * FN * 4 + i * 1024));
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int r = wm * 64 + mb * 32 + r32;
            mean[mb] = (red[r] + red[FM + r]) * (1.0f / FN);
        }
        float q2[2] = {0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < 12; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                // re-pin the block in its home file: the copy below is then a NEW value, and hipcc stops trying to keep
                // the first pass's VGPR copies of all 24 blocks alive for this pass (that is what spilled accumulators)
                if (nb < NA) PIN_A(acca[nb < NA ? nb : 0][mb]); else PIN_V(accv[nb < NA ? 0 : nb - NA][mb]);
                const f32x16 v = nb < NA ? acca[nb < NA ? nb : 0][mb] : accv[nb < NA ? 0 : nb - NA][mb];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float dl = v[e] - mean[mb];
                    q2[mb] = fmaf(dl, dl, q2[mb]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        float* red2 = red + 2 * FM;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            float t = q2[mb];
            t += __shfl_xor(t, 32, 64);
            if (hh == 0) red2[wn * FM + wm * 64 + mb * 32 + r32] = t;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int r = wm * 64 + mb * 32 + r32;
            rstd[mb] = rsqrtf((red2[r] + red2[FM + r]) * (1.0f / FN) + 1e-5f);
        }
    }
    // stores: h fp32 (non-temporal), u = LN(h) bf16, optional bf16 copy of h (last layer: the proj_out operand).
    // ONE base pointer per (output, row block) and compile-time element offsets: with the addresses written as
    // row * ld + col hipcc kept dozens of them live, spilled them, and every reload's vmcnt(0) drained the stores in flight.
    const int cl = wn * 384 + 4 * hh;                              // this lane's column origin; + nb * 32 + 8 g
    const float* gl = lgamma + cl;
    const float* bl = lbeta + cl;
    // Every output row leaves through LDS: a lane holds 4 consecutive columns of ONE row per (block, g), so a direct store
    // instruction would touch 32 rows for 32 bytes each (16 for bf16) — partial-line writes that cost the first version
    // most of a 150 us epilogue.  Each wave stages its 64 rows x 32 columns (fp32: 128 B per row = one cache line; bf16:
    // two column blocks per 128-B line) in a private 16 KiB of the idle ring (16-B chunk q of row r at q ^ (r & 7)), reads
    // them back 8 lanes per row and stores WHOLE lines.  Wave-private staging: LDS operations of one wave execute in order,
    // no barrier.
    FR_BAR();                                                       // every wave is past the statistics' LDS traffic
    char* hst = smem + wid * 16384;                                 // h stage: [64 rows][128 B]
    char* ust = hst + 8192;                                         // u stage: [64 rows][128 B] = 64 bf16 columns
    const int srow = lane >> 3, sq = lane & 7;                      // read-back: row srow (+ 8 i), 16-B chunk sq
    const int grow0 = m0 + wm * 64 + srow;
    float* hrow = (float*)p.out + (size_t)(grow0 < p.M ? grow0 : 0) * p.ldo + wn * 384 + sq * 4;
    bf16* urow = fp.u ? fp.u + (size_t)(grow0 < p.M ? grow0 : 0) * fp.ldu + wn * 384 + sq * 8 : nullptr;
    bf16* orow = p.out2 ? p.out2 + (size_t)(grow0 < p.M ? grow0 : 0) * p.ldo2 + wn * 384 + sq * 8 : nullptr;
#pragma unroll
    for (int nb = 0; nb < 12; ++nb) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            if (nb < NA) PIN_A(acca[nb < NA ? nb : 0][mb]); else PIN_V(accv[nb < NA ? 0 : nb - NA][mb]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = nb * 32 + 8 * g;                         // compile-time
            f32x4 g4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (LN) {
                g4 = *reinterpret_cast<const f32x4*>(gl + co);
                b4 = *reinterpret_cast<const f32x4*>(bl + co);
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const f32x16& v = nb < NA ? acca[nb < NA ? nb : 0][mb] : accv[nb < NA ? 0 : nb - NA][mb];
                const f32x4 v4 = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                const int row = mb * 32 + r32;
                *reinterpret_cast<f32x4*>(hst + row * 128 + (((2 * g + hh) ^ (row & 7)) << 4)) = v4;
                f32x4 y = v4;                                        // bf16 side: LayerNorm output, or the plain copy
                if constexpr (LN) y = (v4 - mean[mb]) * rstd[mb] * g4 + b4;
                u32x2 st;
                st[0] = pack_bf16x2(y[0], y[1]); st[1] = pack_bf16x2(y[2], y[3]);
                *reinterpret_cast<u32x2*>(ust + row * 128 + ((((nb & 1) * 4 + g) ^ (row & 7)) << 4) + hh * 8) = st;
            }
        }
        // read back 8 rows per instruction, whole lines out, four instructions at a time (the data of eight reads in
        // flight plus their addresses was what pushed hipcc into spilling accumulator blocks).  The waits keep the LDS
        // traffic of this wave ordered; the staging writes and the read-back use different vector types, so the asm
        // statements also keep type-based alias analysis from moving a read above the writes it depends on.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i = 4 * half; i < 4 * half + 4; ++i) {
                const int row = srow + 8 * i;
                const u32x4 hv = *reinterpret_cast<const u32x4*>(hst + row * 128 + ((sq ^ (row & 7)) << 4));
                if (grow0 + 8 * i < FR_DIAG_M) store16<true, true>(hrow + (size_t)(8 * i) * p.ldo + nb * 32, hv, 0);   // nt (a run-time plain / nt switch here cost the K = 768 launch 5 us: the branch splits the store block; the A/B itself: no gain from plain stores)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (nb & 1) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int i = 4 * half; i < 4 * half + 4; ++i) {
                    const int row = srow + 8 * i;
                    const u32x4 uv = *reinterpret_cast<const u32x4*>(ust + row * 128 + ((sq ^ (row & 7)) << 4));
                    if (grow0 + 8 * i < FR_DIAG_M) {
                        if (LN) *reinterpret_cast<u32x4*>(urow + (size_t)(8 * i) * fp.ldu + (nb - 1) * 32) = uv;
                        else if (orow) *reinterpret_cast<u32x4*>(orow + (size_t)(8 * i) * p.ldo2 + (nb - 1) * 32) = uv;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool LN, bool RES>
hipError_t launch_fr_t(const FrParams& fp, int grid, hipStream_t s) {
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&gemm_fr_kernel<LN, RES>)}, F_LDS)) return e;
    hipLaunchKernelGGL((gemm_fr_kernel<LN, RES>), dim3(grid), dim3(256), F_LDS, s, fp);
    return hipGetLastError();
}

}  // namespace

// Opt-in A/B kernel since round 5 (csrc/experimental/, built with DITTO_EXPERIMENTAL=1, selected with ditto_set_option("fr_tile",
// 128)): the default N = 768 full-row kernels are gemm_frd.hip (128 rows, weights straight into registers) and gemm_fr64.hip;
// no dispatch rule has selected this LDS-ring form since round 3 (same fp32 h bits as both).  The dispatcher, launch_gemm_fr,
// lives in csrc/gemm_fr.hip.
hipError_t launch_gemm_fr128(const FrParams& fp, hipStream_t s) {
    const bool ln = fp.gamma && fp.u, res = fp.g.residual != nullptr;
    if (ln) return res ? launch_fr_t<true, true>(fp, fp.g.tiles_m, s) : launch_fr_t<true, false>(fp, fp.g.tiles_m, s);
    return res ? launch_fr_t<false, true>(fp, fp.g.tiles_m, s) : launch_fr_t<false, false>(fp, fp.g.tiles_m, s);
}

}  // namespace ditto
