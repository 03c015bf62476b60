// gemm_fr64.hip — the full-row N = 768 GEMM with fused residual + LayerNorm (gemm_fr.hip) on 64-ROW tiles, TWO independent
// workgroups per CU (gfx950).  Same contract, same weight layout, and the SAME BITS as gemm_fr.hip.
//
//     h[M, 768] (fp32, in place) = residual + A[M, K] * W[768, K]^T + bias            (reference DiT.py:148, :155)
//     u[M, 768] (bf16)           = LayerNorm(h) * gamma + beta   (eps 1e-5)           (reference DiT.py:152, :105)
//
// Why: gemm_fr.hip owns 128 rows per workgroup with one wave per SIMD, one workgroup per CU and (at M = 32768) one tile per
// workgroup, so its phases — 100 MB of residual read, the MFMA loop, LayerNorm, 150 MB of stores — run one after the other,
// chip-wide in lockstep: at K = 768 the MFMA loop is 27 % of the launch.  A 64 x 768 tile needs 192 accumulators per lane,
// so two 4-wave workgroups fit a CU (256 registers and 80 KiB of LDS each) and the matrix pipe, the vector memory path and
// the store path each see two clients in DIFFERENT phases once the workgroup that holds the CU's second LDS allocation
// starts late (stagger_ticks): its residual read runs under the first one's MFMA loop, its MFMA loop under the first
// one's LayerNorm and stores.  The price is W traffic: every workgroup streams all of W, so the L2 -> LDS bytes per FLOP
// double against the 128-row tile (measured consequences: DESIGN.md section 8, round 3).
//
//   tile      64 rows x 768 columns; 256 threads = 4 waves side by side in N: wave wn owns all 64 rows x columns
//             [192 wn, +192) = 2 x 6 blocks of v_mfma_f32_32x32x16_bf16 = 192 accumulators, pinned by asm MFMAs: column
//             blocks 0..NA-1 in AGPRs, the rest in VGPRs (hipcc cannot be trusted with them as values, gemm_fr.hip).
//   operands  W stage-major packed Wp[K/16][768][16] (the gemm_fr.hip layout).  A wave DMAs exactly the 6 KiB of a K = 16
//             stage that IT multiplies (its own 192 weight rows = 6 pieces of 1 KiB): the W ring is WAVE-PRIVATE, no
//             barrier guards it, and piece nb of stage s+3 is issued into the bytes of piece nb of stage s right behind the
//             MFMAs that consumed that fragment (three slots give 2-2.5 stages of look-ahead).  A (row-major, the
//             previous kernel's output) is shared by the four waves: slabs of 32 k (64 rows x 64 B = four 1-KiB pieces,
//             one per wave), double-buffered, ONE workgroup barrier per slab (two stages).
//   LDS       W ring 3 x 24 KiB + A 2 x 4 KiB = 80 KiB exactly (two workgroups = the CU's 160 KiB).  W: 16-B chunk h of
//             32-B row r at h ^ ((r >> 3) & 1); A: chunk c of 64-B row r at c ^ (-(r >> 2) & 3) (the 16 lanes of every
//             ds_read_b128 lane group hit 16 different 16-B slots).  The bias row borrows ring slot 2 until the
//             accumulators are initialised (W stage 2 is issued after that); gamma / beta / row statistics / the store
//             staging borrow the idle ring in the epilogue.
//   stage s   (j = s & 1; G = 6 j + nb indexes a 4-register W fragment ring three fragments ahead, across stages)
//             nb = 0..2: 2 MFMAs + W piece nb of stage s+3 each
//             nb = 3   : counted vmcnt (this wave's stage s+1 has landed) ; j = 1: s_barrier (everyone's piece of the next
//                        slab has landed, everyone is done reading this one) + this wave's piece of the slab after next ;
//                        A fragments of stage s+1 ; then as nb = 0..2, W fragments now from stage s+1
//   bits      the K loop starts where gemm_fr.hip's does for the 128-row tile these rows belong to (rot_period), the
//             accumulators start as residual + bias, and the LayerNorm statistics are summed in the same association
//             (per lane ONE chain over a half row — the odd wave of a pair continues the even wave's partial —, + lane ^ 32,
//             half 0 + half 1): h and u are bit-identical to
//             gemm_fr.hip's (tests/test_gpu_kernels.py), so the choice between the two kernels is not a numerics class.
//
// d = 1024 (BASELINE config C5): the same kernel with NBW = 8 column blocks per wave — a 64 x 1024 tile, 256 accumulators per
// lane (all in AGPRs), ONE workgroup per CU with the whole register file, a W ring of FOUR 32-KiB stages (136 KiB of LDS),
// DMA four stages ahead.  M = 16 x 1024 rows make exactly one tile per CU.  There is no 128-row kernel at this width, so
// nothing pins its bits; U8 = the LayerNorm output as fp8 e4m3 (saturating) for the fp8 linear path's next GEMM.
#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int HM = 64, HK = 16;
constexpr int H_ASLAB = HM * 32 * 2;              // 4 KiB: 64 rows x 32 k
constexpr int H_STAGE = 4 * 16384;                // epilogue: 4 x 16 KiB of store staging, then gamma | beta rows, then row statistics
// NBW = 32-column blocks per wave: 6 -> N = 768 (two workgroups per CU), 8 -> N = 1024 (one)
template <int NBW> struct FH {
    static constexpr int HN = 128 * NBW;
    static constexpr int HNS = NBW == 6 ? 3 : 4;                // W ring slots
    static constexpr int W_BYTES = HN * HK * 2;                 // 24 / 32 KiB: one K = 16 stage of W
    static constexpr int WRING = HNS * W_BYTES;                 // 72 / 128 KiB
    static constexpr int ARING = WRING;
    static constexpr int LDS = ARING + 2 * H_ASLAB;             // 80 / 136 KiB
    static constexpr int BIAS = (HNS - 1) * W_BYTES;            // bias row: the last ring slot, until the accumulators are initialised
    static constexpr int GB = H_STAGE;                          // gamma | beta rows (6 / 8 KiB)
    static constexpr int RED = GB + 2 * HN * 4;                 // per-lane partials [2 pairs][2][64] + row sums [2 halves][64] fp32 (1.5 KiB)
    static constexpr int HNA = NBW == 6 ? 4 : 8;                // column blocks whose accumulators live in AGPRs (hipcc splits 256 registers 128 / 128)
    static constexpr int WD = NBW == 6 ? 2 : 4;                 // residual blocks (4 loads each) in flight per wave while the accumulators are initialised
    static constexpr int WPE = NBW == 6 ? 2 : 1;                // waves per SIMD
    // loads a wave may leave in flight at the wait of stage s (after NBW - 3 pieces of stage s + HNS went out), derived in
    // the header for NBW = 6 and the same way for NBW = 8: steady state j = 0 / j = 1, then stages nkt-4, nkt-3
    static constexpr int VM0 = NBW == 6 ? 10 : 22, VM1 = NBW == 6 ? 9 : 16;
    static constexpr int VMT4 = NBW == 6 ? 10 : 17, VMT3 = NBW == 6 ? 6 : 3;
    static constexpr int T4_ISSUES_W = NBW == 6 ? 1 : 0;        // stage nkt-4 still issues W stage nkt-1 with a 3-slot ring
    static_assert(RED + 3 * 2 * HM * 4 <= WRING, "epilogue overlays fit the idle W ring");
};

#ifdef DITTO_DIAG_FR_NOSTORE
#define FH_DIAG_M (p.M - (1 << 30))
#else
#define FH_DIAG_M p.M
#endif
#define FH_BAR() asm volatile("s_barrier" ::: "memory")
#define FH_PIN_A(x) asm volatile("" : "+a"(x))
#define FH_PIN_V(x) asm volatile("" : "+v"(x))

template <int V>
struct HC { static constexpr int value = V; };

DITTO_DEV void hmfma_a(f32x16& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
}
DITTO_DEV void hmfma_v(f32x16& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(a));
}
// The LAST MFMA of an accumulator chain carries its own wait states: an 8-pass MFMA's result may be read by anything but
// the next MFMA of its chain only 11 cycles after issue, hipcc pads nothing behind an asm producer, and it DID place the
// spill of a just-written block between two MFMA statements (ahead of a separate s_nop statement: wrong lanes in u).
DITTO_DEV void hmfma_a_last(f32x16& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+a"(c) : "v"(w), "v"(a));
}
DITTO_DEV void hmfma_v_last(f32x16& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+v"(c) : "v"(w), "v"(a));
}

template <int NBW, bool LN, bool RES, bool U8>
__global__ __launch_bounds__(256, FH<NBW>::WPE) void gemm_fr64_kernel(FrParams fp) {
    constexpr int HN = FH<NBW>::HN, HNS = FH<NBW>::HNS, H_W_BYTES = FH<NBW>::W_BYTES, H_WRING = FH<NBW>::WRING;
    constexpr int H_ARING = FH<NBW>::ARING, H_BIAS = FH<NBW>::BIAS, H_GB = FH<NBW>::GB, H_RED = FH<NBW>::RED, HNA = FH<NBW>::HNA;
    constexpr int WCOLS = 32 * NBW, NBP = HN * 4 / 1024;             // a wave's columns; 1-KiB pieces of an fp32 row of N
    static_assert(!U8 || (LN && NBW % 4 == 0), "fp8 LayerNorm output: four column blocks per 128-B line");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemmParams& p = fp.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);       // = the wave's column quarter
    const int nkt = p.K / HK;                                        // a multiple of 4 (K % 64 == 0)
    const int nslab = nkt >> 1;                                      // 32-k slabs
    // XCD-contiguous tiles (workgroups go to the XCDs round-robin): the 64 workgroups of an XCD hold neighbouring rows
    const int ntile = gridDim.x;
    const int tile = (ntile & 7) == 0 ? (int)(blockIdx.x & 7) * (ntile >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int m0 = tile * HM;
    // K-loop rotation exactly as gemm_fr.hip applies it to the 128-row tile these 64 rows belong to (same sums, same bits)
    const int nslab64 = nkt >> 2;
    const int s0 = fp.rot_period > 0 ? ((((tile >> 1) % fp.rot_period) & 7) * nslab64) >> 3 : 0;   // in 64-k units
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;

    // ---- phase offset: the workgroup holding the CU's SECOND LDS allocation starts late (speed only) ----
    if (fp.stagger_ticks > 0) {
        // HW_REG_LDS_ALLOC (id 6): LDS_BASE in bits [7:0]; s_getreg simm16 = (size-1) << 11 | offset << 6 | id
        const unsigned lds_alloc_base = __builtin_amdgcn_s_getreg((7 << 11) | (0 << 6) | 6);
        if (lds_alloc_base != 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)fp.stagger_ticks) __builtin_amdgcn_s_sleep(8);
        }
    }

    // ---- operand DMA (sources = loop-invariant scalar base + per-lane offset that advances per stage / slab) ----
    const int prow = lane >> 1, ppos = lane & 1;
    const int pc = ppos ^ ((prow >> 3) & 1);                     // W: source chunk landing at position ppos of row prow
    unsigned vwk = (unsigned)(prow * 32 + pc * 16 + s0 * 4 * H_W_BYTES);
    int w_left = nkt - 4 * s0, a_left = nslab - 2 * s0;           // stages / slabs until the rotated K loop wraps to k = 0
    const char* wbase[NBW];                                         // wave-uniform: this wave's own six 32-row pieces
#pragma unroll
    for (int i = 0; i < NBW; ++i) wbase[i] = (const char*)p.W + (size_t)(wid * NBW + i) * 1024;
    unsigned vak;                                                 // A: this wave's piece = rows [16 wid, +16) x 64 B
    {
        const int row = 16 * wid + (lane >> 2);
        int ar = m0 + row;
        ar = ar < p.M ? ar : p.M - 1;
        vak = (unsigned)(((size_t)ar * p.lda + ((lane & 3) ^ ((0 - (row >> 2)) & 3)) * 8) * 2) + (unsigned)(s0 * 128);
    }
    unsigned w_slot = lds_base + (unsigned)(wid * NBW * 1024);      // LDS address of this wave's pieces in the slot the next stage goes to
    unsigned a_buf = lds_base + H_ARING + (unsigned)(wid * 1024); // ... and of this wave's piece in the buffer the next slab goes to
    auto dma = [&](unsigned voff, const char* base, unsigned dst) {
#ifndef DITTO_DIAG_FR_NODMA
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(dst) : "memory");
#endif
    };
    auto issue_w_piece = [&](auto I) {
        constexpr int i = decltype(I)::value;
        dma(vwk, wbase[i], w_slot + (unsigned)(i * 1024));
    };
    auto advance_w = [&]() {
        w_slot = w_slot + H_W_BYTES >= lds_base + H_WRING ? w_slot + H_W_BYTES - H_WRING : w_slot + H_W_BYTES;
        --w_left;
        vwk += w_left == 0 ? (unsigned)H_W_BYTES - (unsigned)nkt * H_W_BYTES : (unsigned)H_W_BYTES;
    };
    auto issue_a_piece = [&]() { dma(vak, (const char*)p.A, a_buf); };
    auto advance_a = [&]() {
        a_buf = a_buf >= lds_base + H_ARING + H_ASLAB ? a_buf - H_ASLAB : a_buf + H_ASLAB;
        --a_left;
        vak += a_left == 0 ? 64u - (unsigned)nslab * 64u : 64u;
    };
    auto issue_w_stage = [&]() {
        issue_w_piece(HC<0>{}); issue_w_piece(HC<1>{}); issue_w_piece(HC<2>{});
        issue_w_piece(HC<3>{}); issue_w_piece(HC<4>{}); issue_w_piece(HC<5>{});
        if constexpr (NBW == 8) { issue_w_piece(HC<6>{}); issue_w_piece(HC<7>{}); }
        advance_w();
    };

    // bias row -> the last ring slot (N fp32 = 3 / 4 pieces of 1 KiB): the oldest loads of the kernel
    if (wid == 0) {
        if (p.bias) {
#pragma unroll
            for (int i = 0; i < NBP; ++i) glds16(p.bias + i * 256 + lane * 4, lds_base + (unsigned)(H_BIAS + i * 1024));
        } else {
#pragma unroll
            for (int i = 0; i < NBP; ++i) *reinterpret_cast<f32x4*>(smem + H_BIAS + i * 1024 + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // ---- fragment addressing: lane reads row (lane & 31) of a 32-row block, 16-B half (lane >> 5) ----
    const int r32 = lane & 31, hh = lane >> 5;
    const int w_off = (wid * WCOLS + r32) * 32 + ((hh ^ ((r32 >> 3) & 1)) << 4);    // in a W slot: + nb * 1024
    const int a_row = r32 * 64;                                                    // + mb * 2048
    const int a_x = (hh ^ ((0 - (r32 >> 2)) & 3)) << 4;                            // stage j of the slab: ^ (j << 5)

    // ---- prologue DMA: A slabs 0 and 1, W stages 0 and 1 (stage 2 follows the accumulator init: its slot holds the bias) ----
    issue_a_piece(); advance_a();
    issue_a_piece(); advance_a();
    issue_w_stage(); issue_w_stage();
    if constexpr (HNS == 4) issue_w_stage();

    // ---- the accumulators START as bias + residual (gemm_fr.hip: the epilogue then only READS them).  48 hand-written
    //      global_load_dwordx4 per lane in the accumulator layout, a window of two (N = 768) or four (N = 1024) 32 x 32 blocks in flight. ----
    const float* lbias = reinterpret_cast<const float*>(smem + H_BIAS);
    const float* lgamma = reinterpret_cast<const float*>(smem + H_GB);     // these two: valid in the epilogue only
    const float* lbeta = lgamma + HN;
    f32x16 acca[HNA][2], accv[HNA < NBW ? NBW - HNA : 1][2];
    {
        const float* rp[2] = {nullptr, nullptr};
        if constexpr (RES) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                int gr = m0 + mb * 32 + r32;
                gr = gr < p.M ? gr : p.M - 1;
                rp[mb] = p.residual + (size_t)gr * p.ldr + wid * WCOLS + 4 * hh;
            }
        }
        constexpr int WD = FH<NBW>::WD, NG = 2 * NBW;                // window depth; groups = (nb, mb) blocks
        f32x4 T[WD][4];                                              // [window slot][g]
        auto issue_group = [&](auto GI, f32x4 (&t)[4]) {             // the 4 loads of block (nb, mb) = (GI / 2, GI % 2)
            constexpr int nb = decltype(GI)::value >> 1, mb = decltype(GI)::value & 1;
            if constexpr (RES) {
                const float* ptr = rp[mb];
                asm volatile("global_load_dwordx4 %0, %4, off offset:%5\n\t"
                             "global_load_dwordx4 %1, %4, off offset:%6\n\t"
                             "global_load_dwordx4 %2, %4, off offset:%7\n\t"
                             "global_load_dwordx4 %3, %4, off offset:%8"
                             : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                             : "v"(ptr), "n"(nb * 128), "n"(nb * 128 + 32), "n"(nb * 128 + 64), "n"(nb * 128 + 96)
                             : "memory");
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto finish_group = [&](auto GI, f32x4 (&t)[4]) {
            constexpr int gi = decltype(GI)::value, nb = gi >> 1, mb = gi & 1;
            constexpr int younger = NG - 1 - gi < WD - 1 ? NG - 1 - gi : WD - 1;   // groups issued after this one and still in flight
            if constexpr (RES)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) : "n"(4 * younger) : "memory");
            if constexpr (gi == 0) {
                if constexpr (!RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                FH_BAR();      // every wave is past a wait that covers wave 0's bias row (the oldest load): visible to all
            }
            f32x16 v;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbias + wid * WCOLS + nb * 32 + 8 * g + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g + e] = t[g][e] + b4[e];
            }
            if constexpr (nb < HNA) { acca[nb < HNA ? nb : 0][mb] = v; FH_PIN_A(acca[nb < HNA ? nb : 0][mb]); }
            else { accv[nb < HNA ? 0 : nb - HNA][mb] = v; FH_PIN_V(accv[nb < HNA ? 0 : nb - HNA][mb]); }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto init_step = [&](auto GI) {
            constexpr int gi = decltype(GI)::value;
            finish_group(GI, T[gi % WD]);
            if constexpr (gi + WD < NG) issue_group(HC<gi + WD>{}, T[gi % WD]);
        };
        issue_group(HC<0>{}, T[0]); issue_group(HC<1>{}, T[1]);
        if constexpr (WD == 4) { issue_group(HC<2>{}, T[2]); issue_group(HC<3>{}, T[3]); }
        init_step(HC<0>{}); init_step(HC<1>{}); init_step(HC<2>{}); init_step(HC<3>{}); init_step(HC<4>{}); init_step(HC<5>{});
        init_step(HC<6>{}); init_step(HC<7>{}); init_step(HC<8>{}); init_step(HC<9>{}); init_step(HC<10>{}); init_step(HC<11>{});
        if constexpr (NBW == 8) { init_step(HC<12>{}); init_step(HC<13>{}); init_step(HC<14>{}); init_step(HC<15>{}); }
    }

    // slabs 0, 1 and W stages 0, 1 have landed for this wave (older than the residual loads); for everyone, and everyone is
    // done with the bias row in slot 2:
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FH_BAR();
    issue_w_stage();             // the W stage whose slot held the bias row (nkt >= 4)
    unsigned c_off = 0;          // W ring byte offset of the stage being multiplied
    unsigned a_cur = H_ARING;    // byte offset of the A slab being multiplied
    bf16x8 a0[2], a1[2], wf[4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) a0[mb] = *reinterpret_cast<const bf16x8*>(smem + a_cur + a_row + mb * 2048 + a_x);
#pragma unroll
    for (int n = 0; n < 3; ++n) wf[n] = *reinterpret_cast<const bf16x8*>(smem + w_off + n * 1024);

    // One stage = K 16.  J: its position in the 32-k A slab.  ACUR: its A fragments (resident), ANXT receives the next
    // stage's.  ISSUE_W: stage s+3 exists (its six pieces go out behind the MFMAs that free their bytes); ISSUE_A: (J = 1)
    // a slab after next exists; NEXT: a next stage exists.  VM: the loads this wave may leave in flight when it needs its
    // stage s+1 (and, at J = 1, its piece of the next slab) landed — a compile-time immediate (header: 10 / 9 in the steady
    // state = the pieces issued since).
    auto stage = [&](auto J, auto ISSUE_W, auto ISSUE_A, auto NEXT, auto VM, bf16x8 (&ACUR)[2], bf16x8 (&ANXT)[2]) {
        constexpr int j = decltype(J)::value, vm = decltype(VM)::value;
        constexpr bool do_w = decltype(ISSUE_W)::value != 0, do_a = decltype(ISSUE_A)::value != 0 && j == 1;
        constexpr bool has_next = decltype(NEXT)::value != 0;
        const char* cur = smem + c_off;
        const unsigned n_off = c_off + H_W_BYTES == H_WRING ? 0u : c_off + H_W_BYTES;
        const char* nxt = smem + n_off;
        const unsigned a_nxt = j == 1 ? (unsigned)(2 * H_ARING + H_ASLAB) - a_cur : a_cur;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int G = NBW * j + nb;
            if (nb == NBW - 3 && has_next) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(vm) : "memory");
                if constexpr (j == 1) {
#ifndef DITTO_DIAG_FR_NOBAR
                    FH_BAR();
#endif
                    if constexpr (do_a) { issue_a_piece(); advance_a(); }
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    ANXT[mb] = *reinterpret_cast<const bf16x8*>(smem + a_nxt + a_row + mb * 2048 + (a_x ^ (((j + 1) & 1) << 5)));
            }
            // W fragment three ahead: this stage's while it has them, then the next stage's first three
            if (nb + 3 < NBW) wf[(G + 3) & 3] = *reinterpret_cast<const bf16x8*>(cur + w_off + (nb + 3) * 1024);
            else if (has_next) wf[(G + 3) & 3] = *reinterpret_cast<const bf16x8*>(nxt + w_off + (nb + 3 - NBW) * 1024);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if constexpr (has_next) {
                    if (nb < HNA) hmfma_a(acca[nb < HNA ? nb : 0][mb], wf[G & 3], ACUR[mb]);
                    else hmfma_v(accv[nb < HNA ? 0 : nb - HNA][mb], wf[G & 3], ACUR[mb]);
                } else {   // LAST stage: the final writes of the block's accumulators
                    if (nb < HNA) hmfma_a_last(acca[nb < HNA ? nb : 0][mb], wf[G & 3], ACUR[mb]);
                    else hmfma_v_last(accv[nb < HNA ? 0 : nb - HNA][mb], wf[G & 3], ACUR[mb]);
                }
            }
            if constexpr (do_w) {
                if (nb == 0) issue_w_piece(HC<0>{});
                if (nb == 1) issue_w_piece(HC<1>{});
                if (nb == 2) issue_w_piece(HC<2>{});
                if (nb == 3) issue_w_piece(HC<3>{});
                if (nb == 4) issue_w_piece(HC<4>{});
                if (nb == 5) issue_w_piece(HC<5>{});
                if constexpr (NBW == 8) {
                    if (nb == 6) issue_w_piece(HC<6>{});
                    if (nb == 7) issue_w_piece(HC<7>{});
                }
            }
        }
        if constexpr (do_w) advance_w();
        if constexpr (j == 1) a_cur = a_nxt;
        c_off = n_off;
    };
    // all slabs but the last two: full issue (the waits leave 10 / 9 loads in flight); then the last four stages, in which
    // the issue stops
    for (int sl = 0; sl + 2 < nslab; ++sl) {
        stage(HC<0>{}, HC<1>{}, HC<1>{}, HC<1>{}, HC<FH<NBW>::VM0>{}, a0, a1);
        stage(HC<1>{}, HC<1>{}, HC<1>{}, HC<1>{}, HC<FH<NBW>::VM1>{}, a1, a0);
    }
    stage(HC<0>{}, HC<FH<NBW>::T4_ISSUES_W>{}, HC<0>{}, HC<1>{}, HC<FH<NBW>::VMT4>{}, a0, a1);   // stage nkt-4 (3-slot ring: issues W stage nkt-1)
    stage(HC<1>{}, HC<0>{}, HC<0>{}, HC<1>{}, HC<FH<NBW>::VMT3>{}, a1, a0);                       // stage nkt-3
    stage(HC<0>{}, HC<0>{}, HC<0>{}, HC<1>{}, HC<0>{}, a0, a1);    // stage nkt-2
    stage(HC<1>{}, HC<0>{}, HC<0>{}, HC<0>{}, HC<0>{}, a1, a0);    // stage nkt-1

    // ---------------- epilogue: the accumulators hold h = residual + bias + A W^T; they are only READ from here on ----------------
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    FH_BAR();                                                       // every wave is out of the main loop: ring and A buffers idle
    float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
    if constexpr (LN) {
        // gamma and beta rows -> the idle ring (3 pieces of 1 KiB each), landed by the first exchange below
        if (wid < 2) {
            const float* src = wid == 0 ? fp.gamma : fp.beta;
#pragma unroll
            for (int i = 0; i < NBP; ++i) glds16(src + i * 256 + lane * 4, lds_base + (unsigned)(H_GB + wid * HN * 4 + i * 1024));
        }
        // Row statistics in gemm_fr.hip's association, bit for bit: there ONE lane sums the 12 column blocks of a half row in
        // one sequential chain; here a half row is split over waves 2 p (blocks 0..5) and 2 p + 1 (blocks 6..11), so the odd
        // wave CONTINUES the even wave's per-lane partial (handed over through LDS: 512 B per pair), then lane + lane ^ 32,
        // then half 0 + half 1.  Two barriers per pass; the odd waves idle for ~100 VALU instructions per pass.
        // The fused forms are written out: gemm_fr.hip's compiler-contracted ones are v - sum / 768 as ONE fma in the variance
        // pass, the rounded mean in the output pass, fma(sum, 1 / 768, eps) under the rsqrt (contraction depends on basic-block
        // structure, which differs here).
        float rsum[2] = {0.f, 0.f};
        float* part = reinterpret_cast<float*>(smem + H_RED);         // [pair][mb][64 lanes]
        float* red = part + 2 * 2 * 64;                               // [half][64 rows]
        auto row_chain = [&](auto PASS, float (&c2)[2]) {
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    // re-pin the block in its home file: the copy below is then a NEW value that cannot be hoisted above this
                    // statement (un-pinned, hipcc read four AGPR blocks into VGPRs at once right behind the barrier)
                    if (nb < HNA) FH_PIN_A(acca[nb < HNA ? nb : 0][mb]); else FH_PIN_V(accv[nb < HNA ? 0 : nb - HNA][mb]);
                    const f32x16 v = nb < HNA ? acca[nb < HNA ? nb : 0][mb] : accv[nb < HNA ? 0 : nb - HNA][mb];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        if constexpr (decltype(PASS)::value == 0) c2[mb] += v[e];
                        else { const float dl = fmaf(rsum[mb], -(1.0f / HN), v[e]); c2[mb] = fmaf(dl, dl, c2[mb]); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
        auto row_reduce = [&](auto PASS) {
            float c2[2] = {0.f, 0.f};
            if (!(wid & 1)) {
                row_chain(PASS, c2);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) part[((wid >> 1) * 2 + mb) * 64 + lane] = c2[mb];
            }
            if constexpr (decltype(PASS)::value == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // gamma / beta have landed
            __syncthreads();
            if (wid & 1) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) c2[mb] = part[((wid >> 1) * 2 + mb) * 64 + lane];
                row_chain(PASS, c2);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    float t = c2[mb];
                    t += __shfl_xor(t, 32, 64);
                    if (hh == 0) red[(wid >> 1) * HM + mb * 32 + r32] = t;
                }
            }
            __syncthreads();
        };
        row_reduce(HC<0>{});
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int r = mb * 32 + r32;
            rsum[mb] = red[r] + red[HM + r];
            mean[mb] = rsum[mb] * (1.0f / HN);
            asm volatile("" : "+v"(mean[mb]));                        // the ROUNDED mean, never re-fused into a consumer
        }
        __syncthreads();                                              // everyone has read the sums before the next pass overwrites them
        row_reduce(HC<1>{});
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int r = mb * 32 + r32;
            rstd[mb] = rsqrtf(fmaf(red[r] + red[HM + r], 1.0f / HN, 1e-5f));
        }
    }
    // stores (gemm_fr.hip): every output row leaves through a wave-private LDS stage so that the stores are whole 128-B
    // lines: h fp32 (nt), u = LN(h) bf16, optional bf16 copy of h.
    const int cl = wid * WCOLS + 4 * hh;                             // this lane's column origin; + nb * 32 + 8 g
    const float* gl = lgamma + cl;
    const float* bl = lbeta + cl;
    char* hst = smem + wid * 16384;                                 // h stage: [64 rows][128 B]
    char* ust = hst + 8192;                                         // u stage: [64 rows][128 B] = 64 bf16 columns
    const int srow = lane >> 3, sq = lane & 7;                      // read-back: row srow (+ 8 i), 16-B chunk sq
    const int grow0 = m0 + srow;
    float* hrow = (float*)p.out + (size_t)(grow0 < p.M ? grow0 : 0) * p.ldo + wid * WCOLS + sq * 4;
    bf16* urow = fp.u ? fp.u + (size_t)(grow0 < p.M ? grow0 : 0) * fp.ldu + wid * WCOLS + sq * 8 : nullptr;
    bf16* orow = p.out2 ? p.out2 + (size_t)(grow0 < p.M ? grow0 : 0) * p.ldo2 + wid * WCOLS + sq * 8 : nullptr;
    // U8: u is fp8 e4m3, [M, ldu] BYTES; a staged 128-B line holds four column blocks
    unsigned char* u8row = U8 && fp.u ? (unsigned char*)fp.u + (size_t)(grow0 < p.M ? grow0 : 0) * fp.ldu + wid * WCOLS + sq * 16 : nullptr;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            if (nb < HNA) FH_PIN_A(acca[nb < HNA ? nb : 0][mb]); else FH_PIN_V(accv[nb < HNA ? 0 : nb - HNA][mb]);
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
                const f32x16& v = nb < HNA ? acca[nb < HNA ? nb : 0][mb] : accv[nb < HNA ? 0 : nb - HNA][mb];
                const f32x4 v4 = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                const int row = mb * 32 + r32;
                *reinterpret_cast<f32x4*>(hst + row * 128 + (((2 * g + hh) ^ (row & 7)) << 4)) = v4;
                f32x4 y = v4;                                        // bf16 side: LayerNorm output, or the plain copy
                if constexpr (LN) y = (v4 - mean[mb]) * rstd[mb] * g4 + b4;
                if constexpr (U8) {
                    // fp8: column 8 g + 4 hh of block nb is byte (nb & 3) * 32 + 8 g + 4 hh of the 128-column line
                    *reinterpret_cast<unsigned*>(ust + row * 128 + ((((nb & 3) * 2 + (g >> 1)) ^ (row & 7)) << 4) + (g & 1) * 8 + hh * 4) =
                        pack_fp8x4(y[0], y[1], y[2], y[3]);
                } else {
                    u32x2 st;
                    st[0] = pack_bf16x2(y[0], y[1]); st[1] = pack_bf16x2(y[2], y[3]);
                    *reinterpret_cast<u32x2*>(ust + row * 128 + ((((nb & 1) * 4 + g) ^ (row & 7)) << 4) + hh * 8) = st;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i = 4 * half; i < 4 * half + 4; ++i) {
                const int row = srow + 8 * i;
                const u32x4 hv = *reinterpret_cast<const u32x4*>(hst + row * 128 + ((sq ^ (row & 7)) << 4));
                if (grow0 + 8 * i < FH_DIAG_M) store16<true, true>(hrow + (size_t)(8 * i) * p.ldo + nb * 32, hv, 0);   // nt (a run-time plain / nt switch here cost the K = 768 launch 5 us: the branch splits the store block; the A/B itself: no gain from plain stores)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (U8 ? (nb & 3) == 3 : (nb & 1)) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int i = 4 * half; i < 4 * half + 4; ++i) {
                    const int row = srow + 8 * i;
                    const u32x4 uv = *reinterpret_cast<const u32x4*>(ust + row * 128 + ((sq ^ (row & 7)) << 4));
                    if (grow0 + 8 * i < FH_DIAG_M) {
                        if constexpr (U8) *reinterpret_cast<u32x4*>(u8row + (size_t)(8 * i) * fp.ldu + (nb - 3) * 32) = uv;
                        else if (LN) *reinterpret_cast<u32x4*>(urow + (size_t)(8 * i) * fp.ldu + (nb - 1) * 32) = uv;
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

template <int NBW, bool LN, bool RES, bool U8>
hipError_t launch_fr64_t(const FrParams& fp, int grid, hipStream_t s) {
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&gemm_fr64_kernel<NBW, LN, RES, U8>)}, FH<NBW>::LDS)) return e;
    hipLaunchKernelGGL((gemm_fr64_kernel<NBW, LN, RES, U8>), dim3(grid), dim3(256), FH<NBW>::LDS, s, fp);
    return hipGetLastError();
}

}  // namespace

bool gemm_fr64_supports(int M, int N, int K, size_t lda, size_t ldw) {
    if ((N != 768 && N != 1024) || K % 64 || K < 64 || M < HM) return false;
    if ((size_t)M * lda * 2 >= (1ull << 32) || (size_t)N * ldw * 2 >= (1ull << 32)) return false;
    return true;
}

// fp.u_fp8: the LayerNorm output is fp8 e4m3 bytes (N = 1024 only)
hipError_t launch_gemm_fr64(const FrParams& fp_in, hipStream_t s) {
    FrParams fp = fp_in;
    fp.g.tiles_m = (fp.g.M + HM - 1) / HM;
    fp.g.tiles_n = 1;
    const bool ln = fp.gamma && fp.u, res = fp.g.residual != nullptr;
    const int grid = fp.g.tiles_m;
    if (fp.g.N == 1024) {
        fp.stagger_ticks = 0;                // one workgroup per CU
        if (fp.u_fp8) {
            if (!ln) return hipErrorInvalidValue;
            return res ? launch_fr64_t<8, true, true, true>(fp, grid, s) : launch_fr64_t<8, true, false, true>(fp, grid, s);
        }
        if (ln) return res ? launch_fr64_t<8, true, true, false>(fp, grid, s) : launch_fr64_t<8, true, false, false>(fp, grid, s);
        return res ? launch_fr64_t<8, false, true, false>(fp, grid, s) : launch_fr64_t<8, false, false, false>(fp, grid, s);
    }
    if (fp.g.N != 768 || fp.u_fp8) return hipErrorInvalidValue;
    if (ln) return res ? launch_fr64_t<6, true, true, false>(fp, grid, s) : launch_fr64_t<6, true, false, false>(fp, grid, s);
    return res ? launch_fr64_t<6, false, true, false>(fp, grid, s) : launch_fr64_t<6, false, false, false>(fp, grid, s);
}

}  // namespace ditto
