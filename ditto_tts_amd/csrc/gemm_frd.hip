// gemm_frd.hip — the full-row N = 768 GEMM with fused residual + LayerNorm (gemm_fr.hip's contract) with the WEIGHTS
// FETCHED STRAIGHT FROM L2 INTO REGISTERS: no W ring in the LDS, no LDS-DMA and no ds_read for the large operand (gfx950).
//
//     h[M, 768] (fp32, in place) = residual + A[M, K] * W[768, K]^T + bias            (reference DiT.py:148, :155)
//     u[M, 768] (bf16)           = LayerNorm(h) * gamma + beta   (eps 1e-5)           (reference DiT.py:152, :105)
//
// Why: gemm_fr.hip's main loop runs at 0.55 us per K = 16 stage where its 24 MFMAs need 0.40: the 24 KiB of W per stage go
// L2 -> LDS by LDS-DMA (the path sustains ~52 GB/s per CU next to MFMAs and fragment reads) and come back out through 12
// ds_read_b128 per wave and stage behind a barrier per stage.  The weights are packed stage-major, Wp[K/16][768][16], so the
// 1 KiB of a (stage, 32-column block) piece is contiguous and lane (r32, hh) of a v_mfma_f32_32x32x16_bf16 operand wants
// exactly the 16 bytes at r32 * 32 + hh * 16 of it: ONE global_load_dwordx4 per fragment whose wave covers eight whole cache
// lines.  tools/probe_wdirect.hip measured that stream at 105 GB/s per CU alone and at 0.44-0.50 us per stage with the 24 MFMAs,
// two stages of look-ahead being the best depth.  That needs W to be wave-private, hence another wave tile:
//
//   tile      128 rows x 768 columns, one workgroup per CU, 4 waves side by side in N: wave wn owns ALL 128 rows x columns
//             [192 wn, +192) = 4 x 6 blocks of 32 x 32 = 384 accumulators (column blocks 0..3 in AGPRs = 256, 4..5 in VGPRs).
//   W         six fragments per stage and wave, a register ring of two stages (48 VGPRs); fragment nb of stage s+2 is
//             loaded into the registers of fragment nb of stage s right behind the four MFMAs that consumed it.  Every
//             fragment is waited for separately with a counted vmcnt (11 younger loads + the A pieces issued since: a
//             compile-time constant per (slab position, nb)), so each load has its full two stages to land.
//   A         shared by the four waves, through the LDS as in gemm_fr.hip: slabs of 64 k (128 rows x 128 B = 16 pieces of
//             1 KiB, 4 per wave, chunk c of row r at c ^ ((r >> 1) & 7)), double-buffered, ONE barrier per slab; 4
//             ds_read_b128 per wave and stage (gemm_fr.hip: 14).
//   K order   as gemm_fr.hip (rotation per 128-row tile included): h is bit-identical to gemm_fr.hip's.  The LayerNorm
//             statistics are summed per quarter row ((q0 + q1) + (q2 + q3)), so u may differ from gemm_fr.hip's in the last
//             bf16 bit of a few elements.
//   LDS       A 2 x 16 KiB + bias row 3 KiB in the loop; the epilogue stages every output row through a wave-private
//             32 KiB (whole-line stores) + gamma | beta + row statistics: 138 KiB.
#include <type_traits>

#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int DM = 128, DN = 768, DK = 16;
constexpr int D_W_BYTES = DN * DK * 2;              // 24 KiB: one K = 16 stage of W in memory
constexpr int D_ASLAB = DM * 64 * 2;                // 16 KiB: 128 rows x 64 k
constexpr int D_BIAS = 2 * D_ASLAB;                 // bias row (3 KiB) behind the two A slabs
constexpr int D_STAGE = 4 * 32768;                  // epilogue: 4 x 32 KiB of store staging (overlays A and bias) ...
constexpr int D_GB = D_STAGE;                       // ... gamma | beta rows (6 KiB) ...
constexpr int D_RED = D_GB + 2 * DN * 4;            // ... row sums [4 quarters][128 rows] fp32 (2 KiB)
constexpr int D_LDS = D_RED + 4 * DM * 4;           // 138 KiB
constexpr int DNA = 4;                              // column blocks (of 6) whose accumulators live in AGPRs

#ifdef DITTO_DIAG_FR_NOSTORE
#define FD_DIAG_M (p.M - (1 << 30))
#else
#define FD_DIAG_M p.M
#endif
#define FD_BAR() asm volatile("s_barrier" ::: "memory")
#define FD_PIN_A(x) asm volatile("" : "+a"(x))
#define FD_PIN_V(x) asm volatile("" : "+v"(x))

template <int V>
struct DC { static constexpr int value = V; };

DITTO_DEV void dmfma_a(f32x16& c, const f32x4& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
}
DITTO_DEV void dmfma_v(f32x16& c, const f32x4& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(a));
}
// last MFMA of a chain: its wait states inside the statement (hazard argument: gemm_fr.hip)
DITTO_DEV void dmfma_a_last(f32x16& c, const f32x4& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+a"(c) : "v"(w), "v"(a));
}
DITTO_DEV void dmfma_v_last(f32x16& c, const f32x4& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+v"(c) : "v"(w), "v"(a));
}

// asm with operands lives in free functions: inside a generic lambda clang rejects asm operands that name captured locals
template <int IMM>
DITTO_DEV void frd_wload(f32x4& dst, unsigned voff, const char* base) {
#ifndef DITTO_DIAG_FRD_NOW      // tools/build_diag.sh: the main loop without its W stream (timing only, wrong results)
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
#else
    asm volatile("" : "+v"(dst) : "v"(voff), "s"(base));
#endif
}
template <int VM>
DITTO_DEV void frd_wait(f32x4& frag) {   // counted wait that ties the fragment's registers: no use moves above it
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(frag) : "n"(VM) : "memory");
}
DITTO_DEV void frd_dma(unsigned voff, const char* base, unsigned dst) {
#ifndef DITTO_DIAG_FR_NODMA
#ifndef DITTO_FRD_A_POLICY     // A/B builds: cache policy of the A slab stream (tools/build_diag.sh -DDITTO_FRD_A_POLICY='"nt"')
#define DITTO_FRD_A_POLICY ""
#endif
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 " DITTO_FRD_A_POLICY ::"v"(voff), "s"(base), "s"(dst) : "memory");
#endif
}

// Vector-memory operations a wave has issued AFTER the load of W fragment (s, nb) when MFMA (s, nb) is about to issue, i.e.
// what may stay in flight at that wait.  Issue order of a stage at slab position j: W0 W1 W2 W3 W4 [A] W5 [A]  (the two A
// pieces only at j < 2 and while a next slab exists), W = the fragments of stage s + 2 while that stage exists.
//   last: this is the last slab (no A issue; stages 2, 3 issue no W).
constexpr int frd_vm(int j, int nb, bool last) {
    int c = 5 - nb;                                                   // stage s-2: W(s, nb+1 .. 5)
    const int j2 = (j + 2) & 3, j1 = (j + 3) & 3;
    const bool a_prev2 = j >= 2 ? !last : true, a_prev1 = j >= 1 ? !last : true, a_this = !last;
    if (a_prev2 && j2 < 2) c += nb <= 4 ? 2 : 1;                     //           its A pieces behind W4 / W5
    const bool w_prev1 = !(last && j == 3), w_this = !(last && j >= 2);
    if (w_prev1) c += 6;                                              // stage s-1: W(s+1, 0 .. 5)
    if (a_prev1 && j1 < 2) c += 2;
    if (w_this) c += nb;                                              // stage s:   W(s+2, 0 .. nb-1)
    if (a_this && j < 2 && nb == 5) c += 1;                           //            the A piece behind W4
    return c;
}
static_assert(frd_vm(0, 0, false) == 11 && frd_vm(0, 5, false) == 12 && frd_vm(1, 0, false) == 13 && frd_vm(2, 0, false) == 15 &&
              frd_vm(2, 5, false) == 14 && frd_vm(3, 4, false) == 13 && frd_vm(3, 5, false) == 12, "steady-state wait counts");
static_assert(frd_vm(0, 3, true) == 11 && frd_vm(2, 0, true) == 11 && frd_vm(2, 5, true) == 6 && frd_vm(3, 0, true) == 5 &&
              frd_vm(3, 5, true) == 0, "last-slab wait counts");

// HB: the residual stream is bf16 in HBM (fp.hb): residual read and h written as bf16 (fp32 only in the accumulators and the
// LayerNorm, which still sees the UNROUNDED fp32 row); p.residual / p.out point to bf16, ldr / ldo in bf16 elements.
#ifdef DITTO_DIAG_FRD_STAMP   // tools/build_diag_one.sh ... gemm_frd.hip -DDITTO_DIAG_FRD_STAMP: s_memtime stamps around the kernel's phases
__device__ unsigned long long g_frd_stamps[1024 * 4 * 8];
DITTO_DEV unsigned long long frd_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define FRD_STAMP(i) const unsigned long long frd_t##i = frd_now()
#else
#define FRD_STAMP(i)
#endif

// accumulator-init group gi of 24 -> its (nb, mb) block (see the kernel's prologue)
constexpr int frd_group_nb(int gi) {
#if defined(FRD_PAIRED) && FRD_PAIRED
    return 2 * (gi >> 3) + (gi & 1);
#else
    return gi >> 2;
#endif
}
constexpr int frd_group_mb(int gi) {
#if defined(FRD_PAIRED) && FRD_PAIRED
    return (gi >> 1) & 3;
#else
    return gi & 3;
#endif
}

template <bool LN, bool RES, bool HB = false>
__global__ __launch_bounds__(256, 1) void gemm_frd_kernel(FrParams fp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    FRD_STAMP(0);
    const GemmParams& p = fp.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);       // = the wave's column quarter
    const int nkt = p.K / DK;                                        // a multiple of 4 (K % 64 == 0)
    const int nslab = nkt >> 2;
    // XCD-contiguous tiles and the K-loop rotation: exactly gemm_fr.hip's (same sums in the same order)
    const int ntile = gridDim.x;
    const int tile = (ntile & 7) == 0 ? (int)(blockIdx.x & 7) * (ntile >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int m0 = tile * DM;
    const int s0 = fp.rot_period > 0 ? (((tile % fp.rot_period) & 7) * nslab) >> 3 : 0;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;

    // ---- A: LDS-DMA in slabs of 64 k, 4 pieces (8 rows x 128 B) per wave and slab ----
    const int arow = lane >> 3, apos = lane & 7;
    unsigned vak[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (wid * 4 + j) + arow;
        int ar = m0 + row;
        ar = ar < p.M ? ar : p.M - 1;
#ifdef DITTO_DIAG_FR_AHOT      // tools/build_diag.sh: every tile reads the A rows of tile 0 (L2-resident, statistically the same data): what does A's HBM / Infinity Cache latency cost?
        ar = row;
#endif
        vak[j] = (unsigned)(((size_t)ar * p.lda + (apos ^ ((row >> 1) & 7)) * 8) * 2) + (unsigned)(s0 * 128);
    }
    int a_left = nslab - s0;                                          // slabs until the rotated K loop wraps to k = 0
    unsigned a_buf = lds_base;                                        // LDS address of the buffer the next slab goes to
    auto issue_a_piece = [&](auto J) {
        constexpr int j = decltype(J)::value;
        frd_dma(vak[j], (const char*)p.A, a_buf + (unsigned)((wid * 4 + j) * 1024));
    };
    auto advance_a = [&]() {
        a_buf = a_buf == lds_base ? lds_base + D_ASLAB : lds_base;
        --a_left;
        const unsigned inc = a_left == 0 ? 128u - (unsigned)nslab * 128u : 128u;
#pragma unroll
        for (int j = 0; j < 4; ++j) vak[j] += inc;
    };

    // ---- W: straight into registers.  Fragment (stage, nb) of this wave = the 1 KiB at stage * 24 KiB + (6 wn + nb) KiB of the
    //      stage-major image; lane (r32, hh) takes its 16 bytes at r32 * 32 + hh * 16.  One per-lane offset for nb 0..3
    //      (immediate nb * 1024), one 4 KiB further for nb 4, 5; both advance 24 KiB per stage and wrap with the rotation. ----
    const int r32 = lane & 31, hh = lane >> 5;
    unsigned vw0 = (unsigned)(wid * 6 * 1024 + r32 * 32 + hh * 16) + (unsigned)s0 * 4u * D_W_BYTES;
    int w_left = nkt - 4 * s0;
    f32x4 wr[2][6];                                                   // the register ring: [stage & 1][nb]
    auto issue_w = [&](auto NB, f32x4& dst) {
        constexpr int nb = decltype(NB)::value;
        if constexpr (nb < 4) frd_wload<nb * 1024>(dst, vw0, (const char*)p.W);
        else frd_wload<(nb - 4) * 1024>(dst, vw0 + 4096u, (const char*)p.W);
    };
    auto wait_frag = [&](auto VM, f32x4& frag) { frd_wait<decltype(VM)::value>(frag); };
    auto advance_w = [&]() {
        --w_left;
        vw0 += w_left == 0 ? (unsigned)D_W_BYTES - (unsigned)nkt * D_W_BYTES : (unsigned)D_W_BYTES;
    };

    // bias row -> LDS (3 pieces of 1 KiB): the oldest loads of the kernel
    if (wid == 0) {
        if (p.bias) {
#pragma unroll
            for (int i = 0; i < 3; ++i) glds16(p.bias + i * 256 + lane * 4, lds_base + (unsigned)(D_BIAS + i * 1024));
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(smem + D_BIAS + i * 1024 + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // A slab 0 (in flight while the accumulators are initialised)
    issue_a_piece(DC<0>{}); issue_a_piece(DC<1>{}); issue_a_piece(DC<2>{}); issue_a_piece(DC<3>{});
    advance_a();

    // A fragment addressing: row (32 mb + r32) x 128 B; stage j of the slab = 16-B chunks 2 j + hh
    const int a_row = r32 * 128;                                      // + mb * 4096
    const int a_x = (hh ^ ((r32 >> 1) & 7)) << 4;                     // ^ (j << 5)

    // ---- accumulators START as bias + residual (the epilogue only READS them): 96 hand-written global_load_dwordx4 per lane
    //      in the accumulator layout, four 32 x 32 blocks (16 loads) in flight ----
    const float* lbias = reinterpret_cast<const float*>(smem + D_BIAS);
    const float* lgamma = reinterpret_cast<const float*>(smem + D_GB);     // these two: valid in the epilogue only
    const float* lbeta = lgamma + DN;
    f32x16 acca[DNA][4], accv[6 - DNA][4];
    {
        const char* rp[4] = {nullptr, nullptr, nullptr, nullptr};
        if constexpr (RES) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                int gr = m0 + mb * 32 + r32;
                gr = gr < p.M ? gr : p.M - 1;
                if constexpr (HB) rp[mb] = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(p.residual) + (size_t)gr * p.ldr + wid * 192 + 4 * hh);
                else rp[mb] = reinterpret_cast<const char*>(p.residual + (size_t)gr * p.ldr + wid * 192 + 4 * hh);
            }
        }
#ifndef FRD_WD
#define FRD_WD 4
#endif
#ifndef FRD_PAIRED
#define FRD_PAIRED 0
#endif
        // window depth (groups of four loads in flight per wave); 24 groups = (nb, mb) blocks.  Order: mb fastest (FRD_PAIRED 0),
        // or the two column blocks that share a row's 128-byte line back to back (bf16 stream: 2 x 64 B; FRD_PAIRED 1).
        // Round 5 A/B (-DFRD_WD=8|12 -DFRD_PAIRED=1, profiles/r05_frd_window_ab.txt): out-proj 65.6 -> 64.2 .. 65.4 us, fc2 137.5 ->
        // 136.2 .. 137.0: inside the noise — the 17 us of this phase are not a too-small load window; defaults kept.
        constexpr int WD = HB ? FRD_WD : 4, NG = 24;                 // (the fp32 form: 16-byte loads, twice the window registers)
        static_assert(WD == 4 || WD == 8 || WD == 12, "window depth");
        using res_t = typename std::conditional<HB, u32x2, f32x4>::type;   // four columns of one row: bf16 x 4 or fp32 x 4
        res_t T[WD][4];
        auto issue_group = [&](auto GI, res_t (&t)[4]) {
            constexpr int nb = frd_group_nb(decltype(GI)::value), mb = frd_group_mb(decltype(GI)::value);
#ifdef DITTO_DIAG_FRD_NORES   // stamps only: the accumulators start from the bias alone (valid numbers, WRONG results) — what do the residual loads cost?
            if constexpr (true) {
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = res_t{};
            } else
#endif
            if constexpr (RES && HB) {
                const char* ptr = rp[mb];
                asm volatile("global_load_dwordx2 %0, %4, off offset:%5\n\t"
                             "global_load_dwordx2 %1, %4, off offset:%6\n\t"
                             "global_load_dwordx2 %2, %4, off offset:%7\n\t"
                             "global_load_dwordx2 %3, %4, off offset:%8"
                             : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                             : "v"(ptr), "n"(nb * 64), "n"(nb * 64 + 16), "n"(nb * 64 + 32), "n"(nb * 64 + 48)
                             : "memory");
            } else if constexpr (RES) {
                const char* ptr = rp[mb];
                asm volatile("global_load_dwordx4 %0, %4, off offset:%5\n\t"
                             "global_load_dwordx4 %1, %4, off offset:%6\n\t"
                             "global_load_dwordx4 %2, %4, off offset:%7\n\t"
                             "global_load_dwordx4 %3, %4, off offset:%8"
                             : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                             : "v"(ptr), "n"(nb * 128), "n"(nb * 128 + 32), "n"(nb * 128 + 64), "n"(nb * 128 + 96)
                             : "memory");
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = res_t{};
            }
        };
        auto finish_group = [&](auto GI, res_t (&t)[4]) {
            constexpr int gi = decltype(GI)::value, nb = frd_group_nb(gi), mb = frd_group_mb(gi);
            constexpr int younger = NG - 1 - gi < WD - 1 ? NG - 1 - gi : WD - 1;
#ifndef DITTO_DIAG_FRD_NORES
            if constexpr (RES)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) : "n"(4 * younger) : "memory");
#else
            if constexpr (gi == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            if constexpr (gi == 0) {
                if constexpr (!RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                FD_BAR();      // every wave is past a wait that covers wave 0's bias row (the oldest load): visible to all
            }
            f32x16 v;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbias + wid * 192 + nb * 32 + 8 * g + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float r;
                    if constexpr (HB) r = __builtin_bit_cast(float, (e & 1) ? (t[g][e >> 1] & 0xFFFF0000u) : (t[g][e >> 1] << 16));
                    else r = t[g][e];
                    v[4 * g + e] = r + b4[e];
                }
            }
            if constexpr (nb < DNA) { acca[nb < DNA ? nb : 0][mb] = v; FD_PIN_A(acca[nb < DNA ? nb : 0][mb]); }
            else { accv[nb < DNA ? 0 : nb - DNA][mb] = v; FD_PIN_V(accv[nb < DNA ? 0 : nb - DNA][mb]); }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto init_step = [&](auto GI) {
            constexpr int gi = decltype(GI)::value;
            finish_group(GI, T[gi % WD]);
            if constexpr (gi + WD < NG) issue_group(DC<gi + WD>{}, T[gi % WD]);
        };
        issue_group(DC<0>{}, T[0]); issue_group(DC<1>{}, T[1]); issue_group(DC<2>{}, T[2]); issue_group(DC<3>{}, T[3]);
        if constexpr (WD > 4) { issue_group(DC<4>{}, T[4 % WD]); issue_group(DC<5>{}, T[5 % WD]); issue_group(DC<6>{}, T[6 % WD]); issue_group(DC<7>{}, T[7 % WD]); }
        if constexpr (WD > 8) { issue_group(DC<8>{}, T[8 % WD]); issue_group(DC<9>{}, T[9 % WD]); issue_group(DC<10>{}, T[10 % WD]); issue_group(DC<11>{}, T[11 % WD]); }
        init_step(DC<0>{}); init_step(DC<1>{}); init_step(DC<2>{}); init_step(DC<3>{}); init_step(DC<4>{}); init_step(DC<5>{});
        init_step(DC<6>{}); init_step(DC<7>{}); init_step(DC<8>{}); init_step(DC<9>{}); init_step(DC<10>{}); init_step(DC<11>{});
        init_step(DC<12>{}); init_step(DC<13>{}); init_step(DC<14>{}); init_step(DC<15>{}); init_step(DC<16>{}); init_step(DC<17>{});
        init_step(DC<18>{}); init_step(DC<19>{}); init_step(DC<20>{}); init_step(DC<21>{}); init_step(DC<22>{}); init_step(DC<23>{});
    }

    // A slab 0 has landed for this wave (older than the residual loads); for everyone:
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FD_BAR();
    FRD_STAMP(1);
    // W stages 0 and 1 -> the register ring (twelve loads, nothing else behind them: the wait counts of frd_vm start here)
    issue_w(DC<0>{}, wr[0][0]); issue_w(DC<1>{}, wr[0][1]); issue_w(DC<2>{}, wr[0][2]);
    issue_w(DC<3>{}, wr[0][3]); issue_w(DC<4>{}, wr[0][4]); issue_w(DC<5>{}, wr[0][5]);
    advance_w();
    issue_w(DC<0>{}, wr[1][0]); issue_w(DC<1>{}, wr[1][1]); issue_w(DC<2>{}, wr[1][2]);
    issue_w(DC<3>{}, wr[1][3]); issue_w(DC<4>{}, wr[1][4]); issue_w(DC<5>{}, wr[1][5]);
    advance_w();
    unsigned a_cur = 0;          // byte offset of the A slab being multiplied
    bf16x8 a0[4], a1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) a0[mb] = *reinterpret_cast<const bf16x8*>(smem + a_cur + a_row + mb * 4096 + a_x);

    // One stage = K 16 at slab position J.  LAST: the last slab (no next slab to fetch; its stages 2, 3 fetch no W).
    // ACUR: this stage's A fragments (resident), ANXT receives the next stage's.
    auto stage = [&](auto J, auto LAST, bf16x8 (&ACUR)[4], bf16x8 (&ANXT)[4]) {
        constexpr int j = decltype(J)::value;
        constexpr bool last = decltype(LAST)::value != 0;
        constexpr bool do_w = !(last && j >= 2), do_a = !last && j < 2, has_next = !(last && j == 3);
        constexpr int slot = j & 1;
        const unsigned a_nxt = j == 3 ? (unsigned)D_ASLAB - a_cur : a_cur;
#pragma unroll
        for (int nb = 0; nb < 6; ++nb) {
            if (nb == 4 && has_next) {
                if constexpr (j == 3) {
                    // the next slab: this wave's four pieces have landed (issued at j = 0, 1; 6 + 4 loads since), everyone's
                    // behind the barrier, which also certifies that every wave is done reading the slab the NEXT issue overwrites
                    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
#ifndef DITTO_DIAG_FR_NOBAR
                    FD_BAR();
#endif
                }
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    ANXT[mb] = *reinterpret_cast<const bf16x8*>(smem + a_nxt + a_row + mb * 4096 + (a_x ^ (((j + 1) & 3) << 5)));
            }
            // fragment (s, nb) has landed: everything but the frd_vm(j, nb) operations issued after it
            if (nb == 0) wait_frag(DC<frd_vm(j, 0, last)>{}, wr[slot][0]);
            if (nb == 1) wait_frag(DC<frd_vm(j, 1, last)>{}, wr[slot][1]);
            if (nb == 2) wait_frag(DC<frd_vm(j, 2, last)>{}, wr[slot][2]);
            if (nb == 3) wait_frag(DC<frd_vm(j, 3, last)>{}, wr[slot][3]);
            if (nb == 4) wait_frag(DC<frd_vm(j, 4, last)>{}, wr[slot][4]);
            if (nb == 5) wait_frag(DC<frd_vm(j, 5, last)>{}, wr[slot][5]);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                if constexpr (has_next) {
                    if (nb < DNA) dmfma_a(acca[nb < DNA ? nb : 0][mb], wr[slot][nb], ACUR[mb]);
                    else dmfma_v(accv[nb < DNA ? 0 : nb - DNA][mb], wr[slot][nb], ACUR[mb]);
                } else {
                    if (nb < DNA) dmfma_a_last(acca[nb < DNA ? nb : 0][mb], wr[slot][nb], ACUR[mb]);
                    else dmfma_v_last(accv[nb < DNA ? 0 : nb - DNA][mb], wr[slot][nb], ACUR[mb]);
                }
            }
            if constexpr (do_w) {   // fragment nb of stage s + 2 into the registers just consumed
                if (nb == 0) issue_w(DC<0>{}, wr[slot][0]);
                if (nb == 1) issue_w(DC<1>{}, wr[slot][1]);
                if (nb == 2) issue_w(DC<2>{}, wr[slot][2]);
                if (nb == 3) issue_w(DC<3>{}, wr[slot][3]);
                if (nb == 4) issue_w(DC<4>{}, wr[slot][4]);
                if (nb == 5) issue_w(DC<5>{}, wr[slot][5]);
            }
            if constexpr (do_a) {   // half of the next slab's four pieces, behind W4 and W5
                if (nb == 4) issue_a_piece(DC<2 * j>{});
                if (nb == 5) issue_a_piece(DC<2 * j + 1>{});
            }
        }
        if constexpr (do_w) advance_w();
        if constexpr (do_a && j == 1) advance_a();
        if constexpr (j == 3) a_cur = a_nxt;
    };
    for (int sl = 0; sl + 1 < nslab; ++sl) {
        stage(DC<0>{}, DC<0>{}, a0, a1);
        stage(DC<1>{}, DC<0>{}, a1, a0);
        stage(DC<2>{}, DC<0>{}, a0, a1);
        stage(DC<3>{}, DC<0>{}, a1, a0);
    }
    stage(DC<0>{}, DC<1>{}, a0, a1);
    stage(DC<1>{}, DC<1>{}, a1, a0);
    stage(DC<2>{}, DC<1>{}, a0, a1);
    stage(DC<3>{}, DC<1>{}, a1, a0);

    // ---------------- epilogue: the accumulators hold h = residual + bias + A W^T; they are only READ from here on ----------------
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    FD_BAR();                                                       // every wave is out of the main loop: the A slabs are idle
    FRD_STAMP(2);
    float mean[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (LN) {
        if (wid < 2) {   // gamma and beta rows -> LDS (3 pieces of 1 KiB each), landed by the first exchange below
            const float* src = wid == 0 ? fp.gamma : fp.beta;
#pragma unroll
            for (int i = 0; i < 3; ++i) glds16(src + i * 256 + lane * 4, lds_base + (unsigned)(D_GB + wid * DN * 4 + i * 1024));
        }
        // Row statistics, two passes like nn.LayerNorm: per lane one chain over this wave's quarter row, + lane ^ 32, then
        // (q0 + q1) + (q2 + q3) through LDS.  The fused forms are written out (contraction would follow basic-block structure).
        float rsum[4] = {0.f, 0.f, 0.f, 0.f};
        float* red = reinterpret_cast<float*>(smem + D_RED);         // [quarter][128 rows]
        auto row_pass = [&](auto PASS, float (&c4)[4]) {
#pragma unroll
            for (int nb = 0; nb < 6; ++nb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    // re-pin the block in its home file: the copy below is then a NEW value that cannot be hoisted above this
                    if (nb < DNA) FD_PIN_A(acca[nb < DNA ? nb : 0][mb]); else FD_PIN_V(accv[nb < DNA ? 0 : nb - DNA][mb]);
                    const f32x16 v = nb < DNA ? acca[nb < DNA ? nb : 0][mb] : accv[nb < DNA ? 0 : nb - DNA][mb];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        if constexpr (decltype(PASS)::value == 0) c4[mb] += v[e];
                        else { const float dl = fmaf(rsum[mb], -(1.0f / DN), v[e]); c4[mb] = fmaf(dl, dl, c4[mb]); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float t = c4[mb];
                t += __shfl_xor(t, 32, 64);
                if (hh == 0) red[wid * DM + mb * 32 + r32] = t;
            }
        };
        float c4[4] = {0.f, 0.f, 0.f, 0.f};
        row_pass(DC<0>{}, c4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // gamma / beta have landed
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int r = mb * 32 + r32;
            rsum[mb] = (red[r] + red[DM + r]) + (red[2 * DM + r] + red[3 * DM + r]);
            mean[mb] = rsum[mb] * (1.0f / DN);
            asm volatile("" : "+v"(mean[mb]));                        // the ROUNDED mean, never re-fused into a consumer
        }
        __syncthreads();                                              // everyone has read the sums before the next pass overwrites them
        float q4[4] = {0.f, 0.f, 0.f, 0.f};
        row_pass(DC<1>{}, q4);
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int r = mb * 32 + r32;
            rstd[mb] = rsqrtf(fmaf((red[r] + red[DM + r]) + (red[2 * DM + r] + red[3 * DM + r]), 1.0f / DN, 1e-5f));
        }
    }
    FRD_STAMP(3);
    // stores (gemm_fr.hip): every output row leaves through a wave-private LDS stage so that the stores are whole 128-B lines:
    // h fp32 (nt), u = LN(h) bf16, optional bf16 copy of h.
    const int cl = wid * 192 + 4 * hh;                             // this lane's column origin; + nb * 32 + 8 g
    const float* gl = lgamma + cl;
    const float* bl = lbeta + cl;
    char* hst = smem + wid * 32768;                                 // h stage: [128 rows][128 B]
    char* ust = hst + 16384;                                        // u stage: [128 rows][128 B] = 64 bf16 columns
    const int srow = lane >> 3, sq = lane & 7;                      // read-back: row srow (+ 8 i), 16-B chunk sq
    const int grow0 = m0 + srow;
    float* hrow = (float*)p.out + (size_t)(grow0 < p.M ? grow0 : 0) * p.ldo + wid * 192 + sq * 4;
    bf16* hbrow = (bf16*)p.out + (size_t)(grow0 < p.M ? grow0 : 0) * p.ldo + wid * 192 + sq * 8;   // HB
    bf16* urow = fp.u ? fp.u + (size_t)(grow0 < p.M ? grow0 : 0) * fp.ldu + wid * 192 + sq * 8 : nullptr;
    bf16* orow = p.out2 ? p.out2 + (size_t)(grow0 < p.M ? grow0 : 0) * p.ldo2 + wid * 192 + sq * 8 : nullptr;
#pragma unroll
    for (int nb = 0; nb < 6; ++nb) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            if (nb < DNA) FD_PIN_A(acca[nb < DNA ? nb : 0][mb]); else FD_PIN_V(accv[nb < DNA ? 0 : nb - DNA][mb]);
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
            for (int mb = 0; mb < 4; ++mb) {
                const f32x16& v = nb < DNA ? acca[nb < DNA ? nb : 0][mb] : accv[nb < DNA ? 0 : nb - DNA][mb];
                const f32x4 v4 = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                const int row = mb * 32 + r32;
                if constexpr (HB) {                                  // h as bf16: the same [128 rows][64 columns] image as u
                    u32x2 sh;
                    sh[0] = pack_bf16x2(v4[0], v4[1]); sh[1] = pack_bf16x2(v4[2], v4[3]);
                    *reinterpret_cast<u32x2*>(hst + row * 128 + ((((nb & 1) * 4 + g) ^ (row & 7)) << 4) + hh * 8) = sh;
                } else {
                    *reinterpret_cast<f32x4*>(hst + row * 128 + (((2 * g + hh) ^ (row & 7)) << 4)) = v4;
                }
                if constexpr (LN || !HB) {
                    f32x4 y = v4;                                    // bf16 side: LayerNorm output, or the plain copy
                    if constexpr (LN) y = (v4 - mean[mb]) * rstd[mb] * g4 + b4;
                    u32x2 st;
                    st[0] = pack_bf16x2(y[0], y[1]); st[1] = pack_bf16x2(y[2], y[3]);
                    *reinterpret_cast<u32x2*>(ust + row * 128 + ((((nb & 1) * 4 + g) ^ (row & 7)) << 4) + hh * 8) = st;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!HB) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 4 * q; i < 4 * q + 4; ++i) {
                    const int row = srow + 8 * i;
                    const u32x4 hv = *reinterpret_cast<const u32x4*>(hst + row * 128 + ((sq ^ (row & 7)) << 4));
                    if (grow0 + 8 * i < FD_DIAG_M) store16<true, true>(hrow + (size_t)(8 * i) * p.ldo + nb * 32, hv, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (nb & 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 4 * q; i < 4 * q + 4; ++i) {
                    const int row = srow + 8 * i;
                    const u32x4 hv = *reinterpret_cast<const u32x4*>(hst + row * 128 + ((sq ^ (row & 7)) << 4));
                    if (grow0 + 8 * i < FD_DIAG_M) *reinterpret_cast<u32x4*>(hbrow + (size_t)(8 * i) * p.ldo + (nb - 1) * 32) = hv;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if ((nb & 1) && (LN || !HB)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 4 * q; i < 4 * q + 4; ++i) {
                    const int row = srow + 8 * i;
                    const u32x4 uv = *reinterpret_cast<const u32x4*>(ust + row * 128 + ((sq ^ (row & 7)) << 4));
                    if (grow0 + 8 * i < FD_DIAG_M) {
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
#ifdef DITTO_DIAG_FRD_STAMP
    {
        FRD_STAMP(4);                                                // (stores issued)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FRD_STAMP(5);                                                // (stores drained)
        const int w = blockIdx.x * 4 + wid;
#ifdef DITTO_DIAG_FRD_STAMP_K   // record only the launches of this depth WITH a LayerNorm (in-model stamps: tools/frd_stamps_model.py)
        if (p.K == DITTO_DIAG_FRD_STAMP_K && LN)
#endif
        if (lane == 0 && w < 1024 * 4) {
            g_frd_stamps[w * 8 + 0] = frd_t1 - frd_t0; g_frd_stamps[w * 8 + 1] = frd_t2 - frd_t1; g_frd_stamps[w * 8 + 2] = frd_t3 - frd_t2;
            g_frd_stamps[w * 8 + 3] = frd_t4 - frd_t3; g_frd_stamps[w * 8 + 4] = frd_t5 - frd_t4; g_frd_stamps[w * 8 + 5] = 1;
        }
    }
#endif
}

template <bool LN, bool RES, bool HB = false>
hipError_t launch_frd_t(const FrParams& fp, int grid, hipStream_t s) {
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&gemm_frd_kernel<LN, RES, HB>)}, D_LDS)) return e;
    hipLaunchKernelGGL((gemm_frd_kernel<LN, RES, HB>), dim3(grid), dim3(256), D_LDS, s, fp);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm_frd(const FrParams& fp_in, hipStream_t s) {
    FrParams fp = fp_in;
    if (fp.g.N != DN || fp.u_fp8) return hipErrorInvalidValue;
    fp.g.tiles_m = (fp.g.M + DM - 1) / DM;
    fp.g.tiles_n = 1;
    const bool ln = fp.gamma && fp.u, res = fp.g.residual != nullptr;
    if (fp.hb) {   // bf16 residual stream (the model path's two launches: always with a residual), or — no residual, no LayerNorm —
                   // a plain product with a bf16 result (the training backward's long-K dgrads)
        if (fp.g.out2 || (!res && ln)) return hipErrorInvalidValue;
        if (!res) return launch_frd_t<false, false, true>(fp, fp.g.tiles_m, s);
        return ln ? launch_frd_t<true, true, true>(fp, fp.g.tiles_m, s) : launch_frd_t<false, true, true>(fp, fp.g.tiles_m, s);
    }
    if (ln) return res ? launch_frd_t<true, true>(fp, fp.g.tiles_m, s) : launch_frd_t<true, false>(fp, fp.g.tiles_m, s);
    return res ? launch_frd_t<false, true>(fp, fp.g.tiles_m, s) : launch_frd_t<false, false>(fp, fp.g.tiles_m, s);
}

#ifdef DITTO_DIAG_FRD_STAMP
}  // namespace ditto
extern "C" int ditto_diag_frd_stamps(unsigned long long* out, int n) {   // raw per-wave records of the LAST launch (diagnostic build only)
    if (n > 1024 * 4 * 8) n = 1024 * 4 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ditto::g_frd_stamps), (size_t)n * 8) != hipSuccess;
}
namespace ditto {
#endif

}  // namespace ditto
