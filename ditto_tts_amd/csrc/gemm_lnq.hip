// gemm_lnq.hip — LayerNorm (norm2) fused INTO the cross-attention q-projection, d = 768 or 1024 (gfx950).
//
//     u = LayerNorm(h) * gamma + beta        (reference src/components/DiT.py:142-143; eps 1e-5, never written to HBM)
//     q[M, 768] (bf16) = u W_q^T + b_q       (the q third of nn.MultiheadAttention's in-projection, :144-148 -> torch
//                                             functional.py; scale * log2(e) is folded into W_q / b_q at model creation)
//
// Before: ln_kernel (h -> u, 27 us, 150 MB) + a tiled d x d GEMM (u -> q, 48 us) = two launches, 250 MB, for 38.7 GFLOP.
// Here a workgroup owns 64 WHOLE rows: one pass over h leaves the 64 normalised rows in the LDS as the bf16 A operand
// (64 x 768 x 2 B = 96 KiB, resident for the whole K loop: the main loop carries no A traffic at all), W_q streams
// straight from L2 into registers as in gemm_frd.hip (stage-major image, one global_load_dwordx4 per MFMA fragment, a
// register ring behind counted vmcnt waits — the only vector-memory operations of the loop, so the count is a constant),
// and the output tile leaves through the same 96 KiB as whole 128-B lines.
//
//   tile      64 rows x 768 columns, one workgroup (4 waves, one per SIMD) per CU; wave wn owns all 64 rows x columns
//             [192 wn, +192): 192 accumulators, all in AGPRs.  M = 32768: 512 tiles = two rounds of the 256 CUs.
//   SHAPE 32  v_mfma_f32_32x32x16_bf16, weights as the instruction's A operand (the accumulator is C^T: a lane owns 4
//             consecutive output columns of one row): 2 x 6 blocks, K = 16 per stage, 48 stages, ring of 4 (or 8) stages.
//   SHAPE 16  v_mfma_f32_16x16x32_bf16, same roles: 4 x 12 blocks, K = 32 per stage, 24 stages, ring of 2 (or 4) stages — the
//             same bytes, MFMA cycles and look-ahead; the A/B of VERDICT r3 item 3 (the chip can hold a higher
//             clock on one MFMA shape than on the other: MI355X_MICROARCH.md, DVFS give-back item 7).
//   W image   SHAPE 32: Wp[K/16][768][16] (launch_repack_bf16_stage_major, group 16); SHAPE 16: Wp[K/32][768][32]
//             (group 32).  Fragment (stage, column block) of a wave = 1 contiguous KiB; a lane takes 16 bytes of it.
//   A image   [64 rows][1536 B], 16-byte chunk c of row r at c ^ (r & 15) (low four bits: a 256-B span holds 16 chunks =
//             all 64 banks, and row strides are multiples of 256 B): conflict-free for the LayerNorm's ds_write_b64, the
//             fragment ds_read_b128 of both shapes, the epilogue's ds_write_b64 and its row-contiguous read-back.
//   LayerNorm one wave per row, the row in registers, statistics in ln_kernel's order (rowwise.hip): for the same h the
//             normalised row is the LayerNorm kernel's bit for bit.  h is fp32 or bf16 (the bf16 residual stream).
#include <type_traits>

#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int QM = 64;
// width D = N = K of the projection: 768 (DiTTO-S) or 1024 (BASELINE config C5, 32x32x16 only)
template <int D>
struct QW {
    static constexpr int AROW = D * 2;             // bytes per A row (1536 / 2048: multiples of 256, the swizzle's span)
    static constexpr int A = QM * AROW;            // 96 / 128 KiB: the normalised rows, later the output tile
    static constexpr int BIAS = A;                 // bias row (3 / 4 KiB) behind it
    static constexpr int LDS = A + D * 4;          // 99 / 132 KiB
    static constexpr int CH = D / 256;             // f32x4 per lane of a row (one wave per row)
};

template <int V>
struct QC { static constexpr int value = V; };

#ifdef DITTO_DIAG_LNQ_STAMP   // tools/build_diag_one.sh ... gemm_lnq.hip -DDITTO_DIAG_LNQ_STAMP: s_memtime stamps around the kernel's phases
__device__ unsigned long long g_lnq_stamps[2048 * 4 * 8];
DITTO_DEV unsigned long long lnq_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define LNQ_STAMP(i) const unsigned long long lnq_t##i = lnq_now()
#else
#define LNQ_STAMP(i)
#endif

// NW = waves per workgroup, side by side in N: 4 (one per SIMD, the whole register file each) or 8 (two per SIMD, round 5)
template <int SHAPE, int D, int NW>
struct Geo;
template <int D, int NW>
struct Geo<32, D, NW> {
    static constexpr int NBW = D / NW / 32, MBW = 2, KS = 16, NKT = D / 16, PER = 8;   // PER: stages per A-swizzle period (one 256-B span)
    using acc_t = f32x16;
};
template <int D, int NW>
struct Geo<16, D, NW> {
    static constexpr int NBW = D / NW / 16, MBW = 4, KS = 32, NKT = D / 32, PER = 4;
    using acc_t = f32x4;
};

DITTO_DEV void q_mfma(f32x16& c, const f32x4& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
}
DITTO_DEV void q_mfma(f32x4& c, const f32x4& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
}
// last MFMA of a chain: its wait states inside the statement (hipcc knows nothing about an asm producer: gemm_fr.hip)
DITTO_DEV void q_mfma_last(f32x16& c, const f32x4& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+a"(c) : "v"(w), "v"(a));
}
DITTO_DEV void q_mfma_last(f32x4& c, const f32x4& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 9" : "+a"(c) : "v"(w), "v"(a));
}
template <int IMM, bool PROLOGUE = false>
DITTO_DEV void q_wload(f32x4& dst, unsigned voff, const char* base) {
#ifdef DITTO_DIAG_LNQ_NOW   // tools/build_diag_one.sh: the kernel WITHOUT its weight stream — only the ring's prologue
                            // loads happen, every later stage multiplies the fragments the registers already hold (valid weights, so
                            // no NaN garbage speeds the rest of the step up).  TIMING ONLY: LayerNorm, LDS reads, MFMAs, epilogue alone
    if constexpr (!PROLOGUE) { asm volatile("" : "+v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory"); return; }
#endif
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
template <int VM>
DITTO_DEV void q_wait(f32x4& frag) {   // counted wait that ties the fragment's registers: no use moves above it
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(frag) : "n"(VM) : "memory");
}

// Loads a wave has issued AFTER the load of W fragment (s, nb) when MFMA (s, nb) is about to issue (issue order: the ring
// prologue stage by stage, then fragment nb of stage s + R right behind the MFMAs of (s, nb)); `rem` = stages after s.
constexpr int lnq_vm(int NBW, int R, int nb, int rem) {
    const int full = rem < R - 1 ? rem : R - 1;          // whole stages s+1 .. s+R-1 that exist
    return (NBW - 1 - nb) + full * NBW + (rem >= R ? nb : 0);
}
static_assert(lnq_vm(6, 4, 0, 47) == 23 && lnq_vm(6, 4, 5, 10) == 23 && lnq_vm(6, 4, 0, 3) == 23 && lnq_vm(6, 4, 5, 3) == 18 &&
              lnq_vm(6, 4, 0, 0) == 5 && lnq_vm(6, 4, 5, 0) == 0 && lnq_vm(12, 2, 3, 9) == 23 && lnq_vm(12, 2, 11, 1) == 12 &&
              lnq_vm(12, 2, 11, 0) == 0, "wait counts");

struct LnqParams {
    const void* h; int ldh;                 // the rows to normalise, fp32 or bf16 (XB)
    const float* gamma; const float* beta;  // of the LayerNorm in FRONT of the product
    const char* Wp;                         // stage-major weight image of the SHAPE
    const float* bias;
    bf16* out; int ldo;                     // the product
    int M;
    int rot_period;                         // > 0: tiles t and t + rot_period start their K loop at the same place (tiles per utterance)
};

// R: depth of the W register ring in stages.  What the ring holds in flight per CU (4 waves x R x NBW KiB) against the L2's
// latency under load is what paces the loop: the weights of a 64-row tile are 1.18 MB, twice the bytes per MFMA of gemm_frd's
// 128-row tile.
// NW = 8 (round 5): two waves per SIMD.  The stamp timeline (profiles/r05_lnq_stamps_final.txt) has the 4-wave kernel spend 37 % of
// a tile normalising its 64 rows and 21 % in its epilogue — both vector work that ONE wave per SIMD issues at 4+ cycles per
// instruction with every latency exposed — against 42 % in the MFMA loop (82 % of it MFMA issue), and the no-weight-stream
// knock-out prices the W stream at 7 % (profiles/r05_floor_diag2.txt): not the bound round 4 took it for.  Eight waves split N
// eight ways (96 columns, 96 accumulators each), so no weight byte is fetched twice and every output element is the same K-ordered
// chain (bit-identical to NW = 4); each wave normalises 8 rows instead of 16 and two waves share a SIMD's vector pipe.
// (Round 5 also ran this tile loop as the cross out-projection + residual + LayerNorm, "MODE 1": 74.0 us against gemm_frd.hip's 65.8 in
// the model, profiles/r05_step_ab_frq.txt; deleted in round 6.)
template <int SHAPE, int R, bool XB, int D = 768, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW / 4) void gemm_lnq_kernel(LnqParams p) {
    static_assert(NW == 4 || NW == 8, "waves per workgroup");
    constexpr int RPW = QM / NW;                                         // rows a wave normalises (16 / 8)
    using G = Geo<SHAPE, D, NW>;
    constexpr int Q_AROW = QW<D>::AROW, Q_BIAS = QW<D>::BIAS, CH = QW<D>::CH, QN = D, QKD = D;
    static_assert(G::NBW <= 12, "fragment list");
    using acc_t = typename G::acc_t;
    constexpr int NBW = G::NBW, MBW = G::MBW, NKT = G::NKT, PER = G::PER;
    constexpr int NPER = NKT / PER;                                       // 6 (8 at D = 1024) periods of one 256-B span of A each
    static_assert(NKT % PER == 0 && PER % R == 0 && R * NBW - 1 <= 63, "period structure / vmcnt range");
    constexpr int W_STAGE = QN * G::KS * 2;                              // bytes of one stage of W (24 / 48 KiB)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    LNQ_STAMP(0);
    // XCD-contiguous tiles (blocks b and b + 8 share an XCD: neighbouring tiles, i.e. neighbouring K-loop phases, on one L2)
    const int ntile = gridDim.x;
    const int tile = (ntile & 7) == 0 ? (int)(blockIdx.x & 7) * (ntile >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int m0 = tile * QM;
    // K-loop rotation by whole periods: the workgroups of an XCD do not all ask their L2 for the same weight lines at once.
    // A function of the tile's place INSIDE its utterance, so an utterance's bits do not depend on its place in the batch.
    const int p0 = p.rot_period > 0 ? (tile % p.rot_period) % NPER : 0;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;

    // bias row -> LDS (3 pieces of 1 KiB), wave 3; landed and visible behind the barrier that ends the LayerNorm
    if (wid == 3) {
        if (p.bias) {
#pragma unroll
            for (int i = 0; i < CH; ++i) glds16(p.bias + i * 256 + lane * 4, lds_base + (unsigned)(Q_BIAS + i * 1024));
        } else {
#pragma unroll
            for (int i = 0; i < CH; ++i) *reinterpret_cast<f32x4*>(smem + Q_BIAS + i * 1024 + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- W: straight into registers.  (The ring's first loads go out AFTER the LayerNorm: an asm load issued between a
    //      compiler-visible load and the compiler's own counted wait for it would make that wait too short.)
    // fragment (stage, nb) of this wave: stage * W_STAGE + (NBW wn + nb) KiB; the lane's 16 bytes inside the KiB:
    //   SHAPE 32: row (lane & 31) x 32 B + (lane >> 5) x 16;   SHAPE 16: row (lane & 15) x 64 B + (lane >> 4) x 16
    const unsigned vw = (unsigned)(wid * NBW * 1024) +
                        (SHAPE == 32 ? (unsigned)((lane & 31) * 32 + (lane >> 5) * 16) : (unsigned)((lane & 15) * 64 + (lane >> 4) * 16));
    const char* wbase = p.Wp + (size_t)p0 * PER * W_STAGE;               // wave-uniform, advanced by W_STAGE per stage issued
    int w_to_wrap = (NPER - p0) * PER;                                    // stages until the rotated loop wraps to k = 0
    auto advance_w = [&]() {
        wbase += W_STAGE;
        if (--w_to_wrap == 0) wbase = p.Wp;
    };
    f32x4 wr[R][NBW];
    auto issue_w = [&](auto NB, f32x4& dst) {
        constexpr int nb = decltype(NB)::value;
        q_wload<(nb & 3) * 1024>(dst, vw + (unsigned)((nb >> 2) * 4096), wbase);
    };
    auto issue_wp = [&](auto NB, f32x4& dst) {   // the ring prologue's loads (the no-weight-stream diagnostic keeps exactly these)
        constexpr int nb = decltype(NB)::value;
        q_wload<(nb & 3) * 1024, true>(dst, vw + (unsigned)((nb >> 2) * 4096), wbase);
    };
    auto issue_stage = [&](f32x4 (&slot)[NBW]) {
        issue_wp(QC<0>{}, slot[0]); issue_wp(QC<1>{}, slot[1]); issue_wp(QC<2>{}, slot[2]);
        if constexpr (NBW > 3) issue_wp(QC<3>{}, slot[3]);
        if constexpr (NBW > 4) { issue_wp(QC<4>{}, slot[4]); issue_wp(QC<5>{}, slot[5]); }
        if constexpr (NBW > 6) { issue_wp(QC<6>{}, slot[6]); issue_wp(QC<7>{}, slot[7]); }
        if constexpr (NBW > 8) {
            issue_wp(QC<8>{}, slot[8]); issue_wp(QC<9>{}, slot[9]); issue_wp(QC<10>{}, slot[10]); issue_wp(QC<11>{}, slot[11]);
        }
        advance_w();
    };

    // ---- LayerNorm of the tile's 64 rows -> LDS (bf16, swizzled): wave w takes rows 16 w .. 16 w + 15, eight at a time ----
    {
        f32x4 g4[CH], b4[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            g4[c] = reinterpret_cast<const f32x4*>(p.gamma)[lane + 64 * c];
            b4[c] = reinterpret_cast<const f32x4*>(p.beta)[lane + 64 * c];
        }
        // all 16 rows of the wave are requested at once (round 5: the first version fetched them in two batches of eight, two memory
        // round trips per tile with the matrix pipe idle); bf16 rows stay packed until their row is normalised
        using raw_t = typename std::conditional<XB, u32x2, f32x4>::type;
        constexpr int RB = XB ? RPW : 8;                                 // rows in flight (fp32 rows: 8, the register file's share)
#pragma unroll
        for (int half = 0; half < RPW / RB; ++half) {
            raw_t raw[RB][CH];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                int gr = m0 + wid * RPW + half * RB + r;
                gr = gr < p.M ? gr : p.M - 1;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if constexpr (XB) raw[r][c] = reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16*>(p.h) + (size_t)gr * p.ldh)[lane + 64 * c];
                    else raw[r][c] = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.h) + (size_t)gr * p.ldh)[lane + 64 * c];
                }
            }
#ifdef DITTO_DIAG_LNQ_STAMP
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (half == 0) { const unsigned long long tl = lnq_now(); if (lane == 0 && blockIdx.x * NW + wid < 2048 * 4) g_lnq_stamps[(blockIdx.x * NW + wid) * 8 + 7] = tl - lnq_t0; }
#endif
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                f32x4 v[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if constexpr (XB) v[c] = f32x4{bf16_lo(raw[r][c][0]), bf16_hi(raw[r][c][0]), bf16_lo(raw[r][c][1]), bf16_hi(raw[r][c][1])};
                    else v[c] = raw[r][c];
                }
                // ln_kernel's arithmetic (rowwise.hip) through the same helpers (common.h): the same bits for the same row
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < CH; ++c) s += ln_sum4(v[c]);
                const float mean = wave_sum(s) / (float)QKD;
                float q = 0.f;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) q = ln_sq_acc(q, v[c][e], mean);
                }
                const float rstd = rsqrtf(wave_sum(q) / (float)QKD + 1e-5f);
                const int row = wid * RPW + half * RB + r;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    u32x2 o;
                    o[0] = pack_bf16x2(ln_norm(v[c][0], mean, rstd, g4[c][0], b4[c][0]), ln_norm(v[c][1], mean, rstd, g4[c][1], b4[c][1]));
                    o[1] = pack_bf16x2(ln_norm(v[c][2], mean, rstd, g4[c][2], b4[c][2]), ln_norm(v[c][3], mean, rstd, g4[c][3], b4[c][3]));
                    const int chunk = (lane + 64 * c) >> 1;                // 16-B chunk of the row holding k = 4 (lane + 64 c) ..
                    *reinterpret_cast<u32x2*>(smem + row * Q_AROW + ((chunk ^ (row & 15)) << 4) + (lane & 1) * 8) = o;
                }
            }
        }
    }
#ifdef DITTO_DIAG_LNQ_STAMP
    { const unsigned long long tv = lnq_now(); if (lane == 0 && blockIdx.x * NW + wid < 2048 * 4) g_lnq_stamps[(blockIdx.x * NW + wid) * 8 + 6] = tv - lnq_t0; }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // (wave 3: the bias pieces)
    __syncthreads();
    LNQ_STAMP(1);

    // ---- accumulators: zero, pinned in AGPRs ----
    acc_t acc[NBW][MBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) {
#pragma unroll
            for (int e = 0; e < (int)(sizeof(acc_t) / 4); ++e) acc[nb][mb][e] = 0.f;
            asm volatile("" : "+a"(acc[nb][mb]));
        }
    
#ifdef DITTO_DIAG_LNQ_STAMP
    const unsigned long long lnq_tinit = lnq_now();
#endif
    // ring prologue: from here on the W loads are the ONLY vector-memory operations until the epilogue's stores
#pragma unroll
    for (int r = 0; r < R; ++r) issue_stage(wr[r]);
#ifdef DITTO_DIAG_LNQ_NOW
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the only loads of the diagnostic: landed before any MFMA reads a fragment
#endif

    // ---- A fragment addressing.  SHAPE 32: lane (m = lane & 31, hh = lane >> 5) of row block mb reads row 32 mb + m, chunk
    //      2 s + hh;  SHAPE 16: lane (m = lane & 15, kq = lane >> 4) reads row 16 mb + m, chunk 4 s + kq.  row & 15 = m & 15 for
    //      every mb, and the chunk's low four bits are ((KCH s) & 15) ^ sub (no carry), so the swizzled byte offset is
    //      row * 1536 + (whole 256-B spans) + ((((KCH s) & 15) << 4) ^ y),  y = (sub ^ (m & 15)) << 4: PER distinct lane terms. ----
    constexpr int KCH = SHAPE == 32 ? 2 : 4;                              // 16-B chunks of A per stage
    const int am = SHAPE == 32 ? (lane & 31) : (lane & 15);
    const int asub = SHAPE == 32 ? (lane >> 5) : (lane >> 4);
    const unsigned ay = (unsigned)((asub ^ (am & 15)) << 4);
    unsigned aoff[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) aoff[j] = (unsigned)(am * Q_AROW) + ((unsigned)(((KCH * j) & 15) << 4) ^ ay);
    constexpr int MB_STRIDE = (SHAPE == 32 ? 32 : 16) * Q_AROW;           // bytes between row blocks
    bf16x8 a0[MBW], a1[MBW];
    auto read_a = [&](bf16x8 (&dst)[MBW], int span_bytes, unsigned lane_off) {
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb)
            dst[mb] = *reinterpret_cast<const bf16x8*>(smem + lane_off + span_bytes + mb * MB_STRIDE);
    };
    read_a(a0, p0 * 256, aoff[0]);

    // One stage: MFMAs of (s, nb) behind the fragment's counted wait, then fragment nb of stage s + R into the same registers;
    // the next stage's A fragments are read behind the first column block.  J: position in the swizzle period (compile time),
    // REM: stages after this one when that is < R (the tail), else R.
    auto stage = [&](auto J, auto REMC, int span_next, f32x4 (&slot)[NBW], bf16x8 (&ACUR)[MBW], bf16x8 (&ANXT)[MBW]) {
        constexpr int j = decltype(J)::value, remc = decltype(REMC)::value;
        constexpr bool last = remc == 0, do_w = remc >= R;
        auto block = [&](auto NB) {
            constexpr int nb = decltype(NB)::value;
            q_wait<lnq_vm(NBW, R, nb, remc)>(slot[nb]);
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) {
                if constexpr (last) q_mfma_last(acc[nb][mb], slot[nb], ACUR[mb]);
                else q_mfma(acc[nb][mb], slot[nb], ACUR[mb]);
            }
            if constexpr (do_w) issue_w(NB, slot[nb]);
            if constexpr (nb == 0 && !last) read_a(ANXT, span_next, aoff[(j + 1) % PER]);
        };
        block(QC<0>{}); block(QC<1>{}); block(QC<2>{});
        if constexpr (NBW > 3) block(QC<3>{});
        if constexpr (NBW > 4) { block(QC<4>{}); block(QC<5>{}); }
        if constexpr (NBW > 6) { block(QC<6>{}); block(QC<7>{}); }
        if constexpr (NBW > 8) { block(QC<8>{}); block(QC<9>{}); block(QC<10>{}); block(QC<11>{}); }
        if constexpr (do_w) advance_w();
    };
    // a period = PER stages = one 256-B span of A (16 chunks); a stage reads the NEXT stage's fragments, which sit in the next
    // period's span when j = PER - 1.  Stage j of a period uses ring slot j % R.
    auto period = [&](auto LASTP, int span, int span_after) {
        constexpr bool lastp = decltype(LASTP)::value != 0;
#define LNQ_REM(j) QC<(lastp ? ((PER - 1 - (j)) < R ? (PER - 1 - (j)) : R) : R)>{}
        stage(QC<0>{}, LNQ_REM(0), span, wr[0 % R], a0, a1);
        stage(QC<1>{}, LNQ_REM(1), span, wr[1 % R], a1, a0);
        stage(QC<2>{}, LNQ_REM(2), span, wr[2 % R], a0, a1);
        if constexpr (PER == 4) {
            stage(QC<3>{}, LNQ_REM(3), span_after, wr[3 % R], a1, a0);
        } else {
            stage(QC<3>{}, LNQ_REM(3), span, wr[3 % R], a1, a0);
            stage(QC<4>{}, LNQ_REM(4), span, wr[4 % R], a0, a1);
            stage(QC<5>{}, LNQ_REM(5), span, wr[5 % R], a1, a0);
            stage(QC<6>{}, LNQ_REM(6), span, wr[6 % R], a0, a1);
            stage(QC<7>{}, LNQ_REM(7), span_after, wr[7 % R], a1, a0);
        }
#undef LNQ_REM
    };
    int pcur = p0;
    for (int pi = 0; pi + 1 < NPER; ++pi) {
        const int pnext = pcur + 1 == NPER ? 0 : pcur + 1;
        period(QC<0>{}, pcur * 256, pnext * 256);
        pcur = pnext;
    }
    period(QC<1>{}, pcur * 256, 0);
    LNQ_STAMP(2);

    // ---------------- epilogue: q = acc + bias -> bf16, staged through the A region (every wave is done reading it) ----------------
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();
    constexpr int CPR = D / 8;                                            // 16-B chunks per output row
    const float* lbias = reinterpret_cast<const float*>(smem + Q_BIAS);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) {
            asm volatile("" : "+a"(acc[nb][mb]));                         // re-pin: the copy below is a NEW value, never hoisted
            const acc_t v = acc[nb][mb];
            if constexpr (SHAPE == 32) {
                const int row = mb * 32 + (lane & 31);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = wid * (D / NW) + nb * 32 + 8 * g + 4 * (lane >> 5);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(lbias + col);
                    u32x2 st;
                    st[0] = pack_bf16x2(v[4 * g] + b[0], v[4 * g + 1] + b[1]);
                    st[1] = pack_bf16x2(v[4 * g + 2] + b[2], v[4 * g + 3] + b[3]);
                    *reinterpret_cast<u32x2*>(smem + row * Q_AROW + (((col >> 3) ^ (row & 15)) << 4) + (col & 7) * 2) = st;
                }
            } else {
                const int row = mb * 16 + (lane & 15);
                const int col = wid * (D / NW) + nb * 16 + 4 * (lane >> 4);
                const f32x4 b = *reinterpret_cast<const f32x4*>(lbias + col);
                u32x2 st;
                st[0] = pack_bf16x2(v[0] + b[0], v[1] + b[1]);
                st[1] = pack_bf16x2(v[2] + b[2], v[3] + b[3]);
                *reinterpret_cast<u32x2*>(smem + row * Q_AROW + (((col >> 3) ^ (row & 15)) << 4) + (col & 7) * 2) = st;
            }
        }
    }
    __syncthreads();
    // read-back: 64 rows x 96 chunks of 16 B, chunk id = tid + 256 i: lanes walk a row's chunks, whole 128-B lines per store
#pragma unroll 4
    for (int i = 0; i < QM * CPR / (NW * 64); ++i) {
        const int id = tid + NW * 64 * i;
        const int row = id / CPR, c = id - row * CPR;
        const u32x4 val = *reinterpret_cast<const u32x4*>(smem + row * Q_AROW + ((c ^ (row & 15)) << 4));
        if (m0 + row < p.M) *reinterpret_cast<u32x4*>(p.out + (size_t)(m0 + row) * p.ldo + c * 8) = val;
    }
#ifdef DITTO_DIAG_LNQ_STAMP
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LNQ_STAMP(3);
        const int w = blockIdx.x * NW + wid;
        if (lane == 0 && w < 2048 * 4) {
            g_lnq_stamps[w * 8 + 0] = lnq_t1 - lnq_t0; g_lnq_stamps[w * 8 + 1] = lnq_t2 - lnq_t1; g_lnq_stamps[w * 8 + 2] = lnq_t3 - lnq_t2;
            g_lnq_stamps[w * 8 + 3] = 1; g_lnq_stamps[w * 8 + 4] = lnq_t0; g_lnq_stamps[w * 8 + 5] = lnq_t3;
        }
    }
#endif
}

template <int SHAPE, int R, bool XB, int D = 768, int NW = 4>
hipError_t launch_lnq_t(const LnqParams& p, hipStream_t s) {
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&gemm_lnq_kernel<SHAPE, R, XB, D, NW>)}, QW<D>::LDS)) return e;
    hipLaunchKernelGGL((gemm_lnq_kernel<SHAPE, R, XB, D, NW>), dim3((p.M + QM - 1) / QM), dim3(NW * 64), QW<D>::LDS, s, p);
    return hipGetLastError();
}

}  // namespace

#ifdef DITTO_DIAG_LNQ_STAMP
}  // namespace ditto
extern "C" int ditto_diag_lnq_stamps(unsigned long long* out, int n) {   // raw per-wave records of the LAST launch (diagnostic build only)
    if (n > 2048 * 4 * 8) n = 2048 * 4 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ditto::g_lnq_stamps), (size_t)n * 8) != hipSuccess;
}
namespace ditto {
#endif

// 8 (default since round 5) = two waves per SIMD, 4 = one (the round-4 form): bit-identical outputs; shape 32 only (the 16x16x32 twin stays at 4)
int g_lnq_waves = [] { const char* e = getenv("DITTO_LNQ_WAVES"); return e ? atoi(e) : 8; }();
int g_lnq_ring = [] { const char* e = getenv("DITTO_LNQ_RING"); return e ? atoi(e) : 0; }();   // 0 = the shape's default depth

// h: fp32 [M, ldh] (h_bf16 false) or bf16 [M, ldh]; Wp: the stage-major image of W_q for `shape` (32: group 16, 16: group 32);
// rot_period: tiles (of 64 rows) per utterance when that is whole, else 0 (no K-loop rotation)
hipError_t launch_gemm_lnq(const void* h, int ldh, bool h_bf16, const float* gamma, const float* beta, const void* Wp,
                           const float* bias, void* out_bf16, int ldo, int M, int d, int shape, int rot_period, hipStream_t s) {
    if ((d != 768 && d != 1024) || M <= 0 || !h || !gamma || !beta || !Wp || !out_bf16 || (shape != 32 && shape != 16))
        return hipErrorInvalidValue;
    if (d == 1024 && (shape != 32 || h_bf16)) return hipErrorInvalidValue;   // C5's width: 32x32x16, fp32 rows
    if (ldh % 4 || ldo % 8) return hipErrorInvalidValue;
    LnqParams p{};
    p.h = h; p.ldh = ldh; p.gamma = gamma; p.beta = beta; p.Wp = (const char*)Wp; p.bias = bias;
    p.out = (bf16*)out_bf16; p.ldo = ldo; p.M = M; p.rot_period = rot_period > 0 ? rot_period : 0;
    const int ring = g_lnq_ring;
    if (d == 1024) return g_lnq_waves == 8 ? launch_lnq_t<32, 4, false, 1024, 8>(p, s) : launch_lnq_t<32, 4, false, 1024>(p, s);
    if (g_lnq_waves == 8 && shape == 32)   // two waves per SIMD (round 5, default): "lnq_waves" 8; in the model 60.8 -> 53.2 us per launch
        return h_bf16 ? launch_lnq_t<32, 4, true, 768, 8>(p, s) : launch_lnq_t<32, 4, false, 768, 8>(p, s);
    // default depth = the shallow ring: in the model (tools/step_ab.py, C2 B = 32, one process) 4 stages 61.3 us against 63.2 for 8
    // (shape 32), 2 stages 70.6 against 72.2 for 4 (shape 16): the loop is not short of bytes in flight
    if (shape == 32) {
        if (ring == 8) return h_bf16 ? launch_lnq_t<32, 8, true>(p, s) : launch_lnq_t<32, 8, false>(p, s);
        return h_bf16 ? launch_lnq_t<32, 4, true>(p, s) : launch_lnq_t<32, 4, false>(p, s);
    }
    if (ring == 4) return h_bf16 ? launch_lnq_t<16, 4, true>(p, s) : launch_lnq_t<16, 4, false>(p, s);
    return h_bf16 ? launch_lnq_t<16, 2, true>(p, s) : launch_lnq_t<16, 2, false>(p, s);
}

}  // namespace ditto
