// gemm_fr.hip — FULL-ROW bf16 MFMA GEMM for the N = d = 768 projections of a DiT block, with the fp32 residual add
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
//   LDS       ring of FIVE stages x 28 KiB (K = 16: A 128 rows x 32 B, W 768 rows x 32 B) = 140 KiB, LDS-DMA three
//             stages ahead behind a counted vmcnt; + bias / gamma / beta rows (9 KiB, DMA'd once) + 2 KiB of row
//             statistics.  The K = 16 instruction is what makes five stages fit: with 16x16x32 MFMAs (K = 32) the
//             same LDS holds two stages, one stage of look-ahead, and the first version ran at 1.93 us per K = 32.
//             16-B chunk h (0 / 1) of 32-B row r sits at h ^ ((r >> 3) & 1): conflict-free ds_read_b128.
//   stage s   { nb = 0..6: 2 MFMAs + one DMA piece of stage s+3 each ; nb = 7 ; nb = 8: vmcnt(14) + s_barrier (stage s+1
//               has landed for everyone, and everyone is past stage s-1), prefetch stage s+1's A fragments ;
//               nb = 9..11: prefetch stage s+1's first W fragments }   W fragments run 3 ahead in a 4-register ring,
//             A fragments ping-pong between two named sets (two stages per loop iteration): one barrier per stage.
//   epilogue  no compiler-visible global load (each would make hipcc wait for every DMA in flight): bias / gamma / beta
//             come from LDS, the fp32 residual tile streams through the idle ring in six 64 KiB chunks by LDS-DMA
//             (double-buffered; 16-B chunk q of row r at q ^ (r & 15)), v = acc + bias + residual goes back into the
//             accumulators' home registers; row sums -> lane l ^ 32 -> the other column half through LDS; mean; the same
//             for sum (v - mean)^2; h stored fp32 (nt), u = LN(v) bf16.  Two-pass statistics, like nn.LayerNorm.
#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int FM = 128, FN = 768, FK = 16, NST = 5;
constexpr int F_A_BYTES = FM * FK * 2;             // 4 KiB
constexpr int F_W_BYTES = FN * FK * 2;             // 24 KiB
constexpr int F_STAGE = F_A_BYTES + F_W_BYTES;     // 28 KiB
constexpr int F_RING = NST * F_STAGE;              // 140 KiB
constexpr int F_VEC = F_RING;                      // bias | gamma | beta rows, 3 KiB each
constexpr int F_RED = F_VEC + 3 * FN * 4;          // row statistics [2 passes][2 column halves][128 rows] fp32
constexpr int F_LDS = F_RED + 2 * 2 * FM * 4;      // 151 KiB
constexpr int F_RES = 2 * FM * 64 * 4;             // one residual chunk: [2 column halves][128 rows][64 fp32] = 64 KiB
constexpr int NA = 7;                              // column blocks whose accumulators live in AGPRs
static_assert(2 * F_RES <= F_RING, "residual double buffer must fit the idle ring");

#define FR_BAR() asm volatile("s_barrier" ::: "memory")
#define PIN_A(x) asm volatile("" : "+a"(x))
#define PIN_V(x) asm volatile("" : "+v"(x))

struct FrParams {
    GemmParams g;
    const float* gamma; const float* beta;   // LayerNorm affine of the fused norm (null: no LayerNorm output)
    bf16* u; int ldu;                         // LayerNorm output
};

template <int V>
struct IC { static constexpr int value = V; };

DITTO_DEV void mfma_a(f32x16& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
}
DITTO_DEV void mfma_v(f32x16& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(a));
}

template <bool LN>
__global__ __launch_bounds__(256, 1) void gemm_fr_kernel(FrParams fp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemmParams& p = fp.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int nkt = p.K / FK;                  // even and >= 4 (K % 32 == 0, K >= 64)
    const int m0 = blockIdx.x * FM;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;

    // ---- stage DMA: 28 pieces of 1 KiB (32 rows x 32 B): 0..3 = A, 4..27 = W; wave w moves pieces 7w .. 7w+6.
    //      Per piece the source is (loop-invariant scalar base) + (per-lane offset that advances 32 B per stage); the
    //      destination goes straight into M0 (s_add_u32 m0, ring slot, piece * 1024).  M0 is not restored: nothing else in
    //      this kernel reads it.  With one wave per SIMD every scalar instruction of the loop is issue time the MFMAs
    //      wait behind (a first version spent 184 SALU instructions per 24 MFMAs on these addresses). ----
    const int prow = lane >> 1, ppos = lane & 1;
    const int pc = ppos ^ ((prow >> 3) & 1);                     // source chunk landing at position ppos of row prow
    // W arrives PACKED stage-major, Wp[K/16][768][16] (launch_pack_w_fr): a stage's 24 KiB are contiguous, every 1-KiB piece
    // a run of whole cache lines.  (Read from the row-major [768][K] image, a piece touched 32 lines for 32 B each and the
    // L2 -> CU path moved 4x the payload: 2.6 us per K = 32 instead of ~1.)
    unsigned vwk = (unsigned)(prow * 32 + pc * 16);                   // W pieces: per-lane byte offset (+ 24 KiB per stage)
    unsigned vak[4];                                             // A pieces (wave 0): row clamp makes them per piece
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ar = m0 + i * 32 + prow;
        ar = ar < p.M ? ar : p.M - 1;
        vak[i] = (unsigned)(((size_t)ar * p.lda + pc * 8) * 2);
    }
    const char* pbase[7];                                        // wave-uniform: SGPR pairs
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int piece = wid * 7 + i;
        pbase[i] = piece < 4 ? (const char*)p.A : (const char*)p.W + (size_t)(piece - 4) * 1024;
    }
    unsigned i_slot = lds_base;                                  // LDS byte address of the ring slot the next stage goes to
    auto issue_piece = [&](auto I) {                             // piece I (0..6) of the stage at the issue cursor
        constexpr int i = decltype(I)::value;
        const int piece = wid * 7 + i;
        const unsigned voff = piece < 4 ? vak[i < 4 ? i : 0] : vwk;   // wave 0's first four pieces are A
        const unsigned dst = i_slot + (unsigned)(piece * 1024);
        const char* base = pbase[i];
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(dst) : "memory");
    };
    auto advance_issue = [&]() {
        i_slot = i_slot + F_STAGE == lds_base + F_RING ? lds_base : i_slot + F_STAGE;
        vwk += F_W_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) vak[i] += FK * 2;
    };
    auto issue_stage = [&]() {
        issue_piece(IC<0>{}); issue_piece(IC<1>{}); issue_piece(IC<2>{}); issue_piece(IC<3>{});
        issue_piece(IC<4>{}); issue_piece(IC<5>{}); issue_piece(IC<6>{});
        advance_issue();
    };

    // bias / gamma / beta rows -> LDS (3 pieces of 1 KiB each = 768 fp32), oldest loads of the kernel
    if (wid < 3) {
        const float* src = wid == 0 ? p.bias : (wid == 1 ? fp.gamma : fp.beta);
        if (src) {
#pragma unroll
            for (int i = 0; i < 3; ++i) glds16(src + i * 256 + lane * 4, lds_base + (unsigned)(F_VEC + wid * FN * 4 + i * 1024));
        }
    }
    // ---- fragment addressing: lane reads row (lane & 31) of a 32-row block, 16-B half (lane >> 5) ----
    const int r32 = lane & 31, hh = lane >> 5;
    const int fpos = (hh ^ ((r32 >> 3) & 1)) << 4;
    const int a_off = (wm * 64 + r32) * 32 + fpos;                         // + mb * 1024
    const int w_off = F_A_BYTES + (wn * 384 + r32) * 32 + fpos;             // + nb * 1024

    // ---- the accumulators START as bias + residual: the fp32 residual tile streams through the (still empty) ring in six
    //      64 KiB chunks by LDS-DMA (double-buffered; 16-B chunk q of row r at q ^ (r & 15): conflict-free reads), so the
    //      epilogue only READS the accumulators.  (Adding the residual afterwards meant writing 384 updated values back into
    //      their AGPR / VGPR homes, which hipcc turned into a scratch copy of every block and reloads behind vmcnt(0) in the
    //      later passes: a 170 us epilogue.)  Chunk c = column blocks nb in {2c, 2c+1} of both column halves: 64 pieces of
    //      1 KiB (4 rows x 256 B), 16 per wave.
    const float* lbias = reinterpret_cast<const float*>(smem + F_VEC);
    const float* lgamma = lbias + FN;
    const float* lbeta = lgamma + FN;
    const bool has_res = p.residual != nullptr;
    const bool has_bias = p.bias != nullptr;
    const int rrow = lane >> 4, rpos = lane & 15;
    auto issue_chunk = [&](int c) {
        const unsigned buf = (unsigned)((c & 1) * F_RES);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int q = wid * 16 + i;                              // wave-uniform: column half q >> 5, rows 4 (q & 31) ..
            const int row = 4 * (q & 31) + rrow;
            int gr = m0 + row;
            gr = gr < p.M ? gr : p.M - 1;
            const int srcq = rpos ^ (row & 15);
            const float* src = p.residual + (size_t)gr * p.ldr + (q >> 5) * 384 + c * 64 + srcq * 4;
            glds16(src, lds_base + buf + (unsigned)(q * 1024));
        }
    };
    f32x16 acca[NA][2], accv[12 - NA][2];
    if (has_res) issue_chunk(0);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        if (has_res) {
            if (c > 0) FR_BAR();                                     // every wave has read chunk c-1: its buffer is free
            if (c + 1 < 6) {
                issue_chunk(c + 1);
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            FR_BAR();                                                // everyone's pieces of chunk c have landed
        } else if (c == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the bias row
            FR_BAR();
        }
        const char* rb = smem + (c & 1) * F_RES + wn * (FM * 64 * 4);
#pragma unroll
        for (int nbl = 0; nbl < 2; ++nbl) {
            const int nb = 2 * c + nbl;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int row = wm * 64 + mb * 32 + r32;
                f32x16 v;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = wn * 384 + nb * 32 + 8 * g + 4 * hh;
                    f32x4 add = {0.f, 0.f, 0.f, 0.f};
                    if (has_bias) add = *reinterpret_cast<const f32x4*>(lbias + col);
                    if (has_res) {
                        const int q = nbl * 8 + 2 * g + hh;
                        add += *reinterpret_cast<const f32x4*>(rb + row * 256 + ((q ^ (row & 15)) << 4));
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * g + e] = add[e];
                }
                if (nb < NA) { acca[nb < NA ? nb : 0][mb] = v; PIN_A(acca[nb < NA ? nb : 0][mb]); }
                else { accv[nb < NA ? 0 : nb - NA][mb] = v; PIN_V(accv[nb < NA ? 0 : nb - NA][mb]); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    FR_BAR();   // every wave is done with the chunk buffers: the ring may fill
    issue_stage(); issue_stage(); issue_stage();                 // stages 0, 1, 2 (nkt >= 4)

    // stage 0 has landed (the two younger stages stay in flight)
    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    FR_BAR();
    bf16x8 a0[2], a1[2], wf[4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) a0[mb] = *reinterpret_cast<const bf16x8*>(smem + a_off + mb * 1024);
#pragma unroll
    for (int n = 0; n < 3; ++n) wf[n] = *reinterpret_cast<const bf16x8*>(smem + w_off + n * 1024);

    unsigned c_off = 0;   // ring byte offset of the stage being multiplied
    // One stage.  ACUR: its A fragments (resident), ANXT receives the next stage's.  ISSUE: the stage three ahead exists
    // and its 7 DMA pieces go out between the first MFMAs; NEXT: a next stage exists; YOUNGER: how many whole stages
    // issued after stage kt+1 are in flight at its wait (compile-time: counted vmcnt needs an immediate).
    auto stage = [&](auto ISSUE, auto NEXT, auto YOUNGER, bf16x8 (&ACUR)[2], bf16x8 (&ANXT)[2]) {
        constexpr bool do_issue = decltype(ISSUE)::value != 0, has_next = decltype(NEXT)::value != 0;
        constexpr int younger = decltype(YOUNGER)::value;
        const char* cur = smem + c_off;
        const unsigned n_off = c_off + F_STAGE == F_RING ? 0u : c_off + F_STAGE;
        const char* nxt = smem + n_off;
#pragma unroll
        for (int nb = 0; nb < 12; ++nb) {
            if (nb + 3 < 12) wf[(nb + 3) & 3] = *reinterpret_cast<const bf16x8*>(cur + w_off + (nb + 3) * 1024);
            if (nb == 8 && has_next) {
                // stage kt+1 has landed: this wave's pieces by the counted vmcnt (the stages issued after it stay in
                // flight), everyone's by the barrier — which also certifies that every wave is past stage kt-1, whose
                // slot the NEXT stage's DMA issue overwrites
                if constexpr (younger >= 2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                else if constexpr (younger == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                FR_BAR();
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) ANXT[mb] = *reinterpret_cast<const bf16x8*>(nxt + a_off + mb * 1024);
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if (nb < NA) mfma_a(acca[nb < NA ? nb : 0][mb], wf[nb & 3], ACUR[mb]);
                else mfma_v(accv[nb < NA ? 0 : nb - NA][mb], wf[nb & 3], ACUR[mb]);
            }
            // LAST stage: these are the final writes of the block's accumulators, and hipcc may read them right behind the
            // asm (it spilled a just-written AGPR block with scratch_store two instructions later: garbage in some lanes
            // of some launches).  An MFMA's result needs its wait states before ANY reader but the next MFMA of its chain.
            if constexpr (!has_next) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
            if constexpr (do_issue) {
                if (nb == 0) issue_piece(IC<0>{});
                if (nb == 1) issue_piece(IC<1>{});
                if (nb == 2) issue_piece(IC<2>{});
                if (nb == 3) issue_piece(IC<3>{});
                if (nb == 4) issue_piece(IC<4>{});
                if (nb == 5) issue_piece(IC<5>{});
                if (nb == 6) issue_piece(IC<6>{});
            }
            if (nb >= 9 && has_next)   // W fragments 0..2 of the next stage, into ring slots the MFMAs above released
                wf[(nb - 9) & 3] = *reinterpret_cast<const bf16x8*>(nxt + w_off + (nb - 9) * 1024);
        }
        if constexpr (do_issue) advance_issue();
        c_off = n_off;
    };
    // stages 0 .. nkt-5 in pairs (the A fragments ping-pong between two NAMED sets), then the four-stage tail in which the
    // issue stops and the counted waits shrink
    for (int kt = 0; kt + 6 <= nkt; kt += 2) {
        stage(IC<1>{}, IC<1>{}, IC<2>{}, a0, a1);
        stage(IC<1>{}, IC<1>{}, IC<2>{}, a1, a0);
    }
    stage(IC<1>{}, IC<1>{}, IC<2>{}, a0, a1);   // stage nkt-4: issues stage nkt-1
    stage(IC<0>{}, IC<1>{}, IC<1>{}, a1, a0);   // stage nkt-3
    stage(IC<0>{}, IC<1>{}, IC<0>{}, a0, a1);   // stage nkt-2
    stage(IC<0>{}, IC<0>{}, IC<0>{}, a1, a0);   // stage nkt-1

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
                if (grow0 + 8 * i < p.M) store16<true, true>(hrow + (size_t)(8 * i) * p.ldo + nb * 32, hv, 0);
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
                    if (grow0 + 8 * i < p.M) {
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

template <bool LN>
hipError_t launch_fr_t(const FrParams& fp, int grid, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fr_kernel<LN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_fr_kernel<LN>), dim3(grid), dim3(256), F_LDS, s, fp);
    return hipGetLastError();
}

}  // namespace

bool gemm_fr_supports(int M, int N, int K, size_t lda, size_t ldw) {
    if (N != FN || K % 32 || K < 64 || M < FM) return false;
    if ((size_t)M * lda * 2 >= (1ull << 32) || (size_t)N * ldw * 2 >= (1ull << 32)) return false;
    return true;
}

hipError_t launch_gemm_fr(const GemmParams& p_in, const float* gamma, const float* beta, void* u_bf16, int ldu,
                          hipStream_t s) {
    FrParams fp;
    fp.g = p_in;
    fp.g.tiles_m = (p_in.M + FM - 1) / FM;
    fp.g.tiles_n = 1;
    fp.gamma = gamma; fp.beta = beta; fp.u = (bf16*)u_bf16; fp.ldu = ldu;
    return (gamma && u_bf16) ? launch_fr_t<true>(fp, fp.g.tiles_m, s) : launch_fr_t<false>(fp, fp.g.tiles_m, s);
}

}  // namespace ditto
