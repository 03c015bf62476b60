// gemm_pp.hip — "ping-pong" persistent bf16 MFMA GEMM for gfx950: 128x256x32 tiles, TWO independent 4-wave
// workgroups per CU, half a tile out of phase.
//
// Same contract and epilogues as gemm.hip (out = A[M,K] * W[N,K]^T, fused epilogue).  Why this structure: in the
// persistent 256x256 kernel (gemm256.hip) the two waves of every SIMD belong to ONE workgroup and move in lockstep — both
// in the main loop (matrix pipe busy, VALU idle), then both in the epilogue (VALU / store issue busy, matrix pipe
// idle).  At K = 768 the epilogue is 23-31 % of the kernel (gelu*sigmoid: 1 750 VALU instructions per wave per tile)
// and nothing overlaps it: two accumulator sets do not fit 256 registers.  The matrix pipe and the VALU are separate
// pipes of a SIMD, so the overlap needs no hand interleaving if the two waves of a SIMD are in DIFFERENT phases.  Here
// they belong to two workgroups, each with its own barrier, its own LDS ring and its own tile sequence; the workgroup
// that gets the second LDS allocation of its CU starts half a tile late, after which one workgroup's epilogue runs
// under the other's main loop and any stall of one (DMA wait, store drain at the tile switch) is matrix time for the
// other.  (In-phase operation is an attractor — the lagging workgroup gets the whole matrix pipe while the leader is in
// its epilogue and catches up — so the offset is created explicitly and must exceed the epilogue length.)
//
//   tile      128 (M) x 256 (N), 256 threads = 4 waves side by side in N; wave wn owns all 128 rows x columns
//             [64 wn, +64): 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16 = 128 fp32 registers — the wave tile and the
//             epilogue code of gemm256.hip, unchanged (12 ds_read_b128 per 32 MFMAs).
//   LDS       ring of 3 stages x 24 KiB (K = 32: A 128 rows x 64 B, W 256 rows x 64 B) = 72 KiB per workgroup, two
//             workgroups = 144 of the CU's 160 KiB.  16-B chunk c of row r sits at c ^ (-(r>>2) & 3) (conflict-free
//             for the hardware's ds_read_b128 lane groups, see gemm_o3.hip); lane-linear image, swizzle on the DMA's
//             source address.
//   pipeline  the K loop is FLAT over the workgroup's tile sequence: stage s+2 is issued (global_load_lds_dwordx4,
//             6 x 1 KiB per wave) right after the barrier that opens stage s, so two stages are always in flight, also
//             across the tile switch — the next tile's first two stages land under the epilogue.
//             per stage:  s_waitcnt vmcnt(6) ; s_barrier ; issue stage s+2 ; 12 ds_read_b128 ; lgkmcnt(0) ; 32 MFMA
//             One barrier per stage: it certifies (a) every wave's pieces of stage s have landed and (b) every wave
//             has retired its reads of stage s-1, whose slot stage s+2 overwrites.
//   tails     rows past M / N are clamped on load and masked on store.
#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int PM = 128, PK = 32;
constexpr int P_A_BYTES = PM * PK * 2;            // 8 KiB
constexpr int P_NSTAGE = 3;
// NB = 16-column blocks per wave: 4 -> 128 x 256 tiles (every epilogue), 3 -> 128 x 192 tiles (the plain epilogues; at
// N = 768 and M = 32768 that is 1024 tiles = exactly two per workgroup slot, where 256-wide tiles make 1.5 rounds)
template <int NB> struct PP {
    static constexpr int PN = 64 * NB;
    static constexpr int W_BYTES = PN * PK * 2;               // 16 / 12 KiB
    static constexpr int STAGE = P_A_BYTES + W_BYTES;         // 24 / 20 KiB
    static constexpr int LDS = P_NSTAGE * STAGE;              // 72 / 60 KiB
    static constexpr int PIECES = STAGE / 1024;               // 24 / 20
    static constexpr int PER_WAVE = PIECES / 4;               // 6 / 5
};

#define PP_BAR() asm volatile("s_barrier" ::: "memory")

template <int EPI, int NB>
__global__ __launch_bounds__(256, 2) void gemm_pp_kernel(GemmParams p) {
    constexpr int PN = PP<NB>::PN, P_STAGE = PP<NB>::STAGE, P_LDS = PP<NB>::LDS, NPW = PP<NB>::PER_WAVE;
    static_assert(NB == 4 || (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_RES_F32 || EPI == EPI_BIAS_F32),
                  "the RoPE / gated epilogues need 64-column spans");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nkt = p.K / PK;
    const int stride = p.tile_stride;

    // ---- phase offset: the workgroup holding the CU's SECOND LDS allocation starts late (speed only) ----
    if (p.stagger_ticks > 0) {
        // HW_REG_LDS_ALLOC (id 6): LDS_BASE in bits [7:0]; s_getreg simm16 = (size-1) << 11 | offset << 6 | id
        const unsigned lds_alloc_base = __builtin_amdgcn_s_getreg((7 << 11) | (0 << 6) | 6);
        const bool second = (p.flags & GF_PP_PARITY) ? (blockIdx.x & 1) : (lds_alloc_base != 0);
        if (second) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)p.stagger_ticks) __builtin_amdgcn_s_sleep(8);
        }
    }

    // ---- DMA addressing: a stage = 24 pieces of 1 KiB (16 rows x 64 B), pieces 0..7 = A, 8..23 = W; wave w moves
    //      pieces 6w .. 6w+5.  Address = (A or W + kt * 64 B) [scalar] + this lane's row / chunk offset [voff] ----
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int prow = lane >> 2, cpos = lane & 3;
    unsigned voff[NPW];
    auto set_issue_tile = [&](int tile) {
        int tm, tn;
        tile_to_mn(xcd_remap(tile, ntiles), p.tiles_m, p.tiles_n, p.group_n, tm, tn);
        const int m0 = tm * PM, n0 = tn * PN;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int piece = wn * NPW + i;
            const int row = (piece < 8 ? piece : piece - 8) * 16 + prow;
            const int c = cpos ^ ((0 - (row >> 2)) & 3);
            if (piece < 8) {
                int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
                voff[i] = (unsigned)(((size_t)ar * p.lda + c * 8) * 2);
            } else {
                int wr = n0 + row; wr = wr < p.w_rows ? wr : p.w_rows - 1;
                voff[i] = (unsigned)(((size_t)wr * p.ldw + c * 8) * 2);
            }
        }
    };
    int i_tile = blockIdx.x, i_kt = 0;      // issue cursor
    unsigned i_slot = 0;                    // LDS byte offset of the slot the next stage goes to
    auto issue_stage = [&]() -> bool {      // wave-uniform
        if (i_tile >= ntiles) return false;
        const char* abase = (const char*)p.A + (size_t)i_kt * (PK * 2);
        const char* wbase = (const char*)p.W + (size_t)i_kt * (PK * 2);
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int piece = wn * NPW + i;
            glds16_so(voff[i], piece < 8 ? abase : wbase, lds_base + i_slot + (unsigned)(piece * 1024));
        }
        i_slot = i_slot + P_STAGE == P_LDS ? 0u : i_slot + P_STAGE;
        if (++i_kt == nkt) {
            i_kt = 0;
            i_tile += stride;
            if (i_tile < ntiles) set_issue_tile(i_tile);
        }
        return true;
    };

    // ---- fragment addressing: lane reads row (lane & 15) of a 16-row block, k-chunk (lane >> 4) ----
    const int frow = lane & 15, fq = lane >> 4;
    const int coff = (fq ^ ((0 - (frow >> 2)) & 3)) << 4;
    const int a_off = frow * 64 + coff;                                 // + m * 16 * 64
    const int w_off = P_A_BYTES + (wn * 16 * NB + frow) * 64 + coff;     // + n * 16 * 64

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    set_issue_tile(tile);
    bool ahead1 = issue_stage();            // stage 0
    bool ahead2 = issue_stage();            // stage 1 (may not exist)
    (void)ahead1;
    unsigned c_slot = 0;                    // LDS byte offset of the slot being multiplied

    f32x4 acc[8][NB];
    for (; tile < ntiles; tile += stride) {
        int tm, tn;
        tile_to_mn(xcd_remap(tile, ntiles), p.tiles_m, p.tiles_n, p.group_n, tm, tn);
        const int m0 = tm * PM, n0 = tn * PN;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nkt; ++kt) {
            // stage (tile, kt) has landed: everything but the one younger stage (if one was issued).  At kt = 0 the
            // previous tile's epilogue stores are younger than this stage's DMA: drain them too (the partner workgroup
            // has the matrix pipe meanwhile).
            if (kt != 0 && ahead2) {
                if constexpr (NPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            PP_BAR();
            const char* cur = smem + c_slot;
            bf16x8 af[8], wf[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) wf[n] = *reinterpret_cast<const bf16x8*>(cur + w_off + n * 16 * 64);
#pragma unroll
            for (int m = 0; m < 8; ++m) af[m] = *reinterpret_cast<const bf16x8*>(cur + a_off + m * 16 * 64);
            // the stage's 6 DMA pieces go out BETWEEN the MFMAs (one per 5-6), not in front of them: a piece costs the
            // issuing wave ~60-180 cycles, which in front of the reads was serial time of every stage; behind an MFMA
            // it overlaps the matrix pipe's own 16 cycles per instruction.  sched_barrier pins the interleaving.
            const bool do_issue = i_tile < ntiles;
            const char* abase = (const char*)p.A + (size_t)i_kt * (PK * 2);
            const char* wbase = (const char*)p.W + (size_t)i_kt * (PK * 2);
            int piece_i = 0;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], af[m], acc[m][n], 0, 0, 0);
                    const int idx = m * NB + n;
                    // piece k goes out behind MFMA ((k + 1) * 8 NB) / NPW - 1: evenly spread, the last behind the last
                    if (piece_i < NPW && idx == ((piece_i + 1) * 8 * NB) / NPW - 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (do_issue) {
                            const int piece = wn * NPW + piece_i;
                            glds16_so(voff[piece_i], piece < 8 ? abase : wbase, lds_base + i_slot + (unsigned)(piece * 1024));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        ++piece_i;
                    }
                }
            }
            if (do_issue) {
                i_slot = i_slot + P_STAGE == P_LDS ? 0u : i_slot + P_STAGE;
                if (++i_kt == nkt) {
                    i_kt = 0;
                    i_tile += stride;
                    if (i_tile < ntiles) set_issue_tile(i_tile);
                }
            }
            ahead2 = do_issue;
            c_slot = c_slot + P_STAGE == P_LDS ? 0u : c_slot + P_STAGE;
        }

        // ---------------- epilogue (gemm_common.h): the next tile's first two stages are already in flight ----------------
        if constexpr (NB == 3) {
            // three 16-column blocks per wave: the plain epilogues, written out (as gemm192.hip)
            const int cb = n0 + wn * 48 + fq * 4;
            f32x4 bias3[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const int c = cb + n * 16;
                bias3[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias && c < p.N) bias3[n] = *reinterpret_cast<const f32x4*>(p.bias + c);
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int row = m0 + m * 16 + frow;
                if (row >= p.M) continue;
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    const int c = cb + n * 16;
                    if (c >= p.N) continue;
                    f32x4 v = acc[m][n] + bias3[n];
                    if constexpr (EPI == EPI_BIAS_BF16) {
                        u32x2 st;
                        st[0] = pack_bf16x2(v[0], v[1]);
                        st[1] = pack_bf16x2(v[2], v[3]);
                        *reinterpret_cast<u32x2*>((bf16*)p.out + (size_t)row * p.ldo + c) = st;
                    } else {
                        if constexpr (EPI == EPI_BIAS_RES_F32) {
                            if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (size_t)row * p.ldr + c);
                        }
                        store16<true>((float*)p.out + (size_t)row * p.ldo + c, __builtin_bit_cast(u32x4, v), p.flags);
                        if constexpr (EPI == EPI_BIAS_RES_F32) {
                            if (p.out2) {
                                u32x2 st;
                                st[0] = pack_bf16x2(v[0], v[1]);
                                st[1] = pack_bf16x2(v[2], v[3]);
                                *reinterpret_cast<u32x2*>(p.out2 + (size_t)row * p.ldo2 + c) = st;
                            }
                        }
                    }
                }
            }
            continue;
        }
        if constexpr (NB == 4) {
        f32x4 bias4[4];
        load_bias(p, n0 + wn * 64, fq, bias4);
        if (epilogue_fast_ok<EPI>(p, m0, n0, PM, PN)) {
            const int cb = n0 + wn * 64, r0 = m0 + frow;
            if constexpr (EPI == EPI_QKV_ROPE) {
                if (cb >= p.rope_cols) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) epilogue_row<EPI_BIAS_BF16, true>(p, r0 + m * 16, cb, acc[m], bias4, fq);
                } else {
#pragma unroll
                    for (int m = 0; m < 8; ++m) epilogue_row<EPI, true>(p, r0 + m * 16, cb, acc[m], bias4, fq);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) epilogue_row<EPI, true>(p, r0 + m * 16, cb, acc[m], bias4, fq);
            }
        } else if (m0 + PM <= p.M) {
#pragma unroll
            for (int m = 0; m < 8; ++m) epilogue_row<EPI>(p, m0 + m * 16 + frow, n0 + wn * 64, acc[m], bias4, fq);
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int row = m0 + m * 16 + frow;
                if (row < p.M) epilogue_row<EPI>(p, row, n0 + wn * 64, acc[m], bias4, fq);
            }
        }
        }   // NB == 4
    }
}

template <int EPI, int NB>
hipError_t launch_pp_t(const GemmParams& p, hipStream_t s) {
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&gemm_pp_kernel<EPI, NB>)}, PP<NB>::LDS)) return e;
    hipLaunchKernelGGL((gemm_pp_kernel<EPI, NB>), dim3(p.tile_stride), dim3(256), PP<NB>::LDS, s, p);
    return hipGetLastError();
}

}  // namespace

bool gemm_pp_supports(const GemmParams& p, GemmEpilogue epi) {
    if (epi == EPI_GATED_FP8) return false;
    if (p.K % PK) return false;
    // 32-bit per-lane byte offsets into A and W
    if ((size_t)p.M * p.lda * 2 >= (1ull << 32) || (size_t)p.w_rows * p.ldw * 2 >= (1ull << 32)) return false;
    return true;
}

hipError_t launch_gemm_pp(const GemmParams& p_in, GemmEpilogue epi, hipStream_t s) {
    static int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    GemmParams p = p_in;
    // 192-wide tiles (three 16-column blocks per wave) exist for the plain epilogues but are OPT-IN (pp_nb = 3): at N = 768,
    // M = 32768 they make exactly two tiles per workgroup slot where 256-wide tiles make 1.5 rounds, yet measured in-model
    // (tools/step_ab.py ...!3) the out-proj went 90.1 -> 105.2 us, q-proj 48 -> 60, fc2 187 -> 241: the extra LDS-DMA bytes
    // per FLOP of the narrower tile outweigh the idle half round.
    const bool plain = epi == EPI_BIAS_BF16 || epi == EPI_BIAS_RES_F32 || epi == EPI_BIAS_F32;
    const bool use192 = plain && p.N % 192 == 0 && g_pp_nb == 3;
    const int PN = use192 ? 192 : 256;
    p.tiles_m = (p.M + PM - 1) / PM;
    p.tiles_n = (p.N + PN - 1) / PN;
    const int ntiles = p.tiles_m * p.tiles_n;
    p.tile_stride = ntiles < 2 * n_cu ? ntiles : 2 * n_cu;
    p.flags = g_gemm_flags & ~(GF_DIAG_NO_STORE | GF_DIAG_NO_EPILOGUE | GF_DIAG_LINEAR_STORE | GF_DIAG_SMALL_OUT);
    p.group_n = pick_group_n(p.tiles_n, p.flags);
    p.k_splits = 1;
    // Phase offset between the two workgroups of a CU: OFF by default.  Measured (tools/gemm_bench.py, M = 32768): an
    // offset of half a tile changes the K = 768 gated GEMM by < 0.5 % and costs the short GEMMs their delay
    // (q-proj 47.7 -> 57.1 us): the workgroups de-phase on their own at the first tile switch (store drain).
    p.stagger_ticks = g_pp_stagger > 0 && ntiles > n_cu ? g_pp_stagger : 0;
    if (use192) {
        switch (epi) {
            case EPI_BIAS_BF16: return launch_pp_t<EPI_BIAS_BF16, 3>(p, s);
            case EPI_BIAS_RES_F32: return launch_pp_t<EPI_BIAS_RES_F32, 3>(p, s);
            case EPI_BIAS_F32: return launch_pp_t<EPI_BIAS_F32, 3>(p, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (epi) {
        case EPI_BIAS_BF16: return launch_pp_t<EPI_BIAS_BF16, 4>(p, s);
        case EPI_BIAS_RES_F32: return launch_pp_t<EPI_BIAS_RES_F32, 4>(p, s);
        case EPI_QKV_ROPE: return launch_pp_t<EPI_QKV_ROPE, 4>(p, s);
        case EPI_GATED: return launch_pp_t<EPI_GATED, 4>(p, s);
        case EPI_BIAS_F32: return launch_pp_t<EPI_BIAS_F32, 4>(p, s);
        case EPI_BIAS_RELU_BF16: return launch_pp_t<EPI_BIAS_RELU_BF16, 4>(p, s);
        default: break;
    }
    return hipErrorInvalidValue;
}

}  // namespace ditto
