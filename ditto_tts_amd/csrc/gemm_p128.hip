// gemm_p128.hip — persistent 256(M) x 128(N) x 64 bf16 MFMA GEMM for gfx950: the structure for N = d GEMMs
// (cross-attn q / out-proj, fc2, the final K-concatenated projection), where 256x256 tiles give only
// 1.5 tiles per CU (M = 32768, N = 768: 384 tiles) and this shape gives exactly 3.
//
// Same contract and epilogues as gemm.hip / gemm256.hip.  Same machinery as gemm256.hip (LDS-DMA behind a
// counted vmcnt, raw barriers, two wave groups staggered by one barrier, source-side swizzle, persistent
// workgroups that start the next tile's DMA before the epilogue); what differs:
//
//   waves     8 = 4 (M) x 2 (N); wave (wm, wn) owns C rows [64 wm, +64) x cols [64 wn, +64):
//             4 x 4 accumulators of v_mfma_f32_16x16x32_bf16 = 64 fp32 registers (16 ds_read_b128 per 32 MFMA).
//   LDS       144 KiB = a RING of 3 K-tile buffers x 3 half-tiles (A_lo, A_hi, B; 128 rows x 128 B = 16 KiB each).
//   K-tile    = 2 phases of 16 MFMAs:  P1: read A (8) + B0 (4) -> C[:, 0:32]     P2: read B1 (4) -> C[:, 32:64]
//             so a buffer's A halves are last read in P1 and its B half in P2.
//   DMA       while tile t is multiplied: P1 issues tile t+2's A_lo + A_hi, P2 issues tile t+2's B — into the
//             buffer tile t-1 used, 2 phases after the last read of each half — then P2 waits vmcnt(6):
//             everything but tile t+2's own 6 loads has landed, i.e. tile t+1 is complete for the next P1.
//             Every half-tile therefore has >= 2 full phases (4 barrier intervals) between issue and wait.
#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int HB = 128 * 64 * 2;   // half-tile bytes: 16 KiB
constexpr int KTB = 3 * HB;        // A_lo | A_hi | B = 48 KiB
constexpr int LDS_P128 = 3 * KTB;  // 144 KiB

template <int V>
struct ICp { static constexpr int value = V; };

#define DITTO_BAR() asm volatile("s_barrier" ::: "memory")

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_p128_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;   // wave group (stagger) = wid >> 2 = wm >> 1
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nkt = p.K / 64;

    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;   // LDS byte address of the dynamic region
    const int srow = lane >> 3, scpos = lane & 7;
    int m0 = 0, n0 = 0;   // origin of the tile whose DMA is being issued
    // half: 0 = A rows 0..127, 1 = A rows 128..255, 2 = B rows 0..127 (of the 128-column tile)
    auto stage = [&](auto BUF, auto HALF, int kt) {
        constexpr int half = decltype(HALF)::value, buf = decltype(BUF)::value;
        if (kt >= nkt) return;
        const int k0 = kt * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;
            const int row = piece * 8 + srow;
            const int c = scpos ^ ((row >> 1) & 7);
            const bf16* src;
            if constexpr (half < 2) {
                int gr = m0 + half * 128 + row;
                gr = gr < p.M ? gr : p.M - 1;
                src = p.A + (size_t)gr * p.lda + k0 + c * 8;
            } else {
                int gr = n0 + row;
                gr = gr < p.w_rows ? gr : p.w_rows - 1;
                src = p.W + (size_t)gr * p.ldw + k0 + c * 8;
            }
            glds16(src, lds_base + (unsigned)(buf * KTB + half * HB + piece * 1024));
        }
    };
    auto stage_tile = [&](auto BUF, int kt) {   // all three halves of K-tile kt into ring slot BUF
        stage(BUF, ICp<0>{}, kt);
        stage(BUF, ICp<1>{}, kt);
        stage(BUF, ICp<2>{}, kt);
    };
    auto prologue = [&](int tile) {
        int tm, tn;
        tile_to_mn(xcd_remap(tile, ntiles), p.tiles_m, p.tiles_n, p.group_n, tm, tn);
        m0 = tm * 256;
        n0 = tn * 128;
        stage_tile(ICp<0>{}, 0);
        stage_tile(ICp<1>{}, 1);
    };

    const int frow = lane & 15, fq = lane >> 4, fswz = frow >> 1;
    const int a_base = (wm >> 1) * HB + ((wm & 1) * 64 + frow) * 128;   // + m*16*128
    const int b_base = 2 * HB + (wn * 64 + frow) * 128;                 // + (bj*32 + n*16)*128
    const int coff0 = ((0 + fq) ^ fswz) << 4, coff1 = ((4 + fq) ^ fswz) << 4;

    bf16x8 af[8], bfr[4];
    f32x4 acc[4][4];

    auto read_A = [&](auto BUF) {
        constexpr int buf = decltype(BUF)::value;
        const char* base = smem + buf * KTB + a_base;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            af[m * 2 + 0] = *reinterpret_cast<const bf16x8*>(base + m * 16 * 128 + coff0);
            af[m * 2 + 1] = *reinterpret_cast<const bf16x8*>(base + m * 16 * 128 + coff1);
        }
    };
    auto read_B = [&](auto BUF, auto BJ) {
        constexpr int bj = decltype(BJ)::value, buf = decltype(BUF)::value;
        const char* base = smem + buf * KTB + b_base + bj * 32 * 128;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            bfr[n * 2 + 0] = *reinterpret_cast<const bf16x8*>(base + n * 16 * 128 + coff0);
            bfr[n * 2 + 1] = *reinterpret_cast<const bf16x8*>(base + n * 16 * 128 + coff1);
        }
    };
    auto mma = [&](auto BJ) {
        constexpr int bj = decltype(BJ)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][bj * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[n * 2 + kk], af[m * 2 + kk],
                                                                                 acc[m][bj * 2 + n], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    constexpr int EPI_STORES = EPI == EPI_BIAS_BF16 ? 8 : EPI == EPI_QKV_ROPE ? 8 : EPI == EPI_GATED ? 4 : 16;
    bool prev_interior = false;

    int tile = blockIdx.x;
    if (tile < ntiles) prologue(tile);

    for (; tile < ntiles; tile += p.tile_stride) {
        const int cur_m0 = m0, cur_n0 = n0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        // K-tile 0 complete: everything older than tile 1's 6 loads (+ the previous tile's epilogue stores,
        // which were issued after this tile's prologue DMA)
        if (nkt > 1) {
            if (prev_interior && (p.flags & GF_RELAXED_WAIT)) {
                if constexpr (EPI_STORES == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else if constexpr (EPI_STORES == 8) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        DITTO_BAR();
        if (wid >= 4) DITTO_BAR();   // stagger the second wave group by one barrier

        // One K-tile from ring slot BUF; its DMA look-ahead (K-tile kt+2) goes to slot BUF2 = (BUF + 2) % 3.
        // The slots are COMPILE-TIME constants on purpose: with a run-time ring index hipcc cannot prove that
        // the in-flight LDS-DMA does not alias the ds_reads and drains it with s_waitcnt vmcnt(0) before the
        // first read of every K-tile (seen in the .s; the kernel then ran at a third of the MFMA rate).
        auto ktile = [&](int kt, auto BUF, auto BUF2) {
            // P1
            read_B(BUF, ICp<0>{});
            read_A(BUF);
            stage(BUF2, ICp<0>{}, kt + 2);
            stage(BUF2, ICp<1>{}, kt + 2);
            DITTO_BAR();
            mma(ICp<0>{});
            DITTO_BAR();
            // P2
            read_B(BUF, ICp<1>{});
            stage(BUF2, ICp<2>{}, kt + 2);
            if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // K-tile kt+1 has landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DITTO_BAR();
            mma(ICp<1>{});
            DITTO_BAR();
        };
        for (int kt = 0; kt < nkt; kt += 3) {
            ktile(kt, ICp<0>{}, ICp<2>{});
            if (kt + 1 < nkt) ktile(kt + 1, ICp<1>{}, ICp<0>{});
            if (kt + 2 < nkt) ktile(kt + 2, ICp<2>{}, ICp<1>{});
        }
        if (wid < 4) DITTO_BAR();    // balance the stagger barrier: every LDS read of this tile has retired

        const int next = tile + p.tile_stride;
        if (next < ntiles) prologue(next);

        prev_interior = (cur_m0 + 256 <= p.M) && (cur_n0 + 128 <= p.N) && !p.out2 &&
                        !(p.flags & (GF_DIAG_NO_STORE | GF_DIAG_NO_EPILOGUE)) && nkt > 1;
        if (p.flags & GF_DIAG_NO_EPILOGUE) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) asm volatile("" ::"v"(acc[m][n]));
            continue;
        }
        f32x4 bias4[4];
        load_bias(p, cur_n0 + wn * 64, fq, bias4);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = cur_m0 + wm * 64 + m * 16 + frow;
            if (row < p.M) epilogue_row<EPI>(p, row, cur_n0 + wn * 64, acc[m], bias4, fq);
        }
    }
}

template <int EPI>
hipError_t launch_p128_t(const GemmParams& p, hipStream_t s) {
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&gemm_p128_kernel<EPI>)}, LDS_P128)) return e;
    hipLaunchKernelGGL((gemm_p128_kernel<EPI>), dim3(p.tile_stride), dim3(512), LDS_P128, s, p);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm_p128(const GemmParams& p_in, GemmEpilogue epi, hipStream_t s) {
    static int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    GemmParams p = p_in;
    const int ntiles = p.tiles_m * p.tiles_n;
    p.tile_stride = ntiles < n_cu ? ntiles : n_cu;
    p.flags = g_gemm_flags;
    p.group_n = pick_group_n(p.tiles_n, p.flags);
    p.stagger_ticks = 0;
    switch (epi) {
        case EPI_BIAS_BF16: return launch_p128_t<EPI_BIAS_BF16>(p, s);
        case EPI_BIAS_RES_F32: return launch_p128_t<EPI_BIAS_RES_F32>(p, s);
        case EPI_QKV_ROPE: return launch_p128_t<EPI_QKV_ROPE>(p, s);
        case EPI_GATED: return launch_p128_t<EPI_GATED>(p, s);
        case EPI_BIAS_F32: return launch_p128_t<EPI_BIAS_F32>(p, s);
        default: break;
    }
    return hipErrorInvalidValue;
}

}  // namespace ditto
