// gemm_o3.hip — 128x256x32 bf16 MFMA GEMM at THREE workgroups per CU (gfx950).
//
// Same contract and epilogues as gemm.hip.  Why another structure: the persistent 256x256 kernel (gemm256.hip) keeps
// its two waves per SIMD in lockstep — both in the main loop, then both in the epilogue — so every tile pays
// MFMA time + epilogue time back to back (K = 768: 17 us + 8..25 us), and a grid of 1.5 or 4.5 tiles per CU
// quantises up.  Here the tile is small enough in LDS (K-tile of 32: 2 x 24 KiB = 48 KiB) and registers (<= 168)
// for three INDEPENDENT workgroups per CU: at any moment one is likely in its epilogue (HBM-bound) while the others
// multiply, three waves per SIMD hide LDS / DMA latency without hand scheduling, and at M = 32768 the grids of the
// model are whole rounds (QKV 2304 tiles = 3.0 rounds of 768 slots, d x d and fc2 768 = 1.0, gated MLP 6144 = 8.0).
//
//   tile      128 (M) x 256 (N) x 32 (K); 256 threads = 4 waves in 2 x 2, each wave 64 x 128:
//             4 x 8 accumulators of v_mfma_f32_16x16x32_bf16 = 128 fp32 registers (same LDS-read : MFMA ratio as the
//             256^2 kernel's 128 x 64 wave tile: 12 ds_read_b128 per 32 MFMAs).
//   LDS       rows of 64 B (32 bf16); 16-B chunk c of row r sits at c ^ (-(r>>2)&3): the 16 lanes of a ds_read_b128 lane
//             group fall in 16 distinct 16-B slots of the 256-B bank row (the hardware serves
//             ds_read_b128 in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: with lane = 16 fq + frow, a
//             group holds for every residue frow & 3 the row quads q = 0, 3 at k-chunk fq and q = 1, 2 at fq ^ 1, and
//             q -> -q & 3 sends those four to four different slots; the first version used (r>>2)&3: 2-way conflicts).  Lane-linear image, swizzle on the
//             DMA's source address.  Two buffers, tile kt+1 in flight while kt is multiplied, two barriers per K-tile.
//   fragments A (4 x 16 B) per K-tile up front, W one 16-column block at a time (8 x 16 B, just in time) to stay
//             under 168 registers.
#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int OM = 128, ON = 256, OK = 32;
constexpr int O_A_BYTES = OM * OK * 2;              // 8 KiB
constexpr int O_W_BYTES = ON * OK * 2;              // 16 KiB
constexpr int O_BUF = O_A_BYTES + O_W_BYTES;        // 24 KiB
constexpr int O_LDS = 2 * O_BUF;                    // 48 KiB

template <int EPI>
__global__ __launch_bounds__(256, 3) void gemm_o3_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int nwg = p.tiles_m * p.tiles_n;
    int tm, tn;
    tile_to_mn(xcd_remap(blockIdx.x, nwg), p.tiles_m, p.tiles_n, p.group_n, tm, tn);
    const int m0 = tm * OM, n0 = tn * ON;

    f32x4 acc[4][8];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one K-tile = 24 pieces of 1 KiB (16 rows x 64 B), 6 per wave: pieces 0..7 = A, 8..23 = W.  A load's address is
    // (A or W base + k0 bytes) [scalar] + this lane's row / chunk offset [one 32-bit register per piece, set once]
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    unsigned voff[6];
    {
        const int prow = lane >> 2, cpos = lane & 3;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int piece = wid * 6 + i;   // wave-uniform: pieces of one wave are all A or all W except wave 1 (6..11)
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
    }
    auto stage = [&](int buf, int kt) {
        const char* abase = (const char*)p.A + (size_t)kt * (OK * 2);
        const char* wbase = (const char*)p.W + (size_t)kt * (OK * 2);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int piece = wid * 6 + i;
            glds16_so(voff[i], piece < 8 ? abase : wbase, lds_base + (unsigned)(buf * O_BUF + piece * 1024));
        }
    };
    const int nkt = p.K / OK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment addressing: lane reads row (lane&15) of a 16-row block, k-chunk (lane>>4) of the 32-wide K-tile
    const int frow = lane & 15, fq = lane >> 4;
    const int coff = (fq ^ ((0 - (frow >> 2)) & 3)) << 4;    // (16*blk + frow)>>2 & 3 == (frow>>2)&3
    const int a_off = (wr * 64 + frow) * 64 + coff;
    const int w_off = O_A_BYTES + (wc * 128 + frow) * 64 + coff;

    for (int kt = 0; kt < nkt; ++kt) {
        const char* cur = smem + (kt & 1) * O_BUF;
        if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
        bf16x8 af[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const bf16x8*>(cur + a_off + m * 16 * 64);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(cur + w_off + n * 16 * 64);
#pragma unroll
            for (int m = 0; m < 4; ++m)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[m], acc[m][n], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt+1 have landed
        __syncthreads();
    }

    // ---------------- epilogue (gemm_common.h): two 64-column spans per wave ----------------
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int cbase = n0 + wc * 128 + half * 64;
        f32x4 bias4[4];
        load_bias(p, cbase, fq, bias4);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = m0 + wr * 64 + m * 16 + frow;
            const f32x4(&a4)[4] = *reinterpret_cast<const f32x4(*)[4]>(&acc[m][half * 4]);
            if (row < p.M) epilogue_row<EPI>(p, row, cbase, a4, bias4, fq);
        }
    }
}

template <int EPI>
hipError_t launch_o3_t(const GemmParams& p, hipStream_t s) {
    hipLaunchKernelGGL((gemm_o3_kernel<EPI>), dim3(p.tiles_m * p.tiles_n), dim3(256), O_LDS, s, p);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm_o3(const GemmParams& p_in, GemmEpilogue epi, hipStream_t s) {
    GemmParams p = p_in;
    p.tiles_m = (p.M + OM - 1) / OM;
    p.tiles_n = (p.N + ON - 1) / ON;
    p.flags = g_gemm_flags & ~(GF_DIAG_NO_STORE | GF_DIAG_NO_EPILOGUE | GF_DIAG_LINEAR_STORE | GF_DIAG_SMALL_OUT);
    p.group_n = pick_group_n(p.tiles_n, p.flags);
    p.k_splits = 1;
    switch (epi) {
        case EPI_BIAS_BF16: return launch_o3_t<EPI_BIAS_BF16>(p, s);
        case EPI_BIAS_RES_F32: return launch_o3_t<EPI_BIAS_RES_F32>(p, s);
        case EPI_QKV_ROPE: return launch_o3_t<EPI_QKV_ROPE>(p, s);
        case EPI_GATED: return launch_o3_t<EPI_GATED>(p, s);
        case EPI_BIAS_F32: return launch_o3_t<EPI_BIAS_F32>(p, s);
        default: break;
    }
    return hipErrorInvalidValue;
}

}  // namespace ditto
