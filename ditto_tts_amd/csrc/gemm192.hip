// gemm192.hip — 256x192x64 persistent bf16 MFMA GEMM for gfx950: the wide-phase structure of gemm256.hip with a
// 192-column tile, for outputs whose width is a multiple of 192 but leaves the 256-wide grid at a fractional number of
// rounds.  At M = 32768, N = 768 (the d x d projections and fc2 of DiTTO-S) 256x256 tiles are 384 = 1.5 per CU — two
// rounds, the second half empty — while 256x192 tiles are 512 = exactly 2 per CU at 3/4 of the work each.
//
//   waves     8 = 2 (M) x 4 (N); wave (wm, wn) owns C rows [128 wm, +128) x cols [48 wn, +48): 8 x 3 accumulators of
//             v_mfma_f32_16x16x32_bf16 = 96 fp32 registers.
//   LDS       112 KiB = 2 K-tile buffers x { A_lo, A_hi (128 rows x 128 B = 16 KiB each) | B (192 rows, 24 KiB) }, same
//             chunk swizzle c ^ ((r>>1)&7) on the DMA source address (48 and 16 are multiples of 16, so a wave's
//             fragment rows keep the (frow>>1) swizzle phase).
//   K-tile    = 2 wide phases of 24 MFMAs:  W1: read B (6) + A0 (8 ds_read_b128) -> C[0..3][*]   W2: read A1 (8) -> C[4..7][*]
//   DMA       per wave 2 loads per A half, 3 per B; order and waits as gemm256's wide schedule with vmcnt(3):
//               W1: o.A_lo+o.A_hi   W2: e'.B, vmcnt(3) -> o complete   W3: e'.A_lo+e'.A_hi   W4: o'.B, vmcnt(3) -> e' complete
//   persistent grid, staggered wave groups, relaxed tile-start wait, staggered start: as gemm256.hip.
//   epilogues EPI_BIAS_RES_F32, EPI_BIAS_F32, EPI_BIAS_BF16 (three 16-column blocks per wave: the RoPE / gated
//             epilogues need 64-column spans and stay on the 256-wide kernels).
#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int H_BYTES = 128 * 64 * 2;            // 16 KiB: one A half
constexpr int B_BYTES = 192 * 64 * 2;            // 24 KiB
constexpr int KT192 = 2 * H_BYTES + B_BYTES;     // 56 KiB
constexpr int LDS192 = 2 * KT192;                // 112 KiB

template <int V>
struct IC192 { static constexpr int value = V; };

#define DITTO_BAR() asm volatile("s_barrier" ::: "memory")

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm192_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nkt = p.K / 64;

    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int srow = lane >> 3, scpos = lane & 7;
    int m0 = 0, n0 = 0;
    auto stage_a = [&](int buf, int half, int kt) {
        if (kt >= nkt) return;
        const int k0b = kt * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;
            const int row = piece * 8 + srow;
            const int c = scpos ^ ((row >> 1) & 7);
            int gr = m0 + half * 128 + row;
            gr = gr < p.M ? gr : p.M - 1;
            glds16((const char*)p.A + (size_t)gr * p.lda * 2 + k0b + c * 16,
                   lds_base + (unsigned)(buf * KT192 + half * H_BYTES + piece * 1024));
        }
    };
    auto stage_b = [&](int buf, int kt) {
        if (kt >= nkt) return;
        const int k0b = kt * 128;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = wid * 3 + i;
            const int row = piece * 8 + srow;      // 0..191
            const int c = scpos ^ ((row >> 1) & 7);
            int gr = n0 + row;
            gr = gr < p.w_rows ? gr : p.w_rows - 1;
            glds16((const char*)p.W + (size_t)gr * p.ldw * 2 + k0b + c * 16,
                   lds_base + (unsigned)(buf * KT192 + 2 * H_BYTES + piece * 1024));
        }
    };
    auto prologue = [&](int tile) {
        int tm, tn;
        tile_to_mn(xcd_remap(tile, ntiles), p.tiles_m, p.tiles_n, p.group_n, tm, tn);
        m0 = tm * 256;
        n0 = tn * 192;
        stage_b(0, 0);
        stage_a(0, 0, 0);
        stage_a(0, 1, 0);
        stage_b(1, 1);
    };

    const int frow = lane & 15, fq = lane >> 4, fswz = frow >> 1;
    const int a_base = wm * H_BYTES + frow * 128;
    const int b_base = 2 * H_BYTES + (wn * 48 + frow) * 128;
    const int coff0 = (fq ^ fswz) << 4, coff1 = ((4 + fq) ^ fswz) << 4;

    u32x4 fa[8], fb[6];
    f32x4 acc[8][3];
    auto read_A = [&](int buf, int ai) {
        const char* base = smem + buf * KT192 + a_base + ai * 64 * 128;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            fa[m * 2 + 0] = *reinterpret_cast<const u32x4*>(base + m * 16 * 128 + coff0);
            fa[m * 2 + 1] = *reinterpret_cast<const u32x4*>(base + m * 16 * 128 + coff1);
        }
    };
    auto read_B = [&](int buf) {
        const char* base = smem + buf * KT192 + b_base;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            fb[n * 2 + 0] = *reinterpret_cast<const u32x4*>(base + n * 16 * 128 + coff0);
            fb[n * 2 + 1] = *reinterpret_cast<const u32x4*>(base + n * 16 * 128 + coff1);
        }
    };
    auto mma = [&](auto AI) {
        constexpr int ai = decltype(AI)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n)
                    acc[ai * 4 + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fb[n * 2 + kk]), __builtin_bit_cast(bf16x8, fa[m * 2 + kk]),
                        acc[ai * 4 + m][n], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto wait_dma = [&](bool ahead) {
        if (ahead) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // stores one wave issues in the epilogue of an interior tile (see gemm256.hip: relaxed tile-start wait)
    constexpr int EPI_STORES = EPI == EPI_BIAS_BF16 ? 16 : 24;
    bool prev_interior = false;

    const int niter = (nkt + 1) / 2;
    int tile = blockIdx.x;
    if (tile < ntiles) prologue(tile);
    if ((p.flags & GF_STAGGER_START) && ntiles >= 8 * p.tile_stride) {
        const int g = (blockIdx.x >> 3) & 3;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)g * (unsigned)p.stagger_ticks;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }

    for (; tile < ntiles; tile += p.tile_stride) {
        const int cur_m0 = m0, cur_n0 = n0;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (prev_interior && nkt > 1 && (p.flags & GF_RELAXED_WAIT)) {
            if constexpr (EPI_STORES == 16) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
        } else {
            wait_dma(nkt > 1);
        }
        DITTO_BAR();
        if (wm == 1) DITTO_BAR();  // stagger the second wave group by one barrier

        for (int it = 0; it < niter; ++it) {
            const int te = 2 * it, to = te + 1;
            const bool odd_valid = to < nkt;
            read_B(0);
            read_A(0, 0);
            stage_a(1, 0, to);
            stage_a(1, 1, to);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            DITTO_BAR();
            mma(IC192<0>{});
            DITTO_BAR();

            read_A(0, 1);
            stage_b(0, te + 2);
            wait_dma(te + 2 < nkt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            DITTO_BAR();
            mma(IC192<1>{});
            DITTO_BAR();

            read_B(1);
            read_A(1, 0);
            stage_a(0, 0, te + 2);
            stage_a(0, 1, te + 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            DITTO_BAR();
            if (odd_valid) mma(IC192<0>{});
            DITTO_BAR();

            read_A(1, 1);
            stage_b(1, to + 2);
            wait_dma(to + 2 < nkt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            DITTO_BAR();
            if (odd_valid) mma(IC192<1>{});
            DITTO_BAR();
        }
        if (wm == 0) DITTO_BAR();  // balance the stagger barrier

        const int next = tile + p.tile_stride;
        if (next < ntiles) prologue(next);

        // ---------------- epilogue: three 16-column blocks per wave ----------------
        prev_interior = (cur_m0 + 256 <= p.M) && (cur_n0 + 192 <= p.N) && !(EPI == EPI_BIAS_RES_F32 && p.out2);
        const int cbase = cur_n0 + wn * 48, c4 = fq * 4;
        f32x4 bias3[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int c = cbase + n * 16 + c4;
            bias3[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && c < p.N) bias3[n] = *reinterpret_cast<const f32x4*>(p.bias + c);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int row = cur_m0 + wm * 128 + m * 16 + frow;
            if (row >= p.M) continue;
            if constexpr (EPI == EPI_BIAS_BF16) {
                u32x2 pk[3];
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    const f32x4 v = acc[m][n] + bias3[n];
                    pk[n][0] = pack_bf16x2(v[0], v[1]);
                    pk[n][1] = pack_bf16x2(v[2], v[3]);
                }
                bf16* rowp = (bf16*)p.out + (size_t)row * p.ldo;
                store_bf16_pair(rowp, cbase, pk[0], pk[1], fq, p.N, p.flags & ~GF_DIAG_LINEAR_STORE, p.ldo);
                const int c = cbase + 32 + c4;
                if (c < p.N) *reinterpret_cast<u32x2*>(rowp + c) = pk[2];
            } else {
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    const int c = cbase + n * 16 + c4;
                    if (c >= p.N) continue;
                    f32x4 v = acc[m][n] + bias3[n];
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
    }
}

template <int EPI>
hipError_t launch192_t(const GemmParams& p, hipStream_t s) {
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&gemm192_kernel<EPI>)}, LDS192)) return e;
    hipLaunchKernelGGL((gemm192_kernel<EPI>), dim3(p.tile_stride), dim3(512), LDS192, s, p);
    return hipGetLastError();
}

}  // namespace

bool gemm192_supports(GemmEpilogue epi) { return epi == EPI_BIAS_BF16 || epi == EPI_BIAS_RES_F32 || epi == EPI_BIAS_F32; }

hipError_t launch_gemm192(const GemmParams& p_in, GemmEpilogue epi, hipStream_t s) {
    static int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    GemmParams p = p_in;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 191) / 192;
    const int ntiles = p.tiles_m * p.tiles_n;
    p.tile_stride = ntiles < n_cu ? ntiles : n_cu;
    p.flags = g_gemm_flags & ~(GF_DIAG_NO_STORE | GF_DIAG_NO_EPILOGUE | GF_DIAG_LINEAR_STORE | GF_DIAG_SMALL_OUT);
    p.group_n = pick_group_n(p.tiles_n, p.flags);
    p.stagger_ticks = (int)((p.K / 64 * 1.1 + 6.0) * 100.0 / 4.0);
    p.k_splits = 1;
    switch (epi) {
        case EPI_BIAS_BF16: return launch192_t<EPI_BIAS_BF16>(p, s);
        case EPI_BIAS_RES_F32: return launch192_t<EPI_BIAS_RES_F32>(p, s);
        case EPI_BIAS_F32: return launch192_t<EPI_BIAS_F32>(p, s);
        default: break;
    }
    return hipErrorInvalidValue;
}

}  // namespace ditto
