// gemm.hip — bf16 MFMA GEMM family for gfx950:  out[M,N] = A[M,K] * W[N,K]^T (+ fused epilogue)
//
// Replaces every F.linear on the DiT path (reference src/components/DiT.py:112-114,153-155,
// src/model/DiTTO.py:83,93 and torch MHA's in/out projections).  W is nn.Linear.weight as stored
// ([out, in], K-contiguous = "B^T input" form), so both operands are read along K.
//
// Roofline: MFMA-bound (dense bf16 ~2.5 PFLOP/s).  Algorithmic FLOPs = 2*M*N*K per launch.
//
// Structure (v1, "128^2 two-barrier" of cdna_hip_programming.md §5):
//   * tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves in 2x2, each wave a 64x64 sub-tile as
//     4x4 v_mfma_f32_16x16x32_bf16 accumulators (64 fp32 regs).
//   * staging by LDS-DMA: __builtin_amdgcn_global_load_lds, 16 B per lane, two LDS buffers
//     (2 x (16 KiB A + 16 KiB W) = 64 KiB -> 2 workgroups / CU); tile kt+1 is in flight while
//     tile kt is multiplied.
//   * LDS image is lane-linear (DMA writes base + lane*16), so the bank swizzle is applied to the
//     per-lane SOURCE address and again on the ds_read_b128 (rule 21): 16-B chunk c of row r is
//     stored at chunk position c ^ ((r>>1)&7) of its 128-B row -> every ds_read_b128 lane group
//     hits 16 distinct 16-B slots of the 256-B bank row (conflict-free).
//   * operands are SWAPPED in the MFMA (weights as the instruction's A, activations as B), so the
//     accumulator holds C^T: a lane owns 4 CONSECUTIVE output columns of one row -> 8/16-byte
//     stores, and the RoPE partner (col +-32) and the fc1/gate partner (col +-16) of a value are in
//     the same lane (no cross-lane traffic in the fused epilogues).
//   * XCD-aware, bijective block->tile map: each XCD (private 4 MiB L2) walks a contiguous range of
//     M-panels, sweeping N inside a panel, so an A panel is fetched from HBM once per XCD.
#include <cstdlib>

#include "gemm_common.h"

namespace ditto {

// 0 = automatic, 128 / 256 = force that tile structure (ditto_set_option("gemm_tile", v); env DITTO_GEMM seeds it)
// GF_STAGGER_START (8) is no longer on by default: after the gated epilogue got shorter the staggered start costs more
// than the store bursts it spreads — same-process A/B at C2 B = 32 (tools/step_ab.py, 8 rounds x 25 steps): gated GEMM
// 291.7 -> 279.1 us per launch, step 13.06 -> 12.99 ms.
int g_gemm_flags = [] { const char* e = getenv("DITTO_GEMM_FLAGS"); return e ? atoi(e) : (GF_RELAXED_WAIT | GF_STORE_NT | GF_WIDE_PHASE); }();   // 321; the variable: A/B runs of whole programs (bench.py)
int g_gemm_group = 0;
int g_pp_stagger = -1;
int g_pp_nb = 0;
// default 3: measured in-model at C2, B = 32 (tools/step_ab.py ~mask): 13.02 -> 12.73 (1) / 12.77 (2) / 12.40 ms (3)
int g_fr_mask = [] { const char* e = getenv("DITTO_FR_MASK"); return e ? atoi(e) : 3; }();
int g_fr_dgrad = [] { const char* e = getenv("DITTO_FR_DGRAD"); return e ? atoi(e) : 3; }();
int g_train_flags = [] { const char* e = getenv("DITTO_TRAIN_FLAGS"); return e ? atoi(e) : 0; }();
int g_fr_rot = [] { const char* e = getenv("DITTO_FR_ROT"); return e ? atoi(e) : 1; }();
int g_fr_tile = [] { const char* e = getenv("DITTO_FR_TILE"); const int v = e ? atoi(e) : 0; return v == 64 || v == 130 ? v : 0; }();   // validated like ditto_set_option
int g_fr64_maxk = [] { const char* e = getenv("DITTO_FR64_MAXK"); return e ? atoi(e) : 1 << 30; }();
int g_fr_stagger = [] { const char* e = getenv("DITTO_FR_STAGGER"); return e ? atoi(e) : 0; }();   // off: worth 6 us isolated at 512 tiles, nothing in the model, and a late second workgroup is pure tail when only a few CUs get one
int g_fr_u_fp8 = 0;
int g_fr_class_rows = 0;   // kernels.h fr_pays: rows of the unsplit batch whose kernel class every launch takes (0 = its own)
thread_local CallOpts t_opts = {-1, -1, -1, -1};   // kernels.h: the calling thread's per-call options (-1 = the process default)
int g_pp_mask = [] { const char* e = getenv("DITTO_PP_MASK"); return e ? atoi(e) : -1; }();   // -1 = built-in rule
int g_gemm_tile = [] { const char* e = getenv("DITTO_GEMM"); return e ? atoi(e) : 0; }();

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
thread_local int g_batch_y = 1;   // grid.y of the launch being built (launch_gemm sets it; the launchers are templates)
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;        // A + W
constexpr int GEMM_LDS = 2 * BUF_BYTES;          // double buffered: 64 KiB

// Issue the LDS-DMA loads of one (A, W) K-tile.  Lane -> (row, chunk position) is linear in LDS;
// the source chunk is the swizzled one.
DITTO_DEV void stage_tile(const GemmParams& p, char* buf, int m0, int n0, int k0, int wid, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wid * 4 + i;                    // 1 KiB piece = 8 rows x 128 B
        const int row = piece * 8 + (lane >> 3);
        const int cpos = lane & 7;
        const int c = cpos ^ ((row >> 1) & 7);
        int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;  // clamp: rows past M are computed, never stored
        int wr = n0 + row; wr = wr < p.w_rows ? wr : p.w_rows - 1;
        const bf16* ga = p.A + (size_t)ar * p.lda + k0 + c * 8;
        const bf16* gw = p.W + (size_t)wr * p.ldw + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)ga, (lds_ptr_t)(buf + piece * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)gw, (lds_ptr_t)(buf + TILE_BYTES + piece * 1024), 16, 0, 0);
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;

    if (gridDim.y > 1) {   // batched: this workgroup's matrices
        const long long zo = blockIdx.y / p.batch_inner, zi = blockIdx.y % p.batch_inner;
        p.A += zo * p.sA[0] + zi * p.sA[1];
        p.W += zo * p.sW[0] + zi * p.sW[1];
        p.out = (char*)p.out + (zo * p.sO[0] + zi * p.sO[1]) * p.out_esz;
        if (p.residual) p.residual += zo * p.sR[0] + zi * p.sR[1];
    }
    const int nwg = p.tiles_m * p.tiles_n;
    int tm, tn;
    const int split = p.k_splits > 1 ? blockIdx.x / nwg : 0;
    tile_to_mn(xcd_remap(p.k_splits > 1 ? blockIdx.x % nwg : blockIdx.x, nwg), p.tiles_m, p.tiles_n, p.group_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    int nkt = p.K / BK, kbase = 0;
    if (p.k_splits > 1) {   // split-K: this workgroup owns K-tiles [kbase, kbase + nkt) and its own output slice
        const int per = (nkt + p.k_splits - 1) / p.k_splits;
        kbase = split * per;
        nkt = nkt - kbase < per ? nkt - kbase : per;
        if (nkt < 0) nkt = 0;
        p.out = (float*)p.out + (size_t)split * p.split_stride;
    }
    if (nkt > 0) stage_tile(p, smem, m0, n0, kbase * BK, wid, lane);
    __syncthreads();  // (hipcc drains the LDS-DMA with vmcnt(0) here)

    // per-lane fragment addressing: row (lane&15) of a 16-row block, k-chunk (lane>>4) of a 32-wide k-step
    const int frow = lane & 15, fq = lane >> 4, fswz = frow >> 1;
    const int a_row_off = (wr * 64 + frow) * 128;
    const int w_row_off = TILE_BYTES + (wc * 64 + frow) * 128;

    for (int kt = 0; kt < nkt; ++kt) {
        char* cur = smem + (kt & 1) * BUF_BYTES;
        if (kt + 1 < nkt) stage_tile(p, smem + ((kt + 1) & 1) * BUF_BYTES, m0, n0, (kbase + kt + 1) * BK, wid, lane);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = ((kk * 4 + fq) ^ fswz) << 4;
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(cur + a_row_off + i * 16 * 128 + coff);
                wf[i] = *reinterpret_cast<const bf16x8*>(cur + w_row_off + i * 16 * 128 + coff);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], af[m], acc[m][n], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---------------- epilogue (gemm_common.h): lane owns row (lane&15) of each 16-row block ----------------
    f32x4 bias4[4];
    load_bias(p, n0 + wc * 64, fq, bias4);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int row = m0 + wr * 64 + m * 16 + frow;
        if (row < p.M) epilogue_row<EPI>(p, row, n0 + wc * 64, acc[m], bias4, fq);
    }
}

// ------------------------------------------------------------------------------------------------
// gemm128_deep: the same 128x128x64 tile for SMALL grids (<= one workgroup per CU: batch-1 serving, where the
// whole GEMM is one round and its time is the latency of one workgroup's K loop).  With one K-tile of look-ahead a
// lone workgroup exposes the full DMA latency on every K step (measured at M = 1024: ~1.6 us per step, 20 us for
// K = 768 whatever the grid size).  Here the LDS ring is 4 tiles deep (128 KiB, affordable at one workgroup per CU), the
// DMA runs 3 tiles ahead behind a COUNTED vmcnt (asm LDS-DMA: in-order retire), one barrier per K step.
// ------------------------------------------------------------------------------------------------
constexpr int DEEP_NB = 4;
constexpr int DEEP_LDS = DEEP_NB * BUF_BYTES;   // 128 KiB

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm128_deep_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    if (gridDim.y > 1) {   // batched: this workgroup's matrices
        const long long zo = blockIdx.y / p.batch_inner, zi = blockIdx.y % p.batch_inner;
        p.A += zo * p.sA[0] + zi * p.sA[1];
        p.W += zo * p.sW[0] + zi * p.sW[1];
        p.out = (char*)p.out + (zo * p.sO[0] + zi * p.sO[1]) * p.out_esz;
        if (p.residual) p.residual += zo * p.sR[0] + zi * p.sR[1];
    }
    const int nwg = p.tiles_m * p.tiles_n;
    int tm, tn;
    const int split = p.k_splits > 1 ? blockIdx.x / nwg : 0;
    tile_to_mn(xcd_remap(p.k_splits > 1 ? blockIdx.x % nwg : blockIdx.x, nwg), p.tiles_m, p.tiles_n, p.group_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    int nkt = p.K / BK, kbase = 0;
    if (p.k_splits > 1) {
        const int per = (nkt + p.k_splits - 1) / p.k_splits;
        kbase = split * per;
        nkt = nkt - kbase < per ? nkt - kbase : per;
        if (nkt < 0) nkt = 0;
        p.out = (float*)p.out + (size_t)split * p.split_stride;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    // per-lane source offsets of this wave's 4 A pieces and 4 W pieces (1 KiB = 8 rows x 128 B each), set once
    unsigned aoff[4], woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wid * 4 + i;
        const int row = piece * 8 + (lane >> 3), cpos = lane & 7;
        const int c = cpos ^ ((row >> 1) & 7);
        int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
        int wrow = n0 + row; wrow = wrow < p.w_rows ? wrow : p.w_rows - 1;
        aoff[i] = (unsigned)(((size_t)ar * p.lda + c * 8) * 2);
        woff[i] = (unsigned)(((size_t)wrow * p.ldw + c * 8) * 2);
    }
    auto stage = [&](int buf, int kt) {   // 8 LDS-DMA loads per wave
        const char* ab = (const char*)p.A + (size_t)(kbase + kt) * (BK * 2);
        const char* wb = (const char*)p.W + (size_t)(kbase + kt) * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wid * 4 + i;
            glds16_so(aoff[i], ab, lds_base + (unsigned)(buf * BUF_BYTES + piece * 1024));
            glds16_so(woff[i], wb, lds_base + (unsigned)(buf * BUF_BYTES + TILE_BYTES + piece * 1024));
        }
    };
#pragma unroll
    for (int t = 0; t < DEEP_NB - 1; ++t)
        if (t < nkt) stage(t, t);

    const int frow = lane & 15, fq = lane >> 4, fswz = frow >> 1;
    const int a_row_off = (wr * 64 + frow) * 128;
    const int w_row_off = TILE_BYTES + (wc * 64 + frow) * 128;

    for (int kt = 0; kt < nkt; ++kt) {
        // tile kt has landed once at most the loads of tiles kt+1, kt+2 (8 each) are still in flight
        const int ahead = nkt - 1 - kt;
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // every wave's pieces of tile kt are visible; every wave is done with tile kt-1's buffer
        if (kt + DEEP_NB - 1 < nkt) stage((kt + DEEP_NB - 1) % DEEP_NB, kt + DEEP_NB - 1);
        const char* cur = smem + (kt % DEEP_NB) * BUF_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = ((kk * 4 + fq) ^ fswz) << 4;
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(cur + a_row_off + i * 16 * 128 + coff);
                wf[i] = *reinterpret_cast<const bf16x8*>(cur + w_row_off + i * 16 * 128 + coff);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], af[m], acc[m][n], 0, 0, 0);
        }
    }

    f32x4 bias4[4];
    load_bias(p, n0 + wc * 64, fq, bias4);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int row = m0 + wr * 64 + m * 16 + frow;
        if (row < p.M) epilogue_row<EPI>(p, row, n0 + wc * 64, acc[m], bias4, fq);
    }
}

template <int EPI>
hipError_t launch_deep_t(const GemmParams& p, hipStream_t s) {
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&gemm128_deep_kernel<EPI>)}, DEEP_LDS)) return e;
    hipLaunchKernelGGL((gemm128_deep_kernel<EPI>),
                       dim3(p.tiles_m * p.tiles_n * (p.k_splits > 1 ? p.k_splits : 1), g_batch_y),
                       dim3(256), DEEP_LDS, s, p);
    return hipGetLastError();
}

template <int EPI>
hipError_t launch_t(const GemmParams& p, hipStream_t s) {
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&gemm128_kernel<EPI>)}, GEMM_LDS)) return e;
    hipLaunchKernelGGL((gemm128_kernel<EPI>), dim3(p.tiles_m * p.tiles_n * (p.k_splits > 1 ? p.k_splits : 1), g_batch_y),
                       dim3(256), GEMM_LDS, s, p);
    return hipGetLastError();
}

}  // namespace

// The fused launch pays where the 256 x 256 kernel is the dgrad's kernel anyway (launch_gemm's wide-output rule below).
bool gemm_gated_bwd_fused_ok(int M, int F) {
    if (M <= 0 || F <= 0 || F % 256) return false;
    const long t256 = (long)((M + 255) / 256) * (F / 256);
    return (g_gemm_tile == 0 || g_gemm_tile == 256) && F >= 2048 && t256 >= 144;
}

hipError_t launch_gemm(const GemmArgs& a, GemmEpilogue epi, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return hipErrorInvalidValue;
    if (a.fp8) {
        if (a.K % 128 || a.N % 16 || a.lda % 16 || a.ldw % 16) return hipErrorInvalidValue;
    } else if (a.K % BK || a.N % 16 || a.lda % 8 || (a.ldw % 8)) {
        return hipErrorInvalidValue;
    }
    GemmParams p;
    p.A = (const bf16*)a.A; p.lda = a.lda; p.W = (const bf16*)a.W; p.bias = a.bias; p.wscale = a.wscale;
    p.ldw = a.ldw ? a.ldw : a.K; p.w_rows = a.w_rows ? a.w_rows : a.N;
    p.residual = a.residual; p.ldr = a.ldr; p.out = a.out; p.ldo = a.ldo;
    p.out2 = (bf16*)a.out2_bf16; p.ldo2 = a.ldo2;
    p.rope_cos = a.rope_cos; p.rope_sin = a.rope_sin; p.rope_rpb = a.rope_rows_per_batch; p.rope_cols = a.rope_cols;
    p.rope_freq_rev = a.rope_freq_rev;
    p.M = a.M; p.N = a.N; p.K = a.K;
    p.k_splits = a.k_splits > 1 ? a.k_splits : 1; p.split_stride = a.split_stride;
    const int nbatch = (a.batch_outer > 1 ? a.batch_outer : 1) * (a.batch_inner > 1 ? a.batch_inner : 1);
    p.batch_inner = a.batch_inner > 1 ? a.batch_inner : 1;
    for (int i = 0; i < 2; ++i) { p.sA[i] = a.sA[i]; p.sW[i] = a.sW[i]; p.sO[i] = a.sO[i]; p.sR[i] = a.sR[i]; }
    p.out_esz = (epi == EPI_BIAS_RES_F32 || epi == EPI_BIAS_F32) ? 4 : 2;
    p.pre = (const bf16*)a.pre_bf16; p.ldpre = a.ldpre; p.colpart = a.colsum_partial;
    if (nbatch > 1 && (a.fp8 || p.k_splits > 1 || nbatch > 65535)) return hipErrorInvalidValue;
    g_batch_y = nbatch;
    if (p.k_splits > 1 && (epi != EPI_BIAS_F32 || a.bias || a.fp8)) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_QKV_ROPE:
            if (a.N % 64 || a.rope_cols % 64 || !a.rope_cos || !a.rope_sin || a.rope_rows_per_batch <= 0)
                return hipErrorInvalidValue;
            break;
        case EPI_GATED:
        case EPI_GATED_FP8:
            if (a.N % 32 || !a.bias) return hipErrorInvalidValue;
            break;
        case EPI_GATED_PRE:   // the two training epilogues exist on the 256 x 256 structure only
            if (a.N % 256 || !a.bias || !a.out2_bf16 || a.fp8 || nbatch > 1 || p.k_splits > 1) return hipErrorInvalidValue;
            break;
        case EPI_GATED_BWD:
            if (a.N % 256 || a.bias || !a.pre_bf16 || !a.colsum_partial || a.ldpre % 8 || a.ldo % 8 || a.fp8 || nbatch > 1 ||
                p.k_splits > 1)
                return hipErrorInvalidValue;
            break;
        default: break;
    }
    if (epi == EPI_GATED_PRE || epi == EPI_GATED_BWD) {
        p.tiles_m = (a.M + 255) / 256;
        p.tiles_n = a.N / 256;
        return launch_gemm256(p, epi, s);
    }
    // Structure choice (measured in-model on MI355X, tools/step_ab.py, one device): the persistent 256x256
    // eight-phase kernel wins once every CU gets >= 4 tiles (QKV: 145 vs 166 us, gated MLP: 307 vs 424 us);
    // at 1.5 tiles per CU (the N = 768 GEMMs) its tile quantisation loses to the 128x128 kernel (d x d 84 vs
    // 69 us, fc2 218 vs 205 us).  DITTO_GEMM=128|256 / ditto_set_option("gemm_tile") force one.
    if (a.fp8) {
        p.tiles_m = (a.M + 255) / 256;
        p.tiles_n = (a.N + 255) / 256;
        return launch_gemm256_fp8(p, epi, s);
    }
    if (epi == EPI_GATED_FP8) return hipErrorInvalidValue;   // fp8 output only from the fp8 GEMM
    // split-K and batched launches exist on the 128x128 structures only
    int forced = p.k_splits > 1 ? 128 : (nbatch > 1 ? 128 : g_gemm_tile);
    // the ReLU epilogue is built for the 128x128 and 256x256 structures only
    if (epi == EPI_BIAS_RELU_BF16 && forced != 128 && forced != 127 && forced != 256)
        forced = (long)((a.M + 255) / 256) * ((a.N + 255) / 256) >= 4 * 256 ? 256 : 128;
    if (forced == 131 && gemm_pp_supports(p, epi)) return launch_gemm_pp(p, epi, s);
    // Ping-pong 128x256 tiles (gemm_pp.hip) by GEMM class.  pp_mask bits: 1 narrow bf16 output (cross q-proj), 2 narrow
    // fp32 in-place residual with K <= 1024 (cross out-proj), 4 narrow fp32 output (final projection), 8 narrow residual
    // with long K (fc2), 16 QKV + RoPE, 32 gated MLP.
    if (forced == 0 && gemm_pp_supports(p, epi) && a.M >= 128) {
        const bool narrow = a.N <= 1024;
        int cls = 0;
        if (epi == EPI_BIAS_BF16 && narrow) cls = 1;
        else if (epi == EPI_BIAS_RES_F32 && narrow) cls = a.K <= 1024 ? 2 : 8;
        else if (epi == EPI_BIAS_F32 && narrow) cls = 4;
        else if (epi == EPI_QKV_ROPE) cls = 16;
        else if (epi == EPI_GATED) cls = 32;
        const long tpp = (long)((a.M + 127) / 128) * ((a.N + 255) / 256);
        // built-in rule (measured IN-MODEL at M = 32768, tools/step_ab.py ^mask: per launch, 256^2 / 192 kernels -> ping-pong):
        // cross out-proj 94.5 -> 89.7 us; q-proj 46.9 -> 52.6, final 91 -> 108, fc2 186 -> 217 (worse; in isolation,
        // with every operand resident in the Infinity Cache, the out-proj showed 96 -> 68 us: bench GEMMs in the model).
        const int mask = g_pp_mask >= 0 ? g_pp_mask : (tpp >= 512 ? 2 : 0);
        if (cls & mask) return launch_gemm_pp(p, epi, s);
    }
    if (forced == 192 && gemm192_supports(epi)) return launch_gemm192(p, epi, s);
    const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    // Wide outputs (N >= 2048: QKV, fc1|gate) take the 256x256 structure from 144 tiles on — re-measured at the end of
    // round 1 (tools/step_ab.py --batch 4 / 8 / 16, per launch): QKV 35.2 -> 31.0 us at B = 4 (144 tiles), 87.1 -> 79.1 at
    // B = 16; gated 51.5 -> 48.3 (B = 4), 99.0 -> 78.3 (B = 8).  Narrow outputs (N = d) keep the 4-round rule: at B = 4
    // the d x d GEMMs take 33 us on it against 18.
    const bool wide256 = a.N >= 2048 && t256 >= 144;
    // Narrow outputs whose 256 x 256 tiles make WHOLE rounds of the 256 CUs (d = 1024 at M = 16384: 64 x 4 tiles, one per CU): no
    // tile quantisation to lose.  Round 5, C5 bf16 in-model (profiles/r05_c5_tile256.txt): fc2 147.5 (256 x 128 ring) -> 116.9 us,
    // the K = 1024 out-projection 57.4 -> 48.4, the final projection 78.5 -> 64.4.  Widths that are multiples of 192 keep the rule
    // below (their 256 x 192 tiles were measured against this kernel in round 1).
    const bool rounds256 = gemm256_whole_rounds(a.M, a.N, a.K);
    if (forced == 256 || (forced == 0 && (t256 >= 4 * 256 || wide256 || rounds256))) {
        p.tiles_m = (a.M + 255) / 256;
        p.tiles_n = (a.N + 255) / 256;
        return launch_gemm256(p, epi, s);
    }
    // 256x128 ring kernel (gemm_p128.hip): N = d GEMMs, exactly 3 tiles per CU at M = 32768, N = 768.  Measured
    // in-model (tools/step_ab.py, one device, after the asm-DMA fix): fc2 (K = 3072) 194 vs 201 us and the final
    // K = 1536 projection 96 vs 104 us against the 128x128 kernel, but the K = 768 d x d GEMMs 70 vs 69 us:
    // automatic only for K >= 1536.
    const long tp128 = (long)((a.M + 255) / 256) * ((a.N + 127) / 128);
    // 256x192 tiles (gemm192.hip) where the output width is a multiple of 192 and whole rounds of them cost no more
    // tile-work than the 256x128 ring's rounds: measured in-model at M = 32768 (tools/step_ab.py): fc2 183.6 vs 189.4 us,
    // the final K = 1536 projection 92.1 vs 97.2 us, the K = 768 d x d GEMMs 69.2 vs 70.1 us (noise): long K only.
    // (K >= 768 since the end of round 1: at M = 16384 the K = 768 d x d GEMMs are exactly one round of 256 such tiles,
    // 32.8 us against 41.2 on the 128x128 kernel; at M = 32768 the two are equal, 69.9 / 71.7.)
    if (forced == 0 && gemm192_supports(epi) && a.N % 192 == 0 && a.K >= 768) {
        const long t192 = (long)((a.M + 255) / 256) * (a.N / 192);
        const long r192 = (t192 + 255) / 256, r128 = (tp128 + 255) / 256;
        if (t192 >= 256 && r192 * 3 <= r128 * 2) return launch_gemm192(p, epi, s);
    }
    if (forced == 129 || (forced == 0 && tp128 >= 2 * 256 && a.K >= 1536)) {
        p.tiles_m = (a.M + 255) / 256;
        p.tiles_n = (a.N + 127) / 128;
        return launch_gemm_p128(p, epi, s);
    }
    p.tiles_m = (a.M + BM - 1) / BM;
    p.tiles_n = (a.N + BN - 1) / BN;
    p.flags = g_gemm_flags & ~(GF_DIAG_NO_STORE | GF_DIAG_NO_EPILOGUE | GF_DIAG_LINEAR_STORE | GF_DIAG_SMALL_OUT);
    p.group_n = pick_group_n(p.tiles_n, p.flags);
    // small grid (one workgroup per CU at most): the deep-prefetch variant; forced with gemm_tile = 127
    const long wgs = (long)p.tiles_m * p.tiles_n * (p.k_splits > 1 ? p.k_splits : 1);
    if (forced == 127 || (forced == 0 && wgs <= 256 && a.K >= 4 * BK)) {
        switch (epi) {
            case EPI_BIAS_BF16: return launch_deep_t<EPI_BIAS_BF16>(p, s);
            case EPI_BIAS_RES_F32: return launch_deep_t<EPI_BIAS_RES_F32>(p, s);
            case EPI_QKV_ROPE: return launch_deep_t<EPI_QKV_ROPE>(p, s);
            case EPI_GATED: return launch_deep_t<EPI_GATED>(p, s);
            case EPI_BIAS_F32: return launch_deep_t<EPI_BIAS_F32>(p, s);
            case EPI_BIAS_RELU_BF16: return launch_deep_t<EPI_BIAS_RELU_BF16>(p, s);
            default: break;
        }
        return hipErrorInvalidValue;
    }
    switch (epi) {
        case EPI_BIAS_BF16: return launch_t<EPI_BIAS_BF16>(p, s);
        case EPI_BIAS_RES_F32: return launch_t<EPI_BIAS_RES_F32>(p, s);
        case EPI_QKV_ROPE: return launch_t<EPI_QKV_ROPE>(p, s);
        case EPI_GATED: return launch_t<EPI_GATED>(p, s);
        case EPI_BIAS_F32: return launch_t<EPI_BIAS_F32>(p, s);
        case EPI_BIAS_RELU_BF16: return launch_t<EPI_BIAS_RELU_BF16>(p, s);
        default: break;
    }
    return hipErrorInvalidValue;
}

}  // namespace ditto
