// gemm_tn.hip — C[m, n] = sum_k X[k, m] * Y[k, n]  (both operands K-MAJOR: rows are the contraction index) for gfx950.
//
// The weight-gradient GEMMs of the backward pass, dW = dY^T X with K = B*N rows of activations (ditto_train.hip): the
// forward GEMM family wants K-contiguous operand rows, which cost a bf16 transpose pass per operand (6 % of the
// training step).  Here both tiles stay in their natural [k][column] layout in LDS and the MFMA operands are read with
// ds_read_b64_tr_b16 (the hardware transpose that feeds V^T in attention.hip), so nothing is transposed in memory.
//
//   tile      128 (m) x 128 (n) x 64 (k); 256 threads = 4 waves in 2 x 2, each wave 64 x 64 as 2 x 2 accumulators of
//             v_mfma_f32_32x32x16_bf16 (64 fp32 registers); 2 workgroups per CU (2 x 32 KiB LDS).
//   LDS       a tile is 64 k-rows of 256 B; the 16-B chunk c of row r sits at c ^ ((r & 3) << 2): the 4 rows a
//             transposed-read block touches land in 4 different 64-B bank quarters.  Lane-linear image (LDS-DMA),
//             swizzle on the source address; rows past K read a zero row (the caller's 256-B zero buffer).
//   ring      gemm_tn_ring_kernel (default): K-tiles of 32 rows in a 4-stage LDS ring (4 x 16 KiB, still 2 workgroups
//             per CU) behind counted s_waitcnt vmcnt, fragments software-pipelined across the tile boundary (reads of
//             the next 16 k-rows in flight under the 4 MFMAs of the current ones).  [measured] the same speed as the
//             first version (K-tile 64, two buffers, vmcnt(0) + read -> wait -> multiply; gemm_flags bit 2048 selects
//             it): 27.6-28.2 ms of backward either way at C2 B = 16 — with two workgroups per CU neither the
//             global->LDS nor the LDS->register latency is what holds the kernel at ~860 TFLOP/s; the tile's 64 FLOP per
//             byte staged (13 TB/s of L2->LDS traffic at that rate) is the suspect, i.e. a 256-wide tile the next step.
//   split-K   blockIdx.y = split: K-tile range per split, partial tiles to out + split * split_stride (fp32), summed in
//             order by launch_reduce_partials — deterministic.
#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int TM = 128, TN = 128, TK = 64;
constexpr int T_TILE = TK * TM * 2;   // 16 KiB per operand tile
constexpr int T_BUF = 2 * T_TILE;     // X | Y
constexpr int T_LDS = 2 * T_BUF;      // double buffered: 64 KiB
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

struct TnParams {
    const bf16* X; int ldx;   // [K, >= m0 + 128] bf16
    const bf16* Y; int ldy;
    const bf16* zero;         // 256 B of zeros (device)
    float* out; int ldo;
    int Mo, No, K;            // output rows (columns of X), output columns (columns of Y), contraction length
    int tiles_m, tiles_n, k_splits;
    size_t split_stride;
};

DITTO_DEV bf16x8 cat4t(bf16x4 a, bf16x4 b) {
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x % p.tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;
    const int split = blockIdx.y;

    int nkt = (p.K + TK - 1) / TK, kbase = 0;
    if (p.k_splits > 1) {
        const int per = (nkt + p.k_splits - 1) / p.k_splits;
        kbase = split * per;
        nkt = nkt - kbase < per ? nkt - kbase : per;
        if (nkt < 0) nkt = 0;
    }
    float* out = p.out + (size_t)split * p.split_stride;

    // DMA: a tile = 16 pieces of 1 KiB (4 k-rows x 256 B); this wave moves pieces 4*wid .. 4*wid+3 of X and of Y
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int prow = lane >> 4, cpos = lane & 15;          // row inside the piece, 16-B chunk position
    auto stage = [&](int buf, int kt) {
        const int k0 = (kbase + kt) * TK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wid * 4 + i;
            const int r = piece * 4 + prow;                 // k-row inside the tile
            const int c = cpos ^ ((r & 3) << 2);            // source chunk for this LDS position
            const int k = k0 + r;
            // columns past the matrix are clamped (their products land in rows / columns that are never stored)
            int cx = m0 + c * 8; cx = cx + 8 <= p.Mo ? cx : (p.Mo >= 8 ? p.Mo - 8 : 0);
            int cy = n0 + c * 8; cy = cy + 8 <= p.No ? cy : (p.No >= 8 ? p.No - 8 : 0);
            const bf16* sx = k < p.K ? p.X + (size_t)k * p.ldx + cx : p.zero + c * 8;
            const bf16* sy = k < p.K ? p.Y + (size_t)k * p.ldy + cy : p.zero + c * 8;
            glds16(sx, lds_base + (unsigned)(buf * T_BUF + piece * 1024));
            glds16(sy, lds_base + (unsigned)(buf * T_BUF + T_TILE + piece * 1024));
        }
    };

    // transposed-read addressing (attention.hip: V^T fragments): lane -> row 4*hh + tr_q (+8), 8 B at column byte
    // 32*((lane>>4)&1) + 8*tr_p of a 32-column block; the row swizzle is a lane constant (rows differ by multiples of 4)
    const int hh = lane >> 5;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int tr_row0 = 4 * hh + tr_q;
    const int tr_colbyte = 32 * ((lane >> 4) & 1) + 8 * tr_p;
    const int tr_swz = (tr_q & 3) << 6;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][0][i] = 0.f; acc[0][1][i] = 0.f; acc[1][0][i] = 0.f; acc[1][1][i] = 0.f; }

    if (nkt > 0) stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const char* xb = smem + (kt & 1) * T_BUF;
        const char* yb = xb + T_TILE;
        if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {   // 16 k-rows per step
            bf16x8 fx[2], fy[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int colx = ((wr * 2 + blk) * 64 + tr_colbyte) ^ tr_swz;
                const int coly = ((wc * 2 + blk) * 64 + tr_colbyte) ^ tr_swz;
                const char* ax = xb + (16 * s4 + tr_row0) * 256 + colx;
                const char* ay = yb + (16 * s4 + tr_row0) * 256 + coly;
                fx[blk] = cat4t(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ax)),
                                __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ax + 8 * 256)));
                fy[blk] = cat4t(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ay)),
                                __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ay + 8 * 256)));
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx[mb], fy[nb], acc[mb][nb], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // accumulator layout of v_mfma_f32_32x32x16: lane holds column n = lane & 31, rows m = (r & 3) + 8 * (r >> 2) + 4 * hh
    const int ncol = lane & 31;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = n0 + (wc * 2 + nb) * 32 + ncol;
            if (col >= p.No) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wr * 2 + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row < p.Mo) out[(size_t)row * p.ldo + col] = acc[mb][nb][r];
            }
        }
}

constexpr int RK = 32;                     // K-tile of the ring kernel
constexpr int R_TILE = RK * TM * 2;        // 8 KiB per operand tile
constexpr int R_STAGE = 2 * R_TILE;        // X | Y
constexpr int R_NST = 4;
constexpr int R_LDS = R_NST * R_STAGE;     // 64 KiB

__global__ __launch_bounds__(256, 2) void gemm_tn_ring_kernel(TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x % p.tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;
    const int split = blockIdx.y;

    int nkt = (p.K + RK - 1) / RK, kbase = 0;
    if (p.k_splits > 1) {
        const int per = (nkt + p.k_splits - 1) / p.k_splits;
        kbase = split * per;
        nkt = nkt - kbase < per ? nkt - kbase : per;
        if (nkt < 0) nkt = 0;
    }
    float* out = p.out + (size_t)split * p.split_stride;

    // DMA: a tile = 8 pieces of 1 KiB (4 k-rows x 256 B) per operand; this wave moves pieces 2*wid, 2*wid+1 of X and Y:
    // 4 loads per wave per tile (the unit the counted waits below are written in)
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int prow = lane >> 4, cpos = lane & 15;
    int cx, cy;
    {
        const int c = cpos ^ (prow << 2);                  // piece rows are 4*piece + prow: (r & 3) == prow
        cx = m0 + c * 8; cx = cx + 8 <= p.Mo ? cx : (p.Mo >= 8 ? p.Mo - 8 : 0);
        cy = n0 + c * 8; cy = cy + 8 <= p.No ? cy : (p.No >= 8 ? p.No - 8 : 0);
    }
    const int czero = (cpos ^ (prow << 2)) * 8;
    auto stage = [&](int kt) {
        const int st = kt & (R_NST - 1);
        const int k0 = (kbase + kt) * RK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;
            const int k = k0 + piece * 4 + prow;
            const bf16* sx = k < p.K ? p.X + (size_t)k * p.ldx + cx : p.zero + czero;
            const bf16* sy = k < p.K ? p.Y + (size_t)k * p.ldy + cy : p.zero + czero;
            glds16(sx, lds_base + (unsigned)(st * R_STAGE + piece * 1024));
            glds16(sy, lds_base + (unsigned)(st * R_STAGE + R_TILE + piece * 1024));
        }
    };

    const int hh = lane >> 5;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int tr_row0 = 4 * hh + tr_q;
    const int tr_colbyte = 32 * ((lane >> 4) & 1) + 8 * tr_p;
    const int tr_swz = (tr_q & 3) << 6;
    int colx[2], coly[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        colx[blk] = ((wr * 2 + blk) * 64 + tr_colbyte) ^ tr_swz;
        coly[blk] = ((wc * 2 + blk) * 64 + tr_colbyte) ^ tr_swz;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][0][i] = 0.f; acc[0][1][i] = 0.f; acc[1][0][i] = 0.f; acc[1][1][i] = 0.f; }

    // Software-pipelined over the tile boundary: the fragments of the next 16 k-rows are always in flight while the
    // 4 MFMAs of the current ones issue:
    //   iteration kt:  read F1 <- tile kt rows 16..31 | MFMA(F0) | wait tile kt+1, barrier, DMA tile kt+3,
    //                  read F0 <- tile kt+1 rows 0..15 | MFMA(F1)
    auto read_frags = [&](const char* xb, int s2, bf16x8 (&fx)[2], bf16x8 (&fy)[2]) {
        const char* yb = xb + R_TILE;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const char* ax = xb + (16 * s2 + tr_row0) * 256 + colx[blk];
            const char* ay = yb + (16 * s2 + tr_row0) * 256 + coly[blk];
            fx[blk] = cat4t(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ax)),
                            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ax + 8 * 256)));
            fy[blk] = cat4t(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ay)),
                            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ay + 8 * 256)));
        }
    };
    auto mma = [&](const bf16x8 (&fx)[2], const bf16x8 (&fy)[2]) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx[mb], fy[nb], acc[mb][nb], 0, 0, 0);
    };
    // a tile has landed once at most the loads of the tiles issued after it are outstanding (4 per tile per wave)
    auto wait_behind = [&](int later) {
        if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // plain s_barrier (no fence): visibility of the DMA'd tile is what the counted vmcnt waits establish, and the stage
    // a DMA overwrites was last read two MFMA groups earlier — a fence here would drain the F1 reads just issued
    bf16x8 f0x[2], f0y[2], f1x[2], f1y[2];
    if (nkt > 0) {
        for (int t = 0; t < R_NST - 1 && t < nkt; ++t) stage(t);           // tiles 0, 1, 2
        wait_behind(nkt - 1 < R_NST - 2 ? nkt - 1 : R_NST - 2);
        asm volatile("s_barrier" ::: "memory");
        read_frags(smem, 0, f0x, f0y);
        for (int kt = 0; kt + 1 < nkt; ++kt) {                             // the last tile is peeled: one path per body
            read_frags(smem + (kt & (R_NST - 1)) * R_STAGE, 1, f1x, f1y);
            __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs (the scheduler sinks them to save registers)
            mma(f0x, f0y);
            __builtin_amdgcn_sched_barrier(0);
            // issued so far: tiles <= kt + 2 (iteration j issues tile j + 3 after its barrier)
            wait_behind(kt + 2 < nkt ? 1 : 0);
            asm volatile("s_barrier" ::: "memory");
            // into the stage of tile kt - 1: its last reads fed MFMA(F1) of the previous iteration.  (NOT the stage of
            // tile kt: this iteration's F1 reads are issued but may still be in the LDS pipeline.)
            if (kt + R_NST - 1 < nkt) stage(kt + R_NST - 1);
            read_frags(smem + ((kt + 1) & (R_NST - 1)) * R_STAGE, 0, f0x, f0y);
            __builtin_amdgcn_sched_barrier(0);
            mma(f1x, f1y);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_frags(smem + ((nkt - 1) & (R_NST - 1)) * R_STAGE, 1, f1x, f1y);
        __builtin_amdgcn_sched_barrier(0);
        mma(f0x, f0y);
        mma(f1x, f1y);
    }

    const int ncol = lane & 31;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = n0 + (wc * 2 + nb) * 32 + ncol;
            if (col >= p.No) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wr * 2 + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row < p.Mo) out[(size_t)row * p.ldo + col] = acc[mb][nb][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// gemm_tn_wide_kernel: the ring kernel on a 256 (m) x 256 (n) tile — 512 threads = 8 waves in 2 (m) x 4 (n), each wave
// 128 x 64 = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16 (128 registers), ONE workgroup per CU (4-stage ring of
// 32 KiB = 128 KiB).  Why: the 128 x 128 kernel moves 64 KiB of L2 -> LDS traffic per 1 024 MFMA cycles of a CU, three
// times what the chip sustains (~20 B/clk/CU), and per 16 k-rows a wave fetches 2 + 2 fragments (8 ds_read_b64_tr_b16 of
// 512 B) for 4 MFMAs: 128 B/clk for the 8 waves of a CU, half the LDS array (~860 TFLOP/s).  Here: 32 KiB per 1 024
// cycles — the bytes per FLOP of the forward 256 x 256 kernel — and 4 + 2 fragments per 8 MFMAs (96 B/clk).  Same LDS image (k-rows of 512 B, 16-B chunk c of row r at c ^ ((r & 3) << 2),
// which permutes the 64-B quarters inside each 256-B half-row), same software pipeline across the tile boundary, same
// counted waits (4 loads per wave per stage), same split-K contract.
// ---------------------------------------------------------------------------------------------------------
constexpr int WM = 256, WN = 256;
constexpr int W_TILE = RK * WM * 2;        // 16 KiB per operand per stage
constexpr int W_STAGE = 2 * W_TILE;        // X | Y
constexpr int W_LDS = R_NST * W_STAGE;     // 128 KiB

__global__ __launch_bounds__(512, 1) void gemm_tn_wide_kernel(TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;                 // wave rows [128 wr, +128), columns [64 wc, +64)
    // 1-D grid of tiles * splits workgroups; consecutive workgroup ids go round the 8 XCDs, so xcd_remap hands every XCD (and its
    // L2) one contiguous run of logical ids, decoded with the column tile fastest, then the row tile, the split slowest: the
    // workgroups of an XCD share their X column blocks (across tn) and Y column blocks (across tm) over one K range.  With the
    // 2-D grid (tile, split) every workgroup's X stream was the only one of its kind on its XCD: fc1|gate pulled ~1.6 GB through
    // the fabric per launch for 450 MB of operands (4.4 TB/s at 365 us).
    const int lid = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n * p.k_splits);
    const int tile = lid % (p.tiles_m * p.tiles_n), split = lid / (p.tiles_m * p.tiles_n);
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * WM, n0 = tn * WN;

    int nkt = (p.K + RK - 1) / RK, kbase = 0;
    if (p.k_splits > 1) {
        const int per = (nkt + p.k_splits - 1) / p.k_splits;
        kbase = split * per;
        nkt = nkt - kbase < per ? nkt - kbase : per;
        if (nkt < 0) nkt = 0;
    }
    float* out = p.out + (size_t)split * p.split_stride;

    // DMA: a stage = 16 pieces of 1 KiB (2 k-rows x 512 B) per operand; this wave moves pieces 2*wid, 2*wid+1 of X and Y
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int prow = lane >> 5, cpos = lane & 31;          // row inside the piece, 16-B chunk position
    // a lane's (k-row, chunk) offsets inside a stage never change: whole K-tiles take a wave-uniform 64-bit base in scalar
    // registers (one scalar add per K-tile) + these 32-bit offsets; only a tile that crosses K falls back to per-lane addresses
    unsigned vox[2], voy[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wid * 2 + i) * 2 + prow;
        const int c = cpos ^ ((r & 3) << 2);
        int cx = m0 + c * 8; cx = cx + 8 <= p.Mo ? cx : (p.Mo >= 8 ? p.Mo - 8 : 0);
        int cy = n0 + c * 8; cy = cy + 8 <= p.No ? cy : (p.No >= 8 ? p.No - 8 : 0);
        vox[i] = (unsigned)(r * p.ldx + cx) * 2u;
        voy[i] = (unsigned)(r * p.ldy + cy) * 2u;
    }
    auto stage = [&](int kt) {
        const int st = kt & (R_NST - 1);
        const int k0 = (kbase + kt) * RK;
        if (k0 + RK <= p.K) {
            const bf16* bx = p.X + (size_t)k0 * p.ldx;
            const bf16* by = p.Y + (size_t)k0 * p.ldy;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int piece = wid * 2 + i;
                glds16_so(vox[i], bx, lds_base + (unsigned)(st * W_STAGE + piece * 1024));
                glds16_so(voy[i], by, lds_base + (unsigned)(st * W_STAGE + W_TILE + piece * 1024));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;
            const int r = piece * 2 + prow;                 // k-row inside the stage
            const int c = cpos ^ ((r & 3) << 2);            // source chunk for this LDS position
            const int k = k0 + r;
            // columns past the matrix are clamped (their products land in rows / columns that are never stored)
            int cx = m0 + c * 8; cx = cx + 8 <= p.Mo ? cx : (p.Mo >= 8 ? p.Mo - 8 : 0);
            int cy = n0 + c * 8; cy = cy + 8 <= p.No ? cy : (p.No >= 8 ? p.No - 8 : 0);
            const bf16* sx = k < p.K ? p.X + (size_t)k * p.ldx + cx : p.zero + (c & 15) * 8;
            const bf16* sy = k < p.K ? p.Y + (size_t)k * p.ldy + cy : p.zero + (c & 15) * 8;
            glds16(sx, lds_base + (unsigned)(st * W_STAGE + piece * 1024));
            glds16(sy, lds_base + (unsigned)(st * W_STAGE + W_TILE + piece * 1024));
        }
    };

    const int hh = lane >> 5;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int tr_row0 = 4 * hh + tr_q;
    const int tr_colbyte = 32 * ((lane >> 4) & 1) + 8 * tr_p;
    const int tr_swz = (tr_q & 3) << 6;
    int colx[4], coly[2];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) colx[blk] = ((wr * 4 + blk) * 64 + tr_colbyte) ^ tr_swz;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) coly[blk] = ((wc * 2 + blk) * 64 + tr_colbyte) ^ tr_swz;

    f32x16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[mb][0][i] = 0.f; acc[mb][1][i] = 0.f; }

    auto read_frags = [&](const char* xb, int s2, bf16x8 (&fx)[4], bf16x8 (&fy)[2]) {
        const char* yb = xb + W_TILE;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const char* ax = xb + (16 * s2 + tr_row0) * 512 + colx[blk];
            fx[blk] = cat4t(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ax)),
                            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ax + 8 * 512)));
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const char* ay = yb + (16 * s2 + tr_row0) * 512 + coly[blk];
            fy[blk] = cat4t(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ay)),
                            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ay + 8 * 512)));
        }
    };
    auto mma = [&](const bf16x8 (&fx)[4], const bf16x8 (&fy)[2]) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx[mb], fy[nb], acc[mb][nb], 0, 0, 0);
    };
    auto wait_behind = [&](int later) {
        if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    bf16x8 f0x[4], f0y[2], f1x[4], f1y[2];
    if (nkt > 0) {
        for (int t = 0; t < R_NST - 1 && t < nkt; ++t) stage(t);           // tiles 0, 1, 2
        wait_behind(nkt - 1 < R_NST - 2 ? nkt - 1 : R_NST - 2);
        asm volatile("s_barrier" ::: "memory");
        read_frags(smem, 0, f0x, f0y);
        // the next half-step's 12 transposed reads go out BETWEEN the 8 MFMAs of the current one (two per MFMA gap: issued as
        // a batch in front of the MFMAs they delayed the first one and queued behind each other; attention_bwd.hip, round 3)
        auto spread = [&]() {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (k < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int kt = 0; kt + 1 < nkt; ++kt) {                             // the last tile is peeled: one path per body
            mma(f0x, f0y);
            read_frags(smem + (kt & (R_NST - 1)) * W_STAGE, 1, f1x, f1y);
            spread();
            wait_behind(kt + 2 < nkt ? 1 : 0);
            asm volatile("s_barrier" ::: "memory");
            if (kt + R_NST - 1 < nkt) stage(kt + R_NST - 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(f1x, f1y);
            read_frags(smem + ((kt + 1) & (R_NST - 1)) * W_STAGE, 0, f0x, f0y);
            spread();
        }
        read_frags(smem + ((nkt - 1) & (R_NST - 1)) * W_STAGE, 1, f1x, f1y);
        __builtin_amdgcn_sched_barrier(0);
        mma(f0x, f0y);
        mma(f1x, f1y);
    }

    const int ncol = lane & 31;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = n0 + (wc * 2 + nb) * 32 + ncol;
            if (col >= p.No) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wr * 4 + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row < p.Mo) out[(size_t)row * p.ldo + col] = acc[mb][nb][r];
            }
        }
}

}  // namespace

// out fp32 [Mo, No] (ld = ldo) = X[K, Mo]^T Y[K, No]; k_splits > 1: partial sums to out + s * split_stride
// wide: the 256 x 256 tile kernel (one workgroup per CU) instead of the 128 x 128 ring kernel
hipError_t launch_gemm_tn(const void* X, int ldx, const void* Y, int ldy, const void* zero256, float* out, int ldo,
                          int Mo, int No, int K, int k_splits, size_t split_stride, hipStream_t s, bool wide) {
    if (!X || !Y || !zero256 || !out || Mo < 8 || No < 8 || K <= 0 || (ldx | ldy) % 8 || (Mo | No) % 8)
        return hipErrorInvalidValue;
    static DevOnce once_t, once_r, once_w;
    if (hipError_t e = set_max_lds_once(once_t, {reinterpret_cast<const void*>(&gemm_tn_kernel)}, T_LDS)) return e;
    if (hipError_t e = set_max_lds_once(once_r, {reinterpret_cast<const void*>(&gemm_tn_ring_kernel)}, R_LDS)) return e;
    if (hipError_t e = set_max_lds_once(once_w, {reinterpret_cast<const void*>(&gemm_tn_wide_kernel)}, W_LDS)) return e;
    TnParams p;
    p.X = (const bf16*)X; p.ldx = ldx; p.Y = (const bf16*)Y; p.ldy = ldy; p.zero = (const bf16*)zero256;
    p.out = out; p.ldo = ldo; p.Mo = Mo; p.No = No; p.K = K;
    p.tiles_m = (Mo + TM - 1) / TM; p.tiles_n = (No + TN - 1) / TN;
    p.k_splits = k_splits > 1 ? k_splits : 1; p.split_stride = split_stride;
    if (wide) {
        p.tiles_m = (Mo + WM - 1) / WM; p.tiles_n = (No + WN - 1) / WN;
        hipLaunchKernelGGL(gemm_tn_wide_kernel, dim3(p.tiles_m * p.tiles_n * p.k_splits), dim3(512), W_LDS, s, p);
        return hipGetLastError();
    }
    if (g_gemm_flags & 2048)
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(p.tiles_m * p.tiles_n, p.k_splits), dim3(256), T_LDS, s, p);
    else
        hipLaunchKernelGGL(gemm_tn_ring_kernel, dim3(p.tiles_m * p.tiles_n, p.k_splits), dim3(256), R_LDS, s, p);
    return hipGetLastError();
}

}  // namespace ditto
