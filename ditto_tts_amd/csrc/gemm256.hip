// gemm256.hip — 256x256x64 eight-phase bf16 MFMA GEMM for gfx950 (1 workgroup of 8 waves per CU).
//
// Same contract and epilogues as gemm.hip (out = A[M,K] * W[N,K]^T, fused epilogue); this is the structure
// for problems whose 256x256 grid covers the chip.  It follows the recipe of cdna_hip_programming.md §5
// ("The 256^2 8-phase template": LDS-DMA staging that stays in flight across raw s_barriers behind a COUNTED
// vmcnt, two wave groups staggered by one barrier so that on every SIMD one wave is in its MFMA segment while
// its partner is in its LDS/DMA segment, st_16x32-style source-side swizzle), re-derived for this layout:
//
//   waves     8 = 2 (M) x 4 (N); wave (wm, wn) owns C rows [128 wm, +128) x cols [64 wn, +64):
//             8 x 4 accumulators of v_mfma_f32_16x16x32_bf16 = 128 fp32 registers.
//   LDS       128 KiB = 2 K-tile buffers x 4 half-tiles (A_lo, A_hi, B_lo, B_hi; 128 rows x 128 B = 16 KiB each).
//             Row r keeps 16-B chunk c at position c ^ ((r>>1)&7)  (conflict-free ds_read_b128; the swizzle is
//             applied to the DMA's per-lane SOURCE address, the LDS image itself is lane-linear).
//   K-tile    = 4 phases, one 64x32 quadrant of the wave tile (16 MFMAs) each:
//               P1: read B0 (4) + A0 (8 ds_read_b128) -> C00     P2: read B1 (4) -> C01
//               P3: read A1 (8)                        -> C11     P4: no reads     -> C10 (A1, B0 still in registers)
//             so a buffer's B halves are last read in P2 and its A halves in P3.
//   phase     = { ds_reads ; issue this phase's DMA (global_load_lds_dwordx4, 2 per wave per half-tile) ;
//                 [P4/P8: s_waitcnt vmcnt(4)] ; s_barrier ; 16 MFMA ; s_barrier }.
//   DMA order (tile e = even buffer, o = odd buffer), each half restaged >= 2 phases after its last read and
//   issued >= 2 phases before the wait that retires it (a first version with one half-tile per phase and
//   vmcnt(2) left the last half one phase of lead and ran at half the MFMA rate):
//               P8: o.B_lo+o.B_hi  P1: o.A_lo  P2: o.A_hi  P3: -  | P4: vmcnt(4) -> o complete, read in P5..P7
//               P4: e.B_lo+e.B_hi  P5: e.A_lo  P6: e.A_hi  P7: -  | P8: vmcnt(4) -> e complete, read in P1..P3
//             (vmcnt(4) leaves exactly the phase's own two half-tiles in flight; vmcnt(0) once the look-ahead
//             tile does not exist).
//   persistent: gridDim = #CUs; a workgroup walks tiles b, b+grid, ... and issues the NEXT tile's first DMA
//             before the current tile's epilogue, so the cold-start latency hides under the stores.
//   stagger   waves 4-7 (wm = 1) run one barrier behind waves 0-3: every barrier interval has one group in
//             its MFMA segment and the other in its read/DMA segment.  Visibility of DMA data is by
//             {every issuing wave's vmcnt ; a barrier the reader has passed}: the readers' phase comes two
//             barriers after both groups' waits.
//   tails     rows past M / N are clamped on load and masked on store; K-tiles past the end are neither
//             loaded nor multiplied.
#include <type_traits>

#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int HALF_BYTES = 128 * 64 * 2;  // 16 KiB
constexpr int KT_BYTES = 4 * HALF_BYTES;  // 64 KiB: A_lo | A_hi | B_lo | B_hi
constexpr int LDS256 = 2 * KT_BYTES;      // 128 KiB
constexpr int LDS256_ALLOC = LDS256 + 4 * 1024;   // + two (bias row, fp8 weight-scale row) pairs of 256 fp32, double-buffered by tile

template <int V>
struct IC { static constexpr int value = V; };

#define DITTO_BAR() asm volatile("s_barrier" ::: "memory")

#ifdef DITTO_DIAG_G256_STAMP   // tools/build_diag.sh: where does a tile spend its cycles?  (s_memtime stamps; timing build only)
constexpr int G256_STAMP_WAVES = 4096;
__device__ unsigned long long g_g256_stamps[G256_STAMP_WAVES * 8];   // per wave: tile start | main loop | end barrier + prologue | epilogue | tiles | 1
DITTO_DEV unsigned long long g256_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define G256_STAMP(var) const unsigned long long var = g256_now()
#define G256_ACC(i, a, b) st_acc[i] += (b) - (a)
#else
#define G256_STAMP(var)
#define G256_ACC(i, a, b)
#endif

// FLAT (the K loop runs on over the tile switch) is a TEMPLATE parameter on purpose: as a run-time mode its three extra scalars (next tile origin, "a next tile
// exists") cost the default kernel 15 % — the main loop is at the SGPR limit, and the spills (v_writelane / v_readlane) landed
// inside it: same-box A/B gated GEMM 275 -> 318 us, QKV 127 -> 146 us, step 11.9 -> 12.6 ms, found only against the previous
// round's library, because an A/B of the FLAG inside the new build compares two equally slowed kernels.
template <int EPI, bool WIDE, bool FP8, bool FLAT = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmParams p) {
    static_assert(!FLAT || WIDE, "the flat K loop exists for the wide-phase schedule");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    // A K-tile is 128 BYTES of every row: 64 bf16 or 128 fp8 elements; the LDS image, the DMA and the swizzle are
    // byte-identical for both, only the fragment chunks and the MFMA differ.
    constexpr int ESZ = FP8 ? 1 : 2;
    const int nkt = p.K / (128 / ESZ);

    // ---- DMA source addressing: this wave moves pieces 2*wid, 2*wid+1 (1 KiB = 8 rows) of every half-tile ----
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;   // LDS byte address of the dynamic region
    const int srow = lane >> 3;   // row inside a piece
    const int scpos = lane & 7;   // chunk position inside the 128-B row
    int m0 = 0, n0 = 0;           // origin of the tile whose DMA is being issued
    // FLAT K loop (even K-tile counts; default since round 4, gemm_flags bit 16384 = GF_NO_FLAT_K turns it off): K-tiles nkt and nkt + 1 of a tile ARE K-tiles 0 and 1 of the workgroup's
    // next tile, so the DMA slots of the last iteration (empty otherwise) carry the next tile's first six half-tiles, the
    // matrix pipe covers their issue (a 1-KiB LDS-DMA piece holds its wave for 60-180 cycles, twelve of them per wave sat
    // between main loop and epilogue), and K-tile 0 of the next tile is resident before this tile's epilogue starts.
    // Second version (round 4).  The first took the decision "this K-tile index belongs to the next tile" per stage call at run
    // time — a wave-uniform branch and a fresh address computation in front of every DMA of the hand-scheduled loop (17 % slower,
    // DESIGN.md section 8b).  Now the per-lane source addresses of K-tile 0 are EXPLICIT values (fsrc, 16 registers), a stage call
    // is `fsrc + kt * 128` whatever tile it belongs to, and ONE block in the last iteration (after its first two stage calls,
    // which still belong to this tile) re-points them at the next tile minus nkt K-tiles and raises the K-tile limit by two.
    [[maybe_unused]] const char* fsrc[4][2];
    [[maybe_unused]] int klim = nkt;   // FLAT: K-tile indices below this are staged
    [[maybe_unused]] auto set_bases = [&](int om, int on, int kshift) {
#pragma unroll
        for (int half = 0; half < 4; ++half)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wid * 2 + i) * 8 + srow;
                const int c = scpos ^ ((row >> 1) & 7);
                if (half < 2) {
                    int gr = om + half * 128 + row;
                    gr = gr < p.M ? gr : p.M - 1;
                    fsrc[half][i] = (const char*)p.A + (size_t)gr * p.lda * ESZ + c * 16 - (long)kshift * 128;
                } else {
                    int gr = on + (half - 2) * 128 + row;
                    gr = gr < p.w_rows ? gr : p.w_rows - 1;
                    fsrc[half][i] = (const char*)p.W + (size_t)gr * p.ldw * ESZ + c * 16 - (long)kshift * 128;
                }
            }
    };
    unsigned bias_slot = 0;       // which of the two 1 KiB bias rows (after the K-tile buffers) the last prologue filled
    auto stage = [&](int buf, auto HALF, int kt) {
        constexpr int half = decltype(HALF)::value;
        if constexpr (FLAT) {
            if (kt >= klim) return;   // wave-uniform; the waits below account for it
#pragma unroll
            for (int i = 0; i < 2; ++i)
                glds16(fsrc[half][i] + (long)kt * 128, lds_base + (unsigned)(buf * KT_BYTES + half * HALF_BYTES + (wid * 2 + i) * 1024));
            return;
        }
        int om = m0, on = n0;
        if (kt >= nkt) return;    // wave-uniform; the waits below account for it
#ifdef DITTO_DIAG_NODMA
        if (kt > 0) return;       // timing experiment: the main loop without its global->LDS traffic (WRONG results)
#endif
        const int k0b = kt * 128;             // byte offset of the K-tile inside a row
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;
            const int row = piece * 8 + srow;  // 0..127 inside the half
            const int c = scpos ^ ((row >> 1) & 7);
            const char* src;
            if constexpr (half < 2) {
                int gr = om + half * 128 + row;
                gr = gr < p.M ? gr : p.M - 1;
#ifdef DITTO_DIAG_G256_AHOT   // tools/build_diag.sh: every tile reads the A rows of row tile 0 (L2-resident, the same statistics): what does A's path cost?
                gr = half * 128 + row;
#endif
#ifdef DITTO_DIAG_G256_WHOT   // ... and the W rows of column tile 0
                (void)on;
#endif
                src = (const char*)p.A + (size_t)gr * p.lda * ESZ + k0b + c * 16;
            } else {
                int gr = on + (half - 2) * 128 + row;
                gr = gr < p.w_rows ? gr : p.w_rows - 1;
#ifdef DITTO_DIAG_G256_WHOT
                gr = (half - 2) * 128 + row;
#endif
                src = (const char*)p.W + (size_t)gr * p.ldw * ESZ + k0b + c * 16;
            }
#if defined(DITTO_G256_A_POLICY) || defined(DITTO_G256_W_POLICY)   // A/B builds: cache policy of the A / W operand streams
#ifndef DITTO_G256_A_POLICY
#define DITTO_G256_A_POLICY ""
#endif
#ifndef DITTO_G256_W_POLICY
#define DITTO_G256_W_POLICY ""
#endif
            if constexpr (half < 2) DITTO_GLDS16_POLICY(src, lds_base + (unsigned)(buf * KT_BYTES + half * HALF_BYTES + piece * 1024), DITTO_G256_A_POLICY);
            else DITTO_GLDS16_POLICY(src, lds_base + (unsigned)(buf * KT_BYTES + half * HALF_BYTES + piece * 1024), DITTO_G256_W_POLICY);
#else
            glds16(src, lds_base + (unsigned)(buf * KT_BYTES + half * HALF_BYTES + piece * 1024));
#endif
        }
    };
    // tile 0 -> buffer 0 (all four halves), tile 1's B halves -> buffer 1: the state the loop's P1 expects
    auto stage_bias = [&](int tn0) {   // the bias row (fp8: + weight scales) of the tile at column origin tn0 -> the other LDS slot
        bias_slot ^= 1u;
        if (wid == 0 && p.bias) {   // read by the tile's FAST epilogue; columns past N are clamped, unused
            int c = tn0 + lane * 4;
            c = c + 4 <= p.N ? c : 0;
            glds16(p.bias + c, lds_base + (unsigned)(LDS256 + bias_slot * 2048));
        }
        if constexpr (FP8) {
            if (wid == 1 && p.wscale) {   // fp8: the per-output-column weight scales of the tile, same route
                int c = tn0 + lane * 4;
                c = c + 4 <= p.N ? c : 0;
                glds16(p.wscale + c, lds_base + (unsigned)(LDS256 + bias_slot * 2048 + 1024));
            }
        }
    };
    auto prologue = [&](int tile) {
        int tm, tn;
        tile_to_mn(xcd_remap(tile, ntiles), p.tiles_m, p.tiles_n, p.group_n, tm, tn);
        m0 = tm * 256;
        n0 = tn * 256;
        if constexpr (FLAT) set_bases(m0, n0, 0);
        stage_bias(n0);
        stage(0, IC<2>{}, 0);
        stage(0, IC<3>{}, 0);
        stage(0, IC<0>{}, 0);
        stage(0, IC<1>{}, 0);
        stage(1, IC<2>{}, 1);
        stage(1, IC<3>{}, 1);
    };

    // ---- fragment addressing ----
    const int frow = lane & 15, fq = lane >> 4, fswz = frow >> 1;
    const int a_base = wm * HALF_BYTES + frow * 128;                                 // + (ai*64 + m*16)*128
    const int b_base = (2 + (wn >> 1)) * HALF_BYTES + ((wn & 1) * 64 + frow) * 128;  // + (bj*32 + n*16)*128
    // bf16: a lane's two 16-B reads are the k-chunks fq and 4+fq (two K = 32 MFMA steps); fp8: chunks 2fq, 2fq+1
    // (one K = 128 step: 32 consecutive bytes per lane).  Operands of both products use the same chunk map, so the
    // k order inside a step is irrelevant.
    const int coff0 = (((FP8 ? 2 * fq : fq)) ^ fswz) << 4, coff1 = (((FP8 ? 2 * fq + 1 : 4 + fq)) ^ fswz) << 4;

    // Fragment registers.  bf16: 16-B fragments, one per (block, k-step).  fp8: the MFMA operand is 8 consecutive
    // registers, so the fragment is allocated as ONE 8-dword vector and the two ds_read_b128 land in its halves
    // (building it later from two 4-dword values made hipcc copy into fresh aligned tuples: 160-210 spills).
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    struct Frags16 { u32x4 a[8], b0[4], b1[4]; };
    struct Frags8 { i32x8 a[4], b0[2], b1[2]; };
    typename std::conditional<FP8, Frags8, Frags16>::type fr;
    f32x4 acc[8][4];

    auto ld8 = [&](const char* ptr) {   // 32 bytes as one 8-dword operand
        const i32x4 lo = *reinterpret_cast<const i32x4*>(ptr + coff0), hi = *reinterpret_cast<const i32x4*>(ptr + coff1);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto read_A = [&](int buf, auto AI) {
        constexpr int ai = decltype(AI)::value;
        const char* base = smem + buf * KT_BYTES + a_base + ai * 64 * 128;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if constexpr (FP8) {
                fr.a[m] = ld8(base + m * 16 * 128);
            } else {
                fr.a[m * 2 + 0] = *reinterpret_cast<const u32x4*>(base + m * 16 * 128 + coff0);
                fr.a[m * 2 + 1] = *reinterpret_cast<const u32x4*>(base + m * 16 * 128 + coff1);
            }
        }
    };
    auto read_B = [&](int buf, auto BJ) {   // B half bj -> fr.b0 (bj = 0) / fr.b1 (bj = 1)
        constexpr int bj = decltype(BJ)::value;
        const char* base = smem + buf * KT_BYTES + b_base + bj * 32 * 128;
        auto& dst = bj == 0 ? fr.b0 : fr.b1;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if constexpr (FP8) {
                dst[n] = ld8(base + n * 16 * 128);
            } else {
                dst[n * 2 + 0] = *reinterpret_cast<const u32x4*>(base + n * 16 * 128 + coff0);
                dst[n * 2 + 1] = *reinterpret_cast<const u32x4*>(base + n * 16 * 128 + coff1);
            }
        }
    };
    auto mma = [&](auto AI, auto BJ) {
        constexpr int ai = decltype(AI)::value, bj = decltype(BJ)::value;
        auto& bf = bj == 0 ? fr.b0 : fr.b1;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8) {
            // 8 x v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, unit e8m0 scales): 32 cycles each, 4x the K of
            // the bf16 instruction -> twice the FLOPs per cycle at the same bytes per K-tile.
            // ONE asm volatile block per cluster: as builtins hipcc SINKS these (pure) calls across the
            // memory-only asm barriers into later phases, keeping several phases' fragments live (160-260 spills,
            // 2.6x slower than bf16); volatile asm keeps its order relative to the barrier statements.  MFMAs that
            // accumulate into different tiles need no wait states between them; hipcc waits for the ds_reads that
            // feed the operands before the statement (it tracks its own LDS loads).
            const int one = 0x7F7F7F7F;   // e8m0 scale 1.0 in every byte
            f32x4* c = &acc[ai * 4][bj * 2];   // c[m * 4 + n]  (acc rows are 4 tiles wide)
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass must not see amdgcn register constraints (it dropped the stub)
#define DITTO_MS(ci, ai_, bi_) "v_mfma_scale_f32_16x16x128_f8f6f4 %" #ci ", %" #bi_ ", %" #ai_ ", %" #ci ", %14, %14 op_sel_hi:[0,0,0]\n\t"
            asm volatile("s_nop 4\n\t"   // VALU-write -> MFMA-read wait states for operands hipcc may have just moved
                         DITTO_MS(0, 8, 12) DITTO_MS(1, 8, 13) DITTO_MS(2, 9, 12) DITTO_MS(3, 9, 13)
                         DITTO_MS(4, 10, 12) DITTO_MS(5, 10, 13) DITTO_MS(6, 11, 12) DITTO_MS(7, 11, 13)
                         : "+v"(c[0]), "+v"(c[1]), "+v"(c[4]), "+v"(c[5]), "+v"(c[8]), "+v"(c[9]), "+v"(c[12]), "+v"(c[13])
                         : "v"(fr.a[0]), "v"(fr.a[1]), "v"(fr.a[2]), "v"(fr.a[3]), "v"(bf[0]), "v"(bf[1]), "v"(one));
#undef DITTO_MS
#else
            (void)one; (void)c; (void)bf;
#endif
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[ai * 4 + m][bj * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, bf[n * 2 + kk]), __builtin_bit_cast(bf16x8, fr.a[m * 2 + kk]),
                            acc[ai * 4 + m][bj * 2 + n], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // counted wait: `ahead` = the DMA this phase issued itself exists (4 loads stay in flight), else drain
    auto wait_dma = [&](bool ahead) {
        if (ahead) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // VMEM instructions one wave issues in the epilogue of a fully interior tile AFTER the next tile's prologue
    // DMA (stores; the compiler's own waits retire the loads).  vmcnt retires in order, so the tile-start wait may
    // leave these (younger) stores in flight and still guarantee the (older) prologue DMA has landed: the store
    // drain of the previous tile then overlaps the first phases instead of stalling the whole workgroup.
    // (EPI_GATED_PRE: 8 + 16 stores, counted as 16; EPI_GATED_BWD: 32 loads + 32 stores + 8 partial-sum stores, counted as 32 —
    // fewer than the truth only makes the wait stricter)
    constexpr int EPI_STORES = (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_RELU_BF16 || EPI == EPI_GATED_PRE) ? 16 : EPI == EPI_QKV_ROPE ? 16
                               : (EPI == EPI_GATED || EPI == EPI_GATED_FP8) ? 8 : 32;
    bool prev_interior = false;   // previous tile of this workgroup was interior (its store count is exact)

    const int niter = (nkt + 1) / 2;
    int tile = blockIdx.x;
    if (tile < ntiles) prologue(tile);
    if ((p.flags & GF_STAGGER_START) && ntiles >= 8 * p.tile_stride) {   // pays only with many rounds per CU
        // All CUs run identical tiles, so without this they alternate in lockstep between an HBM-idle multiply
        // phase and a chip-wide store burst.  Offsetting 4 groups of workgroups by a quarter tile each spreads the
        // stores of some CUs under the MFMA work of the others.  (blockIdx>>3 walks the CUs of one XCD.)
        const int g = (blockIdx.x >> 3) & 3;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)g * (unsigned)p.stagger_ticks;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }

    constexpr bool flat = FLAT;   // the launcher picks the instantiation (even K-tile count, not GF_NO_FLAT_K)
    bool first_tile = true;
#ifdef DITTO_DIAG_G256_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 1, 0, 0}, lb_acc = 0;
#endif
    for (; tile < ntiles; tile += p.tile_stride) {
        G256_STAMP(t_top);
#ifdef DITTO_DIAG_G256_STAMP
        if (st_acc[6]) lb_acc += t_top - st_acc[6];        // loop back: the previous tile's last stamp -> this top
#endif
        const int cur_m0 = m0, cur_n0 = n0;
        const int next = tile + p.tile_stride;
        // flat: the next tile's origin, computed here (integer divisions) and parked in two VECTOR registers until the last
        // iteration — the main loop is at the scalar-register limit
        [[maybe_unused]] int nm0v = 0, nn0v = 0;
        if constexpr (flat) {
            if (next < ntiles) {
                int tm, tn;
                tile_to_mn(xcd_remap(next, ntiles), p.tiles_m, p.tiles_n, p.group_n, tm, tn);
                asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(nm0v), "=v"(nn0v) : "s"(tm * 256), "s"(tn * 256));
            }
        }
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (!flat || first_tile) {
            // buffer 0 complete (everything older than the newest 4 loads: tile 1's B halves, or the previous
            // tile's epilogue stores, which were issued AFTER this tile's prologue DMA)
            if (prev_interior && nkt > 1 && (p.flags & GF_RELAXED_WAIT)) {
                if constexpr (EPI_STORES == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if constexpr (EPI_STORES == 16) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
            } else {
                wait_dma(nkt > 1);
            }
        }   // flat, later tiles: K-tile 0 landed behind the counted wait of the previous tile's last phase; its B halves of K-tile
            // 1 (the newest four loads before the epilogue's stores) are covered by the first iteration's own counted wait
        first_tile = false;
#ifdef DITTO_DIAG_G256_STAMP
        {   // slot 7: tile top -> this wave AT the tile's first barrier (accumulators zeroed, waits done)
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(acc[m][n]));
            const unsigned long long t_b0 = g256_now();
            st_acc[7] += t_b0 - t_top;
        }
#endif
        // (Tried in round 4, flat loop: the stagger set up once per workgroup and kept over the tile switches — group 0's top barrier
        // paired with group 1's last loop barrier, no balance barrier — so that group 0 need not wait for group 1's epilogue.  The
        // wait only moved to group 0's first loop barrier: gated 273.7 -> 276.3 us, QKV 125.9 -> 132.0 in the model
        // (profiles/r04_step_ab_flat_skew.txt).  The switch costs what the two epilogues of a SIMD's wave pair cost back to back
        // — 5 300 + 4 600 ticks in the stamps — however the barriers around them are arranged.)
        DITTO_BAR();
        if (wm == 1) DITTO_BAR();  // stagger the second wave group by one barrier
        G256_STAMP(t_loop);

        if constexpr (WIDE) {
            // Wide-phase schedule: 2 phases of 32 MFMAs per K-tile (half the barriers).  The phase's ds_reads are
            // RETIRED (lgkmcnt(0)) before its first barrier, so (a) the MFMA segment starts the moment the barrier
            // opens and (b) an LDS half may be restaged ONE phase after its last read (guide: WAR rule, second form):
            //   W1: read e.{B0,B1,A0}; DMA o.A      W2: read e.A1; DMA e'.B; vmcnt -> o complete (read in W3, W4)
            //   W3: read o.{B0,B1,A0}; DMA e'.A     W4: read o.A1; DMA o'.B; vmcnt -> e' complete (read in W1', W2')
            for (int it = 0; it < niter; ++it) {
                const int te = 2 * it, to = te + 1;
                const bool odd_valid = to < nkt;
                read_B(0, IC<0>{});
                read_B(0, IC<1>{});
                read_A(0, IC<0>{});
                stage(1, IC<0>{}, to);
                stage(1, IC<1>{}, to);
                if constexpr (FLAT) {
                    if (it == niter - 1 && next < ntiles) {   // wave-uniform, once per tile: from here on the stage calls feed the NEXT tile
                        m0 = __builtin_amdgcn_readfirstlane(nm0v);
                        n0 = __builtin_amdgcn_readfirstlane(nn0v);
                        set_bases(m0, n0, nkt);
                        klim = nkt + 2;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                DITTO_BAR();
                mma(IC<0>{}, IC<0>{});
                mma(IC<0>{}, IC<1>{});
                DITTO_BAR();

                read_A(0, IC<1>{});
                stage(0, IC<2>{}, te + 2);
                stage(0, IC<3>{}, te + 2);
                wait_dma(te + 2 < (FLAT ? klim : nkt));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                DITTO_BAR();
                mma(IC<1>{}, IC<1>{});
                mma(IC<1>{}, IC<0>{});
                DITTO_BAR();

                read_B(1, IC<0>{});
                read_B(1, IC<1>{});
                read_A(1, IC<0>{});
                stage(0, IC<0>{}, te + 2);
                stage(0, IC<1>{}, te + 2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                DITTO_BAR();
                if (odd_valid) {
                    mma(IC<0>{}, IC<0>{});
                    mma(IC<0>{}, IC<1>{});
                }
                DITTO_BAR();

                read_A(1, IC<1>{});
                stage(1, IC<2>{}, to + 2);
                stage(1, IC<3>{}, to + 2);
                wait_dma(to + 2 < (FLAT ? klim : nkt));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                DITTO_BAR();
                if (odd_valid) {
                    mma(IC<1>{}, IC<1>{});
                    mma(IC<1>{}, IC<0>{});
                }
                DITTO_BAR();
            }
        } else {
            for (int it = 0; it < niter; ++it) {
                const int te = 2 * it, to = te + 1;
                const bool odd_valid = to < nkt;
                // ---------------- K-tile te from buffer 0 ----------------
                read_B(0, IC<0>{});
                read_A(0, IC<0>{});
                stage(1, IC<0>{}, to);                    // P1: o.A_lo   (buffer 1's A halves were last read in P7)
                DITTO_BAR();
                mma(IC<0>{}, IC<0>{});
                DITTO_BAR();

                read_B(0, IC<1>{});
                stage(1, IC<1>{}, to);                    // P2: o.A_hi
                DITTO_BAR();
                mma(IC<0>{}, IC<1>{});
                DITTO_BAR();

                read_A(0, IC<1>{});                       // P3: no DMA
                DITTO_BAR();
                mma(IC<1>{}, IC<1>{});
                DITTO_BAR();

                stage(0, IC<2>{}, te + 2);                // P4: e.B_lo + e.B_hi (buffer 0's B halves were last read in P2)
                stage(0, IC<3>{}, te + 2);
                wait_dma(te + 2 < nkt);                   // buffer 1 (tile to: issued P8, P1, P2) has landed
                DITTO_BAR();
                mma(IC<1>{}, IC<0>{});
                DITTO_BAR();

                // ---------------- K-tile to from buffer 1 ----------------
                read_B(1, IC<0>{});
                read_A(1, IC<0>{});
                stage(0, IC<0>{}, te + 2);                // P5: e.A_lo   (buffer 0's A halves were last read in P3)
                DITTO_BAR();
                if (odd_valid) mma(IC<0>{}, IC<0>{});
                DITTO_BAR();

                read_B(1, IC<1>{});
                stage(0, IC<1>{}, te + 2);                // P6: e.A_hi
                DITTO_BAR();
                if (odd_valid) mma(IC<0>{}, IC<1>{});
                DITTO_BAR();

                read_A(1, IC<1>{});                       // P7: no DMA
                DITTO_BAR();
                if (odd_valid) mma(IC<1>{}, IC<1>{});
                DITTO_BAR();

                stage(1, IC<2>{}, to + 2);                // P8: o.B_lo + o.B_hi of the next odd tile
                stage(1, IC<3>{}, to + 2);
                wait_dma(to + 2 < nkt);                   // buffer 0 (tile te+2: issued P4, P5, P6) has landed
                DITTO_BAR();
                if (odd_valid) mma(IC<1>{}, IC<0>{});
                DITTO_BAR();
            }
        }
        G256_STAMP(t_end);
        if (wm == 0) DITTO_BAR();  // balance the stagger barrier: every LDS read of this tile has retired

        // next tile's first K-tiles start streaming in now, under this tile's epilogue (flat mode: they are in already)
        // FAST epilogue: the tile's 256 bias values came in by LDS-DMA with ITS OWN prologue (one 1 KiB piece, wave 0,
        // the oldest load of the prologue), so the epilogue reads them with ds_read and contains no global load at all.
        // (As compiler-visible global loads issued after the next tile's prologue DMA they made hipcc open every
        // epilogue with s_waitcnt vmcnt(0): each tile's epilogue began by waiting for the next tile's first K-tiles.)
        const bool fast_epi = epilogue_fast_ok<EPI>(p, cur_m0, cur_n0, 256, 256) && !(p.flags & GF_DIAG_NO_EPILOGUE) &&
                              (!FP8 || p.wscale);
        const unsigned cur_bias_slot = bias_slot;
        if constexpr (flat) {
            if (next < ntiles) {   // m0 / n0 are the next tile's since the last iteration; its K-tiles 0 and 1 (B halves) are in
                stage_bias(n0);
#pragma unroll
                for (int half = 0; half < 4; ++half)
#pragma unroll
                    for (int i = 0; i < 2; ++i) fsrc[half][i] += (long)nkt * 128;
                klim = nkt;
            }
        } else if (next < ntiles) {
            prologue(next);
        }
        // ---------------- epilogue ----------------
        G256_STAMP(t_epi);
#ifdef DITTO_DIAG_G256_STAMP
        if (!first_tile || true) { G256_ACC(0, t_top, t_loop); G256_ACC(1, t_loop, t_end); G256_ACC(2, t_end, t_epi); st_acc[4] += 1; st_acc[6] = t_epi; }
#endif
        prev_interior = (cur_m0 + 256 <= p.M) && (cur_n0 + 256 <= p.N) && (!p.out2 || EPI == EPI_GATED_PRE) &&
                        !(p.flags & (GF_DIAG_NO_STORE | GF_DIAG_NO_EPILOGUE));
        if (p.flags & GF_DIAG_NO_EPILOGUE) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) asm volatile("" ::"v"(acc[m][n]));   // keep the accumulators live
            continue;
        }
        if constexpr (EPI == EPI_GATED_BWD) {   // no bias; whole-wave-block epilogue with its own loads (gemm_common.h)
            const int pr = 2 * (cur_m0 >> 8) + wm;
            if (cur_m0 + 256 <= p.M) epilogue_gated_bwd<false>(p, cur_m0 + wm * 128, cur_n0 + wn * 64, acc, fq, frow, pr);
            else epilogue_gated_bwd<true>(p, cur_m0 + wm * 128, cur_n0 + wn * 64, acc, fq, frow, pr);
            continue;
        }
        f32x4 bias4[4];
        if (fast_epi) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
                bias4[n] = *reinterpret_cast<const f32x4*>(smem + LDS256 + cur_bias_slot * 2048 + (wn * 64 + n * 16 + fq * 4) * 4);
        } else {
            load_bias(p, cur_n0 + wn * 64, fq, bias4);
        }
        G256_STAMP(t_bias);
#ifdef DITTO_DIAG_G256_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        (void)t_bias;
#endif
        if constexpr (FP8) {   // per-output-column weight scale of the fp8 quantisation
            f32x4 ws4[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int c = cur_n0 + wn * 64 + n * 16 + fq * 4;
                ws4[n] = f32x4{1.f, 1.f, 1.f, 1.f};
                if (fast_epi)
                    ws4[n] = *reinterpret_cast<const f32x4*>(smem + LDS256 + cur_bias_slot * 2048 + 1024 + (wn * 64 + n * 16 + fq * 4) * 4);
                else if (p.wscale && c < p.N) ws4[n] = *reinterpret_cast<const f32x4*>(p.wscale + c);
            }
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] *= ws4[n];
        }
        if (fast_epi) {
            // interior tile, production flags: straight-line epilogue, the 8 row blocks in one basic block
            const int cb = cur_n0 + wn * 64, r0 = cur_m0 + wm * 128 + frow;
            // (The two waves of a SIMD run their epilogues at the same time and the arbiter serves the OLDER one first — the stamps
            // have wave group 0 through in 5 300 ticks and group 1 in 9 900.  s_setprio swapped between the groups halfway through
            // the row blocks changes nothing: gated GEMM 266.0 vs 266.0 us in the model, profiles/r04_step_ab_epi_prio.txt.)
            if constexpr (EPI == EPI_QKV_ROPE) {
                if (cb >= p.rope_cols) {   // a v head: bias only (wave-uniform)
#pragma unroll
                    for (int m = 0; m < 8; ++m) epilogue_row<EPI_BIAS_BF16, true>(p, r0 + m * 16, cb, acc[m], bias4, fq);
                } else {
#pragma unroll
                    for (int m = 0; m < 8; ++m) epilogue_row<EPI, true>(p, r0 + m * 16, cb, acc[m], bias4, fq);
                }
            } else {
#ifdef DITTO_DIAG_G256_EPI_MFMA   // tools/build_diag.sh ("progressive accumulator release"): what would it cost to run
                                  // the NEXT tile's K-tile 0 under this epilogue?  After each row block's epilogue the 8 MFMAs that block
                                  // would issue (4 column blocks x 2 k-steps, fragments from the K-tile 0 that the flat loop already
                                  // has in LDS buffer 0) go into a spare accumulator block.  TIMING ONLY (the products are discarded):
                                  // if the epilogue segment does not grow, the matrix pipe is free under it and the restructure
                                  // could hide one K-tile of twelve; if it grows by what the MFMAs take, there is nothing to win.
                if constexpr (!FP8) {
                    read_B(0, IC<0>{});
                    read_B(0, IC<1>{});
                    f32x4 dacc[2][4];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int n = 0; n < 4; ++n) dacc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        epilogue_row<EPI, true>(p, r0 + m * 16, cb, acc[m], bias4, fq);
                        const char* abase = smem + a_base + (m >> 2) * 64 * 128 + (m & 3) * 16 * 128;
                        const u32x4 af0 = *reinterpret_cast<const u32x4*>(abase + coff0), af1 = *reinterpret_cast<const u32x4*>(abase + coff1);
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            const auto& bf = n < 2 ? fr.b0 : fr.b1;
                            dacc[m & 1][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[(n & 1) * 2 + 0]),
                                                                                   __builtin_bit_cast(bf16x8, af0), dacc[m & 1][n], 0, 0, 0);
                            dacc[m & 1][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[(n & 1) * 2 + 1]),
                                                                                   __builtin_bit_cast(bf16x8, af1), dacc[m & 1][n], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int n = 0; n < 4; ++n) asm volatile("" ::"v"(dacc[i][n]));
                } else
#endif
                {
#pragma unroll
                for (int m = 0; m < 8; ++m) epilogue_row<EPI, true>(p, r0 + m * 16, cb, acc[m], bias4, fq);
                }
            }
        } else if (cur_m0 + 256 <= p.M && !(p.flags & GF_DIAG_SMALL_OUT)) {
            // interior in M: no per-row guard, so the 8 row blocks are ONE basic block and hipcc interleaves their
            // (independent) epilogue arithmetic instead of running 8 short dependent chains one after the other
#pragma unroll
            for (int m = 0; m < 8; ++m)
                epilogue_row<EPI>(p, cur_m0 + wm * 128 + m * 16 + frow, cur_n0 + wn * 64, acc[m], bias4, fq);
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int row = cur_m0 + wm * 128 + m * 16 + frow;
                if (row < p.M)
                    epilogue_row<EPI>(p, (p.flags & GF_DIAG_SMALL_OUT) ? (row & 255) : row, cur_n0 + wn * 64, acc[m], bias4, fq);
            }
        }
#ifdef DITTO_DIAG_G256_STAMP
        { const unsigned long long t_e1 = g256_now(); st_acc[3] += t_e1 - t_epi; st_acc[6] = t_e1; }
#endif
    }
#ifdef DITTO_DIAG_G256_STAMP
    {
        const unsigned long long t_fin = g256_now();
        (void)t_fin;
        st_acc[6] = lb_acc;
        const int w = blockIdx.x * 8 + wid;
        if (lane == 0 && w < G256_STAMP_WAVES)
            for (int i = 0; i < 8; ++i) g_g256_stamps[w * 8 + i] = st_acc[i];
    }
#endif
}

template <int EPI, bool WIDE, bool FP8, bool FLAT = false>
hipError_t launch256_tw(const GemmParams& p, hipStream_t s) {
    static DevOnce lds_once;
    if (hipError_t e = set_max_lds_once(lds_once, {reinterpret_cast<const void*>(&gemm256_kernel<EPI, WIDE, FP8, FLAT>)}, LDS256_ALLOC)) return e;
    hipLaunchKernelGGL((gemm256_kernel<EPI, WIDE, FP8, FLAT>), dim3(p.tile_stride), dim3(512), LDS256_ALLOC, s, p);
    return hipGetLastError();
}
// The diagnostic builds of tools/build_diag.sh that knock out or re-route the operand traffic (no DMA, A / W rows forced hot, cache
// policies) live in the NON-flat stage path: such a build runs the round-3 tile switch for every launch, whatever gemm_flags says,
// so that it measures the knocked-out kernel and not the production one.
#if defined(DITTO_DIAG_NODMA) || defined(DITTO_DIAG_G256_AHOT) || defined(DITTO_DIAG_G256_WHOT) || defined(DITTO_G256_A_POLICY) || \
    defined(DITTO_G256_W_POLICY)
constexpr bool G256_DIAG_NO_FLAT = true;
#else
constexpr bool G256_DIAG_NO_FLAT = false;
#endif

template <int EPI>
hipError_t launch256_t(const GemmParams& p, hipStream_t s) {
    if (!(p.flags & GF_WIDE_PHASE)) return launch256_tw<EPI, false, false>(p, s);
    const int nkt = p.K / 64;
    if constexpr (G256_DIAG_NO_FLAT) return launch256_tw<EPI, true, false>(p, s);
    // flat K loop wherever the K-tile count is even (C2: gated 252.9 -> 248.7 us, QKV 117.5 -> 115.7, step -0.06 .. -0.10 ms at
    // B = 32, -0.02 / -0.05 ms at B = 8 / 16; training step 49.7 -> 48.9 ms: profiles/r04_step_ab_flat2*.txt, r04_train_flat2_ab.txt)
    if (!(p.flags & GF_NO_FLAT_K) && nkt >= 2 && (nkt & 1) == 0) return launch256_tw<EPI, true, false, true>(p, s);
    return launch256_tw<EPI, true, false>(p, s);
}

}  // namespace

#ifdef DITTO_DIAG_G256_STAMP
extern "C" int ditto_diag_g256_stamps(unsigned long long* out) {   // sums of the per-wave records of the LAST launch (diagnostic build only)
    static unsigned long long host[G256_STAMP_WAVES * 8];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_g256_stamps), sizeof(host)) != hipSuccess) return 1;
    for (int i = 0; i < 16; ++i) out[i] = 0;   // [0, 8): waves 0-3 of every workgroup (wm = 0), [8, 16): waves 4-7 (wm = 1, one barrier behind)
    for (int w = 0; w < G256_STAMP_WAVES; ++w)
        for (int i = 0; i < 8; ++i) out[((w & 7) >> 2) * 8 + i] += host[(size_t)w * 8 + i];
    for (size_t i = 0; i < sizeof(host) / sizeof(host[0]); ++i) host[i] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_g256_stamps), host, sizeof(host)) != hipSuccess;
}
#endif

hipError_t launch_gemm256(const GemmParams& p_in, GemmEpilogue epi, hipStream_t s) {
    // persistent grid: one workgroup per CU (128 KiB LDS each), walking tiles b, b + grid, ...
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
    // quarter of the expected tile time: ~1.5 us per K-tile + ~8 us fixed, in 10 ns ticks
    p.stagger_ticks = (int)((p.K / 64 * 1.5 + 8.0) * 100.0 / 4.0);
    // (EPI_GATED_PRE / EPI_GATED_BWD with the four CU groups of an XCD started 4 - 20 us apart: 51.7 - 52.5 ms per training
    // step against 51.8 together, profiles/r04_train_stagger_ab.txt — their HBM-heavy epilogues do not want de-phasing either)
    switch (epi) {
        case EPI_BIAS_BF16: return launch256_t<EPI_BIAS_BF16>(p, s);
        case EPI_BIAS_RES_F32: return launch256_t<EPI_BIAS_RES_F32>(p, s);
        case EPI_QKV_ROPE: return launch256_t<EPI_QKV_ROPE>(p, s);
        case EPI_GATED: return launch256_t<EPI_GATED>(p, s);
        case EPI_GATED_PRE: return launch256_t<EPI_GATED_PRE>(p, s);
        case EPI_GATED_BWD: return launch256_t<EPI_GATED_BWD>(p, s);
        case EPI_BIAS_F32: return launch256_t<EPI_BIAS_F32>(p, s);
        case EPI_BIAS_RELU_BF16: return launch256_t<EPI_BIAS_RELU_BF16>(p, s);
        default: break;
    }
    return hipErrorInvalidValue;
}

// fp8 e4m3 operands (always the wide-phase schedule)
hipError_t launch_gemm256_fp8(const GemmParams& p_in, GemmEpilogue epi, hipStream_t s) {
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
        n_cu = 256;
    GemmParams p = p_in;
    const int ntiles = p.tiles_m * p.tiles_n;
    p.tile_stride = ntiles < n_cu ? ntiles : n_cu;
    p.flags = g_gemm_flags;
    p.group_n = pick_group_n(p.tiles_n, p.flags);
    p.stagger_ticks = (int)((p.K / 128 * 1.5 + 8.0) * 100.0 / 4.0);
    const int nkt = p.K / 128;
    if (!G256_DIAG_NO_FLAT && !(p.flags & GF_NO_FLAT_K) && nkt >= 2 && (nkt & 1) == 0) {   // flat K loop, as the bf16 launcher
        switch (epi) {
            case EPI_BIAS_BF16: return launch256_tw<EPI_BIAS_BF16, true, true, true>(p, s);
            case EPI_BIAS_RES_F32: return launch256_tw<EPI_BIAS_RES_F32, true, true, true>(p, s);
            case EPI_QKV_ROPE: return launch256_tw<EPI_QKV_ROPE, true, true, true>(p, s);
            case EPI_BIAS_F32: return launch256_tw<EPI_BIAS_F32, true, true, true>(p, s);
            case EPI_GATED_FP8: return launch256_tw<EPI_GATED_FP8, true, true, true>(p, s);
            default: break;
        }
        return hipErrorInvalidValue;
    }
    switch (epi) {
        case EPI_BIAS_BF16: return launch256_tw<EPI_BIAS_BF16, true, true>(p, s);
        case EPI_BIAS_RES_F32: return launch256_tw<EPI_BIAS_RES_F32, true, true>(p, s);
        case EPI_QKV_ROPE: return launch256_tw<EPI_QKV_ROPE, true, true>(p, s);
        case EPI_BIAS_F32: return launch256_tw<EPI_BIAS_F32, true, true>(p, s);
        case EPI_GATED_FP8: return launch256_tw<EPI_GATED_FP8, true, true>(p, s);
        default: break;
    }
    return hipErrorInvalidValue;
}

}  // namespace ditto
