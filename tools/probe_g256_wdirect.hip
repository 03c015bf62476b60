// Probe: the persistent 256 x 256 GEMM's main loop with the WEIGHT operand fetched straight from L2 into registers and only
// the activation operand staged through the LDS — is the loop still bound at ~1.6 us per K = 64 tile (gemm256.hip: 64 % MFMA)?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_g256_wdirect.hip -o /tmp/pg && /tmp/pg
// Shape of the gated-MLP GEMM at C2, B = 32: M = 32768, N = 6144, K = 768; 256 workgroups x 12 tiles x 12 K-tiles.
// Layout "1x8": 8 waves side by side in N, a wave owns all 256 rows x 32 columns (128 accumulators, 16x16x32 MFMAs); its W
// fragments (4 KiB per K-tile) come by global_load_dwordx4 from a fragment-major image, two K-tiles ahead in registers (32 VGPRs),
// with NO redundancy between waves; A (32 KiB per K-tile) by LDS-DMA into a ring of 4 K-tiles, ONE barrier per K-tile; 32
// ds_read_b128 per wave and K-tile.  Timing only: the data are random bytes, nothing is stored.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int NA = 4;                       // A ring depth in K-tiles
constexpr int A_KT = 256 * 128;             // 32 KiB: 256 rows x 64 k

__device__ __forceinline__ void dma16(unsigned voff, const char* base, unsigned dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(dst) : "memory");
}
template <int IMM>
__device__ __forceinline__ void wload(f32x4& dst, unsigned voff, const char* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}

// MODE 0: A by DMA + W direct;  1: W direct only (no A DMA);  2: A DMA only (no W loads);  3: neither (MFMA + ds_read floor)
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const char* A, const char* Wf, int lda_bytes, int nkt, int ntiles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    constexpr bool DO_A = MODE == 0 || MODE == 2, DO_W = MODE == 0 || MODE == 1;
    // A DMA: piece = 8 rows x 128 B; wave w moves pieces 4 w .. 4 w + 3 of a K-tile; chunk c of row r sits at c ^ ((r >> 1) & 7)
    const int srow = lane >> 3, scpos = lane & 7;
    unsigned va[4];
    // fragment addressing: lane reads row (lane & 15) of a 16-row block, k-chunks (lane >> 4) and 4 + (lane >> 4)
    const int frow = lane & 15, fq = lane >> 4, fswz = frow >> 1;
    const int a_base = frow * 128;
    const int c0 = ((fq) ^ fswz) << 4, c1 = ((4 + fq) ^ fswz) << 4;
    f32x4 acc[16][2];
#pragma unroll
    for (int m = 0; m < 16; ++m) { acc[m][0] = f32x4{0, 0, 0, 0}; acc[m][1] = f32x4{0, 0, 0, 0}; }
    f32x4 wr[2][4];                                   // W ring: [kt & 1][cb * 2 + ks]
    const unsigned wv = (unsigned)(lane * 16);
    int tile = blockIdx.x;
    // flat K loop over this workgroup's tiles: global K-tile index g = t * nkt + kt
    const int total = ntiles * nkt;
    // tile order as gemm256.hip's: an XCD (blockIdx & 7) owns a contiguous run of the linear tile index, which walks down M inside
    // super-columns of 8 column tiles: at any time the 32 workgroups of an XCD share 4 A panels and 8 W column tiles (L2-resident)
    auto lin = [&](int t) { return (int)(blockIdx.x & 7) * 384 + (int)(blockIdx.x >> 3) + 32 * t; };
    auto tile_rows = [&](int t) { return ((lin(t) % 1024) / 8) * 256; };
    auto tile_cols = [&](int t) { return (lin(t) / 1024) * 8 + lin(t) % 8; };
    auto issue_a1 = [&](int g, int i) {               // piece i (0..3) of this wave, K-tile g -> ring slot g % NA
        if (!DO_A || g >= total) return;
        const int t = g / nkt, kt = g - t * nkt;
        const int m0 = tile_rows(t);
        const int piece = w * 4 + i, row = piece * 8 + srow;
        const int c = scpos ^ ((row >> 1) & 7);
        va[i] = (unsigned)((m0 + row) * lda_bytes + kt * 128 + c * 16);
        dma16(va[i], A, lds_base + (unsigned)((g % NA) * A_KT + piece * 1024));
    };
    auto issue_a = [&](int g) { issue_a1(g, 0); issue_a1(g, 1); issue_a1(g, 2); issue_a1(g, 3); };
    auto wbase = [&](int g) {
        const int t = g / nkt, kt = g - t * nkt;
        // fragment-major image: [col tile 24][kt 12][wave 8][4 frags][1 KiB]
        return Wf + ((size_t)(tile_cols(t) * nkt + kt) * 8 + w) * 4096;
    };
    auto issue_w = [&](int g, f32x4 (&dst)[4]) {      // the wave's four 1-KiB fragments of K-tile g
        if (!DO_W || g >= total) return;
        const char* base = wbase(g);
        wload<0>(dst[0], wv, base); wload<1024>(dst[1], wv, base); wload<2048>(dst[2], wv, base); wload<3072>(dst[3], wv, base);
    };
    (void)tile;
    // prologue in the steady state's issue order (... W(g), A(g+2), W(g+1)): the counted wait below then holds from g = 0 on
    issue_a(0); issue_a(1); issue_w(0, wr[0]); issue_a(2); issue_w(1, wr[1]);
    for (int g = 0; g < total; g += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int gg = g + half;
            // W(gg) has landed when only W(gg+1)'s four loads (+ nothing younger) are in flight; A(gg) pieces are older still
            // W(gg) has landed when only what was issued after it is in flight: A(gg+2) and W(gg+1) = 8 operations (4 without the
            // A stream); A(gg)'s pieces are older still.  A only: A(gg+1), A(gg+2) may be in flight.
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" : "+v"(wr[half][0]), "+v"(wr[half][1]), "+v"(wr[half][2]), "+v"(wr[half][3])::"memory");
            else if (MODE == 1) asm volatile("s_waitcnt vmcnt(4)" : "+v"(wr[half][0]), "+v"(wr[half][1]), "+v"(wr[half][2]), "+v"(wr[half][3])::"memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            const char* at = smem + (gg % NA) * A_KT + a_base;
            bf16x8 af[2][4][2];                        // A fragments of two phases: the next phase's reads under this one's MFMAs
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                af[0][m][0] = *reinterpret_cast<const bf16x8*>(at + m * 16 * 128 + c0);
                af[0][m][1] = *reinterpret_cast<const bf16x8*>(at + m * 16 * 128 + c1);
            }
            const bool wnext = DO_W && gg + 2 < total;
            const char* wb = wnext ? wbase(gg + 2) : Wf;
#pragma unroll
            for (int p = 0; p < 4; ++p) {             // 64 rows per phase: 8 ds_read_b128, 16 MFMAs
                if (p < 3) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        af[(p + 1) & 1][m][0] = *reinterpret_cast<const bf16x8*>(at + ((p + 1) * 4 + m) * 16 * 128 + c0);
                        af[(p + 1) & 1][m][1] = *reinterpret_cast<const bf16x8*>(at + ((p + 1) * 4 + m) * 16 * 128 + c1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
                            acc[p * 4 + m][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, wr[half][cb * 2 + ks]), af[p & 1][m][ks], acc[p * 4 + m][cb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // operand issue BETWEEN the MFMA groups (slot (gg + 3) % 4 = (gg - 1) % 4: everyone is past K-tile gg - 1)
                    if (p < 2) issue_a1(gg + 3, p * 2 + ks);
                    if (p == 3 && wnext) {             // the W registers of k-step ks are free: K-tile gg + 2's fragments go in
                        if (ks == 0) { wload<0>(wr[half][0], wv, wb); wload<2048>(wr[half][2], wv, wb); }
                        else { wload<1024>(wr[half][1], wv, wb); wload<3072>(wr[half][3], wv, wb); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) s += acc[m][0][0] + acc[m][1][3];
    if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int MODE>
float run(const char* A, const char* Wf, int lda, int nkt, int ntiles, float* sink, int reps) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, NA * A_KT);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(512), NA * A_KT, 0, A, Wf, lda, nkt, ntiles, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(512), NA * A_KT, 0, A, Wf, lda, nkt, ntiles, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const int M = 32768, K = 768, nkt = K / 64, ntiles = 12;
    char *A, *Wf; float* sink;
    const size_t a_bytes = (size_t)M * K * 2, w_bytes = (size_t)24 * nkt * 8 * 4096;
    hipMalloc(&A, a_bytes); hipMalloc(&Wf, w_bytes); hipMalloc(&sink, 4096);
    std::vector<unsigned short> h(a_bytes / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)((rand() & 0x807f) | 0x3f00 | ((rand() & 1) << 7));   // +-[0.5, 2)
    hipMemcpy(A, h.data(), a_bytes, hipMemcpyHostToDevice);
    hipMemcpy(Wf, h.data(), w_bytes, hipMemcpyHostToDevice);
    printf("gated-GEMM shape: 256 workgroups x %d tiles x %d K-tiles; gemm256.hip without its epilogue: 229 us = 1.59 us per K-tile\n", ntiles, nkt);
    const char* names[4] = {"A by LDS-DMA + W direct", "W direct only", "A by LDS-DMA only", "neither (MFMA + ds_read)"};
    float us[4];
    us[0] = run<0>(A, Wf, K * 2, nkt, ntiles, sink, 10);
    us[1] = run<1>(A, Wf, K * 2, nkt, ntiles, sink, 10);
    us[2] = run<2>(A, Wf, K * 2, nkt, ntiles, sink, 10);
    us[3] = run<3>(A, Wf, K * 2, nkt, ntiles, sink, 10);
    for (int i = 0; i < 4; ++i)
        printf("  %-28s %8.1f us = %.3f us per K-tile  (%.0f TFLOP/s)\n", names[i], us[i], us[i] / (ntiles * nkt),
               2.0 * M * 6144 * K / us[i] * 1e-6);
    return 0;
}
