// Probe: can a full-row GEMM's W operand go STRAIGHT from L2 into registers, bypassing the LDS?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_wdirect.hip -o /tmp/pw && /tmp/pw
// The full-row kernel's weights are packed stage-major, Wp[K/16][768][16]: the 1 KiB of a (stage, 32-column block) piece
// is contiguous, and lane (r32, hh) of a v_mfma_f32_32x32x16_bf16 B... (here: first) operand wants the 16 bytes at
// r32 * 32 + hh * 16 of it — one global_load_dwordx4 per lane covers the piece in whole cache lines.  Each of the 4 waves of
// a workgroup (one per SIMD, as in gemm_fr.hip) streams ITS 6 pieces per stage D stages ahead in a register ring and runs 24
// MFMAs per stage on them.  Printed: time per K = 16 stage, against the 0.37 us the 24 MFMAs need at 2.1 GHz and the 0.55 us
// the LDS-DMA + ds_read loop of gemm_fr.hip takes (1.10 us per K = 32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int STAGE_BYTES = 768 * 16 * 2;   // 24 KiB

template <int D, bool MFMA, bool LOADS>
__global__ __launch_bounds__(256, 1) void probe(const char* W, int nst, float* sink) {
    const int lane = threadIdx.x & 63, wn = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned voff0 = (unsigned)((wn * 6) * 1024 + (lane & 31) * 32 + (lane >> 5) * 16);
    int st = (int)((blockIdx.x & 7) * (nst / 8));   // rotated start per workgroup, as gemm_fr.hip
    f32x4 wr[D][6];                                  // the register ring: D stages x 6 fragments
    f32x16 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a[2];
    a[0] = __builtin_bit_cast(bf16x8, f32x4{1.f, 2.f, 3.f, (float)lane});
    a[1] = __builtin_bit_cast(bf16x8, f32x4{4.f, 5.f, 6.f, (float)lane});
    auto issue = [&](int slot, int nb, int stage) {
        if constexpr (LOADS) {
            const char* base = W + (size_t)stage * STAGE_BYTES + nb * 1024;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wr[slot][nb]) : "v"(voff0), "s"(base) : "memory");
        } else {
            wr[slot][nb] = f32x4{1.f, 2.f, 3.f, 4.f};
        }
    };
    auto next_stage = [&](int s) { return s + 1 == nst ? 0 : s + 1; };
    int is = st;   // issue cursor
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int nb = 0; nb < 6; ++nb) issue(d, nb, is);
        is = next_stage(is);
    }
    const int iters = nst / D;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            // the stage in ring slot d has landed when at most (D - 1) * 6 younger loads are in flight
            if constexpr (LOADS) {
                if constexpr (D == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if constexpr (D == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if constexpr (D == 4) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
            }
#pragma unroll
            for (int nb = 0; nb < 6; ++nb) {
                asm volatile("" : "+v"(wr[d][nb]));
                const bf16x8 w = __builtin_bit_cast(bf16x8, wr[d][nb]);
                if constexpr (MFMA) {
#pragma unroll
                    for (int r = 0; r < 2; ++r) {   // 4 MFMAs per fragment, as 128 rows x 32 columns
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[nb * 2 + 0]) : "v"(w), "v"(a[0]));
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[nb * 2 + 1]) : "v"(w), "v"(a[1]));
                    }
                } else {
                    acc[nb * 2][0] += wr[d][nb][0];
                }
                issue(d, nb, is);   // the same registers take the fragment of D stages on
            }
            is = next_stage(is);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][7];
#pragma unroll
    for (int d = 0; d < D; ++d) s += wr[d][0][0];
    if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int D, bool MFMA, bool LOADS>
float run(const char* W, int nst, float* sink, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<D, MFMA, LOADS>), dim3(256), dim3(256), 0, 0, W, nst, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<D, MFMA, LOADS>), dim3(256), dim3(256), 0, 0, W, nst, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    for (int K : {768, 3072}) {
        const int nst = K / 16;
        char* W; float* sink;
        hipMalloc(&W, (size_t)nst * STAGE_BYTES); hipMalloc(&sink, 4096);
        std::vector<unsigned short> h((size_t)nst * STAGE_BYTES / 2);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff));   // bf16 around 0.01
        hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        printf("K = %d (%d stages of 24 KiB per workgroup, 256 workgroups, 4 waves each)\n", K, nst);
        const int reps = 20;
#define LINE(D, M, L, name) { float us = run<D, M, L>(W, nst, sink, reps); \
        printf("  D=%d %-28s %8.1f us  = %.3f us per stage, %5.1f GB/s per CU\n", D, name, us, us / nst, L ? 24576.0 / (us / nst) * 1e-3 : 0.0); }
        LINE(3, true, false, "MFMAs only")
        LINE(2, false, true, "loads only")
        LINE(3, false, true, "loads only")
        LINE(4, false, true, "loads only")
        LINE(6, false, true, "loads only")
        LINE(2, true, true, "loads + 24 MFMAs per stage")
        LINE(3, true, true, "loads + 24 MFMAs per stage")
        LINE(4, true, true, "loads + 24 MFMAs per stage")
        LINE(6, true, true, "loads + 24 MFMAs per stage")
        hipFree(W); hipFree(sink);
    }
    return 0;
}
