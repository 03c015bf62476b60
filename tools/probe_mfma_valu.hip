// Probe: do a wave's vector instructions run beside its OWN MFMAs, and does it depend on where the MFMA's accumulator lives?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_valu.hip -o /tmp/pmv && /tmp/pmv
// One workgroup of 4 waves per CU (one wave per SIMD), a loop of NG gaps, each gap = one v_mfma_f32_32x32x16_bf16 + F independent
// vector instructions (asm volatile: the compiler neither reorders nor removes them).  Variants: accumulators in VGPRs ("+v") or
// AGPRs ("+a"); fillers = v_fma_f32 on 8 private registers (chain distance 8), or v_exp_f32; MFMAs on 4 rotating accumulators
// (no back-to-back dependency), or on ONE accumulator (dependent chain).  Printed: cycles per gap (s_memtime), against 32 for the
// MFMA alone and 4 F (8 F for v_exp) for the fillers alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool AGPR>
__device__ __forceinline__ void mfma(f32x16& c, bf16x8 a, bf16x8 b) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// MODE bits: 1 = MFMAs, 2 = fillers.  F fillers per gap; EXP: fillers are v_exp_f32; CHAIN: one accumulator
template <bool AGPR, int MODE, int F, bool EXP, bool CHAIN>
__global__ __launch_bounds__(256, 1) void probe(int iters, float* sink, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a = __builtin_bit_cast(bf16x8, f32x4{1.f, 2.f, 3.f, (float)lane});
    bf16x8 b = __builtin_bit_cast(bf16x8, f32x4{4.f, 5.f, 6.f, (float)lane});
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
    const float k1 = 0.999f, k2 = 0.0001f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if constexpr ((MODE & 1) != 0) mfma<AGPR>(acc[CHAIN ? 0 : g & 3], a, b);
            if constexpr ((MODE & 2) != 0) {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    float& r = x[(g * F + f) & 7];
                    if constexpr (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(k1), "v"(k2));
                }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x16 v = acc[i];
        if constexpr (AGPR) asm volatile("" : "+v"(v));
        s += v[0] + v[7];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <bool AGPR, int MODE, int F, bool EXP, bool CHAIN>
static void run(const char* what, float* sink, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<AGPR, MODE, F, EXP, CHAIN>), dim3(256), dim3(256), 0, 0, iters, sink, cyc);
    hipLaunchKernelGGL((probe<AGPR, MODE, F, EXP, CHAIN>), dim3(256), dim3(256), 0, 0, iters, sink, cyc);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-64s %7.1f ticks per gap\n", what, (double)c / (iters * 8.0));
}

int main() {
    float* sink; unsigned long long* cyc;
    hipMalloc(&sink, 4); hipMalloc(&cyc, 8);
    printf("(s_memtime ticks at 100 MHz: multiply by shader clock / 100 MHz for cycles; compare the rows with each other)\n");
    run<false, 1, 0, false, false>("MFMA alone, 4 accumulators in VGPRs", sink, cyc);
    run<true, 1, 0, false, false>("MFMA alone, 4 accumulators in AGPRs", sink, cyc);
    run<false, 1, 0, false, true>("MFMA alone, ONE accumulator (dependent chain), VGPR", sink, cyc);
    run<false, 2, 4, false, false>("4 v_fma alone", sink, cyc);
    run<false, 2, 6, false, false>("6 v_fma alone", sink, cyc);
    run<false, 2, 8, false, false>("8 v_fma alone", sink, cyc);
    run<false, 2, 4, true, false>("4 v_exp alone", sink, cyc);
    run<false, 3, 4, false, false>("MFMA + 4 v_fma, VGPR accumulators", sink, cyc);
    run<true, 3, 4, false, false>("MFMA + 4 v_fma, AGPR accumulators", sink, cyc);
    run<false, 3, 6, false, false>("MFMA + 6 v_fma, VGPR accumulators", sink, cyc);
    run<true, 3, 6, false, false>("MFMA + 6 v_fma, AGPR accumulators", sink, cyc);
    run<false, 3, 8, false, false>("MFMA + 8 v_fma, VGPR accumulators", sink, cyc);
    run<true, 3, 8, false, false>("MFMA + 8 v_fma, AGPR accumulators", sink, cyc);
    run<false, 3, 4, true, false>("MFMA + 4 v_exp, VGPR accumulators", sink, cyc);
    run<true, 3, 4, true, false>("MFMA + 4 v_exp, AGPR accumulators", sink, cyc);
    run<false, 3, 6, false, true>("MFMA chain + 6 v_fma, VGPR accumulator", sink, cyc);
    return 0;
}
