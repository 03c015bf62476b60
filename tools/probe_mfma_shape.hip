// probe_mfma_shape.hip — the MFMA-shape A/B of VERDICT r3 item 3 in isolation (MI355X_MICROARCH.md, DVFS give-back item 7;
// cdna_hip_programming.md §5.4 rule 28): the SAME 128 x 128 output tile per wave, one wave per SIMD, random bf16 operands,
// as v_mfma_f32_32x32x16_bf16 (4 x 4 blocks, 16 MFMAs per K = 16) and as v_mfma_f32_16x16x32_bf16 (8 x 8 blocks, 64 MFMAs per
// K = 32); 256 accumulators in AGPRs either way.  Three feeding regimes:
//   regs   every operand stays in registers (the bare loop of the guide's measurement)
//   lds    both operands re-read from LDS by ds_read_b128 every K step (the tiled GEMMs' regime)
//   frd    the full-row kernels' regime: the weight operand arrives by global_load_dwordx4 from an L2-resident image two K steps
//          ahead (register ring, counted vmcnt), the activation operand by ds_read_b128
// Reports wall time, TFLOP/s, shader cycles per K = 32 (s_memtime) and the in-kernel clock (s_memtime / s_memrealtime x 100 MHz;
// median over workgroups).      hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_mfma_shape tools/probe_mfma_shape.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void mfma32(f32x16& c, const f32x4& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16(f32x4& c, const f32x4& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int IMM>
__device__ __forceinline__ void wload(f32x4& dst, unsigned voff, const char* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
template <int VM>
__device__ __forceinline__ void wwait(f32x4& frag) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(frag) : "n"(VM) : "memory"); }

__device__ __forceinline__ unsigned hash32(unsigned h) {
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
// a random bf16 pair in [-2, 2): sign, exponent 126..127, random mantissa
__device__ __forceinline__ unsigned rnd_bf16x2(unsigned seed) {
    const unsigned h = hash32(seed);
    const unsigned lo = ((h & 0x8000u) | (0x3F00u + ((h >> 7) & 0x80u)) | (h & 0x7Fu));
    const unsigned g = h >> 16;
    const unsigned hi = ((g & 0x8000u) | (0x3F00u + ((g >> 7) & 0x80u)) | (g & 0x7Fu));
    return lo | (hi << 16);
}

struct Stamp { unsigned long long cyc, rt; };

// MODE 0 regs, 1 lds, 2 frd.  SHAPE 32 / 16.  iters = K steps of 32.
template <int SHAPE, int MODE>
__global__ __launch_bounds__(256, 1) void probe(const char* __restrict__ wimg, float* __restrict__ sink, Stamp* __restrict__ stamps, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[4][2][8192];       // per wave: A fragments | B fragments of one K = 32 (8 KiB each)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // fill this wave's LDS with random bf16
    for (int i = lane; i < 2 * 8192 / 4; i += 64) reinterpret_cast<unsigned*>(lds[wave])[i] = rnd_bf16x2(blockIdx.x * 7919u + wave * 131u + i);
    __syncthreads();
    constexpr int NB = SHAPE == 32 ? 4 : 8;                              // blocks per side of the 128 x 128 wave tile
    constexpr int KS = SHAPE == 32 ? 2 : 1;                              // MFMA k-steps per K = 32
    using acc_t = typename std::conditional<SHAPE == 32, f32x16, f32x4>::type;
    acc_t acc[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int e = 0; e < (int)(sizeof(acc_t) / 4); ++e) acc[i][j][e] = 0.f;
            asm volatile("" : "+a"(acc[i][j]));
        }
    // per K = 32: NB * KS fragments of 1 KiB per operand (both shapes: 8 KiB per operand)
    constexpr int NF = NB * KS;
    f32x4 wf[2][NF];                                                     // weight-side fragments (MODE 2: a two-step ring)
    bf16x8 af[NF];
    const char* abase = lds[wave][0] + lane * 16;
    const char* bbase = lds[wave][1] + lane * 16;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        wf[0][f] = *reinterpret_cast<const f32x4*>(bbase + f * 1024);
        wf[1][f] = wf[0][f];
        af[f] = *reinterpret_cast<const bf16x8*>(abase + f * 1024);
    }
    // MODE 2: this wave's weight stream = 8 KiB per K step out of a 6 MiB L2-resident image, walked round and round
    const unsigned voff = (unsigned)(wave * 8192 + lane * 16);
    const char* wb = wimg;
    int wstep = 0;
    auto issue_w = [&](f32x4 (&dst)[NF]) {
        wload<0>(dst[0], voff, wb); wload<1024>(dst[1], voff, wb); wload<2048>(dst[2], voff, wb); wload<3072>(dst[3], voff, wb);
        wload<0>(dst[4], voff + 4096u, wb); wload<1024>(dst[5], voff + 4096u, wb); wload<2048>(dst[6], voff + 4096u, wb); wload<3072>(dst[7], voff + 4096u, wb);
        wb += 32768; if (++wstep == 192) { wstep = 0; wb = wimg; }
    };
    if constexpr (MODE == 2) { issue_w(wf[0]); issue_w(wf[1]); }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    Stamp st;
    st.cyc = __builtin_amdgcn_s_memtime(); st.rt = __builtin_amdgcn_s_memrealtime();
    auto kstep = [&](f32x4 (&w)[NF]) {
        if constexpr (MODE >= 1) {
#pragma unroll
            for (int f = 0; f < NF; ++f) af[f] = *reinterpret_cast<const bf16x8*>(abase + f * 1024);
        }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int f = 0; f < NF; ++f) w[f] = *reinterpret_cast<const f32x4*>(bbase + f * 1024);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if constexpr (MODE == 2) {   // fragment j of this k-step landed: 15 younger loads in the steady state (8 of the next ring slot + 7 - ...)
                    if (ks * NB + j == 0) wwait<8>(w[0]);
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if constexpr (SHAPE == 32) mfma32(acc[j][i], w[ks * NB + j], af[ks * NB + i]);
                    else mfma16(acc[j][i], w[ks * NB + j], af[ks * NB + i]);
                }
            }
        if constexpr (MODE == 2) issue_w(w);
    };
    for (int it = 0; it < iters; it += 2) { kstep(wf[0]); kstep(wf[1]); }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    Stamp en;
    en.cyc = __builtin_amdgcn_s_memtime(); en.rt = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && wave == 0) { stamps[blockIdx.x].cyc = en.cyc - st.cyc; stamps[blockIdx.x].rt = en.rt - st.rt; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            asm volatile("" : "+a"(acc[i][j]));
            const acc_t v = acc[i][j];
#pragma unroll
            for (int e = 0; e < (int)(sizeof(acc_t) / 4); ++e) s += v[e];
        }
    if (s == 1.2345f) sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int MODE>
int run(const char* name, const char* wimg, float* sink, Stamp* stamps, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 40; ++w) hipLaunchKernelGGL((probe<SHAPE, MODE>), dim3(256), dim3(256), 0, 0, wimg, sink, stamps, iters);   // ~2 s of load first
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<SHAPE, MODE>), dim3(256), dim3(256), 0, 0, wimg, sink, stamps, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<Stamp> h(256);
    CK(hipMemcpy(h.data(), stamps, 256 * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (auto& s : h) { clk.push_back((double)s.cyc / (double)s.rt * 100.0); cyc.push_back((double)s.cyc / iters); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double flop = 256.0 * 4 * 2.0 * 128 * 128 * 32 * iters;
    printf("%-22s wall %8.3f ms  %7.1f TFLOP/s  cycles per K=32 step %7.1f (MFMA floor 512)  in-kernel clock %6.0f MHz\n", name, ms,
           flop / (ms * 1e-3) / 1e12, cyc[128], clk[128]);
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    char* wimg = nullptr; float* sink = nullptr; Stamp* stamps = nullptr;
    CK(hipMalloc(&wimg, 192 * 32768 + 65536));
    CK(hipMalloc(&sink, 256 * 256 * 4));
    CK(hipMalloc(&stamps, 256 * sizeof(Stamp)));
    std::vector<unsigned> hw((192 * 32768 + 65536) / 4);
    unsigned x = 12345u;
    for (auto& v : hw) {   // random bf16 pairs in [-2, 2)
        x = x * 1664525u + 1013904223u;
        const unsigned a = x >> 8;
        v = ((a & 0x8000u) | 0x3F00u | (a & 0xFFu)) | ((((a >> 9) & 0x8000u) | 0x3F00u | ((a >> 16) & 0xFFu)) << 16);
    }
    CK(hipMemcpy(wimg, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    for (int round = 0; round < 2; ++round) {   // interleaved rounds in ONE process (rule 24)
        if (run<32, 0>("32x32x16  regs", wimg, sink, stamps, iters)) return 1;
        if (run<16, 0>("16x16x32  regs", wimg, sink, stamps, iters)) return 1;
        if (run<32, 1>("32x32x16  lds", wimg, sink, stamps, iters)) return 1;
        if (run<16, 1>("16x16x32  lds", wimg, sink, stamps, iters)) return 1;
        if (run<32, 2>("32x32x16  frd (W L2->reg)", wimg, sink, stamps, iters)) return 1;
        if (run<16, 2>("16x16x32  frd (W L2->reg)", wimg, sink, stamps, iters)) return 1;
    }
    return 0;
}
