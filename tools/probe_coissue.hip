// probe_coissue.hip — do the matrix pipe and the vector pipe of ONE SIMD run side by side when they are fed by DIFFERENT waves?
// A workgroup is 8 waves (two per SIMD: waves w and w + 4 share SIMD w); waves 0..3 issue back-to-back v_mfma_f32_32x32x16_bf16
// (independent accumulators), waves 4..7 issue a vector stream of one of several instruction mixes.  Modes: 1 = matrix waves only,
// 2 = vector waves only, 3 = both; the kernel stamps s_memtime around each wave's loop.  If the pipes overlap, mode 3's per-wave cycles
// equal modes 1 and 2; if they serialise, they add.  Round 6 finding (profiles/r06_coissue.txt): v_exp_f32 / v_add_f32 streams lose
// ~25 % beside a matrix-only partner, but v_cvt_pk_bf16_f32 waits for the partner's MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_coissue.hip -o build/probe_coissue && build/probe_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// one "pair step" of the vector stream, on two inputs x0, x1; l0 / l1 / pk are the running results
template <int VMIX>
__device__ __forceinline__ void pair_step(float& x0, float& x1, float& l0, float& l1, unsigned& pk) {
    float e0 = x0, e1 = x1;
    unsigned p = 0;
    if constexpr (VMIX == 0) {          // the attention kernel's: 2 exp, 2 add, 1 cvt_pk
        e0 = __builtin_amdgcn_exp2f(x0); e1 = __builtin_amdgcn_exp2f(x1);
        l0 += e0; l1 += e1;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(e0), "v"(e1));
    } else if constexpr (VMIX == 1) {   // adds only (4 per step)
        asm volatile("v_add_f32 %0, 1.5, %0\n\tv_add_f32 %1, 2.5, %1\n\tv_add_f32 %2, %0, %2\n\tv_add_f32 %3, %1, %3" : "+v"(e0), "+v"(e1), "+v"(l0), "+v"(l1));
    } else if constexpr (VMIX == 2) {   // exps only (2 per step)
        e0 = __builtin_amdgcn_exp2f(x0); e1 = __builtin_amdgcn_exp2f(x1);
        asm volatile("" : "+v"(e0), "+v"(e1));
        l0 = e0; l1 = e1;
    } else if constexpr (VMIX == 3) {   // 2 exp, 2 add (no cvt)
        e0 = __builtin_amdgcn_exp2f(x0); e1 = __builtin_amdgcn_exp2f(x1);
        l0 += e0; l1 += e1;
    } else if constexpr (VMIX == 4) {   // cvt_pk only (1 per step)
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(e0), "v"(e1));
    } else if constexpr (VMIX == 5) {   // 2 exp, truncating pack by v_perm_b32, row sum by v_dot2c_f32_bf16 on the pair
        e0 = __builtin_amdgcn_exp2f(x0); e1 = __builtin_amdgcn_exp2f(x1);
        asm volatile("v_perm_b32 %0, %2, %1, %3" : "=v"(p) : "v"(e0), "v"(e1), "v"(0x07060302u));
        asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %1" : "+v"(l0) : "v"(p));
    } else if constexpr (VMIX == 6) {   // v_perm_b32 only
        asm volatile("v_perm_b32 %0, %2, %1, %3" : "=v"(p) : "v"(e0), "v"(e1), "v"(0x07060302u));
    } else if constexpr (VMIX == 7) {   // v_dot2c_f32_bf16 only
        asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %1" : "+v"(l0) : "v"(__builtin_bit_cast(unsigned, e0)));
    } else if constexpr (VMIX == 9) {   // 2 exp + 2 adds that do NOT consume them
        e0 = __builtin_amdgcn_exp2f(x0); e1 = __builtin_amdgcn_exp2f(x1);
        asm volatile("v_add_f32 %0, 1.5, %0\n\tv_add_f32 %1, 2.5, %1" : "+v"(l0), "+v"(l1));
        asm volatile("" : "+v"(e0), "+v"(e1));
        p = __builtin_bit_cast(unsigned, e0) & __builtin_bit_cast(unsigned, e1);
    } else if constexpr (VMIX == 10) {  // 2 exp, cvt_pk, row sum of the ROUNDED pair by v_dot2c_f32_bf16
        e0 = __builtin_amdgcn_exp2f(x0); e1 = __builtin_amdgcn_exp2f(x1);
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(e0), "v"(e1));
        asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %1" : "+v"(l0) : "v"(p));
    } else if constexpr (VMIX == 11) {  // the same with the three-operand v_dot2_f32_bf16
        e0 = __builtin_amdgcn_exp2f(x0); e1 = __builtin_amdgcn_exp2f(x1);
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(e0), "v"(e1));
        asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(l0) : "v"(p), "v"(0x3f803f80u));
    } else if constexpr (VMIX == 12) {  // 2 exp, cvt_pk (no row sum)
        e0 = __builtin_amdgcn_exp2f(x0); e1 = __builtin_amdgcn_exp2f(x1);
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(e0), "v"(e1));
    } else if constexpr (VMIX == 8) {   // 2 exp, 2 add, truncating pack by v_perm_b32
        e0 = __builtin_amdgcn_exp2f(x0); e1 = __builtin_amdgcn_exp2f(x1);
        l0 += e0; l1 += e1;
        asm volatile("v_perm_b32 %0, %2, %1, %3" : "=v"(p) : "v"(e0), "v"(e1), "v"(0x07060302u));
    }
    pk ^= p;
    asm volatile("" : "+v"(x0), "+v"(x1));
}

// block-structured stream: all 32 exponentials of an iteration first (in place), then the 32 adds and 16 converts that consume them
__device__ __forceinline__ void block_step(float (&x)[32], float& l0, float& l1, unsigned& pk, int variant) {
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
    __builtin_amdgcn_sched_barrier(0);
    if (variant == 2) {   // 32 independent plain operations between the exponentials and their consumers
        float z0 = l0, z1 = l1;
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, 0x3f7fff00, %0\n\tv_mul_f32 %1, 0x3f7ffe00, %1" : "+v"(z0), "+v"(z1));
        l0 = z0; l1 = z1;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (variant == 0 || variant == 2) {
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
            l0 += x[i]; l1 += x[i + 1];
            unsigned p;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(x[i]), "v"(x[i + 1]));
            pk ^= p;
        }
    } else {   // adds first (16 x 2), then the converts
#pragma unroll
        for (int i = 0; i < 32; i += 2) { l0 += x[i]; l1 += x[i + 1]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
            unsigned p;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(x[i]), "v"(x[i + 1]));
            pk ^= p;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_mul_f32 %0, 0x3a83126f, %0" : "+v"(x[i]));   // back to small arguments (32 more plain ops)
    __builtin_amdgcn_sched_barrier(0);
}

template <int VMIX, int PRIO>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, const float* in, int mode, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool mat = wid < 4;
    if ((mat && !(mode & 1)) || (!mat && !(mode & 2))) return;
    if constexpr (PRIO == 1) { if (__builtin_amdgcn_readfirstlane(wid) >= 4) __builtin_amdgcn_s_setprio(3); }   // the vector waves at the top priority
    if constexpr (PRIO == 2) { if (__builtin_amdgcn_readfirstlane(wid) < 4) __builtin_amdgcn_s_setprio(3); }    // or the matrix waves
    float seed = in[tid];
    unsigned long long t0, t1;
    if (mat) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed * 0.5f - i); }
        f32x16 c0, c1, c2, c3;
        for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
        out[blockIdx.x * 512 + tid] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        float x[8], l0 = 0.f, l1 = 0.f;
        unsigned pk = 0;
        for (int i = 0; i < 8; ++i) x[i] = seed * (i + 1) * 1e-3f;
        if constexpr (VMIX >= 20) {
            float y[32];
            for (int i = 0; i < 32; ++i) y[i] = seed * (i + 1) * 1e-3f;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
            for (int it = 0; it < iters; ++it) block_step(y, l0, l1, pk, VMIX - 20);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
            out[blockIdx.x * 512 + tid] = l0 + l1 + __builtin_bit_cast(float, pk) + y[3];
            if (lane == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
            return;
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)     // 16 pair steps per iteration (the partner issues 16 MFMAs = 512 cycles per iteration)
#pragma unroll
                for (int i = 0; i < 8; i += 2) pair_step<VMIX>(x[i], x[i + 1], l0, l1, pk);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
        out[blockIdx.x * 512 + tid] = l0 + l1 + __builtin_bit_cast(float, pk);
    }
    if (lane == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
}

// the pair step with its consumers one step late: this step's exponentials, then the adds and the convert of the PREVIOUS step's
// results (no instruction waits for a transcendental issued just before it).  ORDER 0: exps first; 1: consumers first
template <int ORDER>
__device__ __forceinline__ void pair_step_late(float& x0, float& x1, float& l0, float& l1, unsigned& pk, float& ep0, float& ep1) {
    float n0, n1;
    unsigned p;
    if constexpr (ORDER == 0)
        asm volatile("v_exp_f32 %0, %5\n\tv_exp_f32 %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %3, %8\n\tv_cvt_pk_bf16_f32 %4, %7, %8"
                     : "=&v"(n0), "=&v"(n1), "+v"(l0), "+v"(l1), "=&v"(p) : "v"(x0), "v"(x1), "v"(ep0), "v"(ep1));
    else
        asm volatile("v_add_f32 %2, %2, %7\n\tv_add_f32 %3, %3, %8\n\tv_cvt_pk_bf16_f32 %4, %7, %8\n\tv_exp_f32 %0, %5\n\tv_exp_f32 %1, %6"
                     : "=&v"(n0), "=&v"(n1), "+v"(l0), "+v"(l1), "=&v"(p) : "v"(x0), "v"(x1), "v"(ep0), "v"(ep1));
    ep0 = n0; ep1 = n1;
    pk ^= p;
    asm volatile("" : "+v"(x0), "+v"(x1));
}

// the SAME wave issues both: one pair step of the vector mix behind every MFMA
template <int VMIX>
__global__ __launch_bounds__(512, 2) void ksame(float* out, unsigned long long* cyc, const float* in, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float seed = in[tid & 511];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed * 0.5f - i); }
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
    float x[8], l0 = 0.f, l1 = 0.f, ep0 = 1.f, ep1 = 1.f;
    unsigned pk = 0;
    for (int i = 0; i < 8; ++i) x[i] = seed * (i + 1) * 1e-3f;
    auto step = [&](float& a0, float& a1) {
        if constexpr (VMIX == 13) pair_step_late<0>(a0, a1, l0, l1, pk, ep0, ep1);
        else if constexpr (VMIX == 14) pair_step_late<1>(a0, a1, l0, l1, pk, ep0, ep1);
        else pair_step<VMIX>(a0, a1, l0, l1, pk);
    };
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            step(x[0], x[1]);
            __builtin_amdgcn_sched_barrier(0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            step(x[2], x[3]);
            __builtin_amdgcn_sched_barrier(0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            step(x[4], x[5]);
            __builtin_amdgcn_sched_barrier(0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            step(x[6], x[7]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    out[blockIdx.x * 512 + tid] = c0[0] + c1[1] + c2[2] + c3[3] + l0 + l1 + ep0 + ep1 + __builtin_bit_cast(float, pk);
    if (lane == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int VMIX>
void run_same(const char* name, float* out, unsigned long long* cyc, const float* in) {
    const int nblk = 256, iters = 2000;
    std::vector<unsigned long long> hc(nblk * 8);
    double r[2];
    for (int two = 0; two < 2; ++two) {
        CK(hipMemset(cyc, 0, nblk * 8 * 8));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((ksame<VMIX>), dim3(nblk), dim3(two ? 512 : 256), 0, 0, out, cyc, in, iters);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hc.data(), cyc, nblk * 8 * 8, hipMemcpyDeviceToHost));
        std::vector<double> m;
        for (int b = 0; b < nblk; ++b)
            for (int w = 0; w < (two ? 8 : 4); ++w) m.push_back((double)hc[b * 8 + w] / iters / 16);
        std::sort(m.begin(), m.end());
        r[two] = m[m.size() / 2];
    }
    printf("SAME wave, 1 MFMA + 1 pair step of %-40s: %5.1f cycles per pair at one wave per SIMD, %5.1f at two (per wave; %5.1f per SIMD)\n", name, r[0], r[1], r[1] / 2);
}

template <int VMIX>
void run(const char* name, float* out, unsigned long long* cyc, const float* in) {
    const int nblk = 256, iters = 2000;
    std::vector<unsigned long long> hc(nblk * 8);
    double res[12][2];
    for (int mode : {1, 2, 3, 7, 11}) {
        CK(hipMemset(cyc, 0, nblk * 8 * 8));
        for (int rep = 0; rep < 3; ++rep) {
            if (mode == 7) hipLaunchKernelGGL((k<VMIX, 1>), dim3(nblk), dim3(512), 0, 0, out, cyc, in, 3, iters);
            else if (mode == 11) hipLaunchKernelGGL((k<VMIX, 2>), dim3(nblk), dim3(512), 0, 0, out, cyc, in, 3, iters);
            else hipLaunchKernelGGL((k<VMIX, 0>), dim3(nblk), dim3(512), 0, 0, out, cyc, in, mode, iters);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hc.data(), cyc, nblk * 8 * 8, hipMemcpyDeviceToHost));
        std::vector<double> m, v;
        for (int b = 0; b < nblk; ++b)
            for (int w = 0; w < 8; ++w)
                if (hc[b * 8 + w]) (w < 4 ? m : v).push_back((double)hc[b * 8 + w] / iters);
        auto med = [](std::vector<double>& a) { if (a.empty()) return 0.0; std::sort(a.begin(), a.end()); return a[a.size() / 2]; };
        res[mode][0] = med(m); res[mode][1] = med(v);
    }
    printf("%-56s vector alone %6.1f | beside MFMAs %6.1f (x %.2f), MFMA %.1f | vector wave prio 3: %6.1f, MFMA %.1f | matrix wave prio 3: %6.1f, MFMA %.1f\n", name,
           res[2][1], res[3][1], res[3][1] / res[2][1], res[3][0] / 16, res[7][1], res[7][0] / 16, res[11][1], res[11][0] / 16);
}

int main() {
    float *out, *in; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&in, 512 * 4)); CK(hipMalloc(&cyc, 256 * 8 * 8));
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = 0.37f + 0.001f * i;
    CK(hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice));
    run<0>("2 v_exp + 2 v_add + v_cvt_pk_bf16_f32", out, cyc, in);
    run<3>("2 v_exp + 2 v_add", out, cyc, in);
    run<8>("2 v_exp + 2 v_add + v_perm_b32", out, cyc, in);
    run<5>("2 v_exp + v_perm_b32 + v_dot2c_f32_bf16", out, cyc, in);
    run<1>("4 v_add", out, cyc, in);
    run<2>("2 v_exp", out, cyc, in);
    run<4>("v_cvt_pk_bf16_f32", out, cyc, in);
    run<6>("v_perm_b32", out, cyc, in);
    run<7>("v_dot2c_f32_bf16", out, cyc, in);
    run<9>("2 v_exp + 2 independent v_add (+ v_and)", out, cyc, in);
    run<20>("32 v_exp | 16 x (2 add + cvt) | 32 v_mul", out, cyc, in);
    run<21>("32 v_exp | 32 add | 16 cvt | 32 v_mul", out, cyc, in);
    run<22>("32 v_exp | 32 indep v_mul | 16 x (2 add + cvt) | 32 v_mul", out, cyc, in);
    run_same<0>("2 v_exp + 2 v_add + v_cvt_pk_bf16_f32", out, cyc, in);
    run_same<3>("2 v_exp + 2 v_add", out, cyc, in);
    run_same<1>("4 v_add", out, cyc, in);
    run_same<2>("2 v_exp", out, cyc, in);
    run_same<12>("2 v_exp + v_cvt_pk", out, cyc, in);
    run_same<10>("2 v_exp + v_cvt_pk + v_dot2c_f32_bf16", out, cyc, in);
    run_same<11>("2 v_exp + v_cvt_pk + v_dot2_f32_bf16", out, cyc, in);
    run_same<7>("v_dot2c_f32_bf16", out, cyc, in);
    run_same<13>("2 v_exp | add, add, cvt of the PREVIOUS pair", out, cyc, in);
    run_same<14>("add, add, cvt of the PREVIOUS pair | 2 v_exp", out, cyc, in);
    return 0;
}
