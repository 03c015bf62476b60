// around_loop.hip — the data-format steps either side of the denoise loop, gfx950.
//
//   vq_argmin        VectorQuantizer.forward (reference src/components/VectorQuantizer.py:22-43): nearest codebook
//                    row by squared L2 distance, called once after the loop (src/model/SpeechGenerator.py:117-118).
//   embedding_gather GPT-2 wte lookup for z_text (src/model/SpeechGenerator.py:101-103).
//   code_embed_mean  EnCodec code -> embedding_head lookup, mean over the codebooks, clip to max_length
//                    (src/components/EnCodec.py:35-37 + src/model/SpeechGenerator.py:97-98).
//   linear_update    x <- a[b]*x + c_eps[b]*eps + c_z[b]*z : the update of a strided-DDPM / DDIM step (the paper's
//                    25-step sampler; the reference only has the stride-1 ancestral update).
//   cfg_combine      eps = eps_u + w*(eps_c - eps_u) (classifier-free guidance, paper App. A).
//
// All HBM-bound or tiny next to the loop (VQ: 2*R*K*D = 103 GFLOP fp32 once per 50 x 355 GFLOP x B); the VQ
// result is an INTEGER index, so its distances are computed in fp32 FMAs (no bf16 rounding that could flip a
// near-tie), with the reference's expression order (||x||^2 - 2 x.c) + ||c||^2 and torch.argmin's first-minimum
// tie rule.
#include "common.h"
#include "kernels.h"

namespace ditto {

// ---------------------------------------------------------------------------------------------------------
// cc[k] = sum_j codebook[k][j]^2   (one wave per code)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x, float* __restrict__ out, int rows,
                                                         int d) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int j = lane; j < d; j += 64) { const float v = x[(size_t)r * d + j]; s = fmaf(v, v, s); }
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
}

// ---------------------------------------------------------------------------------------------------------
// 64 latent rows per workgroup; codes in tiles of 64; k in chunks of 32 through LDS; 4x4 register tile per thread.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                        const float* __restrict__ cc, int64_t* __restrict__ idx, int R,
                                                        int K, int D) {
    __shared__ float xs[32][64 + 4];   // [k][row]
    __shared__ float cs[32][64 + 4];   // [k][code]
    __shared__ float xx[64];
    __shared__ float best_d[64][16];
    __shared__ int best_i[64][16];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int r0 = blockIdx.x * 64;

    // ||x||^2 of this block's rows: 4 threads per row
    {
        const int row = tid >> 2, part = tid & 3;
        const int gr = min(r0 + row, R - 1);
        float s = 0.f;
        for (int j = part; j < D; j += 4) { const float v = x[(size_t)gr * D + j]; s = fmaf(v, v, s); }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (part == 0) xx[row] = s;
    }
    float bd[4];
    int bi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { bd[i] = 3.4e38f; bi[i] = 0; }

    for (int c0 = 0; c0 < K; c0 += 64) {
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int k0 = 0; k0 < D; k0 += 32) {
            __syncthreads();
            // stage 64 rows x 32 k of x and of the codebook, transposed to [k][row]
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * i;           // 0..2047
                const int row = e >> 5, k = e & 31;
                const int gr = min(r0 + row, R - 1), gc = min(c0 + row, K - 1);
                const bool kin = k0 + k < D;
                xs[k][row] = kin ? x[(size_t)gr * D + k0 + k] : 0.f;
                cs[k][row] = kin ? cb[(size_t)gc * D + k0 + k] : 0.f;
            }
            __syncthreads();
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[k][ty * 4]);
                const f32x4 cv = *reinterpret_cast<const f32x4*>(&cs[k][tx * 4]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(xv[i], cv[j], acc[i][j]);
            }
        }
        // distances of this code tile; strict '<' keeps the FIRST minimum (codes ascend within a thread)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int code = c0 + tx * 4 + j;
            if (code < K) {
                const float ccj = cc[code];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float dist = __fadd_rn(__fsub_rn(xx[ty * 4 + i], __fmul_rn(2.0f, acc[i][j])), ccj);
                    if (dist < bd[i]) { bd[i] = dist; bi[i] = code; }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) { best_d[ty * 4 + i][tx] = bd[i]; best_i[ty * 4 + i][tx] = bi[i]; }
    __syncthreads();
    if (tid < 64 && r0 + tid < R) {
        float d0 = best_d[tid][0];
        int i0 = best_i[tid][0];
        for (int t = 1; t < 16; ++t) {
            const float dt = best_d[tid][t];
            const int it = best_i[tid][t];
            if (dt < d0 || (dt == d0 && it < i0)) { d0 = dt; i0 = it; }
        }
        idx[r0 + tid] = i0;
    }
}

hipError_t launch_vq_argmin(const float* x, const float* codebook, float* cc_scratch, int64_t* idx, int R, int K, int D,
                            hipStream_t s) {
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((K + 3) / 4), dim3(256), 0, s, codebook, cc_scratch, K, D);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(vq_argmin_kernel, dim3((R + 63) / 64), dim3(256), 0, s, x, codebook, cc_scratch, idx, R, K, D);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// out[i, :] = table[ids[i], :]          (nn.Embedding lookup; one wave per row, 16-B accesses)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embedding_gather_kernel(const float* __restrict__ table,
                                                               const int64_t* __restrict__ ids, float* __restrict__ out,
                                                               int n, int V, int d) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    long long id = ids[i];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);   // nn.Embedding would raise; clamp instead of faulting
    const f32x4* src = reinterpret_cast<const f32x4*>(table + (size_t)id * d);
    f32x4* dst = reinterpret_cast<f32x4*>(out + (size_t)i * d);
    for (int j = lane; j < d / 4; j += 64) dst[j] = src[j];
}
hipError_t launch_embedding_gather(const float* table, const int64_t* ids, float* out, int n, int V, int d,
                                   hipStream_t s) {
    if (d % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embedding_gather_kernel, dim3((n + 3) / 4), dim3(256), 0, s, table, ids, out, n, V, d);
    return hipGetLastError();
}

// out[b, f, :] = mean_c table[codes[b, c, f], :]   for f < Fout (= min(F, max_length))
__global__ __launch_bounds__(256) void code_embed_mean_kernel(const float* __restrict__ table,
                                                              const int64_t* __restrict__ codes,
                                                              float* __restrict__ out, int B, int C, int F, int Fout,
                                                              int V, int d) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B * Fout) return;
    const int b = i / Fout, f = i % Fout;
    f32x4* dst = reinterpret_cast<f32x4*>(out + (size_t)i * d);
    const float inv = 1.0f / (float)C;
    for (int j = lane; j < d / 4; j += 64) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) {
            long long id = codes[((size_t)b * C + c) * F + f];
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);
            acc += reinterpret_cast<const f32x4*>(table + (size_t)id * d)[j];
        }
        dst[j] = acc * inv;
    }
}
hipError_t launch_code_embed_mean(const float* table, const int64_t* codes, float* out, int B, int C, int F, int Fout,
                                  int V, int d, hipStream_t s) {
    if (d % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(code_embed_mean_kernel, dim3((B * Fout + 3) / 4), dim3(256), 0, s, table, codes, out, B, C, F,
                       Fout, V, d);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// x <- a[b]*x + ce[b]*eps + cz[b]*z   (z may be null: the term is dropped)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_update_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                            const float* __restrict__ z, const float* __restrict__ a,
                                                            const float* __restrict__ ce, const float* __restrict__ cz,
                                                            size_t n4_per_utt) {
    const int b = blockIdx.y;
    const float ab = a[b], eb = ce[b], zb = z ? cz[b] : 0.f;
    const size_t base = (size_t)b * n4_per_utt;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4_per_utt;
         i += (size_t)gridDim.x * blockDim.x) {
        f32x4 xv = reinterpret_cast<f32x4*>(x)[base + i];
        const f32x4 ev = reinterpret_cast<const f32x4*>(eps)[base + i];
        f32x4 zv = {0.f, 0.f, 0.f, 0.f};
        if (z) zv = reinterpret_cast<const f32x4*>(z)[base + i];
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] = fmaf(ab, xv[e], fmaf(eb, ev[e], zb * zv[e]));
        reinterpret_cast<f32x4*>(x)[base + i] = xv;
    }
}
hipError_t launch_linear_update(float* x, const float* eps, const float* z, const float* a, const float* ce,
                                const float* cz, int B, size_t elems_per_utt, hipStream_t s) {
    if (elems_per_utt % 4) return hipErrorInvalidValue;
    const size_t n4 = elems_per_utt / 4;
    size_t gx = (n4 + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(linear_update_kernel, dim3((unsigned)gx, B), dim3(256), 0, s, x, eps, z, a, ce, cz, n4);
    return hipGetLastError();
}

// eps2 = [eps_cond (B utterances) ; eps_uncond (B utterances)]  ->  out = eps_u + w * (eps_c - eps_u)
__global__ __launch_bounds__(256) void cfg_combine_kernel(const float* __restrict__ eps2, float* __restrict__ out,
                                                          float w, size_t n4_half) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4_half; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 c = reinterpret_cast<const f32x4*>(eps2)[i];
        const f32x4 u = reinterpret_cast<const f32x4*>(eps2)[n4_half + i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(w, c[e] - u[e], u[e]);
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}
hipError_t launch_cfg_combine(const float* eps2, float* out, float w, size_t elems_half, hipStream_t s) {
    if (elems_half % 4) return hipErrorInvalidValue;
    const size_t n4 = elems_half / 4;
    size_t g = (n4 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3((unsigned)(g ? g : 1)), dim3(256), 0, s, eps2, out, w, n4);
    return hipGetLastError();
}

}  // namespace ditto
