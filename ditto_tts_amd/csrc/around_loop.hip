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
// result is an INTEGER index, so its distances are computed in exact fp32 (the fp32-input MFMA: no bf16 rounding that
// could flip a near-tie), with the reference's expression order (||x||^2 - 2 x.c) + ||c||^2 and torch.argmin's
// first-minimum tie rule.
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
// -2 x.c on the fp32 MATRIX pipe: v_mfma_f32_32x32x2_f32 is an exact fp32 fused-multiply-add chain (bitwise the
// fmaf chain over k, two k per instruction, at the fp32 vector peak of 64 FLOP/clk/SIMD without occupying the VALU),
// so the distances — and therefore every index, near-ties included — are those of the scalar form
// acc = fmaf(x[k], c[k], acc), k ascending, which is what the VALU kernel of round 1 computed.
//
//   workgroup 128 latent rows; 4 waves = 2 (codes) x 2 (rows); codes in tiles of 128, k in steps of 32 through LDS.
//   operands  the CODEBOOK is the instruction's A operand, the latents its B operand: the 32x32 result block then
//             keeps one latent ROW per lane (column = lane & 31) and 16 codes in its 16 registers (code =
//             8 (i/4) + 4 (lane/32) + i%4, ascending in i), so the running (min distance, first index) of a row is
//             updated in-lane with a strict '<' and no cross-lane traffic until the very end.
//   LDS       k de-interleaved at staging time: even k and odd k of a row are separate arrays, because the instruction
//             takes k = 2p from lanes 0-31 and k = 2p+1 from lanes 32-63: a lane's ds_read_b128 then yields its operand
//             for FOUR consecutive instructions, in ascending-k order.  Row pitch 20 floats: conflict-free b128 reads.
//   distance  (||x||^2 - 2 acc) + ||c||^2 with explicit round-to-nearest ops, as VectorQuantizer.py:34-38.
// ---------------------------------------------------------------------------------------------------------
constexpr int VQ_R = 128, VQ_C = 128, VQ_K = 32, VQ_P = 20;   // rows, codes per tile, k per stage, LDS row pitch (floats)
__global__ __launch_bounds__(256, 2) void vq_argmin_mfma_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                                const float* __restrict__ cc, int64_t* __restrict__ idx,
                                                                int R, int K, int D) {
    __shared__ __attribute__((aligned(16))) float xe[VQ_R * VQ_P], xo[VQ_R * VQ_P];   // latents: even / odd k
    __shared__ __attribute__((aligned(16))) float ce[VQ_C * VQ_P], co[VQ_C * VQ_P];   // codes:   even / odd k
    __shared__ float xx[VQ_R];
    __shared__ float best_d[VQ_R][4];
    __shared__ int best_i[VQ_R][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wc = wid >> 1, wr = wid & 1;           // code half / row half of this wave
    const int l32 = lane & 31, half = lane >> 5;
    const int r0 = blockIdx.x * VQ_R;

    // ||x||^2 of this block's rows: 2 threads per row, sequential halves, fixed order
    {
        const int row = tid >> 1, part = tid & 1;
        const int gr = min(r0 + row, R - 1);
        float s = 0.f;
        for (int j = part; j < D; j += 2) { const float v = x[(size_t)gr * D + j]; s = fmaf(v, v, s); }
        s += __shfl_xor(s, 1, 64);
        if (part == 0) xx[row] = s;
    }
    float bd[2] = {3.4e38f, 3.4e38f};
    int bi[2] = {0, 0};
    const float* xbase = half ? xo : xe;
    const float* cbase = half ? co : ce;

    for (int c0 = 0; c0 < K; c0 += VQ_C) {
        f32x16 acc[2][2];   // [code block][row block]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int k0 = 0; k0 < D; k0 += VQ_K) {
            __syncthreads();
            // stage 128 rows x 32 k of the latents and of the codebook; 4 consecutive k per thread, split even / odd
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = tid + 256 * i;           // 0..1023
                const int row = e >> 3, kq = e & 7;    // k = k0 + 4 kq .. + 3
                const int gr = min(r0 + row, R - 1), gc = min(c0 + row, K - 1);
                f32x4 xv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
                const int k = k0 + 4 * kq;
                if (k + 4 <= D && (D & 3) == 0) {
                    xv = *reinterpret_cast<const f32x4*>(x + (size_t)gr * D + k);
                    cv = *reinterpret_cast<const f32x4*>(cb + (size_t)gc * D + k);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (k + t < D) { xv[t] = x[(size_t)gr * D + k + t]; cv[t] = cb[(size_t)gc * D + k + t]; }
                }
                float* pe = xe + row * VQ_P + 2 * kq;
                float* po = xo + row * VQ_P + 2 * kq;
                pe[0] = xv[0]; pe[1] = xv[2]; po[0] = xv[1]; po[1] = xv[3];
                pe = ce + row * VQ_P + 2 * kq;
                po = co + row * VQ_P + 2 * kq;
                pe[0] = cv[0]; pe[1] = cv[2]; po[0] = cv[1]; po[1] = cv[3];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {              // 4 x (4 k-pairs)
                f32x4 av[2], bv[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    av[i] = *reinterpret_cast<const f32x4*>(cbase + (wc * 64 + i * 32 + l32) * VQ_P + 4 * q);
                    bv[i] = *reinterpret_cast<const f32x4*>(xbase + (wr * 64 + i * 32 + l32) * VQ_P + 4 * q);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][t], bv[j][t], acc[i][j], 0, 0, 0);
            }
        }
        // distances of this code tile: a lane owns ONE row per row block and visits its codes in ascending order
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float xr = xx[wr * 64 + j * 32 + l32];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int code = c0 + wc * 64 + i * 32 + (e >> 2) * 8 + half * 4 + (e & 3);
                    if (code < K) {
                        const float dist = __fadd_rn(__fsub_rn(xr, __fmul_rn(2.0f, acc[i][j][e])), cc[code]);
                        if (dist < bd[j]) { bd[j] = dist; bi[j] = code; }
                    }
                }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        best_d[wr * 64 + j * 32 + l32][wc * 2 + half] = bd[j];
        best_i[wr * 64 + j * 32 + l32][wc * 2 + half] = bi[j];
    }
    __syncthreads();
    if (tid < VQ_R && r0 + tid < R) {
        float d0 = best_d[tid][0];
        int i0 = best_i[tid][0];
        for (int t = 1; t < 4; ++t) {
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
    hipLaunchKernelGGL(vq_argmin_mfma_kernel, dim3((R + VQ_R - 1) / VQ_R), dim3(256), 0, s, x, codebook, cc_scratch, idx, R,
                       K, D);
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
