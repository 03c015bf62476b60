// rowwise.hip — HBM-bound kernels of the DiT denoise path (gfx950).
//
// Roofline: every kernel here moves each byte once (algorithmic bytes == HBM traffic) and is judged
// against HBM (~6.3 TB/s achievable of 8 TB/s).  Rules applied (cdna_hip_programming.md G2/G11/G13):
// 16-byte per-lane accesses, one wave64 per row with shuffle reductions (no LDS round trip needed at
// d <= 2048: the row lives in registers between the statistics and the normalise pass), grids of
// >= 4 rows per 256-thread block so that >= 8k blocks cover M = 32k rows.
#include "common.h"
#include "kernels.h"

namespace ditto {

// ------------------------------------------------------------------------------------------------
// LayerNorm / GlobalAdaLN.  Reference: nn.LayerNorm(d) (biased variance, eps 1e-5)
//   src/components/DiT.py:84,89,94 (affine), :23,:38-39 (no affine + scale/shift).
// One wave per row; lane i owns float4 chunks i, i+64, ... (CH of them) => coalesced 1 KiB per
// wave-instruction.  Two-pass statistics on the register copy (mean, then sum (x-mean)^2).
// MODE 0: affine (gamma/beta may be null) -> bf16.   MODE 1: AdaLN modulation -> fp32 + bf16(raw x).
// ------------------------------------------------------------------------------------------------
// MODE 2: as MODE 0 with fp8 e4m3 output (A operand of the fp8 QKV / fc1|gate GEMMs, config C5)
template <int CH, int MODE>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, const float* __restrict__ ttab,
                                                 const float* __restrict__ tmod, const int64_t* __restrict__ t,
                                                 int steps, int rows_per_batch, bf16* __restrict__ out_bf16, int ldo,
                                                 float* __restrict__ out_f32, int M, int d, int x_bf16, int h_bf16,
                                                 const float* __restrict__ g1, const float* __restrict__ be1,
                                                 bf16* __restrict__ u1) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = d >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * d);
    const u32x2* xb = reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16*>(x) + (size_t)row * d);   // x_bf16: bf16 rows
    f32x4 v[CH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = lane + 64 * c;
        if (i < nv) {
            if (x_bf16) {
                const u32x2 w = xb[i];
                v[c] = f32x4{__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xFFFF0000u),
                             __builtin_bit_cast(float, w[1] << 16), __builtin_bit_cast(float, w[1] & 0xFFFF0000u)};
            } else {
                v[c] = xr[i];
            }
            s += ln_sum4(v[c]);
        } else {
            v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = lane + 64 * c;
        if (i < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) q = ln_sq_acc(q, v[c][e], mean);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);

    if constexpr (MODE == 2) {
        unsigned char* orow = reinterpret_cast<unsigned char*>(out_bf16) + (size_t)row * ldo;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = lane + 64 * c;
            if (i < nv) {
                f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
                if (gamma) {
                    g = reinterpret_cast<const f32x4*>(gamma)[i];
                    b = reinterpret_cast<const f32x4*>(beta)[i];
                }
                *reinterpret_cast<unsigned*>(orow + 4 * i) =
                    pack_fp8x4(ln_norm(v[c][0], mean, rstd, g[0], b[0]), ln_norm(v[c][1], mean, rstd, g[1], b[1]),
                               ln_norm(v[c][2], mean, rstd, g[2], b[2]), ln_norm(v[c][3], mean, rstd, g[3], b[3]));
            }
        }
    } else if constexpr (MODE == 0) {
        bf16* orow = out_bf16 + (size_t)row * ldo;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = lane + 64 * c;
            if (i < nv) {
                f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
                if (gamma) {
                    g = reinterpret_cast<const f32x4*>(gamma)[i];
                    b = reinterpret_cast<const f32x4*>(beta)[i];
                }
                u32x2 o;
                o[0] = pack_bf16x2(ln_norm(v[c][0], mean, rstd, g[0], b[0]), ln_norm(v[c][1], mean, rstd, g[1], b[1]));
                o[1] = pack_bf16x2(ln_norm(v[c][2], mean, rstd, g[2], b[2]), ln_norm(v[c][3], mean, rstd, g[3], b[3]));
                *reinterpret_cast<u32x2*>(orow + 4 * i) = o;
            }
        }
    } else {
        const int b_idx = row / rows_per_batch;
        long long ts = t ? t[b_idx] : b_idx;               // t == nullptr: table row = batch index
        ts = ts < 0 ? 0 : (ts >= steps ? steps - 1 : ts);  // nn.Embedding would raise; clamp instead of faulting
        const f32x4* tt = reinterpret_cast<const f32x4*>(ttab + (size_t)ts * 2 * d);
        const f32x4* tm = reinterpret_cast<const f32x4*>(tmod + (size_t)b_idx * 2 * d);
        bf16* orow = out_bf16 + (size_t)row * ldo;
        f32x4* hrow = reinterpret_cast<f32x4*>(out_f32 + (size_t)row * d);
        u32x2* hbrow = reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(out_f32) + (size_t)row * d);   // h_bf16: the bf16 stream
        float s1 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = lane + 64 * c;
            if (i < nv) {
                const f32x4 sc_t = tt[i], sc_x = tm[i], sh_t = tt[nv + i], sh_x = tm[nv + i];
                f32x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float scale = 1.f + sc_t[e] + sc_x[e];          // DiT.py:34
                    const float shift = sh_t[e] + sh_x[e];                // DiT.py:35
                    h[e] = (v[c][e] - mean) * rstd * scale + shift;       // DiT.py:38-39
                }
                if (h_bf16) {
                    u32x2 o;
                    o[0] = pack_bf16x2(h[0], h[1]);
                    o[1] = pack_bf16x2(h[2], h[3]);
                    hbrow[i] = o;
                } else {
                    hrow[i] = h;
                }
                if (out_bf16) {
                    u32x2 o;
                    o[0] = pack_bf16x2(v[c][0], v[c][1]);
                    o[1] = pack_bf16x2(v[c][2], v[c][3]);
                    *reinterpret_cast<u32x2*>(orow + 4 * i) = o;
                }
                v[c] = h;                                                 // kept for the fused norm1 below
                s1 += ln_sum4(h);
            }
        }
        // u1 set: block 0's norm1 (src/components/DiT.py:105) of the row just produced, from the fp32 values in registers
        // (the separate LayerNorm launch in front of block 0 and its 100 MB read disappear)
        if (u1) {
            const float mean1 = wave_sum(s1) / (float)d;
            float q1 = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int i = lane + 64 * c;
                if (i < nv) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) q1 = ln_sq_acc(q1, v[c][e], mean1);
                }
            }
            const float rstd1 = rsqrtf(wave_sum(q1) / (float)d + 1e-5f);
            bf16* urow = u1 + (size_t)row * d;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int i = lane + 64 * c;
                if (i < nv) {
                    const f32x4 g = reinterpret_cast<const f32x4*>(g1)[i], b = reinterpret_cast<const f32x4*>(be1)[i];
                    u32x2 o;
                    o[0] = pack_bf16x2(ln_norm(v[c][0], mean1, rstd1, g[0], b[0]), ln_norm(v[c][1], mean1, rstd1, g[1], b[1]));
                    o[1] = pack_bf16x2(ln_norm(v[c][2], mean1, rstd1, g[2], b[2]), ln_norm(v[c][3], mean1, rstd1, g[3], b[3]));
                    *reinterpret_cast<u32x2*>(urow + 4 * i) = o;
                }
            }
        }
    }
}

template <int MODE>
static hipError_t ln_dispatch(const float* x, const float* gamma, const float* beta, const float* ttab,
                              const float* tmod, const int64_t* t, int steps, int rpb, bf16* ob, int ldo, float* of,
                              int M, int d, hipStream_t s, int x_bf16 = 0, int h_bf16 = 0, const float* g1 = nullptr,
                              const float* be1 = nullptr, bf16* u1 = nullptr) {
    const int ch = (d / 4 + 63) / 64;
    dim3 grid((M + 3) / 4), block(256);
#define LN_CASE(C)                                                                                             \
    case C:                                                                                                    \
        hipLaunchKernelGGL((ln_kernel<C, MODE>), grid, block, 0, s, x, gamma, beta, ttab, tmod, t, steps, rpb, \
                           ob, ldo, of, M, d, x_bf16, h_bf16, g1, be1, u1);                                    \
        break;
    switch (ch) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
        default: return hipErrorInvalidValue;  // d > 2048 (checked with a message in the API layer)
    }
#undef LN_CASE
    return hipGetLastError();
}

hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta, void* out_bf16, int ldo, int M,
                            int d, hipStream_t s) {
    return ln_dispatch<0>(x, gamma, beta, nullptr, nullptr, nullptr, 0, 1, (bf16*)out_bf16, ldo, nullptr, M, d, s);
}

hipError_t launch_layernorm_fp8(const float* x, const float* gamma, const float* beta, void* out_fp8, int ldo, int M,
                                int d, hipStream_t s) {
    return ln_dispatch<2>(x, gamma, beta, nullptr, nullptr, nullptr, 0, 1, (bf16*)out_fp8, ldo, nullptr, M, d, s);
}

hipError_t launch_adaln(const float* x, const float* ttab, const float* tmod, const int64_t* t, int steps,
                        float* h_out, void* raw_bf16, int ldraw, int B, int N, int d, hipStream_t s, bool h_bf16,
                        const float* g1, const float* be1, void* u1_bf16) {
    return ln_dispatch<1>(x, nullptr, nullptr, ttab, tmod, t, steps, N, (bf16*)raw_bf16, ldraw, h_out, B * N, d, s, 0,
                          h_bf16 ? 1 : 0, g1, be1, (bf16*)u1_bf16);
}
// LayerNorm of a BF16 row stream (the bf16 residual stream): x bf16 [M, d]
hipError_t launch_layernorm_xbf16(const void* x_bf16, const float* gamma, const float* beta, void* out_bf16, int ldo, int M,
                                  int d, hipStream_t s) {
    return ln_dispatch<0>((const float*)x_bf16, gamma, beta, nullptr, nullptr, nullptr, 0, 1, (bf16*)out_bf16, ldo, nullptr, M,
                          d, s, 1);
}

// ------------------------------------------------------------------------------------------------
// fp32 -> bf16 cast (text_emb before the cross-attention K/V projection).  8 elements per thread.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                        size_t n8) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i];
        const f32x4 b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
        u32x4 o;
        o[0] = pack_bf16x2(a[0], a[1]);
        o[1] = pack_bf16x2(a[2], a[3]);
        o[2] = pack_bf16x2(b[0], b[1]);
        o[3] = pack_bf16x2(b[2], b[3]);
        reinterpret_cast<u32x4*>(dst)[i] = o;
    }
}
hipError_t launch_cast_bf16(const float* src, void* dst, size_t n, hipStream_t s) {
    if (n % 8) return hipErrorInvalidValue;
    const size_t n8 = n / 8;
    const int grid = (int)((n8 + 255) / 256 < 2048 ? (n8 + 255) / 256 : 2048);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, src, (bf16*)dst, n8);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Split-K finish (small batches: the K = 4d fc2 GEMM on a handful of workgroups, ditto_api.hip run_block):
// out[r, c] = residual[r, c] + bias[c] + sum_s partial[s][r, c], splits added in index order (deterministic);
// optional bf16 copy to out2 (row stride ldo2).  All fp32 arrays are contiguous [M, N]; out may alias residual.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ partial, int nsplit, size_t stride,
                                                            const float* __restrict__ bias, const float* residual,
                                                            float* out, bf16* __restrict__ out2, int ldo2, int N,
                                                            size_t n4) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int nv = N >> 2;
    const size_t row = i / nv;
    const int cv = (int)(i % nv);
    f32x4 acc = reinterpret_cast<const f32x4*>(partial)[i];
    for (int sp = 1; sp < nsplit; ++sp) acc += reinterpret_cast<const f32x4*>(partial + (size_t)sp * stride)[i];
    if (bias) acc += reinterpret_cast<const f32x4*>(bias)[cv];
    if (residual) acc += reinterpret_cast<const f32x4*>(residual)[i];
    reinterpret_cast<f32x4*>(out)[i] = acc;
    if (out2) {
        u32x2 pk;
        pk[0] = pack_bf16x2(acc[0], acc[1]);
        pk[1] = pack_bf16x2(acc[2], acc[3]);
        *reinterpret_cast<u32x2*>(out2 + row * ldo2 + cv * 4) = pk;
    }
}
// The same finish WITH the LayerNorm that follows it (the low-latency class: fc2 -> the next block's norm1, cross out-proj ->
// norm3): one wave per row, the row in registers.  h is summed in splitk_finish_kernel's order and normalised through ln_kernel's
// helpers, so h and u are bit for bit what the finish launch + the LayerNorm launch give — one launch less per use.
template <int CH>
__global__ __launch_bounds__(256) void splitk_finish_ln_kernel(const float* __restrict__ partial, int nsplit, size_t stride,
                                                               const float* __restrict__ bias, const float* residual, float* out,
                                                               bf16* __restrict__ out2, int ldo2, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, bf16* __restrict__ u, int ldu,
                                                               int M, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = d >> 2;
    const size_t r0 = (size_t)row * nv;
    f32x4 v[CH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = lane + 64 * c;
        if (i < nv) {
            f32x4 acc = reinterpret_cast<const f32x4*>(partial)[r0 + i];
            for (int sp = 1; sp < nsplit; ++sp) acc += reinterpret_cast<const f32x4*>(partial + (size_t)sp * stride)[r0 + i];
            if (bias) acc += reinterpret_cast<const f32x4*>(bias)[i];
            if (residual) acc += reinterpret_cast<const f32x4*>(residual)[r0 + i];
            reinterpret_cast<f32x4*>(out)[r0 + i] = acc;
            if (out2) {
                u32x2 pk;
                pk[0] = pack_bf16x2(acc[0], acc[1]);
                pk[1] = pack_bf16x2(acc[2], acc[3]);
                *reinterpret_cast<u32x2*>(out2 + (size_t)row * ldo2 + i * 4) = pk;
            }
            v[c] = acc;
            s += ln_sum4(acc);
        } else {
            v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = lane + 64 * c;
        if (i < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) q = ln_sq_acc(q, v[c][e], mean);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
    bf16* urow = u + (size_t)row * ldu;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = lane + 64 * c;
        if (i < nv) {
            const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[i], b = reinterpret_cast<const f32x4*>(beta)[i];
            u32x2 o;
            o[0] = pack_bf16x2(ln_norm(v[c][0], mean, rstd, g[0], b[0]), ln_norm(v[c][1], mean, rstd, g[1], b[1]));
            o[1] = pack_bf16x2(ln_norm(v[c][2], mean, rstd, g[2], b[2]), ln_norm(v[c][3], mean, rstd, g[3], b[3]));
            *reinterpret_cast<u32x2*>(urow + 4 * i) = o;
        }
    }
}

hipError_t launch_splitk_finish(const float* partial, int nsplit, size_t stride, const float* bias,
                                const float* residual, float* out, void* out2_bf16, int ldo2, int M, int N,
                                hipStream_t s, const float* gamma, const float* beta, void* u_bf16, int ldu) {
    if (!partial || !out || nsplit < 1 || M <= 0 || N <= 0 || N % 4 || (out2_bf16 && ldo2 % 4)) return hipErrorInvalidValue;
    if (u_bf16) {   // + the following LayerNorm (contiguous [M, N] rows: out ld = N)
        if (!gamma || !beta || ldu % 4) return hipErrorInvalidValue;
        const int ch = (N / 4 + 63) / 64;
        dim3 grid((M + 3) / 4), block(256);
#define FIN_CASE(C)                                                                                                        \
    case C:                                                                                                                \
        hipLaunchKernelGGL((splitk_finish_ln_kernel<C>), grid, block, 0, s, partial, nsplit, stride, bias, residual, out,    \
                           (bf16*)out2_bf16, ldo2, gamma, beta, (bf16*)u_bf16, ldu, M, N);                                  \
        break;
        switch (ch) {
            FIN_CASE(1) FIN_CASE(2) FIN_CASE(3) FIN_CASE(4) FIN_CASE(5) FIN_CASE(6) FIN_CASE(7) FIN_CASE(8)
            default: return hipErrorInvalidValue;
        }
#undef FIN_CASE
        return hipGetLastError();
    }
    const size_t n4 = (size_t)M * (N / 4);
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, partial, nsplit, stride,
                       bias, residual, out, (bf16*)out2_bf16, ldo2, N, n4);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// DDPM ancestral update, in place.  Reference: src/model/SpeechGenerator.py:137-145.
// Same operation order as the torch expression; __f*_rn keep hipcc from contracting to FMA so the
// fp32 result is the IEEE sequence torch's CPU kernels produce.
// grid = (blocks per utterance, B): the per-utterance coefficients are wave-uniform.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void p_sample_update_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                              const float* __restrict__ noise,
                                                              const int64_t* __restrict__ t,
                                                              const float* __restrict__ betas,
                                                              const float* __restrict__ alphas,
                                                              const float* __restrict__ acp, size_t n4_per_utt) {
    const int b = blockIdx.y;
    const long long ts = t[b];
    const float beta = betas[ts], alpha = alphas[ts], ac = acp[ts];
    const float inv_sqrt_alpha = __fdiv_rn(1.0f, __fsqrt_rn(alpha));                 // 1 / sqrt(alpha_t)
    const float c_eps = __fdiv_rn(__fsub_rn(1.0f, alpha), __fsqrt_rn(__fsub_rn(1.0f, ac)));  // (1-a)/sqrt(1-acp)
    const float sigma = (ts > 0 ? 1.0f : 0.0f) * __fsqrt_rn(beta);                   // mask * sqrt(beta_t)
    const bool use_noise = (noise != nullptr) && (ts > 0);
    const size_t base = (size_t)b * n4_per_utt;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4_per_utt;
         i += (size_t)gridDim.x * blockDim.x) {
        f32x4 xv = reinterpret_cast<f32x4*>(x)[base + i];
        const f32x4 ev = reinterpret_cast<const f32x4*>(eps)[base + i];
        f32x4 zv = {0.f, 0.f, 0.f, 0.f};
        if (use_noise) zv = reinterpret_cast<const f32x4*>(noise)[base + i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float mean = __fmul_rn(inv_sqrt_alpha, __fsub_rn(xv[e], __fmul_rn(c_eps, ev[e])));
            xv[e] = __fadd_rn(mean, __fmul_rn(sigma, zv[e]));
        }
        reinterpret_cast<f32x4*>(x)[base + i] = xv;
    }
}
hipError_t launch_p_sample_update(float* x, const float* eps, const float* noise, const int64_t* t,
                                  const float* betas, const float* alphas, const float* acp, int B,
                                  size_t elems_per_utt, hipStream_t s) {
    if (elems_per_utt % 4) return hipErrorInvalidValue;
    const size_t n4 = elems_per_utt / 4;
    size_t gx = (n4 + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(p_sample_update_kernel, dim3((unsigned)gx, B), dim3(256), 0, s, x, eps, noise, t, betas,
                       alphas, acp, n4);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Per-utterance counter-based N(0,1): Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3",
// SC'11; the Random123 constants) keyed by the utterance's 64-bit seed, counter = (element quad, step, stream), then
// Box-Muller on the four 32-bit words.  Element i of utterance b at step s depends on (seed[b], s, i) ONLY — not on
// the batch the utterance sits in, its position in it, or the GPU: what makes batch-sharded sampling reproduce the
// unsharded result bit for bit.  The reference draws from torch's global generator
// (src/model/SpeechGenerator.py:141,154), whose stream cannot be sharded; that path stays the default.
// ------------------------------------------------------------------------------------------------
DITTO_DEV void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
// four N(0,1) of (seed, step, quad index): u = ((word >> 8) + 0.5) 2^-24 in (0,1) (the top 24 bits of a Philox word);
// r = sqrt(-2 ln u1); angle = 2 pi u2
DITTO_DEV f32x4 normal4(unsigned long long seed, unsigned step, unsigned long long quad) {
    unsigned w[4];
    philox4x32_10((unsigned)quad, (unsigned)(quad >> 32), step, 0x44695454u /* "DiTT" */, (unsigned)seed,
                  (unsigned)(seed >> 32), w);
    f32x4 z;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(w[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);        // 24 bits: exact in fp32, never 0 or 1
        const float u2 = ((float)(w[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float r = __fsqrt_rn(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln u = -2 ln2 log2 u
        z[2 * h] = r * __builtin_amdgcn_cosf(u2);                                        // v_cos / v_sin take revolutions
        z[2 * h + 1] = r * __builtin_amdgcn_sinf(u2);
    }
    return z;
}
__global__ __launch_bounds__(256) void noise_normal_kernel(float* __restrict__ out, const int64_t* __restrict__ seeds,
                                                           unsigned step, size_t n4_per_utt) {
    const int b = blockIdx.y;
    const unsigned long long seed = (unsigned long long)seeds[b];
    const size_t base = (size_t)b * n4_per_utt;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4_per_utt; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(out)[base + i] = normal4(seed, step, i);
}
hipError_t launch_noise_normal(float* out, const int64_t* seeds, unsigned step, int B, size_t elems_per_utt, hipStream_t s) {
    if (elems_per_utt % 4) return hipErrorInvalidValue;
    const size_t n4 = elems_per_utt / 4;
    size_t gx = (n4 + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(noise_normal_kernel, dim3((unsigned)gx, B), dim3(256), 0, s, out, seeds, step, n4);
    return hipGetLastError();
}
// the DDPM update with the step's noise generated in registers (no z buffer: 2 x 4 bytes per element less HBM traffic,
// one launch less); the noise of utterance b is normal4(seeds[b], step, .), i.e. exactly what launch_noise_normal
// would have written
__global__ __launch_bounds__(256) void p_sample_update_seeded_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                                     const int64_t* __restrict__ seeds, unsigned step,
                                                                     const int64_t* __restrict__ t,
                                                                     const float* __restrict__ betas,
                                                                     const float* __restrict__ alphas,
                                                                     const float* __restrict__ acp, size_t n4_per_utt) {
    const int b = blockIdx.y;
    const long long ts = t[b];
    const float beta = betas[ts], alpha = alphas[ts], ac = acp[ts];
    const float inv_sqrt_alpha = __fdiv_rn(1.0f, __fsqrt_rn(alpha));
    const float c_eps = __fdiv_rn(__fsub_rn(1.0f, alpha), __fsqrt_rn(__fsub_rn(1.0f, ac)));
    const float sigma = (ts > 0 ? 1.0f : 0.0f) * __fsqrt_rn(beta);
    const unsigned long long seed = (unsigned long long)seeds[b];
    const size_t base = (size_t)b * n4_per_utt;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4_per_utt;
         i += (size_t)gridDim.x * blockDim.x) {
        f32x4 xv = reinterpret_cast<f32x4*>(x)[base + i];
        const f32x4 ev = reinterpret_cast<const f32x4*>(eps)[base + i];
        f32x4 zv = {0.f, 0.f, 0.f, 0.f};
        if (ts > 0) zv = normal4(seed, step, i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float mean = __fmul_rn(inv_sqrt_alpha, __fsub_rn(xv[e], __fmul_rn(c_eps, ev[e])));
            xv[e] = __fadd_rn(mean, __fmul_rn(sigma, zv[e]));
        }
        reinterpret_cast<f32x4*>(x)[base + i] = xv;
    }
}
hipError_t launch_p_sample_update_seeded(float* x, const float* eps, const int64_t* seeds, unsigned step, const int64_t* t,
                                         const float* betas, const float* alphas, const float* acp, int B,
                                         size_t elems_per_utt, hipStream_t s) {
    if (elems_per_utt % 4) return hipErrorInvalidValue;
    const size_t n4 = elems_per_utt / 4;
    size_t gx = (n4 + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(p_sample_update_seeded_kernel, dim3((unsigned)gx, B), dim3(256), 0, s, x, eps, seeds, step, t, betas,
                       alphas, acp, n4);
    return hipGetLastError();
}

// q_sample, reference src/model/DiTTO.py:106-126 (bug-for-bug: `buffer` holds clipped betas).
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const int64_t* __restrict__ t,
                                                       const float* __restrict__ buffer, float* __restrict__ out,
                                                       size_t n4_per_utt) {
    const int b = blockIdx.y;
    const float c = buffer[t[b]];
    const float a = __fsqrt_rn(c), s1 = __fsqrt_rn(__fsub_rn(1.0f, c));
    const size_t base = (size_t)b * n4_per_utt;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4_per_utt;
         i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 xv = reinterpret_cast<const f32x4*>(x0)[base + i];
        const f32x4 zv = reinterpret_cast<const f32x4*>(noise)[base + i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(__fmul_rn(a, xv[e]), __fmul_rn(s1, zv[e]));
        reinterpret_cast<f32x4*>(out)[base + i] = o;
    }
}
hipError_t launch_q_sample(const float* x0, const float* noise, const int64_t* t, const float* buffer, float* out,
                           int B, size_t elems_per_utt, hipStream_t s) {
    if (elems_per_utt % 4) return hipErrorInvalidValue;
    const size_t n4 = elems_per_utt / 4;
    size_t gx = (n4 + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)gx, B), dim3(256), 0, s, x0, noise, t, buffer, out, n4);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Timestep -> AdaLN (scale, shift) table, built once per checkpoint at model-create time:
//   ttab[s] = time_mlp( time_embed( t_embedding[s] ) )
// Reference: src/model/DiTTO.py:75-76 (Embedding, Linear, SiLU, Linear) + src/components/DiT.py:14-17,30
// (SiLU, Linear).  One block per timestep; fp32 FMA; vectors staged in LDS.  Init-time only.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_table_kernel(const float* __restrict__ emb, const float* __restrict__ w0,
                                                         const float* __restrict__ b0, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, const float* __restrict__ wt,
                                                         const float* __restrict__ bt, float* __restrict__ ttab,
                                                         int td, int d) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* e = sm;           // [td]
    float* a = sm + td;      // [td]
    float* c = sm + 2 * td;  // [td]
    const int step = blockIdx.x;
    for (int j = threadIdx.x; j < td; j += blockDim.x) e[j] = emb[(size_t)step * td + j];
    __syncthreads();
    for (int j = threadIdx.x; j < td; j += blockDim.x) {
        float acc = 0.f;
        const float* wr = w0 + (size_t)j * td;
        for (int k = 0; k < td; ++k) acc = fmaf(wr[k], e[k], acc);
        a[j] = silu_f(acc + b0[j]);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < td; j += blockDim.x) {
        float acc = 0.f;
        const float* wr = w2 + (size_t)j * td;
        for (int k = 0; k < td; ++k) acc = fmaf(wr[k], a[k], acc);
        c[j] = silu_f(acc + b2[j]);  // SiLU of GlobalAdaLN.time_mlp[0]
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * d; j += blockDim.x) {
        float acc = 0.f;
        const float* wr = wt + (size_t)j * td;
        for (int k = 0; k < td; ++k) acc = fmaf(wr[k], c[k], acc);
        ttab[(size_t)step * 2 * d + j] = acc + bt[j];
    }
}
hipError_t launch_time_table(const float* emb, const float* w0, const float* b0, const float* w2, const float* b2,
                             const float* wt, const float* bt, float* ttab, int steps, int td, int d, hipStream_t s) {
    hipLaunchKernelGGL(time_table_kernel, dim3(steps), dim3(256), 3 * td * sizeof(float), s, emb, w0, b0, w2, b2, wt,
                       bt, ttab, td, d);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Text half of GlobalAdaLN, once per utterance batch (step-invariant):
//   pooled = mean_T(text)  (no mask, src/components/DiT.py:27);  tmod = Linear(SiLU(pooled)) (:19-22,31)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_pool_kernel(const float* __restrict__ text, float* __restrict__ pooled,
                                                        int T, int dt) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6, b = blockIdx.y;
    float acc = 0.f;
    if (col < dt)
        for (int r = g; r < T; r += 4) acc += text[((size_t)b * T + r) * dt + col];
    red[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0 && col < dt)
        pooled[(size_t)b * dt + col] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                        (red[2][threadIdx.x] + red[3][threadIdx.x])) / (float)T;
}
// one wave per output row j of the [2d, dt] matrix
__global__ __launch_bounds__(256) void text_mod_kernel(const float* __restrict__ pooled, const float* __restrict__ wx,
                                                       const float* __restrict__ bx, float* __restrict__ tmod, int dt,
                                                       int d2) {
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (j >= d2) return;
    const float* p = pooled + (size_t)b * dt;
    const float* w = wx + (size_t)j * dt;
    float acc = 0.f;
    for (int k = lane; k < dt; k += 64) acc = fmaf(w[k], silu_f(p[k]), acc);
    acc = wave_sum(acc);
    if (lane == 0) tmod[(size_t)b * d2 + j] = acc + bx[j];
}
hipError_t launch_text_mod(const float* text, const float* wx, const float* bx, float* pooled, float* tmod, int B,
                           int T, int dt, int d, hipStream_t s) {
    hipLaunchKernelGGL(text_pool_kernel, dim3((dt + 63) / 64, B), dim3(256), 0, s, text, pooled, T, dt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(text_mod_kernel, dim3((2 * d + 3) / 4, B), dim3(256), 0, s, pooled, wx, bx, tmod, dt, 2 * d);
    return hipGetLastError();
}

// RotaryEmbedding.forward tables (src/components/DiT.py:56-59): angle = float(n) * inv_freq[j] in fp32.
__global__ __launch_bounds__(256) void rope_table_kernel(const float* __restrict__ inv_freq, float* __restrict__ c,
                                                         float* __restrict__ sn, int N, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * half) return;
    const float ang = __fmul_rn((float)(i / half), inv_freq[i % half]);
    c[i] = cosf(ang);
    sn[i] = sinf(ang);
}
hipError_t launch_rope_tables(const float* inv_freq, float* c, float* sn, int N, int half, hipStream_t s) {
    hipLaunchKernelGGL(rope_table_kernel, dim3((N * half + 255) / 256), dim3(256), 0, s, inv_freq, c, sn, N, half);
    return hipGetLastError();
}

// RotaryEmbedding.apply_rope on fp32 [B,N,H,dh] with the ANGLE table pos [N,dh] the reference passes around
// (src/components/DiT.py:61-72).  Component-level surface only; the model path fuses RoPE into the QKV GEMM.
__global__ __launch_bounds__(256) void apply_rope_f32_kernel(const float* __restrict__ pos, const float* __restrict__ x,
                                                             float* __restrict__ out, size_t total, int N, int H,
                                                             int dh) {
    const int half = dh >> 1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % dh);
        const int n = (int)((i / ((size_t)dh * H)) % N);
        const float a = pos[(size_t)n * dh + j];
        const float rot = j < half ? -x[i + half] : x[i - half];   // _rotate_half: cat(-x2, x1)
        out[i] = x[i] * cosf(a) + rot * sinf(a);
    }
}
hipError_t launch_apply_rope_f32(const float* pos, const float* x, float* out, int B, int N, int H, int dh,
                                 hipStream_t s) {
    const size_t total = (size_t)B * N * H * dh;
    size_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(apply_rope_f32_kernel, dim3((unsigned)(g ? g : 1)), dim3(256), 0, s, pos, x, out, total, N, H, dh);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Weight packing (model-create time): fp32 [rows, cols] -> bf16 with a block row map.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                        int rows, int cols, int dst_ld, int col_off, int blk,
                                                        int mult, int row_off, float scale) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        const size_t dr = (size_t)(r / blk) * ((size_t)blk * mult) + (r % blk) + row_off;
        dst[dr * dst_ld + col_off + c] = (bf16)(src[i] * scale);
    }
}
hipError_t launch_pack_bf16(const float* src, void* dst, int rows, int cols, int dst_ld, int col_off, int blk,
                            int mult, int row_off, hipStream_t s, float scale) {
    const size_t n = (size_t)rows * cols;
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)(g ? g : 1)), dim3(256), 0, s, src, (bf16*)dst, rows, cols,
                       dst_ld, col_off, blk, mult, row_off, scale);
    return hipGetLastError();
}
// fp32 [N, K] (nn.Linear.weight) -> bf16 stage-major [K/16][N][16] for the full-row GEMM (gemm_fr.hip): each K-step's
// N x 16 slab is contiguous
__global__ __launch_bounds__(256) void pack_bf16_stage_major_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                                    int N, int K) {
    const size_t n = (size_t)N * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / K), c = (int)(i % K);
        dst[((size_t)(c >> 4) * N + r) * 16 + (c & 15)] = (bf16)src[i];
    }
}
hipError_t launch_pack_bf16_stage_major(const float* src, void* dst, int N, int K, hipStream_t s) {
    if (K % 16) return hipErrorInvalidValue;   // a ragged last stage would be written past the N * K image
    const size_t n = (size_t)N * K;
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_bf16_stage_major_kernel, dim3((unsigned)(g ? g : 1)), dim3(256), 0, s, src, (bf16*)dst, N, K);
    return hipGetLastError();
}
// the same re-layout of an already packed bf16 [N, K] image (the transposed weights of the dgrad GEMMs)
__global__ __launch_bounds__(256) void repack_bf16_stage_major_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                                      int N, int K, int G) {
    const size_t n8 = (size_t)N * K / 8;                 // 16-B chunks: a chunk stays whole inside its G-element group
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 8;
        const int r = (int)(e / K), c = (int)(e % K);
        *reinterpret_cast<u32x4*>(dst + ((size_t)(c / G) * N + r) * G + (c % G)) = *reinterpret_cast<const u32x4*>(src + e);
    }
}
// group = k per stage: 16 (the 32x32x16 kernels' image Wp[K/16][N][16]) or 32 (the 16x16x32 image Wp[K/32][N][32], gemm_lnq.hip)
hipError_t launch_repack_bf16_stage_major(const void* src, void* dst, int N, int K, hipStream_t s, int group) {
    if ((group != 16 && group != 32) || K % group) return hipErrorInvalidValue;
    const size_t n8 = (size_t)N * K / 8;
    size_t g = (n8 + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(repack_bf16_stage_major_kernel, dim3((unsigned)(g ? g : 1)), dim3(256), 0, s, (const bf16*)src,
                       (bf16*)dst, N, K, group);
    return hipGetLastError();
}
// ------------------------------------------------------------------------------------------------
// Head padding (head_dim % 64 != 0, e.g. the paper's XL shape 1152 / 16 = 72): the attention GEMMs need head widths that
// are multiples of the 64-deep K tile, so inside the block the heads of q / k / v live at a stride of dhp = roundup(dh, 64)
// columns with ZERO pads — zero weight rows and biases in the projections that produce them (q.k^T and P.V are then
// unchanged: the pads add 0), zero weight COLUMNS in the out-projection that consumes them.  These kernels build those
// images; the destination regions are zero-filled first.
// ------------------------------------------------------------------------------------------------
// dst[(r / dh) * dhp + r % dh + dst_row_off][c] = bf16(src[r][c] * scale): the rows of an in-projection, head by head
__global__ __launch_bounds__(256) void pack_bf16_headrows_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                                 int rows, int cols, int dst_ld, int dh, int dhp,
                                                                 int dst_row_off, float scale) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        dst[(size_t)((r / dh) * dhp + r % dh + dst_row_off) * dst_ld + c] = (bf16)(src[i] * scale);
    }
}
// dst[r][(c / dh) * dhp + c % dh] = bf16(src[r][c]): the columns of an out-projection
__global__ __launch_bounds__(256) void pack_bf16_headcols_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                                 int rows, int cols, int dst_ld, int dh, int dhp) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        dst[(size_t)r * dst_ld + (c / dh) * dhp + c % dh] = (bf16)src[i];
    }
}
__global__ void pack_vec_heads_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int dh, int dhp,
                                      int dst_off, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[(i / dh) * dhp + i % dh + dst_off] = src[i] * scale;
}
// h[row, hd * dh + c] += o[row, hd * dhp + c]: the self-attention's head merge + residual (src/components/DiT.py:137-139: no
// out-projection) when the heads were computed at the padded stride
__global__ __launch_bounds__(256) void head_compact_add_kernel(const bf16* __restrict__ o, int ldo, float* __restrict__ h,
                                                               int ldh, size_t n, int d, int dh, int dhp) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / d;
        const int c = (int)(i % d);
        h[row * ldh + c] += (float)o[row * ldo + (c / dh) * dhp + c % dh];
    }
}
static unsigned ew_blocks(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}
hipError_t launch_pack_bf16_headrows(const float* src, void* dst, int rows, int cols, int dst_ld, int dh, int dhp,
                                     int dst_row_off, hipStream_t s, float scale) {
    hipLaunchKernelGGL(pack_bf16_headrows_kernel, dim3(ew_blocks((size_t)rows * cols)), dim3(256), 0, s, src, (bf16*)dst, rows,
                       cols, dst_ld, dh, dhp, dst_row_off, scale);
    return hipGetLastError();
}
hipError_t launch_pack_bf16_headcols(const float* src, void* dst, int rows, int cols, int dst_ld, int dh, int dhp,
                                     hipStream_t s) {
    hipLaunchKernelGGL(pack_bf16_headcols_kernel, dim3(ew_blocks((size_t)rows * cols)), dim3(256), 0, s, src, (bf16*)dst, rows,
                       cols, dst_ld, dh, dhp);
    return hipGetLastError();
}
hipError_t launch_pack_vec_heads(const float* src, float* dst, int n, int dh, int dhp, int dst_off, hipStream_t s, float scale) {
    hipLaunchKernelGGL(pack_vec_heads_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n, dh, dhp, dst_off, scale);
    return hipGetLastError();
}
hipError_t launch_head_compact_add(const void* o_bf16, int ldo, float* h, int ldh, int M, int d, int dh, int dhp, hipStream_t s) {
    hipLaunchKernelGGL(head_compact_add_kernel, dim3(ew_blocks((size_t)M * d)), dim3(256), 0, s, (const bf16*)o_bf16, ldo, h, ldh,
                       (size_t)M * d, d, dh, dhp);
    return hipGetLastError();
}

__global__ void scale_vec_kernel(float* __restrict__ v, int n, float f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] *= f;
}
hipError_t launch_scale_vec(float* v, int n, float f, hipStream_t s) {
    hipLaunchKernelGGL(scale_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, s, v, n, f);
    return hipGetLastError();
}
// fp32 [rows, cols] -> fp8 e4m3 rows with a per-row scale (amax / 448): one wave per row.
__global__ __launch_bounds__(256) void pack_fp8_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst,
                                                       float* __restrict__ scales, int rows, int cols, int dst_ld,
                                                       int blk, int mult, int row_off) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* sr = src + (size_t)r * cols;
    float amax = 0.f;
    for (int c = lane; c < cols; c += 64) amax = fmaxf(amax, fabsf(sr[c]));
    amax = wave_max(amax);
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f, inv = 1.0f / scale;
    const size_t dr = (size_t)(r / blk) * ((size_t)blk * mult) + (r % blk) + row_off;
    if (lane == 0) scales[dr] = scale;
    for (int c = lane * 4; c < cols; c += 256)   // cols % 4 == 0
        *reinterpret_cast<unsigned*>(dst + dr * dst_ld + c) =
            pack_fp8x4(sr[c] * inv, sr[c + 1] * inv, sr[c + 2] * inv, sr[c + 3] * inv);
}
hipError_t launch_pack_fp8(const float* src, void* dst, float* scales, int rows, int cols, int dst_ld, int blk, int mult,
                           int row_off, hipStream_t s) {
    if (cols % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_fp8_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, src, (unsigned char*)dst, scales, rows,
                       cols, dst_ld, blk, mult, row_off);
    return hipGetLastError();
}
__global__ void pack_vec_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int blk, int mult,
                                int row_off) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) dst[(size_t)(r / blk) * ((size_t)blk * mult) + (r % blk) + row_off] = src[r];
}
hipError_t launch_pack_vec(const float* src, float* dst, int rows, int blk, int mult, int row_off, hipStream_t s) {
    hipLaunchKernelGGL(pack_vec_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, src, dst, rows, blk, mult, row_off);
    return hipGetLastError();
}
__global__ void add_vec_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dst,
                               int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = a[i] + b[i];
}
__global__ void fill_i64_kernel(int64_t* __restrict__ dst, int n, int64_t value) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}
hipError_t launch_fill_i64(int64_t* dst, int n, int64_t value, hipStream_t s) {
    hipLaunchKernelGGL(fill_i64_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, n, value);
    return hipGetLastError();
}
hipError_t launch_add_vec(const float* a, const float* b, float* dst, int n, hipStream_t s) {
    hipLaunchKernelGGL(add_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, dst, n);
    return hipGetLastError();
}

}  // namespace ditto
