// train.hip — row / elementwise kernels of the DiT BACKWARD pass for gfx950.
//
// The contractions of the backward pass (dgrad  dX = dY W,  wgrad  dW = dY^T X) run on the forward GEMM family
// (gemm.hip) with pre-transposed operands; everything in this file is HBM-bound glue around them, judged against
// HBM: 16-byte per-lane accesses, one wave per row with shuffle reductions, deterministic two-stage column
// reductions (no atomics => bit-reproducible gradients).
//
//   transpose_bf16      X[rows, cols] -> X^T[cols, ldT] (zero padded to a multiple of 64 along the new K)
//   colsum_*            bias gradients: column sums over the M rows of dY
//   ln_bwd              LayerNorm backward (reference nn.LayerNorm, src/components/DiT.py:84,89,94) and the
//                       scale/shift reductions of GlobalAdaLN's backward (src/components/DiT.py:34-39)
//   gated_bwd           derivative of gelu(a) * sigmoid(g) (src/components/DiT.py:152-154; the forward is a GEMM epilogue)
//   softmax_drop_rows / softmax_bwd_rows   generic-head_dim attention with the train-mode dropout of
//                       nn.MultiheadAttention (src/components/DiT.py:90-91) from a counter-based hash
//   small_linear_*      the [B, .] x [., .] affine maps of the time / text modulation vectors, fp32
#include "common.h"
#include "kernels.h"

namespace ditto {

// ---------------------------------------------------------------------------------------------------------
// dst[c][r] = src[r][c]; dst is [cols, ldT] bf16 with ldT >= rows, columns rows..ldT-1 written as zero.
// 64x64 tile through LDS (row pitch 66 elements = 33 words: the column gather below is conflict-free).
// ---------------------------------------------------------------------------------------------------------
// blockIdx.z = z (batched): src += (z / zi) * s_o + (z % zi) * s_i, dst += z * d_z  (elements)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16* __restrict__ src, int ld, int rows, int cols,
                                                             bf16* __restrict__ dst, int ldT, int zi, long long s_o,
                                                             long long s_i, long long d_z) {
    __shared__ bf16 tile[64][66];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    src += (long long)(blockIdx.z / zi) * s_o + (long long)(blockIdx.z % zi) * s_i;
    dst += (long long)blockIdx.z * d_z;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;          // 512 chunks of 8
        const int r = idx >> 3, ch = idx & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < rows && c0 + ch * 8 < cols) v = *reinterpret_cast<const u32x4*>(src + (size_t)(r0 + r) * ld + c0 + ch * 8);
        unsigned* t = reinterpret_cast<unsigned*>(&tile[r][ch * 8]);
        t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        const int c = idx >> 3, ch = idx & 7;   // output row c (source column), 8 consecutive source rows
        if (c0 + c >= cols) continue;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = tile[ch * 8 + e][c];
        *reinterpret_cast<bf16x8*>(dst + (size_t)(c0 + c) * ldT + r0 + ch * 8) = o;
    }
}
hipError_t launch_transpose_bf16(const void* src, int ld, int rows, int cols, void* dst, int ldT, hipStream_t s, int nz_o,
                                 int nz_i, long long s_o, long long s_i, long long d_z) {
    if (ld % 8 || cols % 8 || ldT % 64 || ldT < rows || nz_o < 1 || nz_i < 1 || nz_o * nz_i > 65535) return hipErrorInvalidValue;
    if ((s_o | s_i | d_z) % 8) return hipErrorInvalidValue;   // 16-byte accesses
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(ldT / 64, (cols + 63) / 64, nz_o * nz_i), dim3(256), 0, s,
                       (const bf16*)src, ld, rows, cols, (bf16*)dst, ldT, nz_i, s_o, s_i, d_z);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Column sums, two deterministic stages.  Stage 1: block = 64 column-lanes (4 columns each) x 4 row-lanes, one
// block per (256-column strip, row chunk) -> partial[chunk][n].  Stage 2: out[j] = sum over chunks.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int ld, int M, int n,
                                                             float* __restrict__ partial, int rows_per_chunk) {
    __shared__ f32x4 red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + tx * 4;
    const int rbeg = blockIdx.y * rows_per_chunk;
    const int rend = rbeg + rows_per_chunk < M ? rbeg + rows_per_chunk : M;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < n) {
        for (int r = rbeg + ty; r < rend; r += 4) {
            if constexpr (sizeof(T) == 4) {
                acc += *reinterpret_cast<const f32x4*>(x + (size_t)r * ld + col);
            } else {
                const u32x2 v = *reinterpret_cast<const u32x2*>(x + (size_t)r * ld + col);
                acc[0] += bf16_lo(v[0]); acc[1] += bf16_hi(v[0]); acc[2] += bf16_lo(v[1]); acc[3] += bf16_hi(v[1]);
            }
        }
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && col < n) {
        const f32x4 r = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
        *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.y * n + col) = r;
    }
}
// The bf16 form the backward calls 36 times per step (bias gradients = column sums of dq, dk | dv, dq | dk | dv): 16-B loads
// (8 columns per lane: 32 column-lanes x 8 row-lanes per 256-column strip) and four rows in flight per lane — the 8-B, one-row-
// at-a-time loop above ran at 1.7 TB/s (30 us for the 50 - 150 MB of a call).  Same partial layout, fixed summation order.
__global__ __launch_bounds__(256) void colsum_partial_bf16x8_kernel(const bf16* __restrict__ x, int ld, int M, int n,
                                                                    float* __restrict__ partial, int rows_per_chunk) {
    __shared__ float red[8][32][9];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int col = blockIdx.x * 256 + tx * 8;
    const int rbeg = blockIdx.y * rows_per_chunk;
    const int rend = rbeg + rows_per_chunk < M ? rbeg + rows_per_chunk : M;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    auto add = [&](const u32x4& v) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[2 * k] += bf16_lo(v[k]); acc[2 * k + 1] += bf16_hi(v[k]); }
    };
    if (col < n) {
        const bf16* base = x + col;
        int r = rbeg + ty;
        for (; r + 24 < rend; r += 32) {
            const u32x4 v0 = *reinterpret_cast<const u32x4*>(base + (size_t)r * ld);
            const u32x4 v1 = *reinterpret_cast<const u32x4*>(base + (size_t)(r + 8) * ld);
            const u32x4 v2 = *reinterpret_cast<const u32x4*>(base + (size_t)(r + 16) * ld);
            const u32x4 v3 = *reinterpret_cast<const u32x4*>(base + (size_t)(r + 24) * ld);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; r < rend; r += 8) add(*reinterpret_cast<const u32x4*>(base + (size_t)r * ld));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[ty][tx][k] = acc[k];
    __syncthreads();
    if (ty == 0 && col < n) {
        float* prow = partial + (size_t)blockIdx.y * n + col;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float r = red[0][tx][k];
#pragma unroll
            for (int j = 1; j < 8; ++j) r += red[j][tx][k];
            prow[k] = r;
        }
    }
}
// out[g][j] = sum_c partial[g][c][j]   (groups of `chunks` partial rows).  Latency-bound (n is a few thousand columns,
// chunks up to 256): block = 16 columns x 16 chunk-lanes, each lane keeps 4 independent running sums so that 4 loads
// are in flight, lanes combined through LDS in a fixed order (deterministic).  The first version (64 columns x 4
// chunk-lanes, one dependent chain of 64 loads on a 12-block grid) took 19 us per call, 110 calls per training step.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int chunks, size_t n,
                                                              float* __restrict__ out) {
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const size_t j = (size_t)blockIdx.x * 16 + tx;
    const int g = blockIdx.y;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (j < n) {
        const float* p = partial + (size_t)g * chunks * n + j;
        int c = ty;
        for (; c + 48 < chunks; c += 64) {
            a0 += p[(size_t)c * n]; a1 += p[(size_t)(c + 16) * n]; a2 += p[(size_t)(c + 32) * n];
            a3 += p[(size_t)(c + 48) * n];
        }
        for (; c < chunks; c += 16) a0 += p[(size_t)c * n];
    }
    red[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && j < n) {
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) r += red[i][tx];
        out[(size_t)g * n + j] = r;
    }
}
// the same sum for LARGE n (split-K weight-gradient partials: a few slices of megabytes each): 16 B per lane, the
// slices added in index order — HBM-bound, every byte read once in 1-KiB wave rows
__global__ __launch_bounds__(256) void reduce_slices_kernel(const float* __restrict__ partial, int chunks, size_t n4,
                                                            float* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(partial) + i;
    f32x4 acc = p[0];
    for (int c = 1; c < chunks; ++c) acc += p[(size_t)c * n4];
    reinterpret_cast<f32x4*>(out)[i] = acc;
}
hipError_t launch_reduce_partials(const float* partial, int chunks, size_t n, float* out, hipStream_t s) {
    if (n >= 65536 && n % 4 == 0 && chunks <= 32) {
        const size_t n4 = n / 4;
        hipLaunchKernelGGL(reduce_slices_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, partial, chunks, n4,
                           out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 15) / 16), 1), dim3(256), 0, s, partial, chunks, n, out);
    return hipGetLastError();
}
static int colsum_chunks(int M, int n) {
    const int strips = (n + 255) / 256;
    int chunks = 1024 / strips;
    if (chunks < 1) chunks = 1;
    if (chunks > 256) chunks = 256;
    const int maxc = (M + 15) / 16;
    if (chunks > maxc) chunks = maxc;
    return chunks < 1 ? 1 : chunks;
}
template <typename T>
static hipError_t colsum_launch(const T* x, int ld, int M, int n, float* out, float* scratch, hipStream_t s) {
    if (n % 4 || ld % 4) return hipErrorInvalidValue;
    const int chunks = colsum_chunks(M, n);
    const int rpc = (M + chunks - 1) / chunks;
    if constexpr (sizeof(T) == 2) {
        if (n % 8 == 0 && ld % 8 == 0 && ((uintptr_t)x & 15) == 0) {
            hipLaunchKernelGGL(colsum_partial_bf16x8_kernel, dim3((n + 255) / 256, chunks), dim3(256), 0, s, (const bf16*)x, ld, M,
                               n, scratch, rpc);
            hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 15) / 16, 1), dim3(256), 0, s, scratch, chunks, (size_t)n, out);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((colsum_partial_kernel<T>), dim3((n + 255) / 256, chunks), dim3(256), 0, s, x, ld, M, n,
                       scratch, rpc);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 15) / 16, 1), dim3(256), 0, s, scratch, chunks, (size_t)n, out);
    return hipGetLastError();
}
hipError_t launch_colsum_f32(const float* x, int ld, int M, int n, float* out, float* scratch, hipStream_t s) {
    return colsum_launch<float>(x, ld, M, n, out, scratch, s);
}
hipError_t launch_colsum_bf16(const void* x, int ld, int M, int n, float* out, float* scratch, hipStream_t s) {
    return colsum_launch<bf16>((const bf16*)x, ld, M, n, out, scratch, s);
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm backward.  y = xhat * gamma + beta, xhat = (x - mean) * rstd (biased variance, eps 1e-5):
//   g = dy * gamma;   dx = rstd * (g - mean(g) - xhat * mean(g * xhat));   dgamma = sum_rows dy * xhat;
//   dbeta = sum_rows dy.
// One wave per row; a workgroup owns `rows_per_chunk` consecutive rows of ONE group (group = utterance for the
// GlobalAdaLN reductions, the whole matrix for a plain LayerNorm) and writes its partial [dgamma | dbeta] row to
// partial[(group * chunks + chunk)][2d]; reduce_partials_kernel finishes.  dx is ADDED into dx_accum (the gradient
// of the residual stream) when dx_accum != nullptr; gamma == nullptr means gamma = 1 (dx then is the no-affine LN).
// ---------------------------------------------------------------------------------------------------------
// EXT (the residual-stream uses of the model's backward): also writes bf16(dx_accum) — the dY operand of the next dgrad /
// wgrad GEMMs — to dx_bf16, and a third partial row sum_rows dx_accum (the bias gradient of the Linear whose output was
// added to the stream there), so that the separate cast and column-sum passes over the 100 MB stream disappear.
// DYB (stream form only): dy is BF16 [rows, d] — the dgrad GEMM in front writes its output once, as bf16 (round 4: 50 MB
// less written there and 50 MB less read here per LayerNorm, 36 per step at C2).
// XB (stream form only): x, the LayerNorm's input row, is BF16 (the training forward on the bf16 residual stream).
template <int CH, bool EXT = false, bool DYB = false, bool XB = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, float* __restrict__ dx_accum,
                                                     float* __restrict__ partial, int d, int rows_per_group,
                                                     int rows_per_chunk, int chunks, bf16* __restrict__ dx_bf16 = nullptr) {
    constexpr int NP = EXT ? 3 : 2;
    __shared__ f32x4 red[4][NP * CH * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int chunk = blockIdx.x, group = blockIdx.y;
    const int nv = d >> 2;
    const int rbeg = chunk * rows_per_chunk;
    const int rend = rbeg + rows_per_chunk < rows_per_group ? rbeg + rows_per_chunk : rows_per_group;
    f32x4 dg[CH], db[CH], gm[CH], dc[EXT ? CH : 1];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        dg[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (EXT) dc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int i = lane + 64 * c;
        gm[c] = (gamma && i < nv) ? reinterpret_cast<const f32x4*>(gamma)[i] : f32x4{1.f, 1.f, 1.f, 1.f};
    }
    for (int r = rbeg + wv; r < rend; r += 4) {
        const size_t row = (size_t)group * rows_per_group + r;
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * d);
        const f32x4* dr = reinterpret_cast<const f32x4*>(dy + row * d);
        const u32x2* drb = reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16*>(dy) + row * d);
        const u32x2* xrb = reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16*>(x) + row * d);
        f32x4 v[CH], gy[CH];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = lane + 64 * c;
            if (i < nv) {
                if constexpr (XB) {
                    const u32x2 t = xrb[i];
                    v[c] = f32x4{bf16_lo(t[0]), bf16_hi(t[0]), bf16_lo(t[1]), bf16_hi(t[1])};
                } else {
                    v[c] = xr[i];
                }
                if constexpr (DYB) {
                    const u32x2 t = drb[i];
                    gy[c] = f32x4{bf16_lo(t[0]), bf16_hi(t[0]), bf16_lo(t[1]), bf16_hi(t[1])};
                } else {
                    gy[c] = dr[i];
                }
                s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
            } else {
                v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                gy[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = lane + 64 * c;
            if (i < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dl = v[c][e] - mean;
                    q += dl * dl;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = lane + 64 * c;
            if (i < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (v[c][e] - mean) * rstd;
                    v[c][e] = xh;
                    dg[c][e] += gy[c][e] * xh;
                    db[c][e] += gy[c][e];
                    const float g = gy[c][e] * gm[c][e];
                    gy[c][e] = g;
                    s1 += g;
                    s2 += g * xh;
                }
            }
        }
        if (dx_accum) {
            const float m1 = wave_sum(s1) / (float)d, m2 = wave_sum(s2) / (float)d;
            f32x4* ar = reinterpret_cast<f32x4*>(dx_accum + row * d);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int i = lane + 64 * c;
                if (i < nv) {
                    f32x4 a = ar[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] += rstd * (gy[c][e] - m1 - v[c][e] * m2);
                    ar[i] = a;
                    if constexpr (EXT) {
                        dc[c] += a;
                        u32x2 pk;
                        pk[0] = pack_bf16x2(a[0], a[1]); pk[1] = pack_bf16x2(a[2], a[3]);
                        reinterpret_cast<u32x2*>(dx_bf16 + row * d)[i] = pk;
                    }
                }
            }
        }
    }
    if (!partial) return;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        red[wv][c * 64 + lane] = dg[c];
        red[wv][(CH + c) * 64 + lane] = db[c];
        if constexpr (EXT) red[wv][(2 * CH + c) * 64 + lane] = dc[c];
    }
    __syncthreads();
    float* prow = partial + ((size_t)group * chunks + chunk) * NP * d;
    for (int idx = threadIdx.x; idx < NP * CH * 64; idx += 256) {
        const int c = (idx / 64) % CH, half = idx / (64 * CH), i = (idx & 63) + 64 * c;
        if (i < nv) {
            const f32x4 r = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
            *reinterpret_cast<f32x4*>(prow + (size_t)half * d + 4 * i) = r;
        }
    }
}
// finish of the extended form: partial [chunks][3d] -> dgamma, dbeta, colsum written straight to their destinations
__global__ __launch_bounds__(256) void ln_bwd_finish3_kernel(const float* __restrict__ partial, int chunks, int d,
                                                             float* __restrict__ o0, float* __restrict__ o1,
                                                             float* __restrict__ o2) {
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + tx, n = 3 * d;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (j < n) {
        const float* p = partial + j;
        int c = ty;
        for (; c + 48 < chunks; c += 64) {
            a0 += p[(size_t)c * n]; a1 += p[(size_t)(c + 16) * n]; a2 += p[(size_t)(c + 32) * n];
            a3 += p[(size_t)(c + 48) * n];
        }
        for (; c < chunks; c += 16) a0 += p[(size_t)c * n];
    }
    red[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && j < n) {
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) r += red[i][tx];
        float* o = j < d ? o0 : (j < 2 * d ? o1 : o2);
        if (o) o[j % d] = r;
    }
}
// workgroups per group: enough to fill the chip at the kernel's 4 waves per SIMD (1 024 workgroups of 4 waves = every
// wave slot of 256 CUs; at the former cap of 512 the d = 768 stream kernel ran at half occupancy: 102 us for 450 MB)
static int ln_bwd_chunks(int rows_per_group, int groups) {
    int chunks = (1024 + groups - 1) / groups;
    const int maxc = (rows_per_group + 15) / 16;
    if (chunks > maxc) chunks = maxc;
    if (chunks > 1024) chunks = 1024;
    return chunks < 1 ? 1 : chunks;
}
size_t ln_bwd_scratch_bytes(int rows_per_group, int groups, int d) {
    return (size_t)groups * ln_bwd_chunks(rows_per_group, groups) * 3 * d * 4;   // 3: the stream form's extra column sum
}
// The residual-stream form (one group): dx_accum += dx; dx_bf16 = bf16(dx_accum); dgamma / dbeta / colsum(dx_accum)
// (the last may be null) written to their own destinations.  scratch: ln_bwd_scratch_bytes(rows, 1, d) * 3 / 2.
hipError_t launch_ln_bwd_stream(const void* dy, bool dy_bf16, const void* x, bool x_bf16, const float* gamma, float* dx_accum,
                                void* dx_bf16, float* dgamma, float* dbeta, float* colsum_or_null, float* scratch, int rows,
                                int d, hipStream_t s) {
    if (d % 4 || d > 2048 || !dx_accum || !dx_bf16 || !scratch || (x_bf16 && !dy_bf16)) return hipErrorInvalidValue;
    const int ch = (d / 4 + 63) / 64;
    const int chunks = ln_bwd_chunks(rows, 1);
    const int rpc = (rows + chunks - 1) / chunks;
    dim3 grid(chunks, 1), block(256);
#define LNB_CASE(C)                                                                                              \
    case C:                                                                                                      \
        if (x_bf16)                                                                                              \
            hipLaunchKernelGGL((ln_bwd_kernel<C, true, true, true>), grid, block, 0, s, (const float*)dy, (const float*)x, gamma,  \
                               dx_accum, scratch, d, rows, rpc, chunks, (bf16*)dx_bf16);                         \
        else if (dy_bf16)                                                                                        \
            hipLaunchKernelGGL((ln_bwd_kernel<C, true, true>), grid, block, 0, s, (const float*)dy, (const float*)x, gamma,       \
                               dx_accum, scratch, d, rows, rpc, chunks, (bf16*)dx_bf16);                         \
        else                                                                                                     \
            hipLaunchKernelGGL((ln_bwd_kernel<C, true>), grid, block, 0, s, (const float*)dy, (const float*)x, gamma, dx_accum,   \
                               scratch, d, rows, rpc, chunks, (bf16*)dx_bf16);                                   \
        break;
    switch (ch) {
        LNB_CASE(1) LNB_CASE(2) LNB_CASE(3) LNB_CASE(4) LNB_CASE(5) LNB_CASE(6) LNB_CASE(7) LNB_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef LNB_CASE
    hipLaunchKernelGGL(ln_bwd_finish3_kernel, dim3((3 * d + 15) / 16), dim3(256), 0, s, scratch, chunks, d, dgamma, dbeta,
                       colsum_or_null);
    return hipGetLastError();
}
// dgb_out fp32 [groups, 2d] = [dgamma | dbeta] per group (may be null together with scratch: dx only)
hipError_t launch_ln_bwd(const float* dy, const float* x, const float* gamma, float* dx_accum, float* dgb_out,
                         float* scratch, int rows_per_group, int groups, int d, hipStream_t s) {
    if (d % 4 || d > 2048) return hipErrorInvalidValue;
    const int ch = (d / 4 + 63) / 64;
    const int chunks = ln_bwd_chunks(rows_per_group, groups);
    const int rpc = (rows_per_group + chunks - 1) / chunks;
    float* partial = dgb_out ? scratch : nullptr;
    dim3 grid(chunks, groups), block(256);
#define LNB_CASE(C)                                                                                              \
    case C:                                                                                                      \
        hipLaunchKernelGGL((ln_bwd_kernel<C>), grid, block, 0, s, dy, x, gamma, dx_accum, partial, d,            \
                           rows_per_group, rpc, chunks);                                                         \
        break;
    switch (ch) {
        LNB_CASE(1) LNB_CASE(2) LNB_CASE(3) LNB_CASE(4) LNB_CASE(5) LNB_CASE(6) LNB_CASE(7) LNB_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef LNB_CASE
    if (dgb_out)
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * d + 15) / 16, groups), dim3(256), 0, s, scratch, chunks,
                           (size_t)(2 * d), dgb_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Gated MLP elementwise.  `pre` bf16 [M, 2F] holds fc1 | gate pre-activations interleaved in blocks of 16
// (columns [32q, 32q+16) = fc1 outputs 16q.., [32q+16, 32q+32) = gate outputs 16q..: the GEMM's packed row order);
// act bf16 [M, F] = gelu_erf(a) * sigmoid(g)        (src/components/DiT.py:152-154)
// backward:  da = dact * sigmoid(g) * gelu'(a),  dg = dact * gelu(a) * sigmoid(g) * (1 - sigmoid(g)).
// One thread = 8 fc1 + 8 gate values.
// ---------------------------------------------------------------------------------------------------------
DITTO_DEV void unpack8(const u32x4 v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_lo(v[i]); f[2 * i + 1] = bf16_hi(v[i]); }
}
DITTO_DEV u32x4 pack8(const float* f) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    return o;
}
// da, dg of one element.  erf by the forward epilogue's rational form z P(z^2) / Q(z^2) (common.h fast_gelu_sigmoid2: 4.3e-7 in
// fp32), so that Phi(a) = (Q + z P) / (2 Q) costs one reciprocal, phi(a) one v_exp, sigmoid one v_exp + one v_rcp: ~35 vector
// instructions per element where erff + __expf + two IEEE divisions took ~90 — the kernel moves 1 GB per launch and was half
// VALU-bound next to it (211 us = 4.8 TB/s, round 3).
DITTO_DEV void gated_bwd_elem(float a, float g, float dy, float& da, float& dg) {
    const float z = __builtin_amdgcn_fmed3f(a * 0.70710678118654752440f, -4.0f, 4.0f);
    const float u = z * z;
    float pn = fmaf(u, 1.9217200275534196e-08f, -1.990321152334218e-06f);
    pn = fmaf(pn, u, 0.00015553680714219809f);
    pn = fmaf(pn, u, 0.0042930529452860355f);
    pn = fmaf(pn, u, 0.05243346840143204f);
    pn = fmaf(pn, u, 0.2139447033405304f);
    pn = fmaf(pn, u, 1.1283786296844482f);
    float qd = fmaf(u, 0.0010980380466207862f, 0.01555109117180109f);
    qd = fmaf(qd, u, 0.12079962342977524f);
    qd = fmaf(qd, u, 0.5229312181472778f);
    qd = fmaf(qd, u, 1.0f);
    const float cdf = fmaf(z, pn, qd) * (0.5f * fast_rcp(qd));                       // Phi(a)
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(a * a * -0.72134752044448170368f);   // phi(a)
    const float sg = fast_rcp(1.0f + __builtin_amdgcn_exp2f(g * -1.4426950408889634f));
    const float t = dy * sg;
    da = t * fmaf(a, pdf, cdf);
    dg = t * (a * cdf) * (1.0f - sg);
}

__global__ __launch_bounds__(256) void gated_bwd_kernel(const bf16* __restrict__ dact, const bf16* __restrict__ pre,
                                                        bf16* __restrict__ dpre, int M, int F) {
    const int per_row = F / 8;
    const size_t n = (size_t)M * per_row;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / per_row), j = (int)(i % per_row);
        const int q = j >> 1, e = j & 1;
        const size_t off = (size_t)m * 2 * F + 32 * q + 8 * e;
        float a[8], g[8], dy[8], da[8], dgt[8];
        unpack8(*reinterpret_cast<const u32x4*>(pre + off), a);
        unpack8(*reinterpret_cast<const u32x4*>(pre + off + 16), g);
        unpack8(*reinterpret_cast<const u32x4*>(dact + (size_t)m * F + 8 * j), dy);
#pragma unroll
        for (int k = 0; k < 8; ++k) gated_bwd_elem(a[k], g[k], dy[k], da[k], dgt[k]);
        *reinterpret_cast<u32x4*>(dpre + off) = pack8(da);
        *reinterpret_cast<u32x4*>(dpre + off + 16) = pack8(dgt);
    }
}
// The same with the column sums of dpre (= the fc1 | gate bias gradients, packed order) as per-chunk partial rows:
// block = 64 column groups x 4 row lanes, blockIdx.y = row chunk; partial[chunk][2F] is finished by reduce_partials_kernel.
__global__ __launch_bounds__(256) void gated_bwd_colsum_kernel(const bf16* __restrict__ dact, const bf16* __restrict__ pre,
                                                               bf16* __restrict__ dpre, float* __restrict__ partial, int M,
                                                               int F, int rows_per_chunk) {
    __shared__ float red[3][64][17];
    const int jl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jl;
    const int q = j >> 1, e = j & 1;
    const int rbeg = blockIdx.y * rows_per_chunk;
    const int rend = rbeg + rows_per_chunk < M ? rbeg + rows_per_chunk : M;
    float sa[8], sg8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sa[k] = 0.f; sg8[k] = 0.f; }
    for (int m = rbeg + rl; m < rend; m += 4) {
        const size_t off = (size_t)m * 2 * F + 32 * q + 8 * e;
        float a[8], g[8], dy[8], da[8], dgt[8];
        unpack8(*reinterpret_cast<const u32x4*>(pre + off), a);
        unpack8(*reinterpret_cast<const u32x4*>(pre + off + 16), g);
        unpack8(*reinterpret_cast<const u32x4*>(dact + (size_t)m * F + 8 * j), dy);
#pragma unroll
        for (int k = 0; k < 8; ++k) gated_bwd_elem(a[k], g[k], dy[k], da[k], dgt[k]);
        const u32x4 pa = pack8(da), pg = pack8(dgt);
        *reinterpret_cast<u32x4*>(dpre + off) = pa;
        *reinterpret_cast<u32x4*>(dpre + off + 16) = pg;
        float ra[8], rg[8];                              // sum what the wgrad GEMM will read: the bf16-rounded values
        unpack8(pa, ra); unpack8(pg, rg);
#pragma unroll
        for (int k = 0; k < 8; ++k) { sa[k] += ra[k]; sg8[k] += rg[k]; }
    }
    if (rl > 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[rl - 1][jl][k] = sa[k]; red[rl - 1][jl][8 + k] = sg8[k]; }
    }
    __syncthreads();
    if (rl == 0) {
        float* prow = partial + (size_t)blockIdx.y * 2 * F + 32 * q + 8 * e;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            prow[k] = (sa[k] + red[0][jl][k]) + (red[1][jl][k] + red[2][jl][k]);
            prow[16 + k] = (sg8[k] + red[0][jl][8 + k]) + (red[1][jl][8 + k] + red[2][jl][8 + k]);
        }
    }
}
static int ew_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}
// colsum_out fp32 [2F] (packed column order) with scratch >= 256 * 2F floats: the fused form; null: elementwise only
hipError_t launch_gated_bwd(const void* dact, const void* pre, void* dpre, int M, int F, hipStream_t s,
                            float* colsum_out, float* scratch) {
    if (F % 16) return hipErrorInvalidValue;
    if (colsum_out && scratch && (F / 8) % 64 == 0) {
        int chunks = 1536 / (F / 8 / 64);
        chunks = chunks > 256 ? 256 : chunks;
        const int maxc = (M + 15) / 16;
        chunks = chunks > maxc ? maxc : (chunks < 1 ? 1 : chunks);
        const int rpc = (M + chunks - 1) / chunks;
        hipLaunchKernelGGL(gated_bwd_colsum_kernel, dim3(F / 8 / 64, chunks), dim3(256), 0, s, (const bf16*)dact,
                           (const bf16*)pre, (bf16*)dpre, scratch, M, F, rpc);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * F + 15) / 16, 1), dim3(256), 0, s, scratch, chunks,
                           (size_t)(2 * F), colsum_out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(gated_bwd_kernel, dim3(ew_grid((size_t)M * F / 8)), dim3(256), 0, s, (const bf16*)dact,
                       (const bf16*)pre, (bf16*)dpre, M, F);
    if (colsum_out && scratch) return launch_colsum_bf16(dpre, 2 * F, M, 2 * F, colsum_out, scratch, s);
    return hipGetLastError();
}

// out fp32 [rows, cols] (ld = cols) <- rows of the packed order: out[r] = src[(r/blk)*(blk*mult) + r%blk + row_off]
__global__ __launch_bounds__(256) void unpack_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          int rows, int cols4, int blk, int mult, int row_off) {
    const size_t n = (size_t)rows * cols4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols4), c = (int)(i % cols4);
        const int pr = (r / blk) * (blk * mult) + r % blk + row_off;
        reinterpret_cast<f32x4*>(dst)[(size_t)r * cols4 + c] = reinterpret_cast<const f32x4*>(src)[(size_t)pr * cols4 + c];
    }
}
hipError_t launch_unpack_rows(const float* src, float* dst, int rows, int cols, int blk, int mult, int row_off,
                              hipStream_t s) {
    if (cols % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(ew_grid((size_t)rows * cols / 4)), dim3(256), 0, s, src, dst, rows,
                       cols / 4, blk, mult, row_off);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void unpack_vec_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                         int blk, int mult, int row_off) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < rows) dst[r] = src[(r / blk) * (blk * mult) + r % blk + row_off];
}
hipError_t launch_unpack_vec(const float* src, float* dst, int rows, int blk, int mult, int row_off, hipStream_t s) {
    hipLaunchKernelGGL(unpack_vec_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, src, dst, rows, blk, mult, row_off);
    return hipGetLastError();
}

// W fp32 [rows, cols] -> W^T bf16 [cols, ldT]: dst[c][map(r)] = src[r][c] with the packed row map of launch_pack_bf16
// (dgrad operand: dX = dY W runs as the forward GEMM with "weight" W^T, K-contiguous).  32x32 LDS tiles.
__global__ __launch_bounds__(256) void pack_bf16_t_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int rows,
                                                          int cols, int ldT, int blk, int mult, int row_off) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) {
            const int pr = (r / blk) * (blk * mult) + r % blk + row_off;
            dst[(size_t)c * ldT + pr] = (bf16)tile[tx][i];
        }
    }
}
hipError_t launch_pack_bf16_t(const float* src, void* dst, int rows, int cols, int ldT, int blk, int mult, int row_off,
                              hipStream_t s) {
    hipLaunchKernelGGL(pack_bf16_t_kernel, dim3((rows + 31) / 32, (cols + 31) / 32), dim3(256), 0, s, src, (bf16*)dst,
                       rows, cols, ldT, blk, mult, row_off);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Dropout of the cross-attention probabilities (nn.MultiheadAttention(dropout=0.1) in train mode,
// src/components/DiT.py:90-91 -> torch functional.py dropout(attn_weights, p)).  torch draws the mask from its
// Philox stream; this path draws it from a counter-based hash of (seed, layer, batch*head, query, key) so the
// backward kernels regenerate it instead of storing a [B,H,N,T] mask.  tests/ restate the same hash in numpy.
// ---------------------------------------------------------------------------------------------------------
// P bf16 [Sq, ld] = dropout(softmax(S * scale)) (first Skv columns; padding columns zero).  One wave per row.
// rows = a stack of (batch, head) score matrices, rows_per_bh each: row -> pair bh0 + row / rows_per_bh, query row % ..
__global__ __launch_bounds__(256) void softmax_drop_rows_kernel(const float* __restrict__ S, bf16* __restrict__ P, int Sq,
                                                                int Skv, int ld, float scale_log2, unsigned seed_lo,
                                                                unsigned seed_hi, int layer, int bh0, int rows_per_bh,
                                                                unsigned thr, float keep_scale) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Sq) return;
    const DropStream stream = thr ? drop_stream(seed_lo, seed_hi, layer, bh0 + row / rows_per_bh) : DropStream{};
    const int qi = row % rows_per_bh;
    const float* s = S + (size_t)row * ld;
    float m = -1e30f;
    for (int j = lane; j < Skv; j += 64) m = fmaxf(m, s[j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < Skv; j += 64) sum += __builtin_amdgcn_exp2f((s[j] - m) * scale_log2);
    const float inv = 1.0f / wave_sum(sum);
    bf16* pr = P + (size_t)row * ld;
    for (int j = lane; j < ld; j += 64) {
        float p = 0.f;
        if (j < Skv) {
            p = __builtin_amdgcn_exp2f((s[j] - m) * scale_log2) * inv;
            if (thr) p = drop_keep(stream, qi, j, thr) ? p * keep_scale : 0.f;
        }
        pr[j] = (bf16)p;
    }
}
// dS bf16 [Sq, ld] = P * (dP - sum_j P dP) * scale, with P = softmax(S * scale) recomputed and
// dP = dPd * mask * keep_scale (dPd fp32 [Sq, ld] = dO V^T).  Padding columns zero.
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ S, const float* __restrict__ dPd,
                                                               bf16* __restrict__ dS, int Sq, int Skv, int ld,
                                                               float scale, float scale_log2, unsigned seed_lo,
                                                               unsigned seed_hi, int layer, int bh0, int rows_per_bh,
                                                               unsigned thr, float keep_scale) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Sq) return;
    const DropStream stream = thr ? drop_stream(seed_lo, seed_hi, layer, bh0 + row / rows_per_bh) : DropStream{};
    const int qi = row % rows_per_bh;
    const float* s = S + (size_t)row * ld;
    const float* dp = dPd + (size_t)row * ld;
    float m = -1e30f;
    for (int j = lane; j < Skv; j += 64) m = fmaxf(m, s[j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < Skv; j += 64) sum += __builtin_amdgcn_exp2f((s[j] - m) * scale_log2);
    const float inv = 1.0f / wave_sum(sum);
    float dsum = 0.f;
    for (int j = lane; j < Skv; j += 64) {
        const float p = __builtin_amdgcn_exp2f((s[j] - m) * scale_log2) * inv;
        float g = dp[j];
        if (thr) g = drop_keep(stream, qi, j, thr) ? g * keep_scale : 0.f;
        dsum += p * g;
    }
    dsum = wave_sum(dsum);
    bf16* out = dS + (size_t)row * ld;
    for (int j = lane; j < ld; j += 64) {
        float v = 0.f;
        if (j < Skv) {
            const float p = __builtin_amdgcn_exp2f((s[j] - m) * scale_log2) * inv;
            float g = dp[j];
            if (thr) g = drop_keep(stream, qi, j, thr) ? g * keep_scale : 0.f;
            v = p * (g - dsum) * scale;
        }
        out[j] = (bf16)v;
    }
}
unsigned dropout_threshold(float p) {
    if (!(p > 0.f)) return 0u;
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 4294967295u : (unsigned)t;
}
// nrows = rows_per_bh * (number of stacked (batch, head) pairs starting at pair bh0)
hipError_t launch_softmax_drop_rows(const float* S, void* P, int nrows, int rows_per_bh, int Skv, int ld, float scale,
                                    uint64_t seed, int layer, int bh0, float p_drop, hipStream_t s) {
    const unsigned thr = dropout_threshold(p_drop);
    hipLaunchKernelGGL(softmax_drop_rows_kernel, dim3((nrows + 3) / 4), dim3(256), 0, s, S, (bf16*)P, nrows, Skv, ld,
                       scale * 1.4426950408889634f, (unsigned)(seed & 0xFFFFFFFFu), (unsigned)(seed >> 32), layer, bh0,
                       rows_per_bh, thr, thr ? 1.0f / (1.0f - p_drop) : 1.0f);
    return hipGetLastError();
}
hipError_t launch_softmax_bwd_rows(const float* S, const float* dPd, void* dS, int nrows, int rows_per_bh, int Skv, int ld,
                                   float scale, uint64_t seed, int layer, int bh0, float p_drop, hipStream_t s) {
    const unsigned thr = dropout_threshold(p_drop);
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((nrows + 3) / 4), dim3(256), 0, s, S, dPd, (bf16*)dS, nrows, Skv, ld,
                       scale, scale * 1.4426950408889634f, (unsigned)(seed & 0xFFFFFFFFu), (unsigned)(seed >> 32), layer,
                       bh0, rows_per_bh, thr, thr ? 1.0f / (1.0f - p_drop) : 1.0f);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// The [B, .] affine maps of the modulation vectors (time_embed, ada_ln.time_mlp, ada_ln.text_mlp:
// src/model/DiTTO.py:40-44, src/components/DiT.py:14-21), fp32, B <= a few hundred rows.
//   fwd    y[b,o]  = sum_i W[o,i] f(x[b,i]) + bias[o]                  f = SiLU or identity
//   bwd_w  dW[o,i] = sum_b dy[b,o] f(x[b,i]);   dbias[o] = sum_b dy[b,o]
//   bwd_x  dx[b,i] = f'(x[b,i]) * sum_o dy[b,o] W[o,i]
// ---------------------------------------------------------------------------------------------------------
DITTO_DEV float silu_grad(float x) {
    const float sg = sigmoid_f(x);
    return sg * (1.0f + x * (1.0f - sg));
}
__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int B, int I, int O, int silu_in) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (o >= O) return;
    float acc = 0.f;
    for (int i = lane; i < I; i += 64) {
        float v = x[(size_t)b * I + i];
        if (silu_in) v = silu_f(v);
        acc += W[(size_t)o * I + i] * v;
    }
    acc = wave_sum(acc);
    if (lane == 0) y[(size_t)b * O + o] = acc + (bias ? bias[o] : 0.f);
}
__global__ __launch_bounds__(256) void small_linear_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 float* __restrict__ dW, float* __restrict__ dbias, int B,
                                                                 int I, int O, int silu_in) {
    const int i = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
    if (i >= I) return;
    float acc = 0.f, bsum = 0.f;
    for (int b = 0; b < B; ++b) {
        float v = x[(size_t)b * I + i];
        if (silu_in) v = silu_f(v);
        const float g = dy[(size_t)b * O + o];
        acc += g * v;
        bsum += g;
    }
    dW[(size_t)o * I + i] = acc;
    if (i == 0 && dbias) dbias[o] = bsum;
}
__global__ __launch_bounds__(256) void small_linear_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                                 const float* __restrict__ x, float* __restrict__ dx, int B,
                                                                 int I, int O, int silu_in) {
    // block = 64 input columns x 4 slices of the O outputs, combined through LDS in a fixed order
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx, b = blockIdx.y;
    float acc = 0.f;
    if (i < I) {
        const int per = (O + 3) / 4, o0 = ty * per, o1 = o0 + per < O ? o0 + per : O;
        for (int o = o0; o < o1; ++o) acc += dy[(size_t)b * O + o] * W[(size_t)o * I + i];
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && i < I) {
        float r = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
        if (silu_in) r *= silu_grad(x[(size_t)b * I + i]);
        dx[(size_t)b * I + i] = r;
    }
}
hipError_t launch_small_linear_fwd(const float* x, const float* W, const float* bias, float* y, int B, int I, int O,
                                   bool silu_in, hipStream_t s) {
    hipLaunchKernelGGL(small_linear_fwd_kernel, dim3((O + 3) / 4, B), dim3(256), 0, s, x, W, bias, y, B, I, O,
                       silu_in ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_small_linear_bwd_w(const float* dy, const float* x, float* dW, float* dbias, int B, int I, int O,
                                     bool silu_in, hipStream_t s) {
    hipLaunchKernelGGL(small_linear_bwd_w_kernel, dim3((I + 255) / 256, O), dim3(256), 0, s, dy, x, dW, dbias, B, I, O,
                       silu_in ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_small_linear_bwd_x(const float* dy, const float* W, const float* x, float* dx, int B, int I, int O,
                                     bool silu_in, hipStream_t s) {
    hipLaunchKernelGGL(small_linear_bwd_x_kernel, dim3((I + 63) / 64, B), dim3(256), 0, s, dy, W, x, dx, B, I, O,
                       silu_in ? 1 : 0);
    return hipGetLastError();
}

// nn.Embedding backward: dtable[ids[b]] += drows[b], b in order (one workgroup: duplicates accumulate
// deterministically); dtable must be zeroed by the caller.
__global__ __launch_bounds__(256) void embedding_scatter_add_kernel(const float* __restrict__ drows,
                                                                    const int64_t* __restrict__ ids,
                                                                    float* __restrict__ dtable, int B, int V, int d) {
    for (int b = 0; b < B; ++b) {
        long long id = ids[b];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        for (int j = threadIdx.x; j < d; j += 256) dtable[(size_t)id * d + j] += drows[(size_t)b * d + j];
        __syncthreads();
    }
}
hipError_t launch_embedding_scatter_add(const float* drows, const int64_t* ids, float* dtable, int B, int V, int d,
                                        hipStream_t s) {
    hipLaunchKernelGGL(embedding_scatter_add_kernel, dim3(1), dim3(256), 0, s, drows, ids, dtable, B, V, d);
    return hipGetLastError();
}

}  // namespace ditto
