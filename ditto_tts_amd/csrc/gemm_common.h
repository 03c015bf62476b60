// gemm_common.h — parameters and fused epilogues shared by the two GEMM tile structures
// (gemm.hip: 128x128 two-barrier; gemm256.hip: 256x256 eight-phase).
//
// Both structures compute C^T with swapped MFMA operands (weights as the instruction's A operand), so a
// lane holds, for ONE output row, 4 consecutive columns of each of the 4 16-column blocks of its wave's
// 64-column span:  acc[n][e] = C[row][cbase + 16*n + 4*fq + e],  fq = lane >> 4.
#pragma once
#include "common.h"
#include "kernels.h"

namespace ditto {

struct GemmParams {
    const bf16* A; int lda;
    const bf16* W; int ldw; int w_rows;
    const float* bias;
    const float* residual; int ldr;
    void* out; int ldo;
    bf16* out2; int ldo2;
    const float* rope_cos; const float* rope_sin; int rope_rpb; int rope_cols;
    int M, N, K;
    int tiles_m, tiles_n;
    int tile_stride;   // gemm256: persistent workgroups walk tiles b, b + stride, ...
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Bias of the lane's 4 x 4 columns, loaded ONCE per wave (the same for every row of the tile): keeps the row
// loop free of dependent global loads.
DITTO_DEV void load_bias(const GemmParams& p, int cbase, int fq, f32x4 (&b)[4]) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int c = cbase + n * 16 + fq * 4;
        b[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && c < p.N) b[n] = *reinterpret_cast<const f32x4*>(p.bias + c);
    }
}

// One output row x the wave's 64-column span.  `row` < M is checked by the caller.
template <int EPI>
DITTO_DEV void epilogue_row(const GemmParams& p, int row, int cbase, const f32x4 (&acc)[4], const f32x4 (&bias)[4],
                            int fq) {
    const int c4 = fq * 4;
    if constexpr (EPI == EPI_GATED) {
        // packed columns: 16 x fc1 | 16 x gate | 16 x fc1 | 16 x gate  (reference src/components/DiT.py:153-155)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int pc = cbase + pr * 32 + c4;
            if (pc >= p.N) continue;
            const f32x4 b1 = bias[2 * pr], bg = bias[2 * pr + 1];
            const f32x4 h = acc[2 * pr], g = acc[2 * pr + 1];
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fast_gelu_erf(h[e] + b1[e]) * fast_sigmoid(g[e] + bg[e]);
            const int oc = cbase / 2 + pr * 16 + c4;
            u32x2 st;
            st[0] = pack_bf16x2(o[0], o[1]);
            st[1] = pack_bf16x2(o[2], o[3]);
            *reinterpret_cast<u32x2*>((bf16*)p.out + (size_t)row * p.ldo + oc) = st;
        }
    } else if constexpr (EPI == EPI_QKV_ROPE) {
        float v[4][4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[n][e] = acc[n][e] + bias[n][e];
        }
        if (cbase < p.rope_cols) {  // a q or k head (width 64): half-split RoPE, pair (j, j+32); DiT.py:52-72
            const int pos = row % p.rope_rpb;
            float r[4][4];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const f32x4 cs = *reinterpret_cast<const f32x4*>(p.rope_cos + (size_t)pos * 32 + n * 16 + c4);
                const f32x4 sn = *reinterpret_cast<const f32x4*>(p.rope_sin + (size_t)pos * 32 + n * 16 + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = v[n][e], hi = v[n + 2][e];
                    r[n][e] = lo * cs[e] - hi * sn[e];      // t*cos + (-t[j+32])*sin
                    r[n + 2][e] = hi * cs[e] + lo * sn[e];  // t*cos + ( t[j-32])*sin
                }
            }
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[n][e] = r[n][e];
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int c = cbase + n * 16 + c4;
            if (c >= p.N) continue;
            u32x2 st;
            st[0] = pack_bf16x2(v[n][0], v[n][1]);
            st[1] = pack_bf16x2(v[n][2], v[n][3]);
            *reinterpret_cast<u32x2*>((bf16*)p.out + (size_t)row * p.ldo + c) = st;
        }
    } else {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int c = cbase + n * 16 + c4;
            if (c >= p.N) continue;
            f32x4 v = acc[n] + bias[n];
            if constexpr (EPI == EPI_BIAS_BF16) {
                u32x2 st;
                st[0] = pack_bf16x2(v[0], v[1]);
                st[1] = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<u32x2*>((bf16*)p.out + (size_t)row * p.ldo + c) = st;
            } else {
                if constexpr (EPI == EPI_BIAS_RES_F32) {
                    if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (size_t)row * p.ldr + c);
                }
                *reinterpret_cast<f32x4*>((float*)p.out + (size_t)row * p.ldo + c) = v;
                if constexpr (EPI == EPI_BIAS_RES_F32) {
                    if (p.out2) {
                        u32x2 st;
                        st[0] = pack_bf16x2(v[0], v[1]);
                        st[1] = pack_bf16x2(v[2], v[3]);
                        *reinterpret_cast<u32x2*>(p.out2 + (size_t)row * p.ldo2 + c) = st;
                    }
                }
            }
        }
    }
}

// gemm256.hip
hipError_t launch_gemm256(const GemmParams& p, GemmEpilogue epi, hipStream_t s);

}  // namespace ditto
