// gemm_fr.hip — FULL-ROW bf16 MFMA GEMM for the N = d = 768 projections of a DiT block, with the fp32 residual add
// AND the next LayerNorm fused into its epilogue (gfx950).
//
//     h[M, 768] (fp32, in place) = residual + A[M, K] * W[768, K]^T + bias            (reference DiT.py:148, :155)
//     u[M, 768] (bf16)           = LayerNorm(h) * gamma + beta   (eps 1e-5)           (reference DiT.py:152, :105 of
//                                                                                      the next block)
//
// Why: a workgroup that owns WHOLE rows of the residual stream can normalise them while they are still in registers.
// The separate LayerNorm launch (28 us: 100 MB of fp32 h read back + 50 MB written, at its HBM roofline) disappears for
// the two LayerNorms that follow a GEMM (24 of the 36 per step), and the out-projection stops re-reading its A panel
// once per column tile.
//
//   tile      128 rows x 768 columns (all of N), one tile per CU at M = 32768.  256 threads = 4 waves, ONE wave per
//             SIMD with the whole 512-entry register file (launch_bounds(256, 1)): wave (wm, wn) owns rows
//             [64 wm, +64) x columns [384 wn, +384) = 4 x 24 accumulators of v_mfma_f32_16x16x32_bf16 = 384 registers.
//             (At two waves per SIMD the 192 accumulators of a half-size wave tile leave no room for fragments.)
//   LDS       two stages of 56 KiB (K = 32: A 128 rows x 64 B, W 768 rows x 64 B) + 2 KiB for the row statistics.
//             16-B chunk c of row r at c ^ (-(r>>2) & 3) (conflict-free ds_read_b128, as gemm_o3 / gemm_pp).
//   stage g   { s_barrier (buffer of stage g-1 is free) ; 14 LDS-DMA pieces of stage g+1 interleaved with the MFMAs ;
//               for n in 0..23: W fragment n (prefetched 3 ahead) x the 4 resident A fragments -> 4 MFMAs ;
//               at n = 20: vmcnt(0) ; s_barrier ; prefetch stage g+1's A fragments and first W fragments }
//             96 MFMAs (1536 cycles) per stage per wave, 28 ds_read_b128; with one wave per SIMD all latency hiding
//             is in-wave: fragment reads run 3 ahead, the DMA one stage ahead, the next stage's first fragments are
//             read under the last 16 MFMAs.
//   epilogue  v = acc + bias + residual (in the accumulators); row sums -> lanes of the row (2 shuffles) -> the two
//             column halves through LDS; mean; sum (v-mean)^2 the same way; h stored fp32 (nt), u = LN(v) stored bf16
//             with the widened 16-B row store (gemm_common.h).  Two-pass statistics, like nn.LayerNorm.
#include "gemm_common.h"

namespace ditto {

namespace {

constexpr int FM = 128, FN = 768, FK = 32;
constexpr int F_A_BYTES = FM * FK * 2;            // 8 KiB
constexpr int F_W_BYTES = FN * FK * 2;            // 48 KiB
constexpr int F_STAGE = F_A_BYTES + F_W_BYTES;    // 56 KiB
constexpr int F_RED = 2 * F_STAGE;                // row-statistics scratch: [2 passes][2 column halves][128 rows] fp32
constexpr int F_LDS = F_RED + 2 * 2 * FM * 4;     // 114 KiB + 2 KiB

#define FR_BAR() asm volatile("s_barrier" ::: "memory")

struct FrParams {
    GemmParams g;
    const float* gamma; const float* beta;   // LayerNorm affine of the fused norm (null: no LayerNorm output)
    bf16* u; int ldu;                         // LayerNorm output
};

// MFMA with the accumulator's register file chosen by the SOURCE: hipcc, given 384 accumulators as plain values,
// shuttled them between the two halves of the 512-entry file (1 591 v_accvgpr moves and 540 scratch accesses per 96
// MFMAs).  As asm statements with an "a" (AGPR) or "v" (VGPR) tied operand the placement is fixed: column blocks
// 0..NA-1 of every row block live in AGPRs, the rest in VGPRs, and the statements keep their program order.
constexpr int NA = 15;   // 4 x 15 x 4 = 240 AGPRs; 4 x 9 x 4 = 144 VGPRs
DITTO_DEV void mfma_a(f32x4& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
}
DITTO_DEV void mfma_v(f32x4& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(a));
}

template <bool LN>
__global__ __launch_bounds__(256, 1) void gemm_fr_kernel(FrParams fp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemmParams& p = fp.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int ntiles = p.tiles_m;
    const int nkt = p.K / FK;
    const int stride = gridDim.x;

    // ---- DMA addressing: a stage = 56 pieces of 1 KiB (16 rows x 64 B): 0..7 = A, 8..55 = W; wave w moves pieces
    //      14w .. 14w+13 (wave 0: the 8 A pieces + 6 W pieces).  NO per-piece address registers: a W piece's address is
    //      (W + K-step + piece * 16 rows) [scalar] + ONE per-lane offset (row-in-piece, swizzled chunk); an A piece's
    //      offset is recomputed when issued (row clamp of the tail tile).  (14 per-piece offsets were spilled to scratch
    //      by hipcc and every reload's vmcnt(0) serialised the DMA stream: 4.6 us per stage instead of 0.8.) ----
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int prow = lane >> 2, cpos = lane & 3;
    const int pc = cpos ^ ((0 - (prow >> 2)) & 3);          // (16 k + prow) >> 2 & 3 == prow >> 2 & 3
    const unsigned vw = (unsigned)(((size_t)prow * p.ldw + pc * 8) * 2);
    int i_tile = blockIdx.x, i_kt = 0;       // issue cursor: (tile, K-step) of the next stage to issue
    unsigned i_buf = 0;                       // LDS byte offset of the buffer it goes to
    auto issue_piece = [&](int i) {           // piece i (0..13) of the stage at the issue cursor
        const int piece = wid * 14 + i;       // wave-uniform
        if (piece < 8) {
            int ar = i_tile * FM + piece * 16 + prow;
            ar = ar < p.M ? ar : p.M - 1;
            const unsigned va = (unsigned)(((size_t)ar * p.lda + pc * 8) * 2);
            glds16_so(va, (const char*)p.A + (size_t)i_kt * (FK * 2), lds_base + i_buf + (unsigned)(piece * 1024));
        } else {
            glds16_so(vw, (const char*)p.W + (size_t)i_kt * (FK * 2) + (size_t)(piece - 8) * 16 * p.ldw * 2,
                      lds_base + i_buf + (unsigned)(piece * 1024));
        }
    };
    auto advance_issue = [&]() {
        i_buf ^= (unsigned)F_STAGE;           // the two buffers sit at 0 and F_STAGE
        if (++i_kt == nkt) {
            i_kt = 0;
            i_tile += stride;
        }
    };

    // ---- fragment addressing ----
    const int frow = lane & 15, fq = lane >> 4;
    const int coff = (fq ^ ((0 - (frow >> 2)) & 3)) << 4;
    const int a_off = (wm * 64 + frow) * 64 + coff;                       // + m * 16 * 64
    const int w_off = F_A_BYTES + (wn * 384 + frow) * 64 + coff;           // + n * 16 * 64

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
#pragma unroll
    for (int i = 0; i < 14; ++i) issue_piece(i);   // stage 0 -> buffer 0
    advance_issue();
    unsigned c_buf = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FR_BAR();

    bf16x8 af[4], wfr[4];
    {
        const char* cur = smem + c_buf;
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const bf16x8*>(cur + a_off + m * 16 * 64);
#pragma unroll
        for (int n = 0; n < 3; ++n) wfr[n] = *reinterpret_cast<const bf16x8*>(cur + w_off + n * 16 * 64);
    }

    f32x4 acca[4][NA], accv[4][24 - NA];
    for (; tile < ntiles; tile += stride) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int n = 0; n < NA; ++n) acca[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 24 - NA; ++n) accv[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        for (int kt = 0; kt < nkt; ++kt) {
            // every wave has retired its reads of the other buffer (stage g-1): it may be overwritten
            FR_BAR();
            const bool do_issue = i_tile < ntiles;
            const char* cur = smem + c_buf;
            bf16x8 afn[4];
#pragma unroll
            for (int n = 0; n < 24; ++n) {
                if (n + 3 < 24) wfr[(n + 3) & 3] = *reinterpret_cast<const bf16x8*>(cur + w_off + (n + 3) * 16 * 64);
                if (n == 20) {
                    // stage g+1 has landed (this wave's pieces: vmcnt; everyone's: barrier): read its A fragments and
                    // first W fragments now, under the last 16 MFMAs of this stage
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    FR_BAR();
                    const char* nb = smem + (c_buf == 0 ? (unsigned)F_STAGE : 0u);
#pragma unroll
                    for (int m = 0; m < 4; ++m) afn[m] = *reinterpret_cast<const bf16x8*>(nb + a_off + m * 16 * 64);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (n < NA) mfma_a(acca[m][n], wfr[n & 3], af[m]);
                    else mfma_v(accv[m][n - NA], wfr[n & 3], af[m]);
                }
                if (n < 14 && do_issue) issue_piece(n);
                if (n >= 21) {   // W fragments 0..2 of the next stage, into the ring slots the last MFMAs have released
                    const char* nb = smem + (c_buf == 0 ? (unsigned)F_STAGE : 0u);
                    wfr[(n - 21) & 3] = *reinterpret_cast<const bf16x8*>(nb + w_off + (n - 21) * 16 * 64);
                }
            }
            if (do_issue) advance_issue();
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = afn[m];
            c_buf = c_buf == 0 ? (unsigned)F_STAGE : 0u;
        }

        // ---------------- epilogue: bias + residual, LayerNorm statistics, stores ----------------
        // The accumulators stay in their home registers (AGPR / VGPR); each pass pulls a value out, works on it and —
        // where it changed — pins it back (PIN_A / PIN_V: an empty asm with a tied operand of the home class), so hipcc
        // does not try to keep all 384 updated values in VGPRs (that spilled 600 registers).
#define PIN_A(x) asm volatile("" : "+a"(x))
#define PIN_V(x) asm volatile("" : "+v"(x))
        const int m0 = tile * FM;
        const int rbase = m0 + wm * 64 + frow;          // + 16 m
        const int cbase = wn * 384 + fq * 4;             // + 16 n
        float s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < 24; ++n) {
            const f32x4 b4 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + cbase + 16 * n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 r = {0.f, 0.f, 0.f, 0.f};
                if (p.residual && rbase + 16 * m < p.M)
                    r = *reinterpret_cast<const f32x4*>(p.residual + (size_t)(rbase + 16 * m) * p.ldr + cbase + 16 * n);
                if (n < NA) {
                    f32x4 v = acca[m][n < NA ? n : 0] + b4 + r;
                    s1[m] += (v[0] + v[1]) + (v[2] + v[3]);
                    acca[m][n < NA ? n : 0] = v;
                    PIN_A(acca[m][n < NA ? n : 0]);
                } else {
                    f32x4 v = accv[m][n < NA ? 0 : n - NA] + b4 + r;
                    s1[m] += (v[0] + v[1]) + (v[2] + v[3]);
                    accv[m][n < NA ? 0 : n - NA] = v;
                    PIN_V(accv[m][n < NA ? 0 : n - NA]);
                }
            }
        }
        float mean[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (LN) {
            float* red = reinterpret_cast<float*>(smem + F_RED);     // [pass][wn][128]
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float t = s1[m];
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                if (fq == 0) red[wn * FM + wm * 64 + 16 * m + frow] = t;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int r = wm * 64 + 16 * m + frow;
                mean[m] = (red[r] + red[FM + r]) * (1.0f / FN);
            }
            float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 24; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f32x4 v = n < NA ? acca[m][n < NA ? n : 0] : accv[m][n < NA ? 0 : n - NA];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dl = v[e] - mean[m];
                        q[m] = fmaf(dl, dl, q[m]);
                    }
                }
            float* red2 = red + 2 * FM;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float t = q[m];
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                if (fq == 0) red2[wn * FM + wm * 64 + 16 * m + frow] = t;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int r = wm * 64 + 16 * m + frow;
                rstd[m] = rsqrtf((red2[r] + red2[FM + r]) * (1.0f / FN) + 1e-5f);
            }
        }
        // stores: h fp32 (non-temporal: its next reader is a LayerNorm / GEMM epilogue far away), u = LN(h) bf16 and the
        // optional bf16 copy of h (last layer: the proj_out operand), two 16-column blocks per 16-byte row store
#pragma unroll
        for (int n = 0; n < 24; n += 2) {
            f32x4 g0, g1, e0, e1;
            if constexpr (LN) {
                g0 = *reinterpret_cast<const f32x4*>(fp.gamma + cbase + 16 * n);
                g1 = *reinterpret_cast<const f32x4*>(fp.gamma + cbase + 16 * (n + 1));
                e0 = *reinterpret_cast<const f32x4*>(fp.beta + cbase + 16 * n);
                e1 = *reinterpret_cast<const f32x4*>(fp.beta + cbase + 16 * (n + 1));
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 v0 = n < NA ? acca[m][n < NA ? n : 0] : accv[m][n < NA ? 0 : n - NA];
                const f32x4 v1 = n + 1 < NA ? acca[m][n + 1 < NA ? n + 1 : 0] : accv[m][n + 1 < NA ? 0 : n + 1 - NA];
                const int row = rbase + 16 * m;
                const bool ok = row < p.M;
                if (ok) {
                    float* hp = (float*)p.out + (size_t)row * p.ldo + cbase + 16 * n;
                    store16<true, true>(hp, __builtin_bit_cast(u32x4, v0), 0);
                    store16<true, true>(hp + 16, __builtin_bit_cast(u32x4, v1), 0);
                }
                if (p.out2) {
                    u32x2 pa, pb;
                    pa[0] = pack_bf16x2(v0[0], v0[1]); pa[1] = pack_bf16x2(v0[2], v0[3]);
                    pb[0] = pack_bf16x2(v1[0], v1[1]); pb[1] = pack_bf16x2(v1[2], v1[3]);
                    store_bf16_pair<false>(p.out2 + (size_t)(ok ? row : 0) * p.ldo2, wn * 384 + 16 * n, pa, pb, fq, ok ? FN : 0);
                }
                if constexpr (LN) {
                    const f32x4 y0 = (v0 - mean[m]) * rstd[m] * g0 + e0;
                    const f32x4 y1 = (v1 - mean[m]) * rstd[m] * g1 + e1;
                    u32x2 pa, pb;
                    pa[0] = pack_bf16x2(y0[0], y0[1]); pa[1] = pack_bf16x2(y0[2], y0[3]);
                    pb[0] = pack_bf16x2(y1[0], y1[1]); pb[1] = pack_bf16x2(y1[2], y1[3]);
                    store_bf16_pair<false>(fp.u + (size_t)(ok ? row : 0) * fp.ldu, wn * 384 + 16 * n, pa, pb, fq, ok ? FN : 0);
                }
            }
        }
#undef PIN_A
#undef PIN_V
    }
}

template <bool LN>
hipError_t launch_fr_t(const FrParams& fp, int grid, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fr_kernel<LN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_fr_kernel<LN>), dim3(grid), dim3(256), F_LDS, s, fp);
    return hipGetLastError();
}

}  // namespace

bool gemm_fr_supports(int M, int N, int K, size_t lda, size_t ldw) {
    if (N != FN || K % FK || M < FM) return false;
    if ((size_t)M * lda * 2 >= (1ull << 32) || (size_t)N * ldw * 2 >= (1ull << 32)) return false;
    return true;
}

hipError_t launch_gemm_fr(const GemmParams& p_in, const float* gamma, const float* beta, void* u_bf16, int ldu,
                          hipStream_t s) {
    static int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    FrParams fp;
    fp.g = p_in;
    fp.g.tiles_m = (p_in.M + FM - 1) / FM;
    fp.g.tiles_n = 1;
    fp.gamma = gamma; fp.beta = beta; fp.u = (bf16*)u_bf16; fp.ldu = ldu;
    const int grid = fp.g.tiles_m < n_cu ? fp.g.tiles_m : n_cu;
    return (gamma && u_bf16) ? launch_fr_t<true>(fp, grid, s) : launch_fr_t<false>(fp, grid, s);
}

}  // namespace ditto
