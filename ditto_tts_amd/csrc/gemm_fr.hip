// gemm_fr.hip — dispatcher of the FULL-ROW bf16 MFMA GEMMs (N = d = 768 or 1024) with the residual add and the next LayerNorm
// fused into their epilogue (reference src/components/DiT.py:148+152, :155 + the next block's :105):
//
//     h[M, N] (fp32 or bf16, in place) = residual + A[M, K] * W[N, K]^T + bias
//     u[M, N] (bf16 / fp8)             = LayerNorm(h) * gamma + beta   (eps 1e-5)
//
// A workgroup that owns WHOLE rows of the residual stream normalises them while they are still in registers: 24 of the 36
// LayerNorm launches of a step are gone.  Kernels (kernels.h fr_launch_kernel picks; all N = 768 forms give the same fp32 h bits):
//   gemm_frd.hip    128 x 768 tiles, one wave per SIMD, W straight from L2 into registers; the bf16 residual stream (default)
//   gemm_fr64.hip   64 x 768 tiles, two workgroups per CU (batches of 11 .. 15 utterances), and 64 x 1024 tiles (BASELINE C5)
//   (rounds 2-3 ran 128 x 768 tiles with the weights through an LDS ring, gemm_fr128: superseded by gemm_frd, deleted in round 6)
#include "gemm_common.h"

namespace ditto {

namespace {
constexpr int FM = 128, FN = 768;
}

bool gemm_fr_supports(int M, int N, int K, size_t lda, size_t ldw) {
    if (N != FN || K % 64 || K < 64 || M < 64) return false;   // 64 <= M < 128 runs the 64-row kernel (kernels.h fr_launch_kernel)
    if ((size_t)M * lda * 2 >= (1ull << 32) || (size_t)N * ldw * 2 >= (1ull << 32)) return false;
    return true;
}

hipError_t launch_gemm_fr(const GemmParams& p_in, const float* gamma, const float* beta, void* u_bf16, int ldu,
                          int rot_period, hipStream_t s, bool u_fp8, bool hb) {
    FrParams fp;
    fp.g = p_in;
    fp.g.flags = g_gemm_flags;
    fp.g.tiles_m = (p_in.M + FM - 1) / FM;
    fp.g.tiles_n = 1;
    fp.gamma = gamma; fp.beta = beta; fp.u = (bf16*)u_bf16; fp.ldu = ldu;
    fp.rot_period = g_fr_rot ? rot_period : 0;
    fp.stagger_ticks = 0;
    fp.u_fp8 = u_fp8;
    fp.hb = hb;
    if (hb && (p_in.N != FN || fr_launch_kernel(p_in.M, p_in.K) != 130)) return hipErrorInvalidValue;   // gemm_frd.hip only
    if (p_in.N == 1024) return launch_gemm_fr64(fp, s);          // d = 1024: 64 x 1024 tiles, one workgroup per CU
    if (u_fp8) return hipErrorInvalidValue;
    // which N = 768 kernel (kernels.h fr_launch_kernel): all produce the same h bits, a speed rule only
    const int kern = fr_launch_kernel(p_in.M, p_in.K);
    if (kern == 130) return launch_gemm_frd(fp, s);
    if (kern == 64) {
        fp.stagger_ticks = g_fr_stagger;
        return launch_gemm_fr64(fp, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace ditto
