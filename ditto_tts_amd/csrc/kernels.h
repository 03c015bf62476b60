// kernels.h — internal launch interface between the C-ABI (ditto_api.hip) and the kernel files.
// Every launcher enqueues on `s` and returns hipGetLastError(); none allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ditto {

// ---------------- rowwise.hip : HBM-bound row / elementwise kernels ----------------
// LayerNorm over the last dim (eps 1e-5), one wave per row.
//   gamma/beta: affine (both or neither).  out_bf16 [M, ldo] bf16.
hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta, void* out_bf16, int ldo,
                            int M, int d, hipStream_t s);
// same, output fp8 e4m3 (saturating), [M, ldo] bytes
hipError_t launch_layernorm_fp8(const float* x, const float* gamma, const float* beta, void* out_fp8, int ldo,
                                int M, int d, hipStream_t s);
// per-row quantisation fp32 [rows, cols] -> fp8 e4m3 with scale[r] = amax(row)/448; row map as launch_pack_bf16
hipError_t launch_pack_fp8(const float* src, void* dst_fp8, float* scales, int rows, int cols, int dst_ld, int blk,
                           int mult, int row_off, hipStream_t s);
// GlobalAdaLN apply: h = LN_noaffine(x) * (1 + ttab[t[b]][j] + tmod[b][j]) + (ttab[t[b]][d+j] + tmod[b][d+j]);
// also writes bf16(x) (the RAW input) to raw_bf16 [M, ldraw] for the fused proj_in.
// h_bf16: h_out is BF16 [M, d] (the bf16 residual stream).  g1 / be1 / u1_bf16 (all or none): also write block 0's norm1,
// u1 bf16 [M, d] = LN(h) * g1 + be1 from the fp32 row in registers (the LayerNorm launch in front of block 0 disappears).
hipError_t launch_adaln(const float* x, const float* ttab, const float* tmod, const int64_t* t, int steps,
                        float* h_out, void* raw_bf16, int ldraw, int B, int N, int d, hipStream_t s, bool h_bf16 = false,
                        const float* g1 = nullptr, const float* be1 = nullptr, void* u1_bf16 = nullptr);
// LayerNorm whose INPUT rows are bf16 [M, d] (the bf16 residual stream)
hipError_t launch_layernorm_xbf16(const void* x_bf16, const float* gamma, const float* beta, void* out_bf16, int ldo, int M,
                                  int d, hipStream_t s);
hipError_t launch_cast_bf16(const float* src, void* dst_bf16, size_t n, hipStream_t s);
// out = residual + bias + sum of `nsplit` fp32 partial products (contiguous [M, N], `stride` elements apart), in order
// gamma / beta / u_bf16 (all or none): also u bf16 [M, ldu] = LayerNorm(out) * gamma + beta in the same launch (one wave per row;
// the same h and u bits as this launch + launch_layernorm)
hipError_t launch_splitk_finish(const float* partial, int nsplit, size_t stride, const float* bias,
                                const float* residual, float* out, void* out2_bf16, int ldo2, int M, int N,
                                hipStream_t s, const float* gamma = nullptr, const float* beta = nullptr,
                                void* u_bf16 = nullptr, int ldu = 0);
hipError_t launch_p_sample_update(float* x, const float* eps, const float* noise, const int64_t* t,
                                  const float* betas, const float* alphas, const float* acp, int B,
                                  size_t elems_per_utt, hipStream_t s);
// Philox4x32-10 + Box-Muller N(0,1) keyed by per-utterance seeds (rowwise.hip): out[b, i] = f(seeds[b], step, i)
hipError_t launch_noise_normal(float* out, const int64_t* seeds, unsigned step, int B, size_t elems_per_utt, hipStream_t s);
hipError_t launch_p_sample_update_seeded(float* x, const float* eps, const int64_t* seeds, unsigned step, const int64_t* t,
                                         const float* betas, const float* alphas, const float* acp, int B,
                                         size_t elems_per_utt, hipStream_t s);
hipError_t launch_q_sample(const float* x0, const float* noise, const int64_t* t, const float* buffer, float* out,
                           int B, size_t elems_per_utt, hipStream_t s);
// ttab[step, 0:2d] = time_mlp(time_embed(t_embedding[step]))
hipError_t launch_time_table(const float* emb, const float* w0, const float* b0, const float* w2, const float* b2,
                             const float* wt, const float* bt, float* ttab, int steps, int td, int d, hipStream_t s);
// pooled[b, :] = mean_T text[b, :, :]   then   tmod[b, 0:2d] = Wx * silu(pooled[b]) + bx
hipError_t launch_text_mod(const float* text, const float* wx, const float* bx, float* pooled_scratch, float* tmod,
                           int B, int T, int dt, int d, hipStream_t s);
hipError_t launch_apply_rope_f32(const float* pos, const float* x, float* out, int B, int N, int H, int dh,
                                 hipStream_t s);
hipError_t launch_rope_tables(const float* inv_freq, float* cos_out, float* sin_out, int N, int half, hipStream_t s);
// dst[(r/blk)*(blk*mult) + r%blk + row_off][col_off + c] = bf16(src[r][c])
hipError_t launch_pack_bf16(const float* src, void* dst_bf16, int rows, int cols, int dst_ld, int col_off, int blk,
                            int mult, int row_off, hipStream_t s, float scale = 1.0f);
hipError_t launch_scale_vec(float* v, int n, float f, hipStream_t s);
// head padding (head_dim % 64 != 0): heads of width dh laid out at a stride of dhp = roundup(dh, 64), pads zero (rowwise.hip)
hipError_t launch_pack_bf16_headrows(const float* src, void* dst, int rows, int cols, int dst_ld, int dh, int dhp,
                                     int dst_row_off, hipStream_t s, float scale = 1.0f);
hipError_t launch_pack_bf16_headcols(const float* src, void* dst, int rows, int cols, int dst_ld, int dh, int dhp, hipStream_t s);
hipError_t launch_pack_vec_heads(const float* src, float* dst, int n, int dh, int dhp, int dst_off, hipStream_t s, float scale = 1.0f);
// h fp32 [M, ldh][:, hd * dh + c] += o bf16 [M, ldo][:, hd * dhp + c]
hipError_t launch_head_compact_add(const void* o_bf16, int ldo, float* h, int ldh, int M, int d, int dh, int dhp, hipStream_t s);
// fp32 [N, K] -> bf16 stage-major [K/16][N][16] (the full-row GEMM's weight layout, gemm_fr.hip)
hipError_t launch_pack_bf16_stage_major(const float* src, void* dst, int N, int K, hipStream_t s);
hipError_t launch_repack_bf16_stage_major(const void* src_bf16, void* dst, int N, int K, hipStream_t s, int group = 16);
// gemm_lnq.hip: q bf16 [M, ldo] = LayerNorm(h; gamma, beta) W_q^T + bias in ONE launch, d = 768 (the LayerNorm output never
// reaches HBM).  h fp32 or bf16 rows; Wp = stage-major image of W_q for the MFMA `shape`: 32 -> group 16, 16 -> group 32.
hipError_t launch_gemm_lnq(const void* h, int ldh, bool h_bf16, const float* gamma, const float* beta, const void* Wp,
                           const float* bias, void* out_bf16, int ldo, int M, int d, int shape, int rot_period, hipStream_t s);
extern int g_attn64p_min_wgs, g_attn64p_min_wgs_plain;   // attention.hip
extern int g_lnq_waves;  // gemm_lnq.hip: waves per workgroup of the d = 768 / shape-32 kernel: 4 (one per SIMD) or 8 (two per SIMD)
extern int g_lnq_ring;   // gemm_lnq.hip: depth of the W register ring in stages (0 = default: 4 for shape 32, 2 for shape 16; 8 / 4 = the deep rings)
// The full-row kernel runs ONE 128-row tile per workgroup, so it needs enough rows to fill the chip: measured in the
// model (tools/step_ab.py --batch b, C2 shapes, fr_mask 3 against 0) it loses below 160 tiles (B = 1: 2.92 vs 1.84 ms per
// step, B = 8: 4.73 vs 4.02, B = 16: 7.00 vs 6.65) and wins from there on (B = 20: 8.37 vs 8.43, B = 24: 9.49 vs 10.17,
// B = 32: 12.1 vs 13.0).  Like the choice of GEMM tile structure this rule depends on the number of rows in the launch: an
// utterance's bits are independent of its batch neighbours WITHIN a class of batch sizes, not across (the full-row kernel
// sums over k in a rotated order and in 32x32x16 steps).  A caller that splits one batch over launches or GPUs pins the
// class of the WHOLE batch for all of them: ditto_set_option("fr_class_rows", rows of the unsplit batch) makes every
// launch decide as that batch would (dist.sample_sharded and SpeechGenerator's seeds= path do it), so that sharding
// changes no bit; 0 (default) = decide on the launch's own rows.  fr_mask 0 gives one class outright.
extern int g_fr_class_rows;   // gemm.hip: the PROCESS default; every reader goes through opt_class_rows()
// Per-call options (include/ditto_hip.h ditto_call_opts, ABI 9).  The class pin, the residual-stream type, the fused full-row
// launches and the fused norm2 + q-projection decide WHICH BITS an utterance gets, so they are a property of a CALL, not of the
// process: an entry point that takes a ditto_call_opts (or runs inside ditto_call_opts_push / _pop of its thread) sees them in this
// THREAD-LOCAL record; -1 = "the process default" (ditto_set_option).  One C-ABI call runs entirely on its calling thread (it only
// enqueues), so two threads with different pins do not meet.
struct CallOpts { int class_rows, resid_bf16, fr_mask, lnq; };
extern thread_local CallOpts t_opts;       // gemm.hip
extern int g_fr_mask, g_resid_bf16, g_lnq;
inline int opt_class_rows() { return t_opts.class_rows >= 0 ? t_opts.class_rows : g_fr_class_rows; }
inline int opt_fr_mask() { return t_opts.fr_mask >= 0 ? t_opts.fr_mask : g_fr_mask; }
inline int opt_resid_bf16() { return t_opts.resid_bf16 >= 0 ? t_opts.resid_bf16 : g_resid_bf16; }
inline int opt_lnq() { return t_opts.lnq >= 0 ? t_opts.lnq : g_lnq; }
// Which N = 768 full-row kernel a batch of `rows` rows takes, 0 = none (the tiled GEMMs + LayerNorm launches).  Measured in the
// model at C2 shapes, tools/step_ab.py --batch b with the class pinned (ms per step, unfused / 64-row / 128-row direct):
//   b = 10: 4.68 / 4.85 / 5.18   11: 5.78 / 5.34 / 5.65   12: 5.78 / 5.37   13: 6.15 / 5.73 / 6.01   14: 6.73 / 6.19 / 6.55
//   15: 6.99 / 6.52 / 6.85   16: 6.43 / 6.46   17: 7.73 / 8.12 / 7.47   18: 7.93 / 8.34 / 7.68   19: 8.30 / 8.75 / 8.16   >= 20: direct
// i.e. the 128-row kernel (gemm_frd.hip) from 136 of its tiles on; below that the 64-row kernel (gemm_fr64.hip, two workgroups
// per CU) from 176 of ITS tiles on — except where the unfused N = d GEMMs' 256 x 192 tiles make exactly one round of the 256
// CUs (16 x 1024 rows; the rule below excludes every row count that rounds up to those 64 tiles of 256 rows, 16129 .. 16384),
// which is the one place in that range where they are not quantised away.
// THREE kernel classes result (what "an utterance's bits do not depend on its batch" is relative to):
//   low-latency  <= 2048 rows: tiled GEMMs, fc2 / final projection / cross out-projection split over K by a K-only rule, fp32 stream
//                (ll_mask bit 0 — fc2's finish also writes the next norm1 — changes no bit; bit 1 — the out-projection as TWO
//                K-splits — changes the summation order of that GEMM: it is part of what defines the class)
//   tiled        up to 17 407 rows: tiled GEMMs + LayerNorm launches, fp32 stream; from 8 192 rows on (round 5, "lnq_min_rows")
//                norm2 is fused into the q-projection (gemm_lnq.hip, two waves per SIMD); inside it 11 264 .. 17 407 rows (176 .. 271 tiles
//                of 64 rows, except the exact 256-row rounds) take the 64-row full-row kernel — same fp32 h bits as the 128-row
//                kernel's fp32 form (u within a bf16 rounding tie), still the fp32 stream
//   full-row     >= 17 408 rows (136 tiles of 128 rows): gemm_frd.hip for both fused launches, norm2 fused into the q-projection and,
//                with "residual_bf16" (default), the BF16 residual stream — h is rounded to bf16 three times per block, the fused
//                LayerNorms (and the AdaLN kernel's block-0 norm1) normalise the UNROUNDED fp32 row they hold in registers, while
//                the A/B path of gemm_flags 32768 (norm1 as its own launch) reads the bf16-rounded h: a bf16 tie apart
// A caller pins a class through the rows it passes (ditto_call_opts.class_rows / "fr_class_rows"): "full-row or not", and inside
// the full-row range also WHICH full-row kernel and with it the stream type.
extern int g_fr_tile;
inline int fr_rule_rows(int rows) {
    const int t128 = (rows + 127) / 128, t64 = (rows + 63) / 64;
    if (t128 >= 136) return 130;
    if (t64 >= 176 && ((rows + 255) / 256) * 4 != 256) return 64;
    return 0;
}
inline bool fr_pays(int M) { return fr_rule_rows(opt_class_rows() > 0 ? opt_class_rows() : M) != 0; }

// gemm.hip's tile rule for narrow outputs (N < 2048, not a multiple of 192): the 256 x 256 persistent kernel where its tiles make
// whole rounds of the 256 CUs.  Also read by fr_fc2_ok (gemm_common.h): at d = 1024 that GEMM + a LayerNorm launch beats the
// 64-row full-row fc2 (116.9 + 18 against 157.7 us at M = 16384: the full-row kernel re-streams all 8 MiB of W per 64 rows).
inline bool gemm256_whole_rounds(int M, int N, int K) {
    if (N >= 2048 || N % 256 || N % 192 == 0 || K < 1024) return false;
    const long t256 = (long)((M + 255) / 256) * (N / 256);
    return t256 >= 256 && t256 % 256 == 0;
}

// d = 1024 runs the 64-row kernel only (gemm_fr64.hip, one workgroup per CU): it needs three quarters of the CUs busy.
inline bool fr_pays_64(int M) {
    const int rows = opt_class_rows() > 0 ? opt_class_rows() : M;
    return (rows + 63) / 64 >= 192;
}

extern int g_train_flags;   // gemm.hip: training-step A/B switches: bit 0 = the rotation's backward as its own pass (not in the dq / dk epilogues), bit 1 = fc2 dgrad and gated backward as two launches, bit 2 = training forward's gated GEMM on the general epilogue
extern int g_fr_dgrad;   // gemm.hip: training backward, long-K dgrads on the full-row kernel: bit 0 fc1|gate (K = 8d), bit 1 QKV (K = 3d)
extern int g_fr_rot;     // gemm.hip: full-row kernel's K-loop rotation: 0 off, 1 on in the model (period = tiles per utterance), > 1 = period for ditto_gemm_ln_bf16 too
// g_fr_tile (declared above): gemm.hip: full-row kernel's tile: 0 = rule, 64 = gemm_fr64.hip for every launch with K <= g_fr64_maxk, 128 = gemm_fr.hip
// g_fr64_maxk (declared below): gemm.hip: longest K that takes the 64-row kernel when fr_tile = 64
extern int g_fr_u_fp8;   // gemm.hip: test hook (ditto_set_option("fr_u_fp8")): ditto_gemm_ln_bf16 writes the LayerNorm output as fp8
extern int g_fr_stagger; // gemm.hip: gemm_fr64's start delay of the second workgroup of a CU (10 ns ticks)
extern int g_fr64_maxk;
// The kernel a full-row LAUNCH of M rows and depth K runs on: 130 = gemm_frd.hip (W straight into registers: the N = 768
// kernel since round 3; in-model fc2 + norm1 162.5 -> 157.0 us against gemm_fr.hip), 64 = gemm_fr64.hip, 128 = gemm_fr.hip.
// fr_tile 0 = the rule of fr_rule_rows on the class rows (a launch outside the rule, e.g. ditto_gemm_ln_bf16 on a small M, takes
// the direct kernel); fr_tile 64 / 128 / 130 force one (64 only up to K = fr64_maxk).  The 128-row kernels need M >= 128.
inline int fr_launch_kernel(int M, int K) {
    int k = g_fr_tile;
    if (k == 0) {
        k = fr_rule_rows(opt_class_rows() > 0 ? opt_class_rows() : M);
        if (k == 0) k = 130;
    } else if (k == 64 && K > g_fr64_maxk) {
        k = 130;
    }
    if (k != 64 && M < 128) k = 64;
    return k;
}
// g_fr_mask (declared above): gemm.hip: 1 = cross out-proj + LayerNorm3, 2 = fc2 + next block's LayerNorm1 on the full-row kernel
// same row map for an fp32 vector (bias)
hipError_t launch_pack_vec(const float* src, float* dst, int rows, int blk, int mult, int row_off, hipStream_t s);
hipError_t launch_add_vec(const float* a, const float* b, float* dst, int n, hipStream_t s);
hipError_t launch_fill_i64(int64_t* dst, int n, int64_t value, hipStream_t s);

// ---------------- gemm.hip : bf16 MFMA GEMM family  out = A[M,K] * W[N,K]^T + bias ... ----------------
enum GemmEpilogue {
    EPI_BIAS_BF16 = 0,    // out bf16 [M, ldo] = acc + bias
    EPI_BIAS_RES_F32 = 1, // out fp32 [M, ldo] = acc + bias + residual (may alias out); optional bf16 copy
    EPI_QKV_ROPE = 2,     // out bf16 [M, ldo]; columns < 2*d_model get half-split RoPE (head_dim 64 only)
    EPI_GATED = 3,        // W rows interleaved [fc1 x16 | gate x16]; out bf16 [M, N/2] = gelu(a) * sigmoid(g)
    EPI_BIAS_F32 = 4,     // out fp32 [M, ldo] = acc + bias
    EPI_GATED_FP8 = 5,    // as EPI_GATED, out fp8 e4m3 [M, N/2] (A operand of an fp8 fc2)
    EPI_BIAS_RELU_BF16 = 6, // out bf16 [M, ldo] = max(acc + bias, 0)  (nn.TransformerDecoderLayer linear1, slp.hip)
    // Training backward of the gated MLP (src/components/DiT.py:152-154) as the epilogue of the fc2 dgrad GEMM: acc = dact
    // [M, N] = dY W2; out bf16 [M, 2N] = [da | dg] in the packed order of the pre-activations `pre_bf16` [M, 2N] (16 x fc1 | 16 x
    // gate); the bf16-rounded column sums (= the fc1 | gate bias gradients) as one partial row per 128-row half tile in
    // `colsum_partial` [2 * ceil(M / 256), 2N] fp32 (the caller sums the rows in order).  256 x 256 kernel only: N % 256 == 0.
    EPI_GATED_BWD = 7,
    // EPI_GATED that ALWAYS writes the pre-activations (out2_bf16 [M, N], packed order) as well: the training forward, as
    // its own instantiation so that it takes the straight-line epilogue the inference kernel has.  256 x 256 kernel only.
    EPI_GATED_PRE = 8
};
struct GemmArgs {
    const void* A; int lda;          // bf16 [M, K], row stride lda (elements)
    const void* W; int ldw;          // bf16 [N, K], K-contiguous rows (nn.Linear.weight layout); ldw = row stride, 0 -> K
    int w_rows;                      // valid rows of W (rows >= w_rows are clamped, their outputs are don't-care); 0 -> N
    const float* bias;               // fp32 [N] (may be null)
    const float* residual; int ldr;  // fp32 [M, ldr] (EPI_BIAS_RES_F32)
    void* out; int ldo;              // bf16 or fp32
    void* out2_bf16; int ldo2;       // optional bf16 copy of the fp32 result (EPI_BIAS_RES_F32)
    const float* rope_cos; const float* rope_sin; int rope_rows_per_batch; int rope_cols;  // EPI_QKV_ROPE
    const float* rope_freq_rev;      // optional, fp32 [32] = inv_freq / (2 pi): angles computed in the epilogue, tables unused
    int M, N, K;
    // fp8 (OCP e4m3) operands: A and W are 1-byte elements, acc is multiplied by wscale[n] (per-output-row weight
    // scale, fp32 [N]) before the bias.  Runs the persistent 256x256 kernel with v_mfma_scale_f32_16x16x128_f8f6f4.
    bool fp8; const float* wscale;
    // split-K (EPI_BIAS_F32 without bias, 128x128 structure): split s of k_splits multiplies K range
    // [s, s+1) * K / k_splits and writes its partial product to out + s * split_stride (fp32 elements); the caller
    // sums the slices (launch_reduce_partials).  The long-K, few-tile wgrad GEMMs of the backward pass.
    int k_splits; size_t split_stride;
    // batched (128x128 structures only): grid.y = batch_outer * batch_inner; batch z = (zo, zi) offsets A / W / out /
    // residual by zo * s?[0] + zi * s?[1] ELEMENTS of their own type.  (Per-(batch, head) attention products of the
    // generic-head_dim path in one launch.)
    int batch_outer, batch_inner;
    long long sA[2], sW[2], sO[2], sR[2];
    const void* pre_bf16; int ldpre;   // EPI_GATED_BWD: the forward's pre-activations, bf16 [M, ldpre]
    float* colsum_partial;             // EPI_GATED_BWD: fp32 [2 * ceil(M / 256), 2N]
};
// may the fc2 dgrad + gated backward of M rows, F = N act columns, run as ONE launch (EPI_GATED_BWD)?
bool gemm_gated_bwd_fused_ok(int M, int F);
hipError_t launch_gemm(const GemmArgs& a, GemmEpilogue epi, hipStream_t s);
extern int g_gemm_tile;  // 0 auto | 128 | 256
extern int g_gemm_flags; // GF_* (gemm_common.h)
extern int g_pp_mask;    // gemm.hip: GEMM classes that take the ping-pong kernel (ditto_set_option("pp_mask")), -1 = rule
extern int g_pp_nb;      // gemm_pp.hip A/B: 0 = rule, 3 / 4 = force 192- / 256-wide tiles (ditto_set_option("pp_nb"))
extern int g_pp_stagger; // gemm_pp.hip: forced phase offset (10 ns ticks), -1 = rule
extern int g_gemm_group; // > 0: forced super-column width of the tile order (A/B); 0 = pick_group_n's rule
extern int g_attn_flags; // attention.hip

// ---------------- around_loop.hip : steps either side of the loop ----------------
hipError_t launch_vq_argmin(const float* x, const float* codebook, float* cc_scratch, int64_t* idx, int R, int K, int D,
                            hipStream_t s);
hipError_t launch_embedding_gather(const float* table, const int64_t* ids, float* out, int n, int V, int d,
                                   hipStream_t s);
hipError_t launch_code_embed_mean(const float* table, const int64_t* codes, float* out, int B, int C, int F, int Fout,
                                  int V, int d, hipStream_t s);
hipError_t launch_linear_update(float* x, const float* eps, const float* z, const float* a, const float* ce,
                                const float* cz, int B, size_t elems_per_utt, hipStream_t s);
hipError_t launch_cfg_combine(const float* eps2, float* out, float w, size_t elems_half, hipStream_t s);

// ---------------- attention.hip ----------------
struct AttnArgs {
    const void* q; int ldq;   // bf16, head h at columns [h*dh, (h+1)*dh)
    const void* k; int ldk;
    const void* v; int ldv;
    void* out_bf16; int ldo;         // used when resid_f32 == nullptr
    float* resid_f32; int ldr;       // if set: resid[row, h*dh + c] = resid_in[...] + O   (self-attention, no out-proj)
    const float* resid_in;           // nullptr -> resid_f32 (in place); the training forward keeps both streams
    bool resid_bf16 = false;         // resid_f32 / resid_in point to BF16 [.., ldr] (the bf16 residual stream; fused dh == 64 path)
    int B, H, Sq, Skv, dh;
    float scale;                     // 1/sqrt(dh)
    void* workspace; size_t workspace_bytes;   // generic (dh != 64) path only
    // training forward: dropout on the probabilities (nn.MultiheadAttention train mode), mask = hash(seed, layer, ...)
    float dropout_p; uint64_t seed; int layer;
    bool force_generic;              // run the GEMM-composed path even at dh == 64
    float* lse_out;                  // fused (dh == 64) path: fp32 [B, H, Sq] log2-domain log-sum-exp for the backward
    bool q_prescaled;                // q already multiplied by scale * log2(e) (packed weights, dh == 64): `scale` unused
    bool causal;                     // key j visible to query i only if j <= i + (Skv - Sq); GEMM-composed path only
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);
size_t attention_workspace_bytes(int B, int H, int Sq, int Skv, int dh);
// the GEMM-composed path's scratch at any head_dim (force_generic / causal)
size_t attention_generic_workspace_bytes(int B, int H, int Sq, int Skv, int dh);
// slp.hip: y = LN(x) * gamma + beta as fp32 and / or bf16 (either may be null)
hipError_t launch_layernorm_dual(const float* x, const float* gamma, const float* beta, float* y_f32, void* y_bf16,
                                 int M, int d, hipStream_t s);
// in-place half-split RoPE on bf16 [M, ld] for `nheads` heads of width dh starting at column 0 (generic path);
// sin_sign = -1 applies the inverse rotation (the backward of RoPE)
// hstride > 0: heads of width dh sit at a stride of hstride columns (padded heads); ncols then counts PHYSICAL columns
hipError_t launch_rope_inplace(void* qk_bf16, int ld, const float* cosT, const float* sinT, int M,
                               int rows_per_batch, int ncols, int dh, hipStream_t s, float sin_sign = 1.0f, int hstride = 0);

// attention backward: dq/dk/dv bf16 (same head layout as q/k/v) from dO bf16; probabilities recomputed
struct AttnBwdArgs {
    const void* q; int ldq; const void* k; int ldk; const void* v; int ldv;
    const void* dout; int lddo;
    void* dq; int lddq; void* dk; int lddk; void* dv; int lddv;
    int B, H, Sq, Skv, dh;
    float scale;
    float dropout_p; uint64_t seed; int layer;
    void* workspace; size_t workspace_bytes;
    // fused path (dh == 64): the forward's log-sum-exp (log2 domain) and the forward output O, given either as bf16
    // (cross-attention) or as the residual stream after / before the segment (self-attention: O = after - before)
    const float* lse;
    const void* o_bf16; int ldo;
    const float* h_after; const float* h_before; int ldh;
    // fused path: the INVERSE half-split RoPE of dq and dk (the backward of the rotation the QKV GEMM's epilogue applied) in
    // the kernels' epilogues, on the fp32 accumulators; tables [Sq, dh / 2] (self-attention: Skv == Sq).  null: no rotation.
    const float* rope_cos; const float* rope_sin;
};
hipError_t launch_attention_bwd(const AttnBwdArgs& a, hipStream_t s);
// true when launch_attention_bwd(a) applies a.rope_cos / a.rope_sin itself (the fused dh == 64 path)
bool attention_bwd_fuses_rope(const AttnBwdArgs& a);
size_t attention_bwd_stats_bytes(int B, int H, int Sq);   // scratch for the per-tile {L, delta} records of the fused backward
hipError_t launch_attention_bwd64(const AttnBwdArgs& a, float* stats, hipStream_t s);
size_t attention_train_workspace_bytes(int B, int H, int Sq, int Skv, int dh);   // forward + backward scratch

// ---------------- gemm_tn.hip : out fp32 [Mo, No] = X[K, Mo]^T Y[K, No], both operands K-major (wgrad) ----------------
hipError_t launch_gemm_tn(const void* X, int ldx, const void* Y, int ldy, const void* zero256, float* out, int ldo,
                          int Mo, int No, int K, int k_splits, size_t split_stride, hipStream_t s, bool wide = false);

// ---------------- train.hip : backward-pass row / elementwise kernels ----------------
// optionally batched: nz_o * nz_i matrices, src of matrix (zo, zi) at + zo * s_o + zi * s_i, dst at + z * d_z (elements)
hipError_t launch_transpose_bf16(const void* src, int ld, int rows, int cols, void* dst, int ldT, hipStream_t s,
                                 int nz_o = 1, int nz_i = 1, long long s_o = 0, long long s_i = 0, long long d_z = 0);
hipError_t launch_reduce_partials(const float* partial, int chunks, size_t n, float* out, hipStream_t s);
hipError_t launch_colsum_f32(const float* x, int ld, int M, int n, float* out, float* scratch, hipStream_t s);
hipError_t launch_colsum_bf16(const void* x, int ld, int M, int n, float* out, float* scratch, hipStream_t s);
size_t ln_bwd_scratch_bytes(int rows_per_group, int groups, int d);
hipError_t launch_ln_bwd_stream(const void* dy, bool dy_bf16, const void* x, bool x_bf16, const float* gamma, float* dx_accum, void* dx_bf16,
                                float* dgamma, float* dbeta, float* colsum_or_null, float* scratch, int rows, int d,
                                hipStream_t s);
hipError_t launch_ln_bwd(const float* dy, const float* x, const float* gamma, float* dx_accum, float* dgb_out,
                         float* scratch, int rows_per_group, int groups, int d, hipStream_t s);
hipError_t launch_gated_bwd(const void* dact, const void* pre, void* dpre, int M, int F, hipStream_t s,
                            float* colsum_out = nullptr, float* scratch = nullptr);
hipError_t launch_unpack_rows(const float* src, float* dst, int rows, int cols, int blk, int mult, int row_off,
                              hipStream_t s);
hipError_t launch_unpack_vec(const float* src, float* dst, int rows, int blk, int mult, int row_off, hipStream_t s);
hipError_t launch_pack_bf16_t(const float* src, void* dst, int rows, int cols, int ldT, int blk, int mult, int row_off,
                              hipStream_t s);
unsigned dropout_threshold(float p);
hipError_t launch_softmax_drop_rows(const float* S, void* P, int nrows, int rows_per_bh, int Skv, int ld, float scale,
                                    uint64_t seed, int layer, int bh0, float p_drop, hipStream_t s);
hipError_t launch_softmax_bwd_rows(const float* S, const float* dPd, void* dS, int nrows, int rows_per_bh, int Skv, int ld,
                                   float scale, uint64_t seed, int layer, int bh0, float p_drop, hipStream_t s);
hipError_t launch_small_linear_fwd(const float* x, const float* W, const float* bias, float* y, int B, int I, int O,
                                   bool silu_in, hipStream_t s);
hipError_t launch_small_linear_bwd_w(const float* dy, const float* x, float* dW, float* dbias, int B, int I, int O,
                                     bool silu_in, hipStream_t s);
hipError_t launch_small_linear_bwd_x(const float* dy, const float* W, const float* x, float* dx, int B, int I, int O,
                                     bool silu_in, hipStream_t s);
hipError_t launch_embedding_scatter_add(const float* drows, const int64_t* ids, float* dtable, int B, int V, int d,
                                        hipStream_t s);

}  // namespace ditto
