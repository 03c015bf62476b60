/* ditto_hip.h — C-ABI of libditto_hip.so: the MI355X (gfx950) DiT denoise path.
 *
 * The reference (Tikai7/DiTTO-TTS) is pure Python with no FFI or operator registry;
 * its boundary for this path is the nn.Module surface.  This header
 * is the *new* boundary underneath that surface: each entry point names the reference
 * Python function (file:line, relative to the reference repo) whose arithmetic it
 * replaces.  The Python binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types, no exceptions.
 *  - Every function returns a ditto_status (0 = ok) and never aborts;
 *    ditto_last_error() returns a thread-local message for the last failure.
 *  - All tensor pointers are DEVICE pointers owned by the caller.  The library
 *    allocates no device memory and never synchronises: one call = one
 *    stream-ordered enqueue on `stream` (a hipStream_t; NULL = the default stream),
 *    so a caller may capture any call into a hipGraph.
 *  - Row-major, contiguous.  B = batch, N = latent length, T = text length,
 *    d = hidden_dim, H = num_heads, dh = d / H, L = num_layers, M = B*N.
 *  - fp32 in / fp32 out at the model boundary (what the reference's callers hold);
 *    bf16 operands with fp32 accumulation inside (MFMA).
 */
#ifndef DITTO_HIP_H
#define DITTO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DITTO_ABI_VERSION 10

typedef enum ditto_status {
    DITTO_OK = 0,
    DITTO_ERR_ARG = 1,        /* null pointer / bad enum */
    DITTO_ERR_SHAPE = 2,      /* unsupported shape (message says which constraint) */
    DITTO_ERR_HIP = 3,        /* a HIP runtime call failed */
    DITTO_ERR_SIZE = 4        /* caller-provided arena / workspace / cond buffer too small */
} ditto_status;

typedef void* ditto_stream_t;            /* hipStream_t */
typedef struct ditto_model* ditto_model_t;

/* Per-call options (ABI 9).  Four switches decide WHICH BITS an utterance gets — the kernel class its launch takes, the type of
 * the residual stream, the fused full-row launches, the fused norm2 + q-projection — so they belong to a CALL, not to the
 * process: the *_opts entry points take them as an argument, and ditto_call_opts_push / _pop put them in force for every call
 * the CALLING THREAD makes in between.  Each field: -1 = inherit (the thread's enclosing push scope, else the process default of
 * ditto_set_option).  A call runs entirely on its calling thread (it only enqueues), so threads with different options do not
 * interfere; handles stay immutable.
 *   class_rows    > 0: decide the kernel class as a batch of that many rows (B * N of the UNSPLIT batch) would — what a caller
 *                 that splits one batch over launches or GPUs passes so that sharding changes no bit ("fr_class_rows");
 *                 0: the launch's own rows.
 *   residual_bf16 0 / 1 ("residual_bf16").      fr_mask 0 .. 3 ("fr_mask").      lnq 0 / 16 / 32 ("lnq").
 *   reserved      must be zero. */
typedef struct ditto_call_opts {
    int32_t class_rows;
    int32_t residual_bf16;
    int32_t fr_mask;
    int32_t lnq;
    int32_t reserved[4];
} ditto_call_opts;

/* DiTTO.__init__ keyword arguments, reference src/model/DiTTO.py:10-19.
 * Shapes: hidden_dim % 64 == 0 and <= 2048; hidden_dim % num_heads == 0 with an EVEN head_dim (half-split RoPE).  head_dim 64
 * runs the fused attention kernels; any other multiple of 64 (e.g. the shipped 1-head / 768 config) the GEMM-composed attention;
 * head_dim % 64 != 0 (the reference takes it: src/components/DiT.py:78-86; e.g. 1152 / 16 = 72) runs forward-only on heads
 * PADDED to the next multiple of 64 inside the block — zero weight rows / columns, so no result changes; the arena, cond and
 * workspace sizes grow accordingly (always take them from the *_bytes queries); training and fp8 linears refuse such shapes. */
typedef struct ditto_config {
    int32_t hidden_dim;        /* d        (768)  */
    int32_t num_layers;        /* L        (12)   */
    int32_t num_heads;         /* H        (12)   */
    int32_t time_dim;          /*          (256)  */
    int32_t text_dim;          /* must equal hidden_dim: reference src/components/DiT.py:90-91 */
    int32_t diffusion_steps;   /* rows of t_embedding (1000; 50 for the benchmark loop) */
    int32_t flags;             /* DITTO_CFG_* */
} ditto_config;

/* QKV, fc1|gate and fc2 GEMMs with OCP fp8 (e4m3) operands on the MX-scaled MFMA (BASELINE config 5): weights
 * quantised per output row at ditto_model_create, activations by the producing LayerNorm / gated-MLP epilogue;
 * fp32 accumulation, bf16 attention, fp32 residual stream unchanged.  Needs hidden_dim % 128 == 0. */
#define DITTO_CFG_FP8_LINEAR 1

/* fp32 device pointers, named after the reference state_dict keys.
 * `blocks.i.attn.out_proj.*` and `blocks.i.rotary.inv_freq` are intentionally absent:
 * the reference never reads them on this path (src/components/DiT.py:134-139). */
typedef struct ditto_layer_weights {
    const float* norm1_weight;  const float* norm1_bias;                 /* blocks.i.norm1.*            [d]      */
    const float* attn_in_proj_weight;  const float* attn_in_proj_bias;   /* blocks.i.attn.in_proj_*     [3d,d],[3d] */
    const float* norm2_weight;  const float* norm2_bias;                 /* blocks.i.norm2.*                     */
    const float* cross_in_proj_weight; const float* cross_in_proj_bias;  /* blocks.i.cross_attn.in_proj_*        */
    const float* cross_out_proj_weight; const float* cross_out_proj_bias;/* blocks.i.cross_attn.out_proj.* [d,d],[d] */
    const float* norm3_weight;  const float* norm3_bias;                 /* blocks.i.norm3.*                     */
    const float* mlp_fc1_weight; const float* mlp_fc1_bias;              /* blocks.i.mlp_fc1.*          [4d,d],[4d] */
    const float* gate_weight;    const float* gate_bias;                 /* blocks.i.gate.*             [4d,d],[4d] */
    const float* mlp_fc2_weight; const float* mlp_fc2_bias;              /* blocks.i.mlp_fc2.*          [d,4d],[d]  */
} ditto_layer_weights;

typedef struct ditto_weights {
    const float* t_embedding_weight;                                     /* [steps, time_dim] */
    const float* time_embed_0_weight; const float* time_embed_0_bias;    /* [td,td],[td] */
    const float* time_embed_2_weight; const float* time_embed_2_bias;
    const float* ada_time_mlp_weight; const float* ada_time_mlp_bias;    /* ada_ln.time_mlp.1.* [2d,td],[2d] */
    const float* ada_text_mlp_weight; const float* ada_text_mlp_bias;    /* ada_ln.text_mlp.1.* [2d,dt],[2d] */
    const float* proj_in_weight;  const float* proj_in_bias;             /* [d,d],[d] */
    const float* proj_out_weight; const float* proj_out_bias;
    const float* rotary_inv_freq;                                        /* rotary.inv_freq [dh/2] */
    const ditto_layer_weights* layers;                                   /* HOST array [num_layers] */
} ditto_weights;

/* ---- library ---------------------------------------------------------------------- */
int         ditto_abi_version(void);
const char* ditto_last_error(void);

/* ---- sizes (host-only arithmetic, no GPU needed) ------------------------------------ */
/* bytes of device arena ditto_model_create packs the weights into */
size_t ditto_arena_bytes(const ditto_config* cfg);
/* bytes of the per-batch conditioning buffer: cached cross-attention K/V of every layer
 * [B*T, L*2d] bf16 + the text half of the AdaLN modulation [B, 2d] fp32 */
size_t ditto_cond_bytes(const ditto_config* cfg, int B, int T);
/* bytes of scratch ditto_forward / ditto_text_precompute need for (B, N, T) */
size_t ditto_workspace_bytes(const ditto_config* cfg, int B, int N, int T);

/* ---- model handle --------------------------------------------------------------------
 * Packs DiTTO's parameters (reference src/model/DiTTO.py:36-64, src/components/DiT.py:78-98)
 * for the kernels: bf16 weights, fused [W_fc1;W_gate] interleave, all layers' cross-attention
 * K/V projections concatenated, [W_proj_in | W_proj_out] concatenated along K, and the
 * timestep -> AdaLN (scale,shift) table  time_mlp(time_embed(t_embedding[t]))  for every t
 * (src/model/DiTTO.py:75-76 + src/components/DiT.py:30: depends on t and weights only).
 * The handle is immutable afterwards; concurrent calls on different streams are safe.
 * If EVERY model-level pointer of `w` (everything except `layers`) is NULL the handle is "blocks-only":
 * ditto_block_forward and ditto_text_precompute (K/V only) work, ditto_forward / ditto_rope_tables refuse. */
int ditto_model_create(const ditto_config* cfg, const ditto_weights* w, void* arena, size_t arena_bytes,
                       ditto_stream_t stream, ditto_model_t* out);
int ditto_model_destroy(ditto_model_t m);

/* RotaryEmbedding.forward (src/components/DiT.py:56-59): cos/sin of n*inv_freq[j],
 * cos_out/sin_out fp32 [N, dh/2] */
int ditto_rope_tables(ditto_model_t m, int N, float* cos_out, float* sin_out, ditto_stream_t stream);

/* Step-invariant text work, once per batch of utterances (the reference redoes it every step,
 * SURVEY.md App. B-9): the text half of GlobalAdaLN (src/components/DiT.py:27,31: mean-pool over T,
 * SiLU, Linear) and every layer's cross-attention K/V in-projection of text_emb
 * (src/components/DiT.py:144-148 -> torch F.multi_head_attention_forward packed in-projection).
 * text fp32 [B,T,text_dim] -> cond (ditto_cond_bytes). */
int ditto_text_precompute(ditto_model_t m, const float* text, int B, int T, void* cond, size_t cond_bytes,
                          void* workspace, size_t workspace_bytes, ditto_stream_t stream);

/* DiTTO.forward(x, text_emb, t)  (src/model/DiTTO.py:66-94) with text_emb given as `cond`.
 * x fp32 [B,N,d], t int64 [B] (device), rope_cos/sin from ditto_rope_tables(N), eps_out fp32 [B,N,d]. */
int ditto_forward(ditto_model_t m, const float* x, const void* cond, const int64_t* t, int B, int N, int T,
                  const float* rope_cos, const float* rope_sin, float* eps_out,
                  void* workspace, size_t workspace_bytes, ditto_stream_t stream);
/* the same call with its options as an argument (opts == NULL: every field inherits) */
int ditto_forward_opts(ditto_model_t m, const float* x, const void* cond, const int64_t* t, int B, int N, int T,
                       const float* rope_cos, const float* rope_sin, float* eps_out,
                       void* workspace, size_t workspace_bytes, ditto_stream_t stream, const ditto_call_opts* opts);
/* Thread-scoped form for every other entry point (ditto_block_forward*, the unit-test kernels, a caller's own helper that
 * makes several calls): `opts` are in force for the calls THIS THREAD makes until the matching pop.  Nests (depth <= 16);
 * fields at -1 inherit the enclosing scope.  ditto_call_opts_pop without a push is an error. */
int ditto_call_opts_push(const ditto_call_opts* opts);
int ditto_call_opts_pop(void);
/* what a call made by this thread NOW, without an opts argument, would run under: every field resolved (scope, else process
 * default), none -1 */
int ditto_call_opts_current(ditto_call_opts* out);

/* One DiT block, DiT.forward(x, text_emb, time_emb, rotary_pos) (src/components/DiT.py:100-157), in place on the
 * fp32 residual stream h [B,N,d].  `layer` selects the packed weights, `cond_layer` the K/V slice of `cond`
 * (a standalone block uses a 1-layer handle and cond_layer = 0).  time_emb is not an input: the reference's
 * block ignores it. */
int ditto_block_forward(ditto_model_t m, int layer, float* h, const void* cond, int cond_layer, int B, int N, int T,
                        const float* rope_cos, const float* rope_sin, void* workspace, size_t workspace_bytes,
                        ditto_stream_t stream);
/* The same block with taps for segment-wise parity (src/components/DiT.py:139 and :148): `tap_self` / `tap_cross`
 * (fp32 [B,N,d], either may be NULL) receive the residual stream after the self-attention segment and after the
 * cross-attention segment; h ends as after the gated MLP (:155). */
int ditto_block_forward_taps(ditto_model_t m, int layer, float* h, const void* cond, int cond_layer, int B, int N,
                             int T, const float* rope_cos, const float* rope_sin, float* tap_self, float* tap_cross,
                             void* workspace, size_t workspace_bytes, ditto_stream_t stream);

/* GlobalAdaLN.forward(x, time_emb, text_emb) as a standalone module (src/components/DiT.py:25-40): explicit
 * time_emb fp32 [B,time_dim] instead of a timestep; weights are the module's own fp32 device tensors
 * (time_mlp.1.*, text_mlp.1.*).  out fp32 [B,N,d]. */
size_t ditto_global_adaln_scratch_bytes(int B, int d, int time_dim, int text_dim);
int ditto_global_adaln(const float* x, const float* time_emb, const float* text_emb, const float* time_w,
                       const float* time_b, const float* text_w, const float* text_b, int B, int N, int T, int d,
                       int time_dim, int text_dim, float* out, void* scratch, size_t scratch_bytes,
                       ditto_stream_t stream);

/* RotaryEmbedding.apply_rope(pos, t) (src/components/DiT.py:61-72): pos fp32 [N,dh] ANGLES (the tensor
 * RotaryEmbedding.forward returns), t fp32 [B,N,H,dh] -> out (not in place). */
int ditto_apply_rope_f32(const float* pos, const float* t, float* out, int B, int N, int H, int dh,
                         ditto_stream_t stream);

/* SpeechGenerator.__p_sample's update (src/model/SpeechGenerator.py:137-145), in place on x:
 *   x <- (x - (1-alpha_t)/sqrt(1-acp_t) * eps)/sqrt(alpha_t) + [t>0]*sqrt(beta_t)*noise
 * betas/alphas/alphas_cumprod fp32 [steps] (device), t int64 [B] (device), noise may be NULL only
 * if every t == 0 is guaranteed by the caller (it is then never read). */
int ditto_p_sample_update(float* x, const float* eps, const float* noise, const int64_t* t,
                          const float* betas, const float* alphas, const float* alphas_cumprod,
                          int B, size_t elems_per_utt, ditto_stream_t stream);

/* One reverse-diffusion step = ditto_forward + ditto_p_sample_update (SpeechGenerator.__p_sample,
 * src/model/SpeechGenerator.py:130-147); eps scratch lives in the workspace. */
int ditto_p_sample(ditto_model_t m, float* x, const void* cond, const int64_t* t, const float* noise,
                   const float* betas, const float* alphas, const float* alphas_cumprod,
                   int B, int N, int T, const float* rope_cos, const float* rope_sin,
                   void* workspace, size_t workspace_bytes, ditto_stream_t stream);
int ditto_p_sample_opts(ditto_model_t m, float* x, const void* cond, const int64_t* t, const float* noise,
                        const float* betas, const float* alphas, const float* alphas_cumprod,
                        int B, int N, int T, const float* rope_cos, const float* rope_sin,
                        void* workspace, size_t workspace_bytes, ditto_stream_t stream, const ditto_call_opts* opts);

/* The sampling loop, SpeechGenerator.__sample_latents (src/model/SpeechGenerator.py:149-164), as ONE stream-ordered
 * enqueue: for t_val = t_begin, t_begin-1, ..., t_end:  x <- p_sample(x, full(B, t_val), cond).  `noise` holds the
 * N(0,1) draws of the steps in execution order, fp32 [t_begin - t_end + 1, B, N, d] (the caller's RNG: the library has
 * none; it may be NULL only when t_begin == t_end == 0).  t_scratch: int64 [B] device scratch the loop fills.
 * A caller that wants the whole loop in a hipGraph captures this single call. */
int ditto_denoise_steps(ditto_model_t m, float* x, const void* cond, int t_begin, int t_end, const float* noise,
                        const float* betas, const float* alphas, const float* alphas_cumprod, int B, int N, int T,
                        const float* rope_cos, const float* rope_sin, int64_t* t_scratch, void* workspace,
                        size_t workspace_bytes, ditto_stream_t stream);
int ditto_denoise_steps_opts(ditto_model_t m, float* x, const void* cond, int t_begin, int t_end, const float* noise,
                             const float* betas, const float* alphas, const float* alphas_cumprod, int B, int N, int T,
                             const float* rope_cos, const float* rope_sin, int64_t* t_scratch, void* workspace,
                             size_t workspace_bytes, ditto_stream_t stream, const ditto_call_opts* opts);

/* Per-utterance counter-based N(0,1) (no reference counterpart: the reference draws `torch.randn_like` from torch's
 * global generator, src/model/SpeechGenerator.py:141,154, a stream that cannot be sharded over GPUs).
 * out[b, i] = Philox4x32-10(key = seeds[b], counter = (i / 4, step, tag)) -> Box-Muller: a function of (seeds[b], step, i)
 * only, so an utterance's noise does not depend on its batch, its place in it or the GPU it runs on (SURVEY.md 8e:
 * per-utterance seeds make the sharded result independent of the world size).  seeds int64 [B] (device).
 * ditto_p_sample_seeded = ditto_p_sample with the step's noise generated inside the update kernel: bit-identical to
 * ditto_noise_normal(step) followed by ditto_p_sample(noise), without the noise buffer. */
int ditto_noise_normal(float* out, const int64_t* seeds, uint32_t step, int B, size_t elems_per_utt,
                       ditto_stream_t stream);
int ditto_p_sample_seeded(ditto_model_t m, float* x, const void* cond, const int64_t* t, const int64_t* seeds,
                          uint32_t step, const float* betas, const float* alphas, const float* alphas_cumprod, int B,
                          int N, int T, const float* rope_cos, const float* rope_sin, void* workspace,
                          size_t workspace_bytes, ditto_stream_t stream);
int ditto_p_sample_seeded_opts(ditto_model_t m, float* x, const void* cond, const int64_t* t, const int64_t* seeds,
                               uint32_t step, const float* betas, const float* alphas, const float* alphas_cumprod, int B,
                               int N, int T, const float* rope_cos, const float* rope_sin, void* workspace,
                               size_t workspace_bytes, ditto_stream_t stream, const ditto_call_opts* opts);

/* DiTTO.q_sample (src/model/DiTTO.py:106-126), bug-for-bug: `buffer` is the module's
 * `alphas_cumprod` buffer, which holds clipped betas.  out may alias x_start. */
int ditto_q_sample(const float* x_start, const float* noise, const int64_t* t, const float* buffer,
                   float* out, int B, size_t elems_per_utt, ditto_stream_t stream);

/* ---- single kernels, exported for unit parity tests ------------------------------------ */
/* nn.LayerNorm(d) eps=1e-5 (src/components/DiT.py:84,89,94,105,143,152): x fp32 [M,d] -> bf16 [M,d];
 * gamma/beta may both be NULL (elementwise_affine=False, src/components/DiT.py:23). */
int ditto_layernorm_bf16(const float* x, const float* gamma, const float* beta, void* out_bf16,
                         int M, int d, ditto_stream_t stream);

/* out[M,N] = A[M,K](bf16, row stride lda) * W[N,K]^T(bf16) + bias[N](fp32) — F.linear.
 * epilogue: 0 = bf16 out; 1 = fp32 out = acc + bias + residual (residual may alias out; may be NULL);
 *           4 = fp32 out = acc + bias;
 *           6 = bf16 out = relu(acc + bias)  (linear1 of nn.TransformerDecoderLayer, src/model/SpeechLP.py:23-28);
 *           3 = gated MLP (src/components/DiT.py:153-155): W/bias rows interleaved in blocks of 16
 *               [16 x mlp_fc1 | 16 x gate | ...], out bf16 [M, N/2] = gelu_erf(a) * sigmoid(g), ldo = N/2 or more. */
int ditto_gemm_bf16(const void* A, int lda, const void* W, const float* bias, const float* residual,
                    void* out, int ldo, int M, int N, int K, int epilogue, ditto_stream_t stream);

/* softmax(q k^T * scale) v per (batch, head), no mask (src/components/DiT.py:131-134;
 * torch functional.py MHA math).  q/k/v/out bf16, element strides given per row; head h occupies
 * columns [h*dh, (h+1)*dh) of each row; rows of batch b start at b*Sq (q,out) / b*Skv (k,v). */
int ditto_attention_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                         void* out, int ldo, int B, int H, int Sq, int Skv, int dh, float scale,
                         void* workspace, size_t workspace_bytes, ditto_stream_t stream);
/* scratch bytes ditto_attention_bf16 needs: 0 at dh = 64 (the fused kernels), the chunked score / probability / V^T arrays of the
 * GEMM-composed path otherwise */
size_t ditto_attention_workspace_bytes(int B, int H, int Sq, int Skv, int dh);

/* Full-row GEMM with the residual add and the FOLLOWING LayerNorm fused (N = 768: csrc/gemm_frd.hip, or its 64-row twin
 * csrc/gemm_fr64.hip under "fr_tile" 64 — 64 <= M < 128 always runs the 64-row twin, same fp32 bits;
 * N = 1024: csrc/gemm_fr64.hip; M >= 64 everywhere):
 *   out fp32 [M,N] = residual + A[M,K] W[N,K]^T + bias   (residual may alias out: the in-place stream update of
 *   src/components/DiT.py:148 / :155),   u bf16 [M,ldu] = LayerNorm(out) * gamma + beta  (eps 1e-5; the norm of :152 / the
 *   next block's :105).  gamma = beta = u = NULL: no LayerNorm output.  K % 64 == 0.
 *   W is NOT the nn.Linear image: it is packed stage-major, Wp[K/16][N][16] (Wp[s][n][j] = W[n][16 s + j]), so that
 *   each K-step's 24 / 32 KiB are contiguous (the kernel moves them by LDS-DMA in whole cache lines). */
int ditto_gemm_ln_bf16(const void* A, int lda, const void* W, const float* bias, const float* residual, float* out,
                       int ldo, const float* gamma, const float* beta, void* u_bf16, int ldu, int M, int N, int K,
                       ditto_stream_t stream);

/* LayerNorm fused INTO the GEMM that consumes it (csrc/gemm_lnq.hip; the model path's norm2 + cross-attention q-projection,
 * src/components/DiT.py:142-145; d = 768, or 1024 = BASELINE config C5):  out bf16 [M, ldo] = (LayerNorm(h) * gamma + beta)
 * W[d, d]^T + bias, eps 1e-5; the normalised rows live in the LDS only.  h: fp32 [M, ldh] or (h_is_bf16, d = 768) bf16 [M, ldh];
 * W: bf16, nn.Linear layout [d out, d in]; bias may be NULL.  mfma_shape 32 / 16 = v_mfma_f32_32x32x16_bf16 / 16x16x32 (two
 * builds of one kernel; d = 1024: 32 only).  w_scratch: d * d * 2 bytes, 256-byte aligned: receives the stage-major image of W the kernel streams (the model keeps
 * these images in its arena).  Any M >= 1. */
int ditto_gemm_lnq_bf16(const void* h, int ldh, int h_is_bf16, const float* gamma, const float* beta, const void* W,
                        const float* bias, void* out_bf16, int ldo, int M, int d, int mfma_shape, void* w_scratch,
                        ditto_stream_t stream);

/* Weight-gradient GEMM of the backward pass (csrc/gemm_tn.hip): out fp32 [Mo, No] = X[K, Mo]^T Y[K, No], both operands
 * K-major bf16 (rows = the contraction index, as activations and their gradients lie in memory), i.e. dW = dY^T X of
 * nn.Linear (what autograd computes for reference src/TrainDiTTO.py:90).  tile = 128 (128x128, two workgroups per CU) or
 * 256 (256x256, one per CU); k_splits > 1: the contraction is split, fp32 partial tiles are summed in slice order
 * (deterministic).  workspace: 256 + k_splits * Mo * No * 4 bytes (256 alone when k_splits <= 1), 256-byte aligned.
 * Mo, No, ldx, ldy multiples of 8. */
int ditto_gemm_tn_bf16(const void* X, int ldx, const void* Y, int ldy, float* out, int ldo, int Mo, int No, int K,
                       int k_splits, int tile, void* workspace, size_t workspace_bytes, ditto_stream_t stream);

/* Process-wide tuning switches (tests / experiments).  "gemm_tile": 0 = automatic choice between the three GEMM
 * tile structures, 128 (128x128) / 129 (persistent 256x128 ring) / 131 (128x256 ping-pong, two workgroups per CU) /
 * 256 (persistent 256x256) = force one.  Results are identical up to fp32 summation order.
 * "pp_mask": GEMM classes that take the ping-pong kernel (bit 0 cross q-proj, 1 cross out-proj, 2 final projection,
 * 3 fc2, 4 QKV + RoPE, 5 gated MLP; -1 = the built-in rule).
 * "fr_mask": N = d = 768 projections on the full-row kernel with the residual add and the following LayerNorm fused
 * (csrc/gemm_fr.hip): bit 0 = cross out-proj + norm3, bit 1 = fc2 + the next block's norm1.
 * "fr_class_rows": the full-row kernel sums over k in a different order than the tiled GEMMs, so an utterance's bits depend on
 * whether its launch took it — a function of the launch's row count (>= 136 tiles of 128 rows, or 176 .. 271 tiles of 64 rows
 * except the 16129 .. 16384 rows that round up to 64 tiles of 256: csrc/kernels.h fr_rule_rows, a measured rule).  A pinned
 * class that a launch cannot take (full-row kernels, fewer than 64 rows in the launch) is an error (DITTO_ERR_SHAPE), never a
 * silent change of class.  A caller that splits ONE batch
 * over several launches or GPUs sets this to the rows (B * N) of the unsplit batch: every launch then decides as that batch
 * would and sharding changes no bit (ditto_tts_amd/dist.py sample_sharded does).  0 (default) = each launch on its own rows.
 * "fr_tile": which N = 768 full-row kernel: 0 (default) / 130 = 128-row tiles with the weights fetched straight from L2 into
 * registers (csrc/gemm_frd.hip); 64 = 64-row tiles, two workgroups per CU (csrc/gemm_fr64.hip) for launches with
 * K <= "fr64_maxk".  Both produce the same fp32 result bit for bit; the LayerNorm output may differ in the last bf16 bit of a few
 * elements (other association of the row statistics).  "fr_stagger": start delay of a CU's second workgroup in the 64-row
 * kernel, 10 ns ticks.  (N = 1024 always runs csrc/gemm_fr64.hip.)
 * "fr_u_fp8": TEST HOOK: ditto_gemm_ln_bf16 at N = 1024 writes u as fp8 e4m3 bytes ([M, ldu] bytes), the form the fp8 linear
 * path's model forward uses for norm3.
 * "fr_dgrad": training backward, the long-K dgrads (N = d = 768) on the same kernel: bit 0 = fc1|gate (K = 8d), bit 1 = QKV.
 * "train_flags": training-step A/B switches: bit 0 = the backward of the self-attention rotation (RoPE) as its own pass over
 * dq | dk instead of inside the attention backward's dq / dk epilogues (head_dim 64; other head dims always take the pass);
 * bit 1 = the fc2 dgrad GEMM and the gated MLP's derivative as two launches instead of one (the derivative as the GEMM's
 * epilogue, from 144 tiles of 256 x 256 on); bit 2 = the training forward's gated GEMM on the general epilogue instead of its
 * own straight-line instantiation; bit 3 = the gradient wrt a LayerNorm's output as fp32 instead of bf16 between the dgrad GEMM
 * and the LayerNorm backward; bit 4 = the tape's residual-stream rows as fp32 instead of bf16 (the bf16 stream exists where the
 * inference forward has it: d = 768, head_dim 64, the full-row kernel class).  The flags must not change between the forward
 * and the backward of one step.
 * "fr_rot": that kernel's K-loop rotation (tiles start their k sum at different places so that the workgroups of an XCD do
 * not all ask the L2 for the same weight lines at once): 0 = off, 1 = on in the model path with period = row tiles per
 * utterance (an utterance's bits do not depend on its place in the batch), > 1 = ditto_gemm_ln_bf16 rotates too, with that
 * period in 128-row tiles.
 * "pp_nb": tile width of the ping-pong kernel's plain epilogues: 0 = rule, 3 = 128x192, 4 = 128x256.
 * "pp_stagger": phase offset between the two workgroups of a CU in the ping-pong GEMM, 10 ns ticks (-1 = built-in rule).
 * "gemm_flags": bit mask for kernel experiments (bit 0 = relaxed tile-start wait, default on; bits 1, 2 are
 * DIAGNOSTIC timing switches that skip stores / the epilogue and produce WRONG results — tools/ only; bit 14 = 16384: the
 * persistent 256x256 kernel takes the round-3 tile switch instead of running its K loop flat over it — A/B, bit-identical
 * results either way).
 * "gemm_group": forced super-column width of the GEMM tile order (A/B tool; 0 = the built-in rule, which a sweep of
 * 3 / 4 / 6 / 12 / 24 at C2 B = 32 did not beat).
 * "residual_bf16": 1 = the residual stream h between the segments of a block lives in HBM as bf16 (fp32 only inside the
 * accumulators and the LayerNorm statistics) for launches of the full-row class at d = 768 / head_dim 64; 0 = fp32 stream.
 * The sampler state x and eps stay fp32 either way.  (DITTO_RESIDUAL_BF16 in the environment sets the initial value.)
 * "lnq_waves": 8 (default) / 4 = waves per workgroup of the fused norm2 + q-projection (bit-identical).
 * "lnq": norm2 fused into the cross-attention q-projection (csrc/gemm_lnq.hip) for launches of the full-row class at d = 768:
 * 0 = LayerNorm launch + tiled GEMM, 32 / 16 = fused, on that MFMA shape.  (DITTO_LNQ sets the initial value.)
 * "splitk_wgs": K-splitting of the long-K GEMMs (fc2, final projection) of small batches, with an ordered fp32 reduce.
 * 0 (default) = the low-latency CLASS: launches of at most 2048 rows (1-2 utterances at N = 1024) split by a rule that depends
 * on K only (K = 3072: 4 splits), so an utterance's bits are independent of its batch neighbours INSIDE the class; across the
 * class boundary they differ in the last bits (as across the full-row class boundary), and "fr_class_rows" pins this class too
 * (-11 % step time at C2 B = 1).  -1 = never split (rounds 1-3 default).  > 0 = the older explicit rule: a workgroup target
 * (256 was the measured choice), under which the K partition depends on the batch size.
 * "qkv_split" (default 3; 0 = off): the QKV GEMM's last 256 columns (v columns) as a second launch where the rest makes whole
 * rounds of 256 x 256 tiles on the CUs, for launches of at most that many rounds (d = 768: B = 8, 16 at N = 1024); bit-identical.
 * "ll_mask" (default 3): the other launch forms of the low-latency class, each a function of the class and of K / Skv only:
 * bit 0 = fc2's split-K finish also writes the next block's norm1 (no bit changes), bit 1 = the cross out-projection as two
 * K-splits whose finish writes norm3.  Bit 1 changes a summation order: it is part of what defines the class (pin it with
 * class_rows as usual).  (ABI 9's bit 2, split-KV attention, measured slower and was removed in ABI 10.) */
int ditto_set_option(const char* name, int value);
/* Reads a switch back (callers that change one temporarily restore what they found: ditto_tts_amd/hip.py batch_class). */
int ditto_get_option(const char* name, int* value);

/* Which launches of one DiT block (reference src/components/DiT.py:148+152, :155+:105) a forward over B utterances of N
 * frames runs on the full-row kernel, under the current "fr_mask" / "fr_class_rows" options: *outproj, *fc2 = 0 / 1.  Host
 * arithmetic only (no GPU call): the full-row kernel addresses its operands with 32-bit byte offsets, so each launch is
 * admitted on the row stride IT reads with (fc2: 4 * hidden_dim). */
int ditto_full_row_plan(const ditto_config* cfg, int B, int N, int* outproj, int* fc2);
/* the same question for a call made with `opts` (NULL: as above); *stream_bf16 (may be NULL) = 1 when that forward carries its
 * residual stream as bf16 (d = 768, head_dim 64, bf16 linears, both fused launches on the 128-row kernel) */
int ditto_full_row_plan_opts(const ditto_config* cfg, int B, int N, const ditto_call_opts* opts, int* outproj, int* fc2,
                             int* stream_bf16);

/* ---- either side of the loop -------------------------------------------------------
 * ditto_vq_argmin: VectorQuantizer.forward (src/components/VectorQuantizer.py:22-43): idx[r] = argmin_k
 *   ||latents[r] - codebook[k]||^2 in fp32, first minimum on ties (torch.argmin); latents fp32 [R,D], codebook
 *   fp32 [K,D], idx int64 [R], scratch fp32 [K].
 * ditto_embedding_gather: nn.Embedding lookup out[i] = table[ids[i]] (GPT-2 wte for z_text,
 *   src/model/SpeechGenerator.py:101-103); table fp32 [V,d], ids int64 [n]; an id outside [0,V) is clamped to the
 *   table (nn.Embedding raises on the host; a stream-ordered kernel must not fault).
 * ditto_code_embed_mean: EnCodec codes -> embedding_head rows, mean over the C codebooks, first Fout frames
 *   (src/components/EnCodec.py:35-37, src/model/SpeechGenerator.py:97-98); codes int64 [B,C,F] -> fp32 [B,Fout,d].
 * ditto_linear_update: x <- a[b]*x + ce[b]*eps + cz[b]*noise (noise may be NULL): one step of a strided-DDPM /
 *   DDIM sampler (the paper's 25-step schedule; the reference only has stride 1).  a/ce/cz fp32 [B] (device).
 * ditto_cfg_combine: classifier-free guidance eps = eps_u + w*(eps_c - eps_u); eps2 fp32 [2,B,...] = [cond; uncond]. */
int ditto_vq_argmin(const float* latents, const float* codebook, int64_t* idx, int R, int K, int D, float* scratch_k,
                    ditto_stream_t stream);
int ditto_embedding_gather(const float* table, const int64_t* ids, float* out, int n, int V, int d,
                           ditto_stream_t stream);
int ditto_code_embed_mean(const float* table, const int64_t* codes, float* out, int B, int C, int F, int Fout, int V,
                          int d, ditto_stream_t stream);
int ditto_linear_update(float* x, const float* eps, const float* noise, const float* a, const float* ce,
                        const float* cz, int B, size_t elems_per_utt, ditto_stream_t stream);
int ditto_cfg_combine(const float* eps2, float* out, float w, size_t elems_half, ditto_stream_t stream);

/* ---- training: the backward of DiTTO.forward, so that the reference's training closure
 * (src/TrainDiTTO.py:55-95: model.train(); loss = mse(model(x_t, text, t), noise); loss.backward()) runs on this
 * library.  Gradients are fp32, in the reference's parameter layout (one pointer per state_dict key, as
 * ditto_weights); bf16 operands / fp32 accumulation inside, deterministic (no atomics).  No gradient is produced
 * for x, text_emb (both come from frozen encoders in the reference, TrainDiTTO.py:66-73) or for the dead
 * blocks.i.attn.out_proj.* (never used by the forward, src/components/DiT.py:134-139).
 *
 * ditto_train_attach : packs the transposed bf16 weight copies the dgrad GEMMs read (dX = dY W) into a caller-owned
 *                      arena; call after ditto_model_create, i.e. after every optimizer step.
 * ditto_train_forward: DiTTO.forward in train mode — as ditto_forward, from raw text_emb (no cached cond), keeping the
 *                      activations in `tape`; dropout_p is nn.MultiheadAttention's cross-attention dropout
 *                      (src/components/DiT.py:90-91), its mask a counter-based hash of (seed, layer, b, h, i, j).
 * ditto_train_backward: consumes the tape, writes every gradient of `grads` (overwrites, does not accumulate). */
typedef struct ditto_layer_grads {
    float* norm1_weight;  float* norm1_bias;
    float* attn_in_proj_weight;  float* attn_in_proj_bias;
    float* norm2_weight;  float* norm2_bias;
    float* cross_in_proj_weight; float* cross_in_proj_bias;
    float* cross_out_proj_weight; float* cross_out_proj_bias;
    float* norm3_weight;  float* norm3_bias;
    float* mlp_fc1_weight; float* mlp_fc1_bias;
    float* gate_weight;    float* gate_bias;
    float* mlp_fc2_weight; float* mlp_fc2_bias;
} ditto_layer_grads;
typedef struct ditto_grads {
    float* t_embedding_weight;
    float* time_embed_0_weight; float* time_embed_0_bias;
    float* time_embed_2_weight; float* time_embed_2_bias;
    float* ada_time_mlp_weight; float* ada_time_mlp_bias;
    float* ada_text_mlp_weight; float* ada_text_mlp_bias;
    float* proj_in_weight;  float* proj_in_bias;
    float* proj_out_weight; float* proj_out_bias;
    float* unused_rotary_inv_freq;                                       /* a buffer, no gradient; keeps the layout of ditto_weights */
    const ditto_layer_grads* layers;                                     /* HOST array [num_layers] */
} ditto_grads;
size_t ditto_train_arena_bytes(const ditto_config* cfg);
size_t ditto_tape_bytes(const ditto_config* cfg, int B, int N, int T);
size_t ditto_train_workspace_bytes(const ditto_config* cfg, int B, int N, int T);
int ditto_train_attach(ditto_model_t m, const ditto_weights* w, void* train_arena, size_t train_arena_bytes,
                       ditto_stream_t stream);
int ditto_train_forward(ditto_model_t m, const float* x, const float* text, const int64_t* t, int B, int N, int T,
                        const float* rope_cos, const float* rope_sin, float dropout_p, uint64_t seed, float* eps_out,
                        void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes, ditto_stream_t stream);
int ditto_train_backward(ditto_model_t m, const ditto_weights* w, const float* grad_eps, const float* x,
                         const int64_t* t, int B, int N, int T, const float* rope_cos, const float* rope_sin,
                         float dropout_p, uint64_t seed, const void* tape, size_t tape_bytes, const ditto_grads* grads,
                         void* workspace, size_t workspace_bytes, ditto_stream_t stream);
/* The same two calls with their options as an argument.  The forward RECORDS in the handle, against the tape's address, how it
 * wrote the tape (bf16 or fp32 residual-stream rows); the backward reads the tape as it was written — whatever the options say by
 * then — and fails (DITTO_ERR_ARG / _SHAPE) on a tape no forward of this handle wrote or one written for another (B, N, T). */
int ditto_train_forward_opts(ditto_model_t m, const float* x, const float* text, const int64_t* t, int B, int N, int T,
                             const float* rope_cos, const float* rope_sin, float dropout_p, uint64_t seed, float* eps_out,
                             void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes, ditto_stream_t stream,
                             const ditto_call_opts* opts);
int ditto_train_backward_opts(ditto_model_t m, const ditto_weights* w, const float* grad_eps, const float* x,
                              const int64_t* t, int B, int N, int T, const float* rope_cos, const float* rope_sin,
                              float dropout_p, uint64_t seed, const void* tape, size_t tape_bytes, const ditto_grads* grads,
                              void* workspace, size_t workspace_bytes, ditto_stream_t stream, const ditto_call_opts* opts);
/* The backward in pieces, for a caller that overlaps the data-parallel gradient exchange with it (ditto_tts_amd/dist.py GradSync:
 * the reduce-scatter / all-gather of the layers already done runs on another stream while the layers below are computed): this
 * call runs layers layer_from, layer_from - 1, ..., layer_to (num_layers > layer_from >= layer_to >= 0) and writes THEIR gradients
 * (grads->layers[l]); the call with layer_from == num_layers - 1 also runs the head (proj_in / proj_out gradients), the call with
 * layer_to == 0 the tail (GlobalAdaLN, time embedding).  Successive calls, top layer first, on ONE stream with ONE workspace (it
 * carries the stream gradient between them) are bit-identical to ditto_train_backward, which is the call (num_layers - 1, 0).
 * ONE gradient crosses a piece boundary: grads->layers[l].mlp_fc2_bias is the column sum taken by the LayerNorm backward of layer
 * l + 1's norm1 (the fused fc2 + norm1 launch of the forward), so it is written by the piece that CONTAINS LAYER l + 1 — the call
 * with layer_to == l + 1 writes layers[l].mlp_fc2_bias as well, and a piece's own top layer's fc2 bias gradient was written by the
 * piece above it (the top layer's: by the call with layer_from == num_layers - 1 itself).  A caller that allocates, zeroes or
 * exchanges gradient buffers per piece must treat layers[layer_to - 1].mlp_fc2_bias as part of the piece (dist.py GradSync does). */
int ditto_train_backward_layers(ditto_model_t m, const ditto_weights* w, const float* grad_eps, const float* x,
                                const int64_t* t, int B, int N, int T, const float* rope_cos, const float* rope_sin,
                                float dropout_p, uint64_t seed, const void* tape, size_t tape_bytes, const ditto_grads* grads,
                                void* workspace, size_t workspace_bytes, ditto_stream_t stream, const ditto_call_opts* opts,
                                int layer_from, int layer_to);
/* building blocks of the backward, exported for unit parity tests:
 * ditto_layernorm_bwd: dx_accum fp32 [M,d] += LN'(dy); dgamma_dbeta fp32 [groups, 2d] (rows_per_group * groups = M;
 *   gamma may be NULL = ones; dx_accum or dgamma_dbeta may be NULL); scratch >= ditto_layernorm_bwd_scratch_bytes.
 * ditto_attention_bwd_bf16: dq/dk/dv bf16 from dout bf16 (layouts as ditto_attention_bf16), dropout as above. */
size_t ditto_layernorm_bwd_scratch_bytes(int rows_per_group, int groups, int d);
int ditto_layernorm_bwd(const float* dy, const float* x, const float* gamma, float* dx_accum, float* dgamma_dbeta,
                        void* scratch, size_t scratch_bytes, int rows_per_group, int groups, int d,
                        ditto_stream_t stream);
size_t ditto_attention_bwd_workspace_bytes(int B, int H, int Sq, int Skv, int dh);
/* lse == NULL: GEMM-composed backward (any head_dim % 64 == 0; `out` unused).  lse != NULL (head_dim 64): the fused
 * two-kernel backward; lse fp32 [B,H,Sq] from ditto_attention_dropout_bf16(lse_out) and `out` that call's output. */
int ditto_attention_bwd_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout,
                             int lddo, const void* out, int ldo, const float* lse, void* dq, int lddq, void* dk, int lddk,
                             void* dv, int lddv, int B, int H, int Sq, int Skv, int dh, float scale, float dropout_p,
                             uint64_t seed, int layer, void* workspace, size_t workspace_bytes, ditto_stream_t stream);
/* training-forward attention: as ditto_attention_bf16 with dropout.  lse_out == NULL: the GEMM-composed path;
 * lse_out != NULL (head_dim 64): the fused kernel, which also writes the log2-domain log-sum-exp per query row. */
int ditto_attention_dropout_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out,
                                 int ldo, float* lse_out, int B, int H, int Sq, int Skv, int dh, float scale,
                                 float dropout_p, uint64_t seed, int layer, void* workspace, size_t workspace_bytes,
                                 ditto_stream_t stream);

/* fp8 building blocks, exported for unit parity tests (all e4m3, OCP):
 * ditto_quantize_rows_fp8: fp32 [rows, cols] -> fp8 [rows, cols] + scales fp32 [rows] (amax/448 per row);
 * ditto_layernorm_fp8: as ditto_layernorm_bf16 with a saturating fp8 result;
 * ditto_gemm_fp8: out = (A_fp8[M,K] * W_fp8[N,K]^T) * wscale[n] + bias ..., epilogue 0 (bf16), 1 (fp32 + residual),
 *                 4 (fp32), 5 (gated MLP, fp8 out [M, N/2], interleaved rows as epilogue 3). */
int ditto_quantize_rows_fp8(const float* src, int rows, int cols, void* dst_fp8, float* scales, ditto_stream_t stream);
int ditto_layernorm_fp8(const float* x, const float* gamma, const float* beta, void* out_fp8, int M, int d,
                        ditto_stream_t stream);
int ditto_gemm_fp8(const void* A, int lda, const void* W, const float* wscale, const float* bias,
                   const float* residual, void* out, int ldo, int M, int N, int K, int epilogue,
                   ditto_stream_t stream);

/* ---- the speech-length predictor's decoder stack ------------------------------------------
 * Second user of the GEMM / attention family: the nn.TransformerDecoder the reference builds at
 * src/model/SpeechLP.py:22-32 (post-norm layers, ReLU feed-forward, batch_first, no final norm) run as
 * src/model/SpeechLP.py:47-55 does in eval mode — causal boolean tgt_mask on the self-attention (:50,57-61), unmasked
 * cross-attention to the text memory (:52), then length_predictor on the LAST position (:54).  The pretrained encoders
 * in front of it (ByT5, EnCodec; :17-19,47-49) are out of scope: the entry point takes their outputs.
 * Weights: one pointer per state_dict key of `transformer.layers.{i}.*` / `length_predictor.*`, device fp32, row-major
 * as PyTorch stores them ([out, in]).  d_model and dim_feedforward must be multiples of 64; the head width
 * d_model / nhead is arbitrary (packed zero-padded to a multiple of 64, which changes no result). */
typedef struct ditto_slp_config {
    int32_t d_model, nhead, num_layers, dim_feedforward, num_classes;
} ditto_slp_config;
typedef struct ditto_slp_layer_weights {
    const float *self_in_proj_weight, *self_in_proj_bias;      /* self_attn.in_proj_weight [3d,d], .in_proj_bias [3d] */
    const float *self_out_proj_weight, *self_out_proj_bias;    /* self_attn.out_proj.weight [d,d], .bias [d] */
    const float *cross_in_proj_weight, *cross_in_proj_bias;    /* multihead_attn.in_proj_* */
    const float *cross_out_proj_weight, *cross_out_proj_bias;  /* multihead_attn.out_proj.* */
    const float *linear1_weight, *linear1_bias;                /* [dff,d], [dff] */
    const float *linear2_weight, *linear2_bias;                /* [d,dff], [d] */
    const float *norm1_weight, *norm1_bias, *norm2_weight, *norm2_bias, *norm3_weight, *norm3_bias;
} ditto_slp_layer_weights;
typedef struct ditto_slp_weights {
    const ditto_slp_layer_weights* layers;                     /* [num_layers] (host array of device pointers) */
    const float *length_predictor_weight, *length_predictor_bias;   /* [num_classes,d], [num_classes] */
} ditto_slp_weights;
typedef struct ditto_slp* ditto_slp_t;

size_t ditto_slp_arena_bytes(const ditto_slp_config* cfg);                       /* 0 + last_error on a bad config */
size_t ditto_slp_workspace_bytes(const ditto_slp_config* cfg, int B, int S, int T);
/* packs the weights (bf16, heads padded) into the caller's 256-byte aligned device arena */
int ditto_slp_create(const ditto_slp_config* cfg, const ditto_slp_weights* w, void* arena, size_t arena_bytes,
                     ditto_stream_t stream, ditto_slp_t* out);
void ditto_slp_destroy(ditto_slp_t m);
/* z_audio fp32 [B,S,d] (tgt), z_text fp32 [B,T,d] (memory) -> logits fp32 [B,num_classes];
 * decoded (optional, may be NULL): fp32 [B,S,d], the decoder output for every position (z_audio_decoded, :52). */
int ditto_slp_forward(ditto_slp_t m, const float* z_audio, const float* z_text, int B, int S, int T, float* logits,
                      float* decoded, void* workspace, size_t workspace_bytes, ditto_stream_t stream);
/* the masked attention of that stack on its own (unit parity tests): as ditto_attention_bf16 with the causal mask
 * key j <= query i + (Skv - Sq); always the GEMM-composed path (any head_dim % 64 == 0, workspace required). */
size_t ditto_attention_causal_workspace_bytes(int B, int H, int Sq, int Skv, int dh);
int ditto_attention_causal_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                                void* out, int ldo, int B, int H, int Sq, int Skv, int dh, float scale,
                                void* workspace, size_t workspace_bytes, ditto_stream_t stream);
/* y = LayerNorm(x) * gamma + beta (eps 1e-5) as fp32 AND bf16 (post-norm residual stream + next GEMM operand);
 * either output may be NULL; y_f32 may alias x. */
int ditto_layernorm_dual(const float* x, const float* gamma, const float* beta, float* y_f32, void* y_bf16, int M,
                         int d, ditto_stream_t stream);

/* ---- profiling aid (bench.py): per-kernel-class HIP-event timing -------------------------
 * When enabled on a handle, ditto_forward brackets every launch with hipEvents on `stream`
 * (eager only; do not enable while capturing a graph).  ditto_profile_read synchronises the
 * recorded events and returns, per class, launches and summed milliseconds since the last reset.
 * Independently, with DITTO_ROCTX=1 in the environment every launch of the path sits inside a roctx range named
 * after its class (roctxRangePushA / Pop from librocprofiler-sdk-roctx, loaded lazily), so `rocprofv3
 * --marker-trace --kernel-trace` separates e.g. the out-projection from fc2 although they share one kernel. */
enum { DITTO_KC_LAYERNORM = 0, DITTO_KC_GEMM_QKV, DITTO_KC_GEMM_QPROJ, DITTO_KC_GEMM_OUTPROJ, DITTO_KC_GEMM_GATED,
       DITTO_KC_GEMM_FC2, DITTO_KC_GEMM_FINAL, DITTO_KC_ATTN_SELF, DITTO_KC_ATTN_CROSS, DITTO_KC_ADALN,
       DITTO_KC_UPDATE, DITTO_KC_COUNT };
int ditto_profile_enable(ditto_model_t m, int enable);
int ditto_profile_read(ditto_model_t m, int32_t* launches /*[DITTO_KC_COUNT]*/, float* ms /*[DITTO_KC_COUNT]*/);
const char* ditto_kernel_class_name(int kc);

#ifdef __cplusplus
}
#endif
#endif /* DITTO_HIP_H */
