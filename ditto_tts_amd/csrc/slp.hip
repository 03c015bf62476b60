// slp.hip — the speech-length predictor's decoder stack on the GEMM / attention family (gfx950).
//
// Reference: src/model/SpeechLP.py.  :22-32 builds nn.TransformerDecoder(nn.TransformerDecoderLayer(d_model,
// nhead, dim_feedforward = d_model * nhead, batch_first=True), num_layers) — PyTorch defaults: post-norm
// (norm_first=False), ReLU, LayerNorm eps 1e-5, no final norm — and :47-55 runs it in eval mode on
// (tgt = audio embeddings, memory = text embeddings) with a causal boolean tgt_mask (:50, :57-61), then applies
// length_predictor to the LAST position.  Per layer (torch/nn/modules/transformer.py, TransformerDecoderLayer.forward):
//     x = LN1(x + out_proj(MHA_causal(x, x, x)))
//     x = LN2(x + out_proj(MHA(x, mem, mem)))
//     x = LN3(x + linear2(relu(linear1(x))))
//
// Layout.  Activations are rows of [B*S, d]: the residual stream fp32 (x), a bf16 copy for the next GEMM's A operand
// (xb), both written by one LayerNorm kernel.  Heads of width dh = d / nhead are packed at a stride of
// dhp = ceil(dh / 64) * 64 columns (byt5-small: d = 1472, 4 heads of 368 -> 384): the padded q/k/v rows of the packed
// in_proj weights and the padded columns of the packed out_proj weight are zero, so q k^T, P V and the out projection
// are unchanged while every contraction length is a multiple of the GEMM K-tile.  Attention runs on the GEMM-composed
// path of attention.hip (scores fp32 -> masked row softmax -> P V) — the stack runs once per utterance, in front of
// hundreds of denoise steps, and is not on the timed path.
#include <memory>
#include <new>

#include "common.h"
#include "model.h"

namespace ditto {

namespace {

// y = LN(x) * gamma + beta -> fp32 and / or bf16.  One wave per row, the row in registers (d <= 256 * CH).
template <int CH>
__global__ __launch_bounds__(256) void ln_dual_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ yf,
                                                      bf16* __restrict__ yb, int M, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = d >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * d);
    f32x4 v[CH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = lane + 64 * c;
        v[c] = i < nv ? xr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (lane + 64 * c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dlt = v[c][e] - mean;
                q += dlt * dlt;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = lane + 64 * c;
        if (i >= nv) continue;
        f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (gamma) g = reinterpret_cast<const f32x4*>(gamma)[i];
        if (beta) b = reinterpret_cast<const f32x4*>(beta)[i];
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (v[c][e] - mean) * rstd * g[e] + b[e];
        if (yf) reinterpret_cast<f32x4*>(yf + (size_t)row * d)[i] = y;
        if (yb) {
            u32x2 pk;
            pk[0] = pack_bf16x2(y[0], y[1]);
            pk[1] = pack_bf16x2(y[2], y[3]);
            reinterpret_cast<u32x2*>(yb + (size_t)row * d)[i] = pk;
        }
    }
}

// weight packing with head padding.  rows: src [nsec * H * dh, cols] -> dst row (sec * H + h) * dhp + r (bf16, ld cols)
__global__ __launch_bounds__(256) void pack_head_rows_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                             int rows, int cols, int dh, int dhp) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        dst[((size_t)(r / dh) * dhp + r % dh) * cols + c] = (bf16)src[i];
    }
}
// columns: src [rows, H * dh] -> dst [rows, H * dhp], column h * dhp + c
__global__ __launch_bounds__(256) void pack_head_cols_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                             int rows, int cols, int dh, int dhp, int ld) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        dst[(size_t)r * ld + (size_t)(c / dh) * dhp + c % dh] = (bf16)src[i];
    }
}
__global__ __launch_bounds__(256) void pack_head_vec_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int n, int dh, int dhp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[(size_t)(i / dh) * dhp + i % dh] = src[i];
}
// logits[b, c] = <x[b * S + S - 1, :], W[c, :]> + bias[c]  (fp32; one wave per output)
__global__ __launch_bounds__(256) void last_row_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int B, int S, int d, int C) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= B * C) return;
    const int b = o / C, c = o % C;
    const float* xr = x + ((size_t)b * S + S - 1) * d;
    const float* wr = W + (size_t)c * d;
    float acc = 0.f;
    for (int i = lane; i < d; i += 64) acc += xr[i] * wr[i];
    acc = wave_sum(acc);
    if (lane == 0) out[o] = acc + (bias ? bias[c] : 0.f);
}

inline unsigned grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g > 4096 ? 4096 : (g ? g : 1));
}

struct SlpLayer {
    const void *Wqkv, *Wo, *Wcq, *Wckv, *Wco, *W1, *W2;
    const float *bqkv, *bo, *bcq, *bckv, *bco, *b1, *b2, *g1, *be1, *g2, *be2, *g3, *be3;
};
struct SlpArena {
    size_t total = 0;
    struct L { size_t Wqkv, Wo, Wcq, Wckv, Wco, W1, W2, bqkv, bo, bcq, bckv, bco, b1, b2, g1, be1, g2, be2, g3, be3; };
    std::vector<L> layers;
    size_t Wh, bh;
};
inline int dhp_of(const ditto_slp_config& c) { return ((c.d_model / c.nhead + 63) / 64) * 64; }
SlpArena plan_slp_arena(const ditto_slp_config& c) {
    SlpArena p;
    const size_t d = c.d_model, dp = (size_t)c.nhead * dhp_of(c), ff = c.dim_feedforward;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += al(b); return o; };
    p.layers.resize(c.num_layers);
    for (auto& l : p.layers) {
        l.Wqkv = take(3 * dp * d * 2); l.Wo = take(d * dp * 2); l.Wcq = take(dp * d * 2); l.Wckv = take(2 * dp * d * 2);
        l.Wco = take(d * dp * 2); l.W1 = take(ff * d * 2); l.W2 = take(d * ff * 2);
        l.bqkv = take(3 * dp * 4); l.bo = take(d * 4); l.bcq = take(dp * 4); l.bckv = take(2 * dp * 4);
        l.bco = take(d * 4); l.b1 = take(ff * 4); l.b2 = take(d * 4);
        l.g1 = take(d * 4); l.be1 = take(d * 4); l.g2 = take(d * 4); l.be2 = take(d * 4); l.g3 = take(d * 4);
        l.be3 = take(d * 4);
    }
    p.Wh = take((size_t)c.num_classes * d * 4); p.bh = take((size_t)c.num_classes * 4);
    p.total = off;
    return p;
}
struct SlpWs { size_t x, r, xb, qkv, ao, memb, kv, h, attn, attn_bytes, total; };
SlpWs plan_slp_ws(const ditto_slp_config& c, int B, int S, int T) {
    SlpWs w;
    const size_t M = (size_t)B * S, Mt = (size_t)B * T, d = c.d_model, dp = (size_t)c.nhead * dhp_of(c);
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += al(b); return o; };
    w.x = take(M * d * 4); w.r = take(M * d * 4); w.xb = take(M * d * 2); w.qkv = take(M * 3 * dp * 2);
    w.ao = take(M * dp * 2); w.memb = take(Mt * d * 2); w.kv = take(Mt * 2 * dp * 2);
    w.h = take(M * (size_t)c.dim_feedforward * 2);
    const size_t a1 = attention_generic_workspace_bytes(B, c.nhead, S, S, dhp_of(c));
    const size_t a2 = attention_generic_workspace_bytes(B, c.nhead, S, T, dhp_of(c));
    w.attn_bytes = a1 > a2 ? a1 : a2;
    w.attn = take(w.attn_bytes);
    w.total = off;
    return w;
}
int check_slp_cfg(const ditto_slp_config* c) {
    if (!c) return fail(DITTO_ERR_ARG, "null ditto_slp_config");
    if (c->d_model <= 0 || c->nhead <= 0 || c->num_layers <= 0 || c->dim_feedforward <= 0 || c->num_classes <= 0)
        return fail(DITTO_ERR_SHAPE, "ditto_slp_config fields must be positive");
    if (c->d_model % c->nhead)
        return fail(DITTO_ERR_SHAPE, "d_model %d is not divisible by nhead %d (nn.MultiheadAttention refuses it too)",
                    c->d_model, c->nhead);
    if (c->d_model % 64 || c->dim_feedforward % 64)
        return fail(DITTO_ERR_SHAPE, "d_model and dim_feedforward must be multiples of 64 (got %d, %d)", c->d_model,
                    c->dim_feedforward);
    if (c->d_model > 2048) return fail(DITTO_ERR_SHAPE, "d_model %d > 2048 is not built", c->d_model);
    return DITTO_OK;
}

}  // namespace

hipError_t launch_layernorm_dual(const float* x, const float* gamma, const float* beta, float* yf, void* yb, int M,
                                 int d, hipStream_t s) {
    if (M <= 0 || d <= 0 || d % 4 || d > 2048) return hipErrorInvalidValue;
    const dim3 grid((M + 3) / 4), block(256);
    const int ch = (d / 4 + 63) / 64;
#define DITTO_LN_DUAL(C)                                                                                   \
    case C: hipLaunchKernelGGL((ln_dual_kernel<C>), grid, block, 0, s, x, gamma, beta, yf, (bf16*)yb, M, d); break;
    switch (ch) {
        DITTO_LN_DUAL(1) DITTO_LN_DUAL(2) DITTO_LN_DUAL(3) DITTO_LN_DUAL(4) DITTO_LN_DUAL(5) DITTO_LN_DUAL(6)
        DITTO_LN_DUAL(7) DITTO_LN_DUAL(8)
        default: return hipErrorInvalidValue;
    }
#undef DITTO_LN_DUAL
    return hipGetLastError();
}

}  // namespace ditto

using namespace ditto;

struct ditto_slp {
    ditto_slp_config cfg;
    int dhp;
    char* arena;
    std::vector<SlpLayer> layers;
    const float *Wh, *bh;
};

extern "C" {

size_t ditto_slp_arena_bytes(const ditto_slp_config* cfg) {
    if (check_slp_cfg(cfg)) return 0;
    return plan_slp_arena(*cfg).total;
}

size_t ditto_slp_workspace_bytes(const ditto_slp_config* cfg, int B, int S, int T) {
    if (check_slp_cfg(cfg)) return 0;
    if (B <= 0 || S <= 0 || T <= 0) { fail(DITTO_ERR_SHAPE, "B, S, T must be positive"); return 0; }
    return plan_slp_ws(*cfg, B, S, T).total;
}

int ditto_slp_create(const ditto_slp_config* cfg, const ditto_slp_weights* w, void* arena, size_t arena_bytes,
                     ditto_stream_t stream, ditto_slp_t* out) {
    if (int rc = check_slp_cfg(cfg)) return rc;
    if (!w || !w->layers || !arena || !out || !w->length_predictor_weight)
        return fail(DITTO_ERR_ARG, "null argument to ditto_slp_create");
    const SlpArena plan = plan_slp_arena(*cfg);
    if (arena_bytes < plan.total) return fail(DITTO_ERR_SIZE, "arena too small: %zu < %zu", arena_bytes, plan.total);
    if ((uintptr_t)arena % 256) return fail(DITTO_ERR_ARG, "arena must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int d = cfg->d_model, H = cfg->nhead, dh = d / H, dhp = dhp_of(*cfg), dp = H * dhp, ff = cfg->dim_feedforward;
    std::unique_ptr<ditto_slp> guard(new (std::nothrow) ditto_slp());
    ditto_slp* m = guard.get();
    if (!m) return fail(DITTO_ERR_ARG, "out of host memory");
    m->cfg = *cfg; m->dhp = dhp; m->arena = (char*)arena; m->layers.resize(cfg->num_layers);
    char* A = (char*)arena;
    HIP_TRY(hipMemsetAsync(A, 0, plan.total, s));   // the head padding
    auto rows = [&](const float* src, size_t dst, int nrows) {   // [nrows, d] rows of heads -> padded rows
        hipLaunchKernelGGL(pack_head_rows_kernel, dim3(grid_for((size_t)nrows * d)), dim3(256), 0, s, src,
                           (bf16*)(A + dst), nrows, d, dh, dhp);
        return hipGetLastError();
    };
    auto cols = [&](const float* src, size_t dst) {              // [d, d] -> [d, dp]
        hipLaunchKernelGGL(pack_head_cols_kernel, dim3(grid_for((size_t)d * d)), dim3(256), 0, s, src, (bf16*)(A + dst),
                           d, d, dh, dhp, dp);
        return hipGetLastError();
    };
    auto vec = [&](const float* src, size_t dst, int n) {        // bias of head rows
        hipLaunchKernelGGL(pack_head_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, (float*)(A + dst), n, dh,
                           dhp);
        return hipGetLastError();
    };
    for (int l = 0; l < cfg->num_layers; ++l) {
        const ditto_slp_layer_weights& lw = w->layers[l];
        const auto& q = plan.layers[l];
        const float* need[] = {lw.self_in_proj_weight, lw.self_in_proj_bias, lw.self_out_proj_weight,
                               lw.self_out_proj_bias, lw.cross_in_proj_weight, lw.cross_in_proj_bias,
                               lw.cross_out_proj_weight, lw.cross_out_proj_bias, lw.linear1_weight, lw.linear1_bias,
                               lw.linear2_weight, lw.linear2_bias, lw.norm1_weight, lw.norm1_bias, lw.norm2_weight,
                               lw.norm2_bias, lw.norm3_weight, lw.norm3_bias};
        for (const float* p : need)
            if (!p) return fail(DITTO_ERR_ARG, "null weight pointer in decoder layer %d", l);
        // in_proj rows are q | k | v, each H heads of dh rows: one padded row map covers all three sections
        HIP_TRY(rows(lw.self_in_proj_weight, q.Wqkv, 3 * d));
        HIP_TRY(vec(lw.self_in_proj_bias, q.bqkv, 3 * d));
        HIP_TRY(cols(lw.self_out_proj_weight, q.Wo));
        HIP_TRY(rows(lw.cross_in_proj_weight, q.Wcq, d));
        HIP_TRY(vec(lw.cross_in_proj_bias, q.bcq, d));
        HIP_TRY(rows(lw.cross_in_proj_weight + (size_t)d * d, q.Wckv, 2 * d));
        HIP_TRY(vec(lw.cross_in_proj_bias + d, q.bckv, 2 * d));
        HIP_TRY(cols(lw.cross_out_proj_weight, q.Wco));
        HIP_TRY(launch_pack_bf16(lw.linear1_weight, A + q.W1, ff, d, d, 0, 1 << 30, 1, 0, s));
        HIP_TRY(launch_pack_bf16(lw.linear2_weight, A + q.W2, d, ff, ff, 0, 1 << 30, 1, 0, s));
        const float* vsrc[] = {lw.self_out_proj_bias, lw.cross_out_proj_bias, lw.linear1_bias, lw.linear2_bias,
                               lw.norm1_weight, lw.norm1_bias, lw.norm2_weight, lw.norm2_bias, lw.norm3_weight,
                               lw.norm3_bias};
        const size_t vdst[] = {q.bo, q.bco, q.b1, q.b2, q.g1, q.be1, q.g2, q.be2, q.g3, q.be3};
        const int vlen[] = {d, d, ff, d, d, d, d, d, d, d};
        for (int i = 0; i < 10; ++i)
            HIP_TRY(hipMemcpyAsync(A + vdst[i], vsrc[i], (size_t)vlen[i] * 4, hipMemcpyDeviceToDevice, s));
        SlpLayer& L = m->layers[l];
        L.Wqkv = A + q.Wqkv; L.Wo = A + q.Wo; L.Wcq = A + q.Wcq; L.Wckv = A + q.Wckv; L.Wco = A + q.Wco;
        L.W1 = A + q.W1; L.W2 = A + q.W2;
        L.bqkv = (const float*)(A + q.bqkv); L.bo = (const float*)(A + q.bo); L.bcq = (const float*)(A + q.bcq);
        L.bckv = (const float*)(A + q.bckv); L.bco = (const float*)(A + q.bco); L.b1 = (const float*)(A + q.b1);
        L.b2 = (const float*)(A + q.b2); L.g1 = (const float*)(A + q.g1); L.be1 = (const float*)(A + q.be1);
        L.g2 = (const float*)(A + q.g2); L.be2 = (const float*)(A + q.be2); L.g3 = (const float*)(A + q.g3);
        L.be3 = (const float*)(A + q.be3);
    }
    HIP_TRY(hipMemcpyAsync(A + plan.Wh, w->length_predictor_weight, (size_t)cfg->num_classes * d * 4,
                           hipMemcpyDeviceToDevice, s));
    if (w->length_predictor_bias)
        HIP_TRY(hipMemcpyAsync(A + plan.bh, w->length_predictor_bias, (size_t)cfg->num_classes * 4,
                               hipMemcpyDeviceToDevice, s));
    m->Wh = (const float*)(A + plan.Wh); m->bh = (const float*)(A + plan.bh);
    *out = guard.release();
    return DITTO_OK;
}

void ditto_slp_destroy(ditto_slp_t m) { delete m; }

int ditto_slp_forward(ditto_slp_t m, const float* z_audio, const float* z_text, int B, int S, int T, float* logits,
                      float* decoded, void* workspace, size_t workspace_bytes, ditto_stream_t stream) {
    if (!m || !z_audio || !z_text || !logits || !workspace) return fail(DITTO_ERR_ARG, "null argument to ditto_slp_forward");
    if (B <= 0 || S <= 0 || T <= 0) return fail(DITTO_ERR_SHAPE, "B, S, T must be positive");
    const ditto_slp_config& c = m->cfg;
    const SlpWs w = plan_slp_ws(c, B, S, T);
    if (workspace_bytes < w.total) return fail(DITTO_ERR_SIZE, "workspace too small: %zu < %zu", workspace_bytes, w.total);
    if ((uintptr_t)workspace % 256) return fail(DITTO_ERR_ARG, "workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    char* W = (char*)workspace;
    const int d = c.d_model, H = c.nhead, dhp = m->dhp, dp = H * dhp, ff = c.dim_feedforward;
    const int M = B * S, Mt = B * T;
    float* x = (float*)(W + w.x);
    float* r = (float*)(W + w.r);
    void *xb = W + w.xb, *qkv = W + w.qkv, *ao = W + w.ao, *memb = W + w.memb, *kv = W + w.kv, *h = W + w.h;
    const float scale = 1.0f / sqrtf((float)(d / H));   // of the TRUE head width (F.multi_head_attention_forward)

    auto gemm = [&](const void* A, int lda, const void* Wt, const float* bias, const float* resid, void* out, int ldo,
                    int Mr, int N, int K, GemmEpilogue e) {
        GemmArgs g{};
        g.A = A; g.lda = lda; g.W = Wt; g.bias = bias; g.residual = resid; g.ldr = ldo; g.out = out; g.ldo = ldo;
        g.M = Mr; g.N = N; g.K = K;
        return launch_gemm(g, e, s);
    };
    auto attn = [&](const void* q, int ldq, const void* k, const void* v, int ldkv, int Skv, bool causal) {
        AttnArgs a{};
        a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldkv; a.v = v; a.ldv = ldkv; a.out_bf16 = ao; a.ldo = dp;
        a.B = B; a.H = H; a.Sq = S; a.Skv = Skv; a.dh = dhp; a.scale = scale;
        a.workspace = W + w.attn; a.workspace_bytes = w.attn_bytes;
        a.force_generic = true; a.causal = causal;
        return launch_attention(a, s);
    };

    HIP_TRY(launch_cast_bf16(z_audio, xb, (size_t)M * d, s));
    HIP_TRY(launch_cast_bf16(z_text, memb, (size_t)Mt * d, s));
    const float* xin = z_audio;
    const bf16* qkv_b = (const bf16*)qkv;
    const bf16* kv_b = (const bf16*)kv;
    for (int l = 0; l < c.num_layers; ++l) {
        const SlpLayer& L = m->layers[l];
        // self-attention block (_sa_block, causal)
        HIP_TRY(gemm(xb, d, L.Wqkv, L.bqkv, nullptr, qkv, 3 * dp, M, 3 * dp, d, EPI_BIAS_BF16));
        HIP_TRY(attn(qkv_b, 3 * dp, qkv_b + dp, qkv_b + 2 * dp, 3 * dp, S, true));
        HIP_TRY(gemm(ao, dp, L.Wo, L.bo, xin, r, d, M, d, dp, EPI_BIAS_RES_F32));
        HIP_TRY(launch_layernorm_dual(r, L.g1, L.be1, x, xb, M, d, s));
        // cross-attention block (_mha_block) on the text memory
        HIP_TRY(gemm(xb, d, L.Wcq, L.bcq, nullptr, qkv, dp, M, dp, d, EPI_BIAS_BF16));
        HIP_TRY(gemm(memb, d, L.Wckv, L.bckv, nullptr, kv, 2 * dp, Mt, 2 * dp, d, EPI_BIAS_BF16));
        HIP_TRY(attn(qkv_b, dp, kv_b, kv_b + dp, 2 * dp, T, false));
        HIP_TRY(gemm(ao, dp, L.Wco, L.bco, x, r, d, M, d, dp, EPI_BIAS_RES_F32));
        HIP_TRY(launch_layernorm_dual(r, L.g2, L.be2, x, xb, M, d, s));
        // feed-forward block (_ff_block)
        HIP_TRY(gemm(xb, d, L.W1, L.b1, nullptr, h, ff, M, ff, d, EPI_BIAS_RELU_BF16));
        HIP_TRY(gemm(h, ff, L.W2, L.b2, x, r, d, M, d, ff, EPI_BIAS_RES_F32));
        const bool last = l + 1 == c.num_layers;
        HIP_TRY(launch_layernorm_dual(r, L.g3, L.be3, x, last ? nullptr : xb, M, d, s));
        xin = x;
    }
    if (decoded) HIP_TRY(hipMemcpyAsync(decoded, x, (size_t)M * d * 4, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(last_row_linear_kernel, dim3((B * c.num_classes + 3) / 4), dim3(256), 0, s, x, m->Wh, m->bh, logits,
                       B, S, d, c.num_classes);
    HIP_TRY(hipGetLastError());
    return DITTO_OK;
}

size_t ditto_attention_causal_workspace_bytes(int B, int H, int Sq, int Skv, int dh) {
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0 || dh <= 0 || dh % 64) return 0;
    return attention_generic_workspace_bytes(B, H, Sq, Skv, dh);
}

int ditto_attention_causal_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo,
                                int B, int H, int Sq, int Skv, int dh, float scale, void* workspace,
                                size_t workspace_bytes, ditto_stream_t stream) {
    if (!q || !k || !v || !out || !workspace) return fail(DITTO_ERR_ARG, "null pointer to ditto_attention_causal_bf16");
    if (dh % 64) return fail(DITTO_ERR_SHAPE, "head_dim must be a multiple of 64");
    if (workspace_bytes < attention_generic_workspace_bytes(B, H, Sq, Skv, dh))
        return fail(DITTO_ERR_SIZE, "attention workspace too small");
    AttnArgs a{};
    a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.out_bf16 = out; a.ldo = ldo;
    a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.dh = dh; a.scale = scale;
    a.workspace = workspace; a.workspace_bytes = workspace_bytes;
    a.force_generic = true; a.causal = true;
    HIP_TRY(launch_attention(a, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_layernorm_dual(const float* x, const float* gamma, const float* beta, float* y_f32, void* y_bf16, int M, int d,
                         ditto_stream_t stream) {
    if (!x || (!y_f32 && !y_bf16)) return fail(DITTO_ERR_ARG, "null pointer to ditto_layernorm_dual");
    if (M <= 0 || d <= 0 || d % 4 || d > 2048) return fail(DITTO_ERR_SHAPE, "need 0 < d <= 2048, d %% 4 == 0");
    HIP_TRY(launch_layernorm_dual(x, gamma, beta, y_f32, y_bf16, M, d, (hipStream_t)stream));
    return DITTO_OK;
}

}  // extern "C"
