// ditto_train.hip — training entry points of libditto_hip.so (include/ditto_hip.h, "training"): the forward that
// keeps its activations and the backward of DiTTO.forward (reference src/model/DiTTO.py:66-94 + src/components/
// DiT.py:25-40,100-157 under autograd, as driven by src/TrainDiTTO.py:55-95).
//
// HBM layout.  `tape` (caller-owned, ditto_tape_bytes): kv bf16[Mt, L*2d] | tmod fp32[B,2d] | text bf16[Mt,d] |
//   pooled fp32[B,dt] | xcat bf16[M,2d] | hs fp32[3L+1][M,d] (the residual stream after AdaLN, after every
//   self-attention, cross-attention and MLP segment) | per layer { u1,u2,u3 bf16[M,d] (LayerNorm outputs) |
//   qkv bf16[M,3d] (RoPE applied) | qc bf16[M,d] | oc bf16[M,d] | pre bf16[M,8d] (fc1|gate pre-activations,
//   interleaved by 16) | act bf16[M,4d] | lse_self, lse_cross fp32[B,H,N] (head_dim 64 only) }.   With 288 GB of HBM nothing is recomputed except the attention
//   probabilities: C2 at B = 32 keeps 17 GB.
// Backward data flow per segment (dh = gradient of the fp32 residual stream, updated in place):
//   dyb = bf16(dh) -> bias grad = colsum(dh) -> wgrad GEMM (dyb^T x act^T, both transposed to K-contiguous) ->
//   dgrad GEMM (dyb x W^T pack) -> elementwise backward -> ... -> LayerNorm backward adds into dh.
#include <cmath>
#include <cstring>

#include "gemm_common.h"
#include "model.h"

using namespace ditto;

namespace {

inline size_t pad64(size_t x) { return (x + 63) & ~(size_t)63; }

struct TapePlan {
    size_t kv, tmod, text, pooled, xcat, hs, hs_stride;
    struct L { size_t u1, u2, u3, qkv, qc, oc, pre, act, lse_s, lse_c; };
    std::vector<L> layers;
    size_t total;
};
TapePlan plan_tape(const ditto_config& c, int B, int N, int T) {
    TapePlan p;
    const size_t d = c.hidden_dim, L = c.num_layers, M = (size_t)B * N, Mt = (size_t)B * T;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    p.kv = take(Mt * L * 2 * d * 2); p.tmod = take((size_t)B * 2 * d * 4); p.text = take(Mt * c.text_dim * 2);
    p.pooled = take((size_t)B * c.text_dim * 4); p.xcat = take(M * 2 * d * 2);
    p.hs_stride = al(M * d * 4);
    p.hs = off; off += (3 * L + 1) * p.hs_stride;
    p.layers.resize(L);
    for (auto& q : p.layers) {
        q.u1 = take(M * d * 2); q.u2 = take(M * d * 2); q.u3 = take(M * d * 2); q.qkv = take(M * 3 * d * 2);
        q.qc = take(M * d * 2); q.oc = take(M * d * 2); q.pre = take(M * 8 * d * 2); q.act = take(M * 4 * d * 2);
        q.lse_s = take((size_t)B * c.num_heads * N * 4); q.lse_c = take((size_t)B * c.num_heads * N * 4);
    }
    p.total = off;
    return p;
}

struct TrainArenaPlan {
    struct L { size_t WqkvT, WcqT, WcoT, W1gT, W2T, Wqkv_u, Wcq_u, bqkv_u, bcq_u, W1gTP, WqkvTP; };
    std::vector<L> layers;
    size_t WoutT, total;
};
TrainArenaPlan plan_train_arena(const ditto_config& c) {
    TrainArenaPlan p;
    const size_t d = c.hidden_dim;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    p.layers.resize(c.num_layers);
    for (auto& q : p.layers) {
        q.WqkvT = take(3 * d * d * 2); q.WcqT = take(d * d * 2); q.WcoT = take(d * d * 2); q.W1gT = take(8 * d * d * 2);
        q.W2T = take(4 * d * d * 2);
        q.Wqkv_u = take(3 * d * d * 2); q.Wcq_u = take(d * d * 2); q.bqkv_u = take(3 * d * 4); q.bcq_u = take(d * 4);
        q.W1gTP = q.WqkvTP = 0;
        if (d == 768) { q.W1gTP = take(8 * d * d * 2); q.WqkvTP = take(3 * d * d * 2); }   // full-row kernel's layout (gemm_fr.hip)
    }
    p.WoutT = take(d * d * 2);
    p.total = off;
    return p;
}

struct TrainWsPlan {
    size_t dh, du, dyb, big1, big2, dkv, zero, wtmp, vtmp, red, dmod, small, wpart, attn, attn_bytes, total;
};
// split-K factor of a wgrad GEMM with `tiles` 128x128 output tiles and `kt` K-tiles of 64.  Cost model, in units of
// one K-tile of one resident workgroup (~1 us; 2 workgroups per CU = 512 slots): rounds(S) * ceil(kt / S) for the
// GEMM + the bytes the ordered reduce moves ((S + 1) tile images at ~3 TB/s) + its launch.  Measured shapes at C2,
// B = 16 (rocprofv3, per launch): d x d 36 tiles S=11 43.8 us; QKV 108 tiles S=4 90.5 us; fc2 144 tiles S=3 104.6 us;
// fc1|gate 288 tiles: S=2 (576 workgroups = 1.1 rounds of 512) 253.9 us -> the model picks S=3 (864 = 1.7 rounds of 86
// K-tiles).  g_wgrad_wgs > 0 (ditto_set_option("wgrad_wgs")) forces the old rule "at least that many workgroups".
int g_wgrad_wgs = 0;
inline int wgrad_splits(long tiles, long kt) {
    const long smax = kt / 8 < 32 ? kt / 8 : 32;
    if (smax <= 1) return 1;
    if (g_wgrad_wgs > 0) {
        long s = (g_wgrad_wgs + tiles - 1) / tiles;
        if (s > smax) s = smax;
        return (int)(s < 1 ? 1 : s);
    }
    const double slots = 512.0, tile_us = 128.0 * 128.0 * 4.0 / 3.0e6;   // one fp32 tile image at 3 TB/s, in us
    long best = 1;
    double best_cost = 1e30;
    for (long s = 1; s <= smax; ++s) {
        const double rounds = (double)((tiles * s + (long)slots - 1) / (long)slots);
        double cost = rounds * (double)((kt + s - 1) / s);
        if (s > 1) cost += 5.0 + (double)(s + 1) * (double)tiles * tile_us;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return (int)best;
}
// The same for the 256 x 256 kernel (gemm_tn_wide_kernel: one workgroup per CU = 256 slots, ~1.6 us per K-tile of 64 of a
// resident workgroup, 256 KiB per fp32 tile image).  Measured at C2, B = 32 (tools/tn_bench.py, best split in brackets,
// against the 128 x 128 kernel's best): d x d 67.5 us [24] vs 85.9 [6]; 2d x d 105.7 [12] vs 137.8; QKV 144.0 [8] vs 184.8;
// fc2 188.0 [6] vs 207.2; fc1|gate 348.0 [3] vs 446.6 — every shape is fastest at tiles * S ~ one round of 256.
inline int wgrad_splits_wide(long tiles, long kt) {
    const long smax = kt / 8 < 32 ? kt / 8 : 32;
    if (smax <= 1) return 1;
    const double slots = 256.0, unit = 1.6, tile_us = 256.0 * 256.0 * 4.0 / 3.0e6;
    long best = 1;
    double best_cost = 1e30;
    for (long s = 1; s <= smax; ++s) {
        const double rounds = (double)((tiles * s + (long)slots - 1) / (long)slots);
        double cost = rounds * (double)((kt + s - 1) / s) * unit;
        if (s > 1) cost += 5.0 + (double)(s + 1) * (double)tiles * tile_us;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return (int)best;
}
constexpr size_t WPART_BYTES = (size_t)(2048 + 320) * 128 * 128 * 4;   // S * tiles: up to ~4 rounds of 512 workgroups
TrainWsPlan plan_train_ws(const ditto_config& c, int B, int N, int T) {
    TrainWsPlan w;
    const size_t d = c.hidden_dim, M = (size_t)B * N, Mt = (size_t)B * T, dh = d / c.num_heads;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    w.dh = take(M * d * 4); w.du = take(M * d * 4); w.dyb = take(M * d * 2);
    w.big1 = take(M * 8 * d * 2); w.big2 = take(M * 4 * d * 2); w.dkv = take(Mt * 2 * d * 2);
    w.zero = take(256);
    w.wtmp = take(8 * d * d * 4); w.vtmp = take(8 * d * 4);
    size_t red = (size_t)256 * 8 * d * 4;   // colsum: <= 256 row chunks x <= 8d columns
    const size_t r2 = ln_bwd_scratch_bytes((int)M, 1, (int)d), r3 = ln_bwd_scratch_bytes(N, B, (int)d);
    red = red > r2 ? red : r2; red = red > r3 ? red : r3;
    const size_t r4 = (size_t)2 * ((M + 255) / 256) * 8 * d * 4;   // fused fc2 dgrad + gated backward: one partial row per 128-row half tile
    red = red > r4 ? red : r4;
    w.red = take(red);
    w.dmod = take((size_t)B * 2 * d * 4);
    w.small = take(6 * al((size_t)B * c.time_dim * 4));
    w.wpart = take(WPART_BYTES);
    const size_t a1 = attention_train_workspace_bytes(B, c.num_heads, N, N, (int)dh);
    const size_t a2 = attention_train_workspace_bytes(B, c.num_heads, N, T, (int)dh);
    w.attn_bytes = a1 > a2 ? a1 : a2;
    w.attn = take(w.attn_bytes);
    w.total = off;
    return w;
}

int check_train(const ditto_model* m) {
    if (!m) return fail(DITTO_ERR_ARG, "null model");
    if (m->blocks_only) return fail(DITTO_ERR_ARG, "training needs a full model handle (not blocks-only)");
    if (m->cfg.flags & DITTO_CFG_FP8_LINEAR) return fail(DITTO_ERR_SHAPE, "training with fp8 linear layers is not supported");
    if (cfg_padded(m->cfg))
        return fail(DITTO_ERR_SHAPE, "training needs head_dim %% 64 == 0 (head_dim %d runs forward-only, on padded heads)",
                    m->cfg.hidden_dim / m->cfg.num_heads);
    return DITTO_OK;
}

}  // namespace

namespace ditto {
void set_wgrad_wgs(int v) { g_wgrad_wgs = v; }
int get_wgrad_wgs() { return g_wgrad_wgs; }
}

extern "C" {

size_t ditto_train_arena_bytes(const ditto_config* cfg) {
    if (check_cfg(cfg) != DITTO_OK || cfg_padded(*cfg)) return 0;
    return plan_train_arena(*cfg).total;
}
size_t ditto_tape_bytes(const ditto_config* cfg, int B, int N, int T) {
    if (check_cfg(cfg) != DITTO_OK || cfg_padded(*cfg) || B <= 0 || N <= 0 || T <= 0) return 0;
    return plan_tape(*cfg, B, N, T).total;
}
size_t ditto_train_workspace_bytes(const ditto_config* cfg, int B, int N, int T) {
    if (check_cfg(cfg) != DITTO_OK || cfg_padded(*cfg) || B <= 0 || N <= 0 || T <= 0) return 0;
    return plan_train_ws(*cfg, B, N, T).total;
}

int ditto_train_attach(ditto_model_t m, const ditto_weights* w, void* train_arena, size_t train_arena_bytes,
                       ditto_stream_t stream) {
    if (int rc = check_train(m)) return rc;
    if (!w || !w->layers || !train_arena) return fail(DITTO_ERR_ARG, "null argument to ditto_train_attach");
    const ditto_config& c = m->cfg;
    const TrainArenaPlan p = plan_train_arena(c);
    if (train_arena_bytes < p.total) return fail(DITTO_ERR_SIZE, "train arena too small: %zu < %zu", train_arena_bytes, p.total);
    if ((uintptr_t)train_arena % 256) return fail(DITTO_ERR_ARG, "train arena must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int d = c.hidden_dim, L = c.num_layers, BIG = 1 << 30;
    char* A = (char*)train_arena;
    m->layersT.resize(L);
    for (int l = 0; l < L; ++l) {
        const ditto_layer_weights& lw = w->layers[l];
        const auto& q = p.layers[l];
        // W [out, in] -> W^T [in, out]: the "weight" of the dgrad GEMM  dX[M,in] = dY[M,out] * W
        HIP_TRY(launch_pack_bf16_t(lw.attn_in_proj_weight, A + q.WqkvT, 3 * d, d, 3 * d, BIG, 1, 0, s));
        HIP_TRY(launch_pack_bf16_t(lw.cross_in_proj_weight, A + q.WcqT, d, d, d, BIG, 1, 0, s));
        HIP_TRY(launch_pack_bf16_t(lw.cross_out_proj_weight, A + q.WcoT, d, d, d, BIG, 1, 0, s));
        HIP_TRY(launch_pack_bf16_t(lw.mlp_fc1_weight, A + q.W1gT, 4 * d, d, 8 * d, 16, 2, 0, s));
        HIP_TRY(launch_pack_bf16_t(lw.gate_weight, A + q.W1gT, 4 * d, d, 8 * d, 16, 2, 16, s));
        HIP_TRY(launch_pack_bf16_t(lw.mlp_fc2_weight, A + q.W2T, d, 4 * d, d, BIG, 1, 0, s));
        // forward copies of the q projections WITHOUT the softmax scale the inference pack folds in: the training
        // kernels (log-sum-exp, backward) work on the reference's own q
        HIP_TRY(launch_pack_bf16(lw.attn_in_proj_weight, A + q.Wqkv_u, 3 * d, d, d, 0, BIG, 1, 0, s));
        HIP_TRY(launch_pack_bf16(lw.cross_in_proj_weight, A + q.Wcq_u, d, d, d, 0, BIG, 1, 0, s));
        HIP_TRY(hipMemcpyAsync(A + q.bqkv_u, lw.attn_in_proj_bias, (size_t)3 * d * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(A + q.bcq_u, lw.cross_in_proj_bias, (size_t)d * 4, hipMemcpyDeviceToDevice, s));
        const void *w1gtp = nullptr, *wqkvtp = nullptr;
        if (d == 768) {
            HIP_TRY(launch_repack_bf16_stage_major(A + q.W1gT, A + q.W1gTP, d, 8 * d, s));
            HIP_TRY(launch_repack_bf16_stage_major(A + q.WqkvT, A + q.WqkvTP, d, 3 * d, s));
            w1gtp = A + q.W1gTP; wqkvtp = A + q.WqkvTP;
        }
        m->layersT[l] = LayerPackT{A + q.WqkvT, A + q.WcqT, A + q.WcoT, A + q.W1gT, A + q.W2T,
                                   A + q.Wqkv_u, A + q.Wcq_u, (const float*)(A + q.bqkv_u), (const float*)(A + q.bcq_u),
                                   w1gtp, wqkvtp};
    }
    HIP_TRY(launch_pack_bf16_t(w->proj_out_weight, A + p.WoutT, d, d, d, BIG, 1, 0, s));
    m->WoutT = A + p.WoutT;
    return DITTO_OK;
}

// The training step on the BF16 residual stream (round 4): the tape keeps h as bf16 rows — written by the AdaLN kernel, updated by
// the self-attention epilogue, read and written by the two full-row launches, read by the LayerNorms and by the LayerNorm
// backward — and the self-attention's output O as bf16 beside h1 (second half of its fp32-sized slot: the backward's
// delta = rowsum(dO . O) needs O, and the difference of two bf16 rows is not it).  Exactly where the inference forward has the
// stream (ditto_api.hip hb_class): d = 768, head_dim 64, both fused launches on gemm_frd.hip; and with du travelling as bf16.
// Decided ONCE per step, by the forward, and recorded in the handle against the tape's address (ditto_model::tapes): the backward
// reads the tape as it was written, whatever the options say by then.
static bool train_stream_bf16(ditto_model_t m, int M) {
    const ditto_config& c = m->cfg;
    const int d = c.hidden_dim;
    if ((g_train_flags & (8 | 16)) || (g_attn_flags & 8192) || d != 768 || d / c.num_heads != 64) return false;
    if (!m->layers[0].WcoP || !m->layers[0].W2P || (opt_fr_mask() & 3) != 3) return false;
    return fr_outproj_ok(M, d) && fr_fc2_ok(M, d) && fr_launch_kernel(M, d) == 130 && fr_launch_kernel(M, 4 * d) == 130;
}

int ditto_train_forward(ditto_model_t m, const float* x, const float* text, const int64_t* t, int B, int N, int T,
                        const float* rope_cos, const float* rope_sin, float dropout_p, uint64_t seed, float* eps_out,
                        void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes, ditto_stream_t stream) {
    return ditto_train_forward_opts(m, x, text, t, B, N, T, rope_cos, rope_sin, dropout_p, seed, eps_out, tape, tape_bytes, workspace,
                                    workspace_bytes, stream, nullptr);
}

int ditto_train_forward_opts(ditto_model_t m, const float* x, const float* text, const int64_t* t, int B, int N, int T,
                             const float* rope_cos, const float* rope_sin, float dropout_p, uint64_t seed, float* eps_out,
                             void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes, ditto_stream_t stream,
                             const ditto_call_opts* opts) {
    if (int rc = check_call_opts(opts)) return rc;
    CallScope scope(opts);
    if (int rc = check_train(m)) return rc;
    if (!x || !text || !t || !rope_cos || !rope_sin || !eps_out || !tape || !workspace || B <= 0 || N <= 0 || T <= 0)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_train_forward");
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail(DITTO_ERR_ARG, "dropout_p must be in [0, 1)");
    if (m->layersT.empty()) return fail(DITTO_ERR_ARG, "ditto_train_attach was not called on this handle");
    const ditto_config& c = m->cfg;
    const TapePlan tp = plan_tape(c, B, N, T);
    const TrainWsPlan wp = plan_train_ws(c, B, N, T);
    if (tape_bytes < tp.total) return fail(DITTO_ERR_SIZE, "tape too small: %zu < %zu", tape_bytes, tp.total);
    if (workspace_bytes < wp.total) return fail(DITTO_ERR_SIZE, "train workspace too small: %zu < %zu", workspace_bytes, wp.total);
    if ((uintptr_t)tape % 256 || (uintptr_t)workspace % 256) return fail(DITTO_ERR_ARG, "tape / workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int d = c.hidden_dim, L = c.num_layers, H = c.num_heads, dh = d / H, M = B * N, Mt = B * T;
    const float scale = 1.0f / sqrtf((float)dh);
    const bool fused_rope = (dh == 64);
    char* tb = (char*)tape;
    char* ws = (char*)workspace;
    char* kv = tb + tp.kv;
    float* tmod = (float*)(tb + tp.tmod);
    void* textbf = tb + tp.text;
    float* pooled = (float*)(tb + tp.pooled);
    char* xcat = tb + tp.xcat;
    auto hs = [&](int i) { return (float*)(tb + tp.hs + (size_t)i * tp.hs_stride); };
    void* attn_ws = ws + wp.attn;

    // text: bf16 copy, every layer's cross-attention K/V projection, the text half of the AdaLN modulation
    HIP_TRY(launch_cast_bf16(text, textbf, (size_t)Mt * c.text_dim, s));
    {
        GemmArgs g{};
        g.A = textbf; g.lda = c.text_dim; g.W = m->Wkv; g.bias = m->bkv; g.out = kv; g.ldo = L * 2 * d;
        g.M = Mt; g.N = L * 2 * d; g.K = d;
        HIP_TRY(launch_gemm(g, EPI_BIAS_BF16, s));
    }
    HIP_TRY(launch_text_mod(text, m->wx, m->bx, pooled, tmod, B, T, c.text_dim, d, s));
    const bool hb = train_stream_bf16(m, M);
    {   // a tape whose forward fails half-way must not look written: drop any older record of this address now, record at the end
        std::lock_guard<std::mutex> lk(m->tape_mu);
        m->tapes.erase(tape);
    }
    if (hb)   // bf16 h0 + block 0's norm1 from the same kernel, as the inference forward
        HIP_TRY(launch_adaln(x, m->ttab, tmod, t, c.diffusion_steps, hs(0), xcat, 2 * d, B, N, d, s, true, m->layers[0].g1,
                             m->layers[0].be1, tb + tp.layers[0].u1));
    else
        HIP_TRY(launch_adaln(x, m->ttab, tmod, t, c.diffusion_steps, hs(0), xcat, 2 * d, B, N, d, s));

    // As in the inference forward (ditto_api.hip run_block): from 160 row tiles on, the cross out-projection + norm3 and
    // fc2 + the next block's norm1 run on the full-row kernel (fr_mask); the LayerNorm outputs land in the tape slots the
    // backward reads (u3, the next block's u1).
    const bool fr_have = d == 768 && m->layers[0].WcoP != nullptr && m->layers[0].W2P != nullptr;
    const bool fr_out = fr_have && (opt_fr_mask() & 1) && fr_outproj_ok(M, d);
    const bool fr_fc2 = fr_have && (opt_fr_mask() & 2) && fr_fc2_ok(M, d);
    const int fr_rot = N % 128 == 0 ? N / 128 : 0;
    for (int l = 0; l < L; ++l) {
        const LayerPack& lp = m->layers[l];
        const LayerPackT& lt = m->layersT[l];
        const auto& q = tp.layers[l];
        float *h0 = hs(3 * l), *h1 = hs(3 * l + 1), *h2 = hs(3 * l + 2), *h3 = hs(3 * l + 3);
        char* qkv = tb + q.qkv;
        // ---- self-attention ----
        if (!hb && !(fr_fc2 && l > 0)) HIP_TRY(launch_layernorm(h0, lp.g1, lp.be1, tb + q.u1, d, M, d, s));
        {
            GemmArgs g{};
            g.A = tb + q.u1; g.lda = d; g.W = lt.Wqkv_u; g.bias = lt.bqkv_u; g.out = qkv; g.ldo = 3 * d;
            g.M = M; g.N = 3 * d; g.K = d;
            g.rope_cos = rope_cos; g.rope_sin = rope_sin; g.rope_rows_per_batch = N; g.rope_cols = 2 * d;
            HIP_TRY(launch_gemm(g, fused_rope ? EPI_QKV_ROPE : EPI_BIAS_BF16, s));
            if (!fused_rope) HIP_TRY(launch_rope_inplace(qkv, 3 * d, rope_cos, rope_sin, M, N, 2 * d, dh, s));
        }
        {
            AttnArgs a{};
            a.q = qkv; a.ldq = 3 * d; a.k = qkv + (size_t)d * 2; a.ldk = 3 * d; a.v = qkv + (size_t)2 * d * 2;
            a.ldv = 3 * d; a.resid_f32 = h1; a.resid_in = h0; a.ldr = d; a.B = B; a.H = H; a.Sq = N; a.Skv = N;
            a.dh = dh; a.scale = scale; a.workspace = attn_ws; a.workspace_bytes = wp.attn_bytes;
            if (dh == 64) a.lse_out = (float*)(tb + q.lse_s);
            if (hb) { a.resid_bf16 = true; a.out_bf16 = (char*)h1 + (size_t)M * d * 2; a.ldo = d; }   // + O for the backward
            HIP_TRY(launch_attention(a, s));
        }
        // ---- cross-attention (dropout on the probabilities in train mode) ----
        if (hb) HIP_TRY(launch_layernorm_xbf16(h1, lp.g2, lp.be2, tb + q.u2, d, M, d, s));
        else HIP_TRY(launch_layernorm(h1, lp.g2, lp.be2, tb + q.u2, d, M, d, s));
        {
            GemmArgs g{};
            g.A = tb + q.u2; g.lda = d; g.W = lt.Wcq_u; g.bias = lt.bcq_u; g.out = tb + q.qc; g.ldo = d;
            g.M = M; g.N = d; g.K = d;
            HIP_TRY(launch_gemm(g, EPI_BIAS_BF16, s));
        }
        {
            AttnArgs a{};
            a.q = tb + q.qc; a.ldq = d; a.k = kv + (size_t)l * 2 * d * 2; a.ldk = L * 2 * d;
            a.v = kv + ((size_t)l * 2 * d + d) * 2; a.ldv = L * 2 * d; a.out_bf16 = tb + q.oc; a.ldo = d;
            a.B = B; a.H = H; a.Sq = N; a.Skv = T; a.dh = dh; a.scale = scale;
            a.workspace = attn_ws; a.workspace_bytes = wp.attn_bytes;
            a.dropout_p = dropout_p; a.seed = seed; a.layer = l;
            if (dh == 64) a.lse_out = (float*)(tb + q.lse_c);
            HIP_TRY(launch_attention(a, s));
        }
        if (fr_out) {
            GemmParams gp{};
            gp.A = (const bf16*)(tb + q.oc); gp.lda = d; gp.W = (const bf16*)lp.WcoP; gp.ldw = d; gp.w_rows = d; gp.bias = lp.bco;
            gp.residual = h1; gp.ldr = d; gp.out = h2; gp.ldo = d; gp.M = M; gp.N = d; gp.K = d;
            HIP_TRY(launch_gemm_fr(gp, lp.g3, lp.be3, tb + q.u3, d, fr_rot, s, false, hb));
        } else {
            GemmArgs g{};
            g.A = tb + q.oc; g.lda = d; g.W = lp.Wco; g.bias = lp.bco; g.residual = h1; g.ldr = d; g.out = h2; g.ldo = d;
            g.M = M; g.N = d; g.K = d;
            HIP_TRY(launch_gemm(g, EPI_BIAS_RES_F32, s));
            // ---- gated MLP: product and pre-activations (for the backward) from one epilogue ----
            HIP_TRY(launch_layernorm(h2, lp.g3, lp.be3, tb + q.u3, d, M, d, s));
        }
        {
            GemmArgs g{};
            g.A = tb + q.u3; g.lda = d; g.W = lp.W1g; g.bias = lp.b1g; g.out = tb + q.act; g.ldo = 4 * d;
            g.out2_bf16 = tb + q.pre; g.ldo2 = 8 * d;   // the gated epilogue also keeps the pre-activations
            g.M = M; g.N = 8 * d; g.K = d;
            // EPI_GATED_PRE = the same epilogue as its own instantiation of the 256 x 256 kernel (straight-line code; train_flags 4: A/B off)
            const bool pre_inst = !(g_train_flags & 4) && gemm_gated_bwd_fused_ok(M, 8 * d);
            HIP_TRY(launch_gemm(g, pre_inst ? EPI_GATED_PRE : EPI_GATED, s));
        }
        if (fr_fc2) {
            GemmParams gp{};
            gp.A = (const bf16*)(tb + q.act); gp.lda = 4 * d; gp.W = (const bf16*)lp.W2P; gp.ldw = 4 * d; gp.w_rows = d;
            gp.bias = lp.b2; gp.residual = h2; gp.ldr = d; gp.out = h3; gp.ldo = d; gp.M = M; gp.N = d; gp.K = 4 * d;
            if (l == L - 1 && hb) {   // the last block's bf16 h goes straight into the final projection's operand slot
                gp.out = xcat + (size_t)d * 2; gp.ldo = 2 * d;
                HIP_TRY(launch_gemm_fr(gp, nullptr, nullptr, nullptr, d, fr_rot, s, false, true));
            } else if (l == L - 1) {
                gp.out2 = (bf16*)(xcat + (size_t)d * 2); gp.ldo2 = 2 * d;
                HIP_TRY(launch_gemm_fr(gp, nullptr, nullptr, nullptr, d, fr_rot, s));
            } else {
                HIP_TRY(launch_gemm_fr(gp, m->layers[l + 1].g1, m->layers[l + 1].be1, tb + tp.layers[l + 1].u1, d, fr_rot, s,
                                       false, hb));
            }
        } else {
            GemmArgs g{};
            g.A = tb + q.act; g.lda = 4 * d; g.W = lp.W2; g.bias = lp.b2; g.residual = h2; g.ldr = d; g.out = h3;
            g.ldo = d; g.M = M; g.N = d; g.K = 4 * d;
            if (l == L - 1) { g.out2_bf16 = xcat + (size_t)d * 2; g.ldo2 = 2 * d; }
            HIP_TRY(launch_gemm(g, EPI_BIAS_RES_F32, s));
        }
    }
    {
        GemmArgs g{};
        g.A = xcat; g.lda = 2 * d; g.W = m->Wfin; g.bias = m->bfin; g.out = eps_out; g.ldo = d; g.M = M; g.N = d;
        g.K = 2 * d;
        HIP_TRY(launch_gemm(g, EPI_BIAS_F32, s));
    }
    {   // every launch of the forward is enqueued: the tape is this handle's.  The map is bounded: a caller that allocates
        // a fresh tape per step would otherwise grow it for ever, and a tape is gigabytes — nobody holds 64 live ones
        std::lock_guard<std::mutex> lk(m->tape_mu);
        if (m->tapes.size() >= 64) m->tapes.clear();
        m->tapes[tape] = ditto_model::TapeRec{B, N, T, hb};
    }
    return DITTO_OK;
}

int ditto_train_backward(ditto_model_t m, const ditto_weights* w, const float* grad_eps, const float* x,
                         const int64_t* t, int B, int N, int T, const float* rope_cos, const float* rope_sin,
                         float dropout_p, uint64_t seed, const void* tape, size_t tape_bytes, const ditto_grads* grads,
                         void* workspace, size_t workspace_bytes, ditto_stream_t stream) {
    return ditto_train_backward_opts(m, w, grad_eps, x, t, B, N, T, rope_cos, rope_sin, dropout_p, seed, tape, tape_bytes, grads,
                                     workspace, workspace_bytes, stream, nullptr);
}

int ditto_train_backward_opts(ditto_model_t m, const ditto_weights* w, const float* grad_eps, const float* x,
                              const int64_t* t, int B, int N, int T, const float* rope_cos, const float* rope_sin,
                              float dropout_p, uint64_t seed, const void* tape, size_t tape_bytes, const ditto_grads* grads,
                              void* workspace, size_t workspace_bytes, ditto_stream_t stream, const ditto_call_opts* opts) {
    if (!m) return fail(DITTO_ERR_ARG, "null model");
    return ditto_train_backward_layers(m, w, grad_eps, x, t, B, N, T, rope_cos, rope_sin, dropout_p, seed, tape, tape_bytes, grads,
                                       workspace, workspace_bytes, stream, opts, m->cfg.num_layers - 1, 0);
}

int ditto_train_backward_layers(ditto_model_t m, const ditto_weights* w, const float* grad_eps, const float* x,
                                const int64_t* t, int B, int N, int T, const float* rope_cos, const float* rope_sin,
                                float dropout_p, uint64_t seed, const void* tape, size_t tape_bytes, const ditto_grads* grads,
                                void* workspace, size_t workspace_bytes, ditto_stream_t stream, const ditto_call_opts* opts,
                                int layer_from, int layer_to) {
    if (int rc = check_call_opts(opts)) return rc;
    CallScope scope(opts);
    if (int rc = check_train(m)) return rc;
    if (layer_from >= m->cfg.num_layers || layer_to < 0 || layer_from < layer_to)
        return fail(DITTO_ERR_ARG, "ditto_train_backward_layers: need num_layers > layer_from >= layer_to >= 0");
    if (!w || !grad_eps || !x || !t || !rope_cos || !rope_sin || !tape || !grads || !grads->layers || !workspace ||
        B <= 0 || N <= 0 || T <= 0)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_train_backward");
    if (m->layersT.empty() || !m->WoutT) return fail(DITTO_ERR_ARG, "ditto_train_attach was not called on this handle");
    const ditto_config& c = m->cfg;
    const TapePlan tp = plan_tape(c, B, N, T);
    const TrainWsPlan wp = plan_train_ws(c, B, N, T);
    if (tape_bytes < tp.total) return fail(DITTO_ERR_SIZE, "tape too small: %zu < %zu", tape_bytes, tp.total);
    if (workspace_bytes < wp.total) return fail(DITTO_ERR_SIZE, "train workspace too small: %zu < %zu", workspace_bytes, wp.total);
    hipStream_t s = (hipStream_t)stream;
    const int d = c.hidden_dim, L = c.num_layers, H = c.num_heads, dhd = d / H, M = B * N, Mt = B * T, td = c.time_dim;
    const float scale = 1.0f / sqrtf((float)dhd);
    const bool lt0_has_fr = m->layersT[0].W1gTP != nullptr;
    const char* tb = (const char*)tape;
    char* ws = (char*)workspace;
    const char* kv = tb + tp.kv;
    const char* xcat = tb + tp.xcat;
    auto hs = [&](int i) { return (const float*)(tb + tp.hs + (size_t)i * tp.hs_stride); };
    float* dh = (float*)(ws + wp.dh);
    float* du = (float*)(ws + wp.du);
    char* dyb = ws + wp.dyb;
    char* big1 = ws + wp.big1;
    char* big2 = ws + wp.big2;
    char* dkv = ws + wp.dkv;
    const void* zero256 = ws + wp.zero;
    float* wtmp = (float*)(ws + wp.wtmp);
    float* vtmp = (float*)(ws + wp.vtmp);
    float* red = (float*)(ws + wp.red);
    float* dmod = (float*)(ws + wp.dmod);
    void* attn_ws = ws + wp.attn;
    float* wpart = (float*)(ws + wp.wpart);
    HIP_TRY(hipMemsetAsync(ws + wp.zero, 0, 256, s));

    // dW[n1, n2] = dY[rows, n1]^T X[rows, n2]: the K-major GEMM (gemm_tn.hip: operands read transposed out of LDS, no
    // transpose passes), split along K when the output has few tiles, partial tiles summed in slice order
    auto wgrad = [&](const void* dY, int ld1, int n1, const void* X, int ld2, int n2, int rows, float* out) -> int {
        // 256 x 256 tiles wherever the output has them (every weight of the d >= 256 configurations); gemm_flags bit 4096
        // keeps the 128 x 128 kernel (A/B)
        const bool wide = n1 >= 256 && n2 >= 256 && !(g_gemm_flags & GF_TN_NARROW);
        const long tiles = wide ? (long)((n1 + 255) / 256) * ((n2 + 255) / 256) : (long)((n1 + 127) / 128) * ((n2 + 127) / 128);
        const int S = wide ? wgrad_splits_wide(tiles, (rows + 63) / 64) : wgrad_splits(tiles, (rows + 63) / 64);
        if (S > 1 && (size_t)S * n1 * n2 * 4 <= WPART_BYTES) {
            HIP_TRY(launch_gemm_tn(dY, ld1, X, ld2, zero256, wpart, n2, n1, n2, rows, S, (size_t)n1 * n2, s, wide));
            HIP_TRY(launch_reduce_partials(wpart, S, (size_t)n1 * n2, out, s));
            return DITTO_OK;
        }
        HIP_TRY(launch_gemm_tn(dY, ld1, X, ld2, zero256, out, n2, n1, n2, rows, 1, 0, s, wide));
        return DITTO_OK;
    };
    // dX[M, n_in] = dY[M, n_out] * W, with Wt = W^T bf16 [n_in, n_out]
    auto dgrad = [&](const void* dY, int n_out, const void* Wt, int n_in, void* out, bool f32) -> int {
        GemmArgs g{};
        g.A = dY; g.lda = n_out; g.W = Wt; g.out = out; g.ldo = n_in; g.M = M; g.N = n_in; g.K = n_out;
        HIP_TRY(launch_gemm(g, f32 ? EPI_BIAS_F32 : EPI_BIAS_BF16, s));
        return DITTO_OK;
    };
    // LayerNorm backward into the stream gradient dh, which also leaves bf16(dh) in dyb (the dY operand of the segment below)
    // and, when asked, the column sums of dh = the bias gradient of the Linear that closes that segment
    // the same on the full-row kernel (N = d = 768 outputs, fp32, no bias / residual / LayerNorm), for the long-K dgrads: its
    // tile reads each dY row once (the 256 x 192 kernel re-reads the dY panel per column tile); WtP = Wt stage-major.
    // Option "fr_dgrad": bit 0 = the fc1|gate dgrad (K = 8d), bit 1 = the QKV dgrad (K = 3d).
    const bool fr_dgrad = lt0_has_fr && fr_pays(M) && gemm_fr_supports(M, d, 8 * d, (size_t)8 * d, (size_t)8 * d);
    // du — the gradient wrt a LayerNorm's output, written by a dgrad GEMM and read once by the LayerNorm backward — travels as
    // BF16 (train_flags 8: fp32, A/B): half the bytes on both sides; dh, the stream gradient it is folded into, stays fp32.
    // the tape's h rows are bf16 and the self-attention's O sits beside h1 — IF the forward that wrote this tape decided so
    bool hb = false;
    {
        std::lock_guard<std::mutex> lk(m->tape_mu);
        const auto it = m->tapes.find(tape);
        if (it == m->tapes.end())
            return fail(DITTO_ERR_ARG, "ditto_train_backward: no ditto_train_forward of this handle wrote the tape at %p", tape);
        if (it->second.B != B || it->second.N != N || it->second.T != T)
            return fail(DITTO_ERR_SHAPE, "ditto_train_backward: the tape was written for (B, N, T) = (%d, %d, %d), not (%d, %d, %d)",
                        it->second.B, it->second.N, it->second.T, B, N, T);
        hb = it->second.hb;
    }
    const bool du_bf16 = !(g_train_flags & 8);
    bool du_is_bf16 = false;   // what the LAST producer of du wrote (the 64-row full-row kernel has no bf16 output)
    auto dgrad_fr = [&](const void* dY, int n_out, const void* WtP, float* out) -> int {
        GemmParams gp{};
        gp.A = (const bf16*)dY; gp.lda = n_out; gp.W = (const bf16*)WtP; gp.ldw = n_out; gp.w_rows = d;
        gp.out = out; gp.ldo = d; gp.M = M; gp.N = d; gp.K = n_out;
        // bf16 output exists on gemm_frd.hip (kernels.h fr_launch_kernel == 130: every row count the full-row dgrads run at)
        const bool hb = du_bf16 && fr_launch_kernel(M, n_out) == 130;
        HIP_TRY(launch_gemm_fr(gp, nullptr, nullptr, nullptr, d, N % 128 == 0 ? N / 128 : 0, s, false, hb));
        du_is_bf16 = hb;
        return DITTO_OK;
    };
    auto ln_back = [&](const float* xin, const float* gamma, float* gw, float* gb, float* next_bias) -> int {
        if (hb && !du_is_bf16) return fail(DITTO_ERR_ARG, "internal: bf16 tape rows with an fp32 du");
        HIP_TRY(launch_ln_bwd_stream(du, du_is_bf16, xin, hb, gamma, dh, dyb, gw, gb, next_bias, red, M, d, s));
        return DITTO_OK;
    };
#define TRY_RC(expr) do { if (int _rc = (expr)) return _rc; } while (0)

    // ---- eps = [bf16(x) | bf16(h_L)] Wfin^T + (b_in + b_out)   (src/model/DiTTO.py:83,93-94) ----
    if (layer_from == L - 1) {   // the head of the backward belongs to the call that starts at the top layer
    HIP_TRY(launch_cast_bf16(grad_eps, dyb, (size_t)M * d, s));
    HIP_TRY(launch_colsum_f32(grad_eps, d, M, d, grads->proj_in_bias, red, s));
    HIP_TRY(hipMemcpyAsync(grads->proj_out_bias, grads->proj_in_bias, (size_t)d * 4, hipMemcpyDeviceToDevice, s));
    TRY_RC(wgrad(dyb, d, d, xcat, 2 * d, d, M, grads->proj_in_weight));
    TRY_RC(wgrad(dyb, d, d, xcat + (size_t)d * 2, 2 * d, d, M, grads->proj_out_weight));
    TRY_RC(dgrad(dyb, d, m->WoutT, d, dh, true));
    }

    // (a call that starts below the top layer continues on the stream gradient dh / its bf16 copy dyb that the call for the layers
    // above left in the workspace: same workspace, same stream, in order)
    for (int l = layer_from; l >= layer_to; --l) {
        const LayerPack& lp = m->layers[l];
        const LayerPackT& lt = m->layersT[l];
        const auto& q = tp.layers[l];
        const ditto_layer_grads& G = grads->layers[l];
        const float *h0 = hs(3 * l), *h1 = hs(3 * l + 1), *h2 = hs(3 * l + 2);

        // ---- gated MLP: h3 = h2 + act W2^T + b2,  act = gelu(a) sigmoid(g),  [a|g] = u3 W1g^T + b1g ----
        if (l == L - 1) {   // below the top layer the block above's last LayerNorm backward left both
            HIP_TRY(launch_cast_bf16(dh, dyb, (size_t)M * d, s));
            HIP_TRY(launch_colsum_f32(dh, d, M, d, G.mlp_fc2_bias, red, s));
        }
        TRY_RC(wgrad(dyb, d, d, tb + q.act, 4 * d, 4 * d, M, G.mlp_fc2_weight));
        if (!(g_train_flags & 2) && gemm_gated_bwd_fused_ok(M, 4 * d)) {
            // ONE launch: dact = dY W2 stays in the accumulators, the epilogue reads the pre-activations and writes [da | dg]
            // and the bias gradients' partial rows (gemm_common.h epilogue_gated_bwd; train_flags 2: A/B off)
            GemmArgs g{};
            g.A = dyb; g.lda = d; g.W = lt.W2T; g.out = big1; g.ldo = 8 * d; g.M = M; g.N = 4 * d; g.K = d;
            g.pre_bf16 = tb + q.pre; g.ldpre = 8 * d; g.colsum_partial = red;
            HIP_TRY(launch_gemm(g, EPI_GATED_BWD, s));
            HIP_TRY(launch_reduce_partials(red, 2 * ((M + 255) / 256), (size_t)8 * d, vtmp, s));
        } else {
            TRY_RC(dgrad(dyb, d, lt.W2T, 4 * d, big2, false));
            HIP_TRY(launch_gated_bwd(big2, tb + q.pre, big1, M, 4 * d, s, vtmp, red));   // + the packed fc1 | gate bias gradients
        }
        HIP_TRY(launch_unpack_vec(vtmp, G.mlp_fc1_bias, 4 * d, 16, 2, 0, s));
        HIP_TRY(launch_unpack_vec(vtmp, G.gate_bias, 4 * d, 16, 2, 16, s));
        TRY_RC(wgrad(big1, 8 * d, 8 * d, tb + q.u3, d, d, M, wtmp));
        HIP_TRY(launch_unpack_rows(wtmp, G.mlp_fc1_weight, 4 * d, d, 16, 2, 0, s));
        HIP_TRY(launch_unpack_rows(wtmp, G.gate_weight, 4 * d, d, 16, 2, 16, s));
        if (fr_dgrad && (g_fr_dgrad & 1)) TRY_RC(dgrad_fr(big1, 8 * d, lt.W1gTP, du));
        else { TRY_RC(dgrad(big1, 8 * d, lt.W1gT, d, du, !du_bf16)); du_is_bf16 = du_bf16; }
        TRY_RC(ln_back(h2, lp.g3, G.norm3_weight, G.norm3_bias, G.cross_out_proj_bias));

        // ---- cross-attention: h2 = h1 + oc Wo^T + bo,  oc = attn(qc, Kc, Vc),  qc = u2 Wq^T + bq ----
        TRY_RC(wgrad(dyb, d, d, tb + q.oc, d, d, M, G.cross_out_proj_weight));
        TRY_RC(dgrad(dyb, d, lt.WcoT, d, big2, false));
        {
            AttnBwdArgs a{};
            a.q = tb + q.qc; a.ldq = d; a.k = kv + (size_t)l * 2 * d * 2; a.ldk = L * 2 * d;
            a.v = kv + ((size_t)l * 2 * d + d) * 2; a.ldv = L * 2 * d; a.dout = big2; a.lddo = d;
            a.dq = big1; a.lddq = d; a.dk = dkv; a.lddk = 2 * d; a.dv = dkv + (size_t)d * 2; a.lddv = 2 * d;
            a.B = B; a.H = H; a.Sq = N; a.Skv = T; a.dh = dhd; a.scale = scale;
            a.dropout_p = dropout_p; a.seed = seed; a.layer = l; a.workspace = attn_ws; a.workspace_bytes = wp.attn_bytes;
            if (dhd == 64) { a.lse = (const float*)(tb + q.lse_c); a.o_bf16 = tb + q.oc; a.ldo = d; }
            HIP_TRY(launch_attention_bwd(a, s));
        }
        HIP_TRY(launch_colsum_bf16(big1, d, M, d, G.cross_in_proj_bias, red, s));
        HIP_TRY(launch_colsum_bf16(dkv, 2 * d, Mt, 2 * d, G.cross_in_proj_bias + d, red, s));
        TRY_RC(wgrad(big1, d, d, tb + q.u2, d, d, M, G.cross_in_proj_weight));
        TRY_RC(wgrad(dkv, 2 * d, 2 * d, tb + tp.text, c.text_dim, d, Mt, G.cross_in_proj_weight + (size_t)d * d));
        TRY_RC(dgrad(big1, d, lt.WcqT, d, du, !du_bf16));
        du_is_bf16 = du_bf16;
        TRY_RC(ln_back(h1, lp.g2, G.norm2_weight, G.norm2_bias, nullptr));

        // ---- self-attention: h1 = h0 + attn(rope(q), rope(k), v), NO out-proj (src/components/DiT.py:134-139) ----
        {
            const char* qkv = tb + q.qkv;
            AttnBwdArgs a{};
            a.q = qkv; a.ldq = 3 * d; a.k = qkv + (size_t)d * 2; a.ldk = 3 * d; a.v = qkv + (size_t)2 * d * 2; a.ldv = 3 * d;
            a.dout = dyb; a.lddo = d;
            a.dq = big1; a.lddq = 3 * d; a.dk = big1 + (size_t)d * 2; a.lddk = 3 * d; a.dv = big1 + (size_t)2 * d * 2;
            a.lddv = 3 * d; a.B = B; a.H = H; a.Sq = N; a.Skv = N; a.dh = dhd; a.scale = scale;
            a.workspace = attn_ws; a.workspace_bytes = wp.attn_bytes;
            if (dhd == 64 && hb) { a.lse = (const float*)(tb + q.lse_s); a.o_bf16 = (const char*)h1 + (size_t)M * d * 2; a.ldo = d; }
            else if (dhd == 64) { a.lse = (const float*)(tb + q.lse_s); a.h_after = h1; a.h_before = h0; a.ldh = d; }
            // the rotation's backward: in the dq / dk epilogues where the kernel can (head_dim 64), else one pass in place
            a.rope_cos = rope_cos; a.rope_sin = rope_sin;
            const bool fused_rope_bwd = (g_train_flags & 1) == 0 && attention_bwd_fuses_rope(a);
            if (!fused_rope_bwd) a.rope_cos = a.rope_sin = nullptr;
            HIP_TRY(launch_attention_bwd(a, s));
            if (!fused_rope_bwd) HIP_TRY(launch_rope_inplace(big1, 3 * d, rope_cos, rope_sin, M, N, 2 * d, dhd, s, -1.0f));
        }
        HIP_TRY(launch_colsum_bf16(big1, 3 * d, M, 3 * d, G.attn_in_proj_bias, red, s));
        TRY_RC(wgrad(big1, 3 * d, 3 * d, tb + q.u1, d, d, M, G.attn_in_proj_weight));
        if (fr_dgrad && (g_fr_dgrad & 2)) TRY_RC(dgrad_fr(big1, 3 * d, lt.WqkvTP, du));
        else { TRY_RC(dgrad(big1, 3 * d, lt.WqkvT, d, du, !du_bf16)); du_is_bf16 = du_bf16; }
        TRY_RC(ln_back(h0, lp.g1, G.norm1_weight, G.norm1_bias, l > 0 ? grads->layers[l - 1].mlp_fc2_bias : nullptr));
    }

    if (layer_to > 0) return DITTO_OK;   // the tail belongs to the call that ends at layer 0
    // ---- GlobalAdaLN (src/components/DiT.py:25-40): h0 = xhat (1 + s_t + s_x) + (b_t + b_x) ----
    // dmod[b] = [sum_n dh xhat | sum_n dh]: the gradient of BOTH branches' (scale, shift) vectors
    HIP_TRY(launch_ln_bwd(dh, x, nullptr, nullptr, dmod, red, N, B, d, s));
    const float* pooled = (const float*)(tb + tp.pooled);
    HIP_TRY(launch_small_linear_bwd_w(dmod, pooled, grads->ada_text_mlp_weight, grads->ada_text_mlp_bias, B,
                                      c.text_dim, 2 * d, true, s));
    {   // time branch: e0 = emb[t]; z1 = W0 e0 + b0; e2 = W2 silu(z1) + b2; mod_t = Wt silu(e2) + bt
        char* sm = ws + wp.small;
        const size_t st = al((size_t)B * td * 4);
        float *e0 = (float*)sm, *z1 = (float*)(sm + st), *e2 = (float*)(sm + 2 * st), *de2 = (float*)(sm + 3 * st),
              *dz1 = (float*)(sm + 4 * st), *de0 = (float*)(sm + 5 * st);
        HIP_TRY(launch_embedding_gather(w->t_embedding_weight, t, e0, B, c.diffusion_steps, td, s));
        HIP_TRY(launch_small_linear_fwd(e0, w->time_embed_0_weight, w->time_embed_0_bias, z1, B, td, td, false, s));
        HIP_TRY(launch_small_linear_fwd(z1, w->time_embed_2_weight, w->time_embed_2_bias, e2, B, td, td, true, s));
        HIP_TRY(launch_small_linear_bwd_w(dmod, e2, grads->ada_time_mlp_weight, grads->ada_time_mlp_bias, B, td, 2 * d, true, s));
        HIP_TRY(launch_small_linear_bwd_x(dmod, w->ada_time_mlp_weight, e2, de2, B, td, 2 * d, true, s));
        HIP_TRY(launch_small_linear_bwd_w(de2, z1, grads->time_embed_2_weight, grads->time_embed_2_bias, B, td, td, true, s));
        HIP_TRY(launch_small_linear_bwd_x(de2, w->time_embed_2_weight, z1, dz1, B, td, td, true, s));
        HIP_TRY(launch_small_linear_bwd_w(dz1, e0, grads->time_embed_0_weight, grads->time_embed_0_bias, B, td, td, false, s));
        HIP_TRY(launch_small_linear_bwd_x(dz1, w->time_embed_0_weight, e0, de0, B, td, td, false, s));
        HIP_TRY(hipMemsetAsync(grads->t_embedding_weight, 0, (size_t)c.diffusion_steps * td * 4, s));
        HIP_TRY(launch_embedding_scatter_add(de0, t, grads->t_embedding_weight, B, c.diffusion_steps, td, s));
    }
#undef TRY_RC
    return DITTO_OK;
}

size_t ditto_layernorm_bwd_scratch_bytes(int rows_per_group, int groups, int d) {
    if (rows_per_group <= 0 || groups <= 0 || d <= 0) return 0;
    return ln_bwd_scratch_bytes(rows_per_group, groups, d);
}
int ditto_layernorm_bwd(const float* dy, const float* x, const float* gamma, float* dx_accum, float* dgamma_dbeta,
                        void* scratch, size_t scratch_bytes, int rows_per_group, int groups, int d,
                        ditto_stream_t stream) {
    if (!dy || !x || rows_per_group <= 0 || groups <= 0 || (!dx_accum && !dgamma_dbeta))
        return fail(DITTO_ERR_ARG, "bad argument to ditto_layernorm_bwd");
    if (d % 4 || d > 2048) return fail(DITTO_ERR_SHAPE, "d must be a multiple of 4 and <= 2048");
    if (dgamma_dbeta && (!scratch || scratch_bytes < ln_bwd_scratch_bytes(rows_per_group, groups, d)))
        return fail(DITTO_ERR_SIZE, "scratch too small for ditto_layernorm_bwd");
    HIP_TRY(launch_ln_bwd(dy, x, gamma, dx_accum, dgamma_dbeta, (float*)scratch, rows_per_group, groups, d,
                          (hipStream_t)stream));
    return DITTO_OK;
}

size_t ditto_attention_bwd_workspace_bytes(int B, int H, int Sq, int Skv, int dh) {
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0 || dh <= 0) return 0;
    return attention_train_workspace_bytes(B, H, Sq, Skv, dh);
}
int ditto_attention_bwd_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout,
                             int lddo, const void* out, int ldo, const float* lse, void* dq, int lddq, void* dk, int lddk,
                             void* dv, int lddv, int B, int H, int Sq, int Skv, int dh, float scale, float dropout_p,
                             uint64_t seed, int layer, void* workspace, size_t workspace_bytes, ditto_stream_t stream) {
    if (!q || !k || !v || !dout || !dq || !dk || !dv || !workspace)
        return fail(DITTO_ERR_ARG, "null pointer to ditto_attention_bwd_bf16");
    if (dh % 64) return fail(DITTO_ERR_SHAPE, "head_dim must be a multiple of 64");
    if ((ldq | ldk | ldv | lddo | lddq | lddk | lddv) % 8) return fail(DITTO_ERR_SHAPE, "row strides must be multiples of 8");
    if (workspace_bytes < attention_train_workspace_bytes(B, H, Sq, Skv, dh))
        return fail(DITTO_ERR_SIZE, "attention backward workspace too small");
    if (lse && (dh != 64 || !out)) return fail(DITTO_ERR_ARG, "the fused backward (lse given) needs head_dim 64 and `out`");
    AttnBwdArgs a{};
    a.lse = lse; a.o_bf16 = out; a.ldo = ldo;
    a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.dout = dout; a.lddo = lddo;
    a.dq = dq; a.lddq = lddq; a.dk = dk; a.lddk = lddk; a.dv = dv; a.lddv = lddv;
    a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.dh = dh; a.scale = scale;
    a.dropout_p = dropout_p; a.seed = seed; a.layer = layer; a.workspace = workspace; a.workspace_bytes = workspace_bytes;
    HIP_TRY(launch_attention_bwd(a, (hipStream_t)stream));
    return DITTO_OK;
}
int ditto_attention_dropout_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out,
                                 int ldo, float* lse_out, int B, int H, int Sq, int Skv, int dh, float scale,
                                 float dropout_p, uint64_t seed, int layer, void* workspace, size_t workspace_bytes,
                                 ditto_stream_t stream) {
    if (!q || !k || !v || !out || !workspace) return fail(DITTO_ERR_ARG, "null pointer to ditto_attention_dropout_bf16");
    if (dh % 64) return fail(DITTO_ERR_SHAPE, "head_dim must be a multiple of 64");
    if (workspace_bytes < attention_train_workspace_bytes(B, H, Sq, Skv, dh))
        return fail(DITTO_ERR_SIZE, "attention workspace too small");
    if (lse_out && dh != 64) return fail(DITTO_ERR_ARG, "lse_out is produced by the fused head_dim-64 kernel only");
    AttnArgs a{};
    a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.out_bf16 = out; a.ldo = ldo;
    a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.dh = dh; a.scale = scale;
    a.workspace = workspace; a.workspace_bytes = workspace_bytes;
    a.dropout_p = dropout_p; a.seed = seed; a.layer = layer; a.force_generic = lse_out == nullptr;
    a.lse_out = lse_out;
    HIP_TRY(launch_attention(a, (hipStream_t)stream));
    return DITTO_OK;
}

}  // extern "C"
