// ditto_api.hip — the C-ABI of libditto_hip.so (include/ditto_hip.h): model handle, weight packing,
// the stream-ordered launch sequence of one DiTTO.forward, the sampler update.
//
// HBM layout (one contiguous caller-owned arena per model, one workspace per (B,N,T), one cond buffer
// per utterance batch).  All offsets 256-byte aligned.
//   arena     : per layer { Wqkv bf16[3d,d] | Wcq bf16[d,d] | Wco bf16[d,d] | W1g bf16[8d,d] (fc1/gate rows
//               interleaved in blocks of 16) | W2 bf16[d,4d] | biases + LN gamma/beta fp32 },
//               Wkv_all bf16[L*2d, d] (cross-attn K,V projections of every layer), Wfin bf16[d, 2d]
//               ([proj_in | proj_out] along K), time table fp32[steps, 2d], text_mlp fp32, inv_freq.
//   cond      : kv_cache bf16[B*T, L*2d] (layer l: K at cols l*2d.., V at l*2d+d..) | tmod fp32[B, 2d]
//   workspace : h fp32[M,d] (residual stream) | u bf16[M,d] | qkv bf16[M,3d] | act bf16[M,4d] |
//               xcat bf16[M,2d] ([bf16(x_raw) | bf16(h_L)]) | eps fp32[M,d] | attention scratch
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <memory>
#include <new>
#include <vector>

#include "model.h"
#include "gemm_common.h"

using namespace ditto;

namespace ditto {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

ArenaPlan plan_arena(const ditto_config& c) {
    ArenaPlan p;
    const size_t d = c.hidden_dim, L = c.num_layers;
    const size_t dp = cfg_dp(c);                                      // attention-side width (= d unless heads are padded)
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    p.layers.resize(L);
    for (size_t l = 0; l < L; ++l) {
        auto& q = p.layers[l];
        const size_t we = (c.flags & DITTO_CFG_FP8_LINEAR) ? 1 : 2;   // bytes per element of the fp8-able weights
        q.Wqkv = take(3 * dp * d * we); q.Wcq = take(dp * d * 2); q.Wco = take(d * dp * 2);
        q.W1g = take(8 * d * d * we); q.W2 = take(4 * d * d * we);
        q.sqkv = take(3 * dp * 4); q.s1g = take(8 * d * 4); q.s2 = take(d * 4);   // fp8 per-row weight scales
        q.bqkv = take(3 * dp * 4); q.bcq = take(dp * 4); q.bco = take(d * 4); q.b1g = take(8 * d * 4); q.b2 = take(d * 4);
        q.g1 = take(d * 4); q.be1 = take(d * 4); q.g2 = take(d * 4); q.be2 = take(d * 4); q.g3 = take(d * 4);
        q.be3 = take(d * 4);
        // stage-major bf16 copies of the N = d projections for the full-row kernels (gemm_fr.hip: d == 768, bf16 linears;
        // gemm_fr64.hip: d == 1024 — the cross out-projection is bf16 in the fp8 configuration too, fc2 only without it)
        const bool fp8c = (c.flags & DITTO_CFG_FP8_LINEAR) != 0;
        const bool pad = dp != d;                                    // padded heads: the tiled GEMMs + LayerNorm launches only
        const bool fr_o = !pad && ((d == 768 && !fp8c) || d == 1024), fr_2 = !pad && (d == 768 || d == 1024) && !fp8c;
        q.WcoP = fr_o ? take(d * d * 2) : 0; q.W2P = fr_2 ? take(4 * d * d * 2) : 0;
        // ... and of the cross q-projection for the fused norm2 + q-projection kernel (gemm_lnq.hip, d == 768): both MFMA shapes' images
        const bool lnq = !pad && ((d == 768 && !fp8c) || d == 1024);   // (the q-projection is a bf16 GEMM in the fp8 configuration too)
        q.WcqP = lnq ? take(d * d * 2) : 0; q.WcqP32 = lnq && d == 768 ? take(d * d * 2) : 0;
    }
    p.Wkv = take(L * 2 * dp * d * 2); p.bkv = take(L * 2 * dp * 4);
    p.Wfin = take(d * 2 * d * 2); p.bfin = take(d * 4);
    p.ttab = take((size_t)c.diffusion_steps * 2 * d * 4);
    p.wx = take(2 * d * (size_t)c.text_dim * 4); p.bx = take(2 * d * 4);
    p.invf = take((d / c.num_heads / 2) * 4);
    p.invf_rev = take((d / c.num_heads / 2) * 4);
    p.total = off;
    return p;
}

// ditto_text_precompute scratch (front of the same workspace): bf16(text) | pooled fp32
static inline size_t text_scratch_bytes(const ditto_config& c, int B, int T) {
    return al((size_t)B * T * c.text_dim * 2) + al((size_t)B * c.text_dim * 4);
}

// Small batches leave the long-K GEMMs of the step on a few dozen workgroups, each walking all of K alone (B = 1:
// fc2 is 48 tiles x 48 K-tiles = 41 us for 4.8 GFLOP).  Those run split-K: `splits` workgroups per output tile write
// fp32 partials, launch_splitk_finish adds them in order together with bias and residual.  Target: g_splitk_wgs
// workgroups in flight, at least 4 K-tiles per split.  Measured in-model (tools/step_ab.py, C2): B = 1 fc2 40.6 ->
// 24.2 us, step 1.87 -> 1.68 ms at 256; B = 2 2.12 -> 2.06 ms; B >= 4 never splits.
// OFF by default (ditto_set_option("splitk_wgs", 256) / DITTO_SPLITK_WGS=256 turn it on): the K-partition depends on
// the batch size, so with it an utterance's result is no longer bit-identical across batch compositions — the
// property tests/test_gpu_model.py::test_full_size_c2_properties holds the default path to.
static int g_splitk_wgs = [] { const char* e = getenv("DITTO_SPLITK_WGS"); return e ? atoi(e) : 0; }();
// "residual_bf16": the residual stream h between the segments of a block lives in HBM as bf16 (fp32 only inside accumulators
// and LayerNorm statistics) wherever the launch takes the full-row class at d = 768 / head_dim 64 (ditto_forward decides)
// "ll_mask": the low-latency class's launch fusions (they change no bit): bit 0 = fc2's split-K finish also writes the NEXT
// block's norm1, bit 1 = the cross out-projection runs as two K-splits whose finish also writes norm3
// (bit 2, round 5: split-KV attention + ordered merge — measured slower at B = 1, 1.59 -> 1.72 ms, profiles/r05_splitkv_ab.txt — was deleted in round 6)
int g_ll_mask = [] { const char* e = getenv("DITTO_LL_MASK"); const int v = e ? atoi(e) : 3; return v < 0 || v > 3 ? 3 : v; }();
// "lnq": norm2 fused into the cross-attention q-projection (gemm_lnq.hip) for launches of the full-row class at d = 768:
// 0 = off (LayerNorm launch + tiled GEMM), 32 / 16 = on, with that MFMA shape (32x32x16 / 16x16x32)
// "lnq_min_rows": > 0: the fused norm2 + q-projection also runs BELOW the full-row class, from that many (class) rows on (A/B)
// Default 8192 since round 5: with the kernel on two waves per SIMD the fused launch wins from 128 of its 64-row tiles on
// (profiles/r05_small_batch_lnq8.txt, same process: B = 8 4.02 -> 3.95 ms, B = 16 6.63 -> 6.47; B = 4 2.61 -> 2.63, hence not lower).
int g_lnq_min_rows = [] { const char* e = getenv("DITTO_LNQ_MIN_ROWS"); return e ? atoi(e) : 8192; }();
int g_lnq = [] { const char* e = getenv("DITTO_LNQ"); const int v = e ? atoi(e) : 32; return v == 0 || v == 16 || v == 32 ? v : 32; }();   // validated like ditto_set_option
int g_resid_bf16 = [] { const char* e = getenv("DITTO_RESIDUAL_BF16"); return e ? (atoi(e) != 0) : 1; }();   // normalised to 0 / 1
// "qkv_split" (round 5): 0 = off; n > 0 = the QKV GEMM's last 256 columns as a second launch where that leaves whole rounds of
// 256 x 256 tiles (run_block), for launches of at most n rounds of such tiles.  Default 3 = B = 8 and B = 16 at N = 1024
// (profiles/r05_qkv_split_ab.txt, same process: QKV 48.6 -> 44.4 us / 75.6 -> 72.3, step 3.87 -> 3.83 / 6.44 -> 6.42 ms; B = 24 and
// B = 32 — 3.4 and 4.5 rounds — do not gain: the small-tile tail launch costs what the half round did).  Bit-identical either way.
int g_qkv_split = [] { const char* e = getenv("DITTO_QKV_SPLIT"); return e ? atoi(e) : 3; }();
int small_batch_k_splits(int M, int N, int K) {
    if (g_splitk_wgs < 0) return 1;                        // -1: never split
    const int ktiles = K / 64;
    if (g_splitk_wgs == 0) {
        // Default since round 4: the LOW-LATENCY CLASS.  A batch of at most 2048 rows (1-2 utterances at N = 1024: the reference's
        // own call pattern, Experiments.ipynb:125-132) leaves its long-K GEMMs (fc2: 48 tiles x 48 K-tiles at B = 1; the final
        // projection) on a fifth of the chip; they run split over K with an ordered fp32 reduce (bias + residual in the reduce):
        // fc2 40.8 -> 24.2 us, step 1.83 -> 1.63 ms at C2 B = 1.  The number of splits depends on K ONLY, so inside the class an
        // utterance's bits do not depend on its batch neighbours; like the full-row class (kernels.h fr_rule_rows) the class is a
        // function of the launch's rows and a caller that splits one batch pins it ("fr_class_rows"): the rows of the UNSPLIT batch.
        const long rows = opt_class_rows() > 0 ? opt_class_rows() : M;
        if (rows > 2048 || ktiles < 24) return 1;
        const int ns = ktiles / 12;                        // K = 3072: 4, 1536: 2, 4096: 5, 2048: 2
        return ns > 8 ? 8 : ns;
    }
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);   // > 0: the explicit rule of rounds 1-3 (a workgroup target)
    if (tiles > 256 || ktiles < 16) return 1;
    long ns = g_splitk_wgs / tiles;
    if (ns > ktiles / 4) ns = ktiles / 4;
    if (ns > 8) ns = 8;
    return ns < 2 ? 1 : (int)ns;
}

// The cross out-projection (K = d) in the low-latency class: two K-splits (its 48 tiles at B = 1 become 96 workgroups of half the
// depth) whose finish also writes norm3 — the LayerNorm launch behind it disappears.  Same class rule as above, K only.
int small_batch_k_splits_outproj(int M, int K) {
    if (g_splitk_wgs != 0 || !(g_ll_mask & 2)) return 1;
    const long rows = opt_class_rows() > 0 ? opt_class_rows() : M;
    return (rows <= 2048 && K / 64 >= 12) ? 2 : 1;
}

WsPlan plan_ws(const ditto_config& c, int B, int N, int T) {
    WsPlan w;
    const size_t d = c.hidden_dim, M = (size_t)B * N, dh = cfg_dhp(c), dp = cfg_dp(c);   // dh: the PHYSICAL head width
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    w.h = take(M * d * 4); w.u = take(M * dp * 2); w.qkv = take(M * 3 * dp * 2); w.act = take(M * 4 * d * 2);
    w.xcat = take(M * 2 * d * 2); w.eps = take(M * d * 4);
    const size_t a1 = attention_workspace_bytes(B, c.num_heads, N, N, (int)dh);
    const size_t a2 = attention_workspace_bytes(B, c.num_heads, N, T, (int)dh);
    w.attn_bytes = a1 > a2 ? a1 : a2;
    w.attn = take(w.attn_bytes);
    // split-K partials of fc2 / the final projection at small batch: sized for the most splits the option allows
    w.splitk_bytes = (long)((M + 127) / 128) * ((d + 127) / 128) <= 256 ? 8 * M * d * 4 : 0;
    w.splitk = take(w.splitk_bytes);
    const size_t tneed = text_scratch_bytes(c, B, T);
    w.total = off > tneed ? off : tneed;
    return w;
}

int check_cfg(const ditto_config* c) {
    if (!c) return fail(DITTO_ERR_ARG, "config is null");
    if (c->hidden_dim <= 0 || c->num_layers <= 0 || c->num_heads <= 0 || c->time_dim <= 0 || c->diffusion_steps <= 0)
        return fail(DITTO_ERR_SHAPE, "non-positive config field");
    if (c->hidden_dim % c->num_heads) return fail(DITTO_ERR_SHAPE, "hidden_dim %% num_heads != 0");
    if (c->text_dim != c->hidden_dim)
        return fail(DITTO_ERR_SHAPE, "text_dim (%d) must equal hidden_dim (%d): reference cross-attn has no kdim/vdim",
                    c->text_dim, c->hidden_dim);
    if (c->hidden_dim % 64) return fail(DITTO_ERR_SHAPE, "hidden_dim must be a multiple of 64 (MFMA K tile)");
    if (c->hidden_dim > 2048) return fail(DITTO_ERR_SHAPE, "hidden_dim > 2048 not supported by the LayerNorm kernel");
    const int dh = c->hidden_dim / c->num_heads;
    if (dh % 2) return fail(DITTO_ERR_SHAPE, "head_dim (%d) must be even (half-split RoPE, src/components/DiT.py:52-54)", dh);
    // head_dim % 64 != 0 (the reference takes any hidden_dim % num_heads == 0, src/components/DiT.py:78-86; e.g. 1152 / 16 = 72)
    // runs with the heads PADDED to the next multiple of 64 inside the block (model.h cfg_dhp): inference only, bf16 linears
    if (dh % 64 && (c->flags & DITTO_CFG_FP8_LINEAR))
        return fail(DITTO_ERR_SHAPE, "fp8 linear layers need head_dim %% 64 == 0 (head_dim %d)", dh);
    if (dh % 64 && 4 * c->hidden_dim < cfg_dp(*c))
        return fail(DITTO_ERR_SHAPE, "head_dim %d pads to %d: more than 4x hidden_dim of attention width is not supported", dh, cfg_dhp(*c));
    if ((c->flags & DITTO_CFG_FP8_LINEAR) && c->hidden_dim % 128)
        return fail(DITTO_ERR_SHAPE, "fp8 linear layers need hidden_dim %% 128 == 0");
    if (c->flags & ~DITTO_CFG_FP8_LINEAR) return fail(DITTO_ERR_ARG, "unknown config flag");
    return DITTO_OK;
}

// A pinned kernel class (fr_class_rows > 0: "decide as the unsplit batch of that many rows would") that says full-row while THIS
// launch cannot run a full-row kernel (fewer rows than one 64-row tile) would silently take the tiled GEMMs, whose last bits
// differ: the promise of the pin — sharding changes no bit — would be broken without a sign.  Fail instead.
int check_class_pin(int M, int d, bool fp8, bool has_fr) {   // has_fr: the model has full-row kernels at all (not on padded heads)
    if (!has_fr || opt_class_rows() <= 0 || !opt_fr_mask() || M >= 64) return DITTO_OK;
    const bool wants = d == 768 ? (!fp8 && fr_rule_rows(opt_class_rows()) != 0) : (d == 1024 && fr_pays_64(M));
    if (!wants) return DITTO_OK;
    return fail(DITTO_ERR_SHAPE, "kernel class pinned to a batch of %d rows (full-row GEMM + LayerNorm kernels), but this launch has "
                                 "%d rows, fewer than one 64-row tile: it cannot take that class and its bits would differ from the "
                                 "unsplit batch's.  Give every shard at least 64 rows, or run the whole batch unfused "
                                 "(ditto_set_option(\"fr_mask\", 0)).", opt_class_rows(), M);
}

int check_call_opts(const ditto_call_opts* o) {
    if (!o) return DITTO_OK;
    if (o->class_rows < -1) return fail(DITTO_ERR_ARG, "ditto_call_opts.class_rows must be >= -1 (-1 inherit, 0 the launch's own rows)");
    if (o->residual_bf16 < -1 || o->residual_bf16 > 1) return fail(DITTO_ERR_ARG, "ditto_call_opts.residual_bf16 must be -1, 0 or 1");
    if (o->fr_mask < -1 || o->fr_mask > 3) return fail(DITTO_ERR_ARG, "ditto_call_opts.fr_mask must be in [-1, 3]");
    if (o->lnq != -1 && o->lnq != 0 && o->lnq != 16 && o->lnq != 32) return fail(DITTO_ERR_ARG, "ditto_call_opts.lnq must be -1, 0, 16 or 32");
    for (int r : o->reserved) if (r) return fail(DITTO_ERR_ARG, "ditto_call_opts.reserved must be zero");
    return DITTO_OK;
}

}  // namespace ditto

namespace {

// roctx ranges per kernel class (DITTO_ROCTX=1): resolved once from the profiler's marker library; absent library or
// unset variable = no-ops.  The ranges are host-side brackets around the enqueue, which is what rocprofv3's marker
// trace correlates kernel dispatches with.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("DITTO_ROCTX");
        if (!e || !*e || *e == '0') return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
static Roctx g_roctx;

struct ProfScope {
    ditto_model* m; hipStream_t s; int kc; hipEvent_t a = nullptr, b = nullptr;
    bool marked = false;
    ProfScope(ditto_model* m_, hipStream_t s_, int kc_) : m(m_), s(s_), kc(kc_) {
        if (g_roctx.push) { g_roctx.push(ditto_kernel_class_name(kc)); marked = true; }
        if (!m->prof) return;
        a = take(); b = take();
        if (a) (void)hipEventRecord(a, s);
    }
    hipEvent_t take() {
        if (!m->pool.empty()) { hipEvent_t e = m->pool.back(); m->pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    ~ProfScope() {
        if (marked) g_roctx.pop();
        if (!m->prof || !a || !b) return;
        (void)hipEventRecord(b, s);
        m->recs.push_back({a, b, kc});
    }
};

}  // namespace

// One DiT block (reference src/components/DiT.py:100-157) on the fp32 residual stream `h`, in place.
static int run_block(ditto_model* m, int l, float* h, void* u, char* qkv, void* act, char* xcat_or_null,
                     void* attn_ws, size_t attn_ws_bytes, float* splitk_ws, size_t splitk_bytes, const char* kv, int kv_layer, int kv_ld,
                     const float* rope_cos, const float* rope_sin, int B, int N, int T, hipStream_t s,
                     float* tap_self = nullptr, float* tap_cross = nullptr, bool ln1_done = false,
                     const float* next_g1 = nullptr, const float* next_be1 = nullptr, bool hb = false) {
    const ditto_config& c = m->cfg;
    const int d = c.hidden_dim, H = c.num_heads, dh = d / H, M = B * N;
    const float scale = 1.0f / sqrtf((float)dh);
    const bool fused_rope = (dh == 64);
    // padded heads (head_dim % 64 != 0): q / k / v and the attention outputs are dp = H * dhp wide, heads at a stride of dhp
    // (zero pads: the packed weights have zero rows / columns there), the attention runs the GEMM-composed path on dhp-wide
    // heads with the TRUE head_dim's softmax scale; the self-attention's head merge + residual is a compaction pass
    const int dhp = cfg_dhp(c), dp = cfg_dp(c);
    const bool pad = dp != d;
    const bool fp8 = (c.flags & DITTO_CFG_FP8_LINEAR) != 0;   // u / act hold fp8 bytes for the fp8 GEMMs
    const LayerPack& lp = m->layers[l];
    // full-row GEMM with the residual add and the FOLLOWING LayerNorm fused (gemm_fr.hip): bit 0 = cross out-proj + norm3,
    // bit 1 = fc2 + the next block's norm1 (ln1_done tells that block its norm1 output is already in u)
    // (the stage-major weight copies exist exactly where a kernel exists: d = 768 bf16, d = 1024 — there the out-projection
    // also in the fp8 configuration, with the LayerNorm output written as fp8)
    const bool fr_out = lp.WcoP && (opt_fr_mask() & 1) && fr_outproj_ok(M, d);
    const bool fr_fc2 = !fp8 && lp.W2P && (opt_fr_mask() & 2) && fr_fc2_ok(M, d);
    const int fr_rot = N % 128 == 0 ? N / 128 : 0;   // tiles per utterance: the K-loop rotation period (gemm_fr.hip)
    if (int rc = ditto::check_class_pin(M, d, fp8, !pad)) return rc;
    // hb: `h` holds BF16 rows (the bf16 residual stream; ditto_forward decides, and only where both fused launches run)
    if (hb && (!fr_out || !fr_fc2 || dh != 64 || tap_self || tap_cross)) return fail(DITTO_ERR_ARG, "internal: bf16 stream outside its class");
        // ---- self-attention (src/components/DiT.py:103-139) ----
        if (!ln1_done) {
            ProfScope ps(m, s, DITTO_KC_LAYERNORM);
            if (fp8) HIP_TRY(launch_layernorm_fp8(h, lp.g1, lp.be1, u, d, M, d, s));
            else if (hb) HIP_TRY(launch_layernorm_xbf16(h, lp.g1, lp.be1, u, d, M, d, s));
            else HIP_TRY(launch_layernorm(h, lp.g1, lp.be1, u, d, M, d, s));
        }
        {
            ProfScope ps(m, s, DITTO_KC_GEMM_QKV);
            GemmArgs g{};
            g.A = u; g.lda = d; g.W = lp.Wqkv; g.bias = lp.bqkv; g.out = qkv; g.ldo = 3 * dp;
            g.M = M; g.N = 3 * dp; g.K = d;
            g.rope_cos = rope_cos; g.rope_sin = rope_sin; g.rope_rows_per_batch = N; g.rope_cols = 2 * d;
            if (fused_rope && !(g_gemm_flags & 1024)) g.rope_freq_rev = m->invf_rev;   // flag 1024: A/B, table loads
            g.fp8 = fp8; g.wscale = fp8 ? lp.sqkv : nullptr;
#ifdef DITTO_DIAG_QKV_PLAIN   // tools/build_diag.sh: the QKV GEMM with the plain bias epilogue (NO RoPE: wrong results) — what
                              // would the 256 x 192 kernel (whole tile rounds at M = 32768; gemm_tile 192) buy this class?
            HIP_TRY(launch_gemm(g, EPI_BIAS_BF16, s));
#else
            // Whole rounds (round 5, "qkv_split"): where the 256 x 256 tiles of the QKV GEMM leave a fractional round of the CUs but
            // the tiles of all but its LAST 256 columns make whole rounds (d = 768: 9 column tiles, M a multiple of 8192 rows:
            // B = 8 is 288 tiles = 1.125 rounds, i.e. two rounds' time), those last columns — all of them v columns, plain bias
            // epilogue — run as a second launch on the small-tile kernel.  Every tiled kernel multiplies with the same MFMA in
            // the same K order: the same bits as the single launch (test).
            const int q_tm = (M + 255) / 256, q_tn = (3 * dp) / 256;
            const bool q_split = g_qkv_split && !fp8 && fused_rope && g_gemm_tile == 0 && (3 * dp) % 256 == 0 && 3 * dp >= 2048 + 256 &&
                                 2 * d <= 3 * dp - 256 && (long)q_tm * q_tn >= 144 && ((long)q_tm * q_tn) % 256 != 0 &&
                                 ((long)q_tm * (q_tn - 1)) % 256 == 0 && (long)q_tm * q_tn <= (long)g_qkv_split * 256;
            if (q_split) {
                GemmArgs gm = g, gt = g;
                gm.N = 3 * dp - 256;
                HIP_TRY(launch_gemm(gm, EPI_QKV_ROPE, s));
                gt.N = 256; gt.W = (const char*)lp.Wqkv + (size_t)gm.N * d * 2; gt.bias = lp.bqkv + gm.N;
                gt.out = qkv + (size_t)gm.N * 2; gt.rope_cols = 0; gt.rope_freq_rev = nullptr;
                HIP_TRY(launch_gemm(gt, EPI_BIAS_BF16, s));
            } else {
            HIP_TRY(launch_gemm(g, fused_rope ? EPI_QKV_ROPE : EPI_BIAS_BF16, s));
            if (!fused_rope) HIP_TRY(launch_rope_inplace(qkv, 3 * dp, rope_cos, rope_sin, M, N, 2 * dp, dh, s, 1.0f, dhp));
            }
#endif
        }
        // norm2 + q-projection in one launch (gemm_lnq.hip) for the full-row class at d = 768; else LayerNorm launch + tiled GEMM
        const int rows_cls = opt_class_rows() > 0 ? opt_class_rows() : M;
        // (d = 1024, BASELINE config C5: the same kernel at that width, 32x32x16 only, from 192 tiles of 64 rows on — the rule of
        // its full-row GEMM)
        const bool lnq = opt_lnq() && lp.WcqP && ((d == 768 && (fr_pays(M) || (g_lnq_min_rows > 0 && rows_cls >= g_lnq_min_rows))) ||
                                              (d == 1024 && fr_pays_64(M)));
        {
            ProfScope ps(m, s, DITTO_KC_ATTN_SELF);
            AttnArgs a{};
            a.q = qkv; a.ldq = 3 * dp; a.k = qkv + (size_t)dp * 2; a.ldk = 3 * dp; a.v = qkv + (size_t)2 * dp * 2;
            a.ldv = 3 * dp; a.B = B; a.H = H; a.Sq = N; a.Skv = N; a.dh = dhp;
            a.scale = scale; a.workspace = attn_ws; a.workspace_bytes = attn_ws_bytes; a.q_prescaled = (dh == 64);
            if (pad) {   // O at the padded stride into `act` (free here), then h[:, hd dh + c] += O[:, hd dhp + c]
                a.out_bf16 = act; a.ldo = dp;
                HIP_TRY(launch_attention(a, s));
                HIP_TRY(launch_head_compact_add(act, dp, h, d, M, d, dh, dhp, s));
            } else {
                a.resid_f32 = h; a.ldr = d; a.resid_bf16 = hb;
                HIP_TRY(launch_attention(a, s));
            }
        }
        if (tap_self) HIP_TRY(hipMemcpyAsync(tap_self, h, (size_t)M * d * 4, hipMemcpyDeviceToDevice, s));
        // ---- cross-attention (src/components/DiT.py:141-148), K/V from the per-utterance cache ----
        if (lnq) {
            ProfScope ps(m, s, DITTO_KC_GEMM_QPROJ);
            const int shape = d == 768 ? opt_lnq() : 32;
            HIP_TRY(launch_gemm_lnq(h, d, hb, lp.g2, lp.be2, shape == 16 ? lp.WcqP32 : lp.WcqP, lp.bcq, qkv, d, M, d, shape,
                                    N % 64 == 0 ? N / 64 : 0, s));
        } else {
            {
                ProfScope ps(m, s, DITTO_KC_LAYERNORM);
                if (hb) HIP_TRY(launch_layernorm_xbf16(h, lp.g2, lp.be2, u, d, M, d, s));
                else HIP_TRY(launch_layernorm(h, lp.g2, lp.be2, u, d, M, d, s));
            }
            ProfScope ps(m, s, DITTO_KC_GEMM_QPROJ);
            GemmArgs g{};
            g.A = u; g.lda = d; g.W = lp.Wcq; g.bias = lp.bcq; g.out = qkv; g.ldo = dp; g.M = M; g.N = dp; g.K = d;
            HIP_TRY(launch_gemm(g, EPI_BIAS_BF16, s));
        }
        {
            ProfScope ps(m, s, DITTO_KC_ATTN_CROSS);
            AttnArgs a{};
            a.q = qkv; a.ldq = dp; a.k = kv + (size_t)kv_layer * 2 * dp * 2; a.ldk = kv_ld;
            a.v = kv + ((size_t)kv_layer * 2 * dp + dp) * 2; a.ldv = kv_ld; a.out_bf16 = u; a.ldo = dp;
            a.B = B; a.H = H; a.Sq = N; a.Skv = T; a.dh = dhp; a.scale = scale;
            a.workspace = attn_ws; a.workspace_bytes = attn_ws_bytes; a.q_prescaled = (dh == 64);
            HIP_TRY(launch_attention(a, s));
        }
        bool ln3_done = false;
        if (fr_out) {
            // out-proj + residual + norm3 in one launch.  A = the attention output in u; the LayerNorm output goes to the first
            // M x d of the qkv buffer (the cross q it held has been consumed) because u is still being read as the A operand
            ProfScope ps(m, s, DITTO_KC_GEMM_OUTPROJ);
            GemmParams gp{};
            gp.A = (const bf16*)u; gp.lda = d; gp.W = (const bf16*)lp.WcoP; gp.ldw = d; gp.w_rows = d; gp.bias = lp.bco;
            gp.residual = h; gp.ldr = d; gp.out = h; gp.ldo = d; gp.M = M; gp.N = d; gp.K = d;
            HIP_TRY(launch_gemm_fr(gp, lp.g3, lp.be3, qkv, d, fr_rot, s, fp8, hb));
        } else {
            ProfScope ps(m, s, DITTO_KC_GEMM_OUTPROJ);
            GemmArgs g{};
            g.A = u; g.lda = dp; g.W = lp.Wco; g.bias = lp.bco; g.residual = h; g.ldr = d; g.out = h; g.ldo = d;
            g.M = M; g.N = d; g.K = dp;
            const int nso = fp8 ? 1 : small_batch_k_splits_outproj(M, dp);
            if (nso > 1 && splitk_ws && splitk_bytes >= (size_t)nso * M * d * 4) {
                // low-latency class: K-splits + a finish that adds bias + residual AND writes norm3 (into qkv: u is the A operand)
                g.bias = nullptr; g.residual = nullptr; g.out = splitk_ws; g.k_splits = nso; g.split_stride = (size_t)M * d;
                HIP_TRY(launch_gemm(g, EPI_BIAS_F32, s));
                HIP_TRY(launch_splitk_finish(splitk_ws, nso, (size_t)M * d, lp.bco, h, h, nullptr, 0, M, d, s, lp.g3, lp.be3, qkv, d));
                ln3_done = true;
            } else {
                HIP_TRY(launch_gemm(g, EPI_BIAS_RES_F32, s));
            }
        }
        if (tap_cross) HIP_TRY(hipMemcpyAsync(tap_cross, h, (size_t)M * d * 4, hipMemcpyDeviceToDevice, s));
        // ---- gated MLP (src/components/DiT.py:150-155) ----
        const void* u3 = (fr_out || ln3_done) ? (const void*)qkv : (const void*)u;   // where norm3's output is
        if (!fr_out && !ln3_done) {
            ProfScope ps(m, s, DITTO_KC_LAYERNORM);
            if (fp8) HIP_TRY(launch_layernorm_fp8(h, lp.g3, lp.be3, u, d, M, d, s));
            else HIP_TRY(launch_layernorm(h, lp.g3, lp.be3, u, d, M, d, s));
        }
        {
            ProfScope ps(m, s, DITTO_KC_GEMM_GATED);
            GemmArgs g{};
            g.A = u3; g.lda = d; g.W = lp.W1g; g.bias = lp.b1g; g.out = act; g.ldo = 4 * d; g.M = M; g.N = 8 * d;
            g.K = d; g.fp8 = fp8; g.wscale = fp8 ? lp.s1g : nullptr;
            HIP_TRY(launch_gemm(g, fp8 ? EPI_GATED_FP8 : EPI_GATED, s));
        }
        if (fr_fc2) {
            // fc2 + residual (+ the NEXT block's norm1 into u, or the bf16 copy of h_L for proj_out on the last block)
            ProfScope ps(m, s, DITTO_KC_GEMM_FC2);
            GemmParams gp{};
            gp.A = (const bf16*)act; gp.lda = 4 * d; gp.W = (const bf16*)lp.W2P; gp.ldw = 4 * d; gp.w_rows = d; gp.bias = lp.b2;
            gp.residual = h; gp.ldr = d; gp.out = h; gp.ldo = d; gp.M = M; gp.N = d; gp.K = 4 * d;
            if (xcat_or_null && hb) { gp.out = xcat_or_null + (size_t)d * 2; gp.ldo = 2 * d; }   // bf16 h_L IS proj_out's operand
            else if (xcat_or_null) { gp.out2 = (bf16*)(xcat_or_null + (size_t)d * 2); gp.ldo2 = 2 * d; }
            HIP_TRY(launch_gemm_fr(gp, next_g1, next_be1, next_g1 ? u : nullptr, d, fr_rot, s, false, hb));
        } else {
            ProfScope ps(m, s, DITTO_KC_GEMM_FC2);
            GemmArgs g{};
            g.A = act; g.lda = 4 * d; g.W = lp.W2; g.bias = lp.b2; g.residual = h; g.ldr = d; g.out = h; g.ldo = d;
            g.M = M; g.N = d; g.K = 4 * d; g.fp8 = fp8; g.wscale = fp8 ? lp.s2 : nullptr;
            void* out2 = xcat_or_null ? xcat_or_null + (size_t)d * 2 : nullptr;   // bf16(h_L) for proj_out
            const int ns = fp8 ? 1 : small_batch_k_splits(M, d, 4 * d);
            if (ns > 1 && splitk_ws && splitk_bytes >= (size_t)ns * M * d * 4) {
                g.bias = nullptr; g.residual = nullptr; g.out = splitk_ws; g.k_splits = ns;
                g.split_stride = (size_t)M * d;
                HIP_TRY(launch_gemm(g, EPI_BIAS_F32, s));
                // (next_g1 set: the finish also writes the NEXT block's norm1 into u — ditto_forward decided it, chain_ll)
                HIP_TRY(launch_splitk_finish(splitk_ws, ns, (size_t)M * d, lp.b2, h, h, out2, 2 * d, M, d, s, next_g1, next_be1,
                                             next_g1 ? u : nullptr, d));
            } else {
                if (next_g1) return fail(DITTO_ERR_ARG, "internal: norm1 chaining without a kernel that writes it");
                if (out2) { g.out2_bf16 = out2; g.ldo2 = 2 * d; }
                HIP_TRY(launch_gemm(g, EPI_BIAS_RES_F32, s));
            }
        }
    return DITTO_OK;
}

extern "C" {

int ditto_abi_version(void) { return DITTO_ABI_VERSION; }
const char* ditto_last_error(void) { return ditto::g_err; }

const char* ditto_kernel_class_name(int kc) {
    static const char* names[DITTO_KC_COUNT] = {"layernorm", "gemm_qkv_rope", "gemm_q_proj", "gemm_out_proj",
                                                "gemm_gated_mlp", "gemm_fc2", "gemm_final", "attn_self", "attn_cross",
                                                "adaln", "p_sample_update"};
    return (kc >= 0 && kc < DITTO_KC_COUNT) ? names[kc] : "?";
}

size_t ditto_arena_bytes(const ditto_config* cfg) {
    if (check_cfg(cfg) != DITTO_OK) return 0;
    return plan_arena(*cfg).total;
}
size_t ditto_cond_bytes(const ditto_config* cfg, int B, int T) {
    if (check_cfg(cfg) != DITTO_OK || B <= 0 || T <= 0) return 0;
    const size_t d = cfg->hidden_dim, dp = cfg_dp(*cfg);
    return al((size_t)B * T * cfg->num_layers * 2 * dp * 2) + al((size_t)B * 2 * d * 4);
}
size_t ditto_workspace_bytes(const ditto_config* cfg, int B, int N, int T) {
    if (check_cfg(cfg) != DITTO_OK || B <= 0 || N <= 0 || T <= 0) return 0;
    return plan_ws(*cfg, B, N, T).total;
}
size_t ditto_attention_workspace_bytes(int B, int H, int Sq, int Skv, int dh) {
    return attention_workspace_bytes(B, H, Sq, Skv, dh);
}

int ditto_model_create(const ditto_config* cfg, const ditto_weights* w, void* arena, size_t arena_bytes,
                       ditto_stream_t stream, ditto_model_t* out) {
    if (int rc = check_cfg(cfg)) return rc;
    if (!w || !arena || !out || !w->layers) return fail(DITTO_ERR_ARG, "null argument to ditto_model_create");
    const ArenaPlan plan = plan_arena(*cfg);
    if (arena_bytes < plan.total)
        return fail(DITTO_ERR_SIZE, "arena too small: %zu < %zu", arena_bytes, plan.total);
    if ((uintptr_t)arena % 256) return fail(DITTO_ERR_ARG, "arena must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int d = cfg->hidden_dim, L = cfg->num_layers, td = cfg->time_dim, dt = cfg->text_dim;
    char* A = (char*)arena;
    std::unique_ptr<ditto_model> guard(new (std::nothrow) ditto_model());
    ditto_model* m = guard.get();
    if (!m) return fail(DITTO_ERR_ARG, "out of host memory");
    m->cfg = *cfg; m->plan = plan; m->arena = A; m->layers.resize(L);
    const int BIG = 1 << 30;
    for (int l = 0; l < L; ++l) {
        const ditto_layer_weights& lw = w->layers[l];
        const auto& q = plan.layers[l];
        const float* need[] = {lw.norm1_weight, lw.norm1_bias, lw.attn_in_proj_weight, lw.attn_in_proj_bias,
                               lw.norm2_weight, lw.norm2_bias, lw.cross_in_proj_weight, lw.cross_in_proj_bias,
                               lw.cross_out_proj_weight, lw.cross_out_proj_bias, lw.norm3_weight, lw.norm3_bias,
                               lw.mlp_fc1_weight, lw.mlp_fc1_bias, lw.gate_weight, lw.gate_bias, lw.mlp_fc2_weight,
                               lw.mlp_fc2_bias};
        for (const float* p : need)
            if (!p) return fail(DITTO_ERR_ARG, "null weight pointer in layer %d", l);
        // self-attention in_proj [3d, d] (q | k | v rows), reference src/components/DiT.py:110-114
        const bool fp8 = (cfg->flags & DITTO_CFG_FP8_LINEAR) != 0;
        // head_dim 64 (fused attention): q leaves its projection already multiplied by scale * log2(e), folded into
        // the q rows of the packed weights / biases here, so the attention kernel's scores are in log2 units
        const bool pre = (d / cfg->num_heads) == 64;
        const float qs = pre ? 1.4426950408889634f / sqrtf((float)(d / cfg->num_heads)) : 1.0f;
        const int dh_ = d / cfg->num_heads, dhp = cfg_dhp(*cfg), dp = cfg_dp(*cfg);
        const bool pad = dp != d;
        if (pad) {
            // padded heads: zero images, then every in-projection's rows / the out-projection's columns head by head
            HIP_TRY(hipMemsetAsync(A + q.Wqkv, 0, (size_t)3 * dp * d * 2, s));
            HIP_TRY(hipMemsetAsync(A + q.Wcq, 0, (size_t)dp * d * 2, s));
            HIP_TRY(hipMemsetAsync(A + q.Wco, 0, (size_t)d * dp * 2, s));
            HIP_TRY(hipMemsetAsync(A + q.bqkv, 0, (size_t)3 * dp * 4, s));
            HIP_TRY(hipMemsetAsync(A + q.bcq, 0, (size_t)dp * 4, s));
            if (l == 0) {
                HIP_TRY(hipMemsetAsync(A + plan.Wkv, 0, (size_t)L * 2 * dp * d * 2, s));
                HIP_TRY(hipMemsetAsync(A + plan.bkv, 0, (size_t)L * 2 * dp * 4, s));
            }
            for (int part = 0; part < 3; ++part) {   // q | k | v rows of attn.in_proj
                HIP_TRY(launch_pack_bf16_headrows(lw.attn_in_proj_weight + (size_t)part * d * d, A + q.Wqkv, d, d, d, dh_, dhp,
                                                  part * dp, s));
                HIP_TRY(launch_pack_vec_heads(lw.attn_in_proj_bias + part * d, (float*)(A + q.bqkv), d, dh_, dhp, part * dp, s));
            }
            HIP_TRY(launch_pack_bf16_headrows(lw.cross_in_proj_weight, A + q.Wcq, d, d, d, dh_, dhp, 0, s));
            HIP_TRY(launch_pack_vec_heads(lw.cross_in_proj_bias, (float*)(A + q.bcq), d, dh_, dhp, 0, s));
            for (int part = 0; part < 2; ++part) {   // k | v rows of cross_attn.in_proj -> the all-layer K/V projection
                HIP_TRY(launch_pack_bf16_headrows(lw.cross_in_proj_weight + (size_t)(1 + part) * d * d, A + plan.Wkv, d, d, d, dh_,
                                                  dhp, l * 2 * dp + part * dp, s));
                HIP_TRY(launch_pack_vec_heads(lw.cross_in_proj_bias + (1 + part) * d, (float*)(A + plan.bkv), d, dh_, dhp,
                                              l * 2 * dp + part * dp, s));
            }
            HIP_TRY(launch_pack_bf16_headcols(lw.cross_out_proj_weight, A + q.Wco, d, d, dp, dh_, dhp, s));
            HIP_TRY(launch_pack_bf16(lw.mlp_fc1_weight, A + q.W1g, 4 * d, d, d, 0, 16, 2, 0, s));
            HIP_TRY(launch_pack_bf16(lw.gate_weight, A + q.W1g, 4 * d, d, d, 0, 16, 2, 16, s));
            HIP_TRY(launch_pack_bf16(lw.mlp_fc2_weight, A + q.W2, d, 4 * d, 4 * d, 0, BIG, 1, 0, s));
        } else {
        if (fp8) {
            HIP_TRY(launch_pack_fp8(lw.attn_in_proj_weight, A + q.Wqkv, (float*)(A + q.sqkv), 3 * d, d, d, BIG, 1, 0, s));
            HIP_TRY(launch_pack_fp8(lw.mlp_fc1_weight, A + q.W1g, (float*)(A + q.s1g), 4 * d, d, d, 16, 2, 0, s));
            HIP_TRY(launch_pack_fp8(lw.gate_weight, A + q.W1g, (float*)(A + q.s1g), 4 * d, d, d, 16, 2, 16, s));
            HIP_TRY(launch_pack_fp8(lw.mlp_fc2_weight, A + q.W2, (float*)(A + q.s2), d, 4 * d, 4 * d, BIG, 1, 0, s));
        } else {
        HIP_TRY(launch_pack_bf16(lw.attn_in_proj_weight, A + q.Wqkv, d, d, d, 0, BIG, 1, 0, s, qs));
        HIP_TRY(launch_pack_bf16(lw.attn_in_proj_weight + (size_t)d * d, A + q.Wqkv, 2 * d, d, d, 0, BIG, 1, d, s));
        HIP_TRY(launch_pack_bf16(lw.mlp_fc1_weight, A + q.W1g, 4 * d, d, d, 0, 16, 2, 0, s));
        HIP_TRY(launch_pack_bf16(lw.gate_weight, A + q.W1g, 4 * d, d, d, 0, 16, 2, 16, s));
        HIP_TRY(launch_pack_bf16(lw.mlp_fc2_weight, A + q.W2, d, 4 * d, 4 * d, 0, BIG, 1, 0, s));
        }
        HIP_TRY(hipMemcpyAsync(A + q.bqkv, lw.attn_in_proj_bias, 3 * d * 4, hipMemcpyDeviceToDevice, s));
        // cross-attention: q rows [0,d) per layer; k,v rows [d,3d) go to the all-layer Wkv
        HIP_TRY(launch_pack_bf16(lw.cross_in_proj_weight, A + q.Wcq, d, d, d, 0, BIG, 1, 0, s, qs));
        HIP_TRY(hipMemcpyAsync(A + q.bcq, lw.cross_in_proj_bias, d * 4, hipMemcpyDeviceToDevice, s));
        if (pre) {
            HIP_TRY(launch_scale_vec((float*)(A + q.bqkv), d, qs, s));
            HIP_TRY(launch_scale_vec((float*)(A + q.bcq), d, qs, s));
            if (fp8) HIP_TRY(launch_scale_vec((float*)(A + q.sqkv), d, qs, s));   // fp8: per-row weight scale carries it
        }
        HIP_TRY(launch_pack_bf16(lw.cross_in_proj_weight + (size_t)d * d, A + plan.Wkv, 2 * d, d, d, 0, BIG, 1,
                                 l * 2 * d, s));
        HIP_TRY(hipMemcpyAsync(A + plan.bkv + (size_t)l * 2 * d * 4, lw.cross_in_proj_bias + d, 2 * d * 4,
                               hipMemcpyDeviceToDevice, s));
        HIP_TRY(launch_pack_bf16(lw.cross_out_proj_weight, A + q.Wco, d, d, d, 0, BIG, 1, 0, s));
        }   // !pad
        const bool fr_o = !pad && ((d == 768 && !fp8) || d == 1024), fr_2 = !pad && (d == 768 || d == 1024) && !fp8;   // as plan_arena
        const bool lnq_w = !pad && ((d == 768 && !fp8) || d == 1024);
        if (fr_o) HIP_TRY(launch_pack_bf16_stage_major(lw.cross_out_proj_weight, A + q.WcoP, d, d, s));
        if (fr_2) HIP_TRY(launch_pack_bf16_stage_major(lw.mlp_fc2_weight, A + q.W2P, d, 4 * d, s));
        if (lnq_w) {   // from the PACKED q-projection (it carries the folded scale * log2(e))
            HIP_TRY(launch_repack_bf16_stage_major(A + q.Wcq, A + q.WcqP, d, d, s, 16));
            if (d == 768) HIP_TRY(launch_repack_bf16_stage_major(A + q.Wcq, A + q.WcqP32, d, d, s, 32));
        }
        HIP_TRY(hipMemcpyAsync(A + q.bco, lw.cross_out_proj_bias, d * 4, hipMemcpyDeviceToDevice, s));
        // gated MLP: rows interleaved [16 x fc1 | 16 x gate] so both halves of a product meet in one lane
        HIP_TRY(launch_pack_vec(lw.mlp_fc1_bias, (float*)(A + q.b1g), 4 * d, 16, 2, 0, s));
        HIP_TRY(launch_pack_vec(lw.gate_bias, (float*)(A + q.b1g), 4 * d, 16, 2, 16, s));
        HIP_TRY(hipMemcpyAsync(A + q.b2, lw.mlp_fc2_bias, d * 4, hipMemcpyDeviceToDevice, s));
        const float* lnsrc[6] = {lw.norm1_weight, lw.norm1_bias, lw.norm2_weight, lw.norm2_bias, lw.norm3_weight,
                                 lw.norm3_bias};
        const size_t lndst[6] = {q.g1, q.be1, q.g2, q.be2, q.g3, q.be3};
        for (int i = 0; i < 6; ++i)
            HIP_TRY(hipMemcpyAsync(A + lndst[i], lnsrc[i], d * 4, hipMemcpyDeviceToDevice, s));
        LayerPack& lp = m->layers[l];
        lp.Wqkv = A + q.Wqkv; lp.Wcq = A + q.Wcq; lp.Wco = A + q.Wco; lp.W1g = A + q.W1g; lp.W2 = A + q.W2;
        lp.bqkv = (const float*)(A + q.bqkv); lp.bcq = (const float*)(A + q.bcq); lp.bco = (const float*)(A + q.bco);
        lp.b1g = (const float*)(A + q.b1g); lp.b2 = (const float*)(A + q.b2);
        lp.sqkv = (const float*)(A + q.sqkv); lp.s1g = (const float*)(A + q.s1g); lp.s2 = (const float*)(A + q.s2);
        if (fr_o) lp.WcoP = A + q.WcoP;
        if (fr_2) lp.W2P = A + q.W2P;
        if (lnq_w) { lp.WcqP = A + q.WcqP; lp.WcqP32 = d == 768 ? A + q.WcqP32 : nullptr; }
        lp.g1 = (const float*)(A + q.g1); lp.be1 = (const float*)(A + q.be1); lp.g2 = (const float*)(A + q.g2);
        lp.be2 = (const float*)(A + q.be2); lp.g3 = (const float*)(A + q.g3); lp.be3 = (const float*)(A + q.be3);
    }
    const float* gneed[] = {w->t_embedding_weight, w->time_embed_0_weight, w->time_embed_0_bias,
                            w->time_embed_2_weight, w->time_embed_2_bias, w->ada_time_mlp_weight,
                            w->ada_time_mlp_bias, w->ada_text_mlp_weight, w->ada_text_mlp_bias, w->proj_in_weight,
                            w->proj_in_bias, w->proj_out_weight, w->proj_out_bias, w->rotary_inv_freq};
    int nnull = 0;
    for (const float* p : gneed) nnull += (p == nullptr);
    if (nnull == (int)(sizeof(gneed) / sizeof(gneed[0]))) {
        // standalone DiT blocks (reference src/components/DiT.py:75-157 used without a DiTTO around them)
        m->blocks_only = true;
        m->Wkv = A + plan.Wkv; m->bkv = (const float*)(A + plan.bkv);
        m->Wfin = nullptr; m->bfin = nullptr; m->ttab = nullptr; m->wx = nullptr; m->bx = nullptr; m->invf = nullptr;
        *out = guard.release();
        return DITTO_OK;
    }
    if (nnull) return fail(DITTO_ERR_ARG, "null global weight pointer (pass ALL of them, or none for a blocks-only handle)");
    // eps = x W_in^T + h_L W_out^T + (b_in + b_out): one GEMM with K = 2d (src/model/DiTTO.py:83,93-94)
    HIP_TRY(launch_pack_bf16(w->proj_in_weight, A + plan.Wfin, d, d, 2 * d, 0, BIG, 1, 0, s));
    HIP_TRY(launch_pack_bf16(w->proj_out_weight, A + plan.Wfin, d, d, 2 * d, d, BIG, 1, 0, s));
    HIP_TRY(launch_add_vec(w->proj_in_bias, w->proj_out_bias, (float*)(A + plan.bfin), d, s));
    HIP_TRY(launch_time_table(w->t_embedding_weight, w->time_embed_0_weight, w->time_embed_0_bias,
                              w->time_embed_2_weight, w->time_embed_2_bias, w->ada_time_mlp_weight,
                              w->ada_time_mlp_bias, (float*)(A + plan.ttab), cfg->diffusion_steps, td, d, s));
    HIP_TRY(hipMemcpyAsync(A + plan.wx, w->ada_text_mlp_weight, (size_t)2 * d * dt * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(A + plan.bx, w->ada_text_mlp_bias, (size_t)2 * d * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(A + plan.invf, w->rotary_inv_freq, (size_t)(d / cfg->num_heads / 2) * 4,
                           hipMemcpyDeviceToDevice, s));
    m->Wkv = A + plan.Wkv; m->bkv = (const float*)(A + plan.bkv); m->Wfin = A + plan.Wfin;
    m->bfin = (const float*)(A + plan.bfin); m->ttab = (const float*)(A + plan.ttab);
    m->wx = (const float*)(A + plan.wx); m->bx = (const float*)(A + plan.bx); m->invf = (const float*)(A + plan.invf);
    HIP_TRY(hipMemcpyAsync(A + plan.invf_rev, w->rotary_inv_freq, (size_t)(d / cfg->num_heads / 2) * 4,
                           hipMemcpyDeviceToDevice, s));
    HIP_TRY(launch_scale_vec((float*)(A + plan.invf_rev), d / cfg->num_heads / 2, 0.15915494309189535f, s));
    m->invf_rev = (const float*)(A + plan.invf_rev);
    *out = guard.release();
    return DITTO_OK;
}

int ditto_model_destroy(ditto_model_t m) {
    if (!m) return DITTO_OK;
    for (auto& r : m->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : m->pool) (void)hipEventDestroy(e);
    delete m;
    return DITTO_OK;
}

int ditto_rope_tables(ditto_model_t m, int N, float* cos_out, float* sin_out, ditto_stream_t stream) {
    if (!m || !cos_out || !sin_out || N <= 0) return fail(DITTO_ERR_ARG, "bad argument to ditto_rope_tables");
    if (m->blocks_only) return fail(DITTO_ERR_ARG, "blocks-only handle has no rotary.inv_freq");
    HIP_TRY(launch_rope_tables(m->invf, cos_out, sin_out, N, m->cfg.hidden_dim / m->cfg.num_heads / 2,
                               (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_text_precompute(ditto_model_t m, const float* text, int B, int T, void* cond, size_t cond_bytes,
                          void* workspace, size_t workspace_bytes, ditto_stream_t stream) {
    if (!m || !text || !cond || !workspace || B <= 0 || T <= 0)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_text_precompute");
    const ditto_config& c = m->cfg;
    if (cond_bytes < ditto_cond_bytes(&c, B, T)) return fail(DITTO_ERR_SIZE, "cond buffer too small");
    const size_t need = text_scratch_bytes(c, B, T);
    if (workspace_bytes < need) return fail(DITTO_ERR_SIZE, "workspace too small for text precompute: %zu < %zu",
                                            workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    const int d = c.hidden_dim, L = c.num_layers, dp = cfg_dp(c);
    char* ws = (char*)workspace;
    void* textbf = ws;
    float* pooled = (float*)(ws + al((size_t)B * T * c.text_dim * 2));
    char* kv = (char*)cond;
    float* tmod = (float*)(kv + al((size_t)B * T * L * 2 * dp * 2));
    HIP_TRY(launch_cast_bf16(text, textbf, (size_t)B * T * c.text_dim, s));
    GemmArgs g{};
    g.A = textbf; g.lda = c.text_dim; g.W = m->Wkv; g.bias = m->bkv; g.out = kv; g.ldo = L * 2 * dp;
    g.M = B * T; g.N = L * 2 * dp; g.K = d;
    HIP_TRY(launch_gemm(g, EPI_BIAS_BF16, s));
    if (!m->blocks_only) HIP_TRY(launch_text_mod(text, m->wx, m->bx, pooled, tmod, B, T, c.text_dim, d, s));
    return DITTO_OK;
}

// thread-scoped options: the stack of what ditto_call_opts_push found in force (kernels.h t_opts is the top)
static thread_local CallOpts t_opt_stack[16];
static thread_local int t_opt_depth = 0;

int ditto_call_opts_push(const ditto_call_opts* opts) {
    if (int rc = check_call_opts(opts)) return rc;
    if (t_opt_depth >= 16) return fail(DITTO_ERR_ARG, "ditto_call_opts_push: more than 16 nested scopes on this thread");
    t_opt_stack[t_opt_depth++] = t_opts;
    CallScope::apply(opts);
    return DITTO_OK;
}

int ditto_call_opts_pop(void) {
    if (t_opt_depth <= 0) return fail(DITTO_ERR_ARG, "ditto_call_opts_pop without a matching push on this thread");
    t_opts = t_opt_stack[--t_opt_depth];
    return DITTO_OK;
}

int ditto_call_opts_current(ditto_call_opts* out) {
    if (!out) return fail(DITTO_ERR_ARG, "null argument to ditto_call_opts_current");
    *out = ditto_call_opts{opt_class_rows(), opt_resid_bf16(), opt_fr_mask(), opt_lnq(), {0, 0, 0, 0}};
    return DITTO_OK;
}

int ditto_forward_opts(ditto_model_t m, const float* x, const void* cond, const int64_t* t, int B, int N, int T,
                       const float* rope_cos, const float* rope_sin, float* eps_out, void* workspace,
                       size_t workspace_bytes, ditto_stream_t stream, const ditto_call_opts* opts) {
    if (int rc = check_call_opts(opts)) return rc;
    CallScope scope(opts);
    return ditto_forward(m, x, cond, t, B, N, T, rope_cos, rope_sin, eps_out, workspace, workspace_bytes, stream);
}

int ditto_forward(ditto_model_t m, const float* x, const void* cond, const int64_t* t, int B, int N, int T,
                  const float* rope_cos, const float* rope_sin, float* eps_out, void* workspace,
                  size_t workspace_bytes, ditto_stream_t stream) {
    if (!m || !x || !cond || !t || !rope_cos || !rope_sin || !eps_out || !workspace || B <= 0 || N <= 0 || T <= 0)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_forward");
    const ditto_config& c = m->cfg;
    if (m->blocks_only) return fail(DITTO_ERR_ARG, "ditto_forward on a blocks-only handle");
    const WsPlan w = plan_ws(c, B, N, T);
    if (workspace_bytes < w.total)
        return fail(DITTO_ERR_SIZE, "workspace too small: %zu < %zu", workspace_bytes, w.total);
    if ((uintptr_t)workspace % 256) return fail(DITTO_ERR_ARG, "workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int d = c.hidden_dim, L = c.num_layers, M = B * N;
    char* ws = (char*)workspace;
    float* h = (float*)(ws + w.h);
    void* u = ws + w.u;
    char* qkv = ws + w.qkv;
    void* act = ws + w.act;
    char* xcat = ws + w.xcat;
    void* attn_ws = ws + w.attn;
    const char* kv = (const char*)cond;
    const float* tmod = (const float*)(kv + al((size_t)B * T * L * 2 * cfg_dp(c) * 2));

    // fc2 on the full-row kernel also emits the NEXT block's norm1 (fr_mask bit 1): that block then skips its LayerNorm
    const bool fp8c = (c.flags & DITTO_CFG_FP8_LINEAR) != 0;
    const bool chain_ln1 = (opt_fr_mask() & 2) && !fp8c && m->layers[0].W2P && fr_fc2_ok(M, d);
    // bf16 residual stream ("residual_bf16"): only where EVERY consumer of h has the bf16 form — d = 768, head_dim 64, both fused
    // launches of a block on gemm_frd.hip (the full-row class); any other launch keeps the fp32 stream
    const bool hb_class = opt_resid_bf16() && !fp8c && d == 768 && d / c.num_heads == 64 && chain_ln1 && (opt_fr_mask() & 1) &&
                          m->layers[0].WcoP && fr_outproj_ok(M, d) &&
                          (g_fr_tile == 130 || (g_fr_tile == 0 && fr_rule_rows(opt_class_rows() > 0 ? opt_class_rows() : M) == 130));
    // (only gemm_frd.hip has the bf16 form: a launch of fewer than 128 rows under a PINNED class would run the 64-row kernel)
    if (hb_class && M < 128)
        return fail(DITTO_ERR_SHAPE, "kernel class pinned to a batch of %d rows (bf16 residual stream on the 128-row full-row kernel), "
                                     "but this launch has %d rows: it cannot take that class.  Give every shard at least 128 rows, or "
                                     "ditto_set_option(\"residual_bf16\", 0).", opt_class_rows(), M);
    const bool hb = hb_class;
    // low-latency class: fc2 runs split over K and its finish launch also writes the next block's norm1 (the same bits as the
    // LayerNorm launch it replaces); decided exactly as run_block will decide the split
    const int ns_fc2 = fp8c ? 1 : small_batch_k_splits(M, d, 4 * d);
    const bool chain_ll = !chain_ln1 && (g_ll_mask & 1) && ns_fc2 > 1 && w.splitk_bytes >= (size_t)ns_fc2 * M * d * 4;
    // block 0's norm1 rides in the GlobalAdaLN kernel (same statistics order as the LayerNorm kernel: the same bits)
    const bool ln1_in_adaln = !fp8c && !(g_gemm_flags & 32768);      // gemm_flags bit 15: A/B, the separate launch
    {   // GlobalAdaLN (src/components/DiT.py:25-40) + bf16 copy of the raw input for proj_in
        ProfScope ps(m, s, DITTO_KC_ADALN);
        HIP_TRY(launch_adaln(x, m->ttab, tmod, t, c.diffusion_steps, h, xcat, 2 * d, B, N, d, s, hb,
                             ln1_in_adaln ? m->layers[0].g1 : nullptr, ln1_in_adaln ? m->layers[0].be1 : nullptr,
                             ln1_in_adaln ? u : nullptr));
    }
    for (int l = 0; l < L; ++l)
        if (int rc = run_block(m, l, h, u, qkv, act, l == L - 1 ? xcat : nullptr, attn_ws, w.attn_bytes,
                               (float*)(ws + w.splitk), w.splitk_bytes, kv, l,
                               L * 2 * cfg_dp(c), rope_cos, rope_sin, B, N, T, s, nullptr, nullptr,
                               ((chain_ln1 || chain_ll) && l > 0) || (ln1_in_adaln && l == 0),
                               (chain_ln1 || chain_ll) && l + 1 < L ? m->layers[l + 1].g1 : nullptr,
                               (chain_ln1 || chain_ll) && l + 1 < L ? m->layers[l + 1].be1 : nullptr, hb))
            return rc;
    {   // eps = proj_in(x_raw) + proj_out(h_L)  (src/model/DiTTO.py:83,93-94), one K = 2d GEMM
        ProfScope ps(m, s, DITTO_KC_GEMM_FINAL);
        GemmArgs g{};
        g.A = xcat; g.lda = 2 * d; g.W = m->Wfin; g.bias = m->bfin; g.out = eps_out; g.ldo = d; g.M = M; g.N = d;
        g.K = 2 * d;
        const int ns = small_batch_k_splits(M, d, 2 * d);
        if (ns > 1 && w.splitk_bytes >= (size_t)ns * M * d * 4) {
            float* part = (float*)(ws + w.splitk);
            g.bias = nullptr; g.out = part; g.k_splits = ns; g.split_stride = (size_t)M * d;
            HIP_TRY(launch_gemm(g, EPI_BIAS_F32, s));
            HIP_TRY(launch_splitk_finish(part, ns, (size_t)M * d, m->bfin, nullptr, eps_out, nullptr, 0, M, d, s));
        } else {
            HIP_TRY(launch_gemm(g, EPI_BIAS_F32, s));
        }
    }
    return DITTO_OK;
}


int ditto_block_forward(ditto_model_t m, int layer, float* h, const void* cond, int cond_layer, int B, int N, int T,
                        const float* rope_cos, const float* rope_sin, void* workspace, size_t workspace_bytes,
                        ditto_stream_t stream) {
    return ditto_block_forward_taps(m, layer, h, cond, cond_layer, B, N, T, rope_cos, rope_sin, nullptr, nullptr,
                                    workspace, workspace_bytes, stream);
}

int ditto_block_forward_taps(ditto_model_t m, int layer, float* h, const void* cond, int cond_layer, int B, int N,
                             int T, const float* rope_cos, const float* rope_sin, float* tap_self, float* tap_cross,
                             void* workspace, size_t workspace_bytes, ditto_stream_t stream) {
    if (!m || !h || !cond || !rope_cos || !rope_sin || !workspace || B <= 0 || N <= 0 || T <= 0)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_block_forward");
    const ditto_config& c = m->cfg;
    if (layer < 0 || layer >= c.num_layers || cond_layer < 0 || cond_layer >= c.num_layers)
        return fail(DITTO_ERR_ARG, "layer index out of range");
    const WsPlan w = plan_ws(c, B, N, T);
    if (workspace_bytes < w.total) return fail(DITTO_ERR_SIZE, "workspace too small: %zu < %zu", workspace_bytes, w.total);
    char* ws = (char*)workspace;
    return run_block(m, layer, h, ws + w.u, ws + w.qkv, ws + w.act, nullptr, ws + w.attn, w.attn_bytes,
                     (float*)(ws + w.splitk), w.splitk_bytes, (const char*)cond, cond_layer, c.num_layers * 2 * cfg_dp(c), rope_cos, rope_sin, B, N, T,
                     (hipStream_t)stream, tap_self, tap_cross);
}


size_t ditto_global_adaln_scratch_bytes(int B, int d, int time_dim, int text_dim) {
    return al((size_t)B * time_dim * 4) + al((size_t)B * text_dim * 4) + 2 * al((size_t)B * 2 * d * 4);
}

int ditto_global_adaln(const float* x, const float* time_emb, const float* text_emb, const float* time_w,
                       const float* time_b, const float* text_w, const float* text_b, int B, int N, int T, int d,
                       int time_dim, int text_dim, float* out, void* scratch, size_t scratch_bytes,
                       ditto_stream_t stream) {
    if (!x || !time_emb || !text_emb || !time_w || !time_b || !text_w || !text_b || !out || !scratch || B <= 0 ||
        N <= 0 || T <= 0)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_global_adaln");
    if (d % 4 || d > 2048) return fail(DITTO_ERR_SHAPE, "d must be a multiple of 4 and <= 2048");
    if (scratch_bytes < ditto_global_adaln_scratch_bytes(B, d, time_dim, text_dim))
        return fail(DITTO_ERR_SIZE, "scratch too small for ditto_global_adaln");
    hipStream_t s = (hipStream_t)stream;
    char* sc = (char*)scratch;
    float* pooled_t = (float*)sc;                 sc += al((size_t)B * time_dim * 4);
    float* pooled_x = (float*)sc;                 sc += al((size_t)B * text_dim * 4);
    float* mod_t = (float*)sc;                    sc += al((size_t)B * 2 * d * 4);
    float* mod_x = (float*)sc;
    // time half: Linear(SiLU(time_emb)) == the text-mod kernel with a "sequence" of length 1
    HIP_TRY(launch_text_mod(time_emb, time_w, time_b, pooled_t, mod_t, B, 1, time_dim, d, s));
    HIP_TRY(launch_text_mod(text_emb, text_w, text_b, pooled_x, mod_x, B, T, text_dim, d, s));
    HIP_TRY(launch_adaln(x, mod_t, mod_x, nullptr, B, out, nullptr, 0, B, N, d, s));
    return DITTO_OK;
}

int ditto_apply_rope_f32(const float* pos, const float* t, float* out, int B, int N, int H, int dh,
                         ditto_stream_t stream) {
    if (!pos || !t || !out || B <= 0 || N <= 0 || H <= 0 || dh <= 0 || (dh & 1))
        return fail(DITTO_ERR_ARG, "bad argument to ditto_apply_rope_f32");
    if (t == out) return fail(DITTO_ERR_ARG, "ditto_apply_rope_f32 is not in-place");
    HIP_TRY(launch_apply_rope_f32(pos, t, out, B, N, H, dh, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_p_sample_update(float* x, const float* eps, const float* noise, const int64_t* t, const float* betas,
                          const float* alphas, const float* alphas_cumprod, int B, size_t elems_per_utt,
                          ditto_stream_t stream) {
    if (!x || !eps || !t || !betas || !alphas || !alphas_cumprod || B <= 0 || elems_per_utt == 0)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_p_sample_update");
    if (elems_per_utt % 4) return fail(DITTO_ERR_SHAPE, "elems_per_utt must be a multiple of 4");
    HIP_TRY(launch_p_sample_update(x, eps, noise, t, betas, alphas, alphas_cumprod, B, elems_per_utt,
                                   (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_p_sample(ditto_model_t m, float* x, const void* cond, const int64_t* t, const float* noise,
                   const float* betas, const float* alphas, const float* alphas_cumprod, int B, int N, int T,
                   const float* rope_cos, const float* rope_sin, void* workspace, size_t workspace_bytes,
                   ditto_stream_t stream) {
    if (!m || !workspace) return fail(DITTO_ERR_ARG, "bad argument to ditto_p_sample");
    const WsPlan w = plan_ws(m->cfg, B, N, T);
    if (workspace_bytes < w.total) return fail(DITTO_ERR_SIZE, "workspace too small: %zu < %zu", workspace_bytes, w.total);
    float* eps = (float*)((char*)workspace + w.eps);
    if (int rc = ditto_forward(m, x, cond, t, B, N, T, rope_cos, rope_sin, eps, workspace, workspace_bytes, stream))
        return rc;
    ProfScope ps(m, (hipStream_t)stream, DITTO_KC_UPDATE);
    return ditto_p_sample_update(x, eps, noise, t, betas, alphas, alphas_cumprod, B,
                                 (size_t)N * m->cfg.hidden_dim, stream);
}

int ditto_p_sample_opts(ditto_model_t m, float* x, const void* cond, const int64_t* t, const float* noise,
                        const float* betas, const float* alphas, const float* alphas_cumprod, int B, int N, int T,
                        const float* rope_cos, const float* rope_sin, void* workspace, size_t workspace_bytes,
                        ditto_stream_t stream, const ditto_call_opts* opts) {
    if (int rc = check_call_opts(opts)) return rc;
    CallScope scope(opts);
    return ditto_p_sample(m, x, cond, t, noise, betas, alphas, alphas_cumprod, B, N, T, rope_cos, rope_sin, workspace,
                          workspace_bytes, stream);
}

int ditto_noise_normal(float* out, const int64_t* seeds, uint32_t step, int B, size_t elems_per_utt,
                       ditto_stream_t stream) {
    if (!out || !seeds || B <= 0 || elems_per_utt == 0) return fail(DITTO_ERR_ARG, "bad argument to ditto_noise_normal");
    if (elems_per_utt % 4) return fail(DITTO_ERR_SHAPE, "elems_per_utt must be a multiple of 4");
    HIP_TRY(launch_noise_normal(out, seeds, step, B, elems_per_utt, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_p_sample_seeded(ditto_model_t m, float* x, const void* cond, const int64_t* t, const int64_t* seeds,
                          uint32_t step, const float* betas, const float* alphas, const float* alphas_cumprod, int B,
                          int N, int T, const float* rope_cos, const float* rope_sin, void* workspace,
                          size_t workspace_bytes, ditto_stream_t stream) {
    if (!m || !workspace || !seeds || !betas || !alphas || !alphas_cumprod)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_p_sample_seeded");
    const WsPlan w = plan_ws(m->cfg, B, N, T);
    if (workspace_bytes < w.total) return fail(DITTO_ERR_SIZE, "workspace too small: %zu < %zu", workspace_bytes, w.total);
    float* eps = (float*)((char*)workspace + w.eps);
    if (int rc = ditto_forward(m, x, cond, t, B, N, T, rope_cos, rope_sin, eps, workspace, workspace_bytes, stream))
        return rc;
    ProfScope ps(m, (hipStream_t)stream, DITTO_KC_UPDATE);
    const size_t per = (size_t)N * m->cfg.hidden_dim;
    if (per % 4) return fail(DITTO_ERR_SHAPE, "elems_per_utt must be a multiple of 4");
    HIP_TRY(launch_p_sample_update_seeded(x, eps, seeds, step, t, betas, alphas, alphas_cumprod, B, per,
                                          (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_p_sample_seeded_opts(ditto_model_t m, float* x, const void* cond, const int64_t* t, const int64_t* seeds,
                               uint32_t step, const float* betas, const float* alphas, const float* alphas_cumprod, int B,
                               int N, int T, const float* rope_cos, const float* rope_sin, void* workspace,
                               size_t workspace_bytes, ditto_stream_t stream, const ditto_call_opts* opts) {
    if (int rc = check_call_opts(opts)) return rc;
    CallScope scope(opts);
    return ditto_p_sample_seeded(m, x, cond, t, seeds, step, betas, alphas, alphas_cumprod, B, N, T, rope_cos, rope_sin, workspace,
                                 workspace_bytes, stream);
}

int ditto_denoise_steps_opts(ditto_model_t m, float* x, const void* cond, int t_begin, int t_end, const float* noise,
                             const float* betas, const float* alphas, const float* alphas_cumprod, int B, int N, int T,
                             const float* rope_cos, const float* rope_sin, int64_t* t_scratch, void* workspace,
                             size_t workspace_bytes, ditto_stream_t stream, const ditto_call_opts* opts) {
    if (int rc = check_call_opts(opts)) return rc;
    CallScope scope(opts);
    return ditto_denoise_steps(m, x, cond, t_begin, t_end, noise, betas, alphas, alphas_cumprod, B, N, T, rope_cos, rope_sin,
                               t_scratch, workspace, workspace_bytes, stream);
}

int ditto_denoise_steps(ditto_model_t m, float* x, const void* cond, int t_begin, int t_end, const float* noise,
                        const float* betas, const float* alphas, const float* alphas_cumprod, int B, int N, int T,
                        const float* rope_cos, const float* rope_sin, int64_t* t_scratch, void* workspace,
                        size_t workspace_bytes, ditto_stream_t stream) {
    if (!m || !x || !t_scratch) return fail(DITTO_ERR_ARG, "bad argument to ditto_denoise_steps");
    if (t_end < 0 || t_begin < t_end || t_begin >= m->cfg.diffusion_steps)
        return fail(DITTO_ERR_ARG, "need 0 <= t_end <= t_begin < diffusion_steps");
    if (!noise && t_begin > 0) return fail(DITTO_ERR_ARG, "noise may be NULL only for the single step t = 0");
    const size_t per_step = (size_t)B * N * m->cfg.hidden_dim;
    for (int tv = t_begin, i = 0; tv >= t_end; --tv, ++i) {
        HIP_TRY(launch_fill_i64(t_scratch, B, tv, (hipStream_t)stream));
        if (int rc = ditto_p_sample(m, x, cond, t_scratch, noise ? noise + (size_t)i * per_step : nullptr, betas, alphas,
                                    alphas_cumprod, B, N, T, rope_cos, rope_sin, workspace, workspace_bytes, stream))
            return rc;
    }
    return DITTO_OK;
}

int ditto_q_sample(const float* x_start, const float* noise, const int64_t* t, const float* buffer, float* out, int B,
                   size_t elems_per_utt, ditto_stream_t stream) {
    if (!x_start || !noise || !t || !buffer || !out || B <= 0) return fail(DITTO_ERR_ARG, "bad argument to ditto_q_sample");
    if (elems_per_utt % 4) return fail(DITTO_ERR_SHAPE, "elems_per_utt must be a multiple of 4");
    HIP_TRY(launch_q_sample(x_start, noise, t, buffer, out, B, elems_per_utt, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_layernorm_bf16(const float* x, const float* gamma, const float* beta, void* out_bf16, int M, int d,
                         ditto_stream_t stream) {
    if (!x || !out_bf16 || M <= 0 || d <= 0 || (gamma == nullptr) != (beta == nullptr))
        return fail(DITTO_ERR_ARG, "bad argument to ditto_layernorm_bf16");
    if (d % 4 || d > 2048) return fail(DITTO_ERR_SHAPE, "d must be a multiple of 4 and <= 2048");
    HIP_TRY(launch_layernorm(x, gamma, beta, out_bf16, d, M, d, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_gemm_bf16(const void* A, int lda, const void* W, const float* bias, const float* residual, void* out,
                    int ldo, int M, int N, int K, int epilogue, ditto_stream_t stream) {
    if (!A || !W || !out) return fail(DITTO_ERR_ARG, "null pointer to ditto_gemm_bf16");
    if (K % 64 || N % 16 || lda % 8) return fail(DITTO_ERR_SHAPE, "need K %% 64 == 0, N %% 16 == 0, lda %% 8 == 0");
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.bias = bias; g.residual = residual; g.ldr = ldo; g.out = out; g.ldo = ldo;
    g.M = M; g.N = N; g.K = K;
    GemmEpilogue e;
    switch (epilogue) {
        case 0: e = EPI_BIAS_BF16; break;
        case 1: e = EPI_BIAS_RES_F32; break;
        case 3:
            if (!bias || N % 32) return fail(DITTO_ERR_ARG, "gated epilogue needs a bias and N %% 32 == 0");
            e = EPI_GATED; break;
        case 4: e = EPI_BIAS_F32; break;
        case 6: e = EPI_BIAS_RELU_BF16; break;
        default: return fail(DITTO_ERR_ARG, "epilogue must be 0, 1, 3, 4 or 6");
    }
    HIP_TRY(launch_gemm(g, e, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_gemm_tn_bf16(const void* X, int ldx, const void* Y, int ldy, float* out, int ldo, int Mo, int No, int K,
                       int k_splits, int tile, void* workspace, size_t workspace_bytes, ditto_stream_t stream) {
    if (!X || !Y || !out || !workspace || Mo < 8 || No < 8 || K <= 0 || (Mo | No | ldx | ldy) % 8 || ldo < No)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_gemm_tn_bf16");
    if (tile != 128 && tile != 256) return fail(DITTO_ERR_ARG, "tile must be 128 or 256");
    if ((uintptr_t)workspace % 256) return fail(DITTO_ERR_ARG, "workspace must be 256-byte aligned");
    const int S = k_splits > 1 ? k_splits : 1;
    const size_t need = 256 + (S > 1 ? (size_t)S * Mo * No * 4 : 0);
    if (workspace_bytes < need) return fail(DITTO_ERR_SIZE, "workspace too small: %zu < %zu", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(workspace, 0, 256, s));
    if (S > 1) {
        if (ldo != No) return fail(DITTO_ERR_ARG, "split-K needs a contiguous output (ldo == No)");
        float* part = (float*)((char*)workspace + 256);
        HIP_TRY(launch_gemm_tn(X, ldx, Y, ldy, workspace, part, No, Mo, No, K, S, (size_t)Mo * No, s, tile == 256));
        HIP_TRY(launch_reduce_partials(part, S, (size_t)Mo * No, out, s));
    } else {
        HIP_TRY(launch_gemm_tn(X, ldx, Y, ldy, workspace, out, ldo, Mo, No, K, 1, 0, s, tile == 256));
    }
    return DITTO_OK;
}

int ditto_gemm_lnq_bf16(const void* h, int ldh, int h_is_bf16, const float* gamma, const float* beta, const void* W,
                        const float* bias, void* out_bf16, int ldo, int M, int d, int mfma_shape, void* w_scratch,
                        ditto_stream_t stream) {
    if (d != 768 && d != 1024) return fail(DITTO_ERR_SHAPE, "ditto_gemm_lnq_bf16: d must be 768 or 1024");
    if (!h || !gamma || !beta || !W || !out_bf16 || !w_scratch || M <= 0 || ldh < d || ldo < d || ldh % 4 || ldo % 8)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_gemm_lnq_bf16");
    if (mfma_shape != 32 && mfma_shape != 16) return fail(DITTO_ERR_ARG, "mfma_shape must be 32 (32x32x16) or 16 (16x16x32)");
    if (d == 1024 && (mfma_shape != 32 || h_is_bf16)) return fail(DITTO_ERR_SHAPE, "d = 1024: 32x32x16 and fp32 rows only");
    if ((uintptr_t)w_scratch % 256) return fail(DITTO_ERR_ARG, "w_scratch must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(launch_repack_bf16_stage_major(W, w_scratch, d, d, s, mfma_shape == 32 ? 16 : 32));
    HIP_TRY(launch_gemm_lnq(h, ldh, h_is_bf16 != 0, gamma, beta, w_scratch, bias, out_bf16, ldo, M, d, mfma_shape,
                            g_fr_rot > 1 ? g_fr_rot : 0, s));   // "fr_rot" > 1: the unit entry rotates too, with that period in tiles
    return DITTO_OK;
}

int ditto_gemm_ln_bf16(const void* A, int lda, const void* W, const float* bias, const float* residual, float* out,
                       int ldo, const float* gamma, const float* beta, void* u_bf16, int ldu, int M, int N, int K,
                       ditto_stream_t stream) {
    if (!A || !W || !out || (gamma == nullptr) != (beta == nullptr) || (gamma == nullptr) != (u_bf16 == nullptr))
        return fail(DITTO_ERR_ARG, "bad argument to ditto_gemm_ln_bf16");
    const bool ok = N == 1024 ? gemm_fr64_supports(M, N, K, (size_t)lda, (size_t)K) : gemm_fr_supports(M, N, K, (size_t)lda, (size_t)K);
    if (!ok || lda % 8)
        return fail(DITTO_ERR_SHAPE, "ditto_gemm_ln_bf16 needs N == 768 (M >= 128) or N == 1024 (M >= 64), K %% 64 == 0, lda %% 8 == 0");
    GemmParams p{};
    p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = K; p.w_rows = N; p.bias = bias;
    p.residual = residual; p.ldr = ldo; p.out = out; p.ldo = ldo; p.M = M; p.N = N; p.K = K;
    if (g_fr_u_fp8 && (N != 1024 || !gamma)) return fail(DITTO_ERR_SHAPE, "fr_u_fp8 (test hook) needs N == 1024 and a LayerNorm output");
    HIP_TRY(launch_gemm_fr(p, gamma, beta, u_bf16, ldu, g_fr_rot > 1 ? g_fr_rot : 0, (hipStream_t)stream, g_fr_u_fp8 != 0));
    return DITTO_OK;
}

int ditto_attention_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo,
                         int B, int H, int Sq, int Skv, int dh, float scale, void* workspace, size_t workspace_bytes,
                         ditto_stream_t stream) {
    if (!q || !k || !v || !out) return fail(DITTO_ERR_ARG, "null pointer to ditto_attention_bf16");
    if (dh % 64) return fail(DITTO_ERR_SHAPE, "head_dim must be a multiple of 64");
    // (head_dim 64: the scratch is OPTIONAL — the split-KV partials of the low-latency class; without it the launch is not split)
    if (dh != 64 && workspace_bytes < attention_workspace_bytes(B, H, Sq, Skv, dh))
        return fail(DITTO_ERR_SIZE, "attention workspace too small");
    AttnArgs a{};
    a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.out_bf16 = out; a.ldo = ldo;
    a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.dh = dh; a.scale = scale;
    a.workspace = workspace; a.workspace_bytes = workspace_bytes;
    a.q_prescaled = (g_attn_flags & 16) && dh == 64;   // unit tests of the pre-scaled-q kernel: q carries scale*log2(e)
    HIP_TRY(launch_attention(a, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_vq_argmin(const float* latents, const float* codebook, int64_t* idx, int R, int K, int D, float* scratch_k,
                    ditto_stream_t stream) {
    if (!latents || !codebook || !idx || !scratch_k || R <= 0 || K <= 0 || D <= 0)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_vq_argmin");
    HIP_TRY(launch_vq_argmin(latents, codebook, scratch_k, idx, R, K, D, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_embedding_gather(const float* table, const int64_t* ids, float* out, int n, int V, int d,
                           ditto_stream_t stream) {
    if (!table || !ids || !out || n <= 0 || V <= 0 || d <= 0) return fail(DITTO_ERR_ARG, "bad argument to ditto_embedding_gather");
    if (d % 4) return fail(DITTO_ERR_SHAPE, "d must be a multiple of 4");
    HIP_TRY(launch_embedding_gather(table, ids, out, n, V, d, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_code_embed_mean(const float* table, const int64_t* codes, float* out, int B, int C, int F, int Fout, int V,
                          int d, ditto_stream_t stream) {
    if (!table || !codes || !out || B <= 0 || C <= 0 || F <= 0 || Fout <= 0 || Fout > F || V <= 0 || d <= 0)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_code_embed_mean");
    if (d % 4) return fail(DITTO_ERR_SHAPE, "d must be a multiple of 4");
    HIP_TRY(launch_code_embed_mean(table, codes, out, B, C, F, Fout, V, d, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_linear_update(float* x, const float* eps, const float* noise, const float* a, const float* ce,
                        const float* cz, int B, size_t elems_per_utt, ditto_stream_t stream) {
    if (!x || !eps || !a || !ce || (noise && !cz) || B <= 0 || elems_per_utt == 0)
        return fail(DITTO_ERR_ARG, "bad argument to ditto_linear_update");
    if (elems_per_utt % 4) return fail(DITTO_ERR_SHAPE, "elems_per_utt must be a multiple of 4");
    HIP_TRY(launch_linear_update(x, eps, noise, a, ce, cz, B, elems_per_utt, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_cfg_combine(const float* eps2, float* out, float w, size_t elems_half, ditto_stream_t stream) {
    if (!eps2 || !out || elems_half == 0) return fail(DITTO_ERR_ARG, "bad argument to ditto_cfg_combine");
    if (elems_half % 4) return fail(DITTO_ERR_SHAPE, "elems_half must be a multiple of 4");
    HIP_TRY(launch_cfg_combine(eps2, out, w, elems_half, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_quantize_rows_fp8(const float* src, int rows, int cols, void* dst_fp8, float* scales, ditto_stream_t stream) {
    if (!src || !dst_fp8 || !scales || rows <= 0 || cols <= 0) return fail(DITTO_ERR_ARG, "bad argument to ditto_quantize_rows_fp8");
    if (cols % 4) return fail(DITTO_ERR_SHAPE, "cols must be a multiple of 4");
    HIP_TRY(launch_pack_fp8(src, dst_fp8, scales, rows, cols, cols, 1 << 30, 1, 0, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_layernorm_fp8(const float* x, const float* gamma, const float* beta, void* out_fp8, int M, int d,
                        ditto_stream_t stream) {
    if (!x || !out_fp8 || M <= 0 || d <= 0 || (gamma == nullptr) != (beta == nullptr))
        return fail(DITTO_ERR_ARG, "bad argument to ditto_layernorm_fp8");
    if (d % 4 || d > 2048) return fail(DITTO_ERR_SHAPE, "d must be a multiple of 4 and <= 2048");
    HIP_TRY(launch_layernorm_fp8(x, gamma, beta, out_fp8, d, M, d, (hipStream_t)stream));
    return DITTO_OK;
}

int ditto_gemm_fp8(const void* A, int lda, const void* W, const float* wscale, const float* bias,
                   const float* residual, void* out, int ldo, int M, int N, int K, int epilogue,
                   ditto_stream_t stream) {
    if (!A || !W || !out) return fail(DITTO_ERR_ARG, "null pointer to ditto_gemm_fp8");
    if (K % 128 || N % 16 || lda % 16) return fail(DITTO_ERR_SHAPE, "need K %% 128 == 0, N %% 16 == 0, lda %% 16 == 0");
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.bias = bias; g.residual = residual; g.ldr = ldo; g.out = out; g.ldo = ldo;
    g.M = M; g.N = N; g.K = K; g.fp8 = true; g.wscale = wscale;
    GemmEpilogue e;
    switch (epilogue) {
        case 0: e = EPI_BIAS_BF16; break;
        case 1: e = EPI_BIAS_RES_F32; break;
        case 4: e = EPI_BIAS_F32; break;
        case 5:
            if (!bias || N % 32) return fail(DITTO_ERR_ARG, "gated epilogue needs a bias and N %% 32 == 0");
            e = EPI_GATED_FP8; break;
        default: return fail(DITTO_ERR_ARG, "epilogue must be 0, 1, 4 or 5");
    }
    HIP_TRY(launch_gemm(g, e, (hipStream_t)stream));
    return DITTO_OK;
}

// plain integer switches: one slot each (ditto_get_option reads them; ditto_set_option validates per name below)
static int* option_slot(const char* name) {
    static const struct { const char* n; int* p; } tab[] = {
        {"gemm_tile", &g_gemm_tile}, {"attn_flags", &g_attn_flags}, {"gemm_flags", &g_gemm_flags}, {"gemm_group", &g_gemm_group},
        {"pp_mask", &g_pp_mask}, {"fr_mask", &g_fr_mask}, {"fr_class_rows", &g_fr_class_rows}, {"fr_dgrad", &g_fr_dgrad},
        {"train_flags", &g_train_flags}, {"fr_u_fp8", &g_fr_u_fp8}, {"fr_tile", &g_fr_tile}, {"fr64_maxk", &g_fr64_maxk},
        {"fr_stagger", &g_fr_stagger}, {"fr_rot", &g_fr_rot}, {"pp_nb", &g_pp_nb}, {"pp_stagger", &g_pp_stagger},
        {"splitk_wgs", &g_splitk_wgs}, {"residual_bf16", &g_resid_bf16}, {"lnq", &g_lnq}, {"lnq_ring", &g_lnq_ring}, {"lnq_waves", &g_lnq_waves}, {"qkv_split", &g_qkv_split}, {"ll_mask", &g_ll_mask}, {"lnq_min_rows", &g_lnq_min_rows}, {"attn64p_min_wgs", &g_attn64p_min_wgs}, {"attn64p_min_wgs_plain", &g_attn64p_min_wgs_plain}};
    for (auto& e : tab) if (!strcmp(name, e.n)) return e.p;
    return nullptr;
}

int ditto_get_option(const char* name, int* value) {
    if (!name || !value) return fail(DITTO_ERR_ARG, "null argument to ditto_get_option");
    if (!strcmp(name, "wgrad_wgs")) { *value = get_wgrad_wgs(); return DITTO_OK; }
    if (const int* p = option_slot(name)) { *value = *p; return DITTO_OK; }
    return fail(DITTO_ERR_ARG, "unknown option '%s'", name);
}

int ditto_set_option(const char* name, int value) {
    if (!name) return fail(DITTO_ERR_ARG, "null option name");
    if (!strcmp(name, "gemm_tile")) {
        if (value != 0 && value != 127 && value != 128 && value != 256 && value != 129 && value != 131 && value != 192)
            return fail(DITTO_ERR_ARG, "gemm_tile must be 0, 127 (128x128 deep prefetch), 128, 129 (256x128 ring), "
                                       "131 (128x256 ping-pong, 2 workgroups/CU), 192 (256x192 where the epilogue allows) or 256");
        g_gemm_tile = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "attn_flags")) {
        g_attn_flags = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "wgrad_wgs")) {
        if (value < 0 || value > 2048) return fail(DITTO_ERR_ARG, "wgrad_wgs must be in [0, 2048] (0 = cost model)");
        set_wgrad_wgs(value);
        return DITTO_OK;
    }
    if (!strcmp(name, "gemm_flags")) {
        g_gemm_flags = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "gemm_group")) {
        if (value < 0 || value > 64) return fail(DITTO_ERR_ARG, "gemm_group must be in [0, 64]");
        g_gemm_group = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "pp_mask")) {
        if (value < -1 || value > 63) return fail(DITTO_ERR_ARG, "pp_mask must be in [-1, 63]");
        g_pp_mask = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "fr_mask")) {
        if (value < 0 || value > 3) return fail(DITTO_ERR_ARG, "fr_mask must be in [0, 3]");
        g_fr_mask = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "fr_class_rows")) {
        if (value < 0) return fail(DITTO_ERR_ARG, "fr_class_rows must be >= 0 (0 = every launch decides on its own rows)");
        g_fr_class_rows = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "fr_dgrad")) {
        if (value < 0 || value > 3) return fail(DITTO_ERR_ARG, "fr_dgrad must be in [0, 3]");
        g_fr_dgrad = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "train_flags")) {
        if (value < 0 || value > 31) return fail(DITTO_ERR_ARG, "train_flags must be in [0, 31]");
        g_train_flags = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "fr_u_fp8")) {   // test hook: ditto_gemm_ln_bf16 writes u as fp8 e4m3 bytes ([M, ldu] bytes; N = 1024)
        g_fr_u_fp8 = value != 0;
        return DITTO_OK;
    }
    if (!strcmp(name, "fr_tile")) {
        if (value != 0 && value != 64 && value != 130) return fail(DITTO_ERR_ARG, "fr_tile must be 0 (rule), 64 or 130 (128 rows, W straight into registers)");
        g_fr_tile = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "fr64_maxk")) {
        if (value < 0) return fail(DITTO_ERR_ARG, "fr64_maxk must be >= 0");
        g_fr64_maxk = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "fr_stagger")) {
        if (value < 0 || value > 100000) return fail(DITTO_ERR_ARG, "fr_stagger must be in [0, 100000] (10 ns ticks)");
        g_fr_stagger = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "fr_rot")) {
        if (value < 0 || value > 4096) return fail(DITTO_ERR_ARG, "fr_rot must be in [0, 4096]");
        g_fr_rot = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "pp_nb")) {
        if (value != 0 && value != 3 && value != 4) return fail(DITTO_ERR_ARG, "pp_nb must be 0 (rule), 3 (192-wide tiles) or 4 (256-wide)");
        g_pp_nb = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "pp_stagger")) {
        if (value < -1 || value > 100000) return fail(DITTO_ERR_ARG, "pp_stagger must be in [-1, 100000] (10 ns ticks; -1 = rule)");
        g_pp_stagger = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "lnq_min_rows")) {
        if (value < 0) return fail(DITTO_ERR_ARG, "lnq_min_rows must be >= 0");
        g_lnq_min_rows = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "attn64p_min_wgs") || !strcmp(name, "attn64p_min_wgs_plain")) {
        if (value < 1) return fail(DITTO_ERR_ARG, "attn64p_min_wgs[_plain] must be >= 1");
        (name[15] ? g_attn64p_min_wgs_plain : g_attn64p_min_wgs) = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "ll_mask")) {
        if (value < 0 || value > 3) return fail(DITTO_ERR_ARG, "ll_mask must be in [0, 3]");
        g_ll_mask = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "lnq_waves")) {
        if (value != 4 && value != 8) return fail(DITTO_ERR_ARG, "lnq_waves must be 4 (one wave per SIMD) or 8 (two)");
        g_lnq_waves = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "qkv_split")) {
        if (value < 0 || value > 64) return fail(DITTO_ERR_ARG, "qkv_split must be in [0, 64]");
        g_qkv_split = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "lnq_ring")) {
        if (value != 0 && value != 2 && value != 4 && value != 8) return fail(DITTO_ERR_ARG, "lnq_ring must be 0 (default), 4 or 8 (shape 32), 2 or 4 (shape 16)");
        g_lnq_ring = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "lnq")) {
        if (value != 0 && value != 16 && value != 32) return fail(DITTO_ERR_ARG, "lnq must be 0 (off), 32 or 16 (MFMA shape)");
        g_lnq = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "residual_bf16")) {
        if (value < 0 || value > 1) return fail(DITTO_ERR_ARG, "residual_bf16 must be 0 or 1");
        g_resid_bf16 = value;
        return DITTO_OK;
    }
    if (!strcmp(name, "splitk_wgs")) {
        if (value < -1 || value > 2048) return fail(DITTO_ERR_ARG, "splitk_wgs must be in [-1, 2048] (-1 never, 0 the low-latency class rule, > 0 a workgroup target)");
        g_splitk_wgs = value;
        return DITTO_OK;
    }
    return fail(DITTO_ERR_ARG, "unknown option '%s'", name);
}

int ditto_full_row_plan(const ditto_config* cfg, int B, int N, int* outproj, int* fc2) {
    return ditto_full_row_plan_opts(cfg, B, N, nullptr, outproj, fc2, nullptr);
}

int ditto_full_row_plan_opts(const ditto_config* cfg, int B, int N, const ditto_call_opts* opts, int* outproj, int* fc2,
                             int* stream_bf16) {
    if (int rc = check_call_opts(opts)) return rc;
    CallScope scope(opts);
    if (int rc = check_cfg(cfg)) return rc;
    if (!outproj || !fc2 || B <= 0 || N <= 0) return fail(DITTO_ERR_ARG, "bad argument to ditto_full_row_plan");
    if ((long long)B * N > 0x7fffffffLL) return fail(DITTO_ERR_SHAPE, "B * N exceeds 2^31 - 1 rows");
    const int d = cfg->hidden_dim, M = B * N;
    const bool fp8c = (cfg->flags & DITTO_CFG_FP8_LINEAR) != 0;          // plan_arena packs the stage-major copies for these:
    if (int rc = check_class_pin(M, d, fp8c, !cfg_padded(*cfg))) return rc;   // what the forward itself would answer
    const bool padc = cfg_padded(*cfg);                                   // padded heads: the tiled path only
    const bool have_o = !padc && ((d == 768 && !fp8c) || d == 1024), have_2 = !padc && (d == 768 || d == 1024) && !fp8c;
    *outproj = have_o && (opt_fr_mask() & 1) && fr_outproj_ok(M, d);
    *fc2 = have_2 && (opt_fr_mask() & 2) && fr_fc2_ok(M, d);
    if (stream_bf16)   // ditto_forward's hb_class
        *stream_bf16 = opt_resid_bf16() && !fp8c && d == 768 && d / cfg->num_heads == 64 && *outproj && *fc2 && M >= 128 &&
                       (g_fr_tile == 130 || (g_fr_tile == 0 && fr_rule_rows(opt_class_rows() > 0 ? opt_class_rows() : M) == 130));
    return DITTO_OK;
}

int ditto_profile_enable(ditto_model_t m, int enable) {
    if (!m) return fail(DITTO_ERR_ARG, "null model");
    m->prof = enable != 0;
    return DITTO_OK;
}

int ditto_profile_read(ditto_model_t m, int32_t* launches, float* ms) {
    if (!m || !launches || !ms) return fail(DITTO_ERR_ARG, "null argument to ditto_profile_read");
    for (auto& r : m->recs) {
        HIP_TRY(hipEventSynchronize(r.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        m->launches[r.kc] += 1;
        m->ms[r.kc] += t;
        m->pool.push_back(r.a);
        m->pool.push_back(r.b);
    }
    m->recs.clear();
    for (int i = 0; i < DITTO_KC_COUNT; ++i) {
        launches[i] = m->launches[i]; ms[i] = m->ms[i];
        m->launches[i] = 0; m->ms[i] = 0.f;
    }
    return DITTO_OK;
}

}  // extern "C"
