// model.h — internal: the model handle, arena / workspace plans and error helpers shared by ditto_api.hip
// (inference entry points) and ditto_train.hip (training entry points).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/ditto_hip.h"
#include "kernels.h"

namespace ditto {

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return ::ditto::fail(DITTO_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct LayerPack {
    const void *Wqkv, *Wcq, *Wco, *W1g, *W2;
    const void *WcqP = nullptr, *WcqP32 = nullptr;   // stage-major images of the cross q-projection for gemm_lnq.hip (d == 768 bf16): k-group 16 / 32
    const void *WcoP = nullptr, *W2P = nullptr;   // stage-major copies for the full-row kernels (d == 768 bf16; d == 1024: WcoP always, W2P bf16)
    const float *sqkv, *s1g, *s2;   // fp8 weight scales (DITTO_CFG_FP8_LINEAR)
    const float *bqkv, *bcq, *bco, *b1g, *b2;
    const float *g1, *be1, *g2, *be2, *g3, *be3;
};
// transposed bf16 copies for the dgrad GEMMs (dX = dY W runs as the forward GEMM with weight W^T), ditto_train_attach
// plus UNSCALED forward copies of the two q projections (the inference pack folds the softmax scale into them)
struct LayerPackT {
    const void *WqkvT, *WcqT, *WcoT, *W1gT, *W2T;
    const void *Wqkv_u, *Wcq_u; const float *bqkv_u, *bcq_u;
    const void *W1gTP, *WqkvTP;   // stage-major copies of W1gT / WqkvT for the full-row kernel (d == 768 only, else null)
};

struct ArenaPlan {
    size_t total = 0;
    struct L { size_t Wqkv, Wcq, Wco, W1g, W2, bqkv, bcq, bco, b1g, b2, g1, be1, g2, be2, g3, be3, sqkv, s1g, s2, WcoP, W2P, WcqP, WcqP32; };
    std::vector<L> layers;
    size_t Wkv, bkv, Wfin, bfin, ttab, wx, bx, invf, invf_rev;
};
ArenaPlan plan_arena(const ditto_config& c);

// head_dim % 64 != 0: inside the block the heads of q / k / v sit at a stride of dhp = roundup(head_dim, 64) columns with zero
// pads (rowwise.hip "Head padding"); dp = H * dhp is the width of the attention-side buffers.  dhp == head_dim otherwise.
inline int cfg_dhp(const ditto_config& c) { const int dh = c.hidden_dim / c.num_heads; return (dh + 63) / 64 * 64; }
inline int cfg_dp(const ditto_config& c) { return c.num_heads * cfg_dhp(c); }
inline bool cfg_padded(const ditto_config& c) { return cfg_dp(c) != c.hidden_dim; }

struct WsPlan { size_t h, u, qkv, act, xcat, eps, attn, attn_bytes, splitk, splitk_bytes, total; };
// K-splits of a long-K, few-tile GEMM [M, N] x K at small batch (1 = none): ditto_api.hip
int small_batch_k_splits(int M, int N, int K);
int small_batch_k_splits_outproj(int M, int K);
WsPlan plan_ws(const ditto_config& c, int B, int N, int T);
int check_cfg(const ditto_config* c);
void set_wgrad_wgs(int v);   // ditto_train.hip: split-K target of the wgrad GEMMs (ditto_set_option("wgrad_wgs"))
int get_wgrad_wgs();
int check_class_pin(int M, int d, bool fp8, bool has_fr);   // ditto_api.hip: a pinned full-row class that this launch cannot take -> error

// ditto_call_opts (ABI 9): validated by check_call_opts, then in force for the length of ONE call through a CallScope on the
// calling thread (kernels.h t_opts); fields left at -1 inherit what is in force around the call (an enclosing
// ditto_call_opts_push scope of the thread, else the process default).
int check_call_opts(const ditto_call_opts* o);
struct CallScope {
    CallOpts saved;
    explicit CallScope(const ditto_call_opts* o) : saved(t_opts) { apply(o); }
    ~CallScope() { t_opts = saved; }
    CallScope(const CallScope&) = delete;
    CallScope& operator=(const CallScope&) = delete;
    static void apply(const ditto_call_opts* o) {
        if (!o) return;
        if (o->class_rows >= 0) t_opts.class_rows = o->class_rows;
        if (o->residual_bf16 >= 0) t_opts.resid_bf16 = o->residual_bf16;
        if (o->fr_mask >= 0) t_opts.fr_mask = o->fr_mask;
        if (o->lnq >= 0) t_opts.lnq = o->lnq;
    }
};

}  // namespace ditto

struct ditto_model {
    ditto_config cfg;
    ditto::ArenaPlan plan;
    char* arena;
    std::vector<ditto::LayerPack> layers;
    const void* Wkv; const float* bkv; const void* Wfin; const float* bfin;
    const float* ttab; const float* wx; const float* bx; const float* invf;
    const float* invf_rev = nullptr;   // inv_freq / (2 pi) for the in-epilogue RoPE angles (head_dim 64)
    bool blocks_only = false;    // created without the model-level weights: only ditto_block_forward works
    // training (ditto_train_attach)
    std::vector<ditto::LayerPackT> layersT;
    const void* WoutT = nullptr;
    // what ditto_train_forward DECIDED for the tape it wrote (keyed by the tape's address): the backward reads the tape as it was
    // written, whatever the options say by then (a bf16 tape read as fp32 is garbage gradients without an error)
    struct TapeRec { int B, N, T; bool hb; };
    std::mutex tape_mu;
    std::unordered_map<const void*, TapeRec> tapes;
    // profiling
    bool prof = false;
    struct Rec { hipEvent_t a, b; int kc; };
    std::vector<Rec> recs;       // pending, un-synchronised
    std::vector<hipEvent_t> pool;
    int32_t launches[DITTO_KC_COUNT] = {0};
    float ms[DITTO_KC_COUNT] = {0};
};
