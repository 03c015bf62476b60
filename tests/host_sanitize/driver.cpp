// Host-side sanitizer driver.
//
// Linked against ASan + UBSan builds of the HOST pass of the three files that do offset / size arithmetic and argument
// validation (csrc/ditto_api.hip, csrc/ditto_train.hip, csrc/slp.hip; device code unchanged: there is no GPU
// sanitizer on this pool) and the ordinary objects of the kernel files.  It needs no GPU: every call below is either host
// arithmetic (the *_bytes queries, ditto_full_row_plan), an argument check that must refuse BEFORE any HIP call, or a
// call that reaches the first HIP call and must come back with DITTO_ERR_HIP without leaking its half-built handle.
// tests/test_host_sanitize.py builds and runs it; any sanitizer report or failed expectation makes the exit code non-zero.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ditto_hip.h"

static int g_fail = 0;
#define EXPECT(cond, ...)                                           \
    do {                                                            \
        if (!(cond)) {                                              \
            ++g_fail;                                               \
            std::fprintf(stderr, "FAIL %s:%d: %s | ", __FILE__, __LINE__, #cond); \
            std::fprintf(stderr, __VA_ARGS__);                      \
            std::fprintf(stderr, "\n");                             \
        }                                                           \
    } while (0)

static bool cfg_valid(const ditto_config& c) {
    if (c.hidden_dim <= 0 || c.num_layers <= 0 || c.num_heads <= 0 || c.time_dim <= 0 || c.diffusion_steps <= 0) return false;
    if (c.hidden_dim % c.num_heads || c.text_dim != c.hidden_dim || c.hidden_dim % 64 || c.hidden_dim > 2048) return false;
    const int dh = c.hidden_dim / c.num_heads, dhp = (dh + 63) / 64 * 64;   // head_dim % 64 != 0 runs on heads padded to dhp
    if (dh % 2) return false;
    if (dh % 64 && ((c.flags & DITTO_CFG_FP8_LINEAR) || 4 * c.hidden_dim < c.num_heads * dhp)) return false;
    if ((c.flags & DITTO_CFG_FP8_LINEAR) && c.hidden_dim % 128) return false;
    if (c.flags & ~DITTO_CFG_FP8_LINEAR) return false;
    return true;
}

static void size_queries() {
    const int ds[] = {0, -64, 64, 96, 128, 192, 256, 768, 1024, 1472, 2048, 2112};
    const int Ls[] = {0, 1, 5, 12, 24};
    const int Hs[] = {0, 1, 3, 4, 12, 16, 23};
    const int flags[] = {0, DITTO_CFG_FP8_LINEAR, 2};
    const int Bs[] = {-1, 0, 1, 3, 21, 32, 256};
    const int Ns[] = {0, 1, 63, 64, 1000, 1024, 4096};
    const int Ts[] = {0, 1, 96, 1024};
    long n_ok = 0, n_bad = 0;
    for (int d : ds) for (int L : Ls) for (int H : Hs) for (int fl : flags) {
        ditto_config c{d, L, H, 256, d, 50, fl};
        const bool ok = cfg_valid(c);
        const size_t arena = ditto_arena_bytes(&c);
        EXPECT((arena > 0) == ok, "arena_bytes d=%d L=%d H=%d flags=%d -> %zu (%s)", d, L, H, fl, arena, ditto_last_error());
        if (!ok) {
            ++n_bad;
            EXPECT(std::strlen(ditto_last_error()) > 0, "refused config without a message");
            EXPECT(ditto_workspace_bytes(&c, 1, 64, 64) == 0 && ditto_cond_bytes(&c, 1, 64) == 0 &&
                   ditto_train_arena_bytes(&c) == 0 && ditto_tape_bytes(&c, 1, 64, 64) == 0 &&
                   ditto_train_workspace_bytes(&c, 1, 64, 64) == 0, "a refused config got a size");
            int a = 7, b = 7;
            EXPECT(ditto_full_row_plan(&c, 1, 64, &a, &b) != DITTO_OK, "full_row_plan accepted a refused config");
            continue;
        }
        ++n_ok;
        EXPECT(arena % 256 == 0, "arena not 256-aligned");
        // bf16 weights alone: L (3 + 1 + 1 + 8 + 4) d^2 + L 2 d^2 (cross K/V) + 2 d^2 elements of 1 (fp8: some) or 2 bytes
        EXPECT(arena >= (size_t)L * 19 * d * d, "arena smaller than its weights: %zu", arena);
        const size_t tarena = ditto_train_arena_bytes(&c);
        const bool padded = (d / H) % 64 != 0;      // padded heads are forward-only: no training sizes
        EXPECT((tarena > 0) == (!(fl & DITTO_CFG_FP8_LINEAR) && !padded) || (tarena > 0 && !padded), "train arena");
        for (int B : Bs) for (int N : Ns) for (int T : Ts) {
            const bool shape_ok = B > 0 && N > 0 && T > 0;
            const size_t ws = ditto_workspace_bytes(&c, B, N, T), cond = ditto_cond_bytes(&c, B, T);
            const size_t tape = ditto_tape_bytes(&c, B, N, T), tws = ditto_train_workspace_bytes(&c, B, N, T);
            EXPECT((ws > 0) == shape_ok, "workspace_bytes B=%d N=%d T=%d -> %zu", B, N, T, ws);
            EXPECT((cond > 0) == (B > 0 && T > 0), "cond_bytes B=%d T=%d -> %zu", B, T, cond);
            if (!shape_ok) {
                EXPECT(tape == 0 && tws == 0, "tape / train workspace for a refused shape");
                continue;
            }
            const size_t M = (size_t)B * N;
            EXPECT(ws % 256 == 0 && cond % 256 == 0, "sizes not 256-aligned");
            // h fp32 + u bf16 + qkv bf16 x3 + act bf16 x4 + xcat bf16 x2 + eps fp32 (u / act one byte under fp8)
            EXPECT(ws >= M * d * (4 + 1 + 6 + 4 + 4 + 4), "workspace too small for its streams: %zu", ws);
            EXPECT(cond >= (size_t)B * T * L * 2 * d * 2, "cond too small for the K/V cache");
            if (tape) EXPECT(tape >= M * d * 4 * (3 * (size_t)L + 1), "tape smaller than its residual snapshots");
            int a = -1, b = -1;
            EXPECT(ditto_full_row_plan(&c, B, N, &a, &b) == DITTO_OK && (a == 0 || a == 1) && (b == 0 || b == 1),
                   "full_row_plan B=%d N=%d", B, N);
            const bool f8 = (fl & DITTO_CFG_FP8_LINEAR) != 0;
            if (padded) EXPECT(a == 0 && b == 0, "full-row plan on padded heads");
            else if (d == 768) { if (f8 || M < 176 * 64 - 63) EXPECT(a == 0 && b == 0, "full-row plan where it cannot run (d = 768)"); }
            else if (d == 1024) {   // 64-row tiles, >= 192 of them; fc2 never under fp8
                if (M < 192 * 64 - 63) EXPECT(a == 0 && b == 0, "full-row plan where it cannot run (d = 1024)");
                if (f8) EXPECT(b == 0, "fp8 fc2 on the full-row kernel");
            } else EXPECT(a == 0 && b == 0, "full-row plan at a width without a kernel");
        }
        // (the workspace is NOT monotone in B by design: batches small enough for the split-K mode reserve its partials)
    }
    EXPECT(n_ok > 20 && n_bad > 20, "grid degenerate: %ld valid, %ld refused", n_ok, n_bad);
    EXPECT(ditto_arena_bytes(nullptr) == 0 && ditto_workspace_bytes(nullptr, 1, 1, 1) == 0 && ditto_cond_bytes(nullptr, 1, 1) == 0,
           "null config got a size");

    // stand-alone kernel scratch queries, including shapes a caller could reach with a long-form batch
    for (int B : {1, 8, 32, 256}) for (int H : {1, 12, 16}) for (int S : {1, 64, 1000, 4096}) for (int dh : {64, 128, 768, 96}) {
        const size_t a = ditto_attention_workspace_bytes(B, H, S, S, dh);
        if (dh == 64) EXPECT(a == 0, "the fused head_dim-64 path needs no workspace, got %zu", a);
        if (dh % 64 == 0 && dh != 64) EXPECT(a > 0, "generic attention without workspace");
        (void)ditto_attention_bwd_workspace_bytes(B, H, S, S, dh);
        (void)ditto_attention_causal_workspace_bytes(B, H, S, S, dh);
    }
    for (int rows : {1, 127, 128, 1024, 1 << 20}) for (int groups : {1, 32}) for (int d : {64, 768, 1024, 2048}) {
        EXPECT(ditto_layernorm_bwd_scratch_bytes(rows, groups, d) > 0, "layernorm_bwd_scratch_bytes(%d, %d, %d)", rows, groups, d);
    }
    for (int B : {1, 32}) for (int d : {256, 768, 1024})
        EXPECT(ditto_global_adaln_scratch_bytes(B, d, 256, d) > 0, "global_adaln_scratch_bytes(%d, %d)", B, d);

    // speech-length predictor: byt5-small geometry (d = 1472, 4 heads of 368 -> padded to 384) and the bad ones
    const ditto_slp_config good[] = {{1472, 4, 4, 1472 * 4, 2048}, {1472, 1, 1, 1472, 2048}, {256, 4, 2, 1024, 16}, {64, 1, 1, 64, 1}};
    for (const auto& sc : good) {
        const size_t a = ditto_slp_arena_bytes(&sc);
        EXPECT(a > 0 && a % 256 == 0, "slp arena d=%d: %zu (%s)", sc.d_model, a, ditto_last_error());
        for (int B : {1, 8}) for (int S : {1, 100, 2048}) for (int T : {1, 128})
            EXPECT(ditto_slp_workspace_bytes(&sc, B, S, T) > 0, "slp workspace B=%d S=%d T=%d", B, S, T);
        EXPECT(ditto_slp_workspace_bytes(&sc, 0, 8, 8) == 0 && ditto_slp_workspace_bytes(&sc, 1, 0, 8) == 0, "slp workspace for an empty batch");
    }
    const ditto_slp_config bad[] = {{0, 1, 1, 64, 4}, {100, 4, 1, 128, 4}, {128, 3, 1, 128, 4}, {128, 2, 0, 128, 4}, {128, 2, 1, 100, 4}, {128, 2, 1, 128, 0}, {2112, 4, 1, 128, 4}};
    for (const auto& sc : bad) EXPECT(ditto_slp_arena_bytes(&sc) == 0, "slp config d=%d h=%d accepted", sc.d_model, sc.nhead);
    EXPECT(ditto_slp_arena_bytes(nullptr) == 0, "slp null config");
}

static void refusals_before_any_gpu_call() {
    const ditto_config c{768, 12, 12, 256, 768, 50, 0};
    ditto_model_t m = nullptr;
    char fake[64];
    void* p = fake;            // a non-null pointer that must never be dereferenced on these paths
    EXPECT(ditto_model_create(nullptr, nullptr, nullptr, 0, nullptr, nullptr) == DITTO_ERR_ARG, "null create");
    ditto_weights w{};
    EXPECT(ditto_model_create(&c, &w, p, 1, nullptr, &m) == DITTO_ERR_ARG, "create without layers");
    std::vector<ditto_layer_weights> lw(12);
    w.layers = lw.data();
    EXPECT(ditto_model_create(&c, &w, p, 1, nullptr, &m) == DITTO_ERR_SIZE, "arena of 1 byte accepted: %s", ditto_last_error());
    const size_t need = ditto_arena_bytes(&c);
    EXPECT(ditto_model_create(&c, &w, reinterpret_cast<void*>((uintptr_t)128), need, nullptr, &m) == DITTO_ERR_ARG, "misaligned arena accepted");
    EXPECT(ditto_model_create(&c, &w, reinterpret_cast<void*>((uintptr_t)256), need, nullptr, &m) == DITTO_ERR_ARG && m == nullptr,
           "null layer weights accepted: %s", ditto_last_error());
    EXPECT(ditto_model_destroy(nullptr) == DITTO_OK, "destroy(null)");

    EXPECT(ditto_forward(nullptr, nullptr, nullptr, nullptr, 1, 1, 1, nullptr, nullptr, nullptr, nullptr, 0, nullptr) == DITTO_ERR_ARG, "forward(null)");
    EXPECT(ditto_text_precompute(nullptr, nullptr, 1, 1, nullptr, 0, nullptr, 0, nullptr) == DITTO_ERR_ARG, "text_precompute(null)");
    EXPECT(ditto_rope_tables(nullptr, 4, nullptr, nullptr, nullptr) == DITTO_ERR_ARG, "rope_tables(null)");
    EXPECT(ditto_p_sample_update(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 4, nullptr) == DITTO_ERR_ARG, "p_sample_update(null)");
    EXPECT(ditto_gemm_bf16(nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 1, 1, 1, 0, nullptr) != DITTO_OK, "gemm(null)");
    EXPECT(ditto_attention_bf16(p, 64, p, 64, p, 64, p, 64, 1, 1, 64, 64, 96, 0.1f, nullptr, 0, nullptr) == DITTO_ERR_SHAPE, "attention head_dim 96");
    EXPECT(ditto_attention_bf16(p, 768, p, 768, p, 768, p, 768, 1, 1, 64, 64, 768, 0.1f, nullptr, 0, nullptr) == DITTO_ERR_SIZE, "generic attention without workspace");
    // the full-row GEMM addresses A with 32-bit byte offsets: refuse M * lda * 2 >= 2^32 for the lda actually passed
    float* fo = reinterpret_cast<float*>(p);
    EXPECT(ditto_gemm_ln_bf16(p, 3072, p, nullptr, nullptr, fo, 768, nullptr, nullptr, nullptr, 0, 699392, 768, 3072, nullptr) == DITTO_ERR_SHAPE, "gemm_ln accepted wrapping A offsets");
    EXPECT(ditto_gemm_ln_bf16(p, 768, p, nullptr, nullptr, fo, 768, nullptr, nullptr, nullptr, 0, 63, 768, 768, nullptr) == DITTO_ERR_SHAPE, "gemm_ln M < 64 (one 64-row tile is the least the full-row kernels take)");
    EXPECT(ditto_gemm_ln_bf16(p, 768, p, nullptr, nullptr, fo, 768, nullptr, nullptr, nullptr, 0, 256, 512, 768, nullptr) == DITTO_ERR_SHAPE, "gemm_ln N != 768");
    EXPECT(ditto_gemm_ln_bf16(p, 768, p, nullptr, nullptr, fo, 768, reinterpret_cast<const float*>(p), nullptr, nullptr, 0, 256, 768, 768, nullptr) == DITTO_ERR_ARG, "gemm_ln gamma without beta");
    EXPECT(ditto_gemm_tn_bf16(nullptr, 8, nullptr, 8, nullptr, 8, 8, 8, 8, 1, 128, nullptr, 0, nullptr) != DITTO_OK, "gemm_tn(null)");
    EXPECT(ditto_vq_argmin(nullptr, nullptr, nullptr, 1, 1, 1, nullptr, nullptr) == DITTO_ERR_ARG, "vq_argmin(null)");
    EXPECT(ditto_train_attach(nullptr, nullptr, nullptr, 0, nullptr) != DITTO_OK, "train_attach(null)");
    EXPECT(ditto_profile_read(nullptr, nullptr, nullptr) == DITTO_ERR_ARG && ditto_profile_enable(nullptr, 1) == DITTO_ERR_ARG, "profile(null)");
    int a = 0, b = 0;
    EXPECT(ditto_full_row_plan(&c, 1 << 20, 1 << 12, &a, &b) == DITTO_ERR_SHAPE, "full_row_plan with B * N > 2^31");
    EXPECT(ditto_full_row_plan(&c, 32, 1024, nullptr, &b) == DITTO_ERR_ARG, "full_row_plan(null)");

    EXPECT(ditto_set_option("no_such_option", 1) == DITTO_ERR_ARG && std::strstr(ditto_last_error(), "no_such_option"), "unknown option");
    EXPECT(ditto_set_option(nullptr, 1) != DITTO_OK, "set_option(null)");
    EXPECT(ditto_set_option("fr_mask", 4) == DITTO_ERR_ARG && ditto_set_option("fr_mask", 3) == DITTO_OK, "fr_mask range");
    EXPECT(ditto_set_option("fr_class_rows", -5) == DITTO_ERR_ARG && ditto_set_option("fr_class_rows", 0) == DITTO_OK, "fr_class_rows range");
    for (int kc = -2; kc < 20; ++kc) EXPECT(ditto_kernel_class_name(kc) != nullptr, "kernel_class_name(%d)", kc);

    const ditto_slp_config sc{256, 4, 2, 1024, 16};
    ditto_slp_t sm = nullptr;
    EXPECT(ditto_slp_create(&sc, nullptr, p, 1 << 20, nullptr, &sm) != DITTO_OK && sm == nullptr, "slp_create(null weights)");
    ditto_slp_destroy(nullptr);
}

// Everything below reaches the first HIP call: on a GPU-less host that call fails, and the entry point must come back
// with DITTO_ERR_HIP and free whatever it had built (LeakSanitizer watches).  On a host WITH a GPU this part is skipped:
// the pointers are not device memory.
static void error_path_after_the_first_hip_call(bool have_gpu) {
    if (have_gpu) return;
    const ditto_config cs[] = {{256, 2, 4, 64, 256, 10, 0}, {768, 3, 12, 256, 768, 50, 0}, {1024, 2, 16, 256, 1024, 50, DITTO_CFG_FP8_LINEAR}};
    for (const auto& c : cs) {
        const size_t need = ditto_arena_bytes(&c);
        void* arena = nullptr;
        if (posix_memalign(&arena, 256, need ? need : 256)) { ++g_fail; return; }
        std::vector<float> dummy(16);
        std::vector<ditto_layer_weights> lw(c.num_layers);
        for (auto& l : lw) {
            const float** f = reinterpret_cast<const float**>(&l);
            for (size_t i = 0; i < sizeof(l) / sizeof(float*); ++i) f[i] = dummy.data();
        }
        ditto_weights w{};
        w.layers = lw.data();
        ditto_model_t m = nullptr;
        const int rc = ditto_model_create(&c, &w, arena, need, nullptr, &m);     // blocks-only form
        EXPECT(rc == DITTO_ERR_HIP && m == nullptr, "create on a GPU-less host: rc %d (%s)", rc, ditto_last_error());
        const float** g = reinterpret_cast<const float**>(&w);
        for (size_t i = 0; i + 1 < sizeof(w) / sizeof(float*); ++i) g[i] = dummy.data();
        const int rc2 = ditto_model_create(&c, &w, arena, need, nullptr, &m);    // full form
        EXPECT(rc2 == DITTO_ERR_HIP && m == nullptr, "create (full) on a GPU-less host: rc %d", rc2);
        std::free(arena);
    }
    const ditto_slp_config sc{256, 4, 2, 1024, 16};
    const size_t sneed = ditto_slp_arena_bytes(&sc);
    void* sarena = nullptr;
    if (posix_memalign(&sarena, 256, sneed)) { ++g_fail; return; }
    std::vector<float> dummy(16);
    std::vector<ditto_slp_layer_weights> slw(sc.num_layers);
    for (auto& l : slw) {
        const float** f = reinterpret_cast<const float**>(&l);
        for (size_t i = 0; i < sizeof(l) / sizeof(float*); ++i) f[i] = dummy.data();
    }
    ditto_slp_weights sw{slw.data(), dummy.data(), dummy.data()};
    ditto_slp_t sm = nullptr;
    const int rc = ditto_slp_create(&sc, &sw, sarena, sneed, nullptr, &sm);
    EXPECT(rc == DITTO_ERR_HIP && sm == nullptr, "slp_create on a GPU-less host: rc %d (%s)", rc, ditto_last_error());
    std::free(sarena);
}

int main(int argc, char** argv) {
    const bool have_gpu = argc > 1 && !std::strcmp(argv[1], "--have-gpu");
    EXPECT(ditto_abi_version() == DITTO_ABI_VERSION, "abi version");
    size_queries();
    refusals_before_any_gpu_call();
    error_path_after_the_first_hip_call(have_gpu);
    if (g_fail) {
        std::fprintf(stderr, "%d expectation(s) failed\n", g_fail);
        return 1;
    }
    std::printf("host sanitizer driver: ok\n");
    return 0;
}
