// probe_attn64p.hip — stand-alone A/B of the head_dim-64 attention forward kernels (no torch, no library): attn64v2 (three waves per
// SIMD, 32 queries per wave — the shipped kernel of rounds 2-5) against attn64p (two waves per SIMD, 64 queries per wave, round 6), its
// knock-out builds (what a launch costs without its softmax / DMA / barrier / fragment reads / MFMAs) and the two ping-pong experiments
// (tools/attn_pingpong_experiments.h); same inputs, interleaved rounds, HIP-event timing, outputs compared element by element.
// Output of the round: profiles/r06_attn_probe.txt.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -fno-slp-vectorize -I include -I ditto_tts_amd/csrc tools/probe_attn64p.hip -o build/probe_attn64p
//   build/probe_attn64p [B H Sq Skv rounds iters]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <type_traits>
#include <vector>

#include "attn_common.h"

namespace ditto {
namespace {
#include "attn64v2.h"
#include "attn64p.h"
#include "attn64q.h"
#include "../../tools/attn_pingpong_experiments.h"
}  // namespace
}  // namespace ditto
using namespace ditto;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static float gauss(unsigned long long& s) {
    auto nxt = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) + 1) / 16777217.0f; };
    const float u1 = nxt(), u2 = nxt();
    return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 12;
    const int Sq = argc > 3 ? atoi(argv[3]) : 1024, Skv = argc > 4 ? atoi(argv[4]) : 1024;
    const int rounds = argc > 5 ? atoi(argv[5]) : 7, iters = argc > 6 ? atoi(argv[6]) : 20;
    const int nset = argc > 7 ? atoi(argv[7]) : 1;   // > 1: rotate over that many copies of q / k / v (operands from HBM, as in the model)
    const float big = argc > 8 ? atof(argv[8]) : 0.f;   // > 0: the queries of the LAST batch item are scaled by this (logits out of attn64q's optimistic range)
    const int d = H * 64;
    const size_t nq = (size_t)B * Sq * d, nk = (size_t)B * Skv * d;
    std::vector<unsigned short> hq(nq), hk(nk), hv(nk);
    unsigned long long s = 12345;
    const float qs = 1.4426950408889634f / 8.0f;
    for (auto& x : hq) x = f2bf(gauss(s) * qs);
    if (big > 0.f) for (size_t i = (size_t)(B - 1) * Sq * d; i < nq; ++i) hq[i] = f2bf(bf2f(hq[i]) * big);
    for (auto& x : hk) x = f2bf(gauss(s));
    for (auto& x : hv) x = f2bf(gauss(s));
    bf16 *q, *k, *v, *o[6], *rin;
    CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&k, nk * 2)); CK(hipMalloc(&v, nk * 2));
    for (auto& x : o) { CK(hipMalloc(&x, nq * 4)); CK(hipMemset(x, 0, nq * 4)); }
    CK(hipMalloc(&rin, nq * 4)); CK(hipMemset(rin, 0, nq * 4));
    CK(hipMemcpy(rin, hv.data(), nq * 2 < nk * 2 ? nq * 2 : nk * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, hq.data(), nq * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(k, hk.data(), nk * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, hv.data(), nk * 2, hipMemcpyHostToDevice));
    std::vector<bf16*> qset(nset, q), kset(nset, k), vset(nset, v);
    for (int i = 1; i < nset; ++i) {
        CK(hipMalloc(&qset[i], nq * 2)); CK(hipMalloc(&kset[i], nk * 2)); CK(hipMalloc(&vset[i], nk * 2));
        CK(hipMemcpy(qset[i], q, nq * 2, hipMemcpyDeviceToDevice)); CK(hipMemcpy(kset[i], k, nk * 2, hipMemcpyDeviceToDevice)); CK(hipMemcpy(vset[i], v, nk * 2, hipMemcpyDeviceToDevice));
    }
    int rot = 0;
    AttnParams p{};
    p.q = q; p.ldq = d; p.k = k; p.ldk = d; p.v = v; p.ldv = d; p.ldo = d; p.resid = nullptr; p.ldr = d; p.resid_in = nullptr;
    p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.scale_log2 = 1.0f;
    hipStream_t st = nullptr;
    const int NV = 14;
    const char* names[NV] = {"attn64v2 (3 waves/SIMD, 32 q/wave)", "attn64p (NBUF = 4)", "attn64q (pipelined, optimistic)",
                             "attn64q, fragments held for block B", "attn64q, held, 6 slots ahead",
                             "attn64q: no softmax", "attn64q: no row-sum adds", "attn64q: no DMA", "attn64q: no DMA, no barrier",
                             "attn64q: no DMA, no barrier, no adds", "attn64q: no DMA, no barrier, no softmax", "attn64p: MFMA only",
                             "attn64p: softmax only (no MFMA)", "attn64p: no MFMA (DMA, LDS, softmax)"};
    auto launch = [&](int var) {
        AttnParams pp = p;
        rot = rot + 1 == nset ? 0 : rot + 1;
        pp.q = qset[rot]; pp.k = kset[rot]; pp.v = vset[rot];
        pp.out = o[var < 5 ? var : 5];
        pp.nqb = var == 0 ? (Sq + 127) / 128 : (Sq + 255) / 256;
        const dim3 grid(pp.nqb * H * B), blk(256);
        switch (var) {
            case 0: hipLaunchKernelGGL((attn64v2_kernel<false, 3>), grid, blk, 0, st, pp); break;
            case 1: hipLaunchKernelGGL((attn64p_kernel<false, 4>), grid, blk, 0, st, pp); break;
            case 2: if (Skv % 64) hipLaunchKernelGGL((attn64q_kernel<false, 0, 4, true, 0, true>), grid, blk, 0, st, pp);
                    else hipLaunchKernelGGL((attn64q_kernel<false>), grid, blk, 0, st, pp);
                    break;
            case 3: hipLaunchKernelGGL((attn64q_kernel<false, 0, 4, true, true>), grid, blk, 0, st, pp); break;
            case 4: hipLaunchKernelGGL((attn64q_kernel<false, 0, 6, true, true>), grid, blk, 0, st, pp); break;
            case 5: hipLaunchKernelGGL((attn64q_kernel<false, 1>), grid, blk, 0, st, pp); break;
            case 6: hipLaunchKernelGGL((attn64q_kernel<false, 16>), grid, blk, 0, st, pp); break;
            case 7: hipLaunchKernelGGL((attn64q_kernel<false, 2>), grid, blk, 0, st, pp); break;
            case 8: hipLaunchKernelGGL((attn64q_kernel<false, 6>), grid, blk, 0, st, pp); break;
            case 9: hipLaunchKernelGGL((attn64q_kernel<false, 22>), grid, blk, 0, st, pp); break;
            case 10: hipLaunchKernelGGL((attn64q_kernel<false, 7>), grid, blk, 0, st, pp); break;
            case 11: hipLaunchKernelGGL((attn64p_kernel<false, 4, 15>), grid, blk, 0, st, pp); break;
            case 12: hipLaunchKernelGGL((attn64p_kernel<false, 4, 128 + 14>), grid, blk, 0, st, pp); break;
            default: hipLaunchKernelGGL((attn64p_kernel<false, 4, 128>), grid, blk, 0, st, pp); break;
        }
    };
    const int NCHK = 5;
    for (int var = 0; var < NV; ++var) { launch(var); launch(var); }
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    // ---- outputs against attn64v2's ----
    std::vector<unsigned short> h0(nq), h1(nq);
    CK(hipMemcpy(h0.data(), o[0], nq * 2, hipMemcpyDeviceToHost));
    for (int var = 1; var < NCHK; ++var) {
        CK(hipMemcpy(h1.data(), o[var], nq * 2, hipMemcpyDeviceToHost));
        double se = 0, sr = 0, mx = 0;
        size_t bad = 0;
        for (size_t i = 0; i < nq; ++i) {
            const double a = bf2f(h0[i]), c = bf2f(h1[i]);
            if (!(c == c)) ++bad;
            se += (a - c) * (a - c); sr += a * a;
            mx = fmax(mx, fabs(a - c));
        }
        printf("%-40s vs attn64v2: rel-L2 %.3e  max-abs %.3e  NaN %zu\n", names[var], sqrt(se / sr), mx, bad);
        if (getenv("PROBE_DUMP") && var == 2) {   // where: per block of 32 queries x 32 columns of head 0, batch 0
            for (int qb = 0; qb < Sq / 32 && qb < 16; ++qb) {
                printf("  q %4d..:", qb * 32);
                for (int cb = 0; cb < 2; ++cb) {
                    double m2 = 0;
                    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
                        const size_t i = (size_t)(qb * 32 + r) * d + cb * 32 + c;
                        m2 = fmax(m2, fabs(bf2f(h0[i]) - bf2f(h1[i])));
                    }
                    printf(" %.2e", m2);
                }
                printf("\n");
            }
        }
    }
    // ---- a direct fp64 check of a few rows of (b, h) = (B-1, H-1) ----
    {
        const int b = B - 1, h = H - 1;
        double worst[NV] = {};
        std::vector<std::vector<unsigned short>> ho(NCHK, std::vector<unsigned short>(nq));
        for (int var = 0; var < NCHK; ++var) CK(hipMemcpy(ho[var].data(), o[var], nq * 2, hipMemcpyDeviceToHost));
        for (int qi : {0, 31, 32, 63, 64, 255, 256, Sq - 1}) {
            if (qi >= Sq) continue;
            std::vector<double> sc(Skv);
            double m = -1e300;
            for (int j = 0; j < Skv; ++j) {
                double a = 0;
                for (int c = 0; c < 64; ++c)
                    a += (double)bf2f(hq[((size_t)b * Sq + qi) * d + h * 64 + c]) * bf2f(hk[((size_t)b * Skv + j) * d + h * 64 + c]);
                sc[j] = a; m = fmax(m, a);
            }
            double l = 0;
            for (int j = 0; j < Skv; ++j) { sc[j] = exp2(sc[j] - m); l += sc[j]; }
            for (int c = 0; c < 64; ++c) {
                double a = 0;
                for (int j = 0; j < Skv; ++j) a += sc[j] * bf2f(hv[((size_t)b * Skv + j) * d + h * 64 + c]);
                a /= l;
                for (int var = 0; var < NCHK; ++var)
                    worst[var] = fmax(worst[var], fabs(a - bf2f(ho[var][((size_t)b * Sq + qi) * d + h * 64 + c])));
            }
        }
        for (int var = 0; var < NCHK; ++var) printf("%-40s vs fp64 (8 rows): max-abs %.3e\n", names[var], worst[var]);
    }
    // ---- timing: interleaved rounds ----
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> t(NV);
    for (int r = 0; r < rounds; ++r)
        for (int var = 0; var < NV; ++var) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) launch(var);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t[var].push_back(ms / iters * 1e3f);
        }
    const double fl = 4.0 * B * H * (double)Sq * Skv * 64;
    for (int var = 0; var < NV; ++var) {
        std::sort(t[var].begin(), t[var].end());
        const float med = t[var][t[var].size() / 2];
        printf("%-40s %8.1f us (min %.1f)  %7.1f TFLOP/s  %.1f %% of 2.5 PF\n", names[var], med, t[var][0], fl / med / 1e6, fl / med / 1e6 / 25.0);
    }
    return 0;
}
