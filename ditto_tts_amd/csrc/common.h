// common.h — shared device helpers for the gfx950 (MI355X, CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <initializer_list>

namespace ditto {

// Raise the dynamic-LDS limit of kernels ONCE PER DEVICE.  hipFuncAttributeMaxDynamicSharedMemorySize is a property of the
// (function, device) pair: a process that runs the model on a second GPU must set it there too, or the first launch of a
// kernel that asks for more than 64 KiB fails on that device.  One DevOnce per call site; bit d = done on device d.  Two
// threads racing on the first call both set the (same) value: benign.  Devices >= 64 set it on every call.
struct DevOnce { std::atomic<unsigned long long> done{0}; };
inline hipError_t set_max_lds_once(DevOnce& once, std::initializer_list<const void*> kernels, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((once.done.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
    for (const void* k : kernels) {
        e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
    }
    if (dev >= 0 && dev < 64) once.done.fetch_or(1ull << dev, std::memory_order_release);
    return hipSuccess;
}

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define DITTO_DEV __device__ __forceinline__

constexpr int WAVE = 64;

// Wave-wide sum / maximum, the total in every lane.  Round 5: on the cross-lane paths of the VECTOR pipe (DPP quad permutes and row
// mirrors, two row broadcasts, one v_readlane) instead of six ds_bpermute round trips through the LDS: the fused norm2 + q-projection
// kernel normalises its 64 rows at ONE wave per SIMD, where each of its 32 reductions per wave was a chain of six ~120-cycle LDS
// latencies — 42 % of the kernel (profiles/r05_lnq_stamps.txt).  Association: the near-to-far butterfly x_i + x_{i^1}, then ^2, ^4, ^8,
// ^16, ^32 — after the quad steps all lanes of a quad hold the same value, so the half-row mirror pairs the same VALUES as lane i ^ 4
// would (likewise the row mirror for i ^ 8), and (R1 + R0), then (R3 + R2) + (R1 + R0) across the four rows is the butterfly's
// (R0 + R1) + (R2 + R3) by commutativity: every kernel that normalises rows through these helpers (ln_kernel, the AdaLN kernel's
// norm1, the split-K finish, gemm_lnq) still produces the SAME bits as the others.
// PRECONDITION: whole 64-lane waves (EXEC all ones) — the total is read from lane 63 and the row broadcasts read lanes 15 / 31 / 47,
// so an inactive lane there gives garbage (the ds_bpermute butterflies of rounds 1-4 summed the active lanes only).  Every call
// site runs 256-thread workgroups with wave-uniform control flow around it.  The association is near-to-far, so LayerNorm bits
// differ from ABI-8 libraries ("bit changes between library versions").
template <int CTRL, int ROW_MASK = 0xF>
DITTO_DEV float dpp_f32(float v, float masked = 0.0f) {   // lanes of masked-off rows read `masked`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, masked), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, 0xF, false));
}
DITTO_DEV float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);            // quad_perm [1,0,3,2]: lane i ^ 1
    v += dpp_f32<0x4E>(v);            // quad_perm [2,3,0,1]: lane i ^ 2
    v += dpp_f32<0x141>(v);           // row_half_mirror: the other quad of the 8-lane group
    v += dpp_f32<0x140>(v);           // row_mirror: the other half of the 16-lane row
    v += dpp_f32<0x142, 0xA>(v);      // row_bcast:15 into rows 1 and 3: R1 + R0, R3 + R2
    v += dpp_f32<0x143, 0xC>(v);      // row_bcast:31 into rows 2 and 3: lane 63 = (R3 + R2) + (R1 + R0)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
DITTO_DEV float wave_max(float v) {
    const float ninf = -__builtin_huge_valf();
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    v = fmaxf(v, dpp_f32<0x142, 0xA>(v, ninf));
    v = fmaxf(v, dpp_f32<0x143, 0xC>(v, ninf));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// LayerNorm row arithmetic shared by ln_kernel (rowwise.hip) and the fused norm2 + q-projection kernel (gemm_lnq.hip), with
// every fused multiply-add written out: left to -ffp-contract=fast the SAME source expression contracts differently in
// different surroundings (contraction follows basic-block structure), and the two kernels must give the same bits.
DITTO_DEV float ln_sum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
DITTO_DEV float ln_sq_acc(float q, float v, float mean) {
    const float dlt = v - mean;
    return __builtin_fmaf(dlt, dlt, q);
}
DITTO_DEV float ln_norm(float v, float mean, float rstd, float g, float b) { return __builtin_fmaf((v - mean) * rstd, g, b); }

// two fp32 -> packed bf16x2 in one dword (hipcc emits v_cvt_pk_bf16_f32; RNE, NaN-preserving)
DITTO_DEV unsigned pack_bf16x2(float lo, float hi) {
    bf16x2 p;
    p[0] = (bf16)lo;
    p[1] = (bf16)hi;
    return __builtin_bit_cast(unsigned, p);
}
DITTO_DEV float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
DITTO_DEV float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }

DITTO_DEV float silu_f(float x) { return x / (1.0f + __expf(-x)); }
DITTO_DEV float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
// nn.GELU() default = exact erf form (reference src/components/DiT.py:96)
DITTO_DEV float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Fast forms for the fused GEMM epilogue, where 64 evaluations per lane sit on the critical path of a tile.
// erf by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16 rounding of the value stored),
// 1/x by v_rcp_f32, e^x by v_exp_f32 (both ~1 ulp).  ~22 VALU per gelu*sigmoid instead of ~56 with erff/expf
// and IEEE division.
DITTO_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
DITTO_DEV float fast_sigmoid(float x) {
    return fast_rcp(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
DITTO_DEV float fast_gelu_erf(float x) {
    const float z = x * 0.70710678118654752440f, az = fabsf(z);
    const float t = fast_rcp(fmaf(0.3275911f, az, 1.0f));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * az * az);
    const float erf_abs = fmaf(-poly, e, 1.0f);             // erf(|z|)
    return 0.5f * x * (1.0f + copysignf(erf_abs, z));
}

// Two (fc1, gate) pairs at once on 2-element vectors: the multiplies / FMAs of the polynomial compile to v_pk_*_f32
// (two fp32 results per instruction; the gated-MLP epilogue is VALU-bound: 4.6 us of a 24 us tile, measured in-model
// with the no-epilogue diagnostic), the four transcendentals per pair stay scalar.  Same formulas as the scalar forms.
typedef __attribute__((ext_vector_type(2))) float f32x2;
// The same formulas with the scalings folded and the sign handled by |x| source modifiers (23 VALU per pair instead of
// 29; the gated epilogue is issue-bound at 1 750 instructions per wave per tile):
//     gelu(x) sigmoid(g) = (x + |x| erf(|x|/sqrt2)) * 1 / (2 + 2 e^-g)
//     erf(|z|) = 1 - poly(t) e^(-x^2/2),   t = 1 / (1 + (p/sqrt2) |x|)          (A&S 7.1.26, as fast_gelu_erf)
// Third form (default): erf as a RATIONAL function, z P(z^2) / Q(z^2) on |z| <= 4 (P of degree 6, Q of degree 4 with
// Q >= 1; least-squares + Lawson-reweighted fit against scipy's erf: 5.4e-8 in fp64, 4.3e-7 evaluated in fp32 — the level
// of the A&S form above), so that the two divisions of the product share ONE reciprocal and the Gaussian factor's
// exponential disappears:
//     gelu(x) sigmoid(g) = (x/2) (Q + z P) / (Q (1 + e^-g)),    z = clamp(x / sqrt2, -4, 4)
// 2 transcendentals per output instead of 4: the gated epilogue runs with the matrix pipe idle and its transcendentals
// (16 cycles each, quarter rate) were 62 % of its arithmetic time.  Q (1 + e^-g) = inf for
// g < -88 gives 0, the limit; beyond |z| = 4 erf is +-1 to 1.5e-8.  The constants are pinned by tests/test_gated_math.py.
// The NEGATIVE tail: at z = -4 the factor Q + z P is not 0 but the fit's residual (+-4e-7 Q), so with the plain x / 2
// in front the result grew like x * 2e-7 for x < -5.66 instead of decaying.  x / 2 is therefore taken from the LOWER-clamped z:
// x / 2 = (x / sqrt2) / sqrt2, and zl = max(x / sqrt2, -4) bounds the tail by 2.8 * 4e-7 (pinned by the CPU test's sweep to
// |x| = 1e4); for x > -5.66 it is x / 2 to 1 ulp.
DITTO_DEV f32x2 fast_gelu_sigmoid2(f32x2 x, f32x2 g) {
    f32x2 zl = x * 0.70710678118654752440f, z;
    // (v_med3_f32 rather than fmaxf / fminf: no NaN-canonicalising v_max x, x in front)
    zl[0] = __builtin_amdgcn_fmed3f(zl[0], -4.0f, 3.0e38f); zl[1] = __builtin_amdgcn_fmed3f(zl[1], -4.0f, 3.0e38f);
    z[0] = __builtin_amdgcn_fmed3f(zl[0], -4.0f, 4.0f); z[1] = __builtin_amdgcn_fmed3f(zl[1], -4.0f, 4.0f);
    const f32x2 u = z * z;
    f32x2 pn = u * 1.9217200275534196e-08f + (-1.990321152334218e-06f);
    pn = pn * u + 0.00015553680714219809f;
    pn = pn * u + 0.0042930529452860355f;
    pn = pn * u + 0.05243346840143204f;
    pn = pn * u + 0.2139447033405304f;
    pn = pn * u + 1.1283786296844482f;
    f32x2 qd = u * 0.0010980380466207862f + 0.01555109117180109f;
    qd = qd * u + 0.12079962342977524f;
    qd = qd * u + 0.5229312181472778f;
    qd = qd * u + 1.0f;
    const f32x2 num = (zl * 0.70710678118654752440f) * (z * pn + qd);   // (x / 2) Q (1 + erf): x Q Phi(x)
    const f32x2 eg = g * (-1.4426950408889634f);
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(eg[0]); e[1] = __builtin_amdgcn_exp2f(eg[1]);
    const f32x2 den = qd * e + qd;                            // Q (1 + e^-g)
    f32x2 r;
    r[0] = fast_rcp(den[0]); r[1] = fast_rcp(den[1]);
    return num * r;
}
// Backward of the same product for two elements (train.hip gated_bwd_elem, whose formulas these are, on 2-vectors):
//   y = gelu(a) sigmoid(g):  da = dy sigmoid(g) (Phi(a) + a phi(a)),  dg = dy gelu(a) sigmoid(g) (1 - sigmoid(g)),
// Phi by the rational erf above (one reciprocal), phi one v_exp, sigmoid one v_exp + one v_rcp.  The epilogue of the fc2
// dgrad GEMM in the training backward (gemm_common.h epilogue_gated_bwd).
DITTO_DEV void gated_bwd2(f32x2 a, f32x2 g, f32x2 dy, f32x2& da, f32x2& dg) {
    f32x2 z = a * 0.70710678118654752440f;
    z[0] = __builtin_amdgcn_fmed3f(z[0], -4.0f, 4.0f); z[1] = __builtin_amdgcn_fmed3f(z[1], -4.0f, 4.0f);
    const f32x2 u = z * z;
    f32x2 pn = u * 1.9217200275534196e-08f + (-1.990321152334218e-06f);
    pn = pn * u + 0.00015553680714219809f;
    pn = pn * u + 0.0042930529452860355f;
    pn = pn * u + 0.05243346840143204f;
    pn = pn * u + 0.2139447033405304f;
    pn = pn * u + 1.1283786296844482f;
    f32x2 qd = u * 0.0010980380466207862f + 0.01555109117180109f;
    qd = qd * u + 0.12079962342977524f;
    qd = qd * u + 0.5229312181472778f;
    qd = qd * u + 1.0f;
    const f32x2 ea = (a * a) * (-0.72134752044448170368f), eg = g * (-1.4426950408889634f);
    f32x2 rq, pe, sg;
    rq[0] = fast_rcp(qd[0]); rq[1] = fast_rcp(qd[1]);
    pe[0] = __builtin_amdgcn_exp2f(ea[0]); pe[1] = __builtin_amdgcn_exp2f(ea[1]);
    sg[0] = fast_rcp(1.0f + __builtin_amdgcn_exp2f(eg[0])); sg[1] = fast_rcp(1.0f + __builtin_amdgcn_exp2f(eg[1]));
    const f32x2 cdf = (z * pn + qd) * (rq * 0.5f);                 // Phi(a)
    const f32x2 pdf = pe * 0.39894228040143267794f;                // phi(a)
    const f32x2 t = dy * sg;
    da = t * (a * pdf + cdf);
    dg = (t * (a * cdf)) * (1.0f - sg);
}
// Fourth form (A/B build -DDITTO_GATED_H): the whole activation in PACKED fp16, in the sigmoid form of the
// Gaussian cdf,
//     gelu(x) sigmoid(g) ~= x / ((1 + e^-p(x)) (1 + e^-g)),   p(x) = 2 sqrt(2/pi) x (1 + 0.044715 x^2)   (the "tanh" GELU),
// 7 v_pk_*_f16 (one pass each, where v_pk_*_f32 takes two) + 4 v_exp_f16 + 2 v_rcp_f16 per PAIR of outputs, against 18 packed
// fp32 operations + 4 transcendentals.  Accuracy (tests/test_gated_math.py; x, g ~ N(0, 1.5)): the form itself is 1.8e-4 rel-L2 from the
// exact erf product (max 4.7e-4 absolute), the fp16 evaluation 4.9e-4 — against the 1.66e-3 of rounding the EXACT value to bf16,
// which is what the epilogue does next: 1.73e-3 after rounding, +4 %.  Limits: e^-p overflows to +inf for x < -4.2 and the
// reciprocal returns 0 (gelu -> 0); for x > 4 e^-p underflows and the product is x sigmoid(g); g < -11: 0; no 0 * inf arises
// ((1 + e) >= 1).  x is kept in fp32 for the final multiply.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
DITTO_DEV f32x2 fast_gelu_sigmoid2_h(f32x2 x, f32x2 g) {
    const h16x2 xh = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(x[0], x[1]));
    const h16x2 gh = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(g[0], g[1]));
    // exponents in log2 units: -log2(e) * 2 sqrt(2/pi) (1 + 0.044715 x^2) x   and   -log2(e) g
    const h16x2 inner = (xh * xh) * (_Float16)(-0.10294324f) + (_Float16)(-2.3022082f);
    const h16x2 ea = xh * inner, eg = gh * (_Float16)(-1.4426950f);
    h16x2 e1, e2;
    e1[0] = __builtin_exp2f16(ea[0]); e1[1] = __builtin_exp2f16(ea[1]);
    e2[0] = __builtin_exp2f16(eg[0]); e2[1] = __builtin_exp2f16(eg[1]);
    const h16x2 den = (e1 + (_Float16)1.0f) * (e2 + (_Float16)1.0f);
    h16x2 r;
    r[0] = __builtin_amdgcn_rcph(den[0]); r[1] = __builtin_amdgcn_rcph(den[1]);
    return x * f32x2{(float)r[0], (float)r[1]};
}
// second packed form (A/B: -DDITTO_GATED_AS): A&S erf, scalings folded, sign by |x| source modifiers
DITTO_DEV f32x2 fast_gelu_sigmoid2_as(f32x2 x, f32x2 g) {
    f32x2 t;
    t[0] = fast_rcp(fmaf(fabsf(x[0]), 0.3275911f * 0.70710678118654752440f, 1.0f));
    t[1] = fast_rcp(fmaf(fabsf(x[1]), 0.3275911f * 0.70710678118654752440f, 1.0f));
    f32x2 poly = t * 1.061405429f + (-1.453152027f);
    poly = poly * t + 1.421413741f;
    poly = poly * t + (-0.284496736f);
    poly = poly * t + 0.254829592f;
    poly = poly * t;
    const f32x2 ea = x * x * (-0.5f * 1.4426950408889634f);
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(ea[0]); e[1] = __builtin_amdgcn_exp2f(ea[1]);
    const f32x2 erf_abs = 1.0f - poly * e;
    f32x2 num;                                               // x + |x| erf(|z|) = 2 gelu(x)
    num[0] = fmaf(fabsf(x[0]), erf_abs[0], x[0]); num[1] = fmaf(fabsf(x[1]), erf_abs[1], x[1]);
    const f32x2 eg = g * (-1.4426950408889634f);
    f32x2 eg2;
    eg2[0] = __builtin_amdgcn_exp2f(eg[0]); eg2[1] = __builtin_amdgcn_exp2f(eg[1]);
    const f32x2 den2 = eg2 * 2.0f + 2.0f;
    f32x2 sg;                                                // sigmoid(g) / 2
    sg[0] = fast_rcp(den2[0]); sg[1] = fast_rcp(den2[1]);
    return num * sg;
}
// first packed form (kept for reference / A/B): separate scaling multiplies, sign by v_bfi
DITTO_DEV f32x2 fast_gelu_sigmoid2_v1(f32x2 x, f32x2 g) {
    const f32x2 z = x * 0.70710678118654752440f;
    const f32x2 az = __builtin_elementwise_abs(z);
    const f32x2 d1 = az * 0.3275911f + 1.0f;
    f32x2 t;
    t[0] = fast_rcp(d1[0]); t[1] = fast_rcp(d1[1]);
    f32x2 poly = t * 1.061405429f + (-1.453152027f);
    poly = poly * t + 1.421413741f;
    poly = poly * t + (-0.284496736f);
    poly = poly * t + 0.254829592f;
    poly = poly * t;
    const f32x2 ea = az * az * (-1.4426950408889634f);
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(ea[0]); e[1] = __builtin_amdgcn_exp2f(ea[1]);
    const f32x2 erf_abs = 1.0f - poly * e;
    f32x2 erf_s;
    erf_s[0] = copysignf(erf_abs[0], z[0]); erf_s[1] = copysignf(erf_abs[1], z[1]);
    const f32x2 gelu = x * 0.5f * (erf_s + 1.0f);
    const f32x2 eg = g * (-1.4426950408889634f);
    f32x2 den;
    den[0] = 1.0f + __builtin_amdgcn_exp2f(eg[0]); den[1] = 1.0f + __builtin_amdgcn_exp2f(eg[1]);
    f32x2 sg;
    sg[0] = fast_rcp(den[0]); sg[1] = fast_rcp(den[1]);
    return gelu * sg;
}

// four fp32 -> packed OCP fp8 e4m3 in one dword (byte 0 = first), saturating at +-448 (v_cvt_pk_fp8_f32 does not
// clamp: an overflow would become NaN in e4m3fn)
DITTO_DEV unsigned pack_fp8x4(float a, float b, float c, float d) {
    const float L = 448.0f;
    a = fminf(fmaxf(a, -L), L); b = fminf(fmaxf(b, -L), L); c = fminf(fmaxf(c, -L), L); d = fminf(fmaxf(d, -L), L);
    int pk = 0;
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, pk, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, pk, true);
    return (unsigned)pk;
}

// Counter-based dropout mask of the training path (train.hip header): keep iff hash(stream, query, key) >= thr.
// The per-(batch, head) stream is TWO full 32-bit mixes (lowbias32; once per kernel launch and wave); the PER-ELEMENT hash
// runs on full-rate instructions only:
//     x = a + (query * C1 + key * C2);  x ^= x >> 13;  x += b;  x ^= x >> 9;  h = (x & 0xFFFFFF) * C
// (v_add, v_lshrrev, v_xad, v_lshrrev, v_xor, v_mul_u32_u24: v_mul_lo_u32 runs at a quarter of their rate, and the mask is
// evaluated in the element loops of the forward and of both backward kernels).  Both stream words enter by ADDs, i.e.
// NON-linearly over GF(2), in front of an xor-fold each.  Round 3 xored ONE word in front of ONE fold:
// fold(a ^ ctr) = fold(a) ^ fold(ctr), so every stream saw the same 24-bit image of (query, key), and the 3.6 % of a
// 1024 x 1024 tile that collide there decided identically in EVERY stream (cross-stream mask correlation rms 0.023, max 0.11).
// One add + one fold still leaves rare stream pairs correlated at 0.01-0.015 (3 of 2016 pairs); with the second add + fold:
// 2016 pairs of 1024 x 1024 masks at p = 0.1 have correlation rms 0.97 sigma, max 3.3 sigma (sigma = 1 / 1024: what
// lowbias32 per element gives), positions that collide in one stream agree in the others at the independent rate 0.82, lag
// correlations along query / key / diagonal within 2.7 sigma (tests/test_oracle_dropout_hash.py).
// oracle/ditto_oracle.py hash_dropout_mask is the same function.
DITTO_DEV unsigned lowbias32(unsigned h) {
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
struct DropStream { unsigned a, b; };
DITTO_DEV DropStream drop_stream(unsigned seed_lo, unsigned seed_hi, int layer, int bh) {
    const unsigned a = lowbias32(seed_lo ^ lowbias32(seed_hi + (unsigned)layer * 0x632BE5ABu + (unsigned)bh * 0x9E3779B1u));
    return DropStream{a, lowbias32(a ^ 0x5BD1E995u)};
}
DITTO_DEV bool drop_keep(DropStream st, int i, int j, unsigned thr) {
    unsigned x = st.a + ((unsigned)i * 0x9E3779B1u + (unsigned)j * 0x85EBCA6Bu);
    x = (x ^ (x >> 13)) + st.b;                          // v_xad_u32
    x ^= x >> 9;
    return (x & 0xFFFFFFu) * 0xD2B74Fu >= thr;           // v_mul_u32_u24 (the mask is free: the instruction reads 24 bits)
}

// Bijective XCD-aware block remap (guide §5 "XCD swizzle must be bijective"): blocks b and b+8
// share an XCD/L2, so give each XCD a contiguous chunk of the logical tile space.
DITTO_DEV int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}

}  // namespace ditto
