// common.h — shared device helpers for the gfx950 (MI355X, CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ditto {

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

DITTO_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DITTO_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

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

// Bijective XCD-aware block remap (guide §5 "XCD swizzle must be bijective"): blocks b and b+8
// share an XCD/L2, so give each XCD a contiguous chunk of the logical tile space.
DITTO_DEV int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}

}  // namespace ditto
