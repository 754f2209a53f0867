// fp32 products on the bf16 matrix pipe: three-way operand split (shared by conv3x3.hip, conv1x1.hip,
// conv_wgrad_split.hip).
//
//   a = hi + mid + lo,   hi = bf16_rn(a), mid = bf16_rn(a - hi), lo = bf16_rn(a - hi - mid)
//   (|mid| <= 2^-9 |a|, |lo| <= 2^-18 |a|, residual <= 2^-27 |a|)
//   a*b ~= hi.hi + hi.mid + mid.hi + hi.lo + mid.mid + lo.hi      (six v_mfma_f32_32x32x16_bf16 terms)
// Every term is exact in the fp32 accumulator; what is dropped (mid.lo, lo.mid, lo.lo, the residuals) is
// <= 2^-25 |a b|, below one fp32 rounding of the product.  Accumulation is fp32, as in the fp32 MFMA.
// Operand map of v_mfma_f32_32x32x16_bf16: lane (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j] /
// B[k = 8h + j][col r] in element j = 0..7 of its 16-byte fragment.
#pragma once
#include "gemm_engine.h"

namespace scat {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {   // round-to-nearest-even, element 0 in the low half
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2_t));
}
// Scalar subtractions on purpose: beside a busy matrix pipe a packed fp32 instruction (v_pk_add_f32) costs ~17 issue
// cycles against 4 for a scalar one (MI355X_MICROARCH.md, "price of one filler beside MFMAs": "an anti-lever ... including
// when the compiler SLP-packs adjacent scalar f32 adds") — the library is built with -fno-slp-vectorize for the same reason.
__device__ __forceinline__ void split3(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = pk_bf16(a, b);
    a -= __uint_as_float(hi << 16);
    b -= __uint_as_float(hi & 0xffff0000u);
    mid = pk_bf16(a, b);
    a -= __uint_as_float(mid << 16);
    b -= __uint_as_float(mid & 0xffff0000u);
    lo = pk_bf16(a, b);
}
// 8 floats -> the three 16-byte fragments
__device__ __forceinline__ void split3x8(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t h, m, l;
        split3(x[2 * q], x[2 * q + 1], h, m, l);
        hi[q] = h; mid[q] = m; lo[q] = l;
    }
}
// (static_for<N>: gemm_engine.h)

// acc += a * b with the six significant terms, small ones first
__device__ __forceinline__ f32x16 mfma_split(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0]), am = __builtin_bit_cast(bf16x8, a[1]),
                 al = __builtin_bit_cast(bf16x8, a[2]);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b[0]), bm = __builtin_bit_cast(bf16x8, b[1]),
                 bl = __builtin_bit_cast(bf16x8, b[2]);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    return c;
}

}  // namespace scat
