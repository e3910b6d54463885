// split_bf16.h -- exact fp32 -> 3 x bf16 splitting for the "bf16x3" matrix-core kernels
// (conv_split.hip, edgeconv_split.hip): x = h + m + l with h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m), round-to-nearest-even, every remainder exact in fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b)
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));     // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float bf16_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// two fp32 -> packed (h, m, l) bf16 pairs, x = h + m + l exactly
__device__ __forceinline__ void split_pair(float a, float b, uint32_t &h, uint32_t &m, uint32_t &l)
{
    // the two remainders are formed as packed fp32 subtractions (v_pk_add_f32: one instruction per pair)
    const f32x2 x = {a, b};
    h = cvt_pk_bf16(a, b);
    const f32x2 hf = {bf16_lo(h), bf16_hi(h)};
    const f32x2 r = x - hf;
    m = cvt_pk_bf16(r[0], r[1]);
    const f32x2 mf = {bf16_lo(m), bf16_hi(m)};
    const f32x2 r2 = r - mf;
    l = cvt_pk_bf16(r2[0], r2[1]);
}

// 8 fp32 -> three uint4 of 8 bf16
__device__ __forceinline__ void split8(const float (&v)[8], uint4 &h, uint4 &m, uint4 &l)
{
    split_pair(v[0], v[1], h.x, m.x, l.x);
    split_pair(v[2], v[3], h.y, m.y, l.y);
    split_pair(v[4], v[5], h.z, m.z, l.z);
    split_pair(v[6], v[7], h.w, m.w, l.w);
}

