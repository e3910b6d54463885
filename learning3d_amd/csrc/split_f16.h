// split_f16.h -- the two operand splits of the "f16x2" matrix-core arithmetic (edgeconv_f16.hip has the derivation):
//   weight-like     W = w c:  H = f16(W),  Hs = f16(H 2^-12),  M = f16(W - H)          (af_split_w)
//   activation-like X = x c:  h = f16(X),  m' = f16((X - h) 2^12)                      (af_split_x)
// two values per call, packed fp16 pairs out; c is a power of two.  VOP3P mix instructions written as inline asm: the
// compiler's hazard recogniser does not see inside, so a result that an MFMA reads DIRECTLY (not through LDS) needs
// >= 4 wait states in between (attention_f16.hip pads with s_nop).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// two fp32 -> packed fp16 (H, Hs, M) of w c, c = 2^S
__device__ __forceinline__ void af_split_w(float a0, float a1, float c, uint32_t &H, uint32_t &Hs, uint32_t &M)
{
    float r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(H) : "v"(a0), "v"(c));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(H) : "v"(a1), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a0), "v"(c), "v"(H));                     // w c - H: exact
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(a1), "v"(c), "v"(H));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, 0" : "=v"(M) : "v"(r0));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, 0" : "+v"(M) : "v"(r1));
    asm("v_pk_mul_f16 %0, %1, %2" : "=v"(Hs) : "v"(H), "s"(0x0C000C00u));                                            // H 2^-12 (packed fp16 2^-12)
}
// two fp32 -> packed fp16 (h, m') of x c, c = 2^T
__device__ __forceinline__ void af_split_x(float a0, float a1, float c, uint32_t &h, uint32_t &m)
{
    float r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a0), "v"(c));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a0), "v"(c), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(a1), "v"(c), "v"(h));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(m) : "v"(r0), "s"(4096.0f));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(m) : "v"(r1), "s"(4096.0f));
}

// two fp32 -> packed fp16 (h, m) of x c with the UNSCALED residual m = f16(x c - h): for operands placed near 2^11 (a subnormal
// residual then costs 2^-25 absolute in plane units, edgeconv_f16b.hip "EF_V2") -- the Hs weight plane is not needed
__device__ __forceinline__ void af_split_x_unscaled(float a0, float a1, float c, uint32_t &h, uint32_t &m)
{
    float r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a0), "v"(c));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a0), "v"(c), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(a1), "v"(c), "v"(h));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, 0" : "=v"(m) : "v"(r0));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, 0" : "+v"(m) : "v"(r1));
}
// the same for operands that are ALREADY in plane units (x = a, no scale): v_cvt_pk_f16_f32 (gfx950) rounds the pair, the rounded
// halves are subtracted back (exact) and rounded again -- six full-rate VALU instructions where the v_fma_mix* forms above cost
// about 2.4x a plain VALU instruction each beside an MFMA stream (LABLOG R2.2; edgeconv_f16b.hip's split).
// s_nop: VALU write -> SDWA read of the same VGPR, a hazard the compiler does not see inside asm.
__device__ __forceinline__ void af_split_x_unscaled_cvt(float a0, float a1, uint32_t &h, uint32_t &m)
{
    float f0, f1, r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a0), "v"(a1));
    asm("v_cvt_f32_f16_e32 %0, %1" : "=v"(f0) : "v"(h));
    asm("s_nop 0\n\tv_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f1) : "v"(h));
    asm("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(a0), "v"(f0));
    asm("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(a1), "v"(f1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(m) : "v"(r0), "v"(r1));
}
