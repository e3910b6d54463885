// Cost of VALU "filler" instructions issued between v_mfma_f32_16x16x32_f16 (one wave per SIMD, 5 independent
// accumulators in VGPRs, B operand in AGPRs): ticks per MFMA with NF fillers of one kind after every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FILL(KIND, i)                                                                                              \
    do {                                                                                                           \
        if (KIND == 0) asm volatile("v_max_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));              \
        if (KIND == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]), "v"(f[(i + 3) & 7])); \
        if (KIND == 2) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));     \
        if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));       \
        if (KIND == 4) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f[i]) : "v"(f[(i + 1) & 7])); \
        if (KIND == 5) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));                                \
        if (KIND == 6) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(g[i]) : "v"(f[(i + 1) & 7]));                              \
        if (KIND == 7) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(f[i]) : "a"(g[(i + 1) & 7]));                               \
        if (KIND == 8) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]), "v"(f[(i + 3) & 7])); \
        if (KIND == 9) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]), "v"(f[(i + 3) & 7])); \
        if (KIND == 10) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));     \
        if (KIND == 11) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7])); \
        if (KIND == 12) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d[i & 3]) : "v"(d[(i + 1) & 3]), "v"(d[(i + 2) & 3]));     \
    } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
// one memory / LDS instruction per 5 MFMAs: 0 global_load_dwordx4, 1 ds_read_b128, 2 global_load_dwordx4 into AGPRs, 3 none
template <int MK>
__global__ __launch_bounds__(256, 1) void km(float *out, unsigned long long *cyc, int iters, const float4 *src)
{
    __shared__ float4 lds[1024];
    f32x4 acc[5];
    u32x4 a = {threadIdx.x, 1, 2, 3}, b[5];
    lds[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
    for (int i = 0; i < 5; i++) { acc[i] = (f32x4){0, 0, 0, 0}; b[i] = (u32x4){threadIdx.x + i, 5, 6, 7}; asm volatile("" : "+a"(b[i])); }
    float4 sum = {0, 0, 0, 0};
    const float4 *p = src + threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 6; rep++) {
            f32x4 v = {0, 0, 0, 0};
            if (MK == 0 || MK == 2) v = *(const f32x4 *)&p[((it * 6 + rep) & 63) * 256];
            if (MK == 1) v = *(const f32x4 *)&lds[(threadIdx.x + rep * 64) & 1023];
            if (MK == 2) asm volatile("" : "+a"(v));
#pragma unroll
            for (int i = 0; i < 5; i++) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "a"(b[i]));
            if (MK == 2) asm volatile("" : "+v"(v));
            sum.x += v[0]; sum.y += v[1]; sum.z += v[2]; sum.w += v[3];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = sum.x + sum.y + sum.z + sum.w;
    for (int i = 0; i < 5; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MK> double runm()
{
    static float *out = nullptr; static unsigned long long *cyc = nullptr; static float4 *src = nullptr; const int nb = 256, iters = 500;
    if (!out) { hipMalloc(&out, 4 * nb * 256); hipMalloc(&cyc, 8 * nb); hipMalloc(&src, 16 * 256 * 64 + 4096); hipMemset(src, 0, 16 * 256 * 64 + 4096); }
    hipLaunchKernelGGL((km<MK>), dim3(nb), dim3(256), 0, 0, out, cyc, iters, src);
    hipLaunchKernelGGL((km<MK>), dim3(nb), dim3(256), 0, 0, out, cyc, iters, src);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, cyc, 8 * nb, hipMemcpyDeviceToHost);
    return h[7] / ((double)iters * 30);
}

template <int KIND, int NF>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[5];
    u32x4 a = {threadIdx.x, 1, 2, 3}, b[5];
    float f[8]; float g[8]; f32x2 d[4];
    for (int i = 0; i < 8; i++) { f[i] = threadIdx.x * 0.001f + i; g[i] = f[i]; asm volatile("" : "+a"(g[i])); }
    for (int i = 0; i < 4; i++) d[i] = (f32x2){f[i], f[i + 4]};
    for (int i = 0; i < 5; i++) { acc[i] = (f32x4){0, 0, 0, 0}; b[i] = (u32x4){threadIdx.x + i, 5, 6, 7}; asm volatile("" : "+a"(b[i])); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 6; rep++)
#pragma unroll
            for (int i = 0; i < 5; i++) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "a"(b[i]));
#pragma unroll
                for (int q = 0; q < NF; q++) FILL(KIND, ((rep * 5 + i) * NF + q) & 7);
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 5; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) { float gv = g[i]; asm volatile("" : "+v"(gv)); s += f[i] + gv; }
    for (int i = 0; i < 4; i++) s += d[i][0] + d[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND, int NF> double run()
{
    static float *out = nullptr; static unsigned long long *cyc = nullptr; const int nb = 256, iters = 500;
    if (!out) { hipMalloc(&out, 4 * nb * 256); hipMalloc(&cyc, 8 * nb); }
    hipLaunchKernelGGL((k<KIND, NF>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<KIND, NF>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, cyc, 8 * nb, hipMemcpyDeviceToHost);
    return h[7] / ((double)iters * 30);
}
template <int KIND> void row(const char *name)
{
    printf("%-24s  1: %6.2f   2: %6.2f   3: %6.2f   4: %6.2f   6: %6.2f  ticks per MFMA\n", name, run<KIND, 1>(), run<KIND, 2>(), run<KIND, 3>(), run<KIND, 4>(), run<KIND, 6>());
}
int main()
{
    printf("fillers per MFMA:\n");
    row<0>("v_max_f32"); row<8>("v_fma_f32"); row<9>("v_max3_f32"); row<10>("v_cndmask_b32");
    row<1>("v_fma_mix_f32"); row<2>("v_fma_mixlo_f16"); row<3>("v_cvt_pk_f16_f32");
    row<4>("v_cvt_f32_f16_sdwa"); row<5>("v_cvt_f32_f16"); row<6>("v_accvgpr_write"); row<7>("v_accvgpr_read");
    row<11>("s_nop1+v_max_f32_dpp"); row<12>("v_pk_mul_f32");
    printf("one memory instruction per 5 MFMAs (ticks per MFMA): none %.2f  global_load_dwordx4 %.2f  -> AGPR %.2f  ds_read_b128 %.2f\n",
           runm<3>(), runm<0>(), runm<2>(), runm<1>());
    return 0;
}
