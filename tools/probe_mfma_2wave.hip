// Does a SECOND wave on the SIMD hide VALU work beside v_mfma_f32_16x16x32_f16?  probe_mfma_filler.hip measured one wave per
// SIMD: one VALU per MFMA is free, every further one costs 2-7.5 cycles.  Here: W waves per SIMD (256 x W threads, one
// workgroup per CU), every wave issues {MFMA, NF fillers} x 30 per iteration; reported = ticks per MFMA *per SIMD*
// (wave time / (MFMAs per wave x W)).  MODE 1: specialised waves -- waves 0-3 issue only MFMAs, waves 4-7 only the fillers
// (the same number the MFMA waves would have carried); both times reported.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/p2w tools/probe_mfma_2wave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FILL(KIND, i)                                                                                              \
    do {                                                                                                           \
        if (KIND == 0) asm volatile("v_max_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));              \
        if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));       \
        if (KIND == 4) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f[i]) : "v"(f[(i + 1) & 7])); \
        if (KIND == 6) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(g[i]) : "v"(f[(i + 1) & 7]));                              \
        if (KIND == 11) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7])); \
    } while (0)

template <int W, int KIND, int NF, int MODE, int PRIO>
__global__ __launch_bounds__(256 * W, 1) void k(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[5];
    u32x4 a = {threadIdx.x, 1, 2, 3}, b[5];
    float f[8]; float g[8];
    for (int i = 0; i < 8; i++) { f[i] = threadIdx.x * 0.001f + i; g[i] = f[i]; asm volatile("" : "+a"(g[i])); }
    for (int i = 0; i < 5; i++) { acc[i] = (f32x4){0, 0, 0, 0}; b[i] = (u32x4){threadIdx.x + i, 5, 6, 7}; asm volatile("" : "+a"(b[i])); }
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || wave < 4, do_fill = MODE == 0 || wave >= 4;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int rep = 0; rep < 6; rep++)
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    if (PRIO) asm volatile("s_setprio 1");
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "a"(b[i]));
                    if (PRIO) asm volatile("s_setprio 0");
#pragma unroll
                    for (int q = 0; q < NF; q++) FILL(KIND, ((rep * 5 + i) * NF + q) & 7);
                }
        }
    } else if (do_mfma) {
        if (PRIO) asm volatile("s_setprio 1");
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int rep = 0; rep < 6; rep++)
#pragma unroll
                for (int i = 0; i < 10; i++)
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i % 5]) : "v"(a), "a"(b[(i + i / 5) % 5]));
        }
    } else {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int rep = 0; rep < 6; rep++)
#pragma unroll
                for (int i = 0; i < 5; i++) {
#pragma unroll
                    for (int q = 0; q < 2 * NF; q++) FILL(KIND, ((rep * 5 + i) * 2 * NF + q) & 7);
                }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 5; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) { float gv = g[i]; asm volatile("" : "+v"(gv)); s += f[i] + gv; }
    out[blockIdx.x * 256 * W + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int W, int KIND, int NF, int MODE, int PRIO> void run(double *tm, double *tf)
{
    static float *out = nullptr; static unsigned long long *cyc = nullptr; const int nb = 256, iters = 500;
    if (!out) { hipMalloc(&out, 4 * nb * 512); hipMalloc(&cyc, 8 * nb * 8); }
    hipLaunchKernelGGL((k<W, KIND, NF, MODE, PRIO>), dim3(nb), dim3(256 * W), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<W, KIND, NF, MODE, PRIO>), dim3(nb), dim3(256 * W), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256 * 8]; hipMemcpy(h, cyc, 8 * nb * 8, hipMemcpyDeviceToHost);
    if (MODE == 0) { *tm = h[7 * 8] / ((double)iters * 30 * W); *tf = h[7 * 8 + (W == 2 ? 4 : 0)] / ((double)iters * 30 * W); }
    else { *tm = h[7 * 8] / ((double)iters * 60); *tf = h[7 * 8 + 4] / ((double)iters * 60); }
}
template <int KIND, int NF> void cell()
{
    double a, b, c, d, e, f2, g, h;
    run<1, KIND, NF, 0, 0>(&a, &b);
    run<2, KIND, NF, 0, 0>(&c, &d);
    run<2, KIND, NF, 0, 1>(&e, &f2);
    run<2, KIND, NF, 1, 1>(&g, &h);
    printf("  NF %d: 1 wave/SIMD %6.2f | 2 waves/SIMD %6.2f (wave4 %6.2f) | +setprio %6.2f | specialised: mfma waves %6.2f, filler waves %6.2f\n", NF, a, c, d, e, g, h);
}
template <int KIND> void row(const char *name)
{
    printf("%s  (ticks per MFMA per SIMD)\n", name);
    cell<KIND, 1>(); cell<KIND, 2>(); cell<KIND, 3>(); cell<KIND, 4>();
}
int main()
{
    row<0>("v_max_f32"); row<3>("v_cvt_pk_f16_f32"); row<4>("v_cvt_f32_f16_sdwa"); row<6>("v_accvgpr_write"); row<11>("s_nop1+v_max_f32_dpp");
    return 0;
}
