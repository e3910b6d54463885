// Issue rate of v_mfma_f32_16x16x32_f16 by operand home (one wave per SIMD, 5 independent accumulators, no memory):
//   form 0: D/C VGPR, A VGPR, B AGPR   form 1: D/C AGPR, A VGPR, B VGPR   form 2: all VGPR   form 3: D/C AGPR, A VGPR, B AGPR
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int FORM>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[5];
    u32x4 a = {threadIdx.x, 1, 2, 3}, b[5];
    for (int i = 0; i < 5; i++) { acc[i] = (f32x4){0, 0, 0, 0}; b[i] = (u32x4){threadIdx.x + i, 5, 6, 7}; }
    if (FORM == 0 || FORM == 3) for (int i = 0; i < 5; i++) asm volatile("" : "+a"(b[i]));
    if (FORM == 1 || FORM == 3) for (int i = 0; i < 5; i++) asm volatile("" : "+a"(acc[i]));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 6; rep++)
#pragma unroll
            for (int i = 0; i < 5; i++) {
                if (FORM == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "a"(b[i]));
                if (FORM == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b[i]));
                if (FORM == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b[i]));
                if (FORM == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "a"(b[i]));
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 5; i++) { f32x4 v = acc[i]; if (FORM == 1 || FORM == 3) asm volatile("" : "+v"(v)); s += v[0] + v[1] + v[2] + v[3]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int FORM> void run(const char *name)
{
    float *out; unsigned long long *cyc; const int nb = 256, iters = 2000;
    hipMalloc(&out, 4 * nb * 256); hipMalloc(&cyc, 8 * nb);
    hipLaunchKernelGGL(k<FORM>, dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<FORM>, dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, 8 * nb, hipMemcpyDeviceToHost);
    double n = (double)iters * 30;
    printf("%-34s %.2f ticks/MFMA (WG 0), %.2f PFLOP/s chip\n", name, h[0] / n, 256.0 * 4 * n * 16384 / (ms * 1e-3) / 1e15);
}
int main()
{
    run<0>("D/C VGPR, A VGPR, B AGPR");
    run<1>("D/C AGPR, A VGPR, B VGPR");
    run<2>("D/C VGPR, A VGPR, B VGPR");
    run<3>("D/C AGPR, A VGPR, B AGPR");
    return 0;
}
