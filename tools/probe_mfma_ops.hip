// Does the fp32 MFMA rate depend on operand variety?  20 accumulators (4 M x 5 N tiles), operands
// taken from register arrays like the real kernels (A varies per (e,mm), B per (q,t,e)).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(256, 1) void k(const float *in, float *out, int iters)
{
    f32x4 acc[4][5], hin[8][5], wa[4];
    for (int m = 0; m < 4; m++) for (int t = 0; t < 5; t++) acc[m][t] = (f32x4){0, 0, 0, 0};
    for (int q = 0; q < 8; q++) for (int t = 0; t < 5; t++) for (int e = 0; e < 4; e++) hin[q][t][e] = in[(threadIdx.x + q * 20 + t * 4 + e) & 511];
    for (int m = 0; m < 4; m++) for (int e = 0; e < 4; e++) wa[m][e] = in[(threadIdx.x * 3 + m * 4 + e) & 511];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int t = 0; t < 5; t++) {
                        if (VAR == 0) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][0], hin[0][0][0], acc[m][t], 0, 0, 0);
                        if (VAR == 1) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][e], hin[0][0][0], acc[m][t], 0, 0, 0);
                        if (VAR == 2) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][e], hin[q][t][e], acc[m][t], 0, 0, 0);
                    }
            if (VAR == 2) { // perturb so the compiler cannot hoist
#pragma unroll
                for (int m = 0; m < 4; m++) wa[m] = wa[(m + 1) & 3];
            }
        }
    }
    float s = 0; for (int m = 0; m < 4; m++) for (int t = 0; t < 5; t++) for (int e = 0; e < 4; e++) s += acc[m][t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int VAR> void run(const char *name, float *in, float *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200, grid = 256;
    hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), 0, 0, in, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), 0, 0, in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.1f TFLOP/s\n", name, (double)iters * 640 * 2048.0 * grid * 4 / ms / 1e9);
}
int main()
{
    float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 4096);
    float h[1024]; for (int i = 0; i < 1024; i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("same A, same B", in, out);
    run<1>("A varies (16 regs), same B", in, out);
    run<2>("A varies, B varies (160 regs)", in, out);
    run<0>("same A, same B (again)", in, out);
    return 0;
}
