// Pure-MFMA ceiling probe: fp32 MFMA issue rate with random (non-zero) operands, no LDS, no barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(const float *in, float *out, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(const float *in, float *out, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int e = 0; e < 4; e++) acc[i][e] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; i++) for (int e = 0; e < 4; e++) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 4096);
    float h[512]; for (int i = 0; i < 512; i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs = 1; wgs <= 4; wgs *= 2) {
        for (int which = 0; which < 2; which++) {
            const int iters = 20000, grid = 256 * wgs;
            auto run = [&]() {
                if (which == 0) hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(k16<20>, dim3(grid), dim3(256), 0, 0, in, out, iters / 5);
            };
            run(); hipDeviceSynchronize();
            hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double nm = which == 0 ? (double)iters * 4 : (double)(iters / 5) * 20;
            const double flop = nm * (which == 0 ? 4096.0 : 2048.0) * grid * 4;
            printf("wgs/CU=%d %s: %.2f ms  %.1f TFLOP/s\n", wgs, which == 0 ? "32x32x2 x4acc" : "16x16x4 x20acc", ms, flop / ms / 1e9);
        }
    }
    return 0;
}
