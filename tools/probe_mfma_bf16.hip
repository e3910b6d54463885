// Pure bf16-MFMA ceiling probe: issue rate with random operands, no LDS, no barriers; short and long
// runs (sustained clocks) and 1/2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(const uint4 *in, float *out, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    uint4 ua = in[threadIdx.x], ub = in[threadIdx.x + 256];
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(const uint4 *in, float *out, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int e = 0; e < 4; e++) acc[i][e] = 0.f;
    uint4 ua = in[threadIdx.x], ub = in[threadIdx.x + 256];
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; i++) for (int e = 0; e < 4; e++) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    uint4 *in; float *out; hipMalloc(&in, 16 * 512); hipMalloc(&out, 4 * 256 * 4096);
    unsigned h[2048]; for (int i = 0; i < 2048; i++) { unsigned r = (i * 2654435761u); h[i] = (r & 0x007f007fu) | 0x3f003e80u | ((r >> 3) & 0x80008000u); }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++)
    for (int wgs = 1; wgs <= 2; wgs *= 2) {
        for (int which = 0; which < 3; which++) {
            const int iters = rep ? 400000 : 20000, grid = 256 * wgs;
            auto run = [&]() {
                if (which == 0) hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, in, out, iters);
                else if (which == 1) hipLaunchKernelGGL(k32<8>, dim3(grid), dim3(256), 0, 0, in, out, iters / 2);
                else hipLaunchKernelGGL(k16<20>, dim3(grid), dim3(256), 0, 0, in, out, iters / 5);
            };
            run(); hipDeviceSynchronize();
            hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double nm = which == 0 ? (double)iters * 4 : which == 1 ? (double)(iters / 2) * 8 : (double)(iters / 5) * 20;
            const double flop = nm * (which == 2 ? 16384.0 : 32768.0) * grid * 4;
            printf("iters=%d wgs/CU=%d %s: %.2f ms  %.1f TFLOP/s\n", iters, wgs, which == 0 ? "32x32x16 x4acc" : which == 1 ? "32x32x16 x8acc" : "16x16x32 x20acc", ms, flop / ms / 1e9);
        }
    }
    return 0;
}
