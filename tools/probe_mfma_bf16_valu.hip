// How many independent VALU ops fit in the shadow of one bf16 MFMA when a SIMD runs a single wave?
// k16<V>: 20 x v_mfma_f32_16x16x32_bf16 per iteration, V x v_max_f32 after each MFMA.
// k32<V>: 8 x v_mfma_f32_32x32x16_bf16 per iteration, V x v_max_f32 after each.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256) void k16(const uint4 *in, float *out, int iters)
{
    f32x4 acc[20];
    for (int i = 0; i < 20; i++) for (int e = 0; e < 4; e++) acc[i][e] = 0.f;
    uint4 ua = in[threadIdx.x], ub = in[threadIdx.x + 256];
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = (float)threadIdx.x * (i + 1);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 20; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < V; q++) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[(i * V + q) & 7]) : "v"(v[(i + q + 3) & 7]));
        }
    }
    float s = 0; for (int i = 0; i < 20; i++) for (int e = 0; e < 4; e++) s += acc[i][e];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V>
__global__ __launch_bounds__(256) void k32(const uint4 *in, float *out, int iters)
{
    f32x16 acc[8];
    for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    uint4 ua = in[threadIdx.x], ub = in[threadIdx.x + 256];
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = (float)threadIdx.x * (i + 1);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < V; q++) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[(i * V + q) & 7]) : "v"(v[(i + q + 3) & 7]));
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V, int W>
void run(const uint4 *in, float *out)
{
    const int iters = 100000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() { if (W == 16) hipLaunchKernelGGL(k16<V>, dim3(grid), dim3(256), 0, 0, in, out, iters / 5);
                      else hipLaunchKernelGGL(k32<V>, dim3(grid), dim3(256), 0, 0, in, out, iters / 2); };
    go(); hipDeviceSynchronize();
    hipEventRecord(e0); go(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double nm = W == 16 ? (double)(iters / 5) * 20 : (double)(iters / 2) * 8;
    printf("%s V=%d: %.2f ms  %.1f cycles/MFMA @2.4GHz  (%.0f TFLOP/s)\n", W == 16 ? "16x16x32" : "32x32x16", V, ms, ms * 1e-3 * 2.4e9 / nm,
           nm * (W == 16 ? 16384.0 : 32768.0) * grid * 4 / ms / 1e9);
}
int main()
{
    uint4 *in; float *out; hipMalloc(&in, 16 * 512); hipMalloc(&out, 4 * 256 * 4096);
    unsigned h[2048]; for (int i = 0; i < 2048; i++) { unsigned r = (i * 2654435761u); h[i] = (r & 0x007f007fu) | 0x3f003e80u | ((r >> 3) & 0x80008000u); }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 16>(in, out); run<1, 16>(in, out); run<2, 16>(in, out); run<3, 16>(in, out); run<4, 16>(in, out); run<6, 16>(in, out);
    run<0, 32>(in, out); run<2, 32>(in, out); run<4, 32>(in, out); run<6, 32>(in, out); run<8, 32>(in, out);
    return 0;
}
