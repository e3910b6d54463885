// Absolute v_mfma_f32_16x16x32_f16 rate (hipEvent time, not s_memtime): one wave per SIMD (256 threads x 256 blocks), two waves per
// SIMD as one 512-thread block per CU, two as two 256-thread blocks per CU, four (1024 threads).  5 independent accumulators per wave.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/pabs tools/probe_mfma_abs.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int T>
__global__ __launch_bounds__(T) void k(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[5];
    u32x4 a = {threadIdx.x, 1, 2, 3}, b[5];
    for (int i = 0; i < 5; i++) { acc[i] = (f32x4){0, 0, 0, 0}; b[i] = (u32x4){threadIdx.x + i, 5, 6, 7}; asm volatile("" : "+a"(b[i])); }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 6; rep++)
#pragma unroll
            for (int i = 0; i < 5; i++) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "a"(b[i]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 5; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * T + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int T> void run(int blocks, const char *name)
{
    static float *out = nullptr; static unsigned long long *cyc = nullptr;
    if (!out) { hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&cyc, 8 * 4096); }
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<T>), dim3(blocks), dim3(T), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<T>), dim3(blocks), dim3(T), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc + 7, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 30 * (T / 64) * blocks;
    printf("%-44s %8.3f ms  %8.1f TFLOP/s   s_memtime ticks per MFMA of one wave %.2f   (ticks per us: %.0f)\n", name, ms, nm * 16384 / ms / 1e9,
           (double)h / (iters * 30.0), (double)h / (ms * 1e3));
}
int main()
{
    run<256>(256, "1 wave/SIMD  (256 thr x 256 blocks)");
    run<512>(256, "2 waves/SIMD (512 thr x 256 blocks)");
    run<256>(512, "2 waves/SIMD (256 thr x 512 blocks)");
    run<1024>(256, "4 waves/SIMD (1024 thr x 256 blocks)");
    run<256>(256, "1 wave/SIMD  again");
    return 0;
}
