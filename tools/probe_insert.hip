// How fast does the (value, index) insertion network of featknn.hip issue with ONE wave per SIMD?
#include "../learning3d_amd/csrc/featknn.hip"
#include <cstdio>
thread_local int g_l3d_last_hip_error = 0;
template <int MODE>
__global__ __launch_bounds__(256) void ins_kernel(float *out, long long *cyc, int iters)
{
    TopK<20> top; top.init();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    float key = (float)(s >> 8);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) topk20_insert(top, key, it);
        else top.insert(key, it);
        key = key * 1.0001f + 3.f;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int i = 0; i < 20; i++) acc += top.v[i] + top.id[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    float *out; long long *cyc; hipMalloc(&out, 4 * 256 * 256); hipMalloc(&cyc, 8 * 256);
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(ins_kernel<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(ins_kernel<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("mode %d (%s): %.1f us total, %.1f ns per insert, s_memtime ticks/insert %.2f\n", mode, mode == 0 ? "asm" : "c++", ms * 1e3, ms * 1e6 / iters, (double)c / iters);
        }
    }
    return 0;
}
