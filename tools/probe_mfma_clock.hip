// What clock does the chip hold while EVERY SIMD issues fp16 MFMAs back to back?  v_mfma_f32_32x32x16_f16 on random operands, 8 waves
// per CU (two per SIMD, four independent accumulators each), for launch lengths from ~20 us to ~5 ms; per launch the shader clock
// ticks (s_memtime) over the constant 100 MHz ticks (s_memrealtime) of workgroup 0, and the achieved PFLOP/s from HIP events.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_clock.hip -o tools/bin/pclk && tools/bin/pclk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void kmfma(const float *in, float *out, long long *ticks, int iters, int valu_per_mfma, float amp)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (_Float16)(amp * in[(threadIdx.x * 8 + e) & 1023]); b[e] = (_Float16)(amp * in[(threadIdx.x * 8 + e + 512) & 1023]); }
    float filler = in[threadIdx.x & 1023];
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            for (int v = 0; v < valu_per_mfma; v++) filler = __builtin_fmaf(filler, 1.0000001f, 0.5f);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = filler;
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x < 8) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main()
{
    float *in, *out; long long *ticks;
    hipMalloc(&in, 4096); hipMalloc(&out, 4 * 512 * 1024); hipMalloc(&ticks, 128);
    float h[1024]; for (int i = 0; i < 1024; i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // amp: operand amplitude -- 0 = all-zero operands (no toggling), 1 = uniform random in [-0.5, 0.5), 4096 = fp16 plane-like magnitudes
    for (float amp : {0.f, 1.f, 4096.f})
        for (int iters : {300, 1500, 6000, 30000}) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kmfma, dim3(256), dim3(512), 0, 0, in, out, ticks, iters, 0, amp);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long t[16]; hipMemcpy(t, ticks, sizeof(t), hipMemcpyDeviceToHost);
                const double flop = (double)iters * 4 * 32768.0 * 256 * 8;
                if (rep == 1)
                    printf("operand amplitude %6.0f, %6d x 4 MFMAs per wave: %8.1f us  %6.3f PFLOP/s  shader clock %5.0f MHz (wg 0), %5.0f MHz (wg 7)\n", amp, iters,
                           ms * 1e3, flop / ms / 1e12, 100.0 * t[0] / t[1], 100.0 * t[14] / t[15]);
            }
        }
    return 0;
}
