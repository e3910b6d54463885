// probe.hip -- what the matrix pipe SUSTAINS on this chip, measured where the roofline is quoted (bench.py): every SIMD issues
// v_mfma_f32_32x32x16_f16 back to back on random operands (8 waves per CU, four independent accumulators each, no memory traffic).
// The dense fp16 peak of the data sheet (2.5 PFLOP/s at 2.4 GHz) assumes the shader clock holds; under this load power management
// drops it to ~1.5 GHz and the loop delivers ~1.55 PFLOP/s (all-zero operands: 2.1-2.2 at 2.15-2.25 GHz) -- tools/probe_mfma_clock.hip,
// profiles/round5_mfma_clock.txt.  The f16x2 kernels' fp32-equivalent ceiling is a third of whichever figure one takes.
#include "common.h"

typedef _Float16 pb_f16x8 __attribute__((ext_vector_type(8)));
typedef float pb_f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_sustained_kernel(int iters, float *__restrict__ sink, long long *__restrict__ ticks)
{
    pb_f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    pb_f16x8 a, b;
    unsigned s = threadIdx.x * 2654435761u + 12345u;
#pragma unroll
    for (int e = 0; e < 8; e++) {                          // uniform in [-0.5, 0.5): the operands of a real layer are not zeros
        s = s * 1664525u + 1013904223u; a[e] = (_Float16)((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f);
        s = s * 1664525u + 1013904223u; b[e] = (_Float16)((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f);
    }
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) v += acc[i][e];
    sink[(size_t)blockIdx.x * 512 + threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = r1 - r0; }      // shader clock ticks | 100 MHz ticks
}

// One launch of 256 workgroups x 8 waves x (4 iters) MFMAs of 32 768 FLOP each = iters * 2.68e11 FLOP; sink: 131 072 floats of scratch,
// ticks: two int64 (shader-clock ticks and constant 100 MHz ticks of workgroup 0 over its loop).  The caller times the launch.
extern "C" int l3d_probe_mfma_sustained(int iters, float *sink, long long *ticks, l3d_stream_t stream)
{
    L3D_REQUIRE(iters > 0 && sink && ticks);
    hipLaunchKernelGGL(mfma_sustained_kernel, dim3(256), dim3(512), 0, (hipStream_t)stream, iters, sink, ticks);
    return l3d_check_launch();
}
