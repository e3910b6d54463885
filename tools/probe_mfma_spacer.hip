// Which spacer restores the single-wave issue rate of v_mfma_f32_16x16x32_bf16?  (back-to-back: ~29
// cycles/MFMA; with two v_max between MFMAs: ~20.)  SP selects the text put after every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SPACER(SP)                                                                   \
    if (SP == 1) asm volatile("v_nop\n\tv_nop");                                     \
    if (SP == 2) asm volatile("s_nop 0\n\ts_nop 0");                                 \
    if (SP == 3) asm volatile("s_nop 1");                                            \
    if (SP == 4) asm volatile("s_nop 3");                                            \
    if (SP == 5) asm volatile("v_nop");                                              \
    if (SP == 6) asm volatile("v_nop\n\tv_nop\n\tv_nop");                            \
    if (SP == 7) asm volatile("s_nop 7");                                            \
    if (SP == 8) asm volatile("v_nop\n\ts_nop 0");
template <int SP>
__global__ __launch_bounds__(256) void k16(const uint4 *in, float *out, int iters)
{
    f32x4 acc[20];
    for (int i = 0; i < 20; i++) for (int e = 0; e < 4; e++) acc[i][e] = 0.f;
    uint4 ua = in[threadIdx.x], ub = in[threadIdx.x + 256];
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 20; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            SPACER(SP)
        }
    }
    float s = 0; for (int i = 0; i < 20; i++) for (int e = 0; e < 4; e++) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int SP>
void run(const uint4 *in, float *out)
{
    const int iters = 20000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() { hipLaunchKernelGGL(k16<SP>, dim3(grid), dim3(256), 0, 0, in, out, iters); };
    go(); hipDeviceSynchronize();
    hipEventRecord(e0); go(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)iters * 20;
    printf("spacer %d: %.2f ms  %.1f cycles/MFMA @2.4GHz  (%.0f TFLOP/s)\n", SP, ms, ms * 1e-3 * 2.4e9 / nm, nm * 16384.0 * grid * 4 / ms / 1e9);
}
int main()
{
    uint4 *in; float *out; hipMalloc(&in, 16 * 512); hipMalloc(&out, 4 * 256 * 4096);
    unsigned h[2048]; for (int i = 0; i < 2048; i++) { unsigned r = (i * 2654435761u); h[i] = (r & 0x007f007fu) | 0x3f003e80u | ((r >> 3) & 0x80008000u); }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    printf("0 none, 1 2xv_nop, 2 2xs_nop0, 3 s_nop1, 4 s_nop3, 5 v_nop, 6 3xv_nop, 7 s_nop7, 8 v_nop+s_nop0\n");
    run<0>(in, out); run<1>(in, out); run<2>(in, out); run<3>(in, out); run<4>(in, out); run<5>(in, out); run<6>(in, out); run<7>(in, out); run<8>(in, out);
    return 0;
}
