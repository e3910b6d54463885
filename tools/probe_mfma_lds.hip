// How much do ds_read_b128 fragment reads slow a stream of v_mfma_f32_32x32x16_f16?  conv_f16's chunk is 14 reads + 24 MFMAs
// per wave, two waves per SIMD.  Variants: reads per 24 MFMAs = 0 / 6 / 14 / 14 with the compiler free to interleave.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_lds.hip -o tools/bin/probe_mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NR, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 40960 / 4; i += 64 * WAVES) ((float *)lds)[i] = 0.001f * (i & 63);
    __syncthreads();
    f32x16 acc[8];
    for (int a = 0; a < 8; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    f16x8 fr[14];
    for (int f = 0; f < 14; f++) fr[f] = *(const f16x8 *)(lds + ((f * 1024 + wave * 512 + lane * 16) % 40960));
    for (int it = 0; it < iters; it++) {
        const unsigned char *base = lds + ((it & 1) * 16);
#pragma unroll
        for (int f = 0; f < NR; f++) fr[f] = *(const f16x8 *)(base + ((f * 2048 + (wave & 3) * 512 + lane * 16) % 40000));
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int c = 0; c < 4; c++)
                    acc[a * 4 + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[8 + p * 2 + a], fr[(p & 1) * 4 + c], acc[a * 4 + c], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < 8; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
    out[blockIdx.x * 64 * WAVES + t] = s;
}

// conv_f16_kernel's own fragment addressing (three 40 KB stages, W planes then x planes, lane halves 4 KB apart), no DMA, no
// barrier: does the address pattern itself cost anything?
template <int KGS, int ROWPAD>
__global__ __launch_bounds__(512) void k_conv(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave & 3, wn = wave >> 2, kgl = lane >> 5;
    for (int i = t; i < 30 * KGS / 4; i += 512) ((float *)lds)[i] = 0.001f * (i & 63);
    __syncthreads();
    f32x16 acc[2][4];
    for (int a = 0; a < 2; a++) for (int c = 0; c < 4; c++) for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;
    const int a_off = kgl * KGS + (wm * 64 + (lane & 31)) * 16 + wm * ROWPAD, b_off = 6 * KGS + kgl * KGS + (wn * 128 + (lane & 31)) * 16 + wn * 2 * ROWPAD;
    int stage = 0;
    for (int it = 0; it < iters; it++) {
        const unsigned char *base = lds + stage * (10 * KGS);
        f16x8 A[2][3], Bf[4][2];
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int c = 0; c < 4; c++) Bf[c][p] = *(const f16x8 *)(base + b_off + c * 512 + p * 2 * KGS);
#pragma unroll
        for (int p = 2; p >= 0; p--)
#pragma unroll
            for (int a = 0; a < 2; a++) A[a][p] = *(const f16x8 *)(base + a_off + a * 512 + p * 2 * KGS);
#pragma unroll
        for (int prod = 0; prod < 3; prod++) {
            const int pa = prod == 0 ? 2 : (prod == 1 ? 1 : 0), pb = prod == 1 ? 1 : 0;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[a][pa], Bf[c][pb], acc[a][c], 0, 0, 0);
        }
        stage = stage == 2 ? 0 : stage + 1;
        asm volatile("" ::: "memory");
    }
    float s = 0;
    for (int a = 0; a < 2; a++) for (int c = 0; c < 4; c++) for (int r = 0; r < 16; r++) s += acc[a][c][r];
    out[blockIdx.x * 512 + t] = s;
}

template <int NR, int WAVES>
void run(const char *name, int blocks_per_cu)
{
    float *out; hipMalloc(&out, 256 * 4 * 512 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NR, WAVES>), dim3(256 * blocks_per_cu), dim3(64 * WAVES), 40960, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = WAVES * blocks_per_cu / 4.0;
    const double cyc = ms * 1e-3 * 2.4e9 / iters;          // per iteration of ALL co-resident waves
    printf("%-44s %7.3f ms  %7.0f cycles / iteration at 2.4 GHz  (%.1f waves per SIMD -> MFMA pipe time %4.0f)  %5.0f TFLOP/s\n", name, ms,
           cyc, waves_per_simd, 24 * 32 * waves_per_simd, 256.0 * blocks_per_cu * WAVES * iters * 24 * 32768.0 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
template <int KGS, int ROWPAD>
void run_conv(int iters, int blocks)
{
    float *out; hipMalloc(&out, (size_t)blocks * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_conv<KGS, ROWPAD>), dim3(blocks), dim3(512), 30 * KGS, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double rounds = blocks / 256.0;
    printf("kg stride %d rowpad %d: conv_f16 addressing, %d blocks x %d chunks: %8.1f us  = %6.0f cycles per chunk (two waves per SIMD; MFMA pipe time 1536)  %5.0f TFLOP/s\n", KGS, ROWPAD, blocks, iters,
           ms * 1e3, ms * 1e-3 * 2.4e9 / (iters * rounds), (double)blocks * 8 * iters * 24 * 32768.0 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main()
{
    run_conv<4096, 0>(2000, 256);
    run_conv<4096 + 16, 0>(2000, 256);
    run_conv<4096 + 32, 0>(2000, 256);
    run_conv<4096 + 64, 0>(2000, 256);
    run_conv<4096 + 128, 0>(2000, 256);
    run_conv<4096 + 256, 0>(2000, 256);
    run_conv<4096 + 512, 0>(2000, 256);
    run_conv<4096, 16>(2000, 256);
    run_conv<4096 + 64, 16>(2000, 256);
    run<0, 8>("8 waves/CU x1, 0 reads per 24 MFMA", 1);
    run<6, 8>("8 waves/CU x1, 6 reads", 1);
    run<14, 8>("8 waves/CU x1, 14 reads", 1);
    run<0, 4>("4 waves/CU x1 (1 per SIMD), 0 reads", 1);
    run<14, 4>("4 waves/CU x1 (1 per SIMD), 14 reads", 1);
    run<14, 4>("4 waves x2 blocks/CU, 14 reads", 2);
    run<0, 4>("4 waves x2 blocks/CU, 0 reads", 2);
    return 0;
}
