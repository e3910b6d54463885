// Where does an LDS-fed fp32 MFMA loop lose against the pure-MFMA ceiling?  Variants add, one at a
// time: operand ds_reads (pipelined), barriers, ds_writes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LD 132

template <int VAR>   // 0: regs only  1: + ds_read operands  2: + 2 barriers / 32 MFMA  3: + 16 ds_write / 32 MFMA  4: 1 barrier
__global__ __launch_bounds__(256) void k(const float *in, float *out, int iters)
{
    __shared__ float As[16][LD], Bs[16][LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 16 * LD; i += 256) { (&As[0][0])[i] = in[i & 511]; (&Bs[0][0])[i] = in[(i + 7) & 511]; }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    float ra = in[tid], rb = in[tid + 256];
    for (int it = 0; it < iters; it++) {
        if (VAR >= 3) {
            if (VAR != 4 || true) {
#pragma unroll
                for (int e = 0; e < 8; e++) { As[(tid & 1) * 8 + e][tid >> 1] = ra; Bs[(tid & 1) * 8 + e][tid >> 1] = rb; }
            }
        }
        if (VAR == 2 || VAR == 3) __syncthreads();
        float av[2][2], bv[2][2];
        if (VAR >= 1) {
            for (int i = 0; i < 2; i++) av[0][i] = As[kh][wm * 64 + i * 32 + l31];
            for (int j = 0; j < 2; j++) bv[0][j] = Bs[kh][wn * 64 + j * 32 + l31];
        } else { av[0][0] = ra; av[0][1] = rb; bv[0][0] = rb; bv[0][1] = ra; av[1][0] = rb; av[1][1] = ra; bv[1][0] = ra; bv[1][1] = rb; }
#pragma unroll
        for (int st = 0; st < 8; st++) {
            const int cr = st & 1, nx = cr ^ 1;
            if (VAR >= 1 && st + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 2; i++) av[nx][i] = As[2 * (st + 1) + kh][wm * 64 + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < 2; j++) bv[nx][j] = Bs[2 * (st + 1) + kh][wn * 64 + j * 32 + l31];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cr][i], bv[cr][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (VAR >= 2) __syncthreads();
    }
    float s = 0; for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}
template <int VAR> void run(const char *name, float *in, float *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs = 1; wgs <= 4; wgs *= 2) {
        const int iters = 2000, grid = 256 * wgs;
        hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), 0, 0, in, out, iters); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), 0, 0, in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s wgs/CU=%d  %.1f TFLOP/s\n", name, wgs, (double)iters * 32 * 4096.0 * grid * 4 / ms / 1e9);
    }
}
int main()
{
    float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 4096);
    float h[1024]; for (int i = 0; i < 1024; i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("regs only", in, out);
    run<1>("+ds_read operands", in, out);
    run<4>("+1 barrier/32mfma", in, out);
    run<2>("+2 barriers/32mfma", in, out);
    run<3>("+2 barriers +16 ds_write", in, out);
    return 0;
}
