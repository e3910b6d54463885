// L2-hit load throughput vs loads in flight: 512-thread workgroups, every workgroup streams a
// slab (shared by all workgroups with the same blockIdx.y, 4 slabs = 3 MB total, L2-resident) with D
// independent 16-byte loads per thread in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ld(const uint4 *w, size_t slab16, int iters, uint4 *out)
{
    const uint4 *p = w + (size_t)(blockIdx.x & 3) * slab16 + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    const int per_it = D * WAVES * 64;
    int off = ((blockIdx.x >> 2) * 7 * per_it) % (int)slab16;
    for (int it = 0; it < iters; it++) {
        uint4 r[D];
#pragma unroll
        for (int d = 0; d < D; d++) r[d] = p[off + d * WAVES * 64];
#pragma unroll
        for (int d = 0; d < D; d++) { acc.x ^= r[d].x; acc.y ^= r[d].y; acc.z ^= r[d].z; acc.w ^= r[d].w; }
        off += per_it; if (off + per_it > (int)slab16) off = 0;
    }
    if (acc.x == 0x12345) out[blockIdx.x * WAVES * 64 + threadIdx.x] = acc;
}
template <int D, int WAVES>
void run(const uint4 *w, size_t slab16, uint4 *out, int grid)
{
    const int iters = 4096 / D;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((ld<D, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, w, slab16, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((ld<D, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, w, slab16, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * WAVES * 64 * 16 * D * iters;
    printf("waves/WG=%d D=%2d grid=%d: %.1f us  %.2f TB/s  (%.1f B/clk/CU @2.4GHz, in flight %d KB/WG)\n", WAVES, D, grid, ms * 1e3,
           bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.4e9, D * WAVES);
}
int main()
{
    const size_t slab16 = 49152;          // 786432 B per slab
    uint4 *w, *out; hipMalloc(&w, 4 * slab16 * 16); hipMalloc(&out, 16ul * 2048 * 1024);
    hipMemset(w, 1, 4 * slab16 * 16);
    for (int grid = 256; grid <= 1024; grid *= 2) {
        run<1, 8>(w, slab16, out, grid); run<3, 8>(w, slab16, out, grid); run<6, 8>(w, slab16, out, grid); run<12, 8>(w, slab16, out, grid);
        run<3, 16>(w, slab16, out, grid); run<6, 16>(w, slab16, out, grid);
    }
    return 0;
}
