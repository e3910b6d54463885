// scatter_det.hip -- deterministic backward of the gather-type ops.
//
// The reference's backward kernels scatter with fp32 atomicAdd (group_points_grad_kernel,
// utils/lib/src/group_points_gpu.cu:8-28; gather_points_grad, sampling_gpu.cu:37-52; three_interpolate_grad,
// interpolate_gpu.cu:185-205): the sum order, hence the low bits of every gradient, changes from run to run.
// Here every TARGET owns its sum and adds its contributions in ascending entry order:
//   1. keys[e] = b * T + idx[b][e]  for all B * E entries; a stable radix sort of (key, e) pairs (rocPRIM --
//      a library sort, not a hot-path kernel) groups the entries of each target, in entry order;
//   2. seg_start[t] = lower_bound(sorted keys, t)            (one thread per target);
//   3. dst[b][c][t] = sum_{p in segment(t)} src[b][c][e_p / div] * weight[e_p]      (one thread per (t, c)).
// One entry point serves the three ops:
//   grouping  : E = npoint * nsample, div = 1, no weight     dst = grad_points [B,C,N]
//   gather    : E = npoint,           div = 1, no weight     dst = grad_points [B,C,N]
//   3-interp  : E = n * 3,            div = 3, weight [B,n,3] dst = grad_points [B,C,m]
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

__global__ __launch_bounds__(256) void sd_keys_kernel(const int32_t *__restrict__ idx, int E, int T, long total,
                                                      uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int b = (int)(e / E);
    const int t = min(max(idx[e], 0), T - 1);
    keys[e] = (uint32_t)((long)b * T + t);
    vals[e] = (uint32_t)e;
}

__global__ __launch_bounds__(256) void sd_segments_kernel(const uint32_t *__restrict__ sorted, long total, long targets,
                                                          uint32_t *__restrict__ start)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t > targets) return;
    long lo = 0, hi = total;                                  // first position whose key >= t
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if ((long)sorted[mid] < t) lo = mid + 1; else hi = mid;
    }
    start[t] = (uint32_t)lo;
}

__global__ __launch_bounds__(256) void sd_sum_kernel(const float *__restrict__ src, const float *__restrict__ weight,
                                                     const uint32_t *__restrict__ order, const uint32_t *__restrict__ start,
                                                     int C, int T, int E, int div, float *__restrict__ dst)
{
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const long g = (long)b * T + t;
    const uint32_t p0 = start[g], p1 = start[g + 1];
    const int S = E / div;                                    // source points per cloud
    const float *sb = src + ((size_t)b * C + c) * S;
    const long ebase = (long)b * E;
    float acc = 0.f;
    for (uint32_t p = p0; p < p1; p++) {
        const uint32_t e = order[p];
        const int el = (int)((long)e - ebase);
        const float v = sb[el / div];
        acc += weight ? v * weight[e] : v;
    }
    dst[((size_t)b * C + c) * T + t] = acc;
}

static size_t sd_align(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t sd_sort_temp_bytes(long total, unsigned end_bit)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                              (uint32_t *)nullptr, (size_t)total, 0u, end_bit, (hipStream_t)0);
    return bytes;
}

static unsigned sd_bits(long targets)
{
    unsigned bits = 1;
    while (bits < 32 && (1L << bits) < targets) bits++;
    return bits;
}

extern "C" size_t l3d_scatter_add_det_workspace_bytes(int B, int T, int E)
{
    if (B <= 0 || T <= 0 || E <= 0) return 0;
    const long total = (long)B * E, targets = (long)B * T;
    return 4 * sd_align((size_t)total * 4) + sd_align((size_t)(targets + 1) * 4) + sd_align(sd_sort_temp_bytes(total, sd_bits(targets))) + 256;
}

extern "C" int l3d_scatter_add_det(const float *src, const int32_t *idx, const float *weight, int B, int C, int T,
                                   int E, int div, void *workspace, float *dst, l3d_stream_t stream)
{
    L3D_REQUIRE(src && idx && workspace && dst && B > 0 && C > 0 && T > 0 && E > 0 && div > 0 && E % div == 0);
    const long total = (long)B * E, targets = (long)B * T;
    if (total >= (1L << 31) || targets >= (1L << 31) || B > 65535 || C > 65535) return L3D_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    unsigned char *w = (unsigned char *)(((size_t)workspace + 255) & ~(size_t)255);
    const size_t seg = sd_align((size_t)total * 4);
    uint32_t *keys_in = (uint32_t *)w, *vals_in = (uint32_t *)(w + seg), *keys_out = (uint32_t *)(w + 2 * seg),
             *vals_out = (uint32_t *)(w + 3 * seg);
    uint32_t *start = (uint32_t *)(w + 4 * seg);
    void *temp = w + 4 * seg + sd_align((size_t)(targets + 1) * 4);
    const unsigned bits = sd_bits(targets);
    size_t temp_bytes = sd_sort_temp_bytes(total, bits);

    hipLaunchKernelGGL(sd_keys_kernel, dim3(l3d_divup(total, 256)), dim3(256), 0, st, idx, E, T, total, keys_in, vals_in);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)total, 0u, bits, st);
    if (e != hipSuccess) { g_l3d_last_hip_error = (int)e; return L3D_ERR_LAUNCH; }
    hipLaunchKernelGGL(sd_segments_kernel, dim3(l3d_divup(targets + 1, 256)), dim3(256), 0, st, (const uint32_t *)keys_out,
                       total, targets, start);
    hipLaunchKernelGGL(sd_sum_kernel, dim3(l3d_divup(T, 256), C, B), dim3(256), 0, st, src, weight, (const uint32_t *)vals_out,
                       (const uint32_t *)start, C, T, E, div, dst);
    return l3d_check_launch();
}
