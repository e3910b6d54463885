// feed.hip -- the data feed of the registration path ON THE DEVICE: at ~90 k clouds/s the
// reference's host pipeline (DataLoader workers, one scipy Rotation per sample, ops/transform_functions.py:271-315)
// would starve the GPU by two orders of magnitude.
//
//   l3d_uniform_clouds   seeded U(lo,hi)^3 clouds generated in place (counter-based hash: no host tensor, no copy)
//   l3d_euler_transform  DCPTransform / DeepGMRTransform for a whole batch: per cloud (anglez, angley, anglex) and a
//                        translation -> source = R template + t and the reference's `igt` ([R^T | t ; 0 0 0 1], the 3x3
//                        block being what scipy's Rotation.apply(np.eye(3)) returns, transform_functions.py:304-310).
//                        R = Rotation.from_euler('zyx', [az, ay, ax]) = Rx(ax) Ry(ay) Rz(az) (extrinsic z, y, x),
//                        evaluated in fp64 like scipy and rounded to fp32 once.
#include "common.h"

__device__ __forceinline__ unsigned feed_mix(unsigned long long x)
{
    // splitmix64 finaliser: every (seed, counter) pair gets an independent 64-bit value
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (unsigned)(x >> 40);                                   // 24 random bits
}

__global__ __launch_bounds__(256) void uniform_clouds_kernel(unsigned long long seed, size_t n, float lo, float hi,
                                                             float *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float u = (float)feed_mix(seed * 0x100000001B3ull + i) * (1.0f / 16777216.0f);     // [0, 1): 24-bit mantissa, exact
    out[i] = lo + (hi - lo) * u;
}

extern "C" int l3d_uniform_clouds(unsigned long long seed, int B, int N, float lo, float hi, float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(out && B > 0 && N > 0);
    const size_t n = (size_t)B * N * 3;
    hipLaunchKernelGGL(uniform_clouds_kernel, dim3((unsigned)l3d_divup((long)n, 256)), dim3(256), 0, (hipStream_t)stream, seed, n, lo, hi, out);
    return l3d_check_launch();
}

__global__ __launch_bounds__(256) void euler_transform_kernel(const float *__restrict__ tmpl, const float *__restrict__ euler_zyx,
                                                              const float *__restrict__ trans, int N,
                                                              float *__restrict__ source, float *__restrict__ igt)
{
    const int b = blockIdx.y;
    const double az = euler_zyx[b * 3], ay = euler_zyx[b * 3 + 1], ax = euler_zyx[b * 3 + 2];
    const double cz = cos(az), sz = sin(az), cy = cos(ay), sy = sin(ay), cx = cos(ax), sx = sin(ax);
    // R = Rx Ry Rz
    const double R[3][3] = {{cy * cz, -cy * sz, sy},
                            {sx * sy * cz + cx * sz, -sx * sy * sz + cx * cz, -sx * cy},
                            {-cx * sy * cz + sx * sz, cx * sy * sz + sx * cz, cx * cy}};
    const double t[3] = {trans[b * 3], trans[b * 3 + 1], trans[b * 3 + 2]};
    if (blockIdx.x == 0 && threadIdx.x < 16) {
        const int r = threadIdx.x >> 2, c = threadIdx.x & 3;
        igt[b * 16 + threadIdx.x] = r == 3 ? (c == 3 ? 1.f : 0.f) : (c == 3 ? (float)t[r] : (float)R[c][r]);     // R^T | t
    }
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float *p = tmpl + ((size_t)b * N + n) * 3;
    const double x = p[0], y = p[1], z = p[2];
    float *o = source + ((size_t)b * N + n) * 3;
    o[0] = (float)(R[0][0] * x + R[0][1] * y + R[0][2] * z + t[0]);
    o[1] = (float)(R[1][0] * x + R[1][1] * y + R[1][2] * z + t[1]);
    o[2] = (float)(R[2][0] * x + R[2][1] * y + R[2][2] * z + t[2]);
}

extern "C" int l3d_euler_transform(const float *tmpl, const float *euler_zyx, const float *trans, int B, int N,
                                   float *source, float *igt, l3d_stream_t stream)
{
    L3D_REQUIRE(tmpl && euler_zyx && trans && source && igt && B > 0 && N > 0 && B <= 65535);
    hipLaunchKernelGGL(euler_transform_kernel, dim3(l3d_divup(N, 256), B), dim3(256), 0, (hipStream_t)stream, tmpl, euler_zyx, trans, N,
                       source, igt);
    return l3d_check_launch();
}
