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

// ---------------------------------------------------------------------------------------------
// PNLKTransform / RPMNetTransform (ops/transform_functions.py:109-192): a twist vector x = (w, v) per cloud ->
//   g  = se3.exp(x)  (ops/se3.py:51-74: R = I + sinc1(t) W + sinc2(t) W^2, p = (I + sinc2(t) W + sinc3(t) W^2) v, t = |w|,
//                    the sinc helpers of ops/sinc.py with their Taylor branch below 0.01)   = the reference's `igt`
//   gt = se3.exp(-x)                                                                         = the reference's `gt`
//   source = R template + p.   Evaluated in fp64, rounded to fp32 once.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void feed_se3_exp(const double w[3], const double v[3], double R[3][3], double p[3])
{
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = sqrt(t2);
    double s1, s2, s3;
    if (t < 0.01) {                                  // ops/sinc.py:14, :100, :129
        s1 = 1 - t2 / 6 * (1 - t2 / 20 * (1 - t2 / 42));
        s2 = 0.5 * (1 - t2 / 12 * (1 - t2 / 30 * (1 - t2 / 56)));
        s3 = 1.0 / 6 * (1 - t2 / 20 * (1 - t2 / 42 * (1 - t2 / 72)));
    } else {
        s1 = sin(t) / t;
        s2 = (1 - cos(t)) / t2;
        s3 = (t - sin(t)) / (t2 * t);
    }
    const double W[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
    double S[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) S[i][j] = W[i][0] * W[0][j] + W[i][1] * W[1][j] + W[i][2] * W[2][j];
    for (int i = 0; i < 3; i++) {
        p[i] = 0;
        for (int j = 0; j < 3; j++) {
            R[i][j] = (i == j ? 1.0 : 0.0) + s1 * W[i][j] + s2 * S[i][j];
            p[i] += ((i == j ? 1.0 : 0.0) + s2 * W[i][j] + s3 * S[i][j]) * v[j];
        }
    }
}

__global__ __launch_bounds__(256) void twist_transform_kernel(const float *__restrict__ tmpl, const float *__restrict__ twist, int N,
                                                              float *__restrict__ source, float *__restrict__ igt, float *__restrict__ gt)
{
    const int b = blockIdx.y;
    const double w[3] = {twist[b * 6], twist[b * 6 + 1], twist[b * 6 + 2]}, v[3] = {twist[b * 6 + 3], twist[b * 6 + 4], twist[b * 6 + 5]};
    double R[3][3], p[3];
    feed_se3_exp(w, v, R, p);
    if (blockIdx.x == 0 && threadIdx.x < 32) {
        const int which = threadIdx.x >> 4, e = threadIdx.x & 15, r = e >> 2, c = e & 3;
        if (which == 0) {
            igt[b * 16 + e] = r == 3 ? (c == 3 ? 1.f : 0.f) : (c == 3 ? (float)p[r] : (float)R[r][c]);
        } else {
            const double wn[3] = {-w[0], -w[1], -w[2]}, vn[3] = {-v[0], -v[1], -v[2]};
            double Rn[3][3], pn[3];
            feed_se3_exp(wn, vn, Rn, pn);
            gt[b * 16 + e] = r == 3 ? (c == 3 ? 1.f : 0.f) : (c == 3 ? (float)pn[r] : (float)Rn[r][c]);
        }
    }
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float *q = tmpl + ((size_t)b * N + n) * 3;
    const double x = q[0], y = q[1], z = q[2];
    float *o = source + ((size_t)b * N + n) * 3;
    o[0] = (float)(R[0][0] * x + R[0][1] * y + R[0][2] * z + p[0]);
    o[1] = (float)(R[1][0] * x + R[1][1] * y + R[1][2] * z + p[1]);
    o[2] = (float)(R[2][0] * x + R[2][1] * y + R[2][2] * z + p[2]);
}

extern "C" int l3d_twist_transform(const float *tmpl, const float *twist, int B, int N, float *source, float *igt, float *gt,
                                   l3d_stream_t stream)
{
    L3D_REQUIRE(tmpl && twist && source && igt && gt && B > 0 && N > 0 && B <= 65535);
    hipLaunchKernelGGL(twist_transform_kernel, dim3(l3d_divup(N, 256), B), dim3(256), 0, (hipStream_t)stream, tmpl, twist, N, source,
                       igt, gt);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// PCRNetTransform.__call__ (ops/transform_functions.py:194-269): pose [B,7] = (quaternion w x y z, translation) ->
//   q = normalised quaternion (create_pose_7d, :218-227: F.normalize, eps 1e-12),
//   source = qrot(q, template) + t,  qrot(q, v) = v + 2 (q_w (q_xyz x v) + q_xyz x (q_xyz x v))   (ops/quaternion.py:35-53)
// in the reference's fp32 operation order (cross products, one multiply-add chain per component: no fusing).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quat_transform_kernel(const float *__restrict__ tmpl, const float *__restrict__ pose7, int N,
                                                             float *__restrict__ source)
{
    const int b = blockIdx.y;
    const float *ps = pose7 + b * 7;
    const float nrm = fmaxf(sqrtf(((ps[0] * ps[0] + ps[1] * ps[1]) + ps[2] * ps[2]) + ps[3] * ps[3]), 1e-12f);
    const float qw = ps[0] / nrm, qx = ps[1] / nrm, qy = ps[2] / nrm, qz = ps[3] / nrm;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float *p = tmpl + ((size_t)b * N + n) * 3;
    const float x = p[0], y = p[1], z = p[2];
    const float ux = qy * z - qz * y, uy = qz * x - qx * z, uz = qx * y - qy * x;              // uv = q_xyz x v
    const float wx = qy * uz - qz * uy, wy = qz * ux - qx * uz, wz = qx * uy - qy * ux;        // uuv = q_xyz x uv
    float *o = source + ((size_t)b * N + n) * 3;
    o[0] = (x + 2.0f * (qw * ux + wx)) + ps[4];
    o[1] = (y + 2.0f * (qw * uy + wy)) + ps[5];
    o[2] = (z + 2.0f * (qw * uz + wz)) + ps[6];
}

extern "C" int l3d_quat_transform(const float *tmpl, const float *pose7, int B, int N, float *source, l3d_stream_t stream)
{
    L3D_REQUIRE(tmpl && pose7 && source && B > 0 && N > 0 && B <= 65535);
    hipLaunchKernelGGL(quat_transform_kernel, dim3(l3d_divup(N, 256), B), dim3(256), 0, (hipStream_t)stream, tmpl, pose7, N, source);
    return l3d_check_launch();
}
