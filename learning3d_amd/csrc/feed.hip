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
        // element (r, c) of [R | p; 0 0 0 1] by compile-time indices: R[r][c] with a per-thread r, c made the matrices private arrays (scratch)
        auto pick = [&](const double (&Rm)[3][3], const double (&pm)[3]) {
            float val = c == 3 ? 1.f : 0.f;                   // row 3
#pragma unroll
            for (int i = 0; i < 3; i++) {
#pragma unroll
                for (int j = 0; j < 3; j++) val = (r == i && c == j) ? (float)Rm[i][j] : val;
                val = (r == i && c == 3) ? (float)pm[i] : val;
            }
            return val;
        };
        if (which == 0) {
            igt[b * 16 + e] = pick(R, p);
        } else {
            const double wn[3] = {-w[0], -w[1], -w[2]}, vn[3] = {-v[0], -v[1], -v[2]};
            double Rn[3][3], pn[3];
            feed_se3_exp(wn, vn, Rn, pn);
            gt[b * 16 + e] = pick(Rn, pn);
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

// ---------------------------------------------------------------------------------------------------------------------
// l3d_sceneflow_batch -- SceneflowDataset.__getitem__ (data_utils/dataloaders.py:400-432) for a whole batch, from a dataset
// that lives in HBM (the processed FlyingThings3D set, `data_processed_maxcut_35_20k_2k_8192`, is ~10 GB: it fits once).
//   per sample b: scene s = scene_idx[b]; rows sample1[b][:] of points1 / color1 / flow / valid_mask1 and rows
//   sample2[b][:] of points2 / color2 (train: np.random.choice without replacement; test: the first npoints rows =
//   sample pointers NULL); centre = np.mean(pos1, 0) of the SAMPLED rows; pos1 -= centre; pos2 -= centre.
// np.mean over axis 0 of a float32 [S,3] array adds the rows one after the other in fp32 and divides in fp64
// (tests/test_oracle_golden.py pins that): one lane per coordinate replays exactly that chain over the rows staged in LDS,
// so the centred clouds are bit-identical to the reference's.
// One workgroup per sample; S <= 8192 (96 KB of LDS for the staged pos1 rows).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sceneflow_batch_kernel(
    const float *__restrict__ points1, const float *__restrict__ points2, const float *__restrict__ color1,
    const float *__restrict__ color2, const float *__restrict__ flow, const unsigned char *__restrict__ mask1,
    const long long *__restrict__ scene_idx, const int *__restrict__ sample1, const int *__restrict__ sample2, int n1, int n2, int S,
    float *__restrict__ o_pos1, float *__restrict__ o_pos2, float *__restrict__ o_color1, float *__restrict__ o_color2,
    float *__restrict__ o_flow, unsigned char *__restrict__ o_mask)
{
    extern __shared__ float sf_pos1[];          // [S][3]
    __shared__ float sf_centre[3];
    const int b = blockIdx.x;
    const size_t s = (size_t)scene_idx[b];
    const float *p1 = points1 + s * n1 * 3, *p2 = points2 + s * n2 * 3;
    const float *c1 = color1 + s * n1 * 3, *c2 = color2 + s * n2 * 3, *fl = flow + s * n1 * 3;
    const unsigned char *mk = mask1 + s * n1;
    const size_t ob = (size_t)b * S;
    for (int i = threadIdx.x; i < S; i += 256) {
        const int r1 = sample1 ? sample1[ob + i] : i;
        const int r2 = sample2 ? sample2[ob + i] : i;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            sf_pos1[i * 3 + c] = p1[(size_t)r1 * 3 + c];
            o_color1[(ob + i) * 3 + c] = c1[(size_t)r1 * 3 + c];
            o_flow[(ob + i) * 3 + c] = fl[(size_t)r1 * 3 + c];
            o_color2[(ob + i) * 3 + c] = c2[(size_t)r2 * 3 + c];
        }
        o_mask[ob + i] = mk[r1];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float acc = 0.f;                         // numpy's axis-0 reduction: row after row, fp32
        for (int i = 0; i < S; i++) acc = acc + sf_pos1[i * 3 + threadIdx.x];
        sf_centre[threadIdx.x] = (float)((double)acc / (double)S);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S; i += 256) {
        const int r2 = sample2 ? sample2[ob + i] : i;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            o_pos1[(ob + i) * 3 + c] = sf_pos1[i * 3 + c] - sf_centre[c];
            o_pos2[(ob + i) * 3 + c] = p2[(size_t)r2 * 3 + c] - sf_centre[c];
        }
    }
}

extern "C" int l3d_sceneflow_batch(const float *points1, const float *points2, const float *color1, const float *color2,
                                   const float *flow, const unsigned char *mask1, const long long *scene_idx, const int *sample1,
                                   const int *sample2, int B, int n1, int n2, int S, float *o_pos1, float *o_pos2, float *o_color1,
                                   float *o_color2, float *o_flow, unsigned char *o_mask, l3d_stream_t stream)
{
    L3D_REQUIRE(points1 && points2 && color1 && color2 && flow && mask1 && scene_idx && o_pos1 && o_pos2 && o_color1 && o_color2 &&
                o_flow && o_mask && B > 0 && n1 > 0 && n2 > 0 && S > 0 && S <= 8192 && S <= n1 && S <= n2);
    L3D_REQUIRE((sample1 == nullptr) == (sample2 == nullptr));
    const size_t lds = (size_t)S * 3 * sizeof(float);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)sceneflow_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { g_l3d_last_hip_error = (int)e; return L3D_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(sceneflow_batch_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, points1, points2, color1, color2, flow,
                       mask1, scene_idx, sample1, sample2, n1, n2, S, o_pos1, o_pos2, o_color1, o_color2, o_flow, o_mask);
    return l3d_check_launch();
}
