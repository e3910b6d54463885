// train.hip -- training-mode BatchNorm around the 1x1-conv GEMMs (reference: the torch modules of
// models/dgcnn.py:34-48 / models/pcn.py in .train(), examples/train_pcn.py:70-91).
//
// conv (HIP GEMM) -> [these kernels] -> next layer.  Batch statistics are formed as PER-CLOUD fp64 partial sums
// (one workgroup per (cloud, channel), fixed summation order, no atomics) which the host adds in global cloud order --
// after an all_gather when the batch is sharded over ranks -- so the statistics are bit-identical for any number of
// GPUs, and the backward's two reductions follow the same scheme.
//   l3d_channel_stats        z [B,C,P]                      -> part [B,C,2] fp64 = (sum z, sum z^2) per (cloud, channel)
//   l3d_bn_act_forward       y = act(z * scale[c] + shift[c])                              (act: 0 none, 1 ReLU)
//   l3d_bn_backward_stats    dy, z, scale, shift, mean, rstd  -> part [B,C,2] fp64 = (sum g, sum g zhat),
//                            g = dy * act'(z scale + shift) (mask recomputed, nothing extra saved; act: 0 none, 1 ReLU, else a
//                            LeakyReLU slope's bits as in common.h), zhat = (z - mean) rstd
//   l3d_bn_act_backward      dz = gr[c] * (g - m1[c] - zhat * m2[c])      (gr = gamma rstd, m1 = sum g / n, m2 = sum g zhat / n)
//   l3d_sum_clouds_f64       tot[j] = part[0][j] + part[1][j] + ... in cloud order (one thread per j: a fixed left-to-right order)
// The per-channel constants of the two backward kernels (mean, rstd, gr, m1, m2) are fp64 and dz is evaluated in fp64 and
// rounded once: sum_p dz is zero by construction, and the weight gradient sums dz x over 10^5..10^6 points, so an fp32
// rounding of m1 (a SYSTEMATIC error, the same for every point) would be multiplied by the point count.  These kernels are
// HBM-bound; the fp64 arithmetic is free.  Eval-mode BatchNorm / plain bias layers use the same kernels with m1 = m2 = 0.
#include "common.h"

// d act(v) / dv for the library's activation code (common.h l3d_act): 0 none, 1 ReLU, else the bits of a LeakyReLU slope
__device__ __forceinline__ float tr_act_grad(float v, int act)
{
    if (!act || v > 0.f) return 1.f;
    return act == 1 ? 0.f : __int_as_float(act);
}

__device__ __forceinline__ double tr_block_sum(double v, double *sh)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void channel_stats_kernel(const float *__restrict__ z, int C, long P, double *__restrict__ part)
{
    __shared__ double sh[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const float *row = z + ((size_t)b * C + c) * P;
    double s = 0.0, q = 0.0;
    if (P % 4 == 0 && (((size_t)row) & 15) == 0) {               // uniform: 16-byte loads, four values per trip (a fixed order all the same)
        for (long p = (long)threadIdx.x * 4; p < P; p += 1024) {
            const float4 v4 = *(const float4 *)(row + p);
            const double v0 = v4.x, v1 = v4.y, v2 = v4.z, v3 = v4.w;
            s += (v0 + v1) + (v2 + v3);
            q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
        }
    } else {
        for (long p = threadIdx.x; p < P; p += 256) {
            const double v = row[p];
            s += v;
            q += v * v;
        }
    }
    s = tr_block_sum(s, sh);
    q = tr_block_sum(q, sh);
    if (threadIdx.x == 0) {
        part[((size_t)b * C + c) * 2] = s;
        part[((size_t)b * C + c) * 2 + 1] = q;
    }
}

extern "C" int l3d_channel_stats(const float *z, int B, int C, long P, double *part, l3d_stream_t stream)
{
    L3D_REQUIRE(z && part && B > 0 && C > 0 && P > 0 && B <= 65535);
    hipLaunchKernelGGL(channel_stats_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, z, C, P, part);
    return l3d_check_launch();
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_act_forward_kernel(const float *__restrict__ z, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, int C, long P, int act,
                                                             float *__restrict__ y)
{
    const int c = blockIdx.y, b = blockIdx.z;
    const float sc = scale[c], sh = shift[c];
    if (VEC) {
        const long p = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
        if (p >= P) return;
        const size_t i = ((size_t)b * C + c) * P + p;
        const float4 z4 = *(const float4 *)(z + i);
        float v[4] = {z4.x * sc + sh, z4.y * sc + sh, z4.z * sc + sh, z4.w * sc + sh};
        if (act)
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = l3d_act(v[e], act);
        *(float4 *)(y + i) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        const long p = (long)blockIdx.x * 256 + threadIdx.x;
        if (p >= P) return;
        const size_t i = ((size_t)b * C + c) * P + p;
        const float v = z[i] * sc + sh;
        y[i] = act ? l3d_act(v, act) : v;
    }
}

extern "C" int l3d_bn_act_forward(const float *z, const float *scale, const float *shift, int B, int C, long P, int act,
                                  float *y, l3d_stream_t stream)
{
    L3D_REQUIRE(z && scale && shift && y && B > 0 && C > 0 && P > 0 && B <= 65535 && C <= 65535);
    if (P % 4 == 0 && ((((size_t)z) | ((size_t)y)) & 15) == 0)
        hipLaunchKernelGGL(bn_act_forward_kernel<true>, dim3((unsigned)l3d_divup(P, 1024), C, B), dim3(256), 0, (hipStream_t)stream, z, scale,
                           shift, C, P, act, y);
    else
        hipLaunchKernelGGL(bn_act_forward_kernel<false>, dim3((unsigned)l3d_divup(P, 256), C, B), dim3(256), 0, (hipStream_t)stream, z, scale,
                           shift, C, P, act, y);
    return l3d_check_launch();
}

// The gradient that arrives at y [B][C][P]: dy (or nothing) plus, for a layer whose output is also max-pooled over runs of K
// consecutive positions (the max over the k neighbours behind an EdgeConv layer, models/dgcnn.py:36-46), dpool [B][C][P/K] at the
// position pidx names -- the dense scatter of l3d_max_last_backward and autograd's add of the two gradient tensors, never formed.
__device__ __forceinline__ float tr_grad_in(const float *__restrict__ dy, const float *__restrict__ dpool,
                                            const unsigned char *__restrict__ pidx, size_t base, size_t pbase, long p, int K, float kinv)
{
    float g = dy ? dy[base + p] : 0.f;
    if (dpool) {
        const long n = (long)(((float)p + 0.5f) * kinv);                 // p / K: exact for every K <= 256 when p < 2^22 (brute force; from p = 4 243 964 on, K = 255 and 81 other K fail)
        if ((int)(p - n * K) == (int)pidx[pbase + n]) g = g + dpool[pbase + n];
    }
    return g;
}

// VEC: four consecutive positions per trip (P % 4 == 0, K % 4 == 0: they lie in one pooled run, one pidx / dpool load serves all
// four) -- with the pooled gradient's two extra loads per element the scalar form went from memory-bound to instruction-bound
// (0.95 -> 1.38 ms over a DGCNN step's five layers)
template <bool VEC>
__global__ __launch_bounds__(256) void bn_backward_stats_kernel(const float *__restrict__ dy, const float *__restrict__ z,
                                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                                const double *__restrict__ mean, const double *__restrict__ rstd,
                                                                int C, long P, int act, double *__restrict__ part,
                                                                const float *__restrict__ dpool, const unsigned char *__restrict__ pidx,
                                                                int K)
{
    __shared__ double sh[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const size_t base = ((size_t)b * C + c) * P, pbase = dpool ? ((size_t)b * C + c) * (P / K) : 0;
    const float sc = scale[c], shf = shift[c], kinv = dpool ? 1.0f / (float)K : 0.f;
    const double mu = mean[c], rs = rstd[c];
    double s = 0.0, q = 0.0;
    if (VEC) {
        for (long p = (long)threadIdx.x * 4; p < P; p += 1024) {
            const float4 z4 = *(const float4 *)(z + base + p);
            float4 g4 = dy ? *(const float4 *)(dy + base + p) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (dpool) {
                const long n = (long)(((float)p + 0.5f) * kinv);
                const int k0 = (int)(p - n * K), at = (int)pidx[pbase + n];
                const float dp = dpool[pbase + n];
                g4.x = at == k0 ? g4.x + dp : g4.x; g4.y = at == k0 + 1 ? g4.y + dp : g4.y;
                g4.z = at == k0 + 2 ? g4.z + dp : g4.z; g4.w = at == k0 + 3 ? g4.w + dp : g4.w;
            }
            const float zv[4] = {z4.x, z4.y, z4.z, z4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float g = gv[e] * tr_act_grad(zv[e] * sc + shf, act);
                s += (double)g;
                q += (double)g * (((double)zv[e] - mu) * rs);
            }
        }
    } else {
        for (long p = threadIdx.x; p < P; p += 256) {
            const float zv = z[base + p];
            const float g = tr_grad_in(dy, dpool, pidx, base, pbase, p, K, kinv) * tr_act_grad(zv * sc + shf, act);
            s += (double)g;
            q += (double)g * (((double)zv - mu) * rs);
        }
    }
    s = tr_block_sum(s, sh);
    q = tr_block_sum(q, sh);
    if (threadIdx.x == 0) {
        part[((size_t)b * C + c) * 2] = s;
        part[((size_t)b * C + c) * 2 + 1] = q;
    }
}

// dy may be NULL when dpool is given (a layer whose output is only pooled); dpool [B][C][P/K], pidx [B][C][P/K] (l3d_max_last's
// arg-max), K = the pooled run length: P % K == 0, P < 2^22
// dpool / pidx / K: the pooled gradient of a layer whose output is also max-pooled over runs of K (NULL / NULL / 0: none)
extern "C" int l3d_bn_backward_stats(const float *dy, const float *z, const float *scale, const float *shift, const double *mean,
                                          const double *rstd, int B, int C, long P, int act, double *part, const float *dpool,
                                          const unsigned char *pidx, int K, l3d_stream_t stream)
{
    L3D_REQUIRE((dy || dpool) && z && scale && shift && mean && rstd && part && B > 0 && C > 0 && P > 0 && B <= 65535);
    L3D_REQUIRE(!dpool || (pidx && K > 0 && K <= 256 && P % K == 0 && P < (1L << 22)));
    const bool vec = P % 4 == 0 && (!dpool || K % 4 == 0) && ((((size_t)dy) | ((size_t)z)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(bn_backward_stats_kernel<true>, dim3(C, B), dim3(256), 0, (hipStream_t)stream, dy, z, scale, shift, mean, rstd, C,
                           P, act, part, dpool, pidx, K);
    else
        hipLaunchKernelGGL(bn_backward_stats_kernel<false>, dim3(C, B), dim3(256), 0, (hipStream_t)stream, dy, z, scale, shift, mean, rstd, C,
                           P, act, part, dpool, pidx, K);
    return l3d_check_launch();
}


template <bool VEC>
__global__ __launch_bounds__(256) void bn_act_backward_kernel(const float *__restrict__ dy, const float *__restrict__ z,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              const double *__restrict__ mean, const double *__restrict__ rstd,
                                                              const double *__restrict__ gr, const double *__restrict__ m1,
                                                              const double *__restrict__ m2, int C, long P, int act,
                                                              float *__restrict__ dz, const float *__restrict__ dpool,
                                                              const unsigned char *__restrict__ pidx, int K)
{
    const int c = blockIdx.y, b = blockIdx.z;
    const size_t base = ((size_t)b * C + c) * P, pbase = dpool ? ((size_t)b * C + c) * (P / K) : 0;
    const float kinv = dpool ? 1.0f / (float)K : 0.f, sc = scale[c], shf = shift[c];
    const double grc = gr[c], m1c = m1[c], m2c = m2[c], mu = mean[c], rs = rstd[c];
    if (VEC) {
        const long p = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
        if (p >= P) return;
        const float4 z4 = *(const float4 *)(z + base + p);
        float4 g4 = dy ? *(const float4 *)(dy + base + p) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (dpool) {
            const long n = (long)(((float)p + 0.5f) * kinv);
            const int k0 = (int)(p - n * K), at = (int)pidx[pbase + n];
            const float dp = dpool[pbase + n];
            g4.x = at == k0 ? g4.x + dp : g4.x; g4.y = at == k0 + 1 ? g4.y + dp : g4.y;
            g4.z = at == k0 + 2 ? g4.z + dp : g4.z; g4.w = at == k0 + 3 ? g4.w + dp : g4.w;
        }
        const float zv[4] = {z4.x, z4.y, z4.z, z4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float g = gv[e] * tr_act_grad(zv[e] * sc + shf, act);
            o[e] = (float)(grc * ((double)g - m1c - ((double)zv[e] - mu) * rs * m2c));
        }
        *(float4 *)(dz + base + p) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        const long p = (long)blockIdx.x * 256 + threadIdx.x;
        if (p >= P) return;
        const size_t i = base + p;
        const float zv = z[i];
        const float g = tr_grad_in(dy, dpool, pidx, base, pbase, p, K, kinv) * tr_act_grad(zv * sc + shf, act);
        dz[i] = (float)(grc * ((double)g - m1c - ((double)zv - mu) * rs * m2c));
    }
}

// dpool / pidx / K: the pooled gradient of a layer whose output is also max-pooled over runs of K (NULL / NULL / 0: none)
extern "C" int l3d_bn_act_backward(const float *dy, const float *z, const float *scale, const float *shift, const double *mean,
                                        const double *rstd, const double *gr, const double *m1, const double *m2, int B, int C, long P,
                                        int act, float *dz, const float *dpool, const unsigned char *pidx, int K, l3d_stream_t stream)
{
    L3D_REQUIRE((dy || dpool) && z && scale && shift && mean && rstd && gr && m1 && m2 && dz && B > 0 && C > 0 && P > 0 && B <= 65535 &&
                C <= 65535);
    L3D_REQUIRE(!dpool || (pidx && K > 0 && K <= 256 && P % K == 0 && P < (1L << 22)));
    const bool vec = P % 4 == 0 && (!dpool || K % 4 == 0) && ((((size_t)dy) | ((size_t)z) | ((size_t)dz)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(bn_act_backward_kernel<true>, dim3((unsigned)l3d_divup(P, 1024), C, B), dim3(256), 0, (hipStream_t)stream, dy, z,
                           scale, shift, mean, rstd, gr, m1, m2, C, P, act, dz, dpool, pidx, K);
    else
        hipLaunchKernelGGL(bn_act_backward_kernel<false>, dim3((unsigned)l3d_divup(P, 256), C, B), dim3(256), 0, (hipStream_t)stream, dy, z,
                           scale, shift, mean, rstd, gr, m1, m2, C, P, act, dz, dpool, pidx, K);
    return l3d_check_launch();
}


__global__ __launch_bounds__(256) void sum_clouds_f64_kernel(const double *__restrict__ part, int B, long M, double *__restrict__ tot)
{
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    double s = 0.0;
    for (int b = 0; b < B; b++) s += part[(size_t)b * M + j];
    tot[j] = s;
}

extern "C" int l3d_sum_clouds_f64(const double *part, int B, long M, double *tot, l3d_stream_t stream)
{
    L3D_REQUIRE(part && tot && B > 0 && M > 0);
    hipLaunchKernelGGL(sum_clouds_f64_kernel, dim3((unsigned)l3d_divup(M, 256)), dim3(256), 0, (hipStream_t)stream, part, B, M, tot);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// max over the last (contiguous) axis of x [R][K] with its arg-max, and the backward of that max: the "max over the k
// neighbours" behind every EdgeConv layer of a DGCNN training step (models/dgcnn.py:36-46 of the reference:
// `x.max(dim=-1, keepdim=True)[0]`).  torch's generic reduction spent 0.34 ms per layer on it (5x the tensor's read time) and its
// backward a zero fill plus an index scatter; here a thread takes a row, the arg-max is the FIRST maximum (torch's rule) in one
// byte, and the backward writes the dense gradient in one pass.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void max_last_kernel(const float *__restrict__ x, long R, int K, float *__restrict__ v,
                                                       unsigned char *__restrict__ idx)
{
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float *row = x + (size_t)r * K;
    float best = row[0];
    int bi = 0;
    bool nan = best != best;
    if ((K & 3) == 0) {
        for (int k = 0; k < K; k += 4) {
            const float4 q = *(const float4 *)(row + k);
            const float e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool take = !nan && (e[u] > best || e[u] != e[u]);      // a NaN wins once and stays (torch.max propagates it)
                best = take ? e[u] : best;
                bi = take ? k + u : bi;
                nan = nan || e[u] != e[u];
            }
        }
    } else {
        for (int k = 1; k < K; k++) {
            const float e = row[k];
            const bool take = !nan && (e > best || e != e);
            best = take ? e : best;
            bi = take ? k : bi;
            nan = nan || e != e;
        }
    }
    v[r] = best;
    idx[r] = (unsigned char)bi;
}

// K % 4 == 0: a thread per float4 of gx (consecutive lanes write consecutive 16 bytes: whole cache lines per wave store; a thread
// per 80-byte row wrote 40 partial lines per store instruction and ran at 1 TB/s); the row's g and idx come through the cache
template <bool VEC>
__global__ __launch_bounds__(256) void max_last_backward_kernel(const float *__restrict__ g, const unsigned char *__restrict__ idx,
                                                                long R, int K, float *__restrict__ gx)
{
    if (VEC) {
        const long f = (long)blockIdx.x * 256 + threadIdx.x, q = K >> 2;
        if (f >= R * q) return;
        const long r = f / q;
        const int k = (int)(f - r * q) << 2, bi = idx[r];
        const float gr = g[r];
        ((float4 *)gx)[f] = make_float4(bi == k ? gr : 0.f, bi == k + 1 ? gr : 0.f, bi == k + 2 ? gr : 0.f, bi == k + 3 ? gr : 0.f);
    } else {
        const long r = (long)blockIdx.x * 256 + threadIdx.x;
        if (r >= R) return;
        const float gr = g[r];
        const int bi = idx[r];
        float *row = gx + (size_t)r * K;
        for (int k = 0; k < K; k++) row[k] = bi == k ? gr : 0.f;
    }
}

// v [R] = max_k x [R][K], idx [R] = the first k that attains it (one byte: K <= 256); x 16-byte aligned
extern "C" int l3d_max_last(const float *x, long R, int K, float *v, unsigned char *idx, l3d_stream_t stream)
{
    L3D_REQUIRE(x && v && idx && R > 0 && K > 0);
    if (K > 256 || (((size_t)x) & 15) || l3d_divup(R, 256) > 0x7fffffffL) return L3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(max_last_kernel, dim3((unsigned)l3d_divup(R, 256)), dim3(256), 0, (hipStream_t)stream, x, R, K, v, idx);
    return l3d_check_launch();
}

// gx [R][K] = g [R] at idx [R], zero elsewhere
extern "C" int l3d_max_last_backward(const float *g, const unsigned char *idx, long R, int K, float *gx, l3d_stream_t stream)
{
    L3D_REQUIRE(g && idx && gx && R > 0 && K > 0);
    if (K > 256 || (((size_t)gx) & 15) || l3d_divup(R * (long)((K + 3) / 4), 256) > 0x7fffffffL) return L3D_ERR_UNSUPPORTED;
    if (K % 4 == 0)
        hipLaunchKernelGGL(max_last_backward_kernel<true>, dim3((unsigned)l3d_divup(R * (K / 4), 256)), dim3(256), 0, (hipStream_t)stream, g,
                           idx, R, K, gx);
    else
        hipLaunchKernelGGL(max_last_backward_kernel<false>, dim3((unsigned)l3d_divup(R, 256)), dim3(256), 0, (hipStream_t)stream, g, idx, R,
                           K, gx);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Per-channel finalisation of a layer's statistics, forward and backward, one launch each: what models/_train.py did with
// ~25 (forward) and ~8 (backward) scalar-sized torch operations per layer -- a FlowNet3D training step has 35 such layers and
// spent a fifth of its launches on them.  Everything in fp64, clouds added in cloud order (as l3d_sum_clouds_f64: the same bits
// for any sharding of the batch).
//   mode 0  train-mode BatchNorm: mean / biased variance from part [B][C][2] = per-cloud (sum z, sum z^2) of the conv output
//           WITHOUT its bias; running statistics updated as torch.nn.BatchNorm does (mean of z + bias, unbiased variance,
//           momentum `mom`) when running_mean is given;
//   mode 1  eval-mode BatchNorm: the running statistics (in z-space the mean is running_mean - bias);
//   mode 2  no BatchNorm: y = act(z + bias).
// Outputs: mean64, rstd64, gr64 = gamma rstd (fp64, the backward's per-channel constants), scale = (float)gr,
// shift = (float)(beta - mean gr): y = act(z scale + shift).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double *__restrict__ part, int B, int C, double n, const float *__restrict__ bias,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta, double eps,
                                                          int mode, double mom, float *__restrict__ running_mean,
                                                          float *__restrict__ running_var, double *__restrict__ mean64,
                                                          double *__restrict__ rstd64, double *__restrict__ gr64,
                                                          float *__restrict__ scale, float *__restrict__ shift)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double b = bias ? (double)bias[c] : 0.0;
    // (initialised here and overwritten by modes 0 / 1: with the three-way if / else if / else of the first version hipcc 7.2 left
    // `mean` undefined on the mode-2 path -- the negation of b had been sunk into a block that path skips)
    double mean = -b, rstd = 1.0;
    if (mode == 0) {
        double t0 = 0.0, t1 = 0.0;
        for (int k = 0; k < B; k++) { t0 += part[((size_t)k * C + c) * 2]; t1 += part[((size_t)k * C + c) * 2 + 1]; }
        mean = t0 / n;
        double var = t1 / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        if (running_mean) {
            const double unbiased = var * (n / (n - 1.0 > 1.0 ? n - 1.0 : 1.0));
            const float m = (float)mom, keep = (float)(1.0 - mom);
            running_mean[c] = running_mean[c] * keep + m * (float)(mean + b);
            running_var[c] = running_var[c] * keep + m * (float)unbiased;
        }
        rstd = 1.0 / sqrt(var + eps);
    }
    if (mode == 1) {
        mean = (double)running_mean[c] - b;
        rstd = 1.0 / sqrt((double)running_var[c] + eps);
    }
    const double g = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0, gr = g * rstd;
    mean64[c] = mean; rstd64[c] = rstd; gr64[c] = gr;
    scale[c] = (float)gr;
    shift[c] = (float)(be - mean * gr);
}

extern "C" int l3d_bn_finalize(const double *part, int B, int C, double n, const float *bias, const float *gamma, const float *beta,
                               double eps, int mode, double momentum, float *running_mean, float *running_var, double *mean64,
                               double *rstd64, double *gr64, float *scale, float *shift, l3d_stream_t stream)
{
    L3D_REQUIRE(C > 0 && mean64 && rstd64 && gr64 && scale && shift && mode >= 0 && mode <= 2);
    L3D_REQUIRE(mode != 0 || (part && B > 0 && n > 0.0));
    L3D_REQUIRE(mode != 1 || (running_mean && running_var));
    L3D_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)l3d_divup(C, 256)), dim3(256), 0, (hipStream_t)stream, part, B, C, n, bias, gamma,
                       beta, eps, mode, momentum, running_mean, running_var, mean64, rstd64, gr64, scale, shift);
    return l3d_check_launch();
}

// backward: part_local [Bl][C][2] = this rank's per-cloud (sum g, sum g zhat), part_all [Ba][C][2] = every rank's (or NULL: the
// local ones) ->  m1, m2 = the batch means the BatchNorm backward subtracts (zeros without batch statistics), and the
// parameter gradients of THIS rank's shard: dbeta = sum g, dgamma = sum g zhat, dbias = 0 with batch statistics (a bias in front
// of BatchNorm cancels) else gr * sum g.
__global__ __launch_bounds__(256) void bn_backward_finalize_kernel(const double *__restrict__ part_local, int Bl,
                                                                   const double *__restrict__ part_all, int Ba, int C, double n,
                                                                   int batch_stats, const double *__restrict__ gr64,
                                                                   double *__restrict__ m1, double *__restrict__ m2,
                                                                   float *__restrict__ dbias, float *__restrict__ dgamma,
                                                                   float *__restrict__ dbeta)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double l0 = 0.0, l1 = 0.0;
    for (int k = 0; k < Bl; k++) { l0 += part_local[((size_t)k * C + c) * 2]; l1 += part_local[((size_t)k * C + c) * 2 + 1]; }
    double t0 = l0, t1 = l1;
    if (part_all) {
        t0 = t1 = 0.0;
        for (int k = 0; k < Ba; k++) { t0 += part_all[((size_t)k * C + c) * 2]; t1 += part_all[((size_t)k * C + c) * 2 + 1]; }
    }
    m1[c] = batch_stats ? t0 / n : 0.0;
    m2[c] = batch_stats ? t1 / n : 0.0;
    if (dbias) dbias[c] = batch_stats ? 0.f : (float)(gr64[c] * l0);
    if (dgamma) dgamma[c] = (float)l1;
    if (dbeta) dbeta[c] = (float)l0;
}

extern "C" int l3d_bn_backward_finalize(const double *part_local, int Bl, const double *part_all, int Ba, int C, double n,
                                        int batch_stats, const double *gr64, double *m1, double *m2, float *dbias, float *dgamma,
                                        float *dbeta, l3d_stream_t stream)
{
    L3D_REQUIRE(part_local && Bl > 0 && C > 0 && gr64 && m1 && m2 && (!part_all || Ba > 0) && (!batch_stats || n > 0.0));
    hipLaunchKernelGGL(bn_backward_finalize_kernel, dim3((unsigned)l3d_divup(C, 256)), dim3(256), 0, (hipStream_t)stream, part_local, Bl,
                       part_all, Ba, C, n, batch_stats, gr64, m1, m2, dbias, dgamma, dbeta);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Backward of the pointer network's LayerNorm (utils/transformer.py:109-119: y = a (x - mean) / (std + eps) + b with the UNBIASED
// std and eps added to std) -- the torch composition is ~8 launches forward and ~20 backward over [rows, C]; forward is
// l3d_layernorm_planes (softcorr.hip), this is its backward in one pass over x and g:
//   xc = x - mean, s = std, d = s + eps, dz = g a
//   dxc_i = dz_i / d - (sum_j dz_j xc_j) xc_i / (d^2 (C-1) s),   dx_i = dxc_i - mean_j dxc_j
//   da_c = sum_rows g z,  db_c = sum_rows g          (z = xc / d)
// A wave per row (the row in registers), a workgroup's rows strided by the grid; every wave keeps da / db partial sums for its
// lane's channels, the four waves of a workgroup are added in wave order through LDS, the workgroups' partials
// [G][2][C] in workgroup order (fp64) by a second kernel: deterministic, no atomics.
// ---------------------------------------------------------------------------------------------
template <int VPL /* float4 per lane */>
__global__ __launch_bounds__(256) void layernorm_ref_backward_kernel(const float *__restrict__ x, const float *__restrict__ a,
                                                                     const float *__restrict__ g, float eps, long rows, int C,
                                                                     float *__restrict__ dx, float *__restrict__ partial)
{
    extern __shared__ float ln_red[];                 // [3 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = C >> 2;
    float4 ga[VPL], sda[VPL], sdb[VPL];
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int q = lane + 64 * i;
        ga[i] = q < c4 ? ((const float4 *)a)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        sda[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        sdb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float4 *xr = (const float4 *)(x + row * C), *gr = (const float4 *)(g + row * C);
        float4 v[VPL], gv[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int q = lane + 64 * i;
            v[i] = q < c4 ? xr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            gv[i] = q < c4 ? gr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float mean = s / (float)C;
        float ss = 0.f, t1 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            if (lane + 64 * i < c4) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;                 // xc
                ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
                t1 += (gv[i].x * ga[i].x * v[i].x + gv[i].y * ga[i].y * v[i].y) + (gv[i].z * ga[i].z * v[i].z + gv[i].w * ga[i].w * v[i].w);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { ss += __shfl_xor(ss, off, 64); t1 += __shfl_xor(t1, off, 64); }
        const float sd = sqrtf(ss / (float)(C - 1)), d = sd + eps, inv = 1.f / d;
        const float k2 = sd > 0.f ? t1 * inv * inv / ((float)(C - 1) * sd) : 0.f;
        float t2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            if (lane + 64 * i < c4) {
                // da / db partial sums first (they need g and z), then g is overwritten by dxc
                sda[i].x += gv[i].x * (v[i].x * inv); sda[i].y += gv[i].y * (v[i].y * inv);
                sda[i].z += gv[i].z * (v[i].z * inv); sda[i].w += gv[i].w * (v[i].w * inv);
                sdb[i].x += gv[i].x; sdb[i].y += gv[i].y; sdb[i].z += gv[i].z; sdb[i].w += gv[i].w;
                gv[i].x = gv[i].x * ga[i].x * inv - k2 * v[i].x; gv[i].y = gv[i].y * ga[i].y * inv - k2 * v[i].y;
                gv[i].z = gv[i].z * ga[i].z * inv - k2 * v[i].z; gv[i].w = gv[i].w * ga[i].w * inv - k2 * v[i].w;
                t2 += (gv[i].x + gv[i].y) + (gv[i].z + gv[i].w);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t2 += __shfl_xor(t2, off, 64);
        const float m2 = t2 / (float)C;
        float4 *dr = (float4 *)(dx + row * C);
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int q = lane + 64 * i;
            if (q < c4) dr[q] = make_float4(gv[i].x - m2, gv[i].y - m2, gv[i].z - m2, gv[i].w - m2);
        }
    }
    // the workgroup's partial: waves 1..3 park theirs in LDS, wave 0 adds them in wave order and writes [2][C]
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int q = lane + 64 * i;
            if (q < c4) {
                ((float4 *)(ln_red + ((size_t)(wave - 1) * 2) * C))[q] = sda[i];
                ((float4 *)(ln_red + ((size_t)(wave - 1) * 2 + 1) * C))[q] = sdb[i];
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
        float *out = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int q = lane + 64 * i;
            if (q < c4) {
                float4 da = sda[i], db = sdb[i];
                for (int w = 0; w < 3; w++) {
                    const float4 pa = ((const float4 *)(ln_red + ((size_t)w * 2) * C))[q], pb = ((const float4 *)(ln_red + ((size_t)w * 2 + 1) * C))[q];
                    da.x += pa.x; da.y += pa.y; da.z += pa.z; da.w += pa.w;
                    db.x += pb.x; db.y += pb.y; db.z += pb.z; db.w += pb.w;
                }
                ((float4 *)out)[q] = da;
                ((float4 *)(out + C))[q] = db;
            }
        }
    }
}

// 64 columns of the [G][2 C] partials per workgroup; thread (column, part) adds workgroups part, part + 4, ... (eight loads in
// flight -- one thread per column walked 256 dependent loads, 63 us), the four parts are added in order through LDS: a fixed order
__global__ __launch_bounds__(256) void layernorm_param_reduce_kernel(const float *__restrict__ partial, int G, int C, float *__restrict__ da,
                                                                     float *__restrict__ db)
{
    __shared__ double red[3][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    double s = 0.0;
    if (c < 2 * C) {
        int k = part;
        for (; k + 28 < G; k += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = partial[(size_t)(k + 4 * u) * 2 * C + c];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; k < G; k += 4) s += partial[(size_t)k * 2 * C + c];
    }
    if (part > 0) red[part - 1][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && c < 2 * C) {
        s = ((s + red[0][threadIdx.x]) + red[1][threadIdx.x]) + red[2][threadIdx.x];
        if (c < C) da[c] = (float)s;
        else db[c - C] = (float)s;
    }
}

static int ln_bwd_groups(long rows) { return (int)(rows + 3) / 4 < 256 ? (int)((rows + 3) / 4) : 256; }

extern "C" size_t l3d_layernorm_backward_workspace_floats(long rows, int C) { return (size_t)ln_bwd_groups(rows) * 2 * C; }

// x, g [rows][C]; a [C]; dx [rows][C]; da, db [C]; workspace: l3d_layernorm_backward_workspace_floats(rows, C) floats
extern "C" int l3d_layernorm_ref_backward(const float *x, const float *a, const float *g, float eps, long rows, int C, float *dx,
                                          float *workspace, float *da, float *db, l3d_stream_t stream)
{
    L3D_REQUIRE(x && a && g && dx && workspace && da && db && rows > 0 && C > 1);
    if (C % 4 || C > 2048 || ((((size_t)x) | ((size_t)g) | ((size_t)dx) | ((size_t)a) | ((size_t)workspace)) & 15)) return L3D_ERR_UNSUPPORTED;
    const int G = ln_bwd_groups(rows);
    dim3 grid((unsigned)G), block(256);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)3 * 2 * C * sizeof(float);
    const int vpl = (C / 4 + 63) / 64;
    if (vpl <= 1)      hipLaunchKernelGGL(layernorm_ref_backward_kernel<1>, grid, block, lds, st, x, a, g, eps, rows, C, dx, workspace);
    else if (vpl <= 2) hipLaunchKernelGGL(layernorm_ref_backward_kernel<2>, grid, block, lds, st, x, a, g, eps, rows, C, dx, workspace);
    else if (vpl <= 4) hipLaunchKernelGGL(layernorm_ref_backward_kernel<4>, grid, block, lds, st, x, a, g, eps, rows, C, dx, workspace);
    else               hipLaunchKernelGGL(layernorm_ref_backward_kernel<8>, grid, block, lds, st, x, a, g, eps, rows, C, dx, workspace);
    int rc = l3d_check_launch();
    if (rc != L3D_OK) return rc;
    hipLaunchKernelGGL(layernorm_param_reduce_kernel, dim3((unsigned)l3d_divup(2 * C, 64)), dim3(256), 0, st, workspace, G, C, da, db);
    return l3d_check_launch();
}
