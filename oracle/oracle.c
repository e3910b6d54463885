/*
 * oracle.c -- CPU restatement of learning3d's point-cloud hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under learning3d_amd/ may import, link or
 * call this file; it exists so tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check the HIP kernels against an independent, scalar,
 * easy-to-read statement of what the reference computes.
 *
 * Parity pinning: every function here is checked (tests/test_oracle_golden.py)
 * against golden vectors produced by importing the real reference
 * (/root/reference, via tests/golden/make_golden.py) -- the reference ships no
 * tests or known-answer vectors of its own.
 *
 * All arithmetic is IEEE fp32 with NO implicit contraction: build with
 * -ffp-contract=off.  Where the reference's result depends on a fused
 * multiply-add (the MKL sgemm dot product inside torch.matmul) the fmaf() is
 * written out explicitly.
 *
 * Citations are relative to /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* helpers                                                                   */
/* ------------------------------------------------------------------------- */

/* K=3 dot product as torch.matmul (MKL sgemm) evaluates it on x86:
 * fma(a2,b2, fma(a1,b1, rn(a0*b0))).  SURVEY.md 8(c) "Third-party arithmetic". */
static inline float dot3_mkl(const float *a, const float *b)
{
    return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
}

/* torch.sum(x**2, dim) over a length-3 axis: sequentially rounded squares. */
static inline float sumsq3(const float *a)
{
    return (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
}

/* ------------------------------------------------------------------------- */
/* a1: knn  -- utils/model_common_utils.py:3-9                               */
/*   inner = -2 * x^T x ; xx = sum(x**2) ;                                   */
/*   pd[i][j] = (-xx[j] - inner[i][j]) - xx[i] ; idx = topk(pd, k) (largest) */
/* xyz is [B,N,3] (the transpose of the reference's [B,3,N] argument).       */
/* Ties: the reference's topk order under exact ties is unspecified; here    */
/* the lower index wins (documented contract).          */
/* ------------------------------------------------------------------------- */
void orc_knn(const float *xyz, int B, int N, int k, int64_t *idx, float *pd_out)
{
    float *xx = (float *)malloc(sizeof(float) * N);
    float *pd = (float *)malloc(sizeof(float) * N);
    unsigned char *used = (unsigned char *)malloc(N);
    for (int b = 0; b < B; b++) {
        const float *p = xyz + (size_t)b * N * 3;
        for (int j = 0; j < N; j++) xx[j] = sumsq3(p + 3 * j);
        for (int i = 0; i < N; i++) {
            for (int j = 0; j < N; j++) {
                float inner = -2.0f * dot3_mkl(p + 3 * i, p + 3 * j);
                float t = (-xx[j]) - inner;
                pd[j] = t - xx[i];
            }
            memset(used, 0, N);
            for (int s = 0; s < k; s++) {
                int best = -1;
                for (int j = 0; j < N; j++) {
                    if (used[j]) continue;
                    if (best < 0 || pd[j] > pd[best]) best = j;
                }
                used[best] = 1;
                idx[((size_t)b * N + i) * k + s] = best;
                if (pd_out) pd_out[((size_t)b * N + i) * k + s] = pd[best];
            }
        }
    }
    free(xx); free(pd); free(used);
}

/* full [B,N,N] matrix of the same pd values (tests use it to reason about ties) */
void orc_knn_pd(const float *xyz, int B, int N, float *pd)
{
    for (int b = 0; b < B; b++) {
        const float *p = xyz + (size_t)b * N * 3;
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                float inner = -2.0f * dot3_mkl(p + 3 * i, p + 3 * j);
                float t = (-sumsq3(p + 3 * j)) - inner;
                pd[((size_t)b * N + i) * N + j] = t - sumsq3(p + 3 * i);
            }
    }
}

/* ------------------------------------------------------------------------- */
/* a3: square_distance -- utils/model_common_utils.py:19-38                  */
/*   dist = -2*src.dst^T ; dist += sum(src^2)[:, :, None] ; dist += sum(dst^2)*/
/* ------------------------------------------------------------------------- */
static inline float sqdist_expanded(const float *s, const float *d)
{
    float v = -2.0f * dot3_mkl(s, d);
    v = v + sumsq3(s);
    v = v + sumsq3(d);
    return v;
}

void orc_square_distance(const float *src, const float *dst, int B, int N, int M, float *out)
{
    for (int b = 0; b < B; b++)
        for (int i = 0; i < N; i++)
            for (int j = 0; j < M; j++)
                out[((size_t)b * N + i) * M + j] =
                    sqdist_expanded(src + ((size_t)b * N + i) * 3, dst + ((size_t)b * M + j) * 3);
}

/* ------------------------------------------------------------------------- */
/* a4: query_ball_point -- utils/model_common_utils.py:102-130               */
/*   keep indices (ascending) with expanded d2 <= r^2 (":117" masks d2 > r^2) */
/*   take the first nsample, pad with the first; an empty ball yields N.     */
/*   cnt (get_cnt=True) is the un-truncated number of hits.                  */
/* ------------------------------------------------------------------------- */
void orc_query_ball_point(float radius, int nsample, const float *xyz, const float *new_xyz,
                          int B, int N, int S, int64_t *idx, int64_t *cnt)
{
    /* python: radius ** 2 is a double; the comparison promotes the fp32 tensor
     * element against a python scalar -> torch compares in fp32 after casting
     * the scalar to fp32. */
    const float r2 = (float)((double)radius * (double)radius);
    for (int b = 0; b < B; b++)
        for (int s = 0; s < S; s++) {
            const float *q = new_xyz + ((size_t)b * S + s) * 3;
            int64_t *o = idx + ((size_t)b * S + s) * nsample;
            int c = 0; int64_t total = 0;
            for (int j = 0; j < N; j++) {
                float d2 = sqdist_expanded(q, xyz + ((size_t)b * N + j) * 3);
                if (!(d2 > r2)) {
                    if (c < nsample) o[c++] = j;
                    total++;
                }
            }
            int64_t first = c ? o[0] : (int64_t)N;
            for (; c < nsample; c++) o[c] = first;
            if (cnt) cnt[(size_t)b * S + s] = total;
        }
}

/* ------------------------------------------------------------------------- */
/* a6: farthest_point_sample(start_with_first_point=True)                    */
/*     utils/model_common_utils.py:58-82                                     */
/*   dist = sum((xyz - centroid)**2, -1) ; distance = min(distance, dist) ;  */
/*   farthest = argmax(distance) (first maximal index)                       */
/* ------------------------------------------------------------------------- */
void orc_farthest_point_sample(const float *xyz, int B, int N, int npoint, int64_t *out)
{
    float *dist = (float *)malloc(sizeof(float) * N);
    for (int b = 0; b < B; b++) {
        const float *p = xyz + (size_t)b * N * 3;
        for (int j = 0; j < N; j++) dist[j] = 1e10f;
        int far = 0;
        for (int i = 0; i < npoint; i++) {
            out[(size_t)b * npoint + i] = far;
            const float *c = p + 3 * far;
            int best = 0;
            for (int j = 0; j < N; j++) {
                float dx = p[3 * j] - c[0], dy = p[3 * j + 1] - c[1], dz = p[3 * j + 2] - c[2];
                float d = (dx * dx + dy * dy) + dz * dz;
                if (d < dist[j]) dist[j] = d;
                if (dist[j] > dist[best]) best = j;
            }
            far = best;
        }
    }
    free(dist);
}

/* ------------------------------------------------------------------------- */
/* a7: knn_point -- utils/model_common_utils.py:84-100                        */
/*   dist = sum(-(pos1-pos2)**2, -1) ; val,idx = topk(dist,k) ; sqrt(-val)   */
/*   pos1 [B,N,3] is the searched set, pos2 [B,M,3] the queries.             */
/* ------------------------------------------------------------------------- */
void orc_knn_point(int k, const float *pos1, const float *pos2, int B, int N, int M,
                   float *val, int64_t *idx)
{
    float *d = (float *)malloc(sizeof(float) * N);
    unsigned char *used = (unsigned char *)malloc(N);
    for (int b = 0; b < B; b++)
        for (int q = 0; q < M; q++) {
            const float *c = pos2 + ((size_t)b * M + q) * 3;
            for (int j = 0; j < N; j++) {
                const float *p = pos1 + ((size_t)b * N + j) * 3;
                float dx = p[0] - c[0], dy = p[1] - c[1], dz = p[2] - c[2];
                /* sum over the last axis of -(diff**2): ((-dx2) + (-dy2)) + (-dz2) */
                d[j] = ((-(dx * dx)) + (-(dy * dy))) + (-(dz * dz));
            }
            memset(used, 0, N);
            for (int s = 0; s < k; s++) {
                int best = -1;
                for (int j = 0; j < N; j++) {
                    if (used[j]) continue;
                    if (best < 0 || d[j] > d[best]) best = j;
                }
                used[best] = 1;
                idx[((size_t)b * M + q) * k + s] = best;
                val[((size_t)b * M + q) * k + s] = sqrtf(-d[best]);
            }
        }
    free(d); free(used);
}

/* ------------------------------------------------------------------------- */
/* a9 / K1c: Chamfer nearest-neighbour search                                */
/*   losses/cuda/chamfer_distance/chamfer_distance.cpp:59-87 (nnsearch)      */
/*   d = (dx*dx + dy*dy) + dz*dz in fp32, strict '<' -> lowest index on ties */
/* ------------------------------------------------------------------------- */
static void nn_one_direction(int B, int n, int m, const float *a, const float *c,
                             float *dist, int32_t *idx)
{
    for (int b = 0; b < B; b++)
        for (int j = 0; j < n; j++) {
            const float *p = a + ((size_t)b * n + j) * 3;
            float best = 0.f; int besti = 0;
            for (int k = 0; k < m; k++) {
                const float *q = c + ((size_t)b * m + k) * 3;
                float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
                float d = (dx * dx + dy * dy) + dz * dz;
                if (k == 0 || d < best) { best = d; besti = k; }
            }
            dist[(size_t)b * n + j] = best;
            idx[(size_t)b * n + j] = besti;
        }
}

void orc_chamfer_forward(const float *xyz1, const float *xyz2, int B, int N, int M,
                         float *dist1, float *dist2, int32_t *idx1, int32_t *idx2)
{   /* chamfer_distance.cpp:90-111 */
    nn_one_direction(B, N, M, xyz1, xyz2, dist1, idx1);
    nn_one_direction(B, M, N, xyz2, xyz1, dist2, idx2);
}

/* K2c: chamfer_distance.cpp:114-177.  Accumulates in double and rounds once so
 * the oracle is order-independent; the HIP kernel is compared with tolerance
 * (the reference GPU kernel itself uses non-deterministic fp32 atomics). */
void orc_chamfer_backward(const float *xyz1, const float *xyz2, int B, int N, int M,
                          const float *gd1, const float *gd2,
                          const int32_t *idx1, const int32_t *idx2,
                          float *g1, float *g2)
{
    size_t n1 = (size_t)B * N * 3, n2 = (size_t)B * M * 3;
    double *a1 = (double *)calloc(n1, sizeof(double));
    double *a2 = (double *)calloc(n2, sizeof(double));
    for (int b = 0; b < B; b++) {
        for (int j = 0; j < N; j++) {
            size_t p = ((size_t)b * N + j) * 3, q = ((size_t)b * M + idx1[(size_t)b * N + j]) * 3;
            float g = gd1[(size_t)b * N + j] * 2;
            for (int c = 0; c < 3; c++) {
                float v = g * (xyz1[p + c] - xyz2[q + c]);
                a1[p + c] += v; a2[q + c] -= v;
            }
        }
        for (int j = 0; j < M; j++) {
            size_t p = ((size_t)b * M + j) * 3, q = ((size_t)b * N + idx2[(size_t)b * M + j]) * 3;
            float g = gd2[(size_t)b * M + j] * 2;
            for (int c = 0; c < 3; c++) {
                float v = g * (xyz2[p + c] - xyz1[q + c]);
                a2[p + c] += v; a1[q + c] -= v;
            }
        }
    }
    for (size_t i = 0; i < n1; i++) g1[i] = (float)a1[i];
    for (size_t i = 0; i < n2; i++) g2[i] = (float)a2[i];
    free(a1); free(a2);
}

/* ------------------------------------------------------------------------- */
/* K7: ball_query_kernel_fast -- utils/lib/src/ball_query_gpu.cu:9-45        */
/*   direct-difference d2, strict '<', first hit back-fills all slots,       */
/*   idx pre-zeroed by the caller (pointnet2_utils.py:246) -> empty ball = 0 */
/* ------------------------------------------------------------------------- */
void orc_ball_query(int B, int N, int S, float radius, int nsample,
                    const float *new_xyz, const float *xyz, int32_t *idx)
{
    const float r2 = radius * radius;
    for (int b = 0; b < B; b++)
        for (int s = 0; s < S; s++) {
            const float *q = new_xyz + ((size_t)b * S + s) * 3;
            int32_t *o = idx + ((size_t)b * S + s) * nsample;
            for (int l = 0; l < nsample; l++) o[l] = 0;
            int cnt = 0;
            for (int k = 0; k < N && cnt < nsample; k++) {
                const float *p = xyz + ((size_t)b * N + k) * 3;
                float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
                float d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 < r2) {
                    if (cnt == 0) for (int l = 0; l < nsample; l++) o[l] = k;
                    o[cnt++] = k;
                }
            }
        }
}

/* K8: group_points_kernel_fast -- utils/lib/src/group_points_gpu.cu:47-66 */
void orc_group_points(int B, int C, int N, int S, int K, const float *points,
                      const int32_t *idx, float *out)
{
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int s = 0; s < S; s++)
                for (int k = 0; k < K; k++)
                    out[(((size_t)b * C + c) * S + s) * K + k] =
                        points[((size_t)b * C + c) * N + idx[((size_t)b * S + s) * K + k]];
}

/* K9: group_points_grad_kernel_fast -- group_points_gpu.cu:8-25 (double accum) */
void orc_group_points_grad(int B, int C, int N, int S, int K, const float *grad_out,
                           const int32_t *idx, float *grad_points)
{
    size_t n = (size_t)B * C * N;
    double *acc = (double *)calloc(n, sizeof(double));
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int s = 0; s < S; s++)
                for (int k = 0; k < K; k++)
                    acc[((size_t)b * C + c) * N + idx[((size_t)b * S + s) * K + k]] +=
                        grad_out[(((size_t)b * C + c) * S + s) * K + k];
    for (size_t i = 0; i < n; i++) grad_points[i] = (float)acc[i];
    free(acc);
}

/* K10: gather_points_kernel_fast -- utils/lib/src/sampling_gpu.cu:8-24 */
void orc_gather_points(int B, int C, int N, int S, const float *points,
                       const int32_t *idx, float *out)
{
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int s = 0; s < S; s++)
                out[((size_t)b * C + c) * S + s] =
                    points[((size_t)b * C + c) * N + idx[(size_t)b * S + s]];
}

/* K11: gather_points_grad_kernel_fast -- sampling_gpu.cu:46-63 */
void orc_gather_points_grad(int B, int C, int N, int S, const float *grad_out,
                            const int32_t *idx, float *grad_points)
{
    size_t n = (size_t)B * C * N;
    double *acc = (double *)calloc(n, sizeof(double));
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int s = 0; s < S; s++)
                acc[((size_t)b * C + c) * N + idx[(size_t)b * S + s]] +=
                    grad_out[((size_t)b * C + c) * S + s];
    for (size_t i = 0; i < n; i++) grad_points[i] = (float)acc[i];
    free(acc);
}

/* ------------------------------------------------------------------------- */
/* K12: furthest_point_sampling_kernel -- sampling_gpu.cu:93-209             */
/*   start at 0; temp[k] = min(d, temp[k]); arg-max with '>'.                */
/*   TIES (round 4: the restatement now follows the kernel here too; before  */
/*   it took the lowest index, which the reference does only for tie-free    */
/*   data): thread tid scans k = tid, tid + T, ... with a strict '>' (lowest */
/*   k of the thread wins); the tree merges slot s with slot s + h for       */
/*   h = T/2, T/4, ... 1 and keeps i1 unless v2 > v1 (:86-91).  Two tied     */
/*   candidates meet at the level h where their thread ids first agree       */
/*   modulo h, and the one whose bit h is CLEAR survives: the winner is the  */
/*   smallest (bit-reversed (k mod T), k), T = opt_n_threads(N) = the        */
/*   largest power of two <= N, at most 1024 (cuda_utils.h:6-14); checked    */
/*   against a thread-by-thread emulation of the kernel in                   */
/*   tests/test_oracle_golden.py.  Clouds clipped to                         */
/*   a box (config 5: N(0,1) clipped to [-2,2]) put duplicate points on its  */
/*   corners, the first places FPS visits: 32 such clouds of 8192 points     */
/*   differ from the lowest-index rule in about two clouds out of three.     */
/*   temp is caller-initialised to 1e10 (pointnet2_utils.py:26).             */
/* ------------------------------------------------------------------------- */
static int orc_brev(int x, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

void orc_furthest_point_sampling(int B, int N, int S, const float *xyz, float *temp, int32_t *idxs)
{
    int pow_2 = (int)(log((double)N) / log(2.0));                /* cuda_utils.h:11-13 */
    int T = 1 << pow_2;
    if (T > 1024) T = 1024;
    if (T < 1) T = 1;
    int bits = 0;
    while ((1 << bits) < T) bits++;
    for (int b = 0; b < B; b++) {
        const float *p = xyz + (size_t)b * N * 3;
        float *t = temp + (size_t)b * N;
        int32_t *o = idxs + (size_t)b * S;
        if (S <= 0) continue;
        int old = 0; o[0] = 0;
        for (int j = 1; j < S; j++) {
            int besti = 0; float best = -1.f;
            float x1 = p[old * 3], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
            for (int k = 0; k < N; k++) {
                float dx = p[k * 3] - x1, dy = p[k * 3 + 1] - y1, dz = p[k * 3 + 2] - z1;
                float d = (dx * dx + dy * dy) + dz * dz;
                float d2 = d < t[k] ? d : t[k];
                t[k] = d2;
                if (d2 > best || (d2 == best && orc_brev(k % T, bits) < orc_brev(besti % T, bits))) { best = d2; besti = k; }
            }
            old = besti; o[j] = old;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* K13: knn_kernel_fast -- utils/lib/src/interpolate_gpu.cu:9-57             */
/*   insertion sort, strict '<' -> ascending d2, earlier index first on ties */
/* ------------------------------------------------------------------------- */
void orc_knn_pair(int B, int N, int M, int k, const float *unknown, const float *known,
                  float *dist2, int32_t *idx)
{
    double *best = (double *)malloc(sizeof(double) * k);
    int *besti = (int *)malloc(sizeof(int) * k);
    for (int b = 0; b < B; b++)
        for (int q = 0; q < N; q++) {
            const float *u = unknown + ((size_t)b * N + q) * 3;
            for (int i = 0; i < k; i++) { best[i] = 1e40; besti[i] = 0; }
            for (int i = 0; i < M; i++) {
                const float *p = known + ((size_t)b * M + i) * 3;
                float dx = u[0] - p[0], dy = u[1] - p[1], dz = u[2] - p[2];
                float d = (dx * dx + dy * dy) + dz * dz;
                for (int j = 0; j < k; j++)
                    if (d < best[j]) {
                        for (int l = k - 1; l > j; l--) { best[l] = best[l - 1]; besti[l] = besti[l - 1]; }
                        best[j] = d; besti[j] = i;
                        break;
                    }
            }
            for (int i = 0; i < k; i++) {
                idx[((size_t)b * N + q) * k + i] = besti[i];
                dist2[((size_t)b * N + q) * k + i] = (float)best[i];   /* 1e40 -> +inf */
            }
        }
    free(best); free(besti);
}

/* K14: three_nn_kernel_fast -- interpolate_gpu.cu:81-124 (= K13 with k=3) */
void orc_three_nn(int B, int N, int M, const float *unknown, const float *known,
                  float *dist2, int32_t *idx)
{
    orc_knn_pair(B, N, M, 3, unknown, known, dist2, idx);
}

/* K15: three_interpolate_kernel_fast -- interpolate_gpu.cu:149-169 */
void orc_three_interpolate(int B, int C, int M, int N, const float *points,
                           const int32_t *idx, const float *weight, float *out)
{
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int n = 0; n < N; n++) {
                const float *w = weight + ((size_t)b * N + n) * 3;
                const int32_t *ix = idx + ((size_t)b * N + n) * 3;
                const float *p = points + ((size_t)b * C + c) * M;
                out[((size_t)b * C + c) * N + n] = (w[0] * p[ix[0]] + w[1] * p[ix[1]]) + w[2] * p[ix[2]];
            }
}

/* K16: three_interpolate_grad_kernel_fast -- interpolate_gpu.cu:192-214 */
void orc_three_interpolate_grad(int B, int C, int N, int M, const float *grad_out,
                                const int32_t *idx, const float *weight, float *grad_points)
{
    size_t tot = (size_t)B * C * M;
    double *acc = (double *)calloc(tot, sizeof(double));
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int n = 0; n < N; n++) {
                const float *w = weight + ((size_t)b * N + n) * 3;
                const int32_t *ix = idx + ((size_t)b * N + n) * 3;
                float g = grad_out[((size_t)b * C + c) * N + n];
                for (int t = 0; t < 3; t++) acc[((size_t)b * C + c) * M + ix[t]] += g * w[t];
            }
    for (size_t i = 0; i < tot; i++) grad_points[i] = (float)acc[i];
    free(acc);
}

/* ------------------------------------------------------------------------- */
/* K3: approxmatch -- losses/cuda/emd_torch/pkg/include/cuda/emd.cuh:7-185   */
/* K4: matchcost   -- emd.cuh:202-244                                        */
/* K5/K6: matchcostgrad1/2 -- emd.cuh:259-323                                */
/*   CUDA-only in the reference (no CPU twin, does not build on torch 2.x):  */
/*   restated from the kernel text.  __expf / rsqrtf are fast-math there, so */
/*   parity for EMD is 1e-4 relative, not bit-exact.        */
/*   match is indexed [b][l*n + k]  (l over xyz2's m points, k over xyz1's   */
/*   n points) exactly as emd.cuh:158 does.                                  */
/* ------------------------------------------------------------------------- */
void orc_emd_approxmatch(int B, int n, int m, const float *xyz1, const float *xyz2, float *match)
{
    float *remainL = (float *)malloc(sizeof(float) * n), *remainR = (float *)malloc(sizeof(float) * m);
    float *ratioL = (float *)malloc(sizeof(float) * n), *ratioR = (float *)malloc(sizeof(float) * m);
    float multiL, multiR;
    if (n >= m) { multiL = 1; multiR = (float)(n / m); } else { multiL = (float)(m / n); multiR = 1; }
    for (int b = 0; b < B; b++) {
        const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
        float *mt = match + (size_t)b * n * m;
        for (size_t j = 0; j < (size_t)n * m; j++) mt[j] = 0;
        for (int j = 0; j < n; j++) remainL[j] = multiL;
        for (int j = 0; j < m; j++) remainR[j] = multiR;
        for (int j = 7; j >= -2; j--) {
            float level = -powf(4.0f, (float)j);
            if (j == -2) level = 0;
            /* ratioL[k] = remainL[k] / sum_l exp(level*d2)*remainR[l]   (emd.cuh:33-70) */
            for (int k = 0; k < n; k++) {
                float suml = 1e-9f;
                for (int l = 0; l < m; l++) {
                    float dx = p2[l * 3] - p1[k * 3], dy = p2[l * 3 + 1] - p1[k * 3 + 1], dz = p2[l * 3 + 2] - p1[k * 3 + 2];
                    float d = level * ((dx * dx + dy * dy) + dz * dz);
                    suml += expf(d) * remainR[l];
                }
                ratioL[k] = remainL[k] / suml;
            }
            /* ratioR[l]: consumption on the right side (emd.cuh:72-118) */
            for (int l = 0; l < m; l++) {
                float sumr = 0;
                for (int k = 0; k < n; k++) {
                    float dx = p2[l * 3] - p1[k * 3], dy = p2[l * 3 + 1] - p1[k * 3 + 1], dz = p2[l * 3 + 2] - p1[k * 3 + 2];
                    float d = level * ((dx * dx + dy * dy) + dz * dz);
                    sumr += expf(d) * ratioL[k];
                }
                sumr *= remainR[l];
                float consumption = fminf(remainR[l] / (sumr + 1e-9f), 1.0f);
                ratioR[l] = consumption * remainR[l];
                remainR[l] = fmaxf(0.0f, remainR[l] - sumr);
            }
            /* match += w ; remainL update (emd.cuh:120-180) */
            for (int k = 0; k < n; k++) {
                float suml = 0;
                for (int l = 0; l < m; l++) {
                    float dx = p2[l * 3] - p1[k * 3], dy = p2[l * 3 + 1] - p1[k * 3 + 1], dz = p2[l * 3 + 2] - p1[k * 3 + 2];
                    float d = level * ((dx * dx + dy * dy) + dz * dz);
                    float w = expf(d) * ratioL[k] * ratioR[l];
                    mt[(size_t)l * n + k] += w;
                    suml += w;
                }
                remainL[k] = fmaxf(0.0f, remainL[k] - suml);
            }
        }
    }
    free(remainL); free(remainR); free(ratioL); free(ratioR);
}

void orc_emd_matchcost(int B, int n, int m, const float *xyz1, const float *xyz2,
                       const float *match, float *cost)
{
    for (int b = 0; b < B; b++) {
        const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
        const float *mt = match + (size_t)b * n * m;
        double s = 0;
        for (int k = 0; k < n; k++)
            for (int l = 0; l < m; l++) {
                float dx = p1[k * 3] - p2[l * 3], dy = p1[k * 3 + 1] - p2[l * 3 + 1], dz = p1[k * 3 + 2] - p2[l * 3 + 2];
                float d = sqrtf((dx * dx + dy * dy) + dz * dz);
                s += (double)(d * mt[(size_t)l * n + k]);
            }
        cost[b] = (float)s;
    }
}

void orc_emd_matchcostgrad(int B, int n, int m, const float *xyz1, const float *xyz2,
                           const float *match, float *grad1, float *grad2)
{
    for (int b = 0; b < B; b++) {
        const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
        const float *mt = match + (size_t)b * n * m;
        for (int k = 0; k < n; k++) {           /* matchcostgrad1: emd.cuh:302-323 */
            double gx = 0, gy = 0, gz = 0;
            for (int l = 0; l < m; l++) {
                float dx = p1[k * 3] - p2[l * 3], dy = p1[k * 3 + 1] - p2[l * 3 + 1], dz = p1[k * 3 + 2] - p2[l * 3 + 2];
                float d = mt[(size_t)l * n + k] / sqrtf(fmaxf((dx * dx + dy * dy) + dz * dz, 1e-20f));
                gx += dx * d; gy += dy * d; gz += dz * d;
            }
            grad1[((size_t)b * n + k) * 3] = (float)gx; grad1[((size_t)b * n + k) * 3 + 1] = (float)gy; grad1[((size_t)b * n + k) * 3 + 2] = (float)gz;
        }
        for (int l = 0; l < m; l++) {           /* matchcostgrad2: emd.cuh:259-299 */
            double gx = 0, gy = 0, gz = 0;
            for (int k = 0; k < n; k++) {
                float dx = p2[l * 3] - p1[k * 3], dy = p2[l * 3 + 1] - p1[k * 3 + 1], dz = p2[l * 3 + 2] - p1[k * 3 + 2];
                float d = mt[(size_t)l * n + k] / sqrtf(fmaxf((dx * dx + dy * dy) + dz * dz, 1e-20f));
                gx += dx * d; gy += dy * d; gz += dz * d;
            }
            grad2[((size_t)b * m + l) * 3] = (float)gx; grad2[((size_t)b * m + l) * 3 + 1] = (float)gy; grad2[((size_t)b * m + l) * 3 + 2] = (float)gz;
        }
    }
}
