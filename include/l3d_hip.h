/*
 * l3d_hip.h -- C ABI of libl3d_hip.so: learning3d's point-cloud hot path as
 * hand-written HIP kernels for AMD MI355X (gfx950 / CDNA4).
 *
 * Conventions (every entry point)
 *   - plain pointers + sizes, no torch / at::Tensor types: all pointers are DEVICE
 *     pointers to dense row-major arrays of the stated shape;
 *   - caller allocates every output (the reference's dominant convention:
 *     losses/cuda/chamfer_distance/chamfer_distance.py:21-25,
 *     utils/lib/pointnet2_utils.py:25-28,246);
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it,
 *     there is no internal synchronisation and no global state, so calls are
 *     safe from several host threads (one per device, as nn.DataParallel does);
 *   - returns L3D_OK (0) or a negative l3d_status; never exit()s, never prints
 *     (the reference printf's and continues, chamfer_distance.cu:152-154, or
 *     exit(-1)s, ball_query_gpu.cu:62-66);
 *   - index outputs are int32 where the reference's native extension returns
 *     IntTensor and int64 where the reference's torch code returns LongTensor.
 *
 * Citations are relative to the reference tree (vinits5/learning3d @ 2025-03-02).
 */
#ifndef L3D_HIP_H_
#define L3D_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *l3d_stream_t; /* hipStream_t */

typedef enum {
    L3D_OK = 0,
    L3D_ERR_INVALID_ARG = -1, /* null pointer, non-positive size, k > supported, ... */
    L3D_ERR_UNSUPPORTED = -2, /* shape outside what the kernels are built for     */
    L3D_ERR_LAUNCH = -3       /* hipGetLastError() != hipSuccess after a launch    */
} l3d_status;

#define L3D_KNN_MAX_K 200 /* same bound as interpolate_gpu.cu:30-31 (best[200]) */

int l3d_version(void);
const char *l3d_status_string(int status);
/* hipError_t recorded by the last failing launch on the calling thread (0 if none) */
int l3d_last_hip_error(void);
/* Measurement aid (bench.py): one launch in which every SIMD issues fp16 MFMAs back to back on random operands -- 256 workgroups x
 * 8 waves x 4 * iters instructions of 32 768 FLOP.  The caller times the launch; ticks[0] / ticks[1] * 100 MHz = the shader clock the
 * chip held (the data-sheet peak assumes 2.4 GHz; under this load it holds ~1.5).  sink: 131 072 floats of scratch. */
int l3d_probe_mfma_sustained(int iters, float *sink, long long *ticks, l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused kNN graph (T1)  == utils/model_common_utils.py:3-9  knn(x, k)
 *   ranks pd[i][j] = (-xx[j] - (-2 x_i.x_j)) - xx[i] exactly as the reference's fp32 ops do
 *   (dot product as an fma chain); writes the k largest, descending;
 *   exact ties resolve to the lower index.  Never materialises [B,N,N].
 *   xyz [B,N,3] fp32, idx [B,N,k] int64.   k <= min(N, L3D_KNN_MAX_K)
 * ------------------------------------------------------------------------------------------- */
int l3d_knn_graph(const float *xyz, int B, int N, int k, int64_t *idx, l3d_stream_t stream);

/* Deterministic backward of the gather-type ops (replaces the fp32-atomicAdd scatters of
 * group_points_grad_kernel, group_points_gpu.cu:8-28; gather_points_grad, sampling_gpu.cu:37-52;
 * three_interpolate_grad, interpolate_gpu.cu:185-205):
 *     dst[b][c][t] = sum over entries e < E with idx[b][e] == t, in ascending e, of
 *                    src[b][c][e / div] * (weight ? weight[b][e] : 1)
 * src fp32 [B][C][E/div]; idx int32 [B][E] with values in [0,T); dst fp32 [B][C][T] (fully written).
 * grouping: E = npoint*nsample, div 1;  gather: E = npoint, div 1;  three_interpolate: E = n*3, div 3, weight [B,n,3].
 * workspace: >= l3d_scatter_add_det_workspace_bytes(B,T,E) bytes.  Same inputs -> same bits, every run. */
size_t l3d_scatter_add_det_workspace_bytes(int B, int T, int E);
int l3d_scatter_add_det(const float *src, const int32_t *idx, const float *weight, int B, int C, int T, int E, int div,
                        void *workspace, float *dst, l3d_stream_t stream);

/* Second half of one dynamic-graph EdgeConv layer (PRNet's DGCNN, models/prnet.py:76-97) on linear
 * pre-activations: the 1x1 conv over (neighbour ; centre) is two per-point products P, Q (one
 * l3d_pointwise_conv with 2*Cout output rows, BN folded), and
 *     out[b][co][i] = act(max_j pq[b][co][idx[b][i][j]] + pq[b][Cout+co][i])
 * equals the reference's max_j act(bn(conv(cat(x_j, x_i)))) up to the rounding of the split sum.
 * pq fp32 [B][2*Cout][N]; idx int64 [B][N][k] (k <= 40); act: activation code as l3d_pointwise_conv's relu;
 * out fp32 [B][Cout][N] with batch stride out_bstride floats (a slice of a concat buffer). */
int l3d_edge_gather_max(const float *pq, const int64_t *idx, int B, int Cout, int N, int k, int act, float *out,
                        long out_bstride, l3d_stream_t stream);

/* knn() of utils/model_common_utils.py:3-9 for FEATURE-space graphs, x [B,C,N] with C % 32 == 0 (PRNet's
 * dynamic DGCNN graphs, models/prnet.py:76-97; C = 3 takes l3d_knn_graph): pd = -xx_j + 2 x_i.x_j - xx_i
 * with the inner product on the matrix cores (bf16x3, fp32-level error) and top-k as the GEMM epilogue, no
 * [B,N,N] tensor.  idx int64 [B,N,k], best first, exact ties -> lower index.  workspace: >=
 * l3d_knn_feature_workspace_bytes(B,C,N) bytes, 16-byte aligned (holds the split copy of x).
 * k > 20 or C % 32 != 0 -> L3D_ERR_UNSUPPORTED; k > N -> L3D_ERR_INVALID_ARG. */
size_t l3d_knn_feature_workspace_bytes(int B, int C, int N);
int l3d_knn_feature(const float *x, int B, int C, int N, int k, void *workspace, int64_t *idx, l3d_stream_t stream);

/* get_graph_feature gather  == utils/model_common_utils.py:141-154
 *   out[b][n][j][0:C] = x[b][idx[b][n][j]][:], out[b][n][j][C:2C] = x[b][n][:]
 *   x [B,N,C], idx [B,N,k] int64, out [B,N,k,2C]  (the reference returns this memory
 *   permuted to [B,2C,N,k]; the host wrapper returns the same strided view). */
int l3d_graph_feature(const float *x, const int64_t *idx, int B, int N, int C, int k, float *out,
                      l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Chamfer  == losses/cuda/chamfer_distance/chamfer_distance.cpp:180-185 (pybind `cd`)
 *   forward : cd.forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2)   (K1, .cu:6-150)
 *   backward: cd.backward_cuda(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
 *             (K2, .cu:158-209) -- deterministic here (gather + segmented sum, no fp32 atomics)
 *   xyz1 [B,N,3], xyz2 [B,M,3], dist1/idx1 [B,N], dist2/idx2 [B,M]; idx int32.
 *   d = (dx*dx + dy*dy) + dz*dz, no contraction; strict '<' => lowest index on ties.  From N * M >= 2^24 pairs per cloud the
 *   candidates are ranked on the fp16 matrix cores and only those inside the ranking's error band are evaluated this way
 *   (chamfer_mfma.hip): the same distances and indices, bit for bit.
 * ------------------------------------------------------------------------------------------- */
int l3d_chamfer_forward(const float *xyz1, const float *xyz2, int B, int N, int M, float *dist1,
                        float *dist2, int32_t *idx1, int32_t *idx2, l3d_stream_t stream);
int l3d_chamfer_backward(const float *xyz1, const float *xyz2, int B, int N, int M,
                         const float *graddist1, const float *graddist2, const int32_t *idx1,
                         const int32_t *idx2, float *gradxyz1, float *gradxyz2, l3d_stream_t stream);
/* Loss tail of losses/chamfer_distance.py:38-40, kept on the device (no host sync per step):
 *   l3d_chamfer_partials: partial[0..3] = (sum sqrt(dist1), sum sqrt(dist2), #dist1, #dist2), fp64,
 *       for this rank's shard -- the 32 bytes the multi-GPU path all-gathers over RCCL;
 *   l3d_chamfer_combine : loss[0] = (S1/N1 + S2/N2) / 2 over `world` gathered partial rows
 *       ([world][4] fp64) -> fp32 scalar, identical on every rank. */
int l3d_chamfer_partials(const float *dist1, const float *dist2, int B, int N, int M, double *partial,
                         l3d_stream_t stream);
int l3d_chamfer_combine(const double *partials, int world, float *loss, l3d_stream_t stream);
/* The same over up to 64 workgroups (the one-workgroup form is a chain of dependent load latencies: 12 us at B=32,
 * N=1024; this one ~4): per-workgroup fp64 partial sums in ws, added in workgroup order by the workgroup that finishes
 * last (deterministic; agrees with l3d_chamfer_loss_local to fp64 rounding of the sum order).  ws:
 * l3d_chamfer_loss_local_ws_bytes() bytes of device memory whose first 8 bytes are zero before the first call (the
 * kernel re-arms them); one ws per stream. */
size_t l3d_chamfer_loss_local_ws_bytes(void);
int l3d_chamfer_loss_local_mb(const float *dist1, const float *dist2, int B, int N, int M, void *ws, double *partial,
                              float *loss, l3d_stream_t stream);
/* Search AND loss tail of ChamferDistanceLoss.forward (losses/chamfer_distance.py:34-43: chamfer_distance() == cd.forward_cuda, then
 * (mean sqrt dist1 + mean sqrt dist2) / 2) in ONE launch where the two-queries-per-lane kernel serves the search (N * M < 2^24 pairs
 * per cloud and >= 256 workgroups: configs 2 and 3), else l3d_chamfer_forward + l3d_chamfer_loss_local_mb.  Outputs: dist / idx as
 * l3d_chamfer_forward (bit for bit), partial[4] as l3d_chamfer_partials, loss[0] fp32.  ws: l3d_chamfer_forward_loss_ws_bytes(B, N, M)
 * bytes of device memory, the first 16 zero before the first call (the kernel re-arms them); one ws per stream. */
size_t l3d_chamfer_forward_loss_ws_bytes(int B, int N, int M);
int l3d_chamfer_forward_loss(const float *xyz1, const float *xyz2, int B, int N, int M, float *dist1, float *dist2, int32_t *idx1,
                             int32_t *idx2, void *ws, double *partial, float *loss, l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * PointNet++ native ops  == utils/lib/src/pointnet2_api.cpp:10-25 (pybind `pointnet2_cuda`)
 * argument order follows the reference wrappers exactly.
 * ------------------------------------------------------------------------------------------- */
/* ball_query_wrapper(b,n,m,radius,nsample,new_xyz,xyz,idx)   K7 ball_query_gpu.cu:9-45
 *   new_xyz [B,m,3], xyz [B,n,3] -> idx [B,m,nsample] int32; strict d2 < r^2; first hit
 *   back-fills; empty ball = 0 (the callee zero-fills, as pointnet2_utils.py:246 did).
 *   workspace: NULL, or b * (16 n + 16448) bytes of device scratch (16-byte aligned): with it, n >= 2048 and nsample <= 64 the
 *   query goes through a per-cloud cell list (a counting sort of the points into cells of edge >= r, then a wave per centroid over
 *   its 27 cells keeping the nsample smallest hit indices) -- the same indices in the same order, ~20x fewer pair evaluations. */
int l3d_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int32_t *idx, void *workspace, l3d_stream_t stream);
/* group_points_wrapper(b,c,n,npoints,nsample,points,idx,out)   K8 group_points_gpu.cu:47-66 */
int l3d_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                     const int32_t *idx, float *out, l3d_stream_t stream);
/* group_points_grad_wrapper(b,c,n,npoints,nsample,grad_out,idx,grad_points)   K9 :8-25
 *   grad_points is zero-filled by the callee. */
int l3d_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                          const int32_t *idx, float *grad_points, l3d_stream_t stream);
/* gather_points_wrapper(b,c,n,npoints,points,idx,out)   K10 sampling_gpu.cu:8-24 */
int l3d_gather_points(int b, int c, int n, int npoints, const float *points, const int32_t *idx,
                      float *out, l3d_stream_t stream);
/* gather_points_grad_wrapper(b,c,n,npoints,grad_out,idx,grad_points)   K11 :46-63 */
int l3d_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                           const int32_t *idx, float *grad_points, l3d_stream_t stream);
/* furthest_point_sampling_wrapper(b,n,m,points,temp,idx)   K12 sampling_gpu.cu:93-209
 *   points [B,n,3]; temp [B,n] scratch (callee initialises it to 1e10); idx [B,m] int32;
 *   starts at index 0. */
int l3d_furthest_point_sampling(int b, int n, int m, const float *points, float *temp,
                                int32_t *idx, l3d_stream_t stream);
/* knn_wrapper(b,n,m,k,unknown,known,dist2,idx)   K13 interpolate_gpu.cu:9-57
 *   unknown [B,n,3] queries, known [B,m,3]; dist2 [B,n,k] (SQUARED), idx [B,n,k] int32,
 *   ascending, lowest index first on ties; slots beyond m hold (+inf, 0). k <= 200. */
int l3d_knn(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2,
            int32_t *idx, l3d_stream_t stream);
/* three_nn_wrapper(b,n,m,unknown,known,dist2,idx)   K14 interpolate_gpu.cu:81-124 */
int l3d_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                 int32_t *idx, l3d_stream_t stream);
/* three_interpolate_wrapper(b,c,m,n,points,idx,weight,out)   K15 interpolate_gpu.cu:149-169 */
int l3d_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx,
                          const float *weight, float *out, l3d_stream_t stream);
/* three_interpolate followed by torch.cat([interpolated, skip], dim=1) (models/flownet3d.py:268-272):
 * out fp32 [B, c + c1, n] = [interpolated (c) | skip (c1)]; skip fp32 [B, c1, n] (NULL when c1 == 0). */
int l3d_three_interpolate_concat(int b, int c, int m, int n, const float *points, const int32_t *idx,
                                 const float *weight, const float *skip, int c1, float *out, l3d_stream_t stream);
/* three_interpolate_grad_wrapper(b,c,n,m,grad_out,idx,weight,grad_points)   K16 :192-214 */
int l3d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                               const int32_t *idx, const float *weight, float *grad_points,
                               l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * torch-level primitives of utils/model_common_utils.py, fused (T4, T7, T8; int64 indices)
 * ------------------------------------------------------------------------------------------- */
/* square_distance(src,dst) :19-38 -> dist [B,N,M] (expanded form, reference rounding order) */
int l3d_square_distance(const float *src, const float *dst, int B, int N, int M, int C, float *dist,
                        l3d_stream_t stream);
/* compute_density(xyz, bandwidth) utils/pointconv_util.py:194-203, fused (no [B,N,N] tensor):
 *   density[b][i] = mean_j exp(-square_distance(xyz,xyz)[i][j] / (2 bw^2)) / (2.5 bw).  xyz [B,N,3] -> density [B,N] */
int l3d_gaussian_density(const float *xyz, int B, int N, float bandwidth, float *density, l3d_stream_t stream);
/* query_ball_point(radius,nsample,xyz,new_xyz,get_cnt) :102-130 (also ppfnet_util.py:96-131 with
 *   itself_indices, pointconv_util.py:85-105): first nsample indices with expanded d2 <= r^2 in
 *   index order, padded with the first (or with itself_indices[b][s] when given, in which case
 *   that index is also excluded from the ball); empty ball -> N.  cnt / itself may be NULL. */
int l3d_query_ball_point(float radius, int nsample, const float *xyz, const float *new_xyz, int B,
                         int N, int S, const int64_t *itself_indices, int64_t *idx, int64_t *cnt,
                         l3d_stream_t stream);
/* index_points(points, idx) :40-56: points [B,N,C], idx [B,S] int64 -> out [B,S,C]
 *   (idx of any rank is flattened to [B,S] by the caller). */
int l3d_index_points(const float *points, const int64_t *idx, int B, int N, int C, int S,
                     float *out, l3d_stream_t stream);
/* farthest_point_sample(xyz, npoint) :58-82: start[b] gives the first centroid (the reference
 *   draws it with torch.randint; start == NULL means index 0); temp [B,N] scratch. */
int l3d_farthest_point_sample(const float *xyz, int B, int N, int npoint, const int64_t *start,
                              float *temp, int64_t *centroids, l3d_stream_t stream);
/* knn_point(k,pos1,pos2) :84-100: pos1 [B,N,3] searched, pos2 [B,M,3] queries ->
 *   val [B,M,k] = sqrt(d2) ascending, idx [B,M,k] int64. */
int l3d_knn_point(int k, const float *pos1, const float *pos2, int B, int N, int M, float *val,
                  int64_t *idx, l3d_stream_t stream);
/* pointconv_util.knn_point(nsample, xyz, new_xyz) utils/pointconv_util.py:107-118: the nsample smallest entries of
 *   square_distance(new_xyz, xyz) = ((-2 q.c) + |q|^2) + |c|^2 (that fp32 rounding sequence) per query, indices only.
 *   The reference calls torch.topk(sorted=False) -- row order unspecified; here nearest first, lowest index first
 *   under exact ties.  xyz [B,N,3] searched, new_xyz [B,S,3] queries -> idx [B,S,nsample] int64. */
int l3d_knn_point_expanded(int nsample, const float *xyz, const float *new_xyz, int B, int N, int S,
                           int64_t *idx, l3d_stream_t stream);

/* QueryAndGroup's tail == utils/lib/pointnet2_utils.py:274-292 in one pass: centred neighbour coordinates
 * (if use_xyz) concatenated with the gathered features.  xyz [B,N,3], new_xyz [B,S,3], features [B,C,N]
 * (NULL if C == 0), idx int32 [B,S,K] -> out [B, 3*use_xyz + C, S, K]. */
int l3d_group_concat(const float *xyz, const float *new_xyz, const float *features, const int32_t *idx, int B,
                     int N, int S, int K, int C, int use_xyz, float *out, l3d_stream_t stream);
/* The grouped conv input of FlowEmbedding / PointNetSetUpConv (models/flownet3d.py:125-180, :182-242) in one pass:
 *   order 0: out = [xyz[idx] - new_xyz | features[idx] | centre broadcast over K]
 *   order 1: out = [features[idx] | xyz[idx] - new_xyz | centre broadcast over K]
 * xyz [B,N,3], new_xyz [B,S,3], features [B,C,N], centre [B,C1,S] (NULL when C1 == 0), idx int32 [B,S,K];
 * out fp32 [B, 3+C+C1, S, K].  Replaces two grouping ops, a subtraction, a repeat and the torch.cat copies. */
int l3d_group_concat2(const float *xyz, const float *new_xyz, const float *features, const float *centre,
                      const int32_t *idx, int B, int N, int S, int K, int C, int C1, int order, float *out,
                      l3d_stream_t stream);
/* The FIRST LAYER of those grouped MLPs without the grouped tensor (models/flownet3d.py:125-180, :182-242, :73-123): conv1
 * is linear in [xyz[idx] - centre | feat[idx] | centre_feat], so with the per-point products U = (s W_feat) feat [B,N,C1],
 * V = (s W_centre) centre_feat + t [B,S,C1] (or NULL, then `shift` [C1] carries t) and wx = s W_xyz [C1][3] (BN scale s and
 * shift t folded in),  out[b][s K + k][:] = act(U[b][idx] + V[b][s] + wx (xyz[idx] - new_xyz[s])),  out [B, S K, C1]
 * channel-last (the next layer's kernels take it as channel_last input).  C1 % 4 == 0, C1 <= 1024. */
int l3d_group_first_layer(const float *U, const float *V, const float *shift, const float *wx, const float *xyz,
                          const float *new_xyz, const int32_t *idx, int B, int N, int S, int K, int C1, int relu,
                          float *out, l3d_stream_t stream);
/* The same with the bound formed inside the kernel from block maxima: maxpart = the 4 x 64 floats l3d_absmax4_partials writes for
 * (U, V, xyz, new_xyz); wxr = max_r sum_d |wx_rd| and shmax = max|shift| come from the layer's parameters.
 * bound = max|U| + (max|V| or shmax) + wxr (max|xyz| + max|new_xyz|). */
int l3d_absmax4_partials(const float *p0, size_t n0, const float *p1, size_t n1, const float *p2, size_t n2, const float *p3,
                         size_t n3, float *out, l3d_stream_t stream);
int l3d_group_first_layer_planes_auto(const float *U, const float *V, const float *shift, const float *wx, const float *xyz,
                                      const float *new_xyz, const int32_t *idx, int B, int N, int S, int K, int C1, int relu,
                                      const float *maxpart, float wxr, float shmax, void *out_img, int *range_flag,
                                      l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Batched 3x3 SVD head  == utils/svd.py:29-58 (T6, without the B host syncs)
 *   src, corr [B,3,N]: centre both, H = src_c corr_c^T, H = U S V^T, R = V U^T with the
 *   det(R) < 0 reflection fix (V[:,2] negated), t = -R mean(src) + mean(corr).
 *   R [B,3,3], t [B,3]; optional H_out [B,3,3] (may be NULL).
 * ------------------------------------------------------------------------------------------- */
int l3d_kabsch(const float *src, const float *corr, int B, int N, float *R, float *t, float *H_out,
               l3d_stream_t stream);
/* rotation only, from given H [B,3,3] (utils/svd.py:38-49) */
int l3d_svd3x3_rotation(const float *H, int B, float *R, l3d_stream_t stream);

/* Soft correspondences of SVDHead == utils/svd.py:22-27, flash-style (softcorr.hip):
 *   scores[i][j] = softmax_j( <src_emb[b][:,i], tgt_emb[b][:,j]> * scale ),  scale = 1/sqrt(C) in the reference
 *   src_corr[b][:,i] = sum_j scores[i][j] * tgt[b][:,j]
 * src_emb [B,C,N], tgt_emb [B,C,M], tgt [B,3,M] -> src_corr [B,3,N], all fp32, channel-first (the
 * reference's layouts).  The [B,N,M] score matrix is never materialised.  workspace: scratch of
 * l3d_soft_correspondence_workspace_floats(B,N,M) floats.  C % 16 != 0 -> L3D_ERR_UNSUPPORTED. */
size_t l3d_soft_correspondence_workspace_floats(int B, int N, int M);
int l3d_soft_correspondence(const float *src_emb, const float *tgt_emb, const float *tgt, int B, int C, int N,
                            int M, float scale, float *workspace, float *src_corr, l3d_stream_t stream);

/* the same with explicit batch strides (in floats) for q, k, v: they may then be channel slices of ONE fused
 * projection output [B, 3*H*D, N] (self-attention) or [B, 2*H*D, M] (keys + values of cross-attention). */
int l3d_attention_forward_strided(const float *q, const float *k, const float *v, int B, int H, int D, int N,
                                  int M, long q_bstride, long k_bstride, long v_bstride, float scale, float *ctx,
                                  l3d_stream_t stream);
/* The same with both GEMMs as f16x2 on the fp16 matrix cores (three fp16 MFMA products per fp32 product instead of bf16x3's
 * six, fp32-level accuracy; attention_f16.hip).  fp16's range is handled inside: one extra pass reads max|q|, max|k|, max|v|
 * into `workspace` (>= 16 bytes of device memory, contents irrelevant) and the operands are scaled by powers of two from
 * them.  Same shapes and strides as l3d_attention_forward_strided.  Output: ctx (fp32, may be NULL) and / or ctx_img, the
 * context as the fp16 activation image of l3d_pointwise_conv_f16 (l3d_f16_image_bytes(1, B N, H D) bytes; may be NULL): the
 * output projection then needs no split pass (|ctx| <= max|v| fixes the plane scale). */
/* The same with the three operand maxima already in `maxima` (uint32 float bits of upper bounds of max|q|, |k|, |v|, e.g. from
 * l3d_pointwise_conv_f16 with amax_out): the pass over q, k, v is not run. */

/* The same computation on the restructured kernel (attention_f16b.hip: 256 queries per workgroup, Q fragments in registers, the
 * probabilities handed from the score accumulators to the second product without leaving registers, one barrier per 32-key
 * tile, unscaled fp16 residuals).  maxima_ready bit 0: workspace already holds the three maxima; bit 1: ctx_img carries an UNSCALED
 * residual plane m = f16(X - h) (for l3d_pointwise_conv_f16 with L3D_CONV_F16_TWO_PLANE) instead of m' = f16((X - h) 2^12). */
int l3d_attention_forward_f16b(const float *q, const float *k, const float *v, int B, int H, int D, int N, int M,
                               long q_bstride, long k_bstride, long v_bstride, float scale, void *workspace, int maxima_ready,
                               float *ctx, void *ctx_img, l3d_stream_t stream);

/* LayerNorm of DCP's pointer network == utils/transformer.py:109-119 (unbiased std, eps added to std):
 *   y[r][:] = a * (x[r][:] - mean_r) / (std_r + eps) + b,   x, y [rows][C] fp32.
 * img == NULL: the fp32 values only (C % 4 == 0, C <= 2048).  img given: additionally (or, with y == NULL, only) the output as the
 * fp16 activation image l3d_pointwise_conv_f16 consumes (l3d_f16_image_bytes(1, rows, C) bytes): the Linear layers behind a
 * LayerNorm then run as f16x2 with no split pass; the plane scale comes from the layer's parameters
 * (|y_c| <= |a_c| sqrt(C-1) + |b_c|); C % 8 == 0, C <= 512; y bit-identical to the img == NULL call. */
int l3d_layernorm_planes(const float *x, const float *a, const float *b, float eps, long rows, int C, float *y, void *img,
                         l3d_stream_t stream);
/* The same LayerNorm over the channels of a CHANNEL-FIRST tensor x [B][C][N] (one normalisation per point), output as
 * y [B][C][N] (or NULL) and / or as the activation image with rows b N + n (img: l3d_f16_image_bytes(1, B N, C) bytes, or NULL):
 * the pointer network keeps the [B,C,N] layout its GEMMs write from end to end.  C in {128, 256, 512}.  flags 1: the image's residual
 * plane is UNSCALED (m = f16(X - h): what l3d_pointwise_conv_f16 reads with L3D_CONV_F16_TWO_PLANE); 0: m' = f16((X - h) 2^12). */
int l3d_layernorm_planes_cf(const float *x, const float *a, const float *b, float eps, int B, int C, int N, float *y,
                            void *img, int flags, l3d_stream_t stream);
/* Residual connection x + sublayer(norm(x)) of utils/transformer.py:82-88 when the sublayer output is channel-first:
 * out[b][n][c] = x[b][n][c] + y[b][c][n];  x, out fp32 [B,N,C], y fp32 [B,C,N] (tiled transpose through LDS). */
int l3d_add_transposed(const float *x, const float *y, int B, int N, int C, float *out, l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Shared-MLP (1x1 conv) stack on fp32 MFMA  (a8)
 * ------------------------------------------------------------------------------------------- */
/* Number of floats of the packed EdgeConv parameter block for channel widths c0(=6)->c1->c2->c3->c4 */
size_t l3d_edgeconv_packed_floats(int c1, int c2, int c3, int c4);
/* Pack + fold (host side, CPU pointers): conv{i}.weight [ci, c(i-1)] row-major and the
 * eval-mode BatchNorm affine (scale[ci], shift[ci]: y = scale * conv + shift) into the
 * fragment-ordered block the kernel streams.  == models/dgcnn.py:13-22 parameters.
 * act_mag[4]: the magnitude (a few standard deviations) expected of each layer's post-ReLU activations (BatchNorm statistics
 * give them); NULL or non-positive entries mean 1.  Only the f16x2 kernel's plane exponents use it. */
int l3d_edgeconv_pack(const float *const w[4], const float *const scale[4], const float *const shift[4],
                      const float *act_mag, int c1, int c2, int c3, int c4, float *packed);
/* Fused EdgeConv stack == models/dgcnn.py:32-46 in eval mode:
 *   graph feature (neighbour, centre) -> 4 x relu(bn(conv1x1)) -> max over k after each ->
 *   concat.  xyz [B,N,3], idx [B,N,k] int64 (k <= 32; DGCNN uses 20), packed from
 *   l3d_edgeconv_pack, pooled [B, N, c1+c2+c3+c4] (CHANNEL-LAST: 64-byte runs per point, the
 *   layout l3d_pointwise_conv consumes with x_channel_last = 1).  Channel widths other than
 *   64/64/128/256 return L3D_ERR_UNSUPPORTED.  The [B,C,N,k] activations never leave the CU. */
int l3d_edgeconv_forward(const float *xyz, const int64_t *idx, int B, int N, int k,
                         const float *packed, int c1, int c2, int c3, int c4, float *pooled,
                         l3d_stream_t stream);
/* Same computation, same packed block, same output; a second kernel design (edgeconv2.hip) that
 * keeps the activations in registers through all four layers (no LDS, no barriers) by chaining the
 * MFMA accumulator layout of one layer into the B operand of the next.  k <= 20. */
/* Same computation, same packed block (its third weight copy), same output layout; layers 2-4 run on
 * the bf16 matrix cores with every fp32 operand split exactly into three bf16 planes and six bf16
 * products per fp32 product ("bf16x3", fp32 accumulate; edgeconv_split.hip) -- fp32-level error at a
 * fraction of the fp32-MFMA time.  k <= 20; packed must be 16-byte aligned. */
int l3d_edgeconv_forward_split(const float *xyz, const int64_t *idx, int B, int N, int k,
                               const float *packed, float *pooled, l3d_stream_t stream);
/* Same computation, same packed block (its fourth weight copy); layers 2-4 as "f16x2" on the fp16 matrix cores
 * (edgeconv_f16.hip): activations X = x 2^T as h = f16(X), m' = f16((X - h) 2^12), weights scaled by a per-layer power of
 * two and split the same way, THREE fp16 MFMA products per fp32 product into one fp32 accumulator -- fp32-level error
 * (tests hold it to the bf16x3 bar) at half of bf16x3's matrix-core work.  T per layer is fixed when the block is packed,
 * from the activation magnitudes the caller expects (l3d_edgeconv_pack's act_mag; BatchNorm statistics give them).
 *   out_mode 0: out = pooled [B,N,512] fp32, channel-last (as the other EdgeConv entry points)
 *   out_mode 1: out = an fp16 activation image of the pooled values (l3d_f16_image_bytes(1, B*N, 512)), the x operand of
 *               l3d_pointwise_conv_f16 -- conv5 then runs without any split pass
 * Range contract: activations must stay below 16x the expected magnitude (fp16 tops out at 65504); the kernel watches
 * the pooled maxima it forms anyway and stores 1 to *range_flag (device or mapped host memory, may be NULL) when a
 * plane value exceeds 60000 -- the outputs are then invalid and the caller re-runs l3d_edgeconv_forward_split.  k <= 20. */
/* The two-plane variant (edgeconv_f16b.hip): same arguments and outputs; two fp16 weight planes per fragment step and an
 * unscaled activation residual (fifth packed copy, csrc/edgeconv_layout.h).  Usable only when the packer could place every
 * layer's weights (packed[l3d_edgeconv_packed_v2_flag_index()] == 1 in the HOST copy of the packed block).
 * out_mode 2: the pooled values as an fp16 activation image like out_mode 1, but with the residual plane unscaled, the
 * x operand of l3d_pointwise_conv_f16 with L3D_CONV_F16_TWO_PLANE. */
int l3d_edgeconv_forward_f16b(const float *xyz, const int64_t *idx, int B, int N, int k, const float *packed, void *out,
                             int out_mode, int *range_flag, l3d_stream_t stream);
int l3d_edgeconv_packed_v2_flag_index(void);
/* Per-point linear layer (Conv1d/Conv2d 1x1 + folded BN + optional ReLU):
 *   y[b][co][n] = act(scale[co] * sum_ci w[co][ci] x[b][ci][n] + shift[co])
 *   x [B,Cin,N] (x_channel_last = 0, torch Conv1d layout) or [B,N,Cin] (x_channel_last = 1),
 *   w [Cout,Cin] (torch conv weight layout), y [B,Cout,N]; scale/shift may be NULL
 *   (identity / zero).  shift is read at [b*shift_bstride + co]: 0 = one vector for the batch,
 *   Cout = one per cloud (a broadcast per-cloud feature concatenated to every point, as in
 *   models/pcn.py:117-119,98-101, becomes a per-cloud shift instead of Cin extra channels).
 *   == models/dgcnn.py:48, models/pointnet.py:22-49, models/pcn.py:84-125 *
 * relu is an activation code for every conv entry point below: 0 none, 1 ReLU, any value > 1 = the IEEE-754 bits
 * of a LeakyReLU negative slope in (0,1) (0.2f -> 0x3E4CCCCD).
 * pool = 0: y [B,Cout,N].  pool = 8, 16, 32, 64: the max over every `pool` consecutive points in the epilogue, y [B,Cout,N/pool]
 * (N % pool == 0): a grouped layer's max over its K neighbours, or -- with 64 and a tiny reduce over the N/64 partial maxima -- a
 * conv followed by a global max-pool whose [B,Cout,N] output is never written. */
int l3d_pointwise_conv(const float *x, int x_channel_last, const float *w, const float *scale,
                       const float *shift, int shift_bstride, int B, int Cin, int Cout, int N,
                       int relu, int pool, float *y, l3d_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * The same per-point linear layer on the bf16 matrix cores with fp32-equivalent results ("bf16x3"):
 * every fp32 operand is split exactly into three bf16 planes (x = h + m + l) and six bf16 MFMA
 * products (hh, hm, mh, hl, lh, mm) are accumulated in fp32 -- product error ~2^-24 relative, the
 * level of an fp32 FMA chain, at 6/16 of the fp32-MFMA cost (conv_split.hip).
 *   l3d_split_bytes(rows, cols): size of the split + tiled image of an fp32 [rows][cols] matrix,
 *     layout [ceil(cols/16)][plane 3][kg 2][rows][8] bf16 (cols zero-padded to 16).
 *   l3d_split_rows: device fp32 [rows][cols] -> that image (weights: rows = Cout, cols = Cin).
 *   l3d_pointwise_conv_split: x_mode 0 = x [B,Cin,N] fp32, 1 = x [B,N,Cin] fp32,
 *     2 = x already split as l3d_split_rows would split the [B*N][Cin] matrix.
 *     Needs Cout % 256 == 0, N % 128 == 0, Cin % 16 == 0, else L3D_ERR_UNSUPPORTED (callers then use
 *     l3d_pointwise_conv).  Other arguments as l3d_pointwise_conv.
 * ------------------------------------------------------------------------------------------- */
size_t l3d_split_bytes(int rows, int cols);
int l3d_split_rows(const float *src, int rows, int cols, void *dst, l3d_stream_t stream);
int l3d_pointwise_conv_split(const void *x, int x_mode, const void *w_split, const float *scale,
                             const float *shift, int shift_bstride, int B, int Cin, int Cout, int N,
                             int relu, int pool, float *y, l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The same 1x1 conv as "f16x2" on the fp16 matrix cores (conv_f16.hip): three fp16 MFMA products per fp32 product --
 * activations x = h + m' 2^-12, weights scaled by a power of two and split into H, Hs = H 2^-12, M -- fp32-level
 * error (tests hold it to the bf16x3 bar) at half of bf16x3's matrix-core work.  Both operands are PRE-SPLIT fp16
 * planes in the tiled layout plane[k / 8][row][8 fp16] (rows = Cout for W, B*N for x), so the kernel moves them
 * global -> LDS by DMA with no registers and no VALU.  Range contract: |x| < 65504 (producers raise *range_flag
 * beyond 60000; results are then invalid and the caller falls back to l3d_pointwise_conv_split).
 *   l3d_f16_image_bytes(kind, rows, cols)    kind 0: bytes of one plane; kind 1: of an activation image (h | m' planes + 16
 *                                            bytes: 2^-T, scratch), the activations being stored times 2^T (T per tensor: see
 *                                            conv_f16.hip); kind 2: of a split weight image of [rows = Cout][cols = Cin]
 *                                            (H | Hs | M planes + 16 bytes: 2^-S, scratch)
 *   l3d_conv_f16_split_weights               w [Cout][Cin] fp32 (device) -> that image (device); two small launches
 *   l3d_split_f16_rows                       x [rows][C] fp32, or [B][C][Npts] with channel_first -> activation image
 *   l3d_pointwise_conv_f16                   y[b][co][n] = act(scale[co] sum_k w[co][k] x[b][n][k] + shift[(b,)co]);
 *                                            Cin % 16 == 0 and (Cout % 256 == 0, N % 256 == 0) or (Cout % 128 == 0,
 *                                            N % 512 == 0: the narrow tile), else L3D_ERR_UNSUPPORTED
 * ------------------------------------------------------------------------------------------- */
size_t l3d_f16_image_bytes(int kind, long rows, int cols);
int l3d_conv_f16_split_weights(const float *w, int Cout, int Cin, void *dst, l3d_stream_t stream);
/* nn.Linear over a handful of rows (PCN's fully connected decoder, models/pcn.py:132-137: rows = clouds):
 * y [R][Cout] = act(x [R][Cin] w [Cout][Cin]^T + bias), fp32 fmaf chains in ascending k; the weight matrix is read once.
 * Cin % 256 == 0 (else L3D_ERR_UNSUPPORTED). */
int l3d_linear_rows(const float *x, const float *w, const float *bias, int R, int Cin, int Cout, int relu, float *y,
                    l3d_stream_t stream);
int l3d_split_f16_rows(const float *x, long rows, int C, int channel_first, int Npts, void *dst, int *range_flag,
                       l3d_stream_t stream);
/* THE f16x2 layer (round 3 exported six spellings of it).  Outputs, in the combinations the kernel family offers
 * (fp32 rows, OR an image and / or pooled maxima):
 *   y         fp32 [B][Cout][N], or NULL
 *   residual  y = residual + act(scale (w x) + shift): residual and y distinct [B][Cout][N] buffers (utils/transformer.py:82-88:
 *             x + sublayer(norm(x)) without a pass over both tensors); Cout % 256 == 0, N % 256 == 0
 *   out_img   the output as the activation image of the NEXT f16x2 layer (l3d_f16_image_bytes(1, B N, Cout) bytes): chains of Linear /
 *             1x1-conv layers stay on the fp16 matrix cores with no split pass in between.  Needs obs = two device floats
 *             {max|shift| over every (b, co), max|scale|} (max|scale| = 1 without a scale); with the weight image's row-sum
 *             maximum and the input image's scale the kernel bounds its outputs and fixes the plane scale itself
 *   ypool     [B][Cout][N/pool] fp32 = the maxima over runs of `pool` (8, 16, 32, 64, 128) consecutive points: pool = 128 for a
 *             global max-pool (models/pooling.py:9-12 after pcn.py:115,124 / pointnet.py:49; a reduce over N/128 values per
 *             channel finishes it), pool = K for the max over a group's K neighbours (models/flownet3d.py:179, :234); the
 *             layer's [B,Cout,N] output is then never written
 *   amax_out  with y: max|y| per group of amax_cdiv output channels (amax_cdiv % 256 == 0) into amax_out[Cout / amax_cdiv]
 *             (uint32 float bits, atomic maximum: the caller zeroes them first): the operand maxima
 *             l3d_attention_forward_f16b wants, from the projection's own epilogue (utils/transformer.py:183-189)
 * flags     L3D_CONV_F16_TWO_PLANE: the input image's residual plane is UNSCALED (m = f16(X - h); written by
 *           l3d_edgeconv_forward_f16b with out_mode 2, by l3d_layernorm_planes_cf / l3d_attention_forward_f16b / this function when
 *           asked): the Hs plane of the weight image is not read (12 instead of 14 LDS fragment reads and 4 instead of 5 DMA pieces
 *           per chunk and wave).  Cout % 256 == 0, N % 256 == 0; outputs: y, y + residual, y + amax_out, or out_img with
 *           L3D_CONV_F16_OUT_UNSCALED.
 *           L3D_CONV_F16_OUT_UNSCALED (with L3D_CONV_F16_TWO_PLANE): out_img carries an unscaled residual plane as well.
 * shift may be per cloud (shift_bstride = Cout, else 0). */
#define L3D_CONV_F16_TWO_PLANE 1
#define L3D_CONV_F16_OUT_UNSCALED 2
#define L3D_CONV_F16_SHIFT_N 4      /* with TWO_PLANE, y only: shift [N] is indexed by the output column -- see l3d_split_f16_operand */
int l3d_pointwise_conv_f16(const void *x_planes, const void *w_planes, const float *scale, const float *shift,
                           int shift_bstride, int B, int Cin, int Cout, int N, int relu, int flags, float *y,
                           const float *residual, void *out_img, const float *obs, float *ypool, int pool,
                           void *amax_out, int amax_cdiv, l3d_stream_t stream);
/* One operand of a training-path product on the f16x2 kernel (an nn.Linear over rows, utils/transformer.py:183-189 of the reference, and
 * its dgrad): x [rows][C] fp32 with row stride `row_stride` -> fp16 planes h | m of x 2^T, UNSCALED residual, T from the window's maximum.
 * kind 0: an activation image (l3d_f16_image_bytes(1, rows, C)); kind 1: the two planes in the slots of a weight image
 * (l3d_f16_image_bytes(2, rows, C)), i.e. the w_planes operand of l3d_pointwise_conv_f16 with L3D_CONV_F16_TWO_PLANE.  With the batch's
 * rows as w_planes (kind 1), the layer's [Cout][Cin] matrix as x_planes (kind 0), B = 1, "Cout" = rows, "N" = Cout and
 * L3D_CONV_F16_SHIFT_N the kernel's output is y [rows][Cout] = x W^T + b, row-major.  rows % 256 == 0 and Cout % 256 == 0 for the GEMM. */
int l3d_split_f16_operand(const float *x, long rows, int C, long row_stride, int kind, void *dst, int *range_flag, l3d_stream_t stream);
/* First layer of a per-point MLP (Cin <= 8; pcn.py:26-33 conv1 3 -> 128, pointnet.py:42) written straight as an activation
 * image: x [B][N][Cin] (channel_last) or [B][Cin][N], w [Cout][Cin], shift [Cout] or NULL, xmax = device float >= max|x|
 * (the plane scale follows from max_r(|shift_r| + xmax sum_c |w_rc|)); raises *range_flag if xmax was not a bound. */
int l3d_first_layer_f16_planes(const float *x, int channel_last, const float *w, const float *shift, const float *xmax,
                               int B, int Cin, int Cout, int N, int relu, void *out_img, int *range_flag, l3d_stream_t stream);

/* PCN's folding decoder == models/pcn.py:84-101 (conv5 -> ReLU -> conv6 -> ReLU -> conv7, + centre) in one
 * kernel (fold_mlp.hip): g [B,N,5] = (grid u, v, centre x, y, z) per fine point, w5g [512,5] = conv5's
 * columns for those five inputs, s5 [B,512] = conv5.bias + conv5.weight[:, 5:] . global_feature (per cloud),
 * w6_split = l3d_split_rows(conv6.weight [512,512]), b6 [512], w7 [3,512], b7 [3], centre [B,N,3]
 * -> out [B,N,3].  The two [B,512,N] activations (2.1 GB each at B=64, N=16384) are never formed. */
int l3d_fold_mlp(const float *g, int CG, const float *w5g, const float *s5, const void *w6_split,
                 const float *b6, const float *w7, const float *b7, const float *centre, int B, int N,
                 float *out, l3d_stream_t stream);
/* The same with the conv6 GEMM as f16x2 on the fp16 matrix cores (fold_mlp_f16.hip): w6_planes is W6's [512][512] weight
 * image from l3d_conv_f16_split_weights; the activation scale is chosen per workgroup from a bound it computes itself. */
int l3d_fold_mlp_f16(const float *g, int CG, const float *w5g, const float *s5, const void *w6_planes, const float *b6,
                     const float *w7, const float *b7, const float *centre, int B, int N, float *out, l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Approximate EMD  == losses/cuda/emd_torch/pkg/include/emd.h:47-50 (pybind `_emd_ext._emd`)
 *   emd_forward(xyz1,xyz2) -> cost [B], match [B,n,m] (indexed [l*n+k], emd.cuh:158); the reference's forward allocates its
 *   own scratch (pkg/src/cuda/emd.cu:18-22) -- here the caller passes `workspace` of l3d_emd_workspace_bytes(B,n,m) bytes
 *   (the ten levels' ratioL / ratioR, the remainders and the cost partials; no initialisation needed).  22 stream-ordered
 *   launches, no host synchronisation.   emd_backward(xyz1,xyz2,match) -> grad1, grad2.
 *   split: lanes per row in the sweeps, 1, 2 or 4; 0 = chosen from B * min(n, m).  The result's bits do not depend on it.
 * ------------------------------------------------------------------------------------------- */
size_t l3d_emd_workspace_bytes(int B, int n, int m);
int l3d_emd_forward(const float *xyz1, const float *xyz2, int B, int n, int m, float *match,
                    float *cost, void *workspace, int split, l3d_stream_t stream);
int l3d_emd_backward(const float *xyz1, const float *xyz2, const float *match, int B, int n, int m,
                     float *grad1, float *grad2, l3d_stream_t stream);

/* CurveNet LPFA grouping == utils/curvenet_util.py:260-291 in one pass: geo [B,9,N,k] = (centre, neighbour, neighbour -
 * centre) coordinates and, when x != NULL, diff [B,C,N,k] = x[:, :, idx] - x[:, :, n].  xyz [B,N,3], x [B,C,N],
 * idx [B,N,k] int64 (the first k of l3d_knn_graph's k + 1, :264). */
int l3d_lpfa_group(const float *xyz, const float *x, const int64_t *idx, int B, int N, int C, int k, float *geo, float *diff,
                   l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Device-side data feed of the registration path (feed.hip)
 *   l3d_uniform_clouds : out [B,N,3] ~ U(lo,hi), generated on the device from (seed, element index): reproducible, no
 *       host tensor and no copy.
 *   l3d_euler_transform == DCPTransform / DeepGMRTransform (ops/transform_functions.py:271-345) for a whole batch:
 *       euler_zyx [B,3] = (anglez, angley, anglex), trans [B,3]; source [B,N,3] = R template + t with
 *       R = scipy Rotation.from_euler('zyx', ...) evaluated in fp64; igt [B,4,4] = [R^T | t ; 0 0 0 1] -- the 3x3
 *       block is Rotation.apply(np.eye(3)), i.e. the transpose, exactly as the reference stores it (:306).
 * ------------------------------------------------------------------------------------------- */
int l3d_uniform_clouds(unsigned long long seed, int B, int N, float lo, float hi, float *out, l3d_stream_t stream);
int l3d_euler_transform(const float *tmpl, const float *euler_zyx, const float *trans, int B, int N, float *source,
                        float *igt, l3d_stream_t stream);
/* PNLKTransform / RPMNetTransform (ops/transform_functions.py:109-192) for a batch: twist [B,6] = (w, v) per cloud;
 * igt [B,4,4] = se3.exp(x) (template -> source), gt [B,4,4] = se3.exp(-x), source = R template + p. */
int l3d_twist_transform(const float *tmpl, const float *twist, int B, int N, float *source, float *igt, float *gt,
                        l3d_stream_t stream);
/* PCRNetTransform.__call__ (ops/transform_functions.py:194-269) for a batch: pose7 [B,7] = (quaternion w x y z, translation);
 * the quaternion is normalised as create_pose_7d does; source = qrot(q, template) + t. */
int l3d_quat_transform(const float *tmpl, const float *pose7, int B, int N, float *source, l3d_stream_t stream);

/* SceneflowDataset.__getitem__ (data_utils/dataloaders.py:400-432) for a batch, from a dataset resident in HBM: points1 /
 * color1 / flow [F,n1,3], valid_mask1 [F,n1] (bytes), points2 / color2 [F,n2,3]; scene_idx [B] int64; sample1 / sample2
 * [B,S] int32 row indices (np.random.choice(n, npoints, replace=False) of the train partition) or both NULL (test partition:
 * the first S rows).  Outputs [B,S,3] (mask [B,S] bytes): pos1 and pos2 minus np.mean(pos1, 0) of the sampled rows, the mean
 * replayed in numpy's order (sequential fp32 adds, fp64 divide) -- bit-identical to the reference.  S <= 8192. */
int l3d_sceneflow_batch(const float *points1, const float *points2, const float *color1, const float *color2, const float *flow,
                        const unsigned char *mask1, const long long *scene_idx, const int *sample1, const int *sample2, int B, int n1,
                        int n2, int S, float *o_pos1, float *o_pos2, float *o_color1, float *o_color2, float *o_flow,
                        unsigned char *o_mask, l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training-mode BatchNorm around the 1x1-conv GEMMs (train.hip).  z / dy / y / dz are [B,C,P]
 * fp32 (P = points, or points x neighbours).  Statistics come out as PER-CLOUD fp64 partial sums [B,C,2] (fixed order,
 * no atomics); the host adds them in global cloud order (after an all_gather when the batch is sharded across ranks),
 * which makes the batch statistics -- and the two backward reductions -- bit-identical for any number of GPUs.
 *   l3d_channel_stats     part[b][c] = (sum_p z, sum_p z^2)
 *   l3d_bn_act_forward    y = act(z scale[c] + shift[c]),  act: 0 none, 1 ReLU, > 1 the fp32 bits of a LeakyReLU slope
 *   l3d_bn_backward_stats part[b][c] = (sum_p g, sum_p g zhat),  g = dy act'(z scale + shift),  zhat = (z - mean[c]) rstd[c]
 *   l3d_bn_act_backward   dz = gr[c] (g - m1[c] - zhat m2[c])
 * ------------------------------------------------------------------------------------------- */
int l3d_channel_stats(const float *z, int B, int C, long P, double *part, l3d_stream_t stream);
int l3d_bn_act_forward(const float *z, const float *scale, const float *shift, int B, int C, long P, int act, float *y,
                       l3d_stream_t stream);
/* mean, rstd, gr, m1, m2 [C] are fp64 and dz is evaluated in fp64 and rounded once (sum_p dz = 0 by construction: an
 * fp32-rounded m1 is a systematic error the weight gradient multiplies by the point count).  m1 = m2 = 0: eval-mode
 * BatchNorm or a plain bias layer (the statistics do not depend on z).
 * dpool / pidx / K (NULL / NULL / 0: none): for a layer whose output y [B][C][P] is ALSO max-pooled over runs of K consecutive
 * positions (the max over the k neighbours behind an EdgeConv layer) the gradient at y is dy (may then be NULL) plus dpool
 * [B][C][P/K] at the position pidx [B][C][P/K] names (l3d_max_last's arg-max) -- the dense scatter and the add of the two gradient
 * tensors are never formed.  K <= 256, P % K == 0, P < 2^22. */
int l3d_bn_backward_stats(const float *dy, const float *z, const float *scale, const float *shift, const double *mean,
                          const double *rstd, int B, int C, long P, int act, double *part, const float *dpool,
                          const unsigned char *pidx, int K, l3d_stream_t stream);
int l3d_bn_act_backward(const float *dy, const float *z, const float *scale, const float *shift, const double *mean,
                        const double *rstd, const double *gr, const double *m1, const double *m2, int B, int C, long P, int act,
                        float *dz, const float *dpool, const unsigned char *pidx, int K, l3d_stream_t stream);
/* tot[j] = part[0][j] + part[1][j] + ... + part[B-1][j]: per-cloud fp64 partials [B][M] added left to right, i.e. in
 * global cloud order whatever B's factorisation into ranks was. */
int l3d_sum_clouds_f64(const double *part, int B, long M, double *tot, l3d_stream_t stream);
/* Per-channel finalisation of a layer's statistics in ONE launch each way (models/_train.py did it with ~25 + ~8 scalar-sized torch
 * operations per layer).  Forward, mode 0 = train-mode BatchNorm from the per-cloud sums part [B][C][2] of the conv output without
 * its bias (clouds added in cloud order, fp64), running statistics updated as torch.nn.BatchNorm does when given; 1 = eval-mode
 * BatchNorm (running statistics); 2 = no BatchNorm.  -> mean64, rstd64, gr64 = gamma rstd (fp64), scale, shift (fp32):
 * y = act(z scale + shift).  Backward: per-cloud (sum g, sum g zhat) of this rank (part_local) and of all ranks (part_all, NULL =
 * the same) -> the batch means m1, m2 (zeros without batch statistics) and this shard's dbias / dgamma / dbeta (each may be NULL). */
int l3d_bn_finalize(const double *part, int B, int C, double n, const float *bias, const float *gamma, const float *beta, double eps,
                    int mode, double momentum, float *running_mean, float *running_var, double *mean64, double *rstd64,
                    double *gr64, float *scale, float *shift, l3d_stream_t stream);
int l3d_bn_backward_finalize(const double *part_local, int Bl, const double *part_all, int Ba, int C, double n, int batch_stats,
                             const double *gr64, double *m1, double *m2, float *dbias, float *dgamma, float *dbeta,
                             l3d_stream_t stream);
/* Max over the last, contiguous axis with its arg-max and the backward of it (the max over the k neighbours behind every
 * EdgeConv layer of a training step, models/dgcnn.py:36-46): v [R] = max_k x [R][K], idx [R] = the FIRST k that attains it
 * (torch's rule; one byte, K <= 256); gx [R][K] = g [R] at idx [R] and zero elsewhere, one dense pass. */
int l3d_max_last(const float *x, long R, int K, float *v, unsigned char *idx, l3d_stream_t stream);
int l3d_max_last_backward(const float *g, const unsigned char *idx, long R, int K, float *gx, l3d_stream_t stream);

/* Backward of the pointer network's LayerNorm (utils/transformer.py:109-119: unbiased std, eps added to std; forward =
 * l3d_layernorm_planes): x, g = dL/dy, dx [rows][C]; a [C]; da, db [C] summed over the rows in a fixed order (workgroup partials in
 * the workspace, then fp64 in workgroup order).  workspace: l3d_layernorm_backward_workspace_floats(rows, C) floats.
 * C % 4 == 0, C <= 2048, 16-byte aligned pointers. */
size_t l3d_layernorm_backward_workspace_floats(long rows, int C);
int l3d_layernorm_ref_backward(const float *x, const float *a, const float *g, float eps, long rows, int C, float *dx,
                               float *workspace, float *da, float *db, l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * A whole narrow set-abstraction layer behind the ball query in one kernel (sa_fused.hip; models/flownet3d.py:108-122
 * PointNetSetAbstraction.forward at sa1 = 3+3 -> 32 -> 32 -> 64, K 16, flownet3d.py:272; pointnet2's 3+D -> 64 -> 64 -> 128):
 *   out [B][C3][S] = max_k relu(s3 (W3 relu(s2 (W2 relu(s1 (W1 [xyz[idx] - new_xyz | feat[idx]]) + t1)) + t2)) + t3)
 * xyz [B][N][3], new_xyz [B][S][3], feat [B][D][N] (NULL when D == 0; 3 + D <= 16), idx int32 [B][S][K] (K in {8, 16, 32, 64}),
 * (C1, C2, C3) in {(32, 32, 64), (64, 64, 128)}, else L3D_ERR_UNSUPPORTED.  params (device floats, 16-byte aligned): per layer the
 * weights in the kernel's LDS order -- w[n][k = 4 s + g] at ((g (NS / RUN) + s / RUN) CP + n) RUN + s % RUN with NS = Cin / 4,
 * RUN = min(4, NS), CP = Cout (+ 16 zero rows when RUN == 2); layer 1's input channels zero-padded to 8 when 3 + D <= 8, else to
 * 16 -- followed by scale [Cout] and shift [Cout] (the folded eval-mode BatchNorm).  fp32 MFMA: an exact fp32 fma chain. */
int l3d_sa_mlp3_fused(const float *xyz, const float *new_xyz, const float *feat, const int32_t *idx, const float *params, int B,
                      int N, int S, int K, int D, int C1, int C2, int C3, float *out, l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Strided batched fp32 GEMM and row softmax (bmm.hip): the TRAINING path of the pointer network and the SVD head -- the
 * products of `attention` (utils/transformer.py:127-132), of its nn.Linear layers (:183-189, :228-238), of the SVD head's scores
 * (utils/svd.py:27-31) and of square_distance (utils/model_common_utils.py:34-37), forward and what autograd derives for them,
 * read where the tensors lie (a transposed operand is its strides swapped).
 *   C[i][j] = act(alpha A[i][j] B[i][j] + bias) (+ C[i][j]),  A [M x K], B [K x N], C [M x N], i < nb1, j < nb2
 *   *_strides: four element strides {batch1, batch2, row, column} (host arrays); flags: 1 accumulate into C, 2 ReLU, 4 bias[n],
 *   8 bias[m]; parts > 1: K split into `parts` ranges through `workspace` (nb1 nb2 parts M N floats), summed in ascending order
 *   (weight gradients: few output tiles, K = every row of the batch).  fp32 MFMA: an exact fma chain per element, ascending k.
 * l3d_softmax_rows: dp == NULL: y = softmax(scale x) over the last axis of [rows][cols]; dp given: y = scale p (dp - sum_j p_j dp_j)
 * with p = x (the backward through the softmax and the score scale).  y may alias x / dp.  cols <= 8192.
 * l3d_colsum_rows: out[c] = sum_r x[r row_stride + c] -- the bias gradient of an nn.Linear over rows (db = 1^T g); a fixed summation
 * tree over chunks of 128 rows: the same bits on every run (workspace: l3d_colsum_rows_workspace_bytes). */
int l3d_bmm_f32(const float *A, const long *a_strides, const float *B, const long *b_strides, float *C, const long *c_strides,
                int nb1, int nb2, int M, int N, int K, float alpha, int flags, const float *bias, int parts, float *workspace,
                l3d_stream_t stream);
int l3d_softmax_rows(const float *x, const float *dp, long rows, int cols, float scale, float *y, l3d_stream_t stream);
size_t l3d_colsum_rows_workspace_bytes(long rows, int cols);
int l3d_colsum_rows(const float *x, long rows, int cols, long row_stride, void *workspace, float *out, l3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Weight gradient of a 1x1 conv / Linear over points (wgrad.hip; the autograd of nn.Conv1d / Conv2d(k=1) in
 * models/dgcnn.py:34-48, models/pcn.py:110-153, models/pointnet.py:51-73 under examples/train_pcn.py:70-91):
 *   dw[co][ci] = sum_b sum_p dz[b][co][p] x[b][ci][p]          dz [B,Cout,P], x [B,Cin,P] fp32
 * exact-fp32 products on the fp32 MFMA, the B*P reduction split into (cloud, pc-point chunk) pieces whose partial sums
 * are added in piece order in fp64: deterministic (no atomics) and accurate to one rounding of <= pc-term fp32 sums.
 * pc <= 0: 2048.  workspace: l3d_wgrad_workspace_bytes(B, Cout, Cin, P, pc) bytes, caller-allocated.
 * ------------------------------------------------------------------------------------------- */
size_t l3d_wgrad_workspace_bytes(int B, int Cout, int Cin, long P, int pc);
int l3d_wgrad(const float *dz, const float *x, int B, int Cout, int Cin, long P, int pc, float *workspace, float *dw,
              l3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* L3D_HIP_H_ */
