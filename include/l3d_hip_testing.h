/* l3d_hip_testing.h -- kernel-selection hooks of libl3d_hip.so for tests and benchmarks.  NOT part of the drop-in boundary
 * (include/l3d_hip.h): each entry point below is its l3d_hip.h namesake with the kernel named explicitly instead of chosen by
 * shape, so that the test suite can hold the alternatives to identical results and the benchmarks can time them side by side. */
#ifndef L3D_HIP_TESTING_H
#define L3D_HIP_TESTING_H
#include "l3d_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* The same with the kernel named explicitly: variant 0 = by shape (what l3d_knn_graph does), 1 = the two-pass insertion
 * kernel (any N, k <= 200), 2 = ranking values on the fp32 matrix cores + selection by rank counting (k <= 24,
 * 256 <= N <= 2048; L3D_ERR_UNSUPPORTED otherwise).  Identical results (lowest index first under exact ties). */
int l3d_knn_graph_variant(const float *xyz, int B, int N, int k, int64_t *idx, int variant, l3d_stream_t stream);

/* The same with the kernel choice as an ARGUMENT (results are bit-identical either way; tests compare them):
 * variant 0 = one (query, candidate) pair per instruction sequence, 1 = auto (what l3d_chamfer_forward does: the packed kernel,
 * and from N * M >= 2^24 pairs per cloud the matrix-core kernel), 2 = always the packed-fp32 kernel (two queries per lane, argmin
 * per chunk of 8), 3 = always chamfer_mfma.hip (candidates ranked on the fp16 matrix cores, the answer fixed by an exact
 * re-evaluation of every candidate inside the ranking's error band). */
int l3d_chamfer_forward_variant(const float *xyz1, const float *xyz2, int B, int N, int M, float *dist1,
                                float *dist2, int32_t *idx1, int32_t *idx2, int variant, l3d_stream_t stream);

/* The same with the kernel choice as an argument (bit-identical results: both add a point's terms in the order of
 * chamfer_distance.cpp:138-176): variant 0 = scan of the partner cloud's selections per point, 2 = selections sorted by
 * target in LDS, one binary search per point (N, M <= 32768, else L3D_ERR_UNSUPPORTED), 1 = auto (l3d_chamfer_backward). */
int l3d_chamfer_backward_variant(const float *xyz1, const float *xyz2, int B, int N, int M,
                                 const float *graddist1, const float *graddist2, const int32_t *idx1,
                                 const int32_t *idx2, float *gradxyz1, float *gradxyz2, int variant,
                                 l3d_stream_t stream);

/* the same with the kernel named: variant 0 = automatic (k <= 4 and >= 65 536 queries: the four-slot kernel of knn_small.hip; otherwise the
 * wave-per-query selection kernel of knn_select.hip when k <= m <= 8192 and (k > 32 or m >= 1024), the lane-per-query kernels
 * of knn.hip for the rest), 1 = lane-per-query, 2 = selection kernel (L3D_ERR_UNSUPPORTED unless k <= m <= 8192), 3 = four-slot
 * kernel (k <= 4).  Results are identical; tests and tools use it. */
int l3d_knn_variant(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2,
                    int32_t *idx, int variant, l3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
