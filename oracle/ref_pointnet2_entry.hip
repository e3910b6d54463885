/* TEST INFRASTRUCTURE -- C entry points onto the REFERENCE's own PointNet++ kernel launchers.
 *
 * oracle/build_ref.py compiles <ref>/utils/lib/src/{ball_query,group_points,interpolate,sampling}_gpu.cu
 * where they lie (hipcc, gfx950, include path oracle/ref_compat/ for the six CUDA runtime names they use)
 * together with this file into oracle/_ref/libref_pointnet2*.so.  Nothing of the reference is copied:
 * this file only declares the launchers (prototypes as in <ref>/utils/lib/src/*_gpu.h) and forwards to them,
 * in the argument order of the pybind wrappers (<ref>/utils/lib/src/pointnet2_api.cpp:10-25) minus the
 * at::Tensor boxing (the wrappers' bodies are `tensor.data<T>()` + the launcher call,
 * e.g. ball_query.cpp:14-25).  Used by tests/ only, on the GPU box, to compare l3d_* with K7-K16. */
#include <hip/hip_runtime.h>

void ball_query_kernel_launcher_fast(int b, int n, int m, float radius, int nsample,
                                     const float* new_xyz, const float* xyz, int* idx, hipStream_t stream);
void group_points_kernel_launcher_fast(int b, int c, int n, int npoints, int nsample,
                                       const float* points, const int* idx, float* out, hipStream_t stream);
void group_points_grad_kernel_launcher_fast(int b, int c, int n, int npoints, int nsample,
                                            const float* grad_out, const int* idx, float* grad_points, hipStream_t stream);
void knn_kernel_launcher_fast(int b, int n, int m, int k, const float* unknown,
                              const float* known, float* dist2, int* idx, hipStream_t stream);
void three_nn_kernel_launcher_fast(int b, int n, int m, const float* unknown,
                                   const float* known, float* dist2, int* idx, hipStream_t stream);
void three_interpolate_kernel_launcher_fast(int b, int c, int m, int n,
                                            const float* points, const int* idx, const float* weight, float* out, hipStream_t stream);
void three_interpolate_grad_kernel_launcher_fast(int b, int c, int n, int m, const float* grad_out,
                                                 const int* idx, const float* weight, float* grad_points, hipStream_t stream);
void gather_points_kernel_launcher_fast(int b, int c, int n, int npoints,
                                        const float* points, const int* idx, float* out, hipStream_t stream);
void gather_points_grad_kernel_launcher_fast(int b, int c, int n, int npoints,
                                             const float* grad_out, const int* idx, float* grad_points, hipStream_t stream);
void furthest_point_sampling_kernel_launcher(int b, int n, int m,
                                             const float* dataset, float* temp, int* idxs, hipStream_t stream);

extern "C" {
void ref_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int* idx, void* s)
{ ball_query_kernel_launcher_fast(b, n, m, radius, nsample, new_xyz, xyz, idx, (hipStream_t)s); }
void ref_group_points(int b, int c, int n, int npoints, int nsample, const float* points, const int* idx, float* out, void* s)
{ group_points_kernel_launcher_fast(b, c, n, npoints, nsample, points, idx, out, (hipStream_t)s); }
void ref_group_points_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out, const int* idx, float* grad_points, void* s)
{ group_points_grad_kernel_launcher_fast(b, c, n, npoints, nsample, grad_out, idx, grad_points, (hipStream_t)s); }
void ref_knn(int b, int n, int m, int k, const float* unknown, const float* known, float* dist2, int* idx, void* s)
{ knn_kernel_launcher_fast(b, n, m, k, unknown, known, dist2, idx, (hipStream_t)s); }
void ref_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2, int* idx, void* s)
{ three_nn_kernel_launcher_fast(b, n, m, unknown, known, dist2, idx, (hipStream_t)s); }
void ref_three_interpolate(int b, int c, int m, int n, const float* points, const int* idx, const float* weight, float* out, void* s)
{ three_interpolate_kernel_launcher_fast(b, c, m, n, points, idx, weight, out, (hipStream_t)s); }
void ref_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out, const int* idx, const float* weight, float* grad_points, void* s)
{ three_interpolate_grad_kernel_launcher_fast(b, c, n, m, grad_out, idx, weight, grad_points, (hipStream_t)s); }
void ref_gather_points(int b, int c, int n, int npoints, const float* points, const int* idx, float* out, void* s)
{ gather_points_kernel_launcher_fast(b, c, n, npoints, points, idx, out, (hipStream_t)s); }
void ref_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out, const int* idx, float* grad_points, void* s)
{ gather_points_grad_kernel_launcher_fast(b, c, n, npoints, grad_out, idx, grad_points, (hipStream_t)s); }
void ref_furthest_point_sampling(int b, int n, int m, const float* dataset, float* temp, int* idxs, void* s)
{ furthest_point_sampling_kernel_launcher(b, n, m, dataset, temp, idxs, (hipStream_t)s); }
}
