/* TEST INFRASTRUCTURE -- C entry points onto the REFERENCE's own Chamfer GPU kernels (K1 / K2).
 *
 * oracle/build_ref.py compiles <ref>/losses/cuda/chamfer_distance/chamfer_distance.cu where it lies together
 * with this file into oracle/_ref/libref_chamfer.so.  The two launchers (chamfer_distance.cu:139-155, :189-209)
 * already take raw pointers; the pybind layer (chamfer_distance.cpp:22-56) only unboxes tensors.  They launch on
 * the legacy default stream, as the reference does.  Used by tests/ only, on the GPU box. */
#include <hip/hip_runtime.h>

void ChamferDistanceKernelLauncher(const int b, const int n, const float* xyz, const int m, const float* xyz2,
                                   float* result, int* result_i, float* result2, int* result2_i);
void ChamferDistanceGradKernelLauncher(const int b, const int n, const float* xyz1, const int m, const float* xyz2,
                                       const float* grad_dist1, const int* idx1, const float* grad_dist2, const int* idx2,
                                       float* grad_xyz1, float* grad_xyz2);

extern "C" {
void ref_chamfer_forward(int b, int n, const float* xyz1, int m, const float* xyz2, float* dist1, int* idx1, float* dist2, int* idx2)
{ ChamferDistanceKernelLauncher(b, n, xyz1, m, xyz2, dist1, idx1, dist2, idx2); }
void ref_chamfer_backward(int b, int n, const float* xyz1, int m, const float* xyz2, const float* gd1, const int* idx1,
                          const float* gd2, const int* idx2, float* g1, float* g2)
{ ChamferDistanceGradKernelLauncher(b, n, xyz1, m, xyz2, gd1, idx1, gd2, idx2, g1, g2); }
}
