/* TEST INFRASTRUCTURE -- C entry points onto the REFERENCE's own EMD kernels.
 *
 * oracle/build_ref.py compiles this file with -I<ref>/losses/cuda/emd_torch/pkg/include so that the
 * `#include "cuda/emd.cuh"` below reads the reference's kernels (approxmatch, matchcost, matchcostgrad1/2,
 * emd.cuh:7-323) and its launchers (emd.cuh:187-199, :246-256, :325-345) WHERE THEY LIE.  The launchers are
 * written against torch 0.4's at::Tensor (`.type()`, `.data<T>()`, AT_DISPATCH_FLOATING_TYPES); the ten
 * lines below stand in for exactly those three names with a raw-pointer box, fp32 only, so the launch
 * configurations (<<<32,512>>>, <<<dim3(32,32),256>>>) and the cudaDeviceSynchronize stay the reference's.
 * The host allocation the reference does in src/cuda/emd.cu:18-22 / :56-57 (zero-filled match, cost,
 * temp = 2*(n+m) floats per cloud, grads) is the caller's job here.  Used by tests/ only, on the GPU box. */
#include "ref_compat/cuda_runtime.h"
#include <vector>

namespace at {
struct RefType {};
struct Tensor {
    float* p;
    RefType type() const { return {}; }
    template <typename T> T* data() const { return reinterpret_cast<T*>(p); }
};
}  // namespace at
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...) { (void)(TYPE); using scalar_t = float; __VA_ARGS__(); }

#include "cuda/emd.cuh"

extern "C" {
/* emd_forward_cuda (src/cuda/emd.cu:8-44): match [b,m,n]-indexed as the kernel does, temp [b,2(n+m)], cost [b] -- all pre-zeroed */
void ref_emd_forward(int b, int n, int m, const float* xyz1, const float* xyz2, float* match, float* temp, float* cost)
{
    at::Tensor t1{const_cast<float*>(xyz1)}, t2{const_cast<float*>(xyz2)}, tm{match}, tt{temp}, tc{cost};
    approxmatchLauncher(b, n, m, t1, t2, tm, tt);
    matchcostLauncher(b, n, m, t1, t2, tm, tc);
}
/* emd_backward_cuda (src/cuda/emd.cu:46-69): grads pre-zeroed */
void ref_emd_backward(int b, int n, int m, const float* xyz1, const float* xyz2, const float* match, float* grad1, float* grad2)
{
    at::Tensor t1{const_cast<float*>(xyz1)}, t2{const_cast<float*>(xyz2)}, tm{const_cast<float*>(match)}, g1{grad1}, g2{grad2};
    matchcostgradLauncher(b, n, m, t1, t2, tm, g1, g2);
}
}
