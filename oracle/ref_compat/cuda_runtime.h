/* TEST INFRASTRUCTURE (oracle/_ref build only) -- lets the reference's own .cu kernels
 * (utils/lib/src/*_gpu.cu, losses/cuda/emd_torch/pkg/include/cuda/emd.cuh) compile with hipcc
 * from where they lie under /root/reference.  The kernels are plain `__global__` code; the only
 * CUDA runtime names they touch are the seven below.  Not part of the product; never shipped. */
#ifndef L3D_REF_COMPAT_CUDA_RUNTIME_H
#define L3D_REF_COMPAT_CUDA_RUNTIME_H
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define cudaError_t            hipError_t
#define cudaSuccess            hipSuccess
#define cudaGetLastError       hipGetLastError
#define cudaGetErrorString     hipGetErrorString
#define cudaStream_t           hipStream_t
#define cudaDeviceSynchronize  hipDeviceSynchronize
#define cudaMemset             hipMemset
using std::max;
using std::min;
#endif
