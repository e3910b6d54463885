/* TEST INFRASTRUCTURE: the reference's *_gpu.h headers only need `at::Tensor` as a name in
 * the (unused here) wrapper prototypes. */
#ifndef L3D_REF_COMPAT_TENSOR_H
#define L3D_REF_COMPAT_TENSOR_H
#include "../../cuda_runtime.h"
namespace at { class Tensor; }
#endif
