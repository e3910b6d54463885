// common.h -- shared helpers for the gfx950 kernels of libl3d_hip.so.
// Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (see build.py)
// so that a*b+c never fuses unless fmaf() is written out: the distance kernels must
// reproduce the reference's fp32 rounding sequence bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/l3d_hip.h"

#define L3D_WAVE 64

extern thread_local int g_l3d_last_hip_error;

static inline int l3d_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_l3d_last_hip_error = (int)e;
        return L3D_ERR_LAUNCH;
    }
    return L3D_OK;
}

#define L3D_REQUIRE(cond) \
    do {                  \
        if (!(cond)) return L3D_ERR_INVALID_ARG; \
    } while (0)

static inline int l3d_divup(long a, long b) { return (int)((a + b - 1) / b); }

// -------------------------------------------------------------------------------------------
// Per-lane sorted top-K list kept entirely in VGPRs (K is a compile-time constant so every
// index below is static).  Keys are "larger is better"; equal keys keep insertion order, so
// scanning candidates in ascending index order yields lowest-index-first on ties.
//   insert(): v[i] <- med3(v[i-1], v[i], key) is exactly the sorted-insert update when
//   v[i-1] >= v[i]; one v_med3_f32 per slot for the keys, cmp+2 cndmask for the payloads.
// -------------------------------------------------------------------------------------------
// Values-only variant: one v_med3_f32 per slot and nothing else.  Used for the first pass of the
// two-pass selection (find the exact K-th best key), where indices are not needed yet.
template <int K>
struct TopKV {
    float v[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < K; i++) v[i] = -INFINITY;
    }
    __device__ __forceinline__ float worst() const { return v[K - 1]; }
    __device__ __forceinline__ void insert(float key) {
#pragma unroll
        for (int i = K - 1; i > 0; i--) v[i] = __builtin_amdgcn_fmed3f(v[i - 1], v[i], key);
        v[0] = fmaxf(v[0], key);
    }
};

template <int K>
struct TopK {
    float v[K];
    int id[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < K; i++) { v[i] = -INFINITY; id[i] = 0; }
    }
    __device__ __forceinline__ float worst() const { return v[K - 1]; }
    __device__ __forceinline__ void insert(float key, int j) {
        bool gt_prev = key > v[K - 1];   // running "key > v[i]" for the slot below
#pragma unroll
        for (int i = K - 1; i > 0; i--) {
            bool gt_up = key > v[i - 1];
            // new id[i]: slot above shifts down | key lands here | unchanged
            id[i] = gt_up ? id[i - 1] : (gt_prev ? j : id[i]);
            v[i] = __builtin_amdgcn_fmed3f(v[i - 1], v[i], key);
            gt_prev = gt_up;
        }
        id[0] = gt_prev ? j : id[0];
        v[0] = fmaxf(v[0], key);
    }
};

// Activation code of the conv entry points (their `relu` argument): 0 none, 1 ReLU, any value > 1 = the
// IEEE-754 bits of a LeakyReLU negative slope in (0, 1) (models/prnet.py:79 uses 0.2).  Call only if act != 0.
#ifdef __HIPCC__
__device__ __forceinline__ float l3d_act(float v, int act)
{
    return act == 1 ? fmaxf(v, 0.f) : fmaxf(v, v * __int_as_float(act));
}
#endif
