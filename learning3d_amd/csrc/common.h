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

// sizes of the f16x2 plane images (conv_f16.hip; exported as l3d_f16_image_bytes(kind, rows, cols))
static inline size_t l3d_f16_plane_bytes(long rows, int cols) { return (size_t)((cols + 7) / 8) * (size_t)rows * 16; }   // one plane, tiled layout
static inline size_t l3d_f16_act_bytes(long rows, int cols) { return 2 * l3d_f16_plane_bytes(rows, cols) + 16; }         // h | m' + {2^-T, scratch}
static inline size_t l3d_conv_f16_weight_bytes(int Cout, int Cin) { return 3 * l3d_f16_plane_bytes(Cout, Cin) + 16; }    // H | Hs | M + {2^-S, maxima}

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

// TopK<20>::insert written out as 4 instructions per slot (compare, two index selects, v_med3).  Left to the
// compiler, the nested selects of common.h's insert() inside featknn.hip's candidate loop became exec-mask branches
// plus ~80 register copies per trip.  Slots are updated from the bottom up, so each reads its upper neighbour's
// OLD value; mask i (= key > v[i]) is computed two slots before its first use: on gfx950 a VALU read of an SGPR
// needs two other instructions after the VALU write, and the hazard recogniser does not look inside asm.  Two
// blocks because an asm statement takes at most 30 operands.
__device__ __forceinline__ void topk20_insert(TopK<20> &t, float key, int j)
{
    unsigned long long m0, m1, m2;
    asm volatile(
        "v_cmp_gt_f32_e64 %[m1], %[key], %[v19]\n\t"
        "v_cmp_gt_f32_e64 %[m0], %[key], %[v18]\n\t"
        "v_cmp_gt_f32_e64 %[m2], %[key], %[v17]\n\t"
        "v_cndmask_b32_e64 %[i19], %[i19], %[j], %[m1]\n\t"
        "v_med3_f32 %[v19], %[v18], %[v19], %[key]\n\t"
        "v_cndmask_b32_e64 %[i19], %[i19], %[i18], %[m0]\n\t"
        "v_cmp_gt_f32_e64 %[m1], %[key], %[v16]\n\t"
        "v_cndmask_b32_e64 %[i18], %[i18], %[j], %[m0]\n\t"
        "v_med3_f32 %[v18], %[v17], %[v18], %[key]\n\t"
        "v_cndmask_b32_e64 %[i18], %[i18], %[i17], %[m2]\n\t"
        "v_cmp_gt_f32_e64 %[m0], %[key], %[v15]\n\t"
        "v_cndmask_b32_e64 %[i17], %[i17], %[j], %[m2]\n\t"
        "v_med3_f32 %[v17], %[v16], %[v17], %[key]\n\t"
        "v_cndmask_b32_e64 %[i17], %[i17], %[i16], %[m1]\n\t"
        "v_cmp_gt_f32_e64 %[m2], %[key], %[v14]\n\t"
        "v_cndmask_b32_e64 %[i16], %[i16], %[j], %[m1]\n\t"
        "v_med3_f32 %[v16], %[v15], %[v16], %[key]\n\t"
        "v_cndmask_b32_e64 %[i16], %[i16], %[i15], %[m0]\n\t"
        "v_cmp_gt_f32_e64 %[m1], %[key], %[v13]\n\t"
        "v_cndmask_b32_e64 %[i15], %[i15], %[j], %[m0]\n\t"
        "v_med3_f32 %[v15], %[v14], %[v15], %[key]\n\t"
        "v_cndmask_b32_e64 %[i15], %[i15], %[i14], %[m2]\n\t"
        "v_cmp_gt_f32_e64 %[m0], %[key], %[v12]\n\t"
        "v_cndmask_b32_e64 %[i14], %[i14], %[j], %[m2]\n\t"
        "v_med3_f32 %[v14], %[v13], %[v14], %[key]\n\t"
        "v_cndmask_b32_e64 %[i14], %[i14], %[i13], %[m1]\n\t"
        "v_cmp_gt_f32_e64 %[m2], %[key], %[v11]\n\t"
        "v_cndmask_b32_e64 %[i13], %[i13], %[j], %[m1]\n\t"
        "v_med3_f32 %[v13], %[v12], %[v13], %[key]\n\t"
        "v_cndmask_b32_e64 %[i13], %[i13], %[i12], %[m0]\n\t"
        "v_cmp_gt_f32_e64 %[m1], %[key], %[v10]\n\t"
        "v_cndmask_b32_e64 %[i12], %[i12], %[j], %[m0]\n\t"
        "v_med3_f32 %[v12], %[v11], %[v12], %[key]\n\t"
        "v_cndmask_b32_e64 %[i12], %[i12], %[i11], %[m2]\n\t"
        "v_cmp_gt_f32_e64 %[m0], %[key], %[v9]\n\t"
        "v_cndmask_b32_e64 %[i11], %[i11], %[j], %[m2]\n\t"
        "v_med3_f32 %[v11], %[v10], %[v11], %[key]\n\t"
        "v_cndmask_b32_e64 %[i11], %[i11], %[i10], %[m1]\n\t"
        "v_cndmask_b32_e64 %[i10], %[i10], %[j], %[m1]\n\t"
        "v_med3_f32 %[v10], %[v9], %[v10], %[key]\n\t"
        "v_cndmask_b32_e64 %[i10], %[i10], %[i9], %[m0]\n\t"
        : [v10] "+v"(t.v[10]), [v11] "+v"(t.v[11]), [v12] "+v"(t.v[12]), [v13] "+v"(t.v[13]), [v14] "+v"(t.v[14]), [v15] "+v"(t.v[15]), [v16] "+v"(t.v[16]), [v17] "+v"(t.v[17]), [v18] "+v"(t.v[18]), [v19] "+v"(t.v[19]), [i10] "+v"(t.id[10]), [i11] "+v"(t.id[11]), [i12] "+v"(t.id[12]), [i13] "+v"(t.id[13]), [i14] "+v"(t.id[14]), [i15] "+v"(t.id[15]), [i16] "+v"(t.id[16]), [i17] "+v"(t.id[17]), [i18] "+v"(t.id[18]), [i19] "+v"(t.id[19]), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2)
        : [v9] "v"(t.v[9]), [i9] "v"(t.id[9]), [key] "v"(key), [j] "v"(j));
    asm volatile(
        "v_cmp_gt_f32_e64 %[m0], %[key], %[v9]\n\t"
        "v_cmp_gt_f32_e64 %[m2], %[key], %[v8]\n\t"
        "v_cmp_gt_f32_e64 %[m1], %[key], %[v7]\n\t"
        "v_cndmask_b32_e64 %[i9], %[i9], %[j], %[m0]\n\t"
        "v_med3_f32 %[v9], %[v8], %[v9], %[key]\n\t"
        "v_cndmask_b32_e64 %[i9], %[i9], %[i8], %[m2]\n\t"
        "v_cmp_gt_f32_e64 %[m0], %[key], %[v6]\n\t"
        "v_cndmask_b32_e64 %[i8], %[i8], %[j], %[m2]\n\t"
        "v_med3_f32 %[v8], %[v7], %[v8], %[key]\n\t"
        "v_cndmask_b32_e64 %[i8], %[i8], %[i7], %[m1]\n\t"
        "v_cmp_gt_f32_e64 %[m2], %[key], %[v5]\n\t"
        "v_cndmask_b32_e64 %[i7], %[i7], %[j], %[m1]\n\t"
        "v_med3_f32 %[v7], %[v6], %[v7], %[key]\n\t"
        "v_cndmask_b32_e64 %[i7], %[i7], %[i6], %[m0]\n\t"
        "v_cmp_gt_f32_e64 %[m1], %[key], %[v4]\n\t"
        "v_cndmask_b32_e64 %[i6], %[i6], %[j], %[m0]\n\t"
        "v_med3_f32 %[v6], %[v5], %[v6], %[key]\n\t"
        "v_cndmask_b32_e64 %[i6], %[i6], %[i5], %[m2]\n\t"
        "v_cmp_gt_f32_e64 %[m0], %[key], %[v3]\n\t"
        "v_cndmask_b32_e64 %[i5], %[i5], %[j], %[m2]\n\t"
        "v_med3_f32 %[v5], %[v4], %[v5], %[key]\n\t"
        "v_cndmask_b32_e64 %[i5], %[i5], %[i4], %[m1]\n\t"
        "v_cmp_gt_f32_e64 %[m2], %[key], %[v2]\n\t"
        "v_cndmask_b32_e64 %[i4], %[i4], %[j], %[m1]\n\t"
        "v_med3_f32 %[v4], %[v3], %[v4], %[key]\n\t"
        "v_cndmask_b32_e64 %[i4], %[i4], %[i3], %[m0]\n\t"
        "v_cmp_gt_f32_e64 %[m1], %[key], %[v1]\n\t"
        "v_cndmask_b32_e64 %[i3], %[i3], %[j], %[m0]\n\t"
        "v_med3_f32 %[v3], %[v2], %[v3], %[key]\n\t"
        "v_cndmask_b32_e64 %[i3], %[i3], %[i2], %[m2]\n\t"
        "v_cmp_gt_f32_e64 %[m0], %[key], %[v0]\n\t"
        "v_cndmask_b32_e64 %[i2], %[i2], %[j], %[m2]\n\t"
        "v_med3_f32 %[v2], %[v1], %[v2], %[key]\n\t"
        "v_cndmask_b32_e64 %[i2], %[i2], %[i1], %[m1]\n\t"
        "v_cndmask_b32_e64 %[i1], %[i1], %[j], %[m1]\n\t"
        "v_med3_f32 %[v1], %[v0], %[v1], %[key]\n\t"
        "v_cndmask_b32_e64 %[i1], %[i1], %[i0], %[m0]\n\t"
        "v_cndmask_b32_e64 %[i0], %[i0], %[j], %[m0]\n\t"
        "v_max_f32 %[v0], %[v0], %[key]\n\t"
        : [v0] "+v"(t.v[0]), [v1] "+v"(t.v[1]), [v2] "+v"(t.v[2]), [v3] "+v"(t.v[3]), [v4] "+v"(t.v[4]), [v5] "+v"(t.v[5]), [v6] "+v"(t.v[6]), [v7] "+v"(t.v[7]), [v8] "+v"(t.v[8]), [v9] "+v"(t.v[9]), [i0] "+v"(t.id[0]), [i1] "+v"(t.id[1]), [i2] "+v"(t.id[2]), [i3] "+v"(t.id[3]), [i4] "+v"(t.id[4]), [i5] "+v"(t.id[5]), [i6] "+v"(t.id[6]), [i7] "+v"(t.id[7]), [i8] "+v"(t.id[8]), [i9] "+v"(t.id[9]), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2)
        : [key] "v"(key), [j] "v"(j));
}


// Activation code of the conv entry points (their `relu` argument): 0 none, 1 ReLU, any value > 1 = the
// IEEE-754 bits of a LeakyReLU negative slope in (0, 1) (models/prnet.py:79 uses 0.2).  Call only if act != 0.
#ifdef __HIPCC__
__device__ __forceinline__ float l3d_act(float v, int act)
{
    return act == 1 ? fmaxf(v, 0.f) : fmaxf(v, v * __int_as_float(act));
}
#endif

// max over aligned groups of `span` (2..32, power of two) consecutive lanes, every lane of a group receiving it.
// The first four levels are DPP modifiers on the VALU (quad_perm, row_half_mirror, row_mirror: no LDS crossbar, no
// wait) -- the pooled conv epilogues ran 4-5 ds_bpermute round trips per accumulator value before and spent more
// time there than in writing the un-pooled tensor; only the 16 <-> 16 exchange of span = 32 still goes through
// ds_bpermute.
#ifdef __HIPCC__
#define L3D_DPP_FMAX(v, CTRL) fmaxf((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (v)), __builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, false)))
__device__ __forceinline__ float l3d_group_max(float v, int span)
{
    if (span >= 2) v = L3D_DPP_FMAX(v, 0xB1);        // quad_perm(1,0,3,2)
    if (span >= 4) v = L3D_DPP_FMAX(v, 0x4E);        // quad_perm(2,3,0,1)
    if (span >= 8) v = L3D_DPP_FMAX(v, 0x141);       // row_half_mirror
    if (span >= 16) v = L3D_DPP_FMAX(v, 0x140);      // row_mirror
    if (span >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    return v;
}

#ifdef __HIPCC__
// Stage `tn` packed xyz points (cbase[3 t + c]) into LDS through `put(t, x, y, z)` with `nthreads` threads, UNR points per thread in
// flight before the first LDS write.  A rolled `for (t = tid; t < tn; t += nthreads) { load; put; }` loop compiles to one round trip
// to memory per trip (load, wait, write): at 64 threads and a 2048-point tile that is 32 dependent round trips per tile.  The loads
// are unconditional, from indices clamped into the tile (a conditional load is a branch with a wait at its join); only `put` is
// predicated.
template <int UNR, typename Put>
__device__ __forceinline__ void l3d_stage_points(const float *__restrict__ cbase, int tn, int tid, int nthreads, Put put)
{
    for (int tb = 0; tb < tn; tb += UNR * nthreads) {
        float x[UNR], y[UNR], z[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const float *cp = cbase + (size_t)min(tb + u * nthreads + tid, tn - 1) * 3;
            x[u] = cp[0]; y[u] = cp[1]; z[u] = cp[2];
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int t = tb + u * nthreads + tid;
            if (t < tn) put(t, x[u], y[u], z[u]);
        }
    }
}
#endif

#endif
