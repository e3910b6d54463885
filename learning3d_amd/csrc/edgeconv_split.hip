// edgeconv_split.hip -- the register-chained EdgeConv stack of edgeconv2.hip with layers 2-4 on the
// bf16 matrix cores ("bf16x3": every fp32 operand split exactly into three bf16 planes, six bf16
// MFMA products per fp32 product, fp32 accumulate -- see conv_split.hip for the error argument).
// models/dgcnn.py:32-46.
//
// Chaining with v_mfma_f32_16x16x32_bf16.  As in edgeconv2.hip every layer is computed transposed,
// D[ch][row] = sum_k W'[ch][k] act[row][k], weights = A operand, activations = B operand, one wave
// owns MT row tiles of 16 rows (4 points x 4*MT neighbours).  The accumulator layout is the same as
// for the fp32 MFMA: lane (j = row, g) register r holds channel 16m + 4g + r of M-tile m.  The bf16
// B operand of lane (j, g) is EIGHT k-slots for row j; k is only a summation index, so k-step s takes
//     slots 0..3 of lane group g  <->  input channel 16(2s)   + 4g + e   (M-tile 2s,   register e)
//     slots 4..7 of lane group g  <->  input channel 16(2s+1) + 4g + e   (M-tile 2s+1, register e)
// i.e. the B operand of k-step s is the pair of previous-layer accumulators (2s, 2s+1) of the same
// lane, after bias (initial accumulator value), ReLU and the exact three-way bf16 split: still no LDS,
// no barriers, no cross-lane traffic.  The A operand is pre-split and pre-permuted on the host
// (l3d_edgeconv_pack, third block) and streamed as 1 KB fragments, consumed strictly linearly.
//
// Loop order is M-tile-pair outer / k-step inner: only two M-tiles of accumulators (2 x MT x 4
// registers) are live, so the dominant register cost is the split input planes (layer 4: 128
// channels x 3 planes = 240 VGPRs for MT = 5) and the kernel fits one wave per SIMD.  Layer 1
// (6 -> 64, K padded to 8) stays on the fp32 MFMA: a K=32 bf16 step would be 3/4 padding.
#include "common.h"
#include "edgeconv_layout.h"
#include "split_bf16.h"

__device__ __forceinline__ float es_quad_max(float v)
{
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    return v;
}

// ReLU in place + max over the wave's neighbours of each point (all 4 lanes of a quad get it)
template <int MT>
__device__ __forceinline__ f32x4 es_relu_pool(f32x4 (&h)[MT])
{
    f32x4 mx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            h[t][r] = fmaxf(h[t][r], 0.f);
            mx[r] = fmaxf(mx[r], h[t][r]);
        }
#pragma unroll
    for (int r = 0; r < 4; r++) mx[r] = es_quad_max(mx[r]);
    return mx;
}

// Two finished M-tiles (2s, 2s+1) -> pooled output + (unless LAST) the three bf16 planes of k-step s
template <int MT, bool LAST>
__device__ __forceinline__ void es_finish_pair(f32x4 (&h0)[MT], f32x4 (&h1)[MT], uint4 (&pl)[3][MT],
                                               float *__restrict__ dst /* channel 16(2s) + 4g of this point */,
                                               bool writer)
{
    const f32x4 m0 = es_relu_pool<MT>(h0);
    const f32x4 m1 = es_relu_pool<MT>(h1);
    if (writer) {
        *(f32x4 *)dst = m0;
        *(f32x4 *)(dst + 16) = m1;
    }
    if (!LAST) {
#pragma unroll
        for (int t = 0; t < MT; t++) {
            split_pair(h0[t][0], h0[t][1], pl[0][t].x, pl[1][t].x, pl[2][t].x);
            split_pair(h0[t][2], h0[t][3], pl[0][t].y, pl[1][t].y, pl[2][t].y);
            split_pair(h1[t][0], h1[t][1], pl[0][t].z, pl[1][t].z, pl[2][t].z);
            split_pair(h1[t][2], h1[t][3], pl[0][t].w, pl[1][t].w, pl[2][t].w);
        }
    }
}

#define ES_BF(u) __builtin_bit_cast(bf16x8, (u))

// One output M-tile pair of a dense layer: 2 x S steps of {prefetch fragment step+2, 6 x MT MFMAs}.
template <int MT, int S, int NSTEP, bool LAST>
__device__ __forceinline__ void es_pair(int mp, const uint4 (&pin)[S][3][MT], uint4 (&po)[3][MT],
                                        const uint4 *__restrict__ wp, uint4 (&a0)[3], uint4 (&a1)[3],
                                        const float *__restrict__ bias, float *__restrict__ prow, bool writer, int g)
{
    f32x4 acc[2][MT];
#pragma unroll
    for (int mm = 0; mm < 2; mm++) {
        const f32x4 bv = *(const f32x4 *)(bias + 16 * (2 * mp + mm) + 4 * g);
#pragma unroll
        for (int t = 0; t < MT; t++) acc[mm][t] = bv;
    }
#pragma unroll
    for (int mm = 0; mm < 2; mm++)
#pragma unroll
        for (int s = 0; s < S; s++) {
            const int step = (2 * mp + mm) * S + s;
            const int nxt = step + 2 < NSTEP ? step + 2 : NSTEP - 1;
            uint4 a2[3];
#pragma unroll
            for (int p = 0; p < 3; p++) a2[p] = wp[(size_t)(nxt * 3 + p) * 64];
            __builtin_amdgcn_sched_barrier(0);
            // six products, smallest first; MT independent accumulators between dependent MFMAs
#pragma unroll
            for (int t = 0; t < MT; t++) acc[mm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ES_BF(a0[2]), ES_BF(pin[s][0][t]), acc[mm][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; t++) acc[mm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ES_BF(a0[0]), ES_BF(pin[s][2][t]), acc[mm][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; t++) acc[mm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ES_BF(a0[1]), ES_BF(pin[s][1][t]), acc[mm][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; t++) acc[mm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ES_BF(a0[1]), ES_BF(pin[s][0][t]), acc[mm][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; t++) acc[mm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ES_BF(a0[0]), ES_BF(pin[s][1][t]), acc[mm][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; t++) acc[mm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ES_BF(a0[0]), ES_BF(pin[s][0][t]), acc[mm][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 3; p++) { a0[p] = a1[p]; a1[p] = a2[p]; }
        }
    es_finish_pair<MT, LAST>(acc[0], acc[1], po, prow + 32 * mp, writer);
}

// One dense layer: S input k-steps (32 channels each, planes in pin), NPAIR output M-tile pairs.
// wl: [step = m*S + s][plane][lane] fragments, read strictly in order, prefetched two steps ahead.
template <int MT, int S, int NPAIR, bool LAST, bool UNROLL>
__device__ __forceinline__ void es_layer(const uint4 (&pin)[S][3][MT], uint4 (&pout)[LAST ? 1 : NPAIR][3][MT],
                                         const uint4 *__restrict__ wl, const float *__restrict__ bias,
                                         float *__restrict__ prow, bool writer, int lane, int g)
{
    constexpr int NSTEP = NPAIR * 2 * S;
    const uint4 *wp = wl + lane;
    uint4 a0[3], a1[3];
#pragma unroll
    for (int p = 0; p < 3; p++) {
        a0[p] = wp[p * 64];
        a1[p] = wp[(3 + p) * 64];
    }
    if (UNROLL) {
#pragma unroll
        for (int mp = 0; mp < NPAIR; mp++)
            es_pair<MT, S, NSTEP, LAST>(mp, pin, pout[LAST ? 0 : mp], wp, a0, a1, bias, prow, writer, g);
    } else {
#pragma unroll 1
        for (int mp = 0; mp < NPAIR; mp++)
            es_pair<MT, S, NSTEP, LAST>(mp, pin, pout[0], wp, a0, a1, bias, prow, writer, g);
    }
}

template <int MT>
__global__ __launch_bounds__(256, 1) void edgeconv_split_kernel(const float *__restrict__ xyz,
                                                                const int64_t *__restrict__ idx, int N, int k,
                                                                const float *__restrict__ packed,
                                                                float *__restrict__ pooled /*[B*N][512]*/)
{
    constexpr int CTOT = EC_C1 + EC_C2 + EC_C3 + EC_C4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int n = (blockIdx.x * 4 + wave) * 4 + (j >> 2);          // this lane's point
    const int nc = min(n, N - 1);
    const bool writer = (n < N) && ((j & 3) == 0);
    float *prow = pooled + ((size_t)b * N + nc) * CTOT + 4 * g;

    // ---- layer 1 on the fp32 MFMA (as edgeconv2.hip): graph feature rows as B operands, k-step s,
    //      lane group g -> channel 4s + g of (neighbour xyz, centre xyz, 0, 0)          dgcnn.py:32
    const float *pc = xyz + ((size_t)b * N + nc) * 3;
    const float cx = pc[0], cy = pc[1], cz = pc[2];
    float b1[MT][2];
#pragma unroll
    for (int t = 0; t < MT; t++) {
        const int jj = 4 * t + (j & 3);
        const int64_t nb = idx[((size_t)b * N + nc) * k + (jj < k ? jj : 0)];   // pad k up to 4*MT with a duplicate
        const float *pn = xyz + ((size_t)b * N + nb) * 3;
        const float nx = pn[0], ny = pn[1], nz = pn[2];
        b1[t][0] = g == 0 ? nx : (g == 1 ? ny : (g == 2 ? nz : cx));
        b1[t][1] = g == 0 ? cy : (g == 1 ? cz : 0.f);
    }
    uint4 p1[EC_C1 / 32][3][MT];
    {
        const f32x2 *w1 = (const f32x2 *)(packed + EC2_OFF_W1);
#pragma unroll
        for (int mp = 0; mp < EC_C1 / 32; mp++) {
            f32x4 h[2][MT];
#pragma unroll
            for (int mm = 0; mm < 2; mm++) {
                const int m = 2 * mp + mm;
                const f32x4 bv = *(const f32x4 *)(packed + EC_OFF_B1 + 16 * m + 4 * g);
                const f32x2 a = w1[m * 64 + lane];
#pragma unroll
                for (int t = 0; t < MT; t++) h[mm][t] = bv;
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int t = 0; t < MT; t++)
                        h[mm][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[t][s], h[mm][t], 0, 0, 0);
            }
            es_finish_pair<MT, false>(h[0], h[1], p1[mp], prow + 32 * mp, writer);
        }
    }

    // ---- layer 2: 64 -> 64
    uint4 p2[EC_C2 / 32][3][MT];
    es_layer<MT, EC_C1 / 32, EC_C2 / 32, false, true>(p1, p2, (const uint4 *)(packed + EC3_OFF_W2), packed + EC_OFF_B2,
                                                      prow + EC_C1, writer, lane, g);
    // ---- layer 3: 64 -> 128
    uint4 p3[EC_C3 / 32][3][MT];
    es_layer<MT, EC_C2 / 32, EC_C3 / 32, false, true>(p2, p3, (const uint4 *)(packed + EC3_OFF_W3), packed + EC_OFF_B3,
                                                      prow + EC_C1 + EC_C2, writer, lane, g);
    // ---- layer 4: 128 -> 256, only max-pooled
    uint4 dummy[1][3][MT];
    es_layer<MT, EC_C3 / 32, EC_C4 / 32, true, false>(p3, dummy, (const uint4 *)(packed + EC3_OFF_W4), packed + EC_OFF_B4,
                                                      prow + EC_C1 + EC_C2 + EC_C3, writer, lane, g);
}

extern "C" int l3d_edgeconv_forward_split(const float *xyz, const int64_t *idx, int B, int N, int k,
                                          const float *packed, float *pooled, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && idx && packed && pooled && B > 0 && N > 0 && k > 0);
    if (k > 20 || B > 65535 || (((size_t)packed) & 15)) return L3D_ERR_UNSUPPORTED;
    dim3 grid(l3d_divup(N, 16), B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (k <= 8)       hipLaunchKernelGGL(edgeconv_split_kernel<2>, grid, block, 0, st, xyz, idx, N, k, packed, pooled);
    else if (k <= 16) hipLaunchKernelGGL(edgeconv_split_kernel<4>, grid, block, 0, st, xyz, idx, N, k, packed, pooled);
    else              hipLaunchKernelGGL(edgeconv_split_kernel<5>, grid, block, 0, st, xyz, idx, N, k, packed, pooled);
    return l3d_check_launch();
}
