// edgeconv2.hip -- the 4-layer EdgeConv stack of models/dgcnn.py:32-46 with the activations kept
// in REGISTERS from the graph-feature gather to the pooled output: no LDS, no barriers.
//
// Idea.  Write every layer transposed:  D[ch][row] = sum_k W'[ch][k] * act[row][k]  with
// v_mfma_f32_16x16x4_f32 taking the WEIGHTS as the A operand (lane (i,g): W'[16m+i][k(g)]) and the
// ACTIVATIONS as the B operand (lane (j,g): act[row 16t+j][k(g)]).  The accumulator layout of that
// instruction puts D[ch = 16m + 4g + r][row = 16t + j] in register r of lane (j,g) -- i.e. lane group
// g of row j holds channels 16m+4g .. 16m+4g+3.  The k index inside an MFMA is only a summation
// index, so the NEXT layer may enumerate its input channels in any order as long as A and B agree:
// choose k-step (q,e) <-> lane group g supplies input channel 16q + 4g + e.  Then the B operand of
// k-step (q,e) for row tile t is EXACTLY accumulator register e of the previous layer's (M-tile q,
// row tile t) -- after bias (folded into the accumulator's initial value) and ReLU (one v_max) --
// and the matching A operand is a plain float4 of 4 consecutive input channels of one output row.
// The transposition that forces the activations through LDS in edgeconv_kernel (mlp.hip) vanishes.
//
// One wave64 owns 4 points x 20 neighbours = 5 row tiles; rows are ordered
//     row 16t + j  <->  (point j>>2, neighbour 4t + (j&3))
// so max over a point's neighbours = 4 v_max across the 5 row tiles + 2 DPP quad-permute v_max.
// Waves are independent; a 256-thread workgroup is just 4 of them on consecutive points.
// Register budget (MT = row tiles = 5): h3 160 + layer-4 accumulators 80 + weight double-buffer 32
// -> one wave per SIMD, which is all an MFMA-bound stream with 20 independent accumulators needs.
#include "common.h"
#include "edgeconv_layout.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float quad_max(float v)
{
    // max over the 4 lanes of a quad: quad_perm(1,0,3,2) then quad_perm(2,3,0,1)
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    return v;
}

// ReLU in place and pooled (max over neighbours) float4 for one M-tile; all 4 lanes of a quad end
// up with the same value, lane (j&3)==0 stores it.
template <int MT>
__device__ __forceinline__ f32x4 relu_pool(f32x4 (&h)[MT])
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
    for (int r = 0; r < 4; r++) mx[r] = quad_max(mx[r]);
    return mx;
}

// One dense layer, NM output M-tiles [m0, m0+NM) written to out[MBASE..], input = NQ M-tiles of the
// previous layer (hin).  out must be pre-initialised with the bias.  wl: this layer's
// [m][q][lane][4] block; the next q-block of weights is prefetched while the current one multiplies.
template <int MT, int NQ, int NM, int NOUT, int MBASE>
__device__ __forceinline__ void chained_layer(const f32x4 (&hin)[NQ][MT], f32x4 (&out)[NOUT][MT],
                                              const f32x4 *__restrict__ wl, int m0, int lane)
{
    f32x4 wa[2][NM];
#pragma unroll
    for (int mm = 0; mm < NM; mm++) wa[0][mm] = wl[((size_t)(m0 + mm) * NQ) * 64 + lane];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int cur = q & 1;
        if (q + 1 < NQ) {
#pragma unroll
            for (int mm = 0; mm < NM; mm++) wa[cur ^ 1][mm] = wl[((size_t)(m0 + mm) * NQ + q + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the prefetch ahead of this block's MFMAs
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int mm = 0; mm < NM; mm++)
#pragma unroll
                for (int t = 0; t < MT; t++)
                    out[MBASE + mm][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cur][mm][e], hin[q][t][e], out[MBASE + mm][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NM, int NOUT, int MBASE, int MT>
__device__ __forceinline__ void init_bias(f32x4 (&out)[NOUT][MT], const float *__restrict__ bias, int m0, int g)
{
#pragma unroll
    for (int mm = 0; mm < NM; mm++) {
        const f32x4 bv = *(const f32x4 *)(bias + 16 * (m0 + mm) + 4 * g);
#pragma unroll
        for (int t = 0; t < MT; t++) out[MBASE + mm][t] = bv;
    }
}

template <int MT>
__global__ __launch_bounds__(256, 1) void edgeconv2_kernel(const float *__restrict__ xyz,
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

    // ---- graph feature rows as layer-1 B operands:  k-step s, lane group g -> channel 4s + g of
    //      (neighbour xyz, centre xyz, 0, 0)                               dgcnn.py:32
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

    // ---- layer 1: 6(8) -> 64
    f32x4 h1[EC_C1 / 16][MT];
    init_bias<EC_C1 / 16, EC_C1 / 16, 0>(h1, packed + EC_OFF_B1, 0, g);
    {
        const f32x2 *w1 = (const f32x2 *)(packed + EC2_OFF_W1);
#pragma unroll
        for (int m = 0; m < EC_C1 / 16; m++) {
            const f32x2 a = w1[m * 64 + lane];
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int t = 0; t < MT; t++)
                    h1[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[t][s], h1[m][t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < EC_C1 / 16; m++) {
        const f32x4 mx = relu_pool<MT>(h1[m]);
        if (writer) *(f32x4 *)(prow + 16 * m) = mx;
    }

    // ---- layer 2: 64 -> 64
    f32x4 h2[EC_C2 / 16][MT];
    init_bias<EC_C2 / 16, EC_C2 / 16, 0>(h2, packed + EC_OFF_B2, 0, g);
    chained_layer<MT, EC_C1 / 16, EC_C2 / 16, EC_C2 / 16, 0>(h1, h2, (const f32x4 *)(packed + EC2_OFF_W2), 0, lane);
#pragma unroll
    for (int m = 0; m < EC_C2 / 16; m++) {
        const f32x4 mx = relu_pool<MT>(h2[m]);
        if (writer) *(f32x4 *)(prow + EC_C1 + 16 * m) = mx;
    }

    // ---- layer 3: 64 -> 128, in two halves of 4 M-tiles
    f32x4 h3[EC_C3 / 16][MT];
    init_bias<4, EC_C3 / 16, 0>(h3, packed + EC_OFF_B3, 0, g);
    chained_layer<MT, EC_C2 / 16, 4, EC_C3 / 16, 0>(h2, h3, (const f32x4 *)(packed + EC2_OFF_W3), 0, lane);
    init_bias<4, EC_C3 / 16, 4>(h3, packed + EC_OFF_B3, 4, g);
    chained_layer<MT, EC_C2 / 16, 4, EC_C3 / 16, 4>(h2, h3, (const f32x4 *)(packed + EC2_OFF_W3), 4, lane);
#pragma unroll
    for (int m = 0; m < EC_C3 / 16; m++) {
        const f32x4 mx = relu_pool<MT>(h3[m]);
        if (writer) *(f32x4 *)(prow + EC_C1 + EC_C2 + 16 * m) = mx;
    }

    // ---- layer 4: 128 -> 256, four groups of 4 M-tiles; activations are only max-pooled
#pragma unroll 1
    for (int grp = 0; grp < EC_C4 / 64; grp++) {
        f32x4 acc[4][MT];
        init_bias<4, 4, 0>(acc, packed + EC_OFF_B4, 4 * grp, g);
        chained_layer<MT, EC_C3 / 16, 4, 4, 0>(h3, acc, (const f32x4 *)(packed + EC2_OFF_W4), 4 * grp, lane);
#pragma unroll
        for (int mm = 0; mm < 4; mm++) {
            const f32x4 mx = relu_pool<MT>(acc[mm]);
            if (writer) *(f32x4 *)(prow + EC_C1 + EC_C2 + EC_C3 + 16 * (4 * grp + mm)) = mx;
        }
    }
}

extern "C" int l3d_edgeconv_forward_chained(const float *xyz, const int64_t *idx, int B, int N, int k,
                                            const float *packed, float *pooled, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && idx && packed && pooled && B > 0 && N > 0 && k > 0);
    if (k > 20 || B > 65535) return L3D_ERR_UNSUPPORTED;
    dim3 grid(l3d_divup(N, 16), B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (k <= 8)       hipLaunchKernelGGL(edgeconv2_kernel<2>, grid, block, 0, st, xyz, idx, N, k, packed, pooled);
    else if (k <= 16) hipLaunchKernelGGL(edgeconv2_kernel<4>, grid, block, 0, st, xyz, idx, N, k, packed, pooled);
    else              hipLaunchKernelGGL(edgeconv2_kernel<5>, grid, block, 0, st, xyz, idx, N, k, packed, pooled);
    return l3d_check_launch();
}
