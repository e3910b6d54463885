// sa_fused.hip -- a whole set-abstraction layer behind the ball query in ONE kernel: gather [xyz[idx] - centre | feat[idx]],
// three shared-MLP layers (1x1 conv + folded BatchNorm + ReLU) and the max over the K neighbours, for the NARROW stacks of a
// first set-abstraction level (reference models/flownet3d.py:108-122 PointNetSetAbstraction with sa1 = 3+3 -> 32 -> 32 -> 64,
// K 16, flownet3d.py:272; pointnet2's ssg levels 3+D -> 64 -> 64 -> 128).  BASELINE configs[4] ran these as a grouping kernel plus
// three conv launches (148 us per 32 clouds x 1024 centroids, 280 MB of [B,C,S,K] activations through HBM); here the activations
// never leave the CU.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 -- an exact fp32 fma chain per output, ascending input channel; y = relu(acc * scale + shift)
// unfused like the layer kernels.  The wide stacks of the deeper levels (67 -> 64 -> 64 -> 128 and up) stay on the f16x2 / bf16x3
// GEMM kernels (models/flownet3d.py _f16_stack): this kernel is for stacks whose weights fit in LDS beside the activations.
//
// A workgroup (4 waves) owns 64 consecutive centroids of one cloud, a wave 16 of them; a wave walks its 16 K rows in tiles of 16
// rows (K = 8: two centroids per tile, 16: one, 32 / 64: two / four tiles per centroid with a running maximum).  Per tile:
//   layer 1   A operand = the gathered values themselves (lane = row m, channel 4 step + lane / 16): no staging
//   layer l   D (row 4 (lane / 16) + r, column lane % 16) -> act -> the wave's LDS tile T[row][(c % 4) (C / 4) + c / 4], so that the
//             next layer's A operand (row lane % 16, channels 4 s + lane / 16 for every s) is C / 16 ds_read_b128
//   weights   LDS, W[n][g][s] = w[n][4 s + g]: the B operand of a column tile is likewise ds_read_b128 runs over s
//   max       over the accumulator's 4 rows, then across the lane groups (xor 16, 32; K = 8: xor 16 only)
// The 64 x C3 maxima are staged in LDS and leave as 256-byte rows of out [B][C3][S].
#include "common.h"
#include "split_bf16.h"          // f32x4

#define SA_CENT 64               // centroids per workgroup

// Where channel c of row `row` sits in a wave's [16 rows x C] activation tile: the next layer's A operand of lane (m = row, g) is the
// channels 4 s + g, s = 0 .. C / 4 - 1, which it reads as C / 16 float4 runs over s -- so the tile is [g][s / 4][row'][s % 4] with
// row' = row rotated within its group of four by g.  Both access patterns are then conflict-free: a 16-lane phase of the b128 read
// (g fixed, m = 0 .. 15) touches 16 consecutive float4, and the 64 lanes of a b32 write (4 rows x 16 channels: bank = 16 (row / 4)
// + 4 ((row + g) % 4) + s % 4) touch 64 different banks.  (The first layout, [row][g][s] with a padded pitch, measured 55 % of its
// LDS cycles as bank conflicts: profiles/round4_pmc_sa_mlp3.txt.)
template <int C>
__device__ __forceinline__ int sa_pos(int row, int c)
{
    const int g = c & 3, s = c >> 2;
    return ((g * (C / 16) + (s >> 2)) * 16 + ((row & ~3) | ((row + g) & 3))) * 4 + (s & 3);
}
// ... and the float4 run q of lane (m, g)
template <int C>
__device__ __forceinline__ int sa_run(int m, int g, int q) { return ((g * (C / 16) + q) * 16 + ((m & ~3) | ((m + g) & 3))) * 4; }

// one layer on a 16-row tile: A fragments a[C_IN / 4] (channel 4 s + lane / 16 of row lane % 16), weights at w (LDS,
// [4][C_IN / 4 / RUN][CP][RUN] with RUN = min(4, C_IN / 4) and CP = C_OUT (+ 16 when RUN == 2): element (n, k = 4 s + g) at
// ((g (NS / RUN) + s / RUN) CP + n) RUN + s % RUN -- consecutive output channels are consecutive 16-byte (8-byte) cells, so the B
// operand of a column tile is conflict-free ds_read_b128 (b64) runs over s), scale / shift at ss (LDS, [2][C_OUT]);
// result v[ct][r] = act value of row 4 (lane / 16) + r, column 16 ct + lane % 16
template <int C_IN, int C_OUT, int SA_TT>
__device__ __forceinline__ void sa_layer(const float (*a)[C_IN / 4], const float *__restrict__ w, const float *__restrict__ ss, int lane,
                                         float (*v)[C_OUT / 16][4])
{
    constexpr int NS = C_IN / 4, NCT = C_OUT / 16;
    const int g = lane >> 4, n = lane & 15;
    f32x4 acc[SA_TT][NCT];
#pragma unroll
    for (int tt = 0; tt < SA_TT; tt++)
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) acc[tt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += 4) {
        constexpr int RUN = NS < 4 ? NS : 4, CP = C_OUT + (RUN == 2 ? 16 : 0);
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) {
            float b[RUN];
            const float *wp = w + ((g * (NS / RUN) + s0 / RUN) * CP + 16 * ct + n) * RUN;
            if constexpr (RUN == 4) {
                const float4 q = *(const float4 *)wp;
                b[0] = q.x; b[1] = q.y; b[2] = q.z; b[3] = q.w;
            } else {
                const float2 q = *(const float2 *)wp;
                b[0] = q.x; b[1] = q.y;
            }
            // the row tiles share the weight fragment: SA_TT NCT independent accumulation chains keep the matrix pipe fed (a chain's
            // next MFMA waits for its previous one's eight passes)
#pragma unroll
            for (int i = 0; i < RUN; i++)
#pragma unroll
                for (int tt = 0; tt < SA_TT; tt++)
                    acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt][s0 + i], b[i], acc[tt][ct], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
        const float sc = ss[16 * ct + n], sh = ss[C_OUT + 16 * ct + n];
#pragma unroll
        for (int tt = 0; tt < SA_TT; tt++)
#pragma unroll
            for (int r = 0; r < 4; r++) v[tt][ct][r] = fmaxf(acc[tt][ct][r] * sc + sh, 0.f);
    }
}

// params (device, floats): W1 | ss1 [2][C1] | W2 | ss2 [2][C2] | W3 | ss3 [2][C3], the weights in sa_layer's LDS order (W1 of a
// C0P = 8 stack carries 16 pad rows per lane group: sa_w1_floats)
template <int C0P, int C1>
constexpr int sa_w1_floats() { return C0P == 8 ? 4 * (C1 + 16) * 2 : C1 * C0P; }

// SA_TT = 16-row tiles a wave works on at once (1 or 2; K, the tile count of a wave, is even): 2 for the 32-32-64 stack, whose layers
// have only two column tiles (two accumulation chains do not keep the matrix pipe fed: 64 -> 56 us at config 5); 1 for 64-64-128,
// which would not fit two waves per SIMD otherwise
template <int C0P, int C1, int C2, int C3, int SA_TT = (C1 <= 32 ? 2 : 1)>
__global__ __launch_bounds__(256) void sa_mlp3_kernel(const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                      const float *__restrict__ feat, const int *__restrict__ idx,
                                                      const float *__restrict__ params, int N, int S, int K, int D,
                                                      float *__restrict__ out)
{
    constexpr int NPAR = sa_w1_floats<C0P, C1>() + 2 * C1 + C2 * C1 + 2 * C2 + C3 * C2 + 2 * C3;
    constexpr int CM = C1 > C2 ? C1 : C2;                             // activation tile: 16 rows x CM floats
    extern __shared__ __attribute__((aligned(16))) float sa_lds[];
    float *par = sa_lds;
    float *tiles = sa_lds + ((NPAR + 3) & ~3);                         // [4 waves][SA_TT][16 CM]
    float *stage = tiles + 4 * SA_TT * 16 * CM;                        // [C3][SA_CENT + 1]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.y, s0 = blockIdx.x * SA_CENT;
    for (int i = t; i < NPAR / 4; i += 256) ((float4 *)par)[i] = ((const float4 *)params)[i];
    __syncthreads();
    const float *w1 = par, *ss1 = w1 + sa_w1_floats<C0P, C1>(), *w2 = ss1 + 2 * C1, *ss2 = w2 + C2 * C1, *w3 = ss2 + 2 * C2, *ss3 = w3 + C3 * C2;
    float *T = tiles + wave * SA_TT * 16 * CM;
    const int g = lane >> 4, m = lane & 15;
    const float *cloud = xyz + (size_t)b * N * 3;
    const float *fb = feat ? feat + (size_t)b * D * N : nullptr;

    float run[C3 / 16];
#pragma unroll
    for (int ct = 0; ct < C3 / 16; ct++) run[ct] = -INFINITY;
    const int ntile = K;                                               // 16 centroids x K rows / 16 (even)
    // The gather runs ahead of the MFMAs: the next pair of tiles' values and the pair after that's neighbour indices (two dependent
    // trips to memory) are in flight while this pair is computed.  One load per value from a SELECTED address and an arithmetic
    // select on what comes back: a `?:` between loaded values or pointers becomes exec-masked branches with a full wait behind
    // each (LABLOG R4.1, R4.11) -- offsets in floats relative to the cloud, opaque to the optimiser.
    auto tile_index = [&](int j, int &sg) {
        const int R = 16 * min(j, ntile - 1) + m, cs = R / K, kk = R - cs * K;
        sg = min(s0 + wave * 16 + cs, S - 1);                          // clamped: a partial last workgroup computes on the last centroid
        return idx[((size_t)b * S + sg) * K + kk];
    };
    const long fdelta = fb ? (long)(((intptr_t)fb - (intptr_t)cloud) / 4) : 0;
    float fx[C0P / 4], fv[C0P / 4];                                    // 1 where the lane's channel is a coordinate / is anything at all
#pragma unroll
    for (int st = 0; st < C0P / 4; st++) {
        const int c = 4 * st + g;
        fx[st] = c < 3 ? 1.f : 0.f;
        fv[st] = c < 3 + D ? 1.f : 0.f;
    }
    auto tile_gather = [&](int nb, int sg, float *a) {
#pragma unroll
        for (int st = 0; st < C0P / 4; st++) {
            const int c = 4 * st + g;
            long off = c < 3 ? (long)nb * 3 + c : (c - 3 < D ? fdelta + (long)(c - 3) * N + nb : 0);
            long coff = ((long)b * S + sg) * 3 + min(c, 2);
            asm volatile("" : "+v"(off), "+v"(coff));
            a[st] = (cloud[off] - new_xyz[coff] * fx[st]) * fv[st];   // exact: x 1 and - 0 change nothing; unused channels x 0
        }
    };
    float a0[SA_TT][C0P / 4], a_nxt[SA_TT][C0P / 4];
    int nb_nxt[SA_TT], sg_nxt[SA_TT];
#pragma unroll
    for (int tt = 0; tt < SA_TT; tt++) {
        int sg;
        const int nb = tile_index(tt, sg);
        tile_gather(nb, sg, a0[tt]);
        nb_nxt[tt] = tile_index(SA_TT + tt, sg_nxt[tt]);
    }
    for (int j = 0; j < ntile; j += SA_TT) {
        int nb_n2[SA_TT], sg_n2[SA_TT];
#pragma unroll
        for (int tt = 0; tt < SA_TT; tt++) {
            tile_gather(nb_nxt[tt], sg_nxt[tt], a_nxt[tt]);            // the next pair's values
            nb_n2[tt] = tile_index(j + 2 * SA_TT + tt, sg_n2[tt]);     // the pair after that's indices
        }
        // ---- layer 1
        float v1[SA_TT][C1 / 16][4];
        sa_layer<C0P, C1, SA_TT>(a0, w1, ss1, lane, v1);
#pragma unroll
        for (int tt = 0; tt < SA_TT; tt++)
#pragma unroll
            for (int ct = 0; ct < C1 / 16; ct++)
#pragma unroll
                for (int r = 0; r < 4; r++) T[tt * 16 * CM + sa_pos<C1>(4 * g + r, 16 * ct + m)] = v1[tt][ct][r];
        // ---- layer 2
        float a1[SA_TT][C1 / 4];
#pragma unroll
        for (int tt = 0; tt < SA_TT; tt++)
#pragma unroll
            for (int q = 0; q < C1 / 16; q++) {
                const float4 x = *(const float4 *)&T[tt * 16 * CM + sa_run<C1>(m, g, q)];
                a1[tt][4 * q] = x.x; a1[tt][4 * q + 1] = x.y; a1[tt][4 * q + 2] = x.z; a1[tt][4 * q + 3] = x.w;
            }
        float v2[SA_TT][C2 / 16][4];
        sa_layer<C1, C2, SA_TT>(a1, w2, ss2, lane, v2);
#pragma unroll
        for (int tt = 0; tt < SA_TT; tt++)
#pragma unroll
            for (int ct = 0; ct < C2 / 16; ct++)
#pragma unroll
                for (int r = 0; r < 4; r++) T[tt * 16 * CM + sa_pos<C2>(4 * g + r, 16 * ct + m)] = v2[tt][ct][r];
        // ---- layer 3 and the maximum over the rows of a centroid
        float a2[SA_TT][C2 / 4];
#pragma unroll
        for (int tt = 0; tt < SA_TT; tt++)
#pragma unroll
            for (int q = 0; q < C2 / 16; q++) {
                const float4 x = *(const float4 *)&T[tt * 16 * CM + sa_run<C2>(m, g, q)];
                a2[tt][4 * q] = x.x; a2[tt][4 * q + 1] = x.y; a2[tt][4 * q + 2] = x.z; a2[tt][4 * q + 3] = x.w;
            }
        float v3[SA_TT][C3 / 16][4];
        sa_layer<C2, C3, SA_TT>(a2, w3, ss3, lane, v3);
#pragma unroll
        for (int tt = 0; tt < SA_TT; tt++) {
            const int jj = j + tt;
            const bool last = K <= 16 || ((16 * jj + 16) % K) == 0;    // this tile ends its centroid(s)
#pragma unroll
            for (int ct = 0; ct < C3 / 16; ct++) {
                float mx = fmaxf(fmaxf(v3[tt][ct][0], v3[tt][ct][1]), fmaxf(v3[tt][ct][2], v3[tt][ct][3]));
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                if (K >= 16) mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                run[ct] = fmaxf(run[ct], mx);
                if (last) {
                    if (K >= 16) {
                        if (g == 0) stage[(16 * ct + m) * (SA_CENT + 1) + wave * 16 + (16 * jj) / K] = run[ct];
                    } else if ((g & 1) == 0) {                         // K = 8: lane groups 0, 1 hold centroid 2 j, groups 2, 3 centroid 2 j + 1
                        stage[(16 * ct + m) * (SA_CENT + 1) + wave * 16 + 2 * jj + (g >> 1)] = run[ct];
                    }
                    run[ct] = -INFINITY;
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < SA_TT; tt++) {
#pragma unroll
            for (int st = 0; st < C0P / 4; st++) a0[tt][st] = a_nxt[tt][st];
            nb_nxt[tt] = nb_n2[tt];
            sg_nxt[tt] = sg_n2[tt];
        }
    }
    __syncthreads();
    float *ob = out + (size_t)b * C3 * S;
    for (int e = t; e < C3 * SA_CENT; e += 256) {
        const int c = e / SA_CENT, i = e % SA_CENT;
        if (s0 + i < S) ob[(size_t)c * S + s0 + i] = stage[c * (SA_CENT + 1) + i];
    }
}

template <int C0P, int C1, int C2, int C3>
static int sa_launch(const float *xyz, const float *new_xyz, const float *feat, const int *idx, const float *params, int B, int N, int S,
                     int K, int D, float *out, hipStream_t st)
{
    constexpr int NPAR = sa_w1_floats<C0P, C1>() + 2 * C1 + C2 * C1 + 2 * C2 + C3 * C2 + 2 * C3;
    constexpr int CM = C1 > C2 ? C1 : C2, SA_TT = C1 <= 32 ? 2 : 1;
    const size_t lds = (size_t)(((NPAR + 3) & ~3) + 4 * SA_TT * 16 * CM + C3 * (SA_CENT + 1)) * sizeof(float);
    hipLaunchKernelGGL((sa_mlp3_kernel<C0P, C1, C2, C3>), dim3((unsigned)l3d_divup(S, SA_CENT), (unsigned)B), dim3(256), lds, st, xyz, new_xyz,
                       feat, idx, params, N, S, K, D, out);
    return l3d_check_launch();
}

// floats of the parameter block for (C0 = 3 + D input channels, widths C1, C2, C3); 0 when the kernel does not take the stack
static size_t l3d_sa_mlp3_param_floats(int D, int C1, int C2, int C3)
{
    const int c0 = 3 + D;
    if (D < 0 || c0 > 16) return 0;
    const bool a = C1 == 32 && C2 == 32 && C3 == 64, b = C1 == 64 && C2 == 64 && C3 == 128;
    if (!a && !b) return 0;
    const int c0p = c0 <= 8 ? 8 : 16;
    return (size_t)(c0p == 8 ? 4 * (C1 + 16) * 2 : C1 * c0p) + 2 * C1 + (size_t)C2 * C1 + 2 * C2 + (size_t)C3 * C2 + 2 * C3;
}

// out [B][C3][S] = max_k relu(s3 (W3 relu(s2 (W2 relu(s1 (W1 [xyz[idx] - new_xyz | feat[idx]]) + t1)) + t2)) + t3)
// xyz [B][N][3], new_xyz [B][S][3], feat [B][D][N] (NULL when D == 0), idx int32 [B][S][K] (K in {8, 16, 32, 64});
// params: per layer the weights in the order sa_layer reads them -- element w[n][k = 4 s + g] at ((g (NS / RUN) + s / RUN) CP + n) RUN
// + s % RUN, NS = Cin / 4, RUN = min(4, NS), CP = Cout (+ 16 zero rows when RUN == 2), layer 1's input channels zero-padded to 8 or
// 16 -- followed by scale [Cout] and shift [Cout] (the folded BatchNorm).
extern "C" int l3d_sa_mlp3_fused(const float *xyz, const float *new_xyz, const float *feat, const int32_t *idx, const float *params,
                                 int B, int N, int S, int K, int D, int C1, int C2, int C3, float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && new_xyz && idx && params && out && (D == 0 || feat) && B > 0 && N > 0 && S > 0 && D >= 0);
    if (l3d_sa_mlp3_param_floats(D, C1, C2, C3) == 0 || (K != 8 && K != 16 && K != 32 && K != 64) || B > 65535 ||
        (((size_t)params) & 15))
        return L3D_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const bool small = 3 + D <= 8;
    if (C1 == 32) return small ? sa_launch<8, 32, 32, 64>(xyz, new_xyz, feat, idx, params, B, N, S, K, D, out, st)
                               : sa_launch<16, 32, 32, 64>(xyz, new_xyz, feat, idx, params, B, N, S, K, D, out, st);
    return small ? sa_launch<8, 64, 64, 128>(xyz, new_xyz, feat, idx, params, B, N, S, K, D, out, st)
                 : sa_launch<16, 64, 64, 128>(xyz, new_xyz, feat, idx, params, B, N, S, K, D, out, st);
}
