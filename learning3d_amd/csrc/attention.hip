// attention.hip -- scaled-dot-product attention of DCP's pointer network (utils/transformer.py:17-25,
// MultiHeadedAttention :120-147) as one flash-style kernel with fp32-level accuracy:
//     ctx[b][h*D + d][i] = sum_j softmax_j( scale * <q[b][h*D + :][i], k[b][h*D + :][j]> ) * v[b][h*D + d][j]
// q [B, H*D, N], k, v [B, H*D, M], ctx [B, H*D, N]: the CHANNEL-FIRST layout the 1x1-conv kernel
// produces for the projections, so heads are plain row ranges and nothing is transposed.  The
// [B,H,N,M] score tensor (537 MB at DCP's B=32, H=4, N=M=1024) is never materialised.  SURVEY.md 8(f)
// rank 1 (the SVDHead half is softcorr.hip).
//
// Both GEMMs run as bf16x3 on the bf16 matrix cores (split_bf16.h: six bf16 MFMA products per fp32
// product, fp32 accumulate):
//   S^T[j][i]  = sum_c K[c][j] Q[c][i]          keys on the MFMA row axis ("swapped QK^T"): a lane holds
//                                               16 keys of ONE query per 32x32 accumulator tile
//   O^T[d][i] += sum_j V[d][j] P^T[j][i]        A operand = V rows (keys contiguous in memory -- k and v
//                                               arrive channel-first, i.e. key-contiguous), B operand =
//                                               the probabilities, split three ways IN-LANE from the S
//                                               accumulators: k-step (a,u) takes registers 8u..8u+7 of
//                                               key tile a, which are keys 32a+16u+{4g..4g+3, 8+4g..8+4g+3}
//                                               for lane group g; V is staged in exactly that slot order.
// Workgroup = 128 queries of one (batch, head), 4 waves, each wave owns 32 queries x all 128 keys of a
// key tile (no cross-wave softmax / merge): S^T 4 tiles + O^T D/32 tiles = 128 accumulator registers.
// Per key tile: 8 chunk iterations of QK^T (K and Q chunks staged + split through double-buffered LDS,
// as conv_split.hip / softcorr.hip), online softmax in registers (log2 units, v_exp_f32; the two lanes
// l, l^32 that share a query column exchange their maxima), 8 k-step iterations of PV with V chunks
// staged through the same LDS buffers.
#include "common.h"
#include "split_bf16.h"

#define AT_TQ 128
#define AT_TK 128
#define AT_REG (128 * 16)                 // one (plane, kg) region: 128 rows x 8 bf16
#define AT_BUF (12 * AT_REG)              // K (or V) 6 regions + Q 6 regions
#define AT_LDS (2 * AT_BUF)
#define AT_NEG (-1.0e30f)

template <int ND /* D / 32 */>
__global__ __launch_bounds__(256) void attention_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                           const float *__restrict__ v, int H, int N, int M,
                                                           float scale, float *__restrict__ ctx, long q_bs, long k_bs,
                                                           long v_bs)
{
    constexpr int D = ND * 32;
    constexpr int NCH = D / 16;                        // QK^T chunks of 16 channels
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i0 = blockIdx.x * AT_TQ, h = blockIdx.y, b = blockIdx.z;
    // batch strides in floats: q, k, v may be channel slices of one fused projection output
    const float *qb = q + (size_t)b * q_bs + (size_t)h * D * N;
    const float *kb = k + (size_t)b * k_bs + (size_t)h * D * M;
    const float *vb = v + (size_t)b * v_bs + (size_t)h * D * M;

    // staging roles: thread -> (row, kg) octet.  QK^T phase: K row = key, Q row = query (channel-first:
    // 8 dword loads strided by M resp. N, coalesced over rows).  PV phase: V row = d (keys contiguous).
    const int srow = t & 127, skg = t >> 7;
    const int qn = min(i0 + srow, N - 1);
    const int st_lds = skg * AT_REG + srow * 16;      // + p * 2 * AT_REG (+ 6 * AT_REG for Q)

    const float sl2 = scale * 1.44269504088896340736f;
    float m_run = AT_NEG, l_run = 0.f;                 // this lane's query column: i0 + wave*32 + (lane&31)
    f32x16 o[ND];
#pragma unroll
    for (int dt = 0; dt < ND; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[dt][r] = 0.f;

    const int frag_kg = (lane >> 5) * AT_REG;
    const int a_off = frag_kg + (lane & 31) * 16;                              // + tile*512 + p*2*AT_REG
    const int b_off = 6 * AT_REG + frag_kg + (wave * 32 + (lane & 31)) * 16;   // + p*2*AT_REG

    for (int j0 = 0; j0 < M; j0 += AT_TK) {
        const int kn = min(j0 + srow, M - 1);
        // ------------------------------------------------------------ S^T = K^T Q over D channels
        f32x16 s[4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) s[a][r] = 0.f;
        float kv[8], qv[8];
#define AT_LOAD_QK(KC)                                                                               \
        do {                                                                                         \
            _Pragma("unroll") for (int e = 0; e < 8; e++) {                                          \
                kv[e] = kb[(size_t)((KC) * 16 + skg * 8 + e) * M + kn];                              \
                qv[e] = qb[(size_t)((KC) * 16 + skg * 8 + e) * N + qn];                              \
            }                                                                                        \
        } while (0)
#define AT_STORE_QK(BUF)                                                                             \
        do {                                                                                         \
            unsigned char *base_ = lds + (BUF) * AT_BUF;                                             \
            uint4 h_, m_, l_;                                                                        \
            split8(kv, h_, m_, l_);                                                                  \
            *(uint4 *)(base_ + st_lds) = h_;                                                         \
            *(uint4 *)(base_ + st_lds + 2 * AT_REG) = m_;                                            \
            *(uint4 *)(base_ + st_lds + 4 * AT_REG) = l_;                                            \
            split8(qv, h_, m_, l_);                                                                  \
            *(uint4 *)(base_ + 6 * AT_REG + st_lds) = h_;                                            \
            *(uint4 *)(base_ + 6 * AT_REG + st_lds + 2 * AT_REG) = m_;                               \
            *(uint4 *)(base_ + 6 * AT_REG + st_lds + 4 * AT_REG) = l_;                               \
        } while (0)

        __syncthreads();                               // the previous tile's PV reads of both buffers are done
        AT_LOAD_QK(0);
        AT_STORE_QK(0);
        __syncthreads();
#pragma unroll 1
        for (int kc = 0; kc < NCH; kc++) {
            const int buf = kc & 1;
            const bool more = kc + 1 < NCH;
            if (more) AT_LOAD_QK(kc + 1);
            const unsigned char *base = lds + buf * AT_BUF;
            bf16x8 Bf[3];
#pragma unroll
            for (int p = 0; p < 3; p++) Bf[p] = *(const bf16x8 *)(base + b_off + p * 2 * AT_REG);
#pragma unroll
            for (int pa = 2; pa >= 0; pa--) {         // key plane l, m, h; products smallest first
                bf16x8 A[4];
#pragma unroll
                for (int a = 0; a < 4; a++) A[a] = *(const bf16x8 *)(base + a_off + a * 512 + pa * 2 * AT_REG);
#pragma unroll
                for (int pb = 2; pb >= 0; pb--) {
                    if (pa + pb > 2) continue;
#pragma unroll
                    for (int a = 0; a < 4; a++) s[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a], Bf[pb], s[a], 0, 0, 0);
                }
            }
            if (more) AT_STORE_QK(buf ^ 1);
            __syncthreads();
        }
#undef AT_LOAD_QK
#undef AT_STORE_QK

        // ------------------------------------------------------------ online softmax (log2 units)
        // this lane's keys: j0 + 32a + (r&3) + 8(r>>2) + 4(lane>>5)
        float smax = AT_NEG;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int j = j0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                s[a][r] = j < M ? s[a][r] * sl2 : AT_NEG;
                smax = fmaxf(smax, s[a][r]);
            }
        smax = fmaxf(smax, __shfl_xor(smax, 32, 64));   // the partner lane holds the column's other 64 keys
        const float m_new = fmaxf(m_run, smax);
        const float alpha = exp2f(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float p = s[a][r] > 0.5f * AT_NEG ? exp2f(s[a][r] - m_new) : 0.f;
                s[a][r] = p;
                lsum += p;
            }
        m_run = m_new;
        l_run = l_run * alpha + lsum;
#pragma unroll
        for (int dt = 0; dt < ND; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[dt][r] *= alpha;

        // ------------------------------------------------------------ O^T += V P^T over the 128 keys
        // k-step ks = 2a + u: keys j0 + 16 ks + {4 kg + e, 8 + 4 kg + e}; V row d = srow (only rows < D staged)
        f32x4 va, vb2;
#define AT_LOAD_V(KS)                                                                                \
        do {                                                                                         \
            if (srow < D) {                                                                          \
                const float *vp_ = vb + (size_t)srow * M;                                            \
                const int ja_ = j0 + 16 * (KS) + 4 * skg, jb_ = ja_ + 8;                             \
                if (ja_ + 3 < M && jb_ + 3 < M && (M & 3) == 0) {                                    \
                    va = *(const f32x4 *)(vp_ + ja_);                                                \
                    vb2 = *(const f32x4 *)(vp_ + jb_);                                               \
                } else {                                                                             \
                    _Pragma("unroll") for (int e = 0; e < 4; e++) {                                  \
                        va[e] = ja_ + e < M ? vp_[ja_ + e] : 0.f;                                    \
                        vb2[e] = jb_ + e < M ? vp_[jb_ + e] : 0.f;                                   \
                    }                                                                                \
                }                                                                                    \
            }                                                                                        \
        } while (0)
#define AT_STORE_V(BUF)                                                                              \
        do {                                                                                         \
            if (srow < D) {                                                                          \
                unsigned char *base_ = lds + (BUF) * AT_BUF;                                         \
                uint4 h_, m_, l_;                                                                    \
                split_pair(va[0], va[1], h_.x, m_.x, l_.x);                                          \
                split_pair(va[2], va[3], h_.y, m_.y, l_.y);                                          \
                split_pair(vb2[0], vb2[1], h_.z, m_.z, l_.z);                                        \
                split_pair(vb2[2], vb2[3], h_.w, m_.w, l_.w);                                        \
                *(uint4 *)(base_ + st_lds) = h_;                                                     \
                *(uint4 *)(base_ + st_lds + 2 * AT_REG) = m_;                                        \
                *(uint4 *)(base_ + st_lds + 4 * AT_REG) = l_;                                        \
            }                                                                                        \
        } while (0)

        AT_LOAD_V(0);
        AT_STORE_V(0);                                 // buffer 0: its last QK^T read was before the loop's final barrier
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            const int buf = ks & 1;
            if (ks + 1 < 8) AT_LOAD_V(ks + 1);
            // probabilities of this k-step -> three bf16 planes (registers 8u..8u+7 of key tile a)
            uint4 ph, pm, pl;
            {
                const int a = ks >> 1, u = ks & 1;
                split_pair(s[a][8 * u + 0], s[a][8 * u + 1], ph.x, pm.x, pl.x);
                split_pair(s[a][8 * u + 2], s[a][8 * u + 3], ph.y, pm.y, pl.y);
                split_pair(s[a][8 * u + 4], s[a][8 * u + 5], ph.z, pm.z, pl.z);
                split_pair(s[a][8 * u + 6], s[a][8 * u + 7], ph.w, pm.w, pl.w);
            }
            const bf16x8 P[3] = {__builtin_bit_cast(bf16x8, ph), __builtin_bit_cast(bf16x8, pm), __builtin_bit_cast(bf16x8, pl)};
            const unsigned char *base = lds + buf * AT_BUF;
#pragma unroll
            for (int pa = 2; pa >= 0; pa--) {          // V plane l, m, h
                bf16x8 A[ND];
#pragma unroll
                for (int dt = 0; dt < ND; dt++) A[dt] = *(const bf16x8 *)(base + a_off + dt * 512 + pa * 2 * AT_REG);
#pragma unroll
                for (int pb = 2; pb >= 0; pb--) {
                    if (pa + pb > 2) continue;
#pragma unroll
                    for (int dt = 0; dt < ND; dt++) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[dt], P[pb], o[dt], 0, 0, 0);
                }
            }
            if (ks + 1 < 8) AT_STORE_V(buf ^ 1);
            __syncthreads();
        }
#undef AT_LOAD_V
#undef AT_STORE_V
    }

    // ---- normalise and store: O^T[d = 32 dt + (r&3) + 8(r>>2) + 4(lane>>5)][i = i0 + 32 wave + (lane&31)]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    const int i = i0 + wave * 32 + (lane & 31);
    if (i < N) {
        float *cb = ctx + ((size_t)b * H + h) * D * N + i;
#pragma unroll
        for (int dt = 0; dt < ND; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                cb[(size_t)d * N] = o[dt][r] * inv;
            }
    }
}

extern "C" int l3d_attention_forward_strided(const float *q, const float *k, const float *v, int B, int H, int D,
                                             int N, int M, long q_bstride, long k_bstride, long v_bstride, float scale,
                                             float *ctx, l3d_stream_t stream)
{
    L3D_REQUIRE(q && k && v && ctx && B > 0 && H > 0 && D > 0 && N > 0 && M > 0);
    if ((D != 32 && D != 64 && D != 128) || B > 65535 || H > 65535 || (((size_t)v) & 15) || (v_bstride & 3))
        return L3D_ERR_UNSUPPORTED;
    dim3 grid(l3d_divup(N, AT_TQ), H, B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (D == 32)      hipLaunchKernelGGL(attention_kernel<1>, grid, block, AT_LDS, st, q, k, v, H, N, M, scale, ctx, q_bstride, k_bstride, v_bstride);
    else if (D == 64) hipLaunchKernelGGL(attention_kernel<2>, grid, block, AT_LDS, st, q, k, v, H, N, M, scale, ctx, q_bstride, k_bstride, v_bstride);
    else              hipLaunchKernelGGL(attention_kernel<4>, grid, block, AT_LDS, st, q, k, v, H, N, M, scale, ctx, q_bstride, k_bstride, v_bstride);
    return l3d_check_launch();
}

