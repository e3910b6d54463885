// softcorr_f16.hip -- softcorr.hip's flash-style soft-correspondence pass (utils/svd.py:22-27) with the score GEMM as "f16x2" on
// the fp16 matrix cores: both embeddings are carried as two fp16 planes of x 2^T with an UNSCALED residual (h = f16(X),
// m = f16(X - h); T from the tensor's maximum so that it sits in [2^11, 2^12): a subnormal residual costs 2^-25 absolute in
// plane units, attention_f16b.hip / edgeconv_f16b.hip), three products per fp32 product (M h + H m + H h) instead of bf16x3's
// six, 8 + 4 fragment reads per chunk and wave instead of 12 + 6.  Tiling, the online softmax and the partial-state merge are
// softcorr.hip's (workgroup = 128 queries x M / KS keys, 4 waves, key tiles of 256, channel chunks of 16 double-buffered in LDS;
// `softcorr_merge_kernel` there combines the partial states).  The maxima come from one l3d_absmax4_partials launch over the two
// embeddings (134 MB read once at DCP's shape).
#include "common.h"
#include "split_bf16.h"          // f32x16
#include "split_f16.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define SF_TQ 128
#define SF_TK 256
#define SF_KREG (256 * 16)
#define SF_QREG (128 * 16 + 64)
#define SF_BUF (4 * SF_KREG + 4 * SF_QREG)      // K: 2 planes x 2 octets, Q: 2 planes x 2 octets
#define SF_VOFF (2 * SF_BUF)
#define SF_LDS (SF_VOFF + SF_TK * 16)
#define SF_NEG (-1.0e30f)

__device__ __forceinline__ int sf_exponent(float mx)
{
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);                 // mx = f 2^e, f in [0.5, 1)
    return 12 - e;
}
// eight fp32 -> two packed fp16 planes (h, m) of x c
__device__ __forceinline__ void sf_split8(const float (&x)[8], float c, uint4 &h, uint4 &m)
{
    af_split_x_unscaled(x[0], x[1], c, h.x, m.x);
    af_split_x_unscaled(x[2], x[3], c, h.y, m.y);
    af_split_x_unscaled(x[4], x[5], c, h.z, m.z);
    af_split_x_unscaled(x[6], x[7], c, h.w, m.w);
}

// partial state layout in the workspace: [B][N][parts][5] = (m, l, o0, o1, o2)
template <int DUMMY>
__global__ __launch_bounds__(256, 2) void softcorr_f16_kernel(const float *__restrict__ src_emb,
                                                          const float *__restrict__ tgt_emb,
                                                          const float *__restrict__ tgt, int C, int N, int M,
                                                          float scale, int ksplit,
                                                              const float *__restrict__ maxpart, float *__restrict__ ws)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int i0 = blockIdx.x * SF_TQ, b = blockIdx.y, ks = blockIdx.z;
    const int nk = C / 16;
    const int keys_per_split = ((M + ksplit * SF_TK - 1) / (ksplit * SF_TK)) * SF_TK;
    const int j_begin = ks * keys_per_split, j_end = min(M, j_begin + keys_per_split);
    const int parts = 4 * ksplit;

    // staging: K rows t (both octets), Q row t & 127, octet t >> 7
    const float *kbase = tgt_emb + (size_t)b * C * M;
    const float *qbase = src_emb + (size_t)b * C * N;
    const int qrow = t & 127, qkg = t >> 7;
    const int qn = min(i0 + qrow, N - 1);
    const int k_lds = t * 16;                                        // + kg * SF_KREG + p * 2 * SF_KREG
    const int q_lds = 4 * SF_KREG + qkg * SF_QREG + qrow * 16;       // + p * 2 * SF_QREG

    // running softmax state: 2 query columns per lane
    float m_run[2] = {SF_NEG, SF_NEG}, l_run[2] = {0.f, 0.f}, o_run[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    // plane scales from the operands' maxima (64 block maxima each, l3d_absmax4_partials): max|x| 2^T in [2^11, 2^12)
    float mq = 0.f, mk = 0.f;
    for (int i = 0; i < 64; i++) { mq = fmaxf(mq, maxpart[i]); mk = fmaxf(mk, maxpart[64 + i]); }
    const int Tq = sf_exponent(mq), Tk = sf_exponent(mk);
    const float cq = ldexpf(1.f, Tq), ck = ldexpf(1.f, Tk);
    const float sl2 = ldexpf(scale * 1.44269504088896340736f, -(Tq + Tk));   // accumulator -> log2 units

    const int a_off = (lane >> 5) * SF_KREG + (wm * 128 + (lane & 31)) * 16;                  // + a*512 + p*2*SF_KREG
    const int b_off = 4 * SF_KREG + (lane >> 5) * SF_QREG + (wn * 64 + (lane & 31)) * 16;     // + c*512 + p*2*SF_QREG

    for (int j0 = j_begin; j0 < j_end; j0 += SF_TK) {
        const int kn = min(j0 + t, M - 1);
        f32x16 acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;

        float kv[2][8], qv[8];
#define SF_LOAD(KC)                                                                                   \
        do {                                                                                          \
            _Pragma("unroll") for (int kg = 0; kg < 2; kg++)                                          \
                _Pragma("unroll") for (int e = 0; e < 8; e++)                                         \
                    kv[kg][e] = kbase[(size_t)((KC) * 16 + kg * 8 + e) * M + kn];                     \
            _Pragma("unroll") for (int e = 0; e < 8; e++)                                             \
                qv[e] = qbase[(size_t)((KC) * 16 + qkg * 8 + e) * N + qn];                            \
        } while (0)
#define SF_STORE(BUF)                                                                                 \
        do {                                                                                          \
            unsigned char *base_ = lds + (BUF) * SF_BUF;                                              \
            uint4 h_, m_;                                                                             \
            _Pragma("unroll") for (int kg = 0; kg < 2; kg++) {                                        \
                sf_split8(kv[kg], ck, h_, m_);                                                        \
                *(uint4 *)(base_ + k_lds + kg * SF_KREG) = h_;                                        \
                *(uint4 *)(base_ + k_lds + kg * SF_KREG + 2 * SF_KREG) = m_;                          \
            }                                                                                         \
            sf_split8(qv, cq, h_, m_);                                                                \
            *(uint4 *)(base_ + q_lds) = h_;                                                           \
            *(uint4 *)(base_ + q_lds + 2 * SF_QREG) = m_;                                             \
        } while (0)

        __syncthreads();                 // previous key tile's LDS reads (operands and V) are done
        {                                // V tile: target coordinates of this key tile
            const float *tb = tgt + (size_t)b * 3 * M;
            const float4 v = {tb[kn], tb[(size_t)M + kn], tb[(size_t)2 * M + kn], 0.f};
            *(float4 *)(lds + SF_VOFF + t * 16) = v;
        }
        SF_LOAD(0);
        SF_STORE(0);
        __syncthreads();
        for (int kc = 0; kc < nk; kc++) {
            const int buf = kc & 1;
            const bool more = kc + 1 < nk;
            if (more) SF_LOAD(kc + 1);
            const unsigned char *base = lds + buf * SF_BUF;
            // products M h, H m, H h (smallest first); the key planes are the A operand, H is read once for its two products
            f16x8 Bf[2][2];
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int c = 0; c < 2; c++) Bf[c][p] = *(const f16x8 *)(base + b_off + c * 512 + p * 2 * SF_QREG);
            f16x8 A[4];
#pragma unroll
            for (int prod = 0; prod < 3; prod++) {
                const int pb = prod == 1 ? 1 : 0;
                if (prod != 2) {
#pragma unroll
                    for (int a = 0; a < 4; a++) A[a] = *(const f16x8 *)(base + a_off + a * 512 + (prod == 0 ? 2 * SF_KREG : 0));
                }
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int c = 0; c < 2; c++)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[a], Bf[c][pb], acc[a][c], 0, 0, 0);
            }
            if (more) SF_STORE(buf ^ 1);
            __syncthreads();
        }
#undef SF_LOAD
#undef SF_STORE

        // ---- online softmax update.  This lane's keys: j0 + wm*128 + a*32 + (r&3) + 8(r>>2) + 4(lane>>5)
        float m_new[2], alpha[2], lsum[2] = {0.f, 0.f}, osum[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            float smax = SF_NEG;
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int j = j0 + wm * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float s = j < j_end ? acc[a][c][r] * sl2 : SF_NEG;
                    acc[a][c][r] = s;
                    smax = fmaxf(smax, s);
                }
            m_new[c] = fmaxf(m_run[c], smax);
            alpha[c] = exp2f(m_run[c] - m_new[c]);
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int jl = wm * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float4 v = *(const float4 *)(lds + SF_VOFF + jl * 16);
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const float p = acc[a][c][r] > 0.5f * SF_NEG ? exp2f(acc[a][c][r] - m_new[c]) : 0.f;
                    lsum[c] += p;
                    osum[c][0] = fmaf(p, v.x, osum[c][0]);
                    osum[c][1] = fmaf(p, v.y, osum[c][1]);
                    osum[c][2] = fmaf(p, v.z, osum[c][2]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);       // keep the 64 V reads from being hoisted into one 256-register burst
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            m_run[c] = m_new[c];
            l_run[c] = l_run[c] * alpha[c] + lsum[c];
#pragma unroll
            for (int d = 0; d < 3; d++) o_run[c][d] = o_run[c][d] * alpha[c] + osum[c][d];
        }
    }

    // ---- partial states out: part index = (ks*2 + wm)*2 + (lane>>5)
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int i = i0 + wn * 64 + c * 32 + (lane & 31);
        if (i < N) {
            float *dst = ws + (((size_t)b * N + i) * parts + (ks * 2 + wm) * 2 + (lane >> 5)) * 5;
            dst[0] = m_run[c]; dst[1] = l_run[c];
            dst[2] = o_run[c][0]; dst[3] = o_run[c][1]; dst[4] = o_run[c][2];
        }
    }
}


// softcorr.hip
void l3d_launch_softcorr_merge(const float *ws, int B, int N, int parts, float *src_corr, hipStream_t st);
int l3d_softcorr_ksplit(int M);

// maxpart: the floats l3d_absmax4_partials(src_emb, B C N, tgt_emb, B C M, NULL, 0, NULL, 0, maxpart) wrote (the first 128 are
// read); workspace as l3d_soft_correspondence (l3d_soft_correspondence_workspace_floats).
extern "C" int l3d_soft_correspondence_f16(const float *src_emb, const float *tgt_emb, const float *tgt, int B, int C, int N,
                                           int M, float scale, const float *maxpart, float *workspace, float *src_corr,
                                           l3d_stream_t stream)
{
    L3D_REQUIRE(src_emb && tgt_emb && tgt && maxpart && workspace && src_corr && B > 0 && C > 0 && N > 0 && M > 0);
    if (C % 16 || B > 65535) return L3D_ERR_UNSUPPORTED;
    const int ksplit = l3d_softcorr_ksplit(M);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(l3d_divup(N, SF_TQ), B, ksplit), block(256);
    hipLaunchKernelGGL(softcorr_f16_kernel<0>, grid, block, SF_LDS, st, src_emb, tgt_emb, tgt, C, N, M, scale, ksplit, maxpart,
                       workspace);
    int rc = l3d_check_launch();
    if (rc != L3D_OK) return rc;
    l3d_launch_softcorr_merge(workspace, B, N, 4 * ksplit, src_corr, st);
    return l3d_check_launch();
}
