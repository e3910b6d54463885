// softcorr.hip -- the soft-correspondence step of SVDHead (utils/svd.py:22-27) as ONE flash-style pass:
//     scores = softmax_j( <src_emb[:, i], tgt_emb[:, j]> / sqrt(C) ),   src_corr[:, i] = sum_j scores[i][j] * tgt[:, j]
// The reference materialises scores [B,N,M] (134 MB at B=32, N=M=1024), reads it for the softmax,
// writes it, transposes it and reads it again for the second matmul.  Here the score tile lives in MFMA
// accumulators, the softmax is online (running max / sum per query) and the "value" matrix is just the
// three target coordinates, so nothing of size N x M ever leaves the CU.  SURVEY.md 8(f) rank 1.
//
// The score GEMM (2 B N M C FLOP = 34 GFLOP at C = 512) runs as bf16x3 on the bf16 matrix cores
// (split_bf16.h: fp32 operands split exactly into three bf16 planes, six products, fp32 accumulate).
// D[j][i] = sum_c K[j][c] Q[i][c] with the KEYS on the MFMA row axis: a lane then holds 64 keys of ONE
// query per accumulator column, so max / exp / sum / weighted coordinate sums are in-lane.
//
// Workgroup = 128 queries x (M / KS) keys of one cloud, 256 threads = 4 waves (2 over keys x 2 over
// queries, wave tile 128 keys x 64 queries), key tiles of 256, channel chunks of 16 double-buffered in
// LDS exactly as conv_split.hip (both operands staged from the channel-first [B,C,N] embeddings with
// per-lane coalesced dword loads and split in flight).  Every lane keeps its own running (m, l, o[3])
// per query column; the 4 x KS partial states of a query (2 lane halves x 2 key waves x KS key splits)
// go to a small workspace and a second kernel merges them -- no cross-wave traffic inside the loop.
#include "common.h"
#include "split_bf16.h"

#define SC_TQ 128
#define SC_TK 256
#define SC_KREG (256 * 16)
#define SC_QREG (128 * 16 + 64)
#define SC_BUF (6 * SC_KREG + 6 * SC_QREG)
#define SC_VOFF (2 * SC_BUF)
#define SC_LDS (SC_VOFF + SC_TK * 16)
#define SC_NEG (-1.0e30f)

// partial state layout in the workspace: [B][N][parts][5] = (m, l, o0, o1, o2)
template <int DUMMY>
__global__ __launch_bounds__(256, 2) void softcorr_kernel(const float *__restrict__ src_emb,
                                                          const float *__restrict__ tgt_emb,
                                                          const float *__restrict__ tgt, int C, int N, int M,
                                                          float scale, int ksplit, float *__restrict__ ws)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int i0 = blockIdx.x * SC_TQ, b = blockIdx.y, ks = blockIdx.z;
    const int nk = C / 16;
    const int keys_per_split = ((M + ksplit * SC_TK - 1) / (ksplit * SC_TK)) * SC_TK;
    const int j_begin = ks * keys_per_split, j_end = min(M, j_begin + keys_per_split);
    const int parts = 4 * ksplit;

    // staging: K rows t (both octets), Q row t & 127, octet t >> 7
    const float *kbase = tgt_emb + (size_t)b * C * M;
    const float *qbase = src_emb + (size_t)b * C * N;
    const int qrow = t & 127, qkg = t >> 7;
    const int qn = min(i0 + qrow, N - 1);
    const int k_lds = t * 16;                                        // + kg * SC_KREG + p * 2 * SC_KREG
    const int q_lds = 6 * SC_KREG + qkg * SC_QREG + qrow * 16;       // + p * 2 * SC_QREG

    // running softmax state: 2 query columns per lane
    float m_run[2] = {SC_NEG, SC_NEG}, l_run[2] = {0.f, 0.f}, o_run[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    const float sl2 = scale * 1.44269504088896340736f;               // scores are compared / exponentiated in log2 units

    const int a_off = (lane >> 5) * SC_KREG + (wm * 128 + (lane & 31)) * 16;                  // + a*512 + p*2*SC_KREG
    const int b_off = 6 * SC_KREG + (lane >> 5) * SC_QREG + (wn * 64 + (lane & 31)) * 16;     // + c*512 + p*2*SC_QREG

    for (int j0 = j_begin; j0 < j_end; j0 += SC_TK) {
        const int kn = min(j0 + t, M - 1);
        f32x16 acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;

        float kv[2][8], qv[8];
#define SC_LOAD(KC)                                                                                   \
        do {                                                                                          \
            _Pragma("unroll") for (int kg = 0; kg < 2; kg++)                                          \
                _Pragma("unroll") for (int e = 0; e < 8; e++)                                         \
                    kv[kg][e] = kbase[(size_t)((KC) * 16 + kg * 8 + e) * M + kn];                     \
            _Pragma("unroll") for (int e = 0; e < 8; e++)                                             \
                qv[e] = qbase[(size_t)((KC) * 16 + qkg * 8 + e) * N + qn];                            \
        } while (0)
#define SC_STORE(BUF)                                                                                 \
        do {                                                                                          \
            unsigned char *base_ = lds + (BUF) * SC_BUF;                                              \
            uint4 h_, m_, l_;                                                                         \
            _Pragma("unroll") for (int kg = 0; kg < 2; kg++) {                                        \
                split8(kv[kg], h_, m_, l_);                                                           \
                *(uint4 *)(base_ + k_lds + kg * SC_KREG) = h_;                                        \
                *(uint4 *)(base_ + k_lds + kg * SC_KREG + 2 * SC_KREG) = m_;                          \
                *(uint4 *)(base_ + k_lds + kg * SC_KREG + 4 * SC_KREG) = l_;                          \
            }                                                                                         \
            split8(qv, h_, m_, l_);                                                                   \
            *(uint4 *)(base_ + q_lds) = h_;                                                           \
            *(uint4 *)(base_ + q_lds + 2 * SC_QREG) = m_;                                             \
            *(uint4 *)(base_ + q_lds + 4 * SC_QREG) = l_;                                             \
        } while (0)

        __syncthreads();                 // previous key tile's LDS reads (operands and V) are done
        {                                // V tile: target coordinates of this key tile
            const float *tb = tgt + (size_t)b * 3 * M;
            const float4 v = {tb[kn], tb[(size_t)M + kn], tb[(size_t)2 * M + kn], 0.f};
            *(float4 *)(lds + SC_VOFF + t * 16) = v;
        }
        SC_LOAD(0);
        SC_STORE(0);
        __syncthreads();
        for (int kc = 0; kc < nk; kc++) {
            const int buf = kc & 1;
            const bool more = kc + 1 < nk;
            if (more) SC_LOAD(kc + 1);
            const unsigned char *base = lds + buf * SC_BUF;
            // operand fragments: the 6 query fragments stay live, the key fragments are fetched one
            // plane at a time (l, m, h) so that 10 fragments are live instead of 18 (register budget:
            // 128 accumulators + staging + softmax state must fit 256 for two workgroups per CU)
            bf16x8 Bf[2][3];
#pragma unroll
            for (int p = 0; p < 3; p++)
#pragma unroll
                for (int c = 0; c < 2; c++) Bf[c][p] = *(const bf16x8 *)(base + b_off + c * 512 + p * 2 * SC_QREG);
#pragma unroll
            for (int pa = 2; pa >= 0; pa--) {
                bf16x8 A[4];
#pragma unroll
                for (int a = 0; a < 4; a++) A[a] = *(const bf16x8 *)(base + a_off + a * 512 + pa * 2 * SC_KREG);
                // products with this key plane, smallest first: l*h | m*m, m*h | h*l, h*m, h*h
#pragma unroll
                for (int pb = 2; pb >= 0; pb--) {
                    if (pa + pb > 2) continue;
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int c = 0; c < 2; c++)
                            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a], Bf[c][pb], acc[a][c], 0, 0, 0);
                }
            }
            if (more) SC_STORE(buf ^ 1);
            __syncthreads();
        }
#undef SC_LOAD
#undef SC_STORE

        // ---- online softmax update.  This lane's keys: j0 + wm*128 + a*32 + (r&3) + 8(r>>2) + 4(lane>>5)
        float m_new[2], alpha[2], lsum[2] = {0.f, 0.f}, osum[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            float smax = SC_NEG;
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int j = j0 + wm * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float s = j < j_end ? acc[a][c][r] * sl2 : SC_NEG;
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
                const float4 v = *(const float4 *)(lds + SC_VOFF + jl * 16);
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const float p = acc[a][c][r] > 0.5f * SC_NEG ? exp2f(acc[a][c][r] - m_new[c]) : 0.f;
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

__global__ __launch_bounds__(256) void softcorr_merge_kernel(const float *__restrict__ ws, int N, int parts,
                                                             float *__restrict__ src_corr /*[B][3][N]*/)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= N) return;
    const float *p = ws + ((size_t)b * N + i) * parts * 5;
    float m = SC_NEG;
    for (int q = 0; q < parts; q++) m = fmaxf(m, p[q * 5]);
    float l = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;
    for (int q = 0; q < parts; q++) {
        const float w = exp2f(p[q * 5] - m);
        l = fmaf(p[q * 5 + 1], w, l);
        o0 = fmaf(p[q * 5 + 2], w, o0); o1 = fmaf(p[q * 5 + 3], w, o1); o2 = fmaf(p[q * 5 + 4], w, o2);
    }
    float *out = src_corr + (size_t)b * 3 * N + i;
    out[0] = o0 / l; out[(size_t)N] = o1 / l; out[(size_t)2 * N] = o2 / l;
}

static int sc_ksplit(int M) { return M >= 2 * SC_TK ? 2 : 1; }

extern "C" size_t l3d_soft_correspondence_workspace_floats(int B, int N, int M)
{
    return (size_t)B * N * 4 * sc_ksplit(M) * 5;
}

extern "C" int l3d_soft_correspondence(const float *src_emb, const float *tgt_emb, const float *tgt, int B, int C,
                                       int N, int M, float scale, float *workspace, float *src_corr,
                                       l3d_stream_t stream)
{
    L3D_REQUIRE(src_emb && tgt_emb && tgt && workspace && src_corr && B > 0 && C > 0 && N > 0 && M > 0);
    if (C % 16 || B > 65535) return L3D_ERR_UNSUPPORTED;
    const int ksplit = sc_ksplit(M);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(l3d_divup(N, SC_TQ), B, ksplit), block(256);
    hipLaunchKernelGGL(softcorr_kernel<0>, grid, block, SC_LDS, st, src_emb, tgt_emb, tgt, C, N, M, scale, ksplit, workspace);
    int rc = l3d_check_launch();
    if (rc != L3D_OK) return rc;
    hipLaunchKernelGGL(softcorr_merge_kernel, dim3(l3d_divup(N, 256), B), dim3(256), 0, st, workspace, N, 4 * ksplit, src_corr);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// LayerNorm as utils/transformer.py:109-119 defines it (NOT nn.LayerNorm): y = a * (x - mean) /
// (std + eps) + b with the UNBIASED standard deviation and eps added to std.  One wave per row, the row
// in registers (C <= 2048), one read and one write of the tensor instead of the ~8 elementwise /
// reduction launches of the torch-op composition (131 us -> ~35 us for [32*1024, 512]).
// ---------------------------------------------------------------------------------------------
template <int VPL /* float4 per lane */>
__global__ __launch_bounds__(256) void layernorm_ref_kernel(const float *__restrict__ x, const float *__restrict__ a,
                                                            const float *__restrict__ bb, float eps, long rows, int C,
                                                            float *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4 *xr = (const float4 *)(x + row * C);
    const int c4 = C >> 2;
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int q = lane + 64 * i;
        v[i] = q < c4 ? xr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        if (lane + 64 * i < c4) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            ss += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float inv = 1.f / (sqrtf(ss / (float)(C - 1)) + eps);
    float4 *yr = (float4 *)(y + row * C);
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int q = lane + 64 * i;
        if (q < c4) {
            const float4 ga = ((const float4 *)a)[q], be = ((const float4 *)bb)[q];
            float4 o;
            o.x = ga.x * (v[i].x - mean) * inv + be.x;
            o.y = ga.y * (v[i].y - mean) * inv + be.y;
            o.z = ga.z * (v[i].z - mean) * inv + be.z;
            o.w = ga.w * (v[i].w - mean) * inv + be.w;
            yr[q] = o;
        }
    }
}

// fp32 values only (C % 4 == 0, C <= 2048): l3d_layernorm_planes with img == NULL
static int layernorm_values(const float *x, const float *a, const float *b, float eps, long rows, int C,
                            float *y, l3d_stream_t stream)
{
    L3D_REQUIRE(x && a && b && y && rows > 0 && C > 1);
    if (C % 4 || C > 2048 || ((((size_t)x) | ((size_t)y) | ((size_t)a) | ((size_t)b)) & 15)) return L3D_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int vpl = (C / 4 + 63) / 64;
    if (vpl <= 1)      hipLaunchKernelGGL(layernorm_ref_kernel<1>, grid, block, 0, st, x, a, b, eps, rows, C, y);
    else if (vpl <= 2) hipLaunchKernelGGL(layernorm_ref_kernel<2>, grid, block, 0, st, x, a, b, eps, rows, C, y);
    else if (vpl <= 4) hipLaunchKernelGGL(layernorm_ref_kernel<4>, grid, block, 0, st, x, a, b, eps, rows, C, y);
    else               hipLaunchKernelGGL(layernorm_ref_kernel<8>, grid, block, 0, st, x, a, b, eps, rows, C, y);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Residual connection of the pointer network's sublayers (utils/transformer.py:82-88, x + sublayer(norm(x))):
// x is [B,N,C] (channel-last), the sublayer's output comes from a 1x1-conv kernel as [B,C,N].  torch adds the
// two through a strided view at ~1/3 of the streaming rate; here 32x32 tiles go through LDS so that both
// reads and the write are coalesced:   out[b][n][c] = x[b][n][c] + y[b][c][n].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_transposed_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                             int N, int C, float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z, n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const float *yb = y + (size_t)b * C * N;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {                           // y rows c0 + r, columns n0 + tx
        const int c = c0 + r, n = n0 + tx;
        tile[r][tx] = (c < C && n < N) ? yb[(size_t)c * N + n] : 0.f;
    }
    __syncthreads();
    const size_t xb = (size_t)b * N * C;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {                           // x / out rows n0 + r, columns c0 + tx
        const int n = n0 + r, c = c0 + tx;
        if (n < N && c < C) out[xb + (size_t)n * C + c] = x[xb + (size_t)n * C + c] + tile[tx][r];
    }
}

extern "C" int l3d_add_transposed(const float *x, const float *y, int B, int N, int C, float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(x && y && out && B > 0 && N > 0 && C > 0);
    if (B > 65535 || l3d_divup(C, 32) > 65535) return L3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(add_transposed_kernel, dim3(l3d_divup(N, 32), l3d_divup(C, 32), B), dim3(256), 0,
                       (hipStream_t)stream, x, y, N, C, out);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// LayerNorm that ALSO emits its output as the fp16 activation image conv_f16.hip consumes (h | m' planes of y 2^T in the
// tiled layout [C/8][rows][8], then 2^-T): every Linear of the pointer network that follows a LayerNorm (q|k|v, q, k|v,
// the feed-forward's first layer) then runs on the f16x2 kernel with no split pass over its input.  The plane scale
// needs no pass over y either: |y_c| <= |a_c| sqrt(C-1) + |b_c| because sum_c z_c^2 <= C-1 for z = (x-mean)/(std+eps)
// with the unbiased std, so T comes from the layer's own parameters and cannot be exceeded.
// A workgroup normalises 16 rows (a wave per row, 4 rows per wave), parks the 16-byte plane cells in LDS as [octet][row]
// and writes them out as 256-byte runs (16 rows of one octet); y itself is written as before (when asked for).  (32 rows per workgroup --
// 512-byte runs, 67 KB of LDS, two workgroups per CU -- was slower: 55 us against the plain kernel's 23.)
// ---------------------------------------------------------------------------------------------
#define LNP_ROWS 16
#define LNP_RS 17                      // row stride of the LDS cell array (bank spread)
template <int VPL /* float4 per lane */>
__global__ __launch_bounds__(256) void layernorm_planes_kernel(const float *__restrict__ x, const float *__restrict__ a,
                                                               const float *__restrict__ bb, float eps, long rows, int C,
                                                               float *__restrict__ y, uint4 *__restrict__ ph, uint4 *__restrict__ pm,
                                                               float *__restrict__ inv_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lnp_lds[];
    const int NO = C >> 3;                                            // octets
    uint4 *cell = (uint4 *)lnp_lds;                                   // [2][NO][LNP_RS]
    float *scr = (float *)(cell + 2 * NO * LNP_RS);                   // [4]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c4 = C >> 2;
    // ---- plane scale from the layer's parameters
    float hm = 0.f;
    const float rt = sqrtf((float)(C - 1));
    for (int c = t; c < C; c += 256) hm = fmaxf(hm, fmaf(fabsf(a[c]), rt, fabsf(bb[c])));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) hm = fmaxf(hm, __shfl_xor(hm, d, 64));
    if (lane == 0) scr[wave] = hm;
    __syncthreads();
    hm = fmaxf(fmaxf(scr[0], scr[1]), fmaxf(scr[2], scr[3])) * 1.000001f;
    int e = 0;
    if (hm > 0.f && hm < 3.0e38f) (void)frexpf(hm, &e);               // hm = f 2^e, f in [0.5, 1)
    const float up = ldexpf(1.f, 12 - e);
    if (blockIdx.x == 0 && t == 0) *inv_out = ldexpf(1.f, e - 12);

    const long row0 = (long)blockIdx.x * LNP_ROWS;
    for (int rr = 0; rr < LNP_ROWS / 4; rr++) {
        const int rslot = rr * 4 + wave;
        const long row = row0 + rslot;
        if (row >= rows) continue;                                    // wave-uniform
        const float4 *xr = (const float4 *)(x + row * C);
        float4 v[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int q = lane + 64 * i;
            v[i] = q < c4 ? xr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float mean = s / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            if (lane + 64 * i < c4) {
                const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
                ss += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        const float inv = 1.f / (sqrtf(ss / (float)(C - 1)) + eps);
        float4 *yr = (float4 *)(y + row * C);
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int q = lane + 64 * i;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < c4) {
                const float4 ga = ((const float4 *)a)[q], be = ((const float4 *)bb)[q];
                o.x = ga.x * (v[i].x - mean) * inv + be.x;
                o.y = ga.y * (v[i].y - mean) * inv + be.y;
                o.z = ga.z * (v[i].z - mean) * inv + be.z;
                o.w = ga.w * (v[i].w - mean) * inv + be.w;
                if (y) yr[q] = o;
            }
            // the odd lane's four values join the even lane's: one octet = channels 8 o .. 8 o + 7
            const float n0 = __shfl_down(o.x, 1, 64), n1 = __shfl_down(o.y, 1, 64), n2 = __shfl_down(o.z, 1, 64), n3 = __shfl_down(o.w, 1, 64);
            if (!(lane & 1) && q < c4) {
                const float xv[8] = {o.x, o.y, o.z, o.w, n0, n1, n2, n3};
                _Float16 h[8], m[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float X = xv[k] * up;
                    h[k] = (_Float16)X;
                    m[k] = (_Float16)((X - (float)h[k]) * 4096.0f);
                }
                const int oc = q >> 1;
                cell[oc * LNP_RS + rslot] = *(const uint4 *)h;
                cell[(NO + oc) * LNP_RS + rslot] = *(const uint4 *)m;
            }
        }
    }
    __syncthreads();
    // ---- contiguous runs: LNP_ROWS consecutive rows of one (plane, octet)
    for (int c = t; c < 2 * NO * LNP_ROWS; c += 256) {
        const int r = c & (LNP_ROWS - 1), po = c / LNP_ROWS;          // po = plane * NO + octet
        const long row = row0 + r;
        if (row < rows) {
            const int oc = po >= NO ? po - NO : po;
            (po >= NO ? pm : ph)[(size_t)oc * rows + row] = cell[po * LNP_RS + r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same LayerNorm on a CHANNEL-FIRST tensor x [B][C][N] (normalised over C for every point), output as the activation
// image and / or as y [B][C][N]: with it the pointer network keeps the layout its GEMMs write ([B,Cout,N]) from end to end --
// no transposed residual adds, no .contiguous() copies around the sublayers (utils/transformer.py, Transformer._forward_cf).
// A workgroup takes 64 consecutive points; wave w of NW holds channels [w C/NW, (w+1) C/NW) of them in registers (one read of x,
// 256-byte runs per channel; 64 channels per wave keep three waves on a SIMD), the partial sums of a point meet in LDS (fp64: the mean and the variance are the
// correctly rounded ones), and every lane writes whole 16-byte plane cells (64 consecutive rows of an octet = 1 KB per
// wave store).  Same scale rule as above.
// ---------------------------------------------------------------------------------------------
template <int CPW /* channels per wave */, int NW /* waves */>
__global__ __launch_bounds__(64 * NW) void layernorm_planes_cf_kernel(const float *__restrict__ x, const float *__restrict__ a,
                                                                  const float *__restrict__ bb, float eps, int Bn, int N,
                                                                  float *__restrict__ y, uint4 *__restrict__ ph, uint4 *__restrict__ pm,
                                                                  float *__restrict__ inv_out, float rs)
{
    constexpr int C = NW * CPW;
    __shared__ double red[2][NW][64];
    __shared__ float scr[NW];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // uniform: row pointers in SGPRs
    const int b = blockIdx.y, n = blockIdx.x * 64 + lane;
    const bool ok = n < N;
    const int c0 = wave * CPW;
    // ---- plane scale from the layer's parameters
    float up = 1.f;
    if (ph) {
        float hm = 0.f;
        const float rt = sqrtf((float)(C - 1));
        for (int c = t; c < C; c += 64 * NW) hm = fmaxf(hm, fmaf(fabsf(a[c]), rt, fabsf(bb[c])));
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) hm = fmaxf(hm, __shfl_xor(hm, d, 64));
        if (lane == 0) scr[wave] = hm;
        __syncthreads();
        hm = scr[0];
#pragma unroll
        for (int w = 1; w < NW; w++) hm = fmaxf(hm, scr[w]);
        hm *= 1.000001f;
        int e = 0;
        if (hm > 0.f && hm < 3.0e38f) (void)frexpf(hm, &e);
        up = ldexpf(1.f, 12 - e);
        if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0) *inv_out = ldexpf(1.f, e - 12);
    }
    // buffer loads: the wave's channel block as the resource (wave-uniform), the channel row as the scalar offset, the point as
    // the only per-lane part of the address -- one VGPR instead of a 64-bit address per load; points beyond N read 0
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(x + ((size_t)b * C + c0) * N), 0, (int)((size_t)CPW * N * 4), 0x00020000);
    const int voff = ok ? n * 4 : 0x7ffffff0;
    float v[CPW];
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < CPW; i++) {
        v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, i * N * 4, 0));
        s += (double)v[i];
    }
    red[0][wave][lane] = s;
    __syncthreads();
    double tot = red[0][0][lane];
#pragma unroll
    for (int w = 1; w < NW; w++) tot += red[0][w][lane];
    const float mean = (float)(tot / (double)C);
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < CPW; i++) { const float d = v[i] - mean; ss += (double)d * (double)d; }    // (fp32 difference as in the row kernel; keeping 64 converted values alive would cost 128 registers)
    red[1][wave][lane] = ss;
    __syncthreads();
    tot = red[1][0][lane];
#pragma unroll
    for (int w = 1; w < NW; w++) tot += red[1][w][lane];
    const float var = (float)(tot / (double)(C - 1));
    const float inv = 1.f / (sqrtf(var) + eps);
    const size_t rows = (size_t)Bn * N, row = (size_t)b * N + n;
#pragma unroll
    for (int o = 0; o < CPW / 8; o++) {
        _Float16 h[8], m[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = o * 8 + k;
            const float val = a[c0 + i] * (v[i] - mean) * inv + bb[c0 + i];
            if (y && ok) (y + ((size_t)b * C + c0 + i) * N)[n] = val;
            const float X = val * up;
            h[k] = (_Float16)X;
            m[k] = (_Float16)((X - (float)h[k]) * rs);           // rs = 2^12 (the three-plane kernel's m') or 1 (unscaled: the two-plane form)
        }
        if (ph && ok) {
            const size_t cell = (size_t)(c0 / 8 + o) * rows + row;
            ph[cell] = *(const uint4 *)h;
            pm[cell] = *(const uint4 *)m;
        }
    }
}

// x [B][C][N] -> y [B][C][N] (or NULL) and / or img = the activation image of y with rows b N + n (l3d_f16_act_bytes(B N, C)
// bytes; or NULL).  C in {128, 256, 512}.  flags 1: the image's residual plane is UNSCALED, m = f16(X - h) (the two-plane form of
// l3d_pointwise_conv_f16 reads it, L3D_CONV_F16_TWO_PLANE); 0: m' = f16((X - h) 2^12).
extern "C" int l3d_layernorm_planes_cf(const float *x, const float *a, const float *b, float eps, int B, int C, int N, float *y,
                                       void *img, int flags, l3d_stream_t stream)
{
    L3D_REQUIRE(x && a && b && (y || img) && B > 0 && N > 0 && C > 1 && (flags & ~1) == 0);
    const float rs = (flags & 1) ? 1.0f : 4096.0f;
    if ((C != 128 && C != 256 && C != 512) || B > 65535 || (((size_t)img) & 15)) return L3D_ERR_UNSUPPORTED;
    const size_t rows = (size_t)B * N, pb = (size_t)(C / 8) * rows * 16;
    unsigned char *d = (unsigned char *)img;
    uint4 *ph = d ? (uint4 *)d : nullptr, *pm = d ? (uint4 *)(d + pb) : nullptr;
    float *inv = d ? (float *)(d + 2 * pb) : nullptr;
    dim3 grid((unsigned)l3d_divup(N, 64), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (C == 512)      hipLaunchKernelGGL((layernorm_planes_cf_kernel<64, 8>), grid, dim3(512), 0, st, x, a, b, eps, B, N, y, ph, pm, inv, rs);
    else if (C == 256) hipLaunchKernelGGL((layernorm_planes_cf_kernel<64, 4>), grid, dim3(256), 0, st, x, a, b, eps, B, N, y, ph, pm, inv, rs);
    else               hipLaunchKernelGGL((layernorm_planes_cf_kernel<32, 4>), grid, dim3(256), 0, st, x, a, b, eps, B, N, y, ph, pm, inv, rs);
    return l3d_check_launch();
}

// y = the layer norm's fp32 values (or NULL: planes only -- every consumer of the pointer network's sublayer norms reads the image, and the
// fp32 copy is a third of this kernel's traffic), plus img = the activation image of y (l3d_f16_act_bytes(rows, C) bytes)
extern "C" int l3d_layernorm_planes(const float *x, const float *a, const float *b, float eps, long rows, int C, float *y,
                                    void *img, l3d_stream_t stream)
{
    L3D_REQUIRE(x && a && b && (img || y) && rows > 0 && C > 1);
    if (!img) return layernorm_values(x, a, b, eps, rows, C, y, stream);
    if (C % 8 || C > 512 || ((((size_t)x) | ((size_t)y) | ((size_t)a) | ((size_t)b) | ((size_t)img)) & 15)) return L3D_ERR_UNSUPPORTED;
    const size_t pb = (size_t)(C / 8) * (size_t)rows * 16;
    unsigned char *d = (unsigned char *)img;
    const size_t lds = (size_t)2 * (C / 8) * LNP_RS * 16 + 64;
    dim3 grid((unsigned)((rows + LNP_ROWS - 1) / LNP_ROWS)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int vpl = (C / 4 + 63) / 64;
    if (vpl <= 1) hipLaunchKernelGGL(layernorm_planes_kernel<1>, grid, block, lds, st, x, a, b, eps, rows, C, y, (uint4 *)d, (uint4 *)(d + pb), (float *)(d + 2 * pb));
    else          hipLaunchKernelGGL(layernorm_planes_kernel<2>, grid, block, lds, st, x, a, b, eps, rows, C, y, (uint4 *)d, (uint4 *)(d + pb), (float *)(d + 2 * pb));
    return l3d_check_launch();
}
