// fold_mlp.hip -- PCN's folding decoder (models/pcn.py:84-101, final_conv = conv5 -> ReLU -> conv6 -> ReLU ->
// conv7, then + centre) as ONE kernel:
//     h5[k]   = relu( sum_{c<CG} W5g[k][c] * g[n][c] + s5[b][k] )          k < 512   (conv5; s5 = bias + W5[:, CG:] . global feature)
//     h6[co]  = relu( sum_k W6[co][k] * h5[k] + b6[co] )                   co < 512  (conv6)
//     out[j]  = sum_co W7[j][co] * h6[co] + b7[j] + centre[n][j]           j < 3     (conv7 + residual)
// The reference (and the layer-by-layer path) materialises h5 and h6 as [B,512,16384] tensors: 2.1 GB each at
// BASELINE config 4, written once and read once.  Here neither exists: conv6 is the bf16x3 GEMM of
// conv_split.hip whose x operand is GENERATED while it is staged (h5 depends on only CG = 5 values per point:
// 5 FMAs + max per element instead of a 4-byte HBM read), and conv7 is folded into conv6's epilogue
// (each lane multiplies its 64 accumulator rows by the three W7 rows and the partial sums are reduced
// over the lane pair, the two co-waves and the two 256-channel halves deterministically).
//
// Tile: 256 points x 256 output channels per pass, both channel halves in sequence inside one workgroup
// (512 threads, 8 waves as 2 x 4, wave tile 128 x 64), K = 512 in 32 chunks of 16, three LDS chunk buffers,
// operand reads one W plane at a time -- the structure of conv_split_kernel.
#include "common.h"
#include "split_bf16.h"

#define FM_C 512                        // conv5 out = conv6 in = conv6 out
#define FM_REGION (256 * 16 + 64)
#define FM_BUF (12 * FM_REGION)
#define FM_W7OFF (3 * FM_BUF)           // W7 as float4 (w7[0][co], w7[1][co], w7[2][co], 0) per co
#define FM_LDS (FM_W7OFF + FM_C * 16)

template <int CG>
__global__ __launch_bounds__(512) void fold_mlp_kernel(const float *__restrict__ g /*[B][N][CG]*/,
                                                       const float *__restrict__ w5g /*[512][CG]*/, const float *__restrict__ s5 /*[B][512]*/,
                                                       const uint4 *__restrict__ w6s /*split [32][3][2][512][8]*/,
                                                       const float *__restrict__ b6, const float *__restrict__ w7 /*[3][512]*/,
                                                       const float *__restrict__ b7, const float *__restrict__ centre /*[B][N][3]*/,
                                                       int N, float *__restrict__ out /*[B][N][3]*/)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int n0 = blockIdx.x * 256, b = blockIdx.y;
    constexpr int nk = FM_C / 16;

    // this thread's point for the x generation: row t & 255, channel octet kg = t >> 8 (wave-uniform)
    const int xrow = t & 255;
    const int xkg = __builtin_amdgcn_readfirstlane(t >> 8);
    const int xn = min(n0 + xrow, N - 1);
    // the point's CG grid / coarse values are re-read from L1 for every K chunk (the pointer passes through an empty asm, so the loads
    // are not hoisted): held in registers across the loop they were the 3 registers this kernel spilled at its 256-register cap
    const float *gp = g + ((size_t)b * N + xn) * CG;
    const float *s5b = s5 + (size_t)b * FM_C;
    const int wrow = t & 255, wkg = t >> 8;
    const int w_lds = wkg * FM_REGION + wrow * 16;
    const int x_lds = 6 * FM_REGION + xkg * FM_REGION + xrow * 16;

    for (int i = t; i < FM_C; i += 512)
        *(float4 *)(lds + FM_W7OFF + i * 16) = make_float4(w7[i], w7[FM_C + i], w7[2 * FM_C + i], 0.f);

    uint4 w0, w1, w2, x0, x1, x2;
    float part[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};           // conv7 partial sums, 2 point columns per lane

    const int frag_kg = (lane >> 5) * FM_REGION;
    const int a_off = frag_kg + (wm * 128 + (lane & 31)) * 16;
    const int b_off = 6 * FM_REGION + frag_kg + (wn * 64 + (lane & 31)) * 16;

#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        const int co0 = half * 256;
        const uint4 *wsrc = w6s + (size_t)wkg * FM_C + co0 + wrow;   // + ((kc*3 + p)*2) * 512

        // h5 octet of chunk KC for this thread's point -> three bf16 planes (x0, x1, x2)
#define FM_GEN_X(KC)                                                                                  \
        do {                                                                                          \
            float hv_[8], gv[CG];                                                                     \
            asm volatile("" : "+v"(gp));                                                              \
            _Pragma("unroll") for (int c = 0; c < CG; c++) gv[c] = gp[c];                             \
            _Pragma("unroll") for (int e = 0; e < 8; e++) {                                           \
                const int k_ = (KC) * 16 + xkg * 8 + e;                 /* wave-uniform: scalar loads */ \
                float a_ = s5b[k_];                                                                   \
                _Pragma("unroll") for (int c = 0; c < CG; c++) a_ = fmaf(w5g[k_ * CG + c], gv[c], a_); \
                hv_[e] = fmaxf(a_, 0.f);                                                              \
            }                                                                                         \
            split8(hv_, x0, x1, x2);                                                                  \
        } while (0)
#define FM_LOAD_W(KC)                                                                                 \
        do {                                                                                          \
            w0 = wsrc[((size_t)(KC) * 3 + 0) * 2 * FM_C];                                             \
            w1 = wsrc[((size_t)(KC) * 3 + 1) * 2 * FM_C];                                             \
            w2 = wsrc[((size_t)(KC) * 3 + 2) * 2 * FM_C];                                             \
        } while (0)
#define FM_STORE(BUF)                                                                                 \
        do {                                                                                          \
            unsigned char *base_ = lds + (BUF) * FM_BUF;                                              \
            *(uint4 *)(base_ + w_lds) = w0;                                                           \
            *(uint4 *)(base_ + w_lds + 2 * FM_REGION) = w1;                                           \
            *(uint4 *)(base_ + w_lds + 4 * FM_REGION) = w2;                                           \
            *(uint4 *)(base_ + x_lds) = x0;                                                           \
            *(uint4 *)(base_ + x_lds + 2 * FM_REGION) = x1;                                           \
            *(uint4 *)(base_ + x_lds + 4 * FM_REGION) = x2;                                           \
        } while (0)

        f32x16 acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;

        __syncthreads();                       // previous half's reads of the chunk buffers (and the W7 image) are done / visible
        FM_LOAD_W(0); FM_GEN_X(0); FM_STORE(0);
        FM_LOAD_W(1); FM_GEN_X(1); FM_STORE(1);
        __syncthreads();
        int buf = 0;
#pragma unroll 1
        for (int kc = 0; kc < nk; kc++) {
            const bool more = kc + 2 < nk;
            const int wbuf = buf == 0 ? 2 : buf - 1;                  // (kc + 2) % 3
            if (more) FM_LOAD_W(kc + 2);
            const unsigned char *base = lds + buf * FM_BUF;
            bf16x8 Bf[2][3];
#pragma unroll
            for (int p = 0; p < 3; p++)
#pragma unroll
                for (int c = 0; c < 2; c++) Bf[c][p] = *(const bf16x8 *)(base + b_off + c * 512 + p * 2 * FM_REGION);
#pragma unroll
            for (int pa = 2; pa >= 0; pa--) {
                bf16x8 A[4];
#pragma unroll
                for (int a = 0; a < 4; a++) A[a] = *(const bf16x8 *)(base + a_off + a * 512 + pa * 2 * FM_REGION);
#pragma unroll
                for (int pb = 2; pb >= 0; pb--) {
                    if (pa + pb > 2) continue;
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int c = 0; c < 2; c++)
                            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a], Bf[c][pb], acc[a][c], 0, 0, 0);
                }
                if (pa == 1 && more) {          // generate + store chunk kc+2 under the h-plane MFMAs
                    FM_GEN_X(kc + 2);
                    FM_STORE(wbuf);
                }
            }
            __syncthreads();
            buf = buf == 2 ? 0 : buf + 1;
        }
#undef FM_GEN_X
#undef FM_LOAD_W
#undef FM_STORE

        // ---- conv6 epilogue + conv7: h6 = relu(acc + b6); part[c][j] += W7[j][co] * h6[co][col c]
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = co0 + wm * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float bias = b6[co];
                const float4 wv = *(const float4 *)(lds + FM_W7OFF + co * 16);
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const float h = fmaxf(acc[a][c][r] + bias, 0.f);
                    part[c][0] = fmaf(wv.x, h, part[c][0]);
                    part[c][1] = fmaf(wv.y, h, part[c][1]);
                    part[c][2] = fmaf(wv.z, h, part[c][2]);
                }
            }
    }

    // ---- reduce the conv7 partials: lane pair (l, l^32), then the two co-waves (wm) through LDS
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int j = 0; j < 3; j++) part[c][j] += __shfl_xor(part[c][j], 32, 64);
    __syncthreads();                                                   // chunk buffers are free
    float *red = (float *)lds;                                         // [256 columns][4]
    if (wm == 1 && lane < 32) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int col = wn * 64 + c * 32 + lane;
            red[col * 4 + 0] = part[c][0]; red[col * 4 + 1] = part[c][1]; red[col * 4 + 2] = part[c][2];
        }
    }
    __syncthreads();
    if (wm == 0 && lane < 32) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int col = wn * 64 + c * 32 + lane;
            const int n = n0 + col;
            if (n < N) {
                const float *ce = centre + ((size_t)b * N + n) * 3;
                float *o = out + ((size_t)b * N + n) * 3;
#pragma unroll
                for (int j = 0; j < 3; j++) o[j] = (part[c][j] + red[col * 4 + j]) + b7[j] + ce[j];
            }
        }
    }
}

extern "C" int l3d_fold_mlp(const float *g, int CG, const float *w5g, const float *s5, const void *w6_split,
                            const float *b6, const float *w7, const float *b7, const float *centre, int B, int N,
                            float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(g && w5g && s5 && w6_split && b6 && w7 && b7 && centre && out && B > 0 && N > 0);
    if (B > 65535 || CG != 5) return L3D_ERR_UNSUPPORTED;
    dim3 grid(l3d_divup(N, 256), B), block(512);
    hipLaunchKernelGGL(fold_mlp_kernel<5>, grid, block, FM_LDS, (hipStream_t)stream, g, w5g, s5, (const uint4 *)w6_split, b6,
                       w7, b7, centre, N, out);
    return l3d_check_launch();
}
