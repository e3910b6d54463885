// fold_mlp_f16.hip -- fold_mlp.hip (PCN's folding decoder conv5 -> ReLU -> conv6 -> ReLU -> conv7 + centre as ONE kernel,
// models/pcn.py:84-101) with the conv6 GEMM as "f16x2" on the fp16 matrix cores: three fp16 MFMA products per fp32 product
// instead of bf16x3's six, fp32-level accuracy.  Structure, tiling and the conv7 reduction are fold_mlp.hip's; what
// changes:
//   W6 arrives as conv_f16.hip's weight image (H | Hs | M planes of W6 2^S, l3d_conv_f16_split_weights);
//   h5, generated while it is staged, is split into two planes h, m' of h5 2^T with T chosen PER WORKGROUP from a bound
//   the workgroup computes itself: |h5[n][k]| <= |s5[b][k]| + sum_c |W5g[k][c]| max_tile |g[n][c]| -- no extra pass, no
//   range flag (the bound cannot be exceeded), and powers of two cancel exactly in the epilogue;
//   per K chunk a wave reads 12 + 4 fragments and issues 24 MFMAs (bf16x3: 12 + 6 and 48).
// Round 3: the residual of h5 is carried UNSCALED (m = f16(h5 2^T - h); the tile's maximum sits in [2^11, 2^12), so a subnormal
// residual costs 2^-25 absolute in plane units, as in edgeconv_f16b.hip) and the Hs = H 2^-12 weight plane is not read: products
// M h + H m + H h, 8 + 4 fragment reads and 4 instead of 5 cell stores per chunk -- the kernel's LDS port was 87 % busy
// (16 KB of reads per wave and 40 KB of stores per chunk against 1536 cycles of matrix-pipe time per SIMD).
#include "common.h"
#include "split_bf16.h"          // f32x16
#include "split_f16.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *ff_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *ff_gbl_ptr_t;

#define FF_C 512                        // conv5 out = conv6 in = conv6 out
#define FF_REGION (256 * 16 + 64)
#define FF_BUF (8 * FF_REGION)          // W: 2 planes (H, M) x 2 octets, x: 2 planes x 2 octets
#define FF_W7OFF (3 * FF_BUF)           // W7 as float4 (w7[0][co], w7[1][co], w7[2][co], 0) per co
#define FF_SCR (FF_W7OFF + FF_C * 16)   // 64 floats of reduction scratch
#define FF_LDS (FF_SCR + 256)

template <int CG>
__global__ __launch_bounds__(512) void fold_mlp_f16_kernel(const float *__restrict__ g /*[B][N][CG]*/,
                                                           const float *__restrict__ w5g /*[512][CG]*/, const float *__restrict__ s5 /*[B][512]*/,
                                                           const uint4 *__restrict__ wH, const uint4 *__restrict__ wHs,
                                                           const uint4 *__restrict__ wM /*planes [64 octets][512 rows]*/,
                                                           const float *__restrict__ winv, const float *__restrict__ b6,
                                                           const float *__restrict__ w7 /*[3][512]*/, const float *__restrict__ b7,
                                                           const float *__restrict__ centre /*[B][N][3]*/, int N,
                                                           float *__restrict__ out /*[B][N][3]*/)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int n0 = blockIdx.x * 256, b = blockIdx.y;
    constexpr int nk = FF_C / 16;

    // this thread's point for the x generation: row t & 255, channel octet kg = t >> 8 (wave-uniform)
    const int xrow = t & 255;
    const int xkg = __builtin_amdgcn_readfirstlane(t >> 8);
    const int xn = min(n0 + xrow, N - 1);
    float gv[CG];
#pragma unroll
    for (int c = 0; c < CG; c++) gv[c] = g[((size_t)b * N + xn) * CG + c];
    const float *s5b = s5 + (size_t)b * FF_C;
    const int wrow = t & 255, wkg = t >> 8;
    // W6 cells go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write): a wave's 64 rows of one octet are
    // 1 KB at a wave-uniform LDS address; the weights of chunk kc + 2 are requested at the top of chunk kc and waited for two
    // barriers later
    const int w_lds = __builtin_amdgcn_readfirstlane(wkg * FF_REGION + (wrow & ~63) * 16);
    const int x_lds = 4 * FF_REGION + xkg * FF_REGION + xrow * 16;

    for (int i = t; i < FF_C; i += 512)
        *(float4 *)(lds + FF_W7OFF + i * 16) = make_float4(w7[i], w7[FF_C + i], w7[2 * FF_C + i], 0.f);

    // ---- the tile's bound on h5 and the plane scale 2^T:  hmax 2^T in [2^11, 2^12)
    float *scr = (float *)(lds + FF_SCR);
    float up, inv;
    {
        float gm[CG];
#pragma unroll
        for (int c = 0; c < CG; c++) {
            gm[c] = fabsf(gv[c]);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) gm[c] = fmaxf(gm[c], __shfl_xor(gm[c], d, 64));
        }
        if (lane == 0)
#pragma unroll
            for (int c = 0; c < CG; c++) scr[wave * 8 + c] = gm[c];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CG; c++) {
            gm[c] = scr[c];
#pragma unroll
            for (int w = 1; w < 8; w++) gm[c] = fmaxf(gm[c], scr[w * 8 + c]);
        }
        float bk = fabsf(s5b[t]);                                     // 512 threads <-> 512 channels of h5
#pragma unroll
        for (int c = 0; c < CG; c++) bk = fmaf(fabsf(w5g[t * CG + c]), gm[c], bk);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) bk = fmaxf(bk, __shfl_xor(bk, d, 64));
        __syncthreads();                                              // scr's first use is over
        if (lane == 0) scr[wave] = bk;
        __syncthreads();
        float hmax = scr[0];
#pragma unroll
        for (int w = 1; w < 8; w++) hmax = fmaxf(hmax, scr[w]);
        hmax *= 1.0000005f;                                           // the fp32 evaluation of h5 may round up past the bound's own rounding
        int e = 0;
        if (hmax > 0.f && hmax < 3.0e38f) (void)frexpf(hmax, &e);   // hmax = f 2^e, f in [0.5, 1)
        const int T = 12 - e;
        up = ldexpf(1.f, T);
        inv = ldexpf(*winv, -T);                                      // 2^-S 2^-T: exact
    }

    float gvu[CG];                                                    // g 2^T: the generated h5 comes out in plane units
#pragma unroll
    for (int c = 0; c < CG; c++) gvu[c] = gv[c] * up;
    uint4 x0, x1;
    float part[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};           // conv7 partial sums, 2 point columns per lane

    const int frag_kg = (lane >> 5) * FF_REGION;
    const int a_off = frag_kg + (wm * 128 + (lane & 31)) * 16;
    const int b_off = 4 * FF_REGION + frag_kg + (wn * 64 + (lane & 31)) * 16;

#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        const int co0 = half * 256;
        const size_t wofs = (size_t)wkg * FF_C + co0 + wrow;          // + kc * 2 * 512: cell [octet 2 kc + kg][row]

        // h5 octet of chunk KC for this thread's point -> two fp16 planes (x0 = h, x1 = m, unscaled residual) of h5 2^T
#define FF_GEN_X(KC)                                                                                  \
        do {                                                                                          \
            float hv_[8];                                                                             \
            _Pragma("unroll") for (int e_ = 0; e_ < 8; e_++) {                                        \
                const int k_ = (KC) * 16 + xkg * 8 + e_;                /* wave-uniform: scalar loads */ \
                float a_ = s5b[k_] * up;            /* h5 2^T: every term scaled by the same power of two, exact */ \
                _Pragma("unroll") for (int c = 0; c < CG; c++) a_ = fmaf(w5g[k_ * CG + c], gvu[c], a_); \
                hv_[e_] = fmaxf(a_, 0.f);                                                             \
            }                                                                                         \
            af_split_x_unscaled_cvt(hv_[0], hv_[1], x0.x, x1.x);                                      \
            af_split_x_unscaled_cvt(hv_[2], hv_[3], x0.y, x1.y);                                      \
            af_split_x_unscaled_cvt(hv_[4], hv_[5], x0.z, x1.z);                                      \
            af_split_x_unscaled_cvt(hv_[6], hv_[7], x0.w, x1.w);                                      \
        } while (0)
#define FF_DMA_W(KC, BUF)                                                                             \
        do {                                                                                          \
            unsigned char *base_ = lds + (BUF) * FF_BUF + w_lds;                                      \
            __builtin_amdgcn_global_load_lds((ff_gbl_ptr_t)(wH + wofs + (size_t)(KC) * 2 * FF_C), (ff_lds_ptr_t)base_, 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((ff_gbl_ptr_t)(wM + wofs + (size_t)(KC) * 2 * FF_C),     \
                                             (ff_lds_ptr_t)(base_ + 2 * FF_REGION), 16, 0, 0);        \
        } while (0)
#define FF_STORE(BUF)                                                                                 \
        do {                                                                                          \
            unsigned char *base_ = lds + (BUF) * FF_BUF;                                              \
            *(uint4 *)(base_ + x_lds) = x0;                                                           \
            *(uint4 *)(base_ + x_lds + 2 * FF_REGION) = x1;                                           \
        } while (0)

        f32x16 acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;

        __syncthreads();                       // previous half's reads of the chunk buffers (and the W7 image) are done / visible
        FF_DMA_W(0, 0); FF_DMA_W(1, 1);
        FF_GEN_X(0); FF_STORE(0);
        FF_GEN_X(1); FF_STORE(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
#pragma unroll 1
        for (int kc = 0; kc < nk; kc++) {
            const bool more = kc + 2 < nk;
            const int wbuf = buf == 0 ? 2 : buf - 1;                  // (kc + 2) % 3
            if (more) FF_DMA_W(kc + 2, wbuf);     // buffer (kc + 2) % 3 was read last in chunk kc - 1: every wave is past that barrier
            const unsigned char *base = lds + buf * FF_BUF;
            f16x8 Bf[2][2];
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int c = 0; c < 2; c++) Bf[c][p] = *(const f16x8 *)(base + b_off + c * 512 + p * 2 * FF_REGION);
            f16x8 A[4];
#pragma unroll
            for (int prod = 0; prod < 3; prod++) {                    // M h, H m, H h: smallest first; H is read once for both
                const int pb = prod == 1 ? 1 : 0;
                if (prod != 2) {
#pragma unroll
                    for (int a = 0; a < 4; a++) A[a] = *(const f16x8 *)(base + a_off + a * 512 + (prod == 0 ? 2 * FF_REGION : 0));
                }
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int c = 0; c < 2; c++)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[a], Bf[c][pb], acc[a][c], 0, 0, 0);
                if (prod == 1 && more) FF_GEN_X(kc + 2);       // generate chunk kc+2's h5 under the MFMAs ...
            }
            if (more) FF_STORE(wbuf);                              // ... and store it behind the last product
            // chunk kc + 1's weights (requested one chunk ago) have landed before anybody reads them; this chunk's two requests may stay in flight
            if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

            __syncthreads();
            buf = buf == 2 ? 0 : buf + 1;
        }
#undef FF_GEN_X
#undef FF_DMA_W
#undef FF_STORE

        // ---- conv6 epilogue + conv7: h6 = relu(acc 2^-(S+T) + b6); part[c][j] += W7[j][co] * h6[co][col c]
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = co0 + wm * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float bias = b6[co];
                const float4 wv = *(const float4 *)(lds + FF_W7OFF + co * 16);
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const float h = fmaxf(acc[a][c][r] * inv + bias, 0.f);
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

// w6_planes: conv_f16.hip's weight image of W6 [512][512] (l3d_conv_f16_split_weights); everything else as l3d_fold_mlp
extern "C" int l3d_fold_mlp_f16(const float *g, int CG, const float *w5g, const float *s5, const void *w6_planes,
                                const float *b6, const float *w7, const float *b7, const float *centre, int B, int N,
                                float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(g && w5g && s5 && w6_planes && b6 && w7 && b7 && centre && out && B > 0 && N > 0);
    if (B > 65535 || CG != 5 || (((size_t)w6_planes) & 15)) return L3D_ERR_UNSUPPORTED;
    const size_t pb = l3d_f16_plane_bytes(FF_C, FF_C);
    const unsigned char *wp = (const unsigned char *)w6_planes;
    dim3 grid(l3d_divup(N, 256), B), block(512);
    hipLaunchKernelGGL(fold_mlp_f16_kernel<5>, grid, block, FF_LDS, (hipStream_t)stream, g, w5g, s5, (const uint4 *)wp,
                       (const uint4 *)(wp + pb), (const uint4 *)(wp + 2 * pb), (const float *)(wp + 3 * pb), b6, w7, b7, centre, N, out);
    return l3d_check_launch();
}
