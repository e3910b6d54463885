// wgrad.hip -- weight gradient of a 1x1 conv / Linear over points, deterministic and fp32-exact per product
// (reference: the autograd of nn.Conv1d / nn.Conv2d(k=1) in models/dgcnn.py:34-48,
// models/pcn.py:110-153, models/pointnet.py:51-73 as driven by examples/train_pcn.py:70-91).
//
//   dW[co][ci] = sum_b sum_p dz[b][co][p] * x[b][ci][p]            dz [B,Cout,P], x [B,Cin,P], both fp32, P contiguous
//
// The reduction axis is B*P (655 360 for DGCNN's EdgeConv layers at B = 32): one GEMM whose K axis is split into
// (cloud, chunk of PC points) pieces.  wgrad_partial_kernel: a 64 (co) x 64 (ci) tile per workgroup and piece on
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation over <= PC terms), partial sums to a workspace
// [pieces][Cout][Cin].  wgrad_reduce_kernel: adds the pieces in a fixed order in fp64 and rounds once.  No atomics: two
// runs give the same bits, and the error is that of <= PC fp32 accumulations plus one rounding, not of B*P.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WG_T 64        // output tile (both axes)
#define WG_K 32        // points per stage
#define WG_LD 66       // LDS row pitch (floats): the 8-wide k runs of the four lanes sharing a row land 16 banks apart

__global__ __launch_bounds__(256) void wgrad_partial_kernel(const float *__restrict__ dz, const float *__restrict__ x, int Cout,
                                                            int Cin, long P, int PC, int nchunk, float *__restrict__ part)
{
    __shared__ float As[WG_K][WG_LD], Bs[WG_K][WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int co0 = blockIdx.x * WG_T, ci0 = blockIdx.y * WG_T;
    const int piece = blockIdx.z, b = piece / nchunk, chunk = piece % nchunk;
    const long p_begin = (long)chunk * PC;
    const long p_end = (p_begin + PC < P) ? p_begin + PC : P;
    const float *dzb = dz + (size_t)b * Cout * P, *xb = x + (size_t)b * Cin * P;
    const bool vec = (P % 4) == 0;

    const int row = tid >> 2, kb = (tid & 3) * 8;
    float ra[8], rb[8];
    auto load8 = [&](float (&r)[8], const float *base, int rr, int rows, long p) {
        const bool rok = rr < rows;
        const float *src = base + (size_t)(rok ? rr : 0) * P;
        if (rok && vec && p + 8 <= p_end) {
            const float4 v0 = *(const float4 *)(src + p), v1 = *(const float4 *)(src + p + 4);
            r[0] = v0.x; r[1] = v0.y; r[2] = v0.z; r[3] = v0.w;
            r[4] = v1.x; r[5] = v1.y; r[6] = v1.z; r[7] = v1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) r[e] = (rok && p + e < p_end) ? src[p + e] : 0.f;
        }
    };

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    const int l31 = lane & 31, kh = lane >> 5;

    load8(ra, dzb, co0 + row, Cout, p_begin + kb);
    load8(rb, xb, ci0 + row, Cin, p_begin + kb);
    for (long p0 = p_begin; p0 < p_end; p0 += WG_K) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; e++) {
            As[kb + e][row] = ra[e];
            Bs[kb + e][row] = rb[e];
        }
        __syncthreads();
        if (p0 + WG_K < p_end) {
            load8(ra, dzb, co0 + row, Cout, p0 + WG_K + kb);
            load8(rb, xb, ci0 + row, Cin, p0 + WG_K + kb);
        }
#pragma unroll
        for (int st = 0; st < WG_K / 2; st++) {
            const float av = As[2 * st + kh][wm * 32 + l31];
            const float bv = Bs[2 * st + kh][wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    // D[row = co][col = ci]; col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    float *out = part + (size_t)piece * Cout * Cin;
    const int ci = ci0 + wn * 32 + l31;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int co = co0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (co < Cout && ci < Cin) out[(size_t)co * Cin + ci] = acc[e];
    }
}

// 64 elements per workgroup, four threads per element: thread (g, e) adds pieces [g q, (g+1) q) of element e in piece order
// (fp64, eight independent loads in flight), the four partial sums meet in LDS and are added in g order -- a fixed tree, the same
// bits every run.  (One thread per element walking all 320 pieces of a DGCNN layer one dependent trip at a time took as long
// as the MFMA kernel that produced them.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, long elems, int pieces, float *__restrict__ dw)
{
    __shared__ double sums[4][64];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + e;
    const int q = (pieces + 3) / 4, k0 = g * q, k1 = min(pieces, k0 + q);
    double s = 0.0;
    if (i < elems) {
        const float *src = part + i;
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[(size_t)(k + u) * elems];
#pragma unroll
            for (int u = 0; u < 8; u++) s += (double)v[u];
        }
        for (; k < k1; k++) s += (double)src[(size_t)k * elems];
    }
    sums[g][e] = s;
    __syncthreads();
    if (g == 0 && i < elems) dw[i] = (float)(((sums[0][e] + sums[1][e]) + sums[2][e]) + sums[3][e]);
}

static inline int wgrad_chunk(long P, int pc)
{
    if (pc <= 0) pc = 2048;
    pc = (pc + WG_K - 1) / WG_K * WG_K;
    return pc;
}

extern "C" size_t l3d_wgrad_workspace_bytes(int B, int Cout, int Cin, long P, int pc)
{
    if (B <= 0 || Cout <= 0 || Cin <= 0 || P <= 0) return 0;
    pc = wgrad_chunk(P, pc);
    const long nchunk = (P + pc - 1) / pc;
    return (size_t)B * nchunk * Cout * Cin * sizeof(float);
}

extern "C" int l3d_wgrad(const float *dz, const float *x, int B, int Cout, int Cin, long P, int pc, float *workspace, float *dw,
                         l3d_stream_t stream)
{
    L3D_REQUIRE(dz && x && workspace && dw && B > 0 && Cout > 0 && Cin > 0 && P > 0);
    pc = wgrad_chunk(P, pc);
    const long nchunk = (P + pc - 1) / pc;
    const long pieces = (long)B * nchunk;
    L3D_REQUIRE(pieces <= 65535 && l3d_divup(Cin, WG_T) <= 65535);
    dim3 grid(l3d_divup(Cout, WG_T), l3d_divup(Cin, WG_T), (unsigned)pieces);
    hipLaunchKernelGGL(wgrad_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, dz, x, Cout, Cin, P, pc, (int)nchunk, workspace);
    int rc = l3d_check_launch();
    if (rc != L3D_OK) return rc;
    const long elems = (long)Cout * Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)l3d_divup(elems, 64)), dim3(256), 0, (hipStream_t)stream, workspace, elems,
                       (int)pieces, dw);
    return l3d_check_launch();
}
