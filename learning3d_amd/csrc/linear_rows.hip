// linear_rows.hip -- nn.Linear over a handful of rows: y [R][Cout] = act(x [R][Cin] w^T + b) for R of the order of the batch
// size (PCN's fully connected decoder, models/pcn.py:132-137: 64 clouds x 1024 -> 1024 -> 1024 -> 3072).  The op is the
// weight matrix read once (4 / 4 / 12 MB): run as a 1x1 conv with the rows as the points of ONE cloud it filled a fraction
// of a CU row and took 120 us per layer.  Here a workgroup owns 16 output channels for 64 rows: its weight slice (64 KB at
// Cin = 1024) streams through LDS once in coalesced 16-byte loads, the x tile (shared by all workgroups, L2-resident) beside
// it; a thread is one row x four channels, the channel's weights arrive as LDS broadcasts, the row's values as conflict-free
// ds_read_b128 (row stride 260 floats).  fp32 fmaf chains in ascending k.
#include "common.h"

#define LR_ROWS 64
#define LR_CO 16
#define LR_KT 256
#define LR_STRIDE (LR_KT + 4)
#define LR_LDS ((LR_ROWS + LR_CO) * LR_STRIDE * 4)

// With one workgroup per CU at most (Cout / 16 of them) nothing hides a chunk's load latency but the workgroup itself: the
// next chunk's 20 float4 per thread are in flight (registers) while this chunk is multiplied out of LDS -- 256-wide chunks,
// four per 1024 input channels (single-buffered 64-wide chunks: 16 exposed round trips, 54 us per layer).
__global__ __launch_bounds__(256) void linear_rows_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, int R, int Cin, int Cout, int relu,
                                                          float *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) float lr_lds[];
    float (*xs)[LR_STRIDE] = (float (*)[LR_STRIDE])lr_lds;
    float (*ws)[LR_STRIDE] = (float (*)[LR_STRIDE])(lr_lds + LR_ROWS * LR_STRIDE);
    const int t = threadIdx.x, r = t & 63, g = t >> 6;
    const int co0 = blockIdx.x * LR_CO, r0 = blockIdx.y * LR_ROWS;
    constexpr int XF = LR_ROWS * LR_KT / 4 / 256, WF = LR_CO * LR_KT / 4 / 256, Q = LR_KT / 4;      // float4 per thread; per tile row
    float4 px[XF], pw[WF];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < XF; j++) {
            const int f = t + 256 * j, row = f / Q, c4 = f % Q;
            px[j] = (r0 + row < R) ? *(const float4 *)(x + (size_t)(r0 + row) * Cin + k0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < WF; j++) {
            const int f = t + 256 * j, row = f / Q, c4 = f % Q;
            pw[j] = (co0 + row < Cout) ? *(const float4 *)(w + (size_t)(co0 + row) * Cin + k0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    fetch(0);
    for (int k0 = 0; k0 < Cin; k0 += LR_KT) {
        __syncthreads();                                  // the previous chunk has been read
#pragma unroll
        for (int j = 0; j < XF; j++) { const int f = t + 256 * j; *(float4 *)&xs[f / Q][(f % Q) * 4] = px[j]; }
#pragma unroll
        for (int j = 0; j < WF; j++) { const int f = t + 256 * j; *(float4 *)&ws[f / Q][(f % Q) * 4] = pw[j]; }
        __syncthreads();
        if (k0 + LR_KT < Cin) fetch(k0 + LR_KT);
#pragma unroll 8
        for (int kk = 0; kk < LR_KT; kk += 4) {
            const float4 xv = *(const float4 *)&xs[r][kk];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float4 wv = *(const float4 *)&ws[g * 4 + c][kk];
                acc[c] = fmaf(xv.x, wv.x, acc[c]);
                acc[c] = fmaf(xv.y, wv.y, acc[c]);
                acc[c] = fmaf(xv.z, wv.z, acc[c]);
                acc[c] = fmaf(xv.w, wv.w, acc[c]);
            }
        }
    }
    if (r0 + r < R) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int co = co0 + g * 4 + c;
            if (co < Cout) {
                float v = acc[c] + (bias ? bias[co] : 0.f);
                if (relu) v = l3d_act(v, relu);
                y[(size_t)(r0 + r) * Cout + co] = v;
            }
        }
    }
}

// y [R][Cout] = act(x [R][Cin] w [Cout][Cin]^T + bias);  Cin % 256 == 0, x and w 16-byte aligned.
extern "C" int l3d_linear_rows(const float *x, const float *w, const float *bias, int R, int Cin, int Cout, int relu, float *y,
                               l3d_stream_t stream)
{
    L3D_REQUIRE(x && w && y && R > 0 && Cin > 0 && Cout > 0);
    if (Cin % LR_KT || ((((size_t)x) | ((size_t)w)) & 15) || l3d_divup(R, LR_ROWS) > 65535) return L3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(linear_rows_kernel, dim3(l3d_divup(Cout, LR_CO), l3d_divup(R, LR_ROWS)), dim3(256), LR_LDS, (hipStream_t)stream,
                       x, w, bias, R, Cin, Cout, relu, y);
    return l3d_check_launch();
}
