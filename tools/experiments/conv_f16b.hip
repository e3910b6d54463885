// conv_f16b.hip -- conv_f16.hip's two-plane GEMM (NPW = 2: products M h + H m + H h on an activation image with an unscaled
// residual; models/dgcnn.py:48, conv5 of the benchmark step) restructured around ONE wave per SIMD.
//
// What the timeline of conv_f16_kernel<false,false,false,2> showed (tools/probe_conv_timeline.hip, per 256 x 256 tile):
// pipeline fill 2.2 us | main loop 34.3 us | epilogue 7.2 us | store drain + relaunch 0.9 us, twice per CU.  The main loop
// is 32 chunks of 2 570 cycles against 1 536 cycles of matrix-pipe time: both waves of a SIMD leave the chunk's barrier together,
// read their 12 fragments together and only then feed the pipe; a wave has no registers left to fetch the next chunk's fragments
// early (128 accumulators + 48 fragment registers of 256; the attempt spilled, LABLOG R2.3).
//
// Here a workgroup is FOUR waves (256 threads, one per SIMD, 512 registers each):
//   * wave tile 128 (co) x 128 (n) = 4 x 4 MFMA tiles of 32 x 32: 256 accumulator registers (AGPRs), 16 fragments per chunk
//     (4 M, 4 H, 4 h, 4 m) instead of 2 x 12 for the same 48 MFMAs per SIMD -- a third fewer LDS reads;
//   * TWO fragment sets (2 x 64 VGPRs): chunk kc+1's fragments are read while chunk kc's 48 MFMAs issue, one ds_read_b128 behind
//     every third MFMA -- nothing but the barrier stands between two chunks' MFMA streams;
//   * FOUR LDS stages of 32 KB (W 16 KB + x 16 KB): chunk kc+3's eight DMA pieces per wave are issued during chunk kc (one
//     behind every sixth MFMA), have landed by the barrier of chunk kc+2, are read into registers during kc+2 and multiplied
//     in kc+3.
// Same operand layout, same tile order, same product order per accumulator as conv_f16_kernel: the results are the same bits.
#include <type_traits>
#include "../../learning3d_amd/csrc/common.h"
#include "../../learning3d_amd/csrc/split_bf16.h"          // f32x16 typedef

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *cb_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *cb_gbl_ptr_t;

#define CB_T 256                       // workgroup tile, both ways
#define CB_REG (CB_T * 16)             // bytes of one (plane, k-octet) region: 256 rows x 16 B
#define CB_STAGE (8 * CB_REG)          // W: H k0, H k1, M k0, M k1 | x: h k0, h k1, m k0, m k1
#define CB_NSTAGE 4
#define CB_LDS (CB_NSTAGE * CB_STAGE)

#ifdef CB_TIMELINE    // tools/probe_conv_timeline.hip
__device__ long long *g_cb_timeline;
#define CBM(i) { if (threadIdx.x == 0) g_cb_timeline[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define CBM(i)
#endif

struct CbFrag { f16x8 A[4][2], B[4][2]; };        // [tile][plane]: A = W (H, M), B = x (h, m)

__global__ __launch_bounds__(256) void conv_f16b_kernel(const uint4 *__restrict__ xh, const uint4 *__restrict__ xm,
                                                        const uint4 *__restrict__ wH, const uint4 *__restrict__ wM,
                                                        const float *__restrict__ winv, const float *__restrict__ xinv,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        int shift_bstride, int Bn, int Cin, int Cout, int N, int relu,
                                                        float *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    CBM(0)
    // tile order: the Cout tiles of one point tile are consecutive slots of ONE XCD (conv_f16.hip)
    int pt, ct;
    {
        const int nct = Cout / CB_T, npt = Bn * (N / CB_T), L = blockIdx.x;
        if (npt % 8 == 0) {
            const int xcd = L & 7, slot = L >> 3;
            ct = slot % nct;
            pt = (slot / nct) * 8 + xcd;
        } else {
            ct = L % nct;
            pt = L / nct;
        }
    }
    const int ntn = N / CB_T;
    const int n0 = (pt % ntn) * CB_T, co0 = ct * CB_T, b = pt / ntn;
    const int nk = Cin / 16;
    const size_t BN = (size_t)Bn * N;

    // ---- DMA: wave w moves plane w of a stage (0: W H, 1: W M, 2: x h, 3: x m): 2 k-octets x 4 pieces of 64 rows (1 KB each,
    // one global_load_lds_dwordx4).  The piece offset (quarter * 1 KB) is the instruction's immediate: it moves both addresses.
    const uint4 *src0, *src1;
    size_t step;                                   // uint4 cells per chunk (two octets)
    {
        const uint4 *pl = wave == 0 ? wH : (wave == 1 ? wM : (wave == 2 ? xh : xm));
        const size_t rows = wave < 2 ? (size_t)Cout : BN;
        const size_t row0 = wave < 2 ? (size_t)co0 : (size_t)b * N + n0;
        src0 = pl + row0 + lane;
        src1 = src0 + rows;
        step = 2 * rows;
    }
    const int dst_w = wave * 2 * CB_REG;           // this wave's two regions of a stage
#define CB_DMA(stage, i)                                                                                              \
    {                                                                                                                 \
        cb_lds_ptr_t d_ = (cb_lds_ptr_t)(lds + (stage) * CB_STAGE + dst_w + ((i) >> 2) * CB_REG);                     \
        cb_gbl_ptr_t s_ = (cb_gbl_ptr_t)(((i) >> 2) ? src1 : src0);                                                   \
        switch ((i) & 3) {                                                                                            \
        case 0: __builtin_amdgcn_global_load_lds(s_, d_, 16, 0, 0); break;                                            \
        case 1: __builtin_amdgcn_global_load_lds(s_, d_, 16, 1024, 0); break;                                         \
        case 2: __builtin_amdgcn_global_load_lds(s_, d_, 16, 2048, 0); break;                                         \
        default: __builtin_amdgcn_global_load_lds(s_, d_, 16, 3072, 0); break;                                        \
        }                                                                                                             \
    }
#define CB_DMA_ADVANCE { src0 += step; src1 += step; }

    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;

    const int kgl = lane >> 5;
    const int a_off = kgl * CB_REG + (wm * 128 + (lane & 31)) * 16;                    // + a * 512 + p * 2 * CB_REG
    const int b_off = 4 * CB_REG + kgl * CB_REG + (wn * 128 + (lane & 31)) * 16;       // + c * 512 + p * 2 * CB_REG
    // fragment j of a chunk, in the order the products need them: M (4), h (4), H (4), m (4)
    auto read_frag = [&](CbFrag &F, const unsigned char *base, int j) {
        const int kind = j >> 2, i = j & 3;
        if (kind == 0) F.A[i][1] = *(const f16x8 *)(base + a_off + i * 512 + 2 * CB_REG);
        else if (kind == 1) F.B[i][0] = *(const f16x8 *)(base + b_off + i * 512);
        else if (kind == 2) F.A[i][0] = *(const f16x8 *)(base + a_off + i * 512);
        else F.B[i][1] = *(const f16x8 *)(base + b_off + i * 512 + 2 * CB_REG);
    };

    // prologue: chunks 0, 1, 2 in flight; chunk 0's fragments into set F0
#pragma unroll
    for (int s = 0; s < 3; s++) {
#pragma unroll
        for (int i = 0; i < 8; i++) CB_DMA(s, i)
        CB_DMA_ADVANCE
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    CBM(1)
    CbFrag F0, F1;
#pragma unroll
    for (int j = 0; j < 16; j++) read_frag(F0, lds, j);

    // one chunk: 48 MFMAs on set F (chunk kc) | 16 fragment reads of chunk kc+1 into set G | 8 DMA pieces of chunk kc+3.
    // TAIL = how many of {DMA issue, two chunks in flight, next chunk} are gone: 0 in the steady state, 1 / 2 / 3 for the last
    // three chunks (compile-time, so the MFMA stream has no branches)
    auto chunk = [&](auto tail_c, int kc, const CbFrag &F, CbFrag &G) {
        constexpr int TAIL = decltype(tail_c)::value;
        constexpr bool has_next = TAIL < 3, more = TAIL < 1;
        // chunk kc+1's pieces (issued during kc-2) have landed; kc+2's eight may still be in flight
        if (has_next) {
            if (TAIL < 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // ... everybody's; and every wave is done with stage (kc+3)%4 = chunk kc-1's
        }
        const unsigned char *nbase = lds + ((kc + 1) & 3) * CB_STAGE;
        const int dstage = (kc + 3) & 3;
#pragma unroll
        for (int n = 0; n < 48; n++) {
            // three products, smallest first: M h, H m, H h
            const int prod = n >> 4, a = (n >> 2) & 3, c = n & 3;
            const int pa = prod == 0 ? 1 : 0, pb = prod == 1 ? 1 : 0;
            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.A[a][pa], F.B[c][pb], acc[a][c], 0, 0, 0);
            if (has_next && n % 3 == 1) {
                __builtin_amdgcn_sched_barrier(0);
                read_frag(G, nbase, n / 3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more && n % 6 == 3) {
                __builtin_amdgcn_sched_barrier(0);
                CB_DMA(dstage, n / 6)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) CB_DMA_ADVANCE
    };
    // nk is even and >= 4 (dispatcher): chunks 0 .. nk-4 run in the steady state
    typedef std::integral_constant<int, 0> T0;
    int kc = 0;
    for (; kc + 4 < nk; kc += 2) {
        chunk(T0(), kc, F0, F1);
        chunk(T0(), kc + 1, F1, F0);
    }
    chunk(T0(), kc, F0, F1);
    chunk(std::integral_constant<int, 1>(), kc + 1, F1, F0);
    chunk(std::integral_constant<int, 2>(), kc + 2, F0, F1);
    chunk(std::integral_constant<int, 3>(), kc + 3, F1, F0);
    CBM(2)

    // ---- epilogue: D[co = 32a + (r&3) + 8(r>>2) + 4(lane>>5)][n = 32c + (lane&31)]
    const float inv = *winv * *xinv;               // 2^-S 2^-T: exact
    float *yb = y + (size_t)b * Cout * N;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + wm * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float sc = (scale ? scale[co] : 1.f) * inv;
            const float sh = shift ? shift[(size_t)b * shift_bstride + co] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float v = acc[a][c][r] * sc + sh;
                if (relu) v = l3d_act(v, relu);
                yb[(size_t)co * N + n0 + wn * 128 + c * 32 + (lane & 31)] = v;
            }
        }
#ifdef CB_TIMELINE
    CBM(3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CBM(4)
    __syncthreads();
    CBM(5)
    if (t == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id)); g_cb_timeline[(size_t)blockIdx.x * 8 + 7] = id; }
#endif
}

extern "C" size_t l3d_f16_plane_bytes(long rows, int cols);

// x_planes: an activation image with an UNSCALED residual (l3d_edgeconv_forward_f16b, out_mode 2); w_planes: a weight image
// (l3d_conv_f16_split_weights; its Hs plane is not read).  Cout % 256 == 0, N % 256 == 0, Cin % 32 == 0, Cin >= 64.
extern "C" int l3d_pointwise_conv_f16b(const void *x_planes, const void *w_planes, const float *scale, const float *shift,
                                       int shift_bstride, int B, int Cin, int Cout, int N, int relu, float *y,
                                       l3d_stream_t stream)
{
    L3D_REQUIRE(x_planes && w_planes && y && B > 0 && Cin > 0 && Cout > 0 && N > 0);
    if (Cout % CB_T || N % CB_T || Cin % 32 || Cin < 64 || B > 65535 || (((size_t)x_planes) & 15) || (((size_t)w_planes) & 15))
        return L3D_ERR_UNSUPPORTED;
    const size_t xpb = l3d_f16_plane_bytes((long)B * N, Cin), wpb = l3d_f16_plane_bytes(Cout, Cin);
    const unsigned char *xp = (const unsigned char *)x_planes, *wp = (const unsigned char *)w_planes;
    dim3 grid((unsigned)((size_t)(N / CB_T) * (Cout / CB_T) * B)), block(256);
    hipLaunchKernelGGL(conv_f16b_kernel, grid, block, CB_LDS, (hipStream_t)stream, (const uint4 *)xp, (const uint4 *)(xp + xpb),
                       (const uint4 *)wp, (const uint4 *)(wp + 2 * wpb), (const float *)(wp + 3 * wpb), (const float *)(xp + 2 * xpb),
                       scale, shift, shift_bstride, B, Cin, Cout, N, relu, y);
    return l3d_check_launch();
}
