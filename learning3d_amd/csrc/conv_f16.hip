// conv_f16.hip -- per-point linear layer (1x1 conv + folded BN + ReLU) as "f16x2" on the fp16 matrix cores:
// three fp16 MFMA products per fp32 product (half of bf16x3's six), fp32-level error.  models/dgcnn.py:48 (conv5,
// 33 % of the benchmark step), pointnet.py:22-49, pcn.py:84-125.
//
// Arithmetic (see edgeconv_f16.hip for the derivation and the error measurements):
//   activation x:  X = x 2^T (T per tensor),  h = f16(X),  m' = f16((X - h) 2^12)
//   weight     w:  W = w 2^S (S per matrix, max|W| in [4,8)),  H = f16(W),  Hs = f16(H 2^-12),  M = f16(W - H)
//   acc = sum_k ( M h + Hs m' + H h ) = 2^(S+T) w.x,     y = act(acc * (scale 2^-S 2^-T) + shift)
// fp16 has 30 binades: T places the tensor's largest magnitude near 2^12 (16x headroom to 65504; values 2^-26 of the
// maximum and larger keep full relative precision, smaller ones an absolute error of 2^-49 of the maximum).  The
// generic splitter takes T from the tensor's own maximum (one extra read pass); a fused producer takes it from what it
// knows about its output (the EdgeConv kernel: BatchNorm statistics) and raises the range flag if that was wrong.
//
// Both operands arrive PRE-SPLIT as fp16 planes in the tiled layout  plane[k / 8][row][8 fp16]  (one 16-byte cell per
// (octet, row); rows = Cout for W, B*N for x): exactly the "8 consecutive k of row i" that a lane of
// v_mfma_f32_32x32x16_f16 consumes, and 64 consecutive rows of one octet are 1 KB contiguous -- one
// global_load_lds_dwordx4 per wave moves them global -> LDS with no registers, no VALU and no ds_write.  W is split
// once per weight version (l3d_conv_f16_split_weights); x is written in this form by its producer (the EdgeConv f16
// kernel's pooled epilogue, out_mode 1) or by l3d_split_f16_rows.  Measured beside the MFMAs (tools/probe_mfma_filler.hip):
// a VALU instruction costs ~7.5 cycles, a packed-fp32 one 17+, a global load ~40 -- an in-kernel split of fp32 x
// (conv_split.hip: 11 VALU per value pair, redone by the 4 workgroups that share an x tile) is what this layout removes.
//
// Kernel.  Workgroup tile 256 (co) x 256 (n), 512 threads = 8 waves (4 x 2), wave tile 64 (co) x 128 (n) = 2 x 4
// MFMA tiles of 32x32 (128 accumulator registers, two waves per SIMD).  With three W planes and two x planes the
// fragment reads per K chunk are 3a + 2c for an a x c wave tile: 14 for 2 x 4 (16 for 4 x 2).  K chunks of 16 (one MFMA
// k-step), THREE LDS stages of 40 KB (W 24 KB + x 16 KB); per chunk every wave issues 5 DMA pieces for chunk kc+2,
// waits for its own pieces of chunk kc (s_waitcnt vmcnt -- the loop has no other vector memory traffic), one barrier,
// 14 ds_read_b128, 24 MFMAs with the five DMA instructions spread between them.  tools/probe_conv_f16.hip (s_memtime
// inside the loop): 2.3 k cycles per chunk and wave against 1.54 k of matrix-pipe time for the SIMD's two waves; the
// DMA data is never waited for (latency hidden), but ISSUING five 1 KB DMA instructions in one block behind the barrier cost
// every wave ~600 cycles (the CU's vector-memory path takes them at 16 cycles apiece and all eight waves queue at once):
// spread out, 99 -> 96 us.  The two epilogues (one per tile round; 134 MB) are exposed, ~17 us each.
// Measured, not kept (LABLOG R2.3): next-chunk x fragments read during this chunk's MFMAs (spills: 112 us); four-wave
// 128 x 256 workgroups, two per CU, which do hide the epilogue (99 us); those with W fragments straight from L2 (110 us);
// all 14 reads ahead of the first MFMA (105 us); first-product fragments read across the barrier (105 us); LDS bank
// padding of the regions (worth 25 % in the bare read + MFMA loop, nothing here); non-temporal epilogue stores (-1 us).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "split_bf16.h"          // f32x4 / f32x16 typedefs
#include "split_f16.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CF_TM 256
#define CF_TN 256
#define CF_WBYTES (6 * 4096)                 // 3 planes x 2 octets x 256 rows x 16 B
#define CF_XBYTES (4 * 4096)                 // 2 planes x 2 octets x 256 rows x 16 B
#define CF_STAGE (CF_WBYTES + CF_XBYTES)
#define CF_NSTAGE 3
#define CF_LDS (CF_NSTAGE * CF_STAGE)

typedef __attribute__((address_space(3))) void *cf_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *cf_gbl_ptr_t;

// ---------------------------------------------------------------------------------------------
// Splitters.  src [R][C] fp32 row-major -> planes [ceil(C/8)][R][8] fp16 (C padded with zeros).
//   weights: H, Hs, M of src * 2^S, S chosen on the device from max|src| (two launches: max, split); *inv = 2^-S
//   activations: h, m' ; raises *range_flag if |x| > 60000
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cf_absmax_kernel(const float *__restrict__ src, size_t n, unsigned *__restrict__ out)
{
    // 16-byte loads, four in flight per thread (round 6: with one 4-byte load per thread and trip and a grid of 256 blocks this read
    // 285 MB in 410 us; the head of the tensor up to the first 16-byte boundary and the tail go through the scalar loop)
    float m = 0.f;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, gsz = (size_t)gridDim.x * 256;
    const size_t head = min(n, (size_t)((16 - ((size_t)src & 15)) & 15) / 4);
    const float4 *v = (const float4 *)(src + head);
    const size_t n4 = (n - head) / 4;
    size_t i = gid;
    for (; i + 3 * gsz < n4; i += 4 * gsz) {
        const float4 a = v[i], b = v[i + gsz], c = v[i + 2 * gsz], d = v[i + 3 * gsz];
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                           fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))),
                           fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
    }
    for (; i < n4; i += gsz) {
        const float4 a = v[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
    }
    for (size_t e = gid; e < head; e += gsz) m = fmaxf(m, fabsf(src[e]));
    for (size_t e = head + 4 * n4 + gid; e < n; e += gsz) m = fmaxf(m, fabsf(src[e]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    // one atomic per workgroup, and only if it would raise the value: thousands of atomics on ONE address are serialised by the L2
    // (8192 of them cost more than reading 67 MB did)
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const unsigned bits = __float_as_uint(m);                               // non-negative floats order like their bits
        if (bits > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, bits);
    }
}

// the same over the [R][C] window of a matrix with row stride `stride` (a column slice of a wider tensor): one workgroup per 64 rows
__global__ __launch_bounds__(256) void cf_absmax_rows_kernel(const float *__restrict__ src, long R, int C, long stride, unsigned *__restrict__ out)
{
    float m = 0.f;
    const long r0 = (long)blockIdx.x * 64, r1 = min(R, r0 + 64);
    for (long r = r0; r < r1; r++)
        for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, fabsf(src[r * stride + c]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

__device__ __forceinline__ int cf_scale_exp(float wmax)
{
    if (!(wmax > 0.f) || !(wmax < INFINITY)) return 0;
    int e;
    frexpf(wmax, &e);                          // wmax = f 2^e, f in [0.5, 1)  ->  wmax 2^(3-e) in [4, 8)
    return 3 - e;
}

__global__ __launch_bounds__(256) void cf_split_w_kernel(const float *__restrict__ src, int R, int C, const unsigned *__restrict__ amax,
                                                         uint4 *__restrict__ pH, uint4 *__restrict__ pHs, uint4 *__restrict__ pM,
                                                         float *__restrict__ inv)
{
    const int S = cf_scale_exp(__uint_as_float(*amax));
    const float up = ldexpf(1.0f, S);
    if (blockIdx.x == 0 && threadIdx.x == 0) *inv = ldexpf(1.0f, -S);
    const int noct = (C + 7) / 8;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)R * noct) return;
    const int row = (int)(id % R), o = (int)(id / R);
    _Float16 H[8], Hs[8], M[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int k = o * 8 + e;
        const float v = (k < C ? src[(size_t)row * C + k] : 0.f) * up;
        H[e] = (_Float16)v;
        M[e] = (_Float16)(v - (float)H[e]);
        Hs[e] = (_Float16)((float)H[e] * 0x1p-12f);
    }
    pH[(size_t)o * R + row] = *(const uint4 *)H;
    pHs[(size_t)o * R + row] = *(const uint4 *)Hs;
    pM[(size_t)o * R + row] = *(const uint4 *)M;
}

// max over rows of sum_c |w[r][c]| (float bits, atomicMax): with |x| <= X it bounds every output of the layer,
// |y_r| <= |shift_r| + |scale_r| X sum_c |w_rc| -- what lets conv_f16_kernel write its OUTPUT as fp16 planes with a scale
// fixed before the first tile is computed.  One wave per row.
__global__ __launch_bounds__(256) void cf_rowsum_kernel(const float *__restrict__ src, int R, int C, unsigned *__restrict__ out)
{
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    float s = 0.f;
    if (row < R)
        for (int c = lane; c < C; c += 64) s += fabsf(src[(size_t)row * C + c]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    __shared__ float ws[4];
    if (lane == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(ws[0], ws[1]), fmaxf(ws[2], ws[3]))));
}

// x [R][C] (row-major, channel-last) or, with CFIRST, x [B][C][Npts] (R = B * Npts) -> h / m' planes
__device__ __forceinline__ int cf_act_exp(float xmax)
{
    if (!(xmax > 0.f) || !(xmax < INFINITY)) return 0;
    int e;
    frexpf(xmax, &e);                          // xmax = f 2^e, f in [0.5, 1)  ->  xmax 2^(12-e) in [2^11, 2^12)
    return 12 - e;
}

template <bool CFIRST>
__global__ __launch_bounds__(256) void cf_split_x_kernel(const float *__restrict__ src, long R, int C, int Npts, const unsigned *__restrict__ amax,
                                                         uint4 *__restrict__ ph, uint4 *__restrict__ pm, float *__restrict__ inv,
                                                         int *__restrict__ range_flag)
{
    const int T = cf_act_exp(__uint_as_float(*amax));
    const float up = ldexpf(1.0f, T);
    if (blockIdx.x == 0 && threadIdx.x == 0) *inv = ldexpf(1.0f, -T);
    const int noct = (C + 7) / 8;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= R * noct) return;
    const long row = id % R;
    const int o = (int)(id / R);
    _Float16 h[8], m[8];
    float big = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int k = o * 8 + e;
        float v = 0.f;
        if (k < C) v = (CFIRST ? src[((size_t)(row / Npts) * C + k) * Npts + row % Npts] : src[(size_t)row * C + k]) * up;
        big = fmaxf(big, fabsf(v));
        h[e] = (_Float16)v;
        m[e] = (_Float16)((v - (float)h[e]) * 4096.0f);
    }
    ph[(size_t)o * R + row] = *(const uint4 *)h;
    pm[(size_t)o * R + row] = *(const uint4 *)m;
    if (!(big <= 60000.f) && range_flag) *(volatile int *)range_flag = 1;         // inf / NaN inputs land here too
}

// Channel-last activations x [R][C] through an LDS transpose: a workgroup takes 64 rows x 32 octets (256 channels); it READS
// with consecutive threads on consecutive octets of one row (1 KB coalesced per 32 threads) and WRITES with consecutive
// threads on consecutive rows of one octet (the planes' own order: 1 KB per 64 threads).  The one-cell-per-thread kernel
// above reads 32 bytes per thread at a stride of a whole row: 150 us for conv5's 67 MB input; this one moves it at HBM speed.
#define CFS_ROWS 64
#define CFS_OCT 32
#define CFS_STRIDE (CFS_ROWS + 1)            // cells per octet line in LDS (+1: bank spread)
__global__ __launch_bounds__(256) void cf_split_x_cl_kernel(const float *__restrict__ src, long R, int C, const unsigned *__restrict__ amax,
                                                            uint4 *__restrict__ ph, uint4 *__restrict__ pm, float *__restrict__ inv,
                                                            int *__restrict__ range_flag, long sstride = 0, float rs = 4096.0f)
{
    if (sstride == 0) sstride = C;                 // row stride of src in floats; rs = 2^12: m' = f16((X - h) 2^12), rs = 1: unscaled residual
    __shared__ uint4 lh[CFS_OCT * CFS_STRIDE], lm[CFS_OCT * CFS_STRIDE];
    const int T = cf_act_exp(__uint_as_float(*amax));
    const float up = ldexpf(1.0f, T);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *inv = ldexpf(1.0f, -T);
    const int t = threadIdx.x;
    const long row0 = (long)blockIdx.x * CFS_ROWS;
    const int o0 = blockIdx.y * CFS_OCT;
    float big = 0.f;
    {
        const int oc = t & 31, o = o0 + oc;
#pragma unroll
        for (int pass = 0; pass < CFS_ROWS / 8; pass++) {
            const int r = pass * 8 + (t >> 5);
            const long row = row0 + r;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = 0.f;
            if (row < R && o * 8 < C) {
                const float *p = src + (size_t)row * sstride + o * 8;
                if (o * 8 + 8 <= C && (C & 3) == 0 && (sstride & 3) == 0) {
                    const f32x4 a = *(const f32x4 *)p, b = *(const f32x4 *)(p + 4);
                    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) if (o * 8 + e < C) v[e] = p[e];
                }
            }
            _Float16 h[8], m[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float x = v[e] * up;
                big = fmaxf(big, fabsf(x));
                h[e] = (_Float16)x;
                m[e] = (_Float16)((x - (float)h[e]) * rs);
            }
            lh[oc * CFS_STRIDE + r] = *(const uint4 *)h;
            lm[oc * CFS_STRIDE + r] = *(const uint4 *)m;
        }
    }
    __syncthreads();
    {
        const int r = t & 63;
        const long row = row0 + r;
        const int noct = (C + 7) / 8;
#pragma unroll
        for (int pass = 0; pass < CFS_OCT / 4; pass++) {
            const int oc = pass * 4 + (t >> 6), o = o0 + oc;
            if (row < R && o < noct) {
                ph[(size_t)o * R + row] = lh[oc * CFS_STRIDE + r];
                pm[(size_t)o * R + row] = lm[oc * CFS_STRIDE + r];
            }
        }
    }
    if (!(big <= 60000.f) && range_flag) *(volatile int *)range_flag = 1;
}

// ---------------------------------------------------------------------------------------------
// The GEMM.  Requires Cin % 16 == 0 and (Cout % 256 == 0, N % 256 == 0) or (Cout % 128 == 0, N % 512 == 0) (dispatcher checks).
// ---------------------------------------------------------------------------------------------
// NARROW: workgroup tile 128 (co) x 512 (n) for layers with Cout % 256 != 0 (FlowNet3D's 128-channel set-conv layers,
// models/flownet3d.py:125-242): the same eight 64 x 128 wave tiles arranged 2 x 4 instead of 4 x 2; W regions halve, x regions
// double (stage 44 KB, 48 DMA instructions per chunk -- four of them duplicates so that every wave issues six).
// AMAX: the fp32 epilogue also reduces max|y| (its own instantiation: the reduction costs the plain kernel its spill-free
// register allocation -- 218 VGPRs against 256 + scratch).
// GROUP: the pooled epilogue takes runs of 8 ... 64 points (a grouped layer's max over K neighbours) instead of 128; its own
// instantiation as well -- with the general code in the common kernel the 128-point pool ran at half speed (312 -> 612 us at PCN's
// conv4) and, one rewrite later, the fp32 epilogue's main loop lost 35 % to a different instruction schedule
// (tools/conv_f16_version_probe.sh, LABLOG R2.4h): this kernel's loop is sensitive to what is compiled around it.
// NPW = 2 (its own instantiation; wide tile, fp32 output only): the activation image carries an UNSCALED residual m = f16(X - h)
// (written by the two-plane EdgeConv kernel, out_mode 2), so the Hs weight plane is not read -- products M h + H m + H h, 12
// instead of 14 ds_read_b128 and 4 instead of 5 DMA pieces per wave and chunk, 32 KB stages.  An unscaled residual below 2^-14
// is a subnormal (2^-25 absolute in plane units): harmless for an image whose magnitudes sit near 2^12.
#ifdef CF_TIMELINE    // tools/probe_conv_timeline.hip: s_memrealtime (100 MHz) marks of every workgroup, through a device global
__device__ long long *g_cf_timeline;
#define CFM(i) { if (threadIdx.x == 0) g_cf_timeline[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define CFM(i)
#endif
// RESID (its own instantiation; wide tile, fp32 output, three weight planes): y = res + act(...), res [B][Cout][N] -- the residual
// connection of the pointer network's sublayers (utils/transformer.py:131-140 of the reference) in the GEMM's epilogue instead
// of a pass over both tensors.  The residual pointer travels in `obs`, which the fp32 epilogue does not otherwise read: the
// kernel's signature, and with it the other instantiations' code, stays as it was.
// OUT2 (two weight planes only; its own instantiation): the plane-image output carries an UNSCALED residual too, so that the next layer
// runs the two-plane form as well -- the pointer network's projections (utils/transformer.py:163-194 of the reference) then all do:
// 119 -> 94 us at 512 -> 1024 over 32768 rows, the difference between the three- and the two-plane main loop.
// SHIFTN (two weight planes, fp32 output; its own instantiation): `shift` is indexed by the output COLUMN n instead of the row co -- the
// bias of an nn.Linear whose rows are the "weight" operand and whose [Cout][Cin] matrix is the "activation" (l3d_split_f16_operand:
// the training path's y [rows][Cout] = x W^T + b, models/_rows.py).
template <bool NARROW, bool AMAX, bool GROUP, int NPW = 3, bool RESID = false, bool OUT2 = false, bool SHIFTN = false>
__global__ __launch_bounds__(512) void conv_f16_kernel(const uint4 *__restrict__ xh, const uint4 *__restrict__ xm,
                                                       const uint4 *__restrict__ wH, const uint4 *__restrict__ wHs,
                                                       const uint4 *__restrict__ wM, const float *__restrict__ winv,
                                                       const float *__restrict__ xinv, const float *__restrict__ scale, const float *__restrict__ shift,
                                                       int shift_bstride, int Bn, int Cin, int Cout, int N, int relu,
                                                       float *__restrict__ y, uint2 *__restrict__ oph, uint2 *__restrict__ opm,
                                                       float *__restrict__ oinv, const float *__restrict__ obs, float *__restrict__ ypool,
                                                       int pool, unsigned *__restrict__ amax_out, int amax_cdiv)
{
    constexpr int TM = NARROW ? 128 : CF_TM, TN = NARROW ? 512 : CF_TN;
    constexpr int WR = TM * 16, XR = TN * 16;                   // bytes of one (plane, octet) region of W / x
    static_assert(NPW == 3 || (NPW == 2 && !NARROW && !GROUP), "two weight planes: wide tile");
    static_assert(!RESID || (!NARROW && !AMAX && !GROUP), "residual epilogue: wide tile, fp32 output");
    static_assert(!OUT2 || (NPW == 2 && !AMAX && !RESID), "unscaled output image: the two-plane form");
    static_assert(!SHIFTN || (NPW == 2 && !AMAX && !RESID && !OUT2), "column-indexed shift: the plain two-plane form");
    constexpr int WBYTES = 2 * NPW * WR, STAGE = WBYTES + 4 * XR;
    constexpr int NPIECE = NARROW ? 6 : (NPW == 3 ? 5 : 4);     // DMA instructions per wave and chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = NARROW ? (wave & 1) : (wave & 3), wn = NARROW ? (wave >> 1) : (wave >> 2);
    CFM(0)
#ifdef CF_TIMELINE
    if (t == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id)); g_cf_timeline[(size_t)blockIdx.x * 8 + 7] = id; }
#endif
    // Tile order (1-D grid).  Workgroup L runs on XCD L % 8, each with its own L2: the Cout tiles of one point tile are
    // consecutive slots of ONE XCD, so the point tile's activation planes come from HBM once instead of once per XCD
    // that happens to host one of its Cout tiles.  It does not change the kernel's time (the reads were hidden), it
    // halves its HBM read traffic.
    int pt, ct;
    {
        const int nct = Cout / TM, npt = Bn * (N / TN), L = blockIdx.x;
        if (npt % 8 == 0) {
            const int xcd = L & 7, slot = L >> 3;
            ct = slot % nct;
            pt = (slot / nct) * 8 + xcd;
        } else {
            ct = L % nct;
            pt = L / nct;
        }
    }
    const int ntn = N / TN;
    const int n0 = (pt % ntn) * TN, co0 = ct * TM, b = pt / ntn;
    const int nk = Cin / 16;
    const size_t BN = (size_t)Bn * N;

    // ---- DMA pieces of 64 rows (1 KB): W 6 regions x TM/64, x 4 regions x TN/64 -- 40 per chunk, 5 per wave (NARROW: 44, 6
    // per wave; wave 7's last four repeat x pieces 0..3: the same bytes to the same place, so that every wave counts alike)
    const uint4 *src[NPIECE];
    size_t stride[NPIECE];
    int dst[NPIECE];
    constexpr int WQ = TM / 64, XQ = TN / 64, NWP = 2 * NPW * WQ, NXP = 4 * XQ;
#pragma unroll
    for (int i = 0; i < NPIECE; i++) {
        int q = wave * NPIECE + i;
        if (q >= NWP + NXP) q -= NXP;
        if (q < NWP) {
            const int reg = q / WQ, p = reg >> 1, kg = reg & 1, quarter = q % WQ;
            const uint4 *pl = p == 0 ? wH : ((NPW == 3 && p == 1) ? wHs : wM);
            src[i] = pl + (size_t)kg * Cout + co0 + quarter * 64 + lane;
            stride[i] = 2 * (size_t)Cout;
            dst[i] = reg * WR + quarter * 1024;
        } else {
            const int q2 = q - NWP, reg = q2 / XQ, p = reg >> 1, kg = reg & 1, quarter = q2 % XQ;
            const uint4 *pl = p == 0 ? xh : xm;
            src[i] = pl + (size_t)kg * BN + (size_t)b * N + n0 + quarter * 64 + lane;
            stride[i] = 2 * BN;
            dst[i] = WBYTES + reg * XR + quarter * 1024;
        }
    }
    auto issue_one = [&](int stage, int i) {
#ifdef CF_ABL      // timing ablation (results are garbage): bit 0 = no W pieces after the prologue, bit 1 = no x pieces
        {
            int q_ = wave * NPIECE + i;
            if (q_ >= NWP + NXP) q_ -= NXP;
            if (((CF_ABL & 1) && q_ < NWP) || ((CF_ABL & 2) && q_ >= NWP)) return;
        }
#endif
        __builtin_amdgcn_global_load_lds((cf_gbl_ptr_t)src[i], (cf_lds_ptr_t)(lds + stage * STAGE + dst[i]), 16, 0, 0);
        src[i] += stride[i];
    };
    auto issue = [&](int stage) {
#pragma unroll
        for (int i = 0; i < NPIECE; i++) issue_one(stage, i);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;

    const int kgl = lane >> 5;
    const int a_off = kgl * WR + (wm * 64 + (lane & 31)) * 16;                       // + a*512 + p*2*WR
    const int b_off = WBYTES + kgl * XR + (wn * 128 + (lane & 31)) * 16;             // + c*512 + p*2*XR

    issue(0);
    if (nk > 1) issue(1);
    int stage = 0;
#ifdef CF_TIMING      // tools/probe_conv_f16.hip: where a wave's chunk time goes (s_memtime sums, written over y)
    long long tq[5] = {0, 0, 0, 0, 0}, tp = __builtin_amdgcn_s_memtime();
#define CFT(n) { const long long now_ = __builtin_amdgcn_s_memtime(); tq[n] += now_ - tp; tp = now_; }
#else
#define CFT(n)
#endif
    for (int kc = 0; kc < nk; kc++) {
        CFT(4)
        // this wave's pieces of chunk kc have landed (chunk kc+1's five may still be in flight) ...
        if (kc + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NARROW) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (NPW == 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        CFT(0)
        __builtin_amdgcn_s_barrier();            // ... and so have everybody else's; stage (kc+2)%3 was last read in chunk kc-1
#ifdef CF_TIMELINE
        if (kc == 0) CFM(1)
#endif
        CFT(1)
        const int nst = stage == 0 ? 2 : stage - 1;                                   // (kc + 2) % 3
        const bool more = kc + 2 < nk;
        CFT(2)
        const unsigned char *base = lds + stage * STAGE;
#if defined(CF_ABL) && (CF_ABL & 8)     // timing ablation: the twelve fragments are read in the first chunk only
        static_assert(true, "");
        f16x8 A[2][NPW], Bf[4][2];
        if (kc == 0 || ((const volatile int *)winv)[0] == 0x7fffffff)
#else
        f16x8 A[2][NPW], Bf[4][2];
#endif
        {
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int c = 0; c < 4; c++) Bf[c][p] = *(const f16x8 *)(base + b_off + c * 512 + p * 2 * XR);
#pragma unroll
        for (int p = NPW - 1; p >= 0; p--)
#pragma unroll
            for (int a = 0; a < 2; a++) A[a][p] = *(const f16x8 *)(base + a_off + a * 512 + p * 2 * WR);
        }
        // three products, smallest first: M h, Hs m', H h
        // The five DMA instructions of chunk kc+2 are spread over the chunk's 24 MFMAs, one behind every fifth: the CU's
        // vector-memory path takes 16 cycles per 1 KB instruction and all eight waves share it -- issued as one block behind
        // the barrier (40 instructions at once) they cost each wave ~600 cycles of issue stall (tools/probe_conv_f16.hip)
        // during which it feeds the matrix pipe nothing.
#pragma unroll
        for (int n = 0; n < 24; n++) {
            const int prod = n >> 3, a = (n >> 2) & 1, c = n & 3;
            const int pa = NPW == 3 ? (prod == 0 ? 2 : (prod == 1 ? 1 : 0)) : (prod == 0 ? 1 : 0);      // M, Hs | H, H
            const int pb = prod == 1 ? 1 : 0;
            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[a][pa], Bf[c][pb], acc[a][c], 0, 0, 0);
            constexpr int GAP = NARROW ? 4 : (NPW == 3 ? 5 : 6);  // behind MFMA 3, 8, 13, 18, 23 (NARROW: 3, 7, .. 23; two planes: 3, 9, 15, 21)
            if (n % GAP == 3) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) issue_one(nst, n / GAP);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stage = stage == 2 ? 0 : stage + 1;
        CFT(3)
    }
#ifdef CF_TIMING
    if (lane == 0) {
        long long *o = (long long *)y + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int n = 0; n < 5; n++) o[n] = tq[n];
    }
    {
        float keep = 0.f;                        // keep the MFMAs alive
        for (int a = 0; a < 2; a++) for (int c = 0; c < 4; c++) for (int r = 0; r < 16; r++) keep += acc[a][c][r];
        if (keep == 12345.678f) y[0] = keep;
    }
    return;
#endif

    CFM(2)
    // ---- epilogue: D[co = 32a + (r&3) + 8(r>>2) + 4(lane>>5)][n = 32c + (lane&31)]
    const float inv = *winv * *xinv;             // 2^-S 2^-T: exact
    if (oph || ypool) {
        // (1) Output as the fp16 plane image of the NEXT f16x2 layer ([Cout/8][B N][8], h | m' of y 2^To) instead of fp32: the
        // scale is fixed from a bound, |y| <= max|shift| + max|scale| (max_r sum_c |w_rc|) max|x| with max|x| <= 2^12 2^-T
        // (obs = {max|shift|, max|scale|}, the row-sum maximum sits behind 2^-S in the weight image).  A lane holds 4
        // consecutive channels of an octet, its partner (lane ^ 32) the other 4: each writes its 8-byte half of the cell.
        // (2) ypool [B][Cout][N/pool]: maxima over runs of `pool` consecutive points (lanes of a 32-point tile by shuffles, tiles
        // in registers).  pool = 128: a global max-pool (pcn.py:115,124, pointnet.py:49 + pooling.py) finishes with a reduce over
        // N/128 values; pool = K: the max over a group's K neighbours (flownet3d.py:179, :234) -- in both cases the layer's
        // [B,Cout,N] output is never written.  Both outputs may be asked for (pcn.py:115-119 pools conv2's output AND feeds it on).
        float up = 1.f;
        if (oph) {
            float bound = fmaf(obs[1] * winv[2], 4096.0f * *xinv, obs[0]) * 1.000001f;
            int e = 0;
            if (bound > 0.f && bound < 3.0e38f) (void)frexpf(bound, &e);
            up = ldexpf(1.f, 12 - e);
            if (blockIdx.x == 0 && t == 0) *oinv = ldexpf(1.f, e - 12);
        }
      if constexpr (!GROUP) {
        const size_t rows = (size_t)Bn * N;
        const int half = lane >> 5, NP = N / 128;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const int cob = co0 + wm * 64 + a * 32 + 8 * gq + 4 * half;          // this lane's 4 channels: cob .. cob + 3
                float sc[4], sh[4], vmax[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    sc[u] = (scale ? scale[cob + u] : 1.f) * inv;
                    sh[u] = shift ? shift[(size_t)b * shift_bstride + cob + u] : 0.f;
                    vmax[u] = -INFINITY;
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        v[u] = acc[a][c][4 * gq + u] * sc[u] + sh[u];
                        if (relu) v[u] = l3d_act(v[u], relu);
                        vmax[u] = fmaxf(vmax[u], v[u]);
                    }
                    if (oph) {
                        uint32_t h0, h1, m0, m1;
                        if constexpr (OUT2) {
                            af_split_x_unscaled(v[0], v[1], up, h0, m0);
                            af_split_x_unscaled(v[2], v[3], up, h1, m1);
                        } else {
                            af_split_x(v[0], v[1], up, h0, m0);
                            af_split_x(v[2], v[3], up, h1, m1);
                        }
                        const size_t row = (size_t)b * N + n0 + wn * 128 + c * 32 + (lane & 31);
                        const size_t cellh = ((size_t)(cob >> 3) * rows + row) * 2 + half;
                        oph[cellh] = make_uint2(h0, h1);
                        opm[cellh] = make_uint2(m0, m1);
                    }
                }
                if (ypool) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
#pragma unroll
                        for (int d = 1; d < 32; d <<= 1) vmax[u] = fmaxf(vmax[u], __shfl_xor(vmax[u], d, 64));
                    }
                    if ((lane & 31) == 0) {
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            ypool[((size_t)b * Cout + cob + u) * NP + (n0 + wn * 128) / 128] = vmax[u];
                    }
                }
            }
      } else {
        const size_t rows = (size_t)Bn * N;
        const int half = lane >> 5, NP = N / pool;               // pool: 8, 16, 32, 64 or 128 consecutive points per maximum
        const int span = pool < 32 ? pool : 32;                  // lanes of one 32-point tile that share a maximum
        // (unroll(full): with a plain `unroll` the compiler kept these two loops rolled -- the body is long -- and indexed a private copy of
        // the accumulators: 576 bytes of scratch)
#pragma clang loop unroll(full)
        for (int a = 0; a < 2; a++)
#pragma clang loop unroll(full)
            for (int gq = 0; gq < 4; gq++) {
                const int cob = co0 + wm * 64 + a * 32 + 8 * gq + 4 * half;          // this lane's 4 channels: cob .. cob + 3
                float sc[4], sh[4], pv[4][4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    sc[u] = (scale ? scale[cob + u] : 1.f) * inv;
                    sh[u] = shift ? shift[(size_t)b * shift_bstride + cob + u] : 0.f;
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        v[u] = acc[a][c][4 * gq + u] * sc[u] + sh[u];
                        if (relu) v[u] = l3d_act(v[u], relu);
                        pv[c][u] = v[u];
                    }
                    if (oph) {
                        uint32_t h0, h1, m0, m1;
                        af_split_x(v[0], v[1], up, h0, m0);
                        af_split_x(v[2], v[3], up, h1, m1);
                        const size_t row = (size_t)b * N + n0 + wn * 128 + c * 32 + (lane & 31);
                        const size_t cellh = ((size_t)(cob >> 3) * rows + row) * 2 + half;
                        oph[cellh] = make_uint2(h0, h1);
                        opm[cellh] = make_uint2(m0, m1);
                    }
                }
                if (ypool) {
                    // tiles of one run first (registers), then across the `span` lanes of a tile: shuffles with constant distances
                    // (DPP, not ds_bpermute) and as few of them as the run length allows -- 5 per channel for pool = 128
                    if (pool >= 64) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            pv[0][u] = fmaxf(pv[0][u], pv[1][u]);
                            pv[2][u] = fmaxf(pv[2][u], pv[3][u]);
                            if (pool == 128) pv[0][u] = fmaxf(pv[0][u], pv[2][u]);
                        }
                    }
                    // tiles per maximum: 4 (pool 128), 2 (pool 64), 1 (pool <= 32).  Written out per case: `if (c % step) continue` with a
                    // runtime step made pv[][] an indexed private array (576 bytes of scratch)
                    auto finish = [&](auto c_c) {
                        constexpr int c = decltype(c_c)::value;
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            float m = pv[c][u];
#pragma unroll
                            for (int d = 1; d < 32; d <<= 1)
                                if (d < span) m = fmaxf(m, __shfl_xor(m, d, 64));
                            pv[c][u] = m;
                        }
                        if (((lane & 31) & (span - 1)) == 0) {
                            const int n = n0 + wn * 128 + c * 32 + (lane & 31);
#pragma unroll
                            for (int u = 0; u < 4; u++) ypool[((size_t)b * Cout + cob + u) * NP + n / pool] = pv[c][u];
                        }
                    };
                    finish(std::integral_constant<int, 0>{});
                    if (pool <= 64) finish(std::integral_constant<int, 2>{});
                    if (pool <= 32) { finish(std::integral_constant<int, 1>{}); finish(std::integral_constant<int, 3>{}); }
                }
            }
      }
        return;
    }
    // the GROUP instantiation is only launched with pooled maxima (cf_launch): without this its fp32 epilogue is compiled too, with the
    // loop over the accumulator blocks left rolled -- an indexed private copy of the accumulators, 576 bytes of scratch
    if constexpr (GROUP) return;
    float *yb = y + (size_t)b * Cout * N;
    const float *rb = RESID ? obs + (size_t)b * Cout * N : nullptr;
    float shn[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (SHIFTN) {
        if (shift) {
#pragma unroll
            for (int c = 0; c < 4; c++) shn[c] = shift[n0 + wn * 128 + c * 32 + (lane & 31)];
        }
    }
    float amax_nan = 0.f;
    float amax_run = 0.f;                          // AMAX: max |y| of this lane's 128 outputs (round 6: a running maximum; parking |v| in
                                                   // the dead accumulators for a later reduction cost the instantiation 12 - 16 spilled registers)
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float sc = (scale ? scale[co] : 1.f) * inv;
            const float sh = (!SHIFTN && shift) ? shift[(size_t)b * shift_bstride + co] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float v = acc[a][c][r] * sc + (SHIFTN ? shn[c] : sh);
                if (relu) v = l3d_act(v, relu);
                if constexpr (RESID) v = rb[(size_t)co * N + n0 + wn * 128 + c * 32 + (lane & 31)] + v;
                // DGCNN's conv5 (the two-plane instantiation): 134 MB of fp32 output that nothing on the chip reads back soon.  As ordinary
                // stores its dirty lines are still being written back while the NEXT launches run -- the step's EdgeConv kernel is
                // 128 us behind a kNN launch and 140 us behind this kernel (tools/ec_instep_probe.py: its dependent index / coordinate
                // gathers queue behind the write-back).  Nontemporal stores cost this kernel 1.5 us and give EdgeConv 3-5 back.
#if defined(CF_ABL) && (CF_ABL & 4)     // timing ablation: no output stores (a store that never fires keeps the accumulators alive)
                if (v == 12345.678f) yb[(size_t)co * N + n0 + wn * 128 + c * 32 + (lane & 31)] = v;
#else
#ifndef CF_STORE
#define CF_STORE 1     // the two-plane instantiation's output stores: 0 plain, 1 nt (default), 2 sc1 (write-through, agent), 3 sc0 sc1 (system)
#endif
                float *dst_ = &yb[(size_t)co * N + n0 + wn * 128 + c * 32 + (lane & 31)];
                if constexpr (NPW == 2 && !AMAX && !RESID && !SHIFTN) {
                    if (CF_STORE == 1) __builtin_nontemporal_store(v, dst_);
                    else if (CF_STORE == 2) __hip_atomic_store(dst_, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else if (CF_STORE == 3) __hip_atomic_store(dst_, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    else *dst_ = v;
                } else *dst_ = v;
#endif
                // (asm: left to the compiler the maxima are re-associated into ONE tree behind the last store, every v alive until then)
                if constexpr (AMAX) {
                    asm volatile("v_max_f32_e64 %0, %0, |%1|" : "+v"(amax_run) : "v"(v));
                    asm volatile("v_fma_f32 %0, %1, 0, %0" : "+v"(amax_nan) : "v"(v));      // v_max drops a NaN: 0 * v keeps it (and an inf)
                }
            }
        }
#ifdef CF_TIMELINE
    CFM(3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CFM(4)
    __syncthreads();
    CFM(5)
#endif
    if constexpr (AMAX) {
        // max|y| per group of amax_cdiv output channels (amax_cdiv % 256 == 0: a workgroup's 256 channels lie in one group), as
        // float bits: the consumer's operand scale (attention_f16.hip takes max|q|, |k|, |v| of a fused q|k|v projection from
        // here instead of a pass over the three tensors).  One atomic per workgroup, skipped when it would not raise the value.
        float *red = (float *)lds;                 // the stages are dead: every wave is past its last fragment read ...
        float big = amax_run;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) big = fmaxf(big, __shfl_xor(big, d, 64));
        if (__builtin_amdgcn_ballot_w64(amax_nan != amax_nan) != 0) big = __builtin_nanf("");   // fmaxf would drop it
        __syncthreads();                           // ... after this barrier
        if (lane == 0) red[wave] = big;
        __syncthreads();
        if (t == 0) {
            float m = red[0];
            for (int w = 1; w < 8; w++) m = (m != m || red[w] != red[w]) ? __builtin_nanf("") : fmaxf(m, red[w]);
            unsigned *dst = amax_out + co0 / amax_cdiv;
            if (!(m <= __uint_as_float(__atomic_load_n(dst, __ATOMIC_RELAXED)))) atomicMax(dst, __float_as_uint(m));   // NaN goes through too
        }
    }
}

#ifdef CF_PERSIST    // tools/experiments/conv_f16_persist.inc: persistent two-tile form with an LDS-transposed 16-byte epilogue (LABLOG R6.1; measured, slower in the step)
#include "../../tools/experiments/conv_f16_persist.inc"
#endif
#ifdef CF_HALF       // tools/experiments/conv_f16_half.inc: 256-thread workgroups, two per CU (LABLOG R4.4; measured, not faster)
#include "../../tools/experiments/conv_f16_half.inc"
#endif
// Sizes of the plane images (common.h: l3d_f16_plane_bytes / l3d_f16_act_bytes / l3d_conv_f16_weight_bytes), one exported spelling:
// kind 0 = ONE fp16 plane of a [rows][cols] matrix in the tiled layout; 1 = an activation image (h | m' planes + 16 bytes:
// 2^-T, scratch); 2 = a weight image of [rows = Cout][cols = Cin] (H | Hs | M planes + 16 bytes: 2^-S, |w| maximum, row-sum maximum)
extern "C" size_t l3d_f16_image_bytes(int kind, long rows, int cols)
{
    return kind == 0 ? l3d_f16_plane_bytes(rows, cols) : (kind == 1 ? l3d_f16_act_bytes(rows, cols) : l3d_conv_f16_weight_bytes((int)rows, cols));
}

// weights [Cout][Cin] fp32 -> dst = the weight image (device); two small launches
extern "C" int l3d_conv_f16_split_weights(const float *w, int Cout, int Cin, void *dst, l3d_stream_t stream)
{
    L3D_REQUIRE(w && dst && Cout > 0 && Cin > 0);
    hipStream_t st = (hipStream_t)stream;
    const size_t pb = l3d_f16_plane_bytes(Cout, Cin);
    unsigned char *d = (unsigned char *)dst;
    float *inv = (float *)(d + 3 * pb);
    unsigned *amax = (unsigned *)(d + 3 * pb + 4);
    if (hipMemsetAsync(amax, 0, 8, st) != hipSuccess) return L3D_ERR_LAUNCH;       // |w| maximum and row-sum maximum
    const size_t n = (size_t)Cout * Cin;
    const long nblk = l3d_divup((long)n, 256);
    hipLaunchKernelGGL(cf_absmax_kernel, dim3((unsigned)(nblk > 256 ? 256 : nblk)), dim3(256), 0, st, w, n, amax);
    hipLaunchKernelGGL(cf_rowsum_kernel, dim3((unsigned)l3d_divup(Cout, 4)), dim3(256), 0, st, w, Cout, Cin, amax + 1);
    const long cells = (long)Cout * ((Cin + 7) / 8);
    hipLaunchKernelGGL(cf_split_w_kernel, dim3((unsigned)l3d_divup(cells, 256)), dim3(256), 0, st, w, Cout, Cin, (const unsigned *)amax,
                       (uint4 *)d, (uint4 *)(d + pb), (uint4 *)(d + 2 * pb), inv);
    return l3d_check_launch();
}

// activations: x [rows][C] (channel_first = 0) or [B][C][Npts] (channel_first = 1, rows = B*Npts) -> dst = activation image
// (T from the tensor's own maximum: one read pass for the maximum, one for the split)
extern "C" int l3d_split_f16_rows(const float *x, long rows, int C, int channel_first, int Npts, void *dst, int *range_flag,
                                  l3d_stream_t stream)
{
    L3D_REQUIRE(x && dst && rows > 0 && C > 0 && (!channel_first || (Npts > 0 && rows % Npts == 0)));
    hipStream_t st = (hipStream_t)stream;
    const size_t pb = l3d_f16_plane_bytes(rows, C);
    unsigned char *d = (unsigned char *)dst;
    float *inv = (float *)(d + 2 * pb);
    unsigned *amax = (unsigned *)(d + 2 * pb + 4);
    if (hipMemsetAsync(amax, 0, 4, st) != hipSuccess) return L3D_ERR_LAUNCH;
    const size_t n = (size_t)rows * C;
    const long nblk = l3d_divup((long)n, 4096);
    hipLaunchKernelGGL(cf_absmax_kernel, dim3((unsigned)(nblk > 2048 ? 2048 : nblk)), dim3(256), 0, st, x, n, amax);
    const long cells = rows * ((C + 7) / 8);
    if (channel_first)
        hipLaunchKernelGGL(cf_split_x_kernel<true>, dim3((unsigned)l3d_divup(cells, 256)), dim3(256), 0, st, x, rows, C,
                           Npts, (const unsigned *)amax, (uint4 *)d, (uint4 *)(d + pb), inv, range_flag);
    else
        hipLaunchKernelGGL(cf_split_x_cl_kernel, dim3((unsigned)l3d_divup(rows, CFS_ROWS), (unsigned)l3d_divup((C + 7) / 8, CFS_OCT)), dim3(256), 0,
                           st, x, rows, C, (const unsigned *)amax, (uint4 *)d, (uint4 *)(d + pb), inv, range_flag);
    return l3d_check_launch();
}

// One operand of a training-path GEMM (models/_rows.py: an nn.Linear over rows and its dgrad on the two-plane form of the f16x2 kernel):
// x [rows][C] fp32 with row stride `row_stride` -> two fp16 planes h | m of x 2^T with an UNSCALED residual, T from the window's own maximum.
//   kind 0: an activation image (l3d_f16_image_bytes(1, rows, C)): the x_planes operand of l3d_pointwise_conv_f16 with L3D_CONV_F16_TWO_PLANE
//   kind 1: the same two planes in the slots of a WEIGHT image (l3d_f16_image_bytes(2, rows, C): H at plane 0, M at plane 2, 2^-T behind
//           them) -- the w_planes operand of the two-plane form, which reads exactly those.  With the rows of a batch as the "weight" and
//           the layer's [Cout][Cin] matrix as the "activation" the kernel's [B][Cout][N] output IS y [rows][Cout], row-major.
extern "C" int l3d_split_f16_operand(const float *x, long rows, int C, long row_stride, int kind, void *dst, int *range_flag, l3d_stream_t stream)
{
    L3D_REQUIRE(x && dst && rows > 0 && C > 0 && row_stride >= C && (kind == 0 || kind == 1));
    if ((((size_t)dst) & 15) || rows > 2147483647L) return L3D_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const size_t pb = l3d_f16_plane_bytes(rows, C);
    unsigned char *d = (unsigned char *)dst;
    uint4 *ph = (uint4 *)d, *pm = (uint4 *)(d + (kind ? 2 : 1) * pb);
    float *inv = (float *)(d + (kind ? 3 : 2) * pb);
    unsigned *amax = (unsigned *)(inv + 1);
    if (hipMemsetAsync(inv, 0, 16, st) != hipSuccess) return L3D_ERR_LAUNCH;
    if (row_stride == C) {
        const size_t n = (size_t)rows * C;
        const long nblk = l3d_divup((long)n, 4096);
        hipLaunchKernelGGL(cf_absmax_kernel, dim3((unsigned)(nblk > 2048 ? 2048 : nblk)), dim3(256), 0, st, x, n, amax);
    } else {
        hipLaunchKernelGGL(cf_absmax_rows_kernel, dim3((unsigned)l3d_divup(rows, 64L)), dim3(256), 0, st, x, rows, C, row_stride, amax);
    }
    hipLaunchKernelGGL(cf_split_x_cl_kernel, dim3((unsigned)l3d_divup(rows, (long)CFS_ROWS), (unsigned)l3d_divup((C + 7) / 8, CFS_OCT)), dim3(256), 0,
                       st, x, rows, C, (const unsigned *)amax, ph, pm, inv, range_flag, row_stride, 1.0f);
    return l3d_check_launch();
}

// x_act: an activation image (l3d_f16_act_bytes), w_planes: a weight image (l3d_conv_f16_weight_bytes)
static int cf_launch(const void *x_planes, const void *w_planes, const float *scale, const float *shift, int shift_bstride, int B,
                     int Cin, int Cout, int N, int relu, float *y, void *out_img, const float *obs, float *ypool, int pool,
                     unsigned *amax_out, int amax_cdiv, hipStream_t st, bool two_plane = false, const float *resid = nullptr,
                     bool out_unscaled = false, bool shift_n = false)
{
    // wide tile (256 x 256) when Cout allows it, else the narrow one (128 x 512)
    const bool narrow = Cout % CF_TM != 0;
    const int tm = narrow ? 128 : CF_TM, tn = narrow ? 512 : CF_TN;
    if (Cout % tm || N % tn || Cin % 16 || B > 65535 || (((size_t)x_planes) & 15) || (((size_t)w_planes) & 15) ||
        (((size_t)out_img) & 15) || (amax_out && amax_cdiv % tm))
        return L3D_ERR_UNSUPPORTED;
    if (ypool && pool != 8 && pool != 16 && pool != 32 && pool != 64 && pool != 128) return L3D_ERR_UNSUPPORTED;
    const size_t xpb = l3d_f16_plane_bytes((long)B * N, Cin), wpb = l3d_f16_plane_bytes(Cout, Cin);
    const size_t opb = l3d_f16_plane_bytes((long)B * N, Cout);
    const unsigned char *xp = (const unsigned char *)x_planes, *wp = (const unsigned char *)w_planes;
    unsigned char *op = (unsigned char *)out_img;
    dim3 grid((unsigned)((size_t)(N / tn) * (Cout / tm) * B)), block(512);
    if (!ypool) pool = 128;
#define CF_ARGS (const uint4 *)xp, (const uint4 *)(xp + xpb), (const uint4 *)wp, (const uint4 *)(wp + wpb), (const uint4 *)(wp + 2 * wpb), \
                (const float *)(wp + 3 * wpb), (const float *)(xp + 2 * xpb), scale, shift, shift_bstride, B, Cin, Cout, N, relu, y,      \
                (uint2 *)op, op ? (uint2 *)(op + opb) : nullptr, op ? (float *)(op + 2 * opb) : nullptr, obs, ypool, pool, amax_out, amax_cdiv
    const size_t nlds = 3 * (6 * 128 * 16 + 4 * 512 * 16);
    const bool group = ypool && pool != 128;
    if (group && amax_out) return L3D_ERR_UNSUPPORTED;
    constexpr size_t lds2 = 3 * (4 * CF_TM * 16 + 4 * CF_TN * 16);             // the two-plane form's three stages
    if (out_unscaled && !(two_plane && out_img && !ypool)) return L3D_ERR_UNSUPPORTED;
    if (shift_n) {                                 // the training path's rows-as-weights product: plain two-plane form, bias along n
        if (!two_plane || narrow || group || amax_out || ypool || out_img || resid || !y || shift_bstride) return L3D_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((conv_f16_kernel<false, false, false, 2, false, false, true>), grid, block, lds2, st, CF_ARGS);
        return l3d_check_launch();
    }
    if (resid) {
        if (narrow || group || amax_out || ypool || out_img || !y) return L3D_ERR_UNSUPPORTED;
        obs = resid;
        if (two_plane) hipLaunchKernelGGL((conv_f16_kernel<false, false, false, 2, true>), grid, block, lds2, st, CF_ARGS);
        else hipLaunchKernelGGL((conv_f16_kernel<false, false, false, 3, true>), grid, block, CF_LDS, st, CF_ARGS);
        return l3d_check_launch();
    }
    if (two_plane && (amax_out || out_img)) {
        // the pointer network's projections on two-plane images: operand maxima out of the epilogue, or an (unscaled) plane image
        if (narrow || group || ypool || (amax_out && !y) || (out_img && (!out_unscaled || amax_out))) return L3D_ERR_UNSUPPORTED;
        if (amax_out) hipLaunchKernelGGL((conv_f16_kernel<false, true, false, 2>), grid, block, lds2, st, CF_ARGS);
        else hipLaunchKernelGGL((conv_f16_kernel<false, false, false, 2, false, true>), grid, block, lds2, st, CF_ARGS);
        return l3d_check_launch();
    }
    if (two_plane) {
        if (narrow || group || amax_out || ypool || out_img || !y) return L3D_ERR_UNSUPPORTED;
#ifdef CF_HALF
        hipLaunchKernelGGL(conv_f16_half_kernel, dim3((unsigned)((size_t)(N / CFH_TN) * (Cout / CFH_TM) * B)), dim3(256), CFH_LDS, st,
                           (const uint4 *)xp, (const uint4 *)(xp + xpb), (const uint4 *)wp, (const uint4 *)(wp + 2 * wpb),
                           (const float *)(wp + 3 * wpb), (const float *)(xp + 2 * xpb), scale, shift, shift_bstride, B, Cin, Cout, N, relu, y);
#else
#ifdef CF_PERSIST
        {
            // persistent form: one workgroup per CU (a multiple of 8, so that a workgroup's tiles stay on its XCD's Cout group)
            static bool ok = [] { return hipFuncSetAttribute((const void *)conv_f16_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             CFP_LDS) == hipSuccess; }();
            static const bool off = [] { const char *e = getenv("L3D_CONV5_PERSIST"); return e && e[0] == '0'; }();
            const int ntiles = (int)grid.x;
            if (ok && !off && Cin >= 32) {
                const int wgs = ntiles < 256 ? ntiles : 256;
                hipLaunchKernelGGL(conv_f16_persist_kernel, dim3(wgs), block, CFP_LDS, st, (const uint4 *)xp, (const uint4 *)(xp + xpb),
                                   (const uint4 *)wp, (const uint4 *)(wp + 2 * wpb), (const float *)(wp + 3 * wpb), (const float *)(xp + 2 * xpb),
                                   scale, shift, shift_bstride, B, Cin, Cout, N, relu, y, ntiles);
                return l3d_check_launch();
            }
        }
#endif
        hipLaunchKernelGGL((conv_f16_kernel<false, false, false, 2>), grid, block, 3 * (4 * CF_TM * 16 + 4 * CF_TN * 16), st, CF_ARGS);
#endif
        return l3d_check_launch();
    }
    if (narrow && group)         hipLaunchKernelGGL((conv_f16_kernel<true, false, true>), grid, block, nlds, st, CF_ARGS);
    else if (narrow && amax_out) hipLaunchKernelGGL((conv_f16_kernel<true, true, false>), grid, block, nlds, st, CF_ARGS);
    else if (narrow)             hipLaunchKernelGGL((conv_f16_kernel<true, false, false>), grid, block, nlds, st, CF_ARGS);
    else if (group)              hipLaunchKernelGGL((conv_f16_kernel<false, false, true>), grid, block, CF_LDS, st, CF_ARGS);
    else if (amax_out)           hipLaunchKernelGGL((conv_f16_kernel<false, true, false>), grid, block, CF_LDS, st, CF_ARGS);
    else                         hipLaunchKernelGGL((conv_f16_kernel<false, false, false>), grid, block, CF_LDS, st, CF_ARGS);
#undef CF_ARGS
    return l3d_check_launch();
}

// THE entry point of the f16x2 layer (one kernel family, one name; round 3 had six spellings of it).  Outputs, any combination
// the kernel family offers:
//   y         fp32 [B][Cout][N], or NULL
//   residual  y = residual + act(...): residual and y [B][Cout][N] fp32, distinct buffers (wide tile, fp32 output only)
//   out_img   the output as the activation image of the NEXT f16x2 layer (l3d_f16_act_bytes(B N, Cout) bytes) instead of / beside
//             ypool; needs obs = two device floats {max|shift| over every (b, co), max|scale|} (max|scale| = 1 without a scale)
//             from which, with the weight image's row-sum maximum and the input image's scale, the kernel fixes the plane scale
//   ypool     [B][Cout][N/pool] fp32 maxima over runs of `pool` (8, 16, 32, 64 or 128) consecutive points: pool = 128 makes a
//             global max-pool a reduce over N/128 values per channel (pcn.py:110-124), pool = K the max over a group's K
//             neighbours (flownet3d.py:179, :234); the [B,Cout,N] output is then never written
//   amax_out  max|y| per group of amax_cdiv output channels (amax_cdiv % 256 == 0) as float bits, atomicMax into
//             amax_out[Cout / amax_cdiv] (the caller zeroes them; needs y): transformer.py:183-189's fused q|k|v projection
//             hands the attention kernel its operand maxima
// flags: 1 (L3D_CONV_F16_TWO_PLANE) -- the input image's residual plane is UNSCALED (m = f16(X - h), written by
//        l3d_edgeconv_forward_f16b with out_mode 2, l3d_layernorm_planes / l3d_attention_forward_f16b / this kernel when asked): the Hs
//        plane of the weight image is not read (wide tile; y, y + residual, y + amax_out, or out_img with flag 2).
//        2 (L3D_CONV_F16_OUT_UNSCALED) -- out_img gets an unscaled residual plane as well (needs flag 1).
//        4 (L3D_CONV_F16_SHIFT_N) -- shift [N] is indexed by the output column (needs flag 1; y only): with w_planes = the rows of a batch
//        (l3d_split_f16_operand kind 1) and x_planes = a layer's [Cout][Cin] matrix (kind 0), y [1][rows][Cout] = x W^T + b.
// shift may be per cloud (shift_bstride = Cout).  Cin % 16 == 0; Cout % 256 == 0 and N % 256 == 0, or Cout % 128 == 0 and N % 512 == 0.
extern "C" int l3d_pointwise_conv_f16(const void *x_planes, const void *w_planes, const float *scale, const float *shift,
                                      int shift_bstride, int B, int Cin, int Cout, int N, int relu, int flags, float *y,
                                      const float *residual, void *out_img, const float *obs, float *ypool, int pool,
                                      void *amax_out, int amax_cdiv, l3d_stream_t stream)
{
    L3D_REQUIRE(x_planes && w_planes && (y || out_img || ypool) && (!out_img || obs) && (!residual || y) && (!amax_out || (y && amax_cdiv > 0)) &&
                B > 0 && Cin > 0 && Cout > 0 && N > 0 && (flags & ~7) == 0);
    if (y && (out_img || ypool)) return L3D_ERR_UNSUPPORTED;                  // the epilogue writes fp32 rows OR planes / pooled maxima
    return cf_launch(x_planes, w_planes, scale, shift, shift_bstride, B, Cin, Cout, N, relu, y, out_img, obs, ypool, pool,
                     (unsigned *)amax_out, amax_cdiv, (hipStream_t)stream, (flags & 1) != 0, residual, (flags & 2) != 0, (flags & 4) != 0);
}

// ---------------------------------------------------------------------------------------------
// First layer of a per-point MLP (Cin <= 8: xyz, or xyz + a few features) written straight as an activation image:
// pcn.py:26-33 / pointnet.py:42 conv1 (3 -> 64/128).  24 FMAs per 8 outputs on the VALU; one thread per (row, octet) cell, rows
// fastest so that the plane writes are 1 KB per wave.  The plane scale comes from a bound computed by every workgroup from the
// weights, max_r(|shift_r| + xmax sum_c |w_rc|), xmax = a device float >= max|x| (the host passes x.abs().max()).
// ---------------------------------------------------------------------------------------------
#define CFN_MAXCIN 8
__global__ __launch_bounds__(256) void cf_first_layer_kernel(const float *__restrict__ x, int channel_last, const float *__restrict__ w,
                                                             const float *__restrict__ shift, const float *__restrict__ xmax, int Cin,
                                                             int Cout, int Npts, long R, int relu, uint4 *__restrict__ ph,
                                                             uint4 *__restrict__ pm, float *__restrict__ oinv, int *__restrict__ range_flag)
{
    __shared__ float red[4];
    const int t = threadIdx.x;
    const float xm = *xmax;
    float bnd = 0.f;
    for (int r = t; r < Cout; r += 256) {
        float s = 0.f;
        for (int c = 0; c < Cin; c++) s += fabsf(w[r * Cin + c]);
        bnd = fmaxf(bnd, fmaf(s, xm, shift ? fabsf(shift[r]) : 0.f));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) bnd = fmaxf(bnd, __shfl_xor(bnd, d, 64));
    if ((t & 63) == 0) red[t >> 6] = bnd;
    __syncthreads();
    bnd = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * 1.000001f;
    int e = 0;
    if (bnd > 0.f && bnd < 3.0e38f) (void)frexpf(bnd, &e);
    const float up = ldexpf(1.f, 12 - e);
    if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0) *oinv = ldexpf(1.f, e - 12);
    const long row = (long)blockIdx.x * 256 + t;
    const int o = blockIdx.y;
    if (row >= R) return;
    float xv[CFN_MAXCIN];
#pragma unroll
    for (int c = 0; c < CFN_MAXCIN; c++)
        xv[c] = c < Cin ? (channel_last ? x[(size_t)row * Cin + c] : x[((size_t)(row / Npts) * Cin + c) * Npts + row % Npts]) : 0.f;
    float v[8], big = 0.f;
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int co = o * 8 + u;
        float s = 0.f;
        if (co < Cout) {
#pragma unroll
            for (int c = 0; c < CFN_MAXCIN; c++)
                if (c < Cin) s = fmaf(w[co * Cin + c], xv[c], s);
            s += shift ? shift[co] : 0.f;
            if (relu) s = l3d_act(s, relu);
        }
        v[u] = s;
        big = fmaxf(big, fabsf(s * up));
    }
    uint32_t h[4], m[4];
#pragma unroll
    for (int u = 0; u < 4; u++) af_split_x(v[2 * u], v[2 * u + 1], up, h[u], m[u]);
    ph[(size_t)o * R + row] = make_uint4(h[0], h[1], h[2], h[3]);
    pm[(size_t)o * R + row] = make_uint4(m[0], m[1], m[2], m[3]);
    if (!(big <= 60000.f) && range_flag) *(volatile int *)range_flag = 1;      // xmax was not a bound (or inf / NaN)
}

// x [B][N][Cin] (channel_last) or [B][Cin][N]; w [Cout][Cin]; shift [Cout] or NULL; xmax: device float >= max|x|;
// out_img: l3d_f16_act_bytes(B N, Cout) bytes.  Cin <= 8.
extern "C" int l3d_first_layer_f16_planes(const float *x, int channel_last, const float *w, const float *shift, const float *xmax,
                                          int B, int Cin, int Cout, int N, int relu, void *out_img, int *range_flag,
                                          l3d_stream_t stream)
{
    L3D_REQUIRE(x && w && xmax && out_img && B > 0 && Cin > 0 && Cout > 0 && N > 0);
    if (Cin > CFN_MAXCIN || (((size_t)out_img) & 15)) return L3D_ERR_UNSUPPORTED;
    const long R = (long)B * N;
    const size_t pb = l3d_f16_plane_bytes(R, Cout);
    unsigned char *d = (unsigned char *)out_img;
    dim3 grid((unsigned)l3d_divup(R, 256), (unsigned)((Cout + 7) / 8));
    hipLaunchKernelGGL(cf_first_layer_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, channel_last, w, shift, xmax, Cin, Cout, N, R,
                       relu, (uint4 *)d, (uint4 *)(d + pb), (float *)(d + 2 * pb), range_flag);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// The factored first layer of a grouped MLP (grouping.hip, l3d_group_first_layer: act(U[idx] + V + Wx (xyz[idx] - centre)))
// written straight as the activation image of the next f16x2 layer -- models/flownet3d.py:125-242's 128-channel stacks then run
// conv2 / conv3 on the fp16 matrix cores (narrow tile) with no fp32 activation in between.  A workgroup takes 64 rows (s, k) of
// one cloud and all C1/8 octets: pass 1 reads with consecutive threads on consecutive octets of one gathered row (C1 * 4
// contiguous bytes), computes, splits and parks the 16-byte cells in LDS; pass 2 writes them with consecutive threads on
// consecutive rows of one octet (the planes' own order, 1 KB per wave) -- cf_split_x_cl_kernel's pattern.
// The plane scale comes from *bound >= max|output| (the host adds max|U| + max|V| + max_r sum_d |wx_rd| * max|coordinate difference|).
// ---------------------------------------------------------------------------------------------
#define GFP_ROWS 64
#define GFP_STRIDE (GFP_ROWS + 1)
template <int NO /* octets = C1 / 8: 8, 16 or 32 */>
__global__ __launch_bounds__(256) void group_first_layer_planes_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                                       const float *__restrict__ shift, const float *__restrict__ wx,
                                                                       const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                       const int32_t *__restrict__ idx, int N, int S, int K, int act,
                                                                       const float *__restrict__ bound, uint4 *__restrict__ ph,
                                                                       uint4 *__restrict__ pm, float *__restrict__ oinv,
                                                                       int *__restrict__ range_flag, const float *__restrict__ maxpart,
                                                                       float wxr, float shmax)
{
    __shared__ uint4 lh[NO * GFP_STRIDE], lm[NO * GFP_STRIDE];
    __shared__ float mx4[4];
    constexpr int C1 = NO * 8, RPP = 256 / NO;                   // rows per pass
    const int t = threadIdx.x, b = blockIdx.y;
    const long SK = (long)S * K, e0 = (long)blockIdx.x * GFP_ROWS, rows = (long)gridDim.y * SK;
    int ex = 0;
    {
        float bd;
        if (maxpart) {
            // the bound from l3d_absmax4_partials' [4][64] block maxima (max|U|, max|V|, max|xyz|, max|new_xyz|): wave j reduces
            // tensor j; bound = max|U| + (max|V| or max|shift|) + (max_r sum_d |wx_rd|) (max|xyz| + max|new_xyz|)
            float m = maxpart[t];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
            if ((t & 63) == 0) mx4[t >> 6] = m;
            __syncthreads();
            bd = mx4[0] + (V ? mx4[1] : shmax) + wxr * (mx4[2] + mx4[3]);
        } else {
            bd = *bound;
        }
        bd *= 1.000001f;
        if (bd > 0.f && bd < 3.0e38f) (void)frexpf(bd, &ex);
    }
    const float up = ldexpf(1.f, 12 - ex);
    if (blockIdx.x == 0 && b == 0 && t == 0) *oinv = ldexpf(1.f, ex - 12);
    const int oc = t % NO, r0 = t / NO;
    float w0[8], w1[8], w2[8], sh[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const float *w = wx + (size_t)(8 * oc + u) * 3;
        w0[u] = w[0]; w1[u] = w[1]; w2[u] = w[2];
        sh[u] = shift ? shift[8 * oc + u] : 0.f;
    }
    float big = 0.f;
#pragma unroll 2
    for (int pass = 0; pass < GFP_ROWS / RPP; pass++) {
        const int r = pass * RPP + r0;
        const long e = e0 + r;
        uint4 hc = make_uint4(0, 0, 0, 0), mc = hc;
        if (e < SK) {
            const int s_ = (int)(e / K), j = idx[(size_t)b * SK + e];
            const float *p = xyz + ((size_t)b * N + j) * 3, *q = new_xyz + ((size_t)b * S + s_) * 3;
            const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
            const f32x4 *up4 = (const f32x4 *)(U + ((size_t)b * N + j) * C1 + 8 * oc);
            f32x4 ua = up4[0], ub = up4[1];
            if (V) {
                const f32x4 *vp4 = (const f32x4 *)(V + ((size_t)b * S + s_) * C1 + 8 * oc);
                ua += vp4[0]; ub += vp4[1];
            }
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                float x = (u < 4 ? ua[u] : ub[u - 4]) + sh[u];
                x = fmaf(w2[u], dz, fmaf(w1[u], dy, fmaf(w0[u], dx, x)));
                if (act) x = l3d_act(x, act);
                v[u] = x;
                big = fmaxf(big, fabsf(x * up));
            }
            af_split_x(v[0], v[1], up, hc.x, mc.x);
            af_split_x(v[2], v[3], up, hc.y, mc.y);
            af_split_x(v[4], v[5], up, hc.z, mc.z);
            af_split_x(v[6], v[7], up, hc.w, mc.w);
        }
        lh[oc * GFP_STRIDE + r] = hc;
        lm[oc * GFP_STRIDE + r] = mc;
    }
    __syncthreads();
    {
        const int r = t & 63;
        const long e = e0 + r;
        if (e < SK) {
            const size_t row = (size_t)b * SK + e;
#pragma unroll
            for (int pass = 0; pass < NO / 4; pass++) {
                const int o = pass * 4 + (t >> 6);
                ph[(size_t)o * rows + row] = lh[o * GFP_STRIDE + r];
                pm[(size_t)o * rows + row] = lm[o * GFP_STRIDE + r];
            }
        }
    }
    if (!(big <= 60000.f) && range_flag) *(volatile int *)range_flag = 1;        // *bound was not a bound (or inf / NaN)
}

// as l3d_group_first_layer (grouping.hip), output = activation image (l3d_f16_act_bytes(B S K, C1) bytes) instead of fp32 rows;
// bound: device float >= max|output|.  C1 = 64, 128 or 256.
static int gfp_launch(const float *U, const float *V, const float *shift, const float *wx, const float *xyz, const float *new_xyz,
                      const int32_t *idx, int B, int N, int S, int K, int C1, int relu, const float *bound, const float *maxpart,
                      float wxr, float shmax, void *out_img, int *range_flag, hipStream_t st)
{
    if (B > 65535 || (C1 != 64 && C1 != 128 && C1 != 256) || (((size_t)U | (size_t)V | (size_t)out_img) & 15)) return L3D_ERR_UNSUPPORTED;
    const long R = (long)B * S * K;
    const size_t pb = l3d_f16_plane_bytes(R, C1);
    unsigned char *d = (unsigned char *)out_img;
    dim3 grid((unsigned)l3d_divup((long)S * K, GFP_ROWS), B), block(256);
#define GFP_ARGS U, V, shift, wx, xyz, new_xyz, idx, N, S, K, relu, bound, (uint4 *)d, (uint4 *)(d + pb), (float *)(d + 2 * pb), range_flag, \
                 maxpart, wxr, shmax
    if (C1 == 64)       hipLaunchKernelGGL(group_first_layer_planes_kernel<8>, grid, block, 0, st, GFP_ARGS);
    else if (C1 == 128) hipLaunchKernelGGL(group_first_layer_planes_kernel<16>, grid, block, 0, st, GFP_ARGS);
    else                hipLaunchKernelGGL(group_first_layer_planes_kernel<32>, grid, block, 0, st, GFP_ARGS);
#undef GFP_ARGS
    return l3d_check_launch();
}

// Block maxima of up to four fp32 tensors in one launch: out[j][blk] = max |p_j| over block blk's stride of tensor j (64 blocks per
// tensor; zeros for an absent tensor).  No atomics, no pre-zeroing: the consumer reduces the 4 x 64 values itself.
__global__ __launch_bounds__(256) void absmax4_partials_kernel(const float *__restrict__ p0, size_t n0, const float *__restrict__ p1,
                                                               size_t n1, const float *__restrict__ p2, size_t n2,
                                                               const float *__restrict__ p3, size_t n3, float *__restrict__ out)
{
    __shared__ float red[4];
    const int j = blockIdx.y;
    const float *p = j == 0 ? p0 : (j == 1 ? p1 : (j == 2 ? p2 : p3));
    const size_t n = j == 0 ? n0 : (j == 1 ? n1 : (j == 2 ? n2 : n3));
    float m = 0.f;
    if (p) {
        const size_t n4 = (((size_t)p & 15) == 0) ? n >> 2 : 0;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            const float4 v = ((const float4 *)p)[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(p[i]));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[j * 64 + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

extern "C" int l3d_absmax4_partials(const float *p0, size_t n0, const float *p1, size_t n1, const float *p2, size_t n2,
                                    const float *p3, size_t n3, float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(out);
    hipLaunchKernelGGL(absmax4_partials_kernel, dim3(64, 4), dim3(256), 0, (hipStream_t)stream, p0, n0, p1, n1, p2, n2, p3, n3, out);
    return l3d_check_launch();
}

// l3d_group_first_layer_planes with the bound formed inside the kernel: maxpart = l3d_absmax4_partials(U, V, xyz, new_xyz)'s 256
// floats, wxr = max_r sum_d |wx_rd| and shmax = max|shift| (functions of the layer's parameters) -- one launch in front of the layer
// instead of a dozen reductions and scalar operations (NaN / inf in an input -> the range flag, as before).
extern "C" int l3d_group_first_layer_planes_auto(const float *U, const float *V, const float *shift, const float *wx, const float *xyz,
                                                 const float *new_xyz, const int32_t *idx, int B, int N, int S, int K, int C1, int relu,
                                                 const float *maxpart, float wxr, float shmax, void *out_img, int *range_flag,
                                                 l3d_stream_t stream)
{
    L3D_REQUIRE(U && wx && xyz && new_xyz && idx && maxpart && out_img && B > 0 && N > 0 && S > 0 && K > 0 && C1 > 0);
    return gfp_launch(U, V, shift, wx, xyz, new_xyz, idx, B, N, S, K, C1, relu, nullptr, maxpart, wxr, shmax, out_img, range_flag,
                      (hipStream_t)stream);
}
