// bmm.hip -- strided batched fp32 GEMM on the fp32 matrix cores, for the TRAINING path of the pointer network and the SVD head
// (SURVEY.md 8 row f3): C[i][j] = act(alpha * A[i][j] B[i][j] + bias) (+ C[i][j]) over a two-level batch (cloud i, head j), every
// operand addressed through four element strides {batch1, batch2, row, column} -- so q k^T, p v, p^T dO, dS^T q, x W^T, dy W and
// dy^T x (reference utils/transformer.py:127-132 `attention`, :183-189 the nn.Linear layers; utils/svd.py:27-31 the SVD head's
// scores; utils/model_common_utils.py:19-38 square_distance -- and what autograd derives for them) are all the same kernel reading
// the tensors where they lie: no transposed copies.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- an exact fp32 fma chain per output element, ascending k (split-K: ascending k within a
// part, parts summed in ascending order by the reduction kernel): deterministic, and the fp32 result torch's own matmul is compared
// against in tests/test_gpu_grad_routes.py.  Peak 157.3 TFLOP/s (MI355X_MICROARCH.md); this is not an inference kernel -- the
// inference path's GEMMs are the f16x2 kernels of conv_f16.hip / attention_f16b.hip.
//
// Workgroup: 256 threads, tile 128 x 128, four waves of 64 x 64 (2 x 2 MFMA tiles, 64 accumulator registers), K in chunks of 16
// through LDS ([k][m] and [k][n], pitch 160 floats, two buffers); the next chunk's values are in flight in registers while the
// current one is multiplied.  A thread's 8 values run along whichever axis of the operand is contiguous (two 16-byte loads when the
// host says the operand is 16-byte aligned along it, scalars otherwise and at the edges).
#include "common.h"
#include "split_bf16.h"          // f32x16
#include <type_traits>

#define BM_T 128
#define BM_K 16
#define BM_P (BM_T + 32)          // pitch 160: the two k rows a wave reads per MFMA operand fall on disjoint bank halves
#ifndef BM_OCC
#define BM_OCC 4                   // waves per SIMD the register allocation is bounded for: <= 128 registers, so that FOUR workgroups share a
                                   // CU -- a Linear layer over 32768 rows x 512 channels is 1024 tiles = one round of the chip (at the three the
                                   // 134-138 registers of a bound of 2 allowed, a second round ran on a third of the slots: 210 -> 183 us)
#endif

struct BmOperand {
    const float *p;
    long s1, s2, sr, sc;         // element strides: batch level 1, batch level 2, row, column
    int unused;
};

// V values (8 or 4) of a [16 V rows x 16 k] operand tile for thread t.  MODE is a property of the operand, fixed for the launch (a
// template parameter: with the three forms behind run-time tests every load sat in an exec-masked region with a wait at its end,
// and nothing was in flight under the MFMAs -- LABLOG R4.8):
//   1  16-byte loads along k: row t / (16 / V), k (t % (16 / V)) V .. + V - 1; needs sc == 1, alignment, K % 16 == 0
//   2  16-byte loads along the rows: k = t >> 4, rows (t & 15) V .. + V - 1;   needs sr == 1, alignment, K % 16 == 0, R % 8 == 0
//   0  anything else: scalar loads, rows fastest when sr == 1
// In the vector forms nothing is conditional: a row index beyond R is clamped to a valid one -- what is loaded there only reaches
// output rows / columns beyond M / N, which are not stored.  The scalar form multiplies by a 0 / 1 mask instead of selecting (a
// select on a loaded value becomes a branch with a wait behind it).  `sr` / `sc` = the stride along the operand's rows / along k
// (for B, which is given as [K x N], the caller passes them swapped).
template <int V>
struct BmFrag { float v[V]; };

template <int V, int MODE>
__device__ __forceinline__ BmFrag<V> bm_fetch(const float *__restrict__ base, long sr, long sc, int r0, int k0, int R, int K, int t)
{
    BmFrag<V> f;
    if constexpr (MODE == 2) {
        const int k = k0 + (t >> 4), r = min(r0 + (t & 15) * V, R - V);
#pragma unroll
        for (int j = 0; j < V; j += 4) {
            const float4 a = *(const float4 *)(base + (long)k * sc + r + j);
            f.v[j] = a.x; f.v[j + 1] = a.y; f.v[j + 2] = a.z; f.v[j + 3] = a.w;
        }
    } else if constexpr (MODE == 1) {
        constexpr int TPR = 16 / V;                     // threads per row
        const int r = min(r0 + t / TPR, R - 1), k = k0 + (t % TPR) * V;
#pragma unroll
        for (int j = 0; j < V; j += 4) {
            const float4 a = *(const float4 *)(base + (long)r * sr + k + j);
            f.v[j] = a.x; f.v[j + 1] = a.y; f.v[j + 2] = a.z; f.v[j + 3] = a.w;
        }
    } else if (sr == 1) {
        const int k = k0 + (t >> 4), r = r0 + (t & 15) * V;
        const bool kin = k < K;
        const long kc = min(k, K - 1);
        // unconditional loads from clamped indices, then a SELECT on the loaded value (a 0 / 1 factor would turn an Inf / NaN
        // sitting at the clamped index into NaN inside a tail that torch.matmul does not read)
        float ld[V];
#pragma unroll
        for (int i = 0; i < V; i++) ld[i] = base[kc * sc + min(r + i, R - 1)];
#pragma unroll
        for (int i = 0; i < V; i++) f.v[i] = (kin && r + i < R) ? ld[i] : 0.f;
    } else {
        constexpr int TPR = 16 / V;
        const int r = r0 + t / TPR, k = k0 + (t % TPR) * V;
        const bool rin = r < R;
        const long rc = min(r, R - 1);
        float ld[V];
#pragma unroll
        for (int i = 0; i < V; i++) ld[i] = base[rc * sr + (long)min(k + i, K - 1) * sc];
#pragma unroll
        for (int i = 0; i < V; i++) f.v[i] = (rin && k + i < K) ? ld[i] : 0.f;
    }
    return f;
}

// ... and where they go in the [k][row] LDS tile
template <int V, int MODE>
__device__ __forceinline__ void bm_stash(float (*__restrict__ tile)[BM_P], const BmFrag<V> &f, long sr, int t)
{
    if (MODE == 2 || (MODE == 0 && sr == 1)) {
        const int k = t >> 4, r = (t & 15) * V;
#pragma unroll
        for (int j = 0; j < V; j += 4) *(float4 *)&tile[k][r + j] = make_float4(f.v[j], f.v[j + 1], f.v[j + 2], f.v[j + 3]);
    } else {
        constexpr int TPR = 16 / V;
        const int r = t / TPR, k = (t % TPR) * V;
#pragma unroll
        for (int i = 0; i < V; i++) tile[k + i][r] = f.v[i];
    }
}

// grid: x = column tiles, y = row tiles, z = (batch1 * nb2 + batch2) * parts + part.  parts > 1: every part writes the raw sums of
// its K range into ws [z][M][N]; bm_reduce_kernel finishes.
// YT = 32-column MFMA tiles per wave: 2 -> workgroup tile 128 x 128 (64 accumulators), 1 -> 128 x 64 (32) for products whose
// 128 x 128 tiling would leave CUs with a single workgroup (a Linear layer over 8192 rows x 512 channels is 256 tiles).
// LDS is double-buffered: one barrier per chunk (chunk i + 1 is stashed into the other buffer behind chunk i's MFMAs; every wave
// has passed the barrier in front of chunk i, so nobody still reads that buffer).
// AM / BM = the fetch form of the two operands (bm_fetch)
template <int YT, int AM, int BM>
__global__ __launch_bounds__(256, (YT == 2 && (AM == 0 || BM == 0)) ? 2 : BM_OCC) void bmm_f32_kernel(BmOperand A, BmOperand B, float *__restrict__ C, long c1, long c2, long cr, long cc,
                                                      int nb2, int M, int N, int K, float alpha, int flags,
                                                      const float *__restrict__ bias, int parts, float *__restrict__ ws)
{
    constexpr int TN = 64 * YT, VB = 4 * YT;
    __shared__ __attribute__((aligned(16))) float As[2][BM_K][BM_P];
    __shared__ __attribute__((aligned(16))) float Bs[2][BM_K][BM_P];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int part = blockIdx.z % parts, bz = blockIdx.z / parts;
    const int i1 = bz / nb2, i2 = bz % nb2;
    // Tile order.  Workgroups are dispatched x fastest and workgroup L runs on XCD L % 8, each with its own L2: left alone, the column
    // tiles of one row tile land on DIFFERENT XCDs and every one of them fetches the row tile's A rows from HBM / the memory-side cache
    // again (133 MB fetched for 67 MB of x at 32768 x 512 x 512, profiles/round6_pmc_bmm12.txt).  Remapped so that a row tile's column
    // tiles are consecutive slots of ONE XCD (conv_f16.hip's order).
    int bx = blockIdx.x, by = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int L = blockIdx.y * gridDim.x + blockIdx.x, xcd = L & 7, slot = L >> 3;
        bx = slot % gridDim.x;
        by = (slot / gridDim.x) * 8 + xcd;
    }
    const int m0 = by * BM_T, n0 = bx * TN;
    const float *__restrict__ a = A.p + (long)i1 * A.s1 + (long)i2 * A.s2;
    const float *__restrict__ b = B.p + (long)i1 * B.s1 + (long)i2 * B.s2;
    // this part's K range: chunks of 16, split evenly
    const int nchunk = (K + BM_K - 1) / BM_K;
    const int cpp = (nchunk + parts - 1) / parts;
    const int kbeg = part * cpp * BM_K, kend = min(K, kbeg + cpp * BM_K);

    f32x16 acc[2][YT];
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < YT; y++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[x][y][r] = 0.f;

    // B is [K x N]: as a [N rows x K] operand its row stride is sc and its k stride sr.
    // Chunk c + 2 is requested while chunk c is multiplied and chunk c + 1 waits in registers: a chunk's MFMAs take ~0.85 us per
    // wave, less than a trip to HBM under load -- with one chunk of look-ahead the stash at the end of every chunk waited for its
    // loads (51 % matrix-pipe utilisation at two waves per SIMD, LABLOG R4.8).  Chunk c lives in register set c & 1.
    BmFrag<8> fa[2];
    BmFrag<VB> fb[2];
    if (kbeg < kend) {
        fa[0] = bm_fetch<8, AM>(a, A.sr, A.sc, m0, kbeg, M, kend, t);
        fb[0] = bm_fetch<VB, BM>(b, B.sc, B.sr, n0, kbeg, N, kend, t);
        if (kbeg + BM_K < kend) {
            fa[1] = bm_fetch<8, AM>(a, A.sr, A.sc, m0, kbeg + BM_K, M, kend, t);
            fb[1] = bm_fetch<VB, BM>(b, B.sc, B.sr, n0, kbeg + BM_K, N, kend, t);
        }
        bm_stash<8, AM>(As[0], fa[0], A.sr, t);
        bm_stash<VB, BM>(Bs[0], fb[0], B.sc, t);
    }
    const int kl = lane >> 5, rl = lane & 31;
    auto chunk = [&](int k0, auto parity) {
        constexpr int P = decltype(parity)::value;        // k0's chunk index & 1; LDS buffer P holds it
        __syncthreads();                                  // chunk k0 is in buffer P; buffer P ^ 1 is free
        if (k0 + 2 * BM_K < kend) {                       // register set P is free too (its chunk was stashed): chunk + 2 goes there
            fa[P] = bm_fetch<8, AM>(a, A.sr, A.sc, m0, k0 + 2 * BM_K, M, kend, t);
            fb[P] = bm_fetch<VB, BM>(b, B.sc, B.sr, n0, k0 + 2 * BM_K, N, kend, t);
        }
        // the fragments of k-step s + 1 are read before the MFMAs of k-step s are issued (left to itself the compiler read a step's four
        // values, waited for them, issued the step's MFMAs and only then read the next: every step ended in an exposed LDS round trip)
        float av[2][2], bv[2][YT];
#pragma unroll
        for (int x = 0; x < 2; x++) av[0][x] = As[P][kl][wm * 64 + 32 * x + rl];
#pragma unroll
        for (int y = 0; y < YT; y++) bv[0][y] = Bs[P][kl][wn * 32 * YT + 32 * y + rl];
#pragma unroll
        for (int kk = 0; kk < BM_K; kk += 2) {
            const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
            if (kk + 2 < BM_K) {
#pragma unroll
                for (int x = 0; x < 2; x++) av[nxt][x] = As[P][kk + 2 + kl][wm * 64 + 32 * x + rl];
#pragma unroll
                for (int y = 0; y < YT; y++) bv[nxt][y] = Bs[P][kk + 2 + kl][wn * 32 * YT + 32 * y + rl];
            }
#pragma unroll
            for (int x = 0; x < 2; x++)
#pragma unroll
                for (int y = 0; y < YT; y++) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][x], bv[cur][y], acc[x][y], 0, 0, 0);
        }
        // ... and the order is pinned (the scheduler folds the two register sets back into one otherwise): two steps' reads, then
        // [a step's MFMAs, the reads of the step after next]; a step is two LDS instructions (ds_read2_b32 pairs) and 2 YT MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int i = 0; i < BM_K / 2 - 2; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * YT, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * YT, 0);
        if (k0 + BM_K < kend) {                           // chunk + 1 (register set P ^ 1, requested a chunk ago) into the other buffer
            bm_stash<8, AM>(As[P ^ 1], fa[P ^ 1], A.sr, t);
            bm_stash<VB, BM>(Bs[P ^ 1], fb[P ^ 1], B.sc, t);
        }
    };
    for (int k0 = kbeg; k0 < kend; k0 += 2 * BM_K) {
        chunk(k0, std::integral_constant<int, 0>{});
        if (k0 + BM_K < kend) chunk(k0 + BM_K, std::integral_constant<int, 1>{});
    }

    // D[m = 32x + (r&3) + 8(r>>2) + 4(lane>>5)][n = 32y + (lane&31)]
    float *__restrict__ c = parts > 1 ? ws + (size_t)blockIdx.z * M * N : C + (long)i1 * c1 + (long)i2 * c2;
    const long sr = parts > 1 ? N : cr, sc = parts > 1 ? 1 : cc;
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m0 + wm * 64 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= M) continue;
#pragma unroll
            for (int y = 0; y < YT; y++) {
                const int n = n0 + wn * 32 * YT + 32 * y + (lane & 31);
                if (n >= N) continue;
                float v = acc[x][y][r];
                if (parts == 1) {
                    v = alpha * v;
                    if (flags & 4) v = v + bias[n];
                    if (flags & 8) v = v + bias[m];
                    if (flags & 2) v = fmaxf(v, 0.f);
                    if (flags & 1) v = c[(long)m * sr + (long)n * sc] + v;
                }
                c[(long)m * sr + (long)n * sc] = v;
            }
        }
}

__global__ __launch_bounds__(256) void bm_reduce_kernel(const float *__restrict__ ws, int parts, float *__restrict__ C, long c1, long c2,
                                                        long cr, long cc, int nb2, int M, int N, float alpha, int flags,
                                                        const float *__restrict__ bias)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x, per = (long)M * N;
    const int bz = blockIdx.y;
    if (e >= per) return;
    const int m = (int)(e / N), n = (int)(e % N);
    float s = 0.f;
    for (int p = 0; p < parts; p++) s += ws[((size_t)bz * parts + p) * per + e];
    float v = alpha * s;
    if (flags & 4) v = v + bias[n];
    if (flags & 8) v = v + bias[m];
    if (flags & 2) v = fmaxf(v, 0.f);
    float *c = C + (long)(bz / nb2) * c1 + (long)(bz % nb2) * c2 + (long)m * cr + (long)n * cc;
    if (flags & 1) v = *c + v;
    *c = v;
}

// the fetch form of an operand seen as [R rows x K] with strides s = {batch1, batch2, row, k} (bm_fetch)
static int bm_mode(const float *p, const long s[4], int R, int K)
{
    const bool al = (((size_t)p) & 15) == 0 && s[0] % 4 == 0 && s[1] % 4 == 0 && K % BM_K == 0;
    if (s[3] == 1 && al && s[2] % 4 == 0) return 1;
    if (s[2] == 1 && al && s[3] % 4 == 0 && R % 8 == 0) return 2;
    return 0;
}

// C[i][j] = act(alpha A[i][j] B[i][j] + bias) (+ C[i][j]),  A [M x K], B [K x N], C [M x N], i < nb1, j < nb2;
// *_strides = {batch1, batch2, row, column} in elements (a transposed operand is its strides swapped; a broadcast one has batch
// strides 0).  flags: 1 accumulate into C, 2 ReLU, 4 bias[n], 8 bias[m].  parts > 1: split K into `parts` ranges whose partial
// products go through `workspace` (nb1 nb2 parts M N floats) and are summed in ascending order -- for products with few output
// tiles and a long K (weight gradients: K = every row of the batch).
extern "C" int l3d_bmm_f32(const float *A, const long *a_strides, const float *B, const long *b_strides, float *C,
                           const long *c_strides, int nb1, int nb2, int M, int N, int K, float alpha, int flags,
                           const float *bias, int parts, float *workspace, l3d_stream_t stream)
{
    L3D_REQUIRE(A && B && C && a_strides && b_strides && c_strides && nb1 > 0 && nb2 > 0 && M > 0 && N > 0 && K > 0 &&
                (flags & ~15) == 0 && (!(flags & 12) || bias) && parts >= 1 && (parts == 1 || workspace));
    const long nz = (long)nb1 * nb2 * parts;
    if (nz > 65535 || l3d_divup(M, BM_T) > 65535) return L3D_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    BmOperand a{A, a_strides[0], a_strides[1], a_strides[2], a_strides[3], 0};
    BmOperand b{B, b_strides[0], b_strides[1], b_strides[2], b_strides[3], 0};
    // B as a [N rows x K] operand: "along k" is its row stride, "along rows" its column stride
    const long bsw[4] = {b_strides[0], b_strides[1], b_strides[3], b_strides[2]};
    const int am = bm_mode(A, a_strides, M, K), bm = bm_mode(B, bsw, N, K);
    // 128 x 128 tiles when they fill the chip twice over, 128 x 64 otherwise
    const long wide = (long)l3d_divup(N, 128) * l3d_divup(M, BM_T) * nz;
    const bool yt2 = wide >= 512 && N > 64;
    dim3 grid((unsigned)l3d_divup(N, yt2 ? 128 : 64), (unsigned)l3d_divup(M, BM_T), (unsigned)nz);
#define BM_LAUNCH(YT, AM, BMD)                                                                                                             \
    hipLaunchKernelGGL((bmm_f32_kernel<YT, AM, BMD>), grid, dim3(256), 0, st, a, b, C, c_strides[0], c_strides[1], c_strides[2], c_strides[3], \
                       nb2, M, N, K, alpha, flags, bias, parts, workspace)
#define BM_PICK_B(YT, AM)                                                                                                                   \
    do { if (bm == 1) BM_LAUNCH(YT, AM, 1); else if (bm == 2) BM_LAUNCH(YT, AM, 2); else BM_LAUNCH(YT, AM, 0); } while (0)
#define BM_PICK_A(YT)                                                                                                                       \
    do { if (am == 1) BM_PICK_B(YT, 1); else if (am == 2) BM_PICK_B(YT, 2); else BM_PICK_B(YT, 0); } while (0)
    if (yt2) BM_PICK_A(2); else BM_PICK_A(1);
#undef BM_PICK_A
#undef BM_PICK_B
#undef BM_LAUNCH
    if (parts > 1) {
        if (l3d_check_launch() != 0) return L3D_ERR_LAUNCH;
        dim3 rg((unsigned)l3d_divup((long)M * N, 256), (unsigned)(nb1 * nb2));
        hipLaunchKernelGGL(bm_reduce_kernel, rg, dim3(256), 0, st, (const float *)workspace, parts, C, c_strides[0], c_strides[1],
                           c_strides[2], c_strides[3], nb2, M, N, alpha, flags, bias);
    }
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// softmax over the last axis of [rows][cols] (reference utils/transformer.py:131, utils/svd.py:29), in place or not, and its
// backward ds = p (dp - sum_j p_j dp_j): one workgroup per row, values in registers (cols <= 256 * SM_PER), fixed-order
// reductions (a wave's butterfly, then the four waves in order).
// ---------------------------------------------------------------------------------------------
#define SM_PER 32
__device__ __forceinline__ float sm_block_reduce(float v, bool is_max, float *red)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const float o = __shfl_xor(v, d, 64);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : ((red[0] + red[1]) + (red[2] + red[3]));
}

// NIT = 256-column chunks a thread holds (cols <= 256 NIT): a template parameter so that the loads are an unrolled batch with no
// condition around them (clamped column index; the values beyond `cols` are replaced after the loads have all been issued) -- with a
// run-time bound every load was followed by its own wait (tools/isa_audit.py)
template <int NIT>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float *__restrict__ x, long rows, int cols, float scale, float *__restrict__ y)
{
    __shared__ float red[4];
    const long row = blockIdx.x;
    const float *xr = x + row * cols;
    float v[NIT], big = -INFINITY;
#pragma unroll
    for (int i = 0; i < NIT; i++) v[i] = xr[min(i * 256 + (int)threadIdx.x, cols - 1)];
#pragma unroll
    for (int i = 0; i < NIT; i++) {
        v[i] = i * 256 + (int)threadIdx.x < cols ? v[i] * scale : -INFINITY;
        big = fmaxf(big, v[i]);
    }
    big = sm_block_reduce(big, true, red);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; i++) {
        v[i] = i * 256 + (int)threadIdx.x < cols ? expf(v[i] - big) : 0.f;
        sum += v[i];
    }
    sum = sm_block_reduce(sum, false, red);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < NIT; i++) {
        const int c = i * 256 + threadIdx.x;
        if (c < cols) y[row * cols + c] = v[i] * inv;
    }
}

template <int NIT>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float *__restrict__ p, const float *__restrict__ dp, long rows, int cols,
                                                               float scale, float *__restrict__ ds)
{
    __shared__ float red[4];
    const long row = blockIdx.x;
    float pv[NIT], gv[NIT], dot = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; i++) {
        const int c = min(i * 256 + (int)threadIdx.x, cols - 1);
        pv[i] = p[row * cols + c];
        gv[i] = dp[row * cols + c];
    }
#pragma unroll
    for (int i = 0; i < NIT; i++) {
        const float in = i * 256 + (int)threadIdx.x < cols ? 1.f : 0.f;
        pv[i] *= in;
        dot += pv[i] * gv[i];
    }
    dot = sm_block_reduce(dot, false, red);
#pragma unroll
    for (int i = 0; i < NIT; i++) {
        const int c = i * 256 + threadIdx.x;
        if (c < cols) ds[row * cols + c] = scale * (pv[i] * (gv[i] - dot));
    }
}

// forward (dp == NULL): y = softmax(scale * x) over the last axis; backward (dp given): y = scale * p (dp - sum p dp) with p = x.
// y may alias x (forward) or dp (backward).  cols <= 8192.
extern "C" int l3d_softmax_rows(const float *x, const float *dp, long rows, int cols, float scale, float *y, l3d_stream_t stream)
{
    L3D_REQUIRE(x && y && rows > 0 && cols > 0);
    if (cols > 256 * SM_PER || rows > 2147483647L) return L3D_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int nit = l3d_divup(cols, 256);
#define SM_GO(NIT)                                                                                                        \
    do {                                                                                                                  \
        if (dp) hipLaunchKernelGGL(softmax_rows_bwd_kernel<NIT>, dim3((unsigned)rows), dim3(256), 0, st, x, dp, rows, cols, scale, y); \
        else hipLaunchKernelGGL(softmax_rows_kernel<NIT>, dim3((unsigned)rows), dim3(256), 0, st, x, rows, cols, scale, y);            \
    } while (0)
    if (nit <= 1) SM_GO(1); else if (nit <= 2) SM_GO(2); else if (nit <= 4) SM_GO(4); else if (nit <= 8) SM_GO(8);
    else if (nit <= 16) SM_GO(16); else SM_GO(32);
#undef SM_GO
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Column sums of [rows][cols] (row stride given): the bias gradient of an nn.Linear over rows, db = 1^T g (what autograd derives for
// utils/transformer.py:183-189).  As a GEMM with M = 1 it cost 62 us for 32768 x 512 (one MFMA row in 128 used); this is a read at
// bandwidth.  Deterministic: a fixed summation tree -- rows in chunks of 128 (four groups of 32 in ascending order, then the groups in
// order), the chunks' sums in four ascending quarters, then the quarters in order.
// ---------------------------------------------------------------------------------------------
#define CS_ROWS 128
// one workgroup: 128 rows x 256 columns.  VEC: four row groups of 32 rows x 64 lanes of float4 (16 MB in flight over the chip at 32768 x 512),
// the groups' sums added in order through LDS; else one thread per column over the 128 rows (any stride / alignment).
template <bool VEC>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ x, long rows, int cols, long stride,
                                                             float *__restrict__ ws)
{
    const long r0 = (long)blockIdx.y * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
    if constexpr (VEC) {
        __shared__ float4 part[4][64];
        const int q = threadIdx.x & 63, grp = threadIdx.x >> 6, c = blockIdx.x * 256 + 4 * q;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < cols) {
            const long ra = r0 + 32 * grp, rb = min(r1, ra + 32);
            const float *p = x + ra * stride + c;
            long r = ra;
            for (; r + 8 <= rb; r += 8) {
                float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = *(const float4 *)(p + (long)i * stride);
#pragma unroll
                for (int i = 0; i < 8; i++) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
                p += 8 * stride;
            }
            for (; r < rb; r++, p += stride) {
                const float4 v = *(const float4 *)p;
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        part[grp][q] = s;
        __syncthreads();
        if (grp == 0 && c < cols) {
#pragma unroll
            for (int g = 1; g < 4; g++) { const float4 o = part[g][q]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
            *(float4 *)(ws + (size_t)blockIdx.y * cols + c) = s;
        }
    } else {
        const int c = blockIdx.x * 256 + threadIdx.x;
        if (c >= cols) return;
        const float *p = x + r0 * stride + c;
        float s = 0.f;
        long r = r0;
        for (; r + 8 <= r1; r += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = p[(long)i * stride];
#pragma unroll
            for (int i = 0; i < 8; i++) s += v[i];
            p += 8 * stride;
        }
        for (; r < r1; r++, p += stride) s += *p;
        ws[(size_t)blockIdx.y * cols + c] = s;
    }
}

// 64 columns per workgroup; four groups of threads each add a quarter of the chunks in ascending order (16 loads in flight), the four
// quarters are added in order
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float *__restrict__ ws, int chunks, int cols, float *__restrict__ out)
{
    __shared__ float part[4][64];
    const int q = threadIdx.x & 63, grp = threadIdx.x >> 6, c = blockIdx.x * 64 + q;
    const int per = (chunks + 3) / 4, i0 = grp * per, i1 = min(chunks, i0 + per);
    float s = 0.f;
    if (c < cols) {
        int i = i0;
        for (; i + 16 <= i1; i += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = ws[(size_t)(i + u) * cols + c];
#pragma unroll
            for (int u = 0; u < 16; u++) s += v[u];
        }
        for (; i < i1; i++) s += ws[(size_t)i * cols + c];
    }
    part[grp][q] = s;
    __syncthreads();
    if (grp == 0 && c < cols) out[c] = ((part[0][q] + part[1][q]) + part[2][q]) + part[3][q];
}

extern "C" size_t l3d_colsum_rows_workspace_bytes(long rows, int cols)
{
    return rows > 0 && cols > 0 ? (size_t)l3d_divup(rows, (long)CS_ROWS) * cols * sizeof(float) : 0;
}

// out[c] = sum_r x[r * row_stride + c], r < rows, c < cols
extern "C" int l3d_colsum_rows(const float *x, long rows, int cols, long row_stride, void *workspace, float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(x && out && workspace && rows > 0 && cols > 0 && row_stride >= cols);
    const long chunks = l3d_divup(rows, (long)CS_ROWS);
    if (chunks > 65535) return L3D_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)l3d_divup(cols, 256), (unsigned)chunks);
    if (cols % 4 == 0 && row_stride % 4 == 0 && (((size_t)x) & 15) == 0 && (((size_t)workspace) & 15) == 0)
        hipLaunchKernelGGL(colsum_partial_kernel<true>, grid, dim3(256), 0, st, x, rows, cols, row_stride, (float *)workspace);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<false>, grid, dim3(256), 0, st, x, rows, cols, row_stride, (float *)workspace);
    if (l3d_check_launch() != 0) return L3D_ERR_LAUNCH;
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)l3d_divup(cols, 64)), dim3(256), 0, st, (const float *)workspace, (int)chunks,
                       cols, out);
    return l3d_check_launch();
}
