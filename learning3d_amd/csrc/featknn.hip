// featknn.hip -- k nearest neighbours in FEATURE space: knn(x, k) of
// utils/model_common_utils.py:3-9 for x [B,C,N] with C != 3 (the dynamic graphs of PRNet's DGCNN, models/prnet.py:76-97;
// PointConv's knn_point, utils/pointconv_util.py:107-121; CurveNet's LPFA, utils/curvenet_util.py:260-291), without the [B,N,N]
// inner-product / distance matrices:
//     inner = -2 x^T x;  xx = sum_c x^2;  pd = -xx_j - inner_ij - xx_i;  idx = topk(pd, k)
//
// Round 6 rewrite (rounds 2-5: six bf16 products per fp32 product, a sorted (value, index) list of 2 K registers per lane, one wave per
// SIMD with every unit's load -> LDS -> barrier -> MFMA chain exposed: 125 us at B 32 / C 64 / N 1024 / k 20 against ~10 us of matrix
// work, 151 spilled registers at K = 64; that kernel is tools/experiments/featknn_v1.hip).  Now:
//
//   * the inner product is an f16x2 GEMM (split_f16.h: X = x 2^T per ROW, h = f16(X), m = f16(X - h); products m h' + h m' + h h' on
//     v_mfma_f32_32x32x16_f16, fp32 accumulate: THREE fp16 products per fp32 product, fp32-level error -- the same arithmetic as the
//     shared-MLP kernels).  The row scale is a power of two taken from the row's own largest magnitude by the thread that splits the
//     row: no pass for a tensor maximum, and 2^-T_i 2^-T_j comes back out exactly.
//   * one pre-pass (featknn_split_kernel) writes the two fp16 planes in MFMA operand order [plane][C/8][Np][8 fp16] (16-byte cells:
//     the 8 consecutive channels a lane of the MFMA consumes) and aux = (-|x_j|^2, 2^-T_j) per row; the SAME planes are both operands.
//   * featknn_kernel: a workgroup = 128 queries x the whole cloud, EIGHT waves (two per SIMD): wave = (query block of 32) x (key half):
//     the 64-key tiles 2t + half of every 128 keys.  Keys are the MFMA's rows and stream through a double-buffered LDS stage as
//     global_load_lds DMA pieces (no registers, no VALU) shared by the four query blocks; queries are the MFMA's COLUMNS: a lane holds
//     16 candidates of ONE query per 32 x 32 tile, and four lanes (two waves x two half-waves) serve a query.
//   * selection is knn_mfma.hip's: no sorted lists in registers.  Sweep 0 keeps per-lane GROUP maxima (one v_max3 per two values);
//     the T-th largest of a lane's 16 (K = 64: 32) group maxima, T = ceil(K / 4), minimised over the query's four lanes, is a bound thr that at
//     least K candidates reach -- and few more (~25 for K = 20 at N = 1024).  Sweep 1 recomputes the tiles (bit-identical) and appends
//     every candidate >= thr as a 64-bit key (order-preserving value bits << 32 | ~index) to the query's list (LDS counter, list in the
//     workspace).  Rank: four threads per query count the keys above each key; rank r < k writes idx[q][r].  Equal values: the lower
//     index ranks first, as l3d_knn_graph.
//   * a list that overflows (more than CAP candidates at or above the bound: clouds of many identical points) sends its query to an
//     exact wave-per-query selection at the end of the same kernel (plain fp32 dot products, k rounds of arg-max): slow, rare, and
//     what guarantees an answer for every input.
// Indices cannot be bit-pinned to the reference here (its sgemm's summation order is MKL's); the parity test bounds every returned
// neighbour by the exact k-th distance and compares with the reference's own op sequence on the CPU (> 99.9 % identical indices).
#include <type_traits>
#include "common.h"
#include "split_bf16.h"          // f32x16
#include "split_f16.h"

typedef _Float16 fk_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long fk_u64;
typedef __attribute__((address_space(3))) void *fk_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *fk_gbl_ptr_t;
// byte offset of an LDS object inside the workgroup's allocation (what a ds_* instruction in inline asm takes)
__device__ __forceinline__ unsigned fk_lds_off(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}

#ifdef FK_TIMELINE     // tools/fk_timeline.py: s_memrealtime (100 MHz) marks of every workgroup in the (unused, K <= 20) list area of the workspace
#define FKM(i) { if (threadIdx.x == 0) ((long long *)lists)[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define FKM(i)
#endif
#ifndef FK_ABL
#define FK_ABL 0      // timing ablations (tools/build_variant_lib.py; results are garbage): 1 no rank phase, 2 nothing reaches the bound, 4 one sweep, 8 no epilogue
#endif
#define FK_QT 128                          // queries per workgroup (4 column blocks of 32)
#define FK_STAGE 33792                     // 2 key halves x 2 planes x 8 octets x 64 rows x 16 B (128 keys x 64 channels) + 1 KB: the tile's aux
#define FK_AUXT 256                        // floats of aux per 128-key tile: -|x_row|^2 x 128 | 2^-T | max |x_row| | (unused)
#define FK_NSTAGE 2
#define FK_DUMP_OFF (FK_NSTAGE * FK_STAGE) // [8 waves][17 rows][64 lanes] floats (row 16 = scratch for lanes without a hit)
#define FK_DUMP_BYTES (8 * 17 * 64 * 4)
#define FK_CNT_OFF (FK_DUMP_OFF + FK_DUMP_BYTES)           // int qcnt[128]
#define FK_FLAG_OFF (FK_CNT_OFF + 128 * 4)                 // int flagged[1 .. 128], count at [129]
#define FK_LIST_OFF (FK_FLAG_OFF + 132 * 4)                // LDS lists (K <= 20): float lv[128][65], unsigned short li[128][66]
#define FK_LDS_CAP 64
#define FK_LVS 65                                          // row strides: odd in 4-byte words, or a wave's 16 - 32 queries meet in one bank
#define FK_LIS 66
#define FK_LDS_LL (FK_LIST_OFF + 128 * FK_LVS * 4 + 128 * FK_LIS * 2)
#define FK_MAXNP 16384                     // the exact fallback keeps one query's Np ranking values in the two stages

__host__ __device__ constexpr int fk_cap(int KC) { return KC <= 20 ? FK_LDS_CAP : (KC <= 32 ? 96 : 192); }   // list capacity per query

// x [B][C][N] fp32 -> planes [B][2][Cp/8][Np][8 fp16] (h | m of x 2^T, T per 128-ROW TILE from the tile's own largest magnitude; zero
// rows / channels beyond N / C) and aux [B][Np/128][256] = per tile: -sum_c x^2 of its 128 rows (-inf on the padding rows), then
// [128] = 2^-T, [129] = the tile's largest |x_row|.  A workgroup = one tile: thread (row t & 127, part t >> 7) takes a quarter of the
// row's channels / octets.  (A scale per tile rather than per tensor: no pass for a tensor maximum in front of the split; rather
// than per row: the epilogue then owes every value a multiplication.  A row far below its tile's maximum keeps the tile's absolute
// resolution, 2^-22 of |x_max| |x_i| in a ranking value -- the tolerance the parity test grants is 4e-6 of the largest |x|^2.)
__global__ __launch_bounds__(512) void featknn_split_kernel(const float *__restrict__ x, int C, int Cp, int N, int Np,
                                                            uint4 *__restrict__ planes, float *__restrict__ aux)
{
    __shared__ float red[4][128];
    __shared__ float tmax[8];
    const int t = threadIdx.x, r = t & 127, part = t >> 7;
    const int n = blockIdx.x * 128 + r, b = blockIdx.y;
    const float *xb = x + (size_t)b * C * N;
    const int noct = Cp / 8;
    uint4 *ph = planes + (size_t)b * 2 * noct * Np, *pm = ph + (size_t)noct * Np;
    // The thread's values once, with every load in flight (round 6: the maximum was a loop of one dependent load per channel -- 16 trips
    // to memory for C = 64, 12 us for 16 MB of traffic): up to four octets per thread (C <= 128) stay in registers for the split below;
    // wider inputs take the maximum eight loads at a time and read the values again (L2).
    constexpr int KEEP = 4;
    const bool keep = noct <= 4 * KEEP;
    float kv[KEEP][8];
    float big = 0.f;
    if (keep) {
#pragma unroll
        for (int i = 0; i < KEEP; i++) {
            const int o = part + 4 * i;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int c = min(o * 8 + u, C - 1);                    // clamped: no condition around the load
                kv[i][u] = xb[(size_t)c * N + min(n, N - 1)];
            }
        }
#pragma unroll
        for (int i = 0; i < KEEP; i++)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bool in = n < N && (part + 4 * i) * 8 + u < C;
                kv[i][u] = in ? kv[i][u] : 0.f;
                big = fmaxf(big, fabsf(kv[i][u]));
            }
    } else if (n < N) {
        for (int o = part; o < noct; o += 4) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = xb[(size_t)min(o * 8 + u, C - 1) * N + n];
#pragma unroll
            for (int u = 0; u < 8; u++) big = fmaxf(big, o * 8 + u < C ? fabsf(v[u]) : 0.f);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) big = fmaxf(big, __shfl_xor(big, d, 64));
    if ((t & 63) == 0) tmax[t >> 6] = big;
    __syncthreads();
    big = fmaxf(fmaxf(fmaxf(tmax[0], tmax[1]), fmaxf(tmax[2], tmax[3])), fmaxf(fmaxf(tmax[4], tmax[5]), fmaxf(tmax[6], tmax[7])));
    int e = 0;
    const bool fin = big > 0.f && big < INFINITY;
    if (fin) (void)frexpf(big, &e);                                 // big = f 2^e, f in [0.5, 1): big 2^(12 - e) in [2^11, 2^12)
    const int T = fin ? 12 - e : 0;
    const float up = ldexpf(1.0f, T);
    float s = 0.f;
    auto emit = [&](int o, const float (&v)[8]) {
#pragma unroll
        for (int u = 0; u < 8; u++) s = s + v[u] * v[u];           // x ** 2 then sum: no fused multiply-add
        uint4 h, m;
        af_split_x_unscaled(v[0], v[1], up, h.x, m.x);
        af_split_x_unscaled(v[2], v[3], up, h.y, m.y);
        af_split_x_unscaled(v[4], v[5], up, h.z, m.z);
        af_split_x_unscaled(v[6], v[7], up, h.w, m.w);
        ph[(size_t)o * Np + n] = h;
        pm[(size_t)o * Np + n] = m;
    };
    if (keep) {
#pragma unroll
        for (int i = 0; i < KEEP; i++)
            if (part + 4 * i < noct) emit(part + 4 * i, kv[i]);
    } else {
        for (int o = part; o < noct; o += 4) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = (n < N && o * 8 + u < C) ? xb[(size_t)(o * 8 + u) * N + n] : 0.f;
            emit(o, v);
        }
    }
    red[part][r] = s;
    __syncthreads();
    float *at = aux + ((size_t)b * (Np / 128) + blockIdx.x) * FK_AUXT;
    float xx = 0.f;
    if (part == 0) {
        xx = (red[0][r] + red[1][r]) + (red[2][r] + red[3][r]);
        at[r] = n < N ? -xx : -INFINITY;
        xx = n < N ? xx : 0.f;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) xx = fmaxf(xx, __shfl_xor(xx, d, 64));
    }
    __syncthreads();
    if (part == 0 && (t & 63) == 0) tmax[t >> 6] = xx;
    __syncthreads();
    if (t == 0) {
        at[128] = ldexpf(1.0f, -T);
        at[129] = sqrtf(fmaxf(tmax[0], tmax[1])) * 1.000001f;
    }
}

// T largest of the values inserted so far, descending (TopKV's v_med3 network, common.h)
template <int T>
struct FkTop {
    float v[T];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < T; i++) v[i] = -INFINITY;
    }
    __device__ __forceinline__ void insert(float key) {
#pragma unroll
        for (int i = T - 1; i > 0; i--) v[i] = __builtin_amdgcn_fmed3f(v[i - 1], v[i], key);
        v[0] = fmaxf(v[0], key);
    }
};

// KC: list-length class (20, 32, 64: k <= KC).  KC = 20 keeps the queries' lists in LDS (value + 16-bit index, 64 entries), the longer
// classes in the workspace (64-bit keys).  NCH_RES: channel chunks of 64 whose query fragments stay in registers (1 or 2: C <= 128);
// 0: any C, the query fragments of a chunk are fetched from L2 one unit ahead.
template <int KC, int NCH_RES>
__global__ __launch_bounds__(512) void featknn_kernel(const uint4 *__restrict__ planes, const float *__restrict__ aux,
                                                       const float *__restrict__ x, int C, int Cp, int N, int Np, int k,
                                                       fk_u64 *__restrict__ lists, int64_t *__restrict__ idx_out)
{
    constexpr int CAP = fk_cap(KC), T = KC / 4 + 1;            // 4 T >= KC + 4 group maxima per query decide its bound
    constexpr bool LL = KC <= 20;                              // lists in LDS
    constexpr bool STREAM = NCH_RES == 0;
    constexpr int NBF = STREAM ? 2 : NCH_RES;                 // query fragment sets held in registers
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int qb = wave & 3, kh = wave >> 2;                  // query block, key half
    const int col = lane & 31, hf = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * FK_QT;
    const int ql = qb * 32 + col, q = q0 + ql;                // q < Np always
    const int noct = Cp / 8, nch = Cp / 64, nkt = Np / 128, U = nkt * nch;
    const uint4 *pb = planes + (size_t)b * 2 * noct * Np;
    const float *ab = aux + (size_t)b * nkt * FK_AUXT;
    float *dump = (float *)(lds + FK_DUMP_OFF) + wave * 17 * 64;
    float *tv = (float *)lds;                                  // [128][4 T + 1]: the lanes' T largest group maxima (between the sweeps, in the idle stages)
    int *qcnt = (int *)(lds + FK_CNT_OFF);
    int *flags = (int *)(lds + FK_FLAG_OFF);
    float *lv = (float *)(lds + FK_LIST_OFF);
    unsigned short *li = (unsigned short *)(lds + FK_LIST_OFF + 128 * FK_LVS * 4);
    static_assert((4 * T + 1) * 128 * 4 <= FK_NSTAGE * FK_STAGE, "the bound exchange fits the stages");

    FKM(0)
    if (t < 128) qcnt[t] = 0;
    if (t == 0) flags[129] = 0;
    const float xxq = -ab[(q >> 7) * FK_AUXT + (q & 127)], sq2 = 2.0f * ab[(q >> 7) * FK_AUXT + 128];
    // Sweep 0 ranks with ONE product (h h').  What the two dropped products can add to a ranking value is below
    // 2 (2^-11 + 2^-11 + 2^-22) |x_i| |x_j| (+ the accumulations' own rounding) < 0x1.1p-9 |x_i| max_j |x_j|: the bound taken from the
    // one-product group maxima is lowered by that much and stays a value that at least KC exact ranking values reach.
    float sxmax = 0.f;
    for (int kt_ = 0; kt_ < nkt; kt_++) sxmax = fmaxf(sxmax, ab[(size_t)kt_ * FK_AUXT + 129]);
    const float margin = 0x1.1p-9f * sqrtf(xxq) * sxmax;

    // ---- DMA pieces of a unit (64 keys per half x 64 channels): piece z = wave * 4 + i -> (half z >> 4, plane (z >> 3) & 1, octet z & 7),
    // 64 rows of 16 B = 1 KB, one global_load_lds_dwordx4 per wave; wave 0 adds the tile's aux (1 KB) on the tile's last chunk
    auto issue = [&](int u, int stage) {
        const int kt = u / nch, ch = u - kt * nch;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int z = wave * 4 + i, half = z >> 4, p = (z >> 3) & 1, o = z & 7;
            const uint4 *src = pb + ((size_t)(p * noct + ch * 8 + o)) * Np + kt * 128 + half * 64 + lane;
            __builtin_amdgcn_global_load_lds((fk_gbl_ptr_t)src, (fk_lds_ptr_t)(lds + stage * FK_STAGE + z * 1024), 16, 0, 0);
        }
        if (ch == nch - 1 && wave == 0)
            __builtin_amdgcn_global_load_lds((fk_gbl_ptr_t)((const uint4 *)(ab + (size_t)kt * FK_AUXT) + lane),
                                             (fk_lds_ptr_t)(lds + stage * FK_STAGE + 32768), 16, 0, 0);
    };
    // query fragments of channel chunk ch: k-step s, lane (col, hf) -> octet 2 s + hf of row q, planes h and m
    fk_f16x8 Bh[NBF][4], Bm[NBF][4];
    auto load_b = [&](int ch, int set) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const size_t cell = (size_t)(ch * 8 + 2 * s + hf) * Np + q;
            Bh[set][s] = __builtin_bit_cast(fk_f16x8, pb[cell]);
            Bm[set][s] = __builtin_bit_cast(fk_f16x8, pb[(size_t)noct * Np + cell]);
        }
    };
    if constexpr (!STREAM) {
#pragma unroll
        for (int c_ = 0; c_ < NCH_RES; c_++)
            if (c_ < nch) load_b(c_, c_);
        // a use of every fragment register HERE: the compiler waits for the loads in front of it.  Left to their first use inside the
        // unit loop, that wait (s_waitcnt vmcnt(0), every iteration) also waited for the DMA pieces issued a few instructions earlier --
        // every unit then cost an L2 round trip (1.7 us per unit instead of 0.5: tools/fk_timeline.py)
#pragma unroll
        for (int c_ = 0; c_ < NCH_RES; c_++)
#pragma unroll
            for (int s_ = 0; s_ < 4; s_++) asm volatile("" :: "v"(Bh[c_][s_]), "v"(Bm[c_][s_]));
    }
    const int a_off = kh * 16384 + hf * 1024 + col * 16;      // + plane * 8192 + s * 2048 + a * 512

    float thr = -INFINITY;
    auto run_sweep = [&](auto sweep_c) {
        constexpr int sweep = decltype(sweep_c)::value;
        float mr[2][16];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) mr[a][r] = -INFINITY;
        f32x16 acc[2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
        __syncthreads();                                       // the previous sweep's last stage reads are done (and the setup above)
        dump[16 * 64 + lane] = 0.f;
        // Double buffer: unit u + 1's pieces are issued at the top of unit u (right behind the barrier that retires their stage) and
        // waited for, with everything else this wave has in flight, at the top of unit u + 1.  (A third stage with counted waits,
        // conv_f16.hip's scheme, measured no faster and needs the 34 KB of the dump area: LABLOG R6.3.)
        if constexpr (STREAM) load_b(0, 0);
        issue(0, 0);
        int kt = 0, ch = 0, stage = 0;
        // one unit; PAR = u & 1 at compile time when the query fragments stream (their two register sets alternate: with a runtime
        // parity the compiler made the sets an indexed private array -- scratch)
        auto unit = [&](int u, auto par_c) {
            constexpr int PAR = decltype(par_c)::value;
            const int chn = ch + 1 == nch ? 0 : ch + 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // ... everybody's; the other stage was last read in unit u - 1
            if constexpr (STREAM) {
                if (u + 1 < U) load_b(chn, PAR ^ 1);
            }
            if (u + 1 < U) issue(u + 1, stage ^ 1);
            const unsigned char *base = lds + stage * FK_STAGE + a_off;
            const int set = STREAM ? PAR : ch;
            // the fragments of k-step s + 1 are read while the products of step s run, and no further ahead (sched_barrier): the
            // register budget decides whether two sets of query fragments fit beside them
            auto products = [&](const fk_f16x8 (&bh)[4], const fk_f16x8 (&bm)[4]) {
                fk_f16x8 Ah[2][2], Am[2][2];
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    Ah[0][a] = *(const fk_f16x8 *)(base + a * 512);
                    if (sweep == 1) Am[0][a] = *(const fk_f16x8 *)(base + 8192 + a * 512);
                }
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if (s < 3) {
#pragma unroll
                        for (int a = 0; a < 2; a++) {
                            Ah[(s + 1) & 1][a] = *(const fk_f16x8 *)(base + (s + 1) * 2048 + a * 512);
                            if (sweep == 1) Am[(s + 1) & 1][a] = *(const fk_f16x8 *)(base + 8192 + (s + 1) * 2048 + a * 512);
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 2; a++) {          // smallest first: m h', h m', h h'  (sweep 0: h h' alone, see margin)
                        if (sweep == 1) {
                            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Am[s & 1][a], bh[s], acc[a], 0, 0, 0);
                            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s & 1][a], bm[s], acc[a], 0, 0, 0);
                        }
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s & 1][a], bh[s], acc[a], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if constexpr (STREAM) products(Bh[PAR], Bm[PAR]);
            else if (NBF == 1 || set == 0) products(Bh[0], Bm[0]);
            else products(Bh[NBF - 1], Bm[NBF - 1]);

            if (ch == nch - 1 && !(FK_ABL & 8)) {
                // ---- epilogue of key tile kt: p0 = -xx_j + 2 <x_i, x_j> for this lane's 2 x 16 candidates (the reference's pd = p0 - xx_i is
                // formed for the collected candidates only: x -> fl(x - xx_i) is monotone, so selecting on p0 selects the same set)
                // The tile's aux arrived as a DMA piece; read with compiler-visible loads, the compiler orders them behind EVERY LDS DMA it has
                // seen -- s_waitcnt vmcnt(0) right behind the pieces just issued for unit u + 1, an L2 round trip per tile.  The piece of
                // this unit has landed (the wait and the barrier at the top): inline asm reads, waited for by hand.
                const unsigned aux_addr = fk_lds_off(lds + stage * FK_STAGE + 32768);
                float fsc;
                f32x4 nq[2][4];
                {
                    const unsigned arow = aux_addr + (unsigned)(kh * 64 + 4 * hf) * 4u;
                    // ONE statement, reads and wait together: between separate asm statements the compiler may move a register whose
                    // load it cannot know is still in flight
                    asm volatile("ds_read_b32 %0, %9 offset:512\n\t"
                                 "ds_read_b128 %1, %10\n\tds_read_b128 %2, %10 offset:32\n\tds_read_b128 %3, %10 offset:64\n\tds_read_b128 %4, %10 offset:96\n\t"
                                 "ds_read_b128 %5, %10 offset:128\n\tds_read_b128 %6, %10 offset:160\n\tds_read_b128 %7, %10 offset:192\n\tds_read_b128 %8, %10 offset:224\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(fsc), "=&v"(nq[0][0]), "=&v"(nq[0][1]), "=&v"(nq[0][2]), "=&v"(nq[0][3]), "=&v"(nq[1][0]), "=&v"(nq[1][1]),
                                   "=&v"(nq[1][2]), "=&v"(nq[1][3])
                                 : "v"(aux_addr), "v"(arow)
                                 : "memory");
                }
                const int krow = kh * 64 + 4 * hf;                       // row of the 128-key tile: + 32 a + (r & 3) + 8 (r >> 2)
                const float fkt = sq2 * fsc;                             // 2 x 2^-T_i x 2^-T_j: exact
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    float pd[16];
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const f32x4 n4 = nq[a][g];
                        pd[4 * g] = fmaf(acc[a][4 * g], fkt, n4[0]);         pd[4 * g + 1] = fmaf(acc[a][4 * g + 1], fkt, n4[1]);
                        pd[4 * g + 2] = fmaf(acc[a][4 * g + 2], fkt, n4[2]); pd[4 * g + 3] = fmaf(acc[a][4 * g + 3], fkt, n4[3]);
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
                    if (sweep == 0) {
#pragma unroll
                        for (int r = 0; r < 16; r++) mr[a][r] = fmaxf(mr[a][r], pd[r]);
                    } else {
                        unsigned m = 0;
#pragma unroll
                        for (int r = 0; r < 16; r++) m |= pd[r] >= thr ? (1u << r) : 0u;
                        if (!(FK_ABL & 16) && __builtin_amdgcn_ballot_w64(m != 0) != 0) {
                            // Every LDS access of the collection as inline asm, for the reason the aux reads are: the compiler puts
                            // s_waitcnt vmcnt(0) in front of any LDS access that may alias a DMA destination -- here, behind the pieces just
                            // issued for the next unit.  The 16 values go to this wave's dump rows (a lane's column), a hit's value comes
                            // back by its register number; that read and the slot atomic are in flight together.
                            const unsigned daddr = fk_lds_off(dump + lane);
                            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:256\n\tds_write_b32 %0, %3 offset:512\n\tds_write_b32 %0, %4 offset:768\n\t"
                                         "ds_write_b32 %0, %5 offset:1024\n\tds_write_b32 %0, %6 offset:1280\n\tds_write_b32 %0, %7 offset:1536\n\t"
                                         "ds_write_b32 %0, %8 offset:1792\n\tds_write_b32 %0, %9 offset:2048\n\tds_write_b32 %0, %10 offset:2304\n\t"
                                         "ds_write_b32 %0, %11 offset:2560\n\tds_write_b32 %0, %12 offset:2816\n\tds_write_b32 %0, %13 offset:3072\n\t"
                                         "ds_write_b32 %0, %14 offset:3328\n\tds_write_b32 %0, %15 offset:3584\n\tds_write_b32 %0, %16 offset:3840"
                                         :: "v"(daddr), "v"(pd[0]), "v"(pd[1]), "v"(pd[2]), "v"(pd[3]), "v"(pd[4]), "v"(pd[5]), "v"(pd[6]), "v"(pd[7]),
                                            "v"(pd[8]), "v"(pd[9]), "v"(pd[10]), "v"(pd[11]), "v"(pd[12]), "v"(pd[13]), "v"(pd[14]), "v"(pd[15])
                                         : "memory");
                            const int cand0 = kt * 128 + krow + 32 * a;
#pragma unroll 1
                            do {
                                const bool has = m != 0;
                                const int e = has ? __builtin_ctz(m) : 16;
                                m &= m - 1;
                                float v;
                                int slot;
                                asm volatile("ds_read_b32 %0, %2\n\tds_add_rtn_u32 %1, %3, %4\n\ts_waitcnt lgkmcnt(0)"
                                             : "=&v"(v), "=&v"(slot) : "v"(daddr + (unsigned)e * 256u), "v"(fk_lds_off(qcnt + ql)), "v"(has ? 1 : 0) : "memory");
                                v = v - xxq;                                           // the reference's pd
                                const unsigned cand = (unsigned)(cand0 + (e & 3) + 8 * (e >> 2));
                                if (has && slot < CAP) {           // else: the count says so (rank phase)
                                    if constexpr (LL) {
                                        asm volatile("ds_write_b32 %0, %1" :: "v"(fk_lds_off(lv + ql * FK_LVS + slot)), "v"(v) : "memory");
                                        asm volatile("ds_write_b16 %0, %1" :: "v"(fk_lds_off(li + ql * FK_LIS + slot)), "v"(cand) : "memory");
                                    } else {
                                        const unsigned bits = __float_as_uint(v + 0.0f);                         // -0 -> +0
                                        const unsigned sk = bits ^ ((unsigned)((int)bits >> 31) | 0x80000000u);   // unsigned order == float order
                                        lists[((size_t)b * Np + q) * CAP + slot] = ((fk_u64)sk << 32) | (unsigned)(~cand);
                                    }
                                }
                            } while (__builtin_amdgcn_ballot_w64(m != 0) != 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);         // one row block's 16 values at a time
                }
            }
            ch++;
            if (ch == nch) { ch = 0; kt++; }
            stage ^= 1;
        };
        if constexpr (STREAM) {
#pragma unroll 1
            for (int u = 0; u < U; u += 2) {
                unit(u, std::integral_constant<int, 0>{});
                if (u + 1 < U) unit(u + 1, std::integral_constant<int, 1>{});
            }
        } else {
#pragma unroll 1
            for (int u = 0; u < U; u++) unit(u, std::integral_constant<int, 0>{});
        }
        if (sweep == 0) {
            FKM(2)
            // ---- the bound: the KC-th largest of the 4 T values the query's four lanes offer (each its T largest group maxima: distinct
            // candidates, so at least KC candidates reach it)
            FkTop<T> top;
            top.init();
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int r = 0; r < 16; r++) top.insert(mr[a][r]);
            __syncthreads();                                   // every wave is past its last stage read
            float *mine = tv + ql * (4 * T + 1) + (kh * 2 + hf) * T;
#pragma unroll
            for (int i = 0; i < T; i++) mine[i] = top.v[i];
            __syncthreads();
            int cnt[T];
#pragma unroll
            for (int i = 0; i < T; i++) cnt[i] = 0;
            for (int f = 0; f < 4 * T; f++) {
                const float v = tv[ql * (4 * T + 1) + f];
#pragma unroll
                for (int i = 0; i < T; i++) cnt[i] += v > top.v[i] ? 1 : 0;
            }
            // the smallest own value with fewer than KC values above it (own values descend, so the counts ascend); the query's bound is
            // the smallest such offer of its four lanes
            float offer = INFINITY;
#pragma unroll
            for (int i = 0; i < T; i++) offer = cnt[i] < KC ? top.v[i] : offer;
            __syncthreads();
            tv[ql * 5 + kh * 2 + hf] = offer;
            __syncthreads();
            const float *t4 = tv + ql * 5;
            thr = fmaxf(fminf(fminf(t4[0], t4[1]), fminf(t4[2], t4[3])) - margin, -3.0e38f);     // -inf (fewer than KC real candidates): every finite one
            if (FK_ABL & 2) thr = INFINITY;
        }
    };
    FKM(1)
    run_sweep(std::integral_constant<int, 0>{});
    FKM(3)
    if (!(FK_ABL & 4)) run_sweep(std::integral_constant<int, 1>{});
    FKM(4)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the asm LDS writes of the collection are not the compiler's to wait for)
    __syncthreads();

    // ---- rank: four threads per query, thread part ranks keys part, part + 4, ...; a key of rank r < k gives idx[q][r]
    if (FK_ABL & 32) return;
    if constexpr (LL) {
        const int rq = t >> 2, part = t & 3, gq = q0 + rq;
        const int M = qcnt[rq];
        const bool ranked = gq < N && M <= CAP && !(FK_ABL & 1);
        int *outl = (int *)lds;                                 // [128][k] ranks -> indices, in the idle stages; written out coalesced below
        // own keys part, part + 4, ... in registers, ONE pass over the list; no branches: a compare written with && / || became an
        // exec-mask region and a wait per element here (52 us of a 112 us kernel).  OWN = 8 covers lists of up to 32 keys (the mean
        // is 24, the 99th percentile 34); a wave with a longer list among its 16 queries takes the 16-key form.
        auto rank_lists = [&](auto own_c) {
            constexpr int OWN = decltype(own_c)::value;
            const float *V = lv + rq * FK_LVS;
            const unsigned short *I = li + rq * FK_LIS;
            // (value, index) as ONE 64-bit key -- order-preserving value bits << 32 | ~index: larger = nearer, the lower index first under
            // equal values -- so that a comparison is v_cmp_gt_u64 + v_addc instead of three compares and two logic operations
            auto key_of = [](float v, unsigned i) {
                const unsigned bits = __float_as_uint(v + 0.0f);                               // -0 -> +0
                const unsigned sk = bits ^ ((unsigned)((int)bits >> 31) | 0x80000000u);
                return ((fk_u64)sk << 32) | (unsigned)(~i);
            };
            fk_u64 own[OWN];
            int oi[OWN], rank[OWN];
#pragma unroll
            for (int j = 0; j < OWN; j++) {
                const int o = part + 4 * j;
                oi[j] = o < M ? (int)I[o] : -1;
                own[j] = o < M ? key_of(V[o], (unsigned)oi[j]) : ~(fk_u64)0;
                rank[j] = 0;
            }
#pragma unroll 4
            for (int f = 0; f < M; f++) {
                const fk_u64 kf = key_of(V[f], (unsigned)I[f]);
#pragma unroll
                for (int j = 0; j < OWN; j++) rank[j] += (int)(kf > own[j]);
            }
#pragma unroll
            for (int j = 0; j < OWN; j++)
                if (part + 4 * j < M && rank[j] < k) outl[rq * k + rank[j]] = oi[j];
        };
        if (__builtin_amdgcn_ballot_w64(ranked && M > 32) != 0) {
            if (ranked) rank_lists(std::integral_constant<int, FK_LDS_CAP / 4>{});
        } else {
            if (ranked) rank_lists(std::integral_constant<int, 8>{});
        }
        if (!ranked && gq < N && part == 0) flags[1 + atomicAdd(&flags[129], 1)] = rq;     // overflowed: the exact selection below
        __syncthreads();
        {
            // the workgroup's rows of idx_out are one contiguous block: 8-byte stores, 512 B per wave instruction (a row of an
            // overflowed query carries stale values here; the fallback below overwrites it)
            const int rows = min(FK_QT, N - q0);
            int64_t *dst = idx_out + ((size_t)b * N + q0) * k;
            for (int e = t; e < rows * k; e += 512) dst[e] = outl[e];
        }
    } else {
        // lists in the workspace: staged through the (idle) key stages, 32 queries at a time, then ranked from LDS
        fk_u64 *Ls = (fk_u64 *)lds;
        static_assert(32 * CAP * 8 <= FK_NSTAGE * FK_STAGE, "a batch of lists fits the stages");
        for (int bq = 0; bq < 128; bq += 32) {
            __syncthreads();
            for (int e = t; e < 32 * CAP; e += 512) {
                const int rq = bq + e / CAP, o = e % CAP;
                if (o < min(qcnt[rq], CAP))
                    Ls[e] = __hip_atomic_load(&lists[((size_t)b * Np + q0 + rq) * CAP + o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            const int rq = bq + (t >> 4), part = t & 15, gq = q0 + rq;
            const int M = qcnt[rq];
            if (gq < N && M <= CAP && !(FK_ABL & 1)) {
                const fk_u64 *L = Ls + (t >> 4) * CAP;
                int64_t *dst = idx_out + ((size_t)b * N + gq) * k;
                constexpr int OWN = CAP / 16;
                fk_u64 own[OWN];
                int rank[OWN];
#pragma unroll
                for (int j = 0; j < OWN; j++) {
                    const int o = part + 16 * j;
                    own[j] = o < M ? L[o] : ~0ull;
                    rank[j] = 0;
                }
#pragma unroll 4
                for (int f = 0; f < M; f++) {
                    const fk_u64 kf = L[f];
#pragma unroll
                    for (int j = 0; j < OWN; j++) rank[j] += (int)(kf > own[j]);
                }
#pragma unroll
                for (int j = 0; j < OWN; j++)
                    if (part + 16 * j < M && rank[j] < k) dst[rank[j]] = (int64_t)(unsigned)(~(unsigned)own[j]);
            } else if (gq < N && part == 0) {
                flags[1 + atomicAdd(&flags[129], 1)] = rq;
            }
        }
    }
    __syncthreads();
    FKM(5)
    const int nfl = flags[129];
    if (nfl == 0) return;
    // ---- exact fallback: a wave per overflowed query, its Np ranking values in the wave's eighth of the (idle) stages when they fit
    // there (Np <= 2048), else wave 0 alone with all of it
    const bool wide = Np > 2048;
    if (wide && wave != 0) return;
    float *vals = (float *)lds + (wide ? 0 : wave * 2048);
    const float *xb = x + (size_t)b * C * N;
    for (int f = wide ? 0 : wave; f < nfl; f += wide ? 1 : 8) {
        const int gq = q0 + flags[1 + f];
        const float xxg = -ab[(gq >> 7) * FK_AUXT + (gq & 127)];
        for (int j0 = 0; j0 < Np; j0 += 64) {
            const int j = j0 + lane;
            float dot = 0.f;
            if (j < N)
                for (int c = 0; c < C; c++) dot = fmaf(xb[(size_t)c * N + gq], xb[(size_t)c * N + j], dot);
            vals[j] = j < N ? fmaf(2.0f, dot, ab[(j >> 7) * FK_AUXT + (j & 127)]) - xxg : -INFINITY;
        }
        int64_t *dst = idx_out + ((size_t)b * N + gq) * k;
        for (int o = 0; o < k; o++) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int j = lane; j < Np; j += 64) {
                const float v = vals[j];
                if (v > bv) { bv = v; bi = j; }                 // ascending j: the first of equal values stays
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                const float ov = __shfl_xor(bv, d, 64);
                const int oi = __shfl_xor(bi, d, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (lane == 0) { dst[o] = bi; vals[bi] = -INFINITY; }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

extern "C" size_t l3d_knn_feature_workspace_bytes(int B, int C, int N)
{
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    const size_t Np = (size_t)l3d_divup(N, 128) * 128, Cp = (size_t)l3d_divup(C, 64) * 64;
    // fp16 planes (h | m) | aux (-|x|^2, 2^-T) | the queries' candidate lists at the longest capacity (k > 20 only; k <= 20 keeps them in LDS)
    return (size_t)B * Cp * Np * 4 + (size_t)B * Np * 8 + (size_t)B * Np * fk_cap(64) * 8;
}

template <int KC, int NCH_RES>
static int fk_launch(const uint4 *planes, const float *aux, const float *x, int B, int C, int Cp, int N, int Np, int k, fk_u64 *lists,
                     int64_t *idx, hipStream_t st)
{
    const size_t nlds = KC <= 20 ? FK_LDS_LL : FK_LIST_OFF;
    static const bool ok = hipFuncSetAttribute((const void *)featknn_kernel<KC, NCH_RES>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)nlds) == hipSuccess;
    if (!ok) return L3D_ERR_LAUNCH;
    hipLaunchKernelGGL((featknn_kernel<KC, NCH_RES>), dim3(Np / FK_QT, B), dim3(512), nlds, st, planes, aux, x, C, Cp, N, Np, k, lists, idx);
    return l3d_check_launch();
}

extern "C" int l3d_knn_feature(const float *x, int B, int C, int N, int k, void *workspace, int64_t *idx,
                               l3d_stream_t stream)
{
    L3D_REQUIRE(x && workspace && idx && B > 0 && C > 0 && N > 0 && k > 0);
    if (k > N) return L3D_ERR_INVALID_ARG;
    const int Np = l3d_divup(N, 128) * 128, Cp = l3d_divup(C, 64) * 64;          // any C: padded with zero channels
    if (k > 64 || B > 65535 || Np > FK_MAXNP || (((size_t)workspace) & 15)) return L3D_ERR_UNSUPPORTED;
    unsigned char *ws = (unsigned char *)workspace;
    uint4 *planes = (uint4 *)ws;
    float *aux = (float *)(ws + (size_t)B * Cp * Np * 4);
    fk_u64 *lists = (fk_u64 *)(ws + (size_t)B * Cp * Np * 4 + (size_t)B * Np * 8);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(featknn_split_kernel, dim3(Np / 128, B), dim3(512), 0, st, x, C, Cp, N, Np, planes, aux);
    const int nch = Cp / 64;
#define FK_GO(KK)                                                                                                        \
    (nch == 1 ? fk_launch<KK, 1>(planes, aux, x, B, C, Cp, N, Np, k, lists, idx, st)                                     \
              : (nch == 2 ? fk_launch<KK, 2>(planes, aux, x, B, C, Cp, N, Np, k, lists, idx, st)                         \
                          : fk_launch<KK, 0>(planes, aux, x, B, C, Cp, N, Np, k, lists, idx, st)))
    // list-length classes: k <= 20 (the DGCNN / PRNet graphs), <= 32, <= 64
    const int rc = k <= 20 ? FK_GO(20) : (k <= 32 ? FK_GO(32) : FK_GO(64));
#undef FK_GO
    return rc;
}
