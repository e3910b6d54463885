// featknn.hip -- k nearest neighbours in FEATURE space: knn(x, k) of
// utils/model_common_utils.py:3-9 for x [B,C,N] with C = 32..512 (the dynamic graphs of PRNet's DGCNN,
// models/prnet.py:76-97), without the [B,N,N] inner-product / distance matrices:
//     inner = -2 x^T x;  xx = sum_c x^2;  pd = -xx_j - inner_ij - xx_i;  idx = topk(pd, k)
// The inner product is a real GEMM here (2*C flops per pair), so it runs on the matrix cores in the bf16x3
// arithmetic of conv_split.hip (x = h + m + l exactly, six bf16 products, fp32 accumulate: fp32-level error),
// with the top-k selection as the GEMM's epilogue, flash-attention style:
//   * featknn_split_kernel: one pass over x -> the three bf16 planes in MFMA operand order
//     [B][C/16][plane 3][kg 2][Np][8 bf16] (Np = N rounded up to 128, zero padded) and nxx = -sum_c x^2
//     (-inf on the padding).  The SAME buffer is both GEMM operands (keys and queries).
//   * featknn_kernel: workgroup = 4 waves = 128 queries; wave = 32 queries (MFMA columns) x 128-key tiles
//     (MFMA rows, "swapped" orientation: a lane owns ONE query and 64 of the tile's 128 keys, its partner
//     lane l^32 the other 64).  Key chunks of 32 channels go through a double-buffered LDS tile shared by
//     the four waves; query fragments come straight from L2 one unit ahead.  After the last channel chunk
//     the lane forms pd exactly in the reference's op order, keeps candidates that beat its current k-th
//     best in a bit mask and runs the (value, index) insertion network only for those.  The two lanes of a
//     pair merge their lists once at the end (ties -> lower index first, as knn.hip).
// Indices cannot be bit-pinned to the reference here (its sgemm's summation order is MKL's); the parity
// test bounds every returned neighbour by the exact k-th distance.
#include "common.h"
#include "split_bf16.h"

#ifndef FK_PROBE
#define FK_PROBE 0                                   // tools/probe_featknn.hip: 1 = no insertions, 2 = GEMM only
#endif
#ifdef FK_COUNT
__device__ unsigned long long fk_trip_counter;
#endif
#define FK_REGION (128 * 16 + 64)
#define FK_BUF (12 * FK_REGION)                      // 32 channels x 128 keys x 3 planes
#define FK_NXOFF (2 * FK_BUF)                        // [2][128] floats
#define FK_SCROFF (FK_NXOFF + 2 * 128 * 4)           // [4 waves][17][64] floats (row 16 = -inf)
#define FK_LDS (FK_SCROFF + 4 * 17 * 64 * 4)

// C: the tensor's channels; Cp: C rounded up to a multiple of 32 (the GEMM's K chunk) -- the pad channels are zeros,
// which change neither the dot products nor |x|^2
__global__ __launch_bounds__(256) void featknn_split_kernel(const float *__restrict__ x, int C, int Cp, int N, int Np,
                                                            uint4 *__restrict__ xs, float *__restrict__ nxx)
{
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (n >= Np) return;
    const float *xb = x + (size_t)b * C * N;
    uint4 *xsb = xs + (size_t)b * (Cp / 16) * 6 * Np;
    float s = 0.f;
    for (int c8 = 0; c8 < Cp / 8; c8++) {              // kc16 = c8 >> 1, kg = c8 & 1
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (n < N && c8 * 8 + e < C) ? xb[(size_t)(c8 * 8 + e) * N + n] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) s = s + v[e] * v[e];          // x ** 2 then sum: no fused multiply-add
        uint4 h, m, l;
        split8(v, h, m, l);
        const size_t base = ((size_t)(c8 >> 1) * 6 + (c8 & 1)) * Np + n;
        xsb[base] = h;
        xsb[base + 2 * (size_t)Np] = m;
        xsb[base + 4 * (size_t)Np] = l;
    }
    nxx[(size_t)b * Np + n] = n < N ? -s : -INFINITY;
}

// (value desc, index asc) insertion: used once, to merge the two lists of a lane pair
template <int K>
__device__ __forceinline__ void fk_insert_lex(TopK<K> &t, float key, int j)
{
#define FK_BEFORE(i) (key > t.v[i] || (key == t.v[i] && j < t.id[i]))
    bool b_prev = FK_BEFORE(K - 1);
#pragma unroll
    for (int i = K - 1; i > 0; i--) {
        const bool b_up = FK_BEFORE(i - 1);
        t.id[i] = b_up ? t.id[i - 1] : (b_prev ? j : t.id[i]);
        t.v[i] = b_up ? t.v[i - 1] : (b_prev ? key : t.v[i]);
        b_prev = b_up;
    }
    t.id[0] = b_prev ? j : t.id[0];
    t.v[0] = b_prev ? key : t.v[0];
#undef FK_BEFORE
}

template <int K>
__global__ __launch_bounds__(256, K <= 20 ? 2 : 1) void featknn_kernel(const uint4 *__restrict__ xs, const float *__restrict__ nxx,
                                                         int C, int N, int Np, int k, int64_t *__restrict__ idx_out,
                                                         float *__restrict__ part_v, int *__restrict__ part_i)
{
    // Key-range split (round 5): with B N / 128 < 512 workgroups a CU holds ONE and every unit's load -> LDS -> barrier -> MFMA chain
    // runs exposed (LABLOG R5.4).  gridDim.z parts each rank their share of the key tiles for the same 128 queries (a second
    // workgroup per CU to switch to) and leave their sorted K-lists in part_v / part_i; featknn_merge_kernel merges them.
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.y, q0 = blockIdx.x * 128;
    const int nch = C / 32, nkt_all = Np / 128;
    const int kt0 = (int)((long)nkt_all * blockIdx.z / gridDim.z), kt1 = (int)((long)nkt_all * (blockIdx.z + 1) / gridDim.z);
    const int U = (kt1 - kt0) * nch;
    const uint4 *xsb = xs + (size_t)b * (C / 16) * 6 * Np;
    const float *nxb = nxx + (size_t)b * Np;
    float *nxl = (float *)(lds + FK_NXOFF);
    float *scr = (float *)(lds + FK_SCROFF) + wave * 17 * 64;
    scr[16 * 64 + lane] = -INFINITY;

    // staging: thread -> key row t & 127 of regions 2j + (t >> 7), j < 6; region r of channel chunk ch sits
    // at ((ch * 12 + r) * Np + key) in the split buffer
    const int srow = t & 127, shalf = t >> 7;
    const int s_lds = shalf * FK_REGION + srow * 16;
    // query fragments: lane (i = l & 31, kg = l >> 5) -> 8 channels of query q0 + 32 wave + i
    const int qrow = q0 + wave * 32 + (lane & 31);
    const size_t q_off = (size_t)(lane >> 5) * Np + qrow;
    const int a_off = (lane >> 5) * FK_REGION + (lane & 31) * 16;
    const float xxq = -nxb[qrow];                                  // xx_i (qrow < Np always)

    uint4 k0, k1, k2, k3, k4, k5;            // key chunk in flight
    uint4 qn[2][3];                          // query fragments of the next unit
    bf16x8 qc[2][3];                         // ... of the current unit
    float nxr = 0.f;

#define FK_LOAD(KT, CH)                                                                                   \
    do {                                                                                                  \
        const uint4 *src_ = xsb + ((size_t)(CH) * 12 + shalf) * Np + (KT) * 128 + srow;                   \
        k0 = src_[0];                                                                                     \
        k1 = src_[2 * (size_t)Np];                                                                        \
        k2 = src_[4 * (size_t)Np];                                                                        \
        k3 = src_[6 * (size_t)Np];                                                                        \
        k4 = src_[8 * (size_t)Np];                                                                        \
        k5 = src_[10 * (size_t)Np];                                                                       \
        const uint4 *qs_ = xsb + (size_t)(CH) * 12 * Np + q_off;                                          \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; s_++)                                                  \
            _Pragma("unroll") for (int p_ = 0; p_ < 3; p_++) qn[s_][p_] = qs_[(size_t)((s_ * 3 + p_) * 2) * Np]; \
        if ((CH) == 0 && t < 128) nxr = nxb[(KT) * 128 + t];                                              \
    } while (0)
#define FK_STORE(BUF, KT, CH)                                                                             \
    do {                                                                                                  \
        unsigned char *base_ = lds + (BUF) * FK_BUF + s_lds;                                              \
        *(uint4 *)(base_) = k0;                                                                           \
        *(uint4 *)(base_ + 2 * FK_REGION) = k1;                                                           \
        *(uint4 *)(base_ + 4 * FK_REGION) = k2;                                                           \
        *(uint4 *)(base_ + 6 * FK_REGION) = k3;                                                           \
        *(uint4 *)(base_ + 8 * FK_REGION) = k4;                                                           \
        *(uint4 *)(base_ + 10 * FK_REGION) = k5;                                                          \
        if ((CH) == 0 && t < 128) nxl[((KT) & 1) * 128 + t] = nxr;                                        \
    } while (0)
#define FK_QSWAP()                                                                                        \
    do {                                                                                                  \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; s_++)                                                  \
            _Pragma("unroll") for (int p_ = 0; p_ < 3; p_++) qc[s_][p_] = __builtin_bit_cast(bf16x8, qn[s_][p_]); \
    } while (0)

    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    static_assert(K == 20 || K == 32 || K == 64, "instantiated list lengths");
#ifdef FK_COUNT
    int trips = 0;
#endif
    TopK<K> top;
    top.init();
    float thr = -INFINITY, thrp = -INFINITY;

    int kt = kt0, ch = 0;                    // current unit
    int ktn = kt0, chn = 0;                  // next unit to fetch
    FK_LOAD(kt0, 0);
    FK_STORE(0, kt0, 0);
    FK_QSWAP();
    chn = 1;
    if (chn == nch) { chn = 0; ktn = kt0 + 1; }
    __syncthreads();

#pragma unroll 1
    for (int u = 0; u < U; u++) {
        const bool more = u + 1 < U;
        if (more) FK_LOAD(ktn, chn);
        const unsigned char *base = lds + (u & 1) * FK_BUF + a_off;
        // six plane steps (k-step s, key plane pa = l, m, h); the fragments of step i + 1 are read while the
        // products of step i run, and no further ahead (sched_barrier): the register budget is what matters here
        bf16x8 A[2][4];
#pragma unroll
        for (int a = 0; a < 4; a++) A[0][a] = *(const bf16x8 *)(base + (2 * 2) * FK_REGION + a * 512);
#pragma unroll
        for (int st = 0; st < 6; st++) {
            const int s = st / 3, pa = 2 - st % 3;
            if (st < 5) {
                const int sn = (st + 1) / 3, pn = 2 - (st + 1) % 3;
#pragma unroll
                for (int a = 0; a < 4; a++)
                    A[(st + 1) & 1][a] = *(const bf16x8 *)(base + ((sn * 3 + pn) * 2) * FK_REGION + a * 512);
            }
#pragma unroll
            for (int pb = 2; pb >= 0; pb--) {
                if (pa + pb > 2) continue;
#pragma unroll
                for (int a = 0; a < 4; a++)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[st & 1][a], qc[s][pb], acc[a], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) FK_STORE((u + 1) & 1, ktn, chn);

#if FK_PROBE == 2
        if (false) {
#else
        if (ch == nch - 1) {
#endif
            // ---- epilogue of key tile kt: pd = (-xx_j - (-2 inner_ij)) - xx_i, candidates -> top-K
            const float *nx = nxl + (kt & 1) * 128 + 4 * (lane >> 5);
#pragma unroll
            for (int a = 0; a < 4; a++) {
                unsigned mask = 0;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float4 n4 = *(const float4 *)(nx + a * 32 + g * 8);
                    const float nv[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = g * 4 + e;
                        const float pd = fmaf(2.0f, acc[a][r], nv[e]) - xxq;
                        scr[r * 64 + lane] = pd;
                        mask |= (pd > thr && pd >= thrp) ? (1u << r) : 0u;
                        acc[a][r] = 0.f;
                    }
                }
                // one candidate per trip, the NEXT one's LDS read issued before the network runs.  A lane that
                // has run out of candidates reads row 16 of its scratch column, which holds -inf (a no-op
                // insertion): no per-lane boolean is carried around the loop -- a divergent i1 phi here made the
                // compiler add a flow block and ~120 register copies per trip.
                unsigned bp = min((unsigned)(__ffs((int)mask) - 1), 16u);
                mask &= mask - 1;
                float cv = scr[bp * 64 + lane];
                const int rbase = kt * 128 + a * 32 + 4 * (lane >> 5);
                // hand-rotated (if + do-while): ballot is convergent, so the compiler may not rotate a while
                // loop itself, and the unrotated form carries every TopK register through two extra blocks
#if FK_PROBE == 1
                if (false) {
#else
                if (__builtin_amdgcn_ballot_w64(bp < 16u) != 0) {
#endif
#pragma unroll 1
                    do {
#ifdef FK_COUNT
                        trips++;
#endif
                        const unsigned bn = min((unsigned)(__ffs((int)mask) - 1), 16u);
                        mask &= mask - 1;
                        const float cn = scr[bn * 64 + lane];
                        if constexpr (K == 20) topk20_insert(top, cv, rbase + (int)((bp & 3) + 8 * (bp >> 2)));   // asm network (common.h)
                        else top.insert(cv, rbase + (int)((bp & 3) + 8 * (bp >> 2)));
                        bp = bn;
                        cv = cn;
                    } while (__builtin_amdgcn_ballot_w64(bp < 16u) != 0);
                }
                // a candidate has to beat this lane's k-th best AND (at least tie) the partner lane's, which
                // ranks the other 64 keys of each tile for the same query.  (The k-th best of the pair's UNION,
                // max_i min(a[i-1], b[K-1-i]), cuts the insertions by a quarter but costs more than it saves.)
                thr = top.worst();
                thrp = __shfl_xor(thr, 32, 64);
            }
        }
        FK_QSWAP();
        __syncthreads();
        ch++;
        if (ch == nch) { ch = 0; kt++; }
        chn++;
        if (chn == nch) { chn = 0; ktn++; }
    }
#undef FK_LOAD
#undef FK_STORE
#undef FK_QSWAP

#ifdef FK_COUNT
    if (lane == 0) atomicAdd(&fk_trip_counter, (unsigned long long)trips);
#endif
    // ---- merge the two half-lists of each lane pair (the key buffers are free after the last barrier)
    float *mv = (float *)lds + wave * (2 * K * 32);
    int *mi = (int *)mv + K * 32;
    if (lane >= 32) {
#pragma unroll
        for (int i = 0; i < K; i++) { mv[i * 32 + lane - 32] = top.v[i]; mi[i * 32 + lane - 32] = top.id[i]; }
    }
    __syncthreads();
    if (lane < 32) {
#pragma unroll 1
        for (int i = 0; i < K; i++) fk_insert_lex<K>(top, mv[i * 32 + lane], mi[i * 32 + lane]);
        if (qrow < N) {
            if (gridDim.z == 1) {
                int64_t *dst = idx_out + ((size_t)b * N + qrow) * k;
#pragma unroll
                for (int i = 0; i < K; i++)
                    if (i < k) dst[i] = top.id[i];
            } else {
                const size_t o = (((size_t)b * N + qrow) * gridDim.z + blockIdx.z) * K;
#pragma unroll
                for (int i = 0; i < K; i++) { part_v[o + i] = top.v[i]; part_i[o + i] = top.id[i]; }
            }
        }
    }
}

// idx[q][0..k) = the k best of the parts' sorted lists (value descending, equal values: lower index first -- the order inside a list
// and what one list over all keys would hold); one thread per query, <= 4 list heads
template <int K>
__global__ __launch_bounds__(256) void featknn_merge_kernel(const float *__restrict__ part_v, const int *__restrict__ part_i, long nq, int parts,
                                                            int k, int64_t *__restrict__ idx_out)
{
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const float *v = part_v + (size_t)q * parts * K;
    const int *id = part_i + (size_t)q * parts * K;
    int head[4] = {0, 0, 0, 0};
    for (int o = 0; o < k; o++) {
        int best = -1;
        float bv = 0.f;
        int bi = 0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (p >= parts || head[p] >= K) continue;
            const float pv = v[p * K + head[p]];
            const int pi = id[p * K + head[p]];
            if (best < 0 || pv > bv || (pv == bv && pi < bi)) { best = p; bv = pv; bi = pi; }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) head[p] += (p == best);
        idx_out[(size_t)q * k + o] = bi;
    }
}

// key-range parts: enough workgroups for two per CU (512), at most 4, at most one per key tile
// (measured at k = 20, profiles/round5_featknn_bench.txt: B 32, N 1024: C 64 128 -> 134 us (the per-part epilogues cost more than the
// overlap returns: no split), C 128 168 -> 163, C 256 244 -> 220; B 8, N 1024, k 40: 477 -> 381)
static inline int fk_parts(int B, int Cp, int Np)
{
    const long wgs = (long)B * (Np / 128);
    if (Cp < 128 && wgs > 128) return 1;
    int p = wgs >= 512 ? 1 : (int)((512 + wgs - 1) / wgs);
    if (p > 4) p = 4;
    if (p > Np / 128) p = Np / 128;
    return p < 1 ? 1 : p;
}

extern "C" size_t l3d_knn_feature_workspace_bytes(int B, int C, int N)
{
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    const size_t Np = (size_t)l3d_divup(N, 128) * 128, Cp = (size_t)l3d_divup(C, 32) * 32;
    const int parts = fk_parts(B, (int)Cp, (int)Np);
    // split planes | -|x|^2 | (parts > 1) the parts' sorted lists, values and indices, at the longest list length (64)
    return (size_t)B * Cp * Np * 6 + (size_t)B * Np * 4 + (parts > 1 ? (size_t)B * N * parts * 64 * 8 : 0);
}

extern "C" int l3d_knn_feature(const float *x, int B, int C, int N, int k, void *workspace, int64_t *idx,
                               l3d_stream_t stream)
{
    L3D_REQUIRE(x && workspace && idx && B > 0 && C > 0 && N > 0 && k > 0);
    if (k > N) return L3D_ERR_INVALID_ARG;
    if (k > 64 || B > 65535 || (((size_t)workspace) & 15)) return L3D_ERR_UNSUPPORTED;
    const int Np = l3d_divup(N, 128) * 128, Cp = l3d_divup(C, 32) * 32;      // any C: padded with zero channels
    uint4 *xs = (uint4 *)workspace;
    float *nxx = (float *)((unsigned char *)workspace + (size_t)B * Cp * Np * 6);
    const int parts = fk_parts(B, Cp, Np);
    float *pv = nxx + (size_t)B * Np;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(featknn_split_kernel, dim3(l3d_divup(Np, 256), B), dim3(256), 0, st, x, C, Cp, N, Np, xs, nxx);
    dim3 grid(Np / 128, B, parts), block(256);
    const long nq = (long)B * N;
    const dim3 mgrid((unsigned)((nq + 255) / 256));
    // k <= 20: the asm insertion network; 20 < k <= 64: the generic one (list lengths 32 / 64)
#define FK_GO(KK)                                                                                                                       \
    do {                                                                                                                                \
        int *pi = (int *)(pv + (size_t)nq * parts * KK);                                                                                \
        hipLaunchKernelGGL(featknn_kernel<KK>, grid, block, FK_LDS, st, (const uint4 *)xs, (const float *)nxx, Cp, N, Np, k, idx, pv, pi); \
        if (parts > 1) hipLaunchKernelGGL(featknn_merge_kernel<KK>, mgrid, dim3(256), 0, st, (const float *)pv, (const int *)pi, nq, parts, k, idx); \
    } while (0)
    if (k <= 20) FK_GO(20);
    else if (k <= 32) FK_GO(32);
    else FK_GO(64);
#undef FK_GO
    return l3d_check_launch();
}
