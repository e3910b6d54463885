// knn_mfma.hip -- k nearest neighbours of a cloud in itself (utils/model_common_utils.py:3-9, `knn`), ranking values
// on the fp32 matrix cores, selection by thresholding + rank counting.  k <= 24, 256 <= N <= 2048 (the DGCNN / DCP
// shapes: N = 1024, k = 20); everything else stays on knn.hip's two-pass insertion kernel.
//
// The reference ranks pd[i][j] = (-xx_j + 2 <x_i, x_j>) - xx_i in fp32 with the dot product as the fma chain
// fma(z,z', fma(y,y', x*x')) (MKL sgemm, K = 3).  v_mfma_f32_32x32x2_f32 accumulates its two k-slots as exactly
// that kind of chain, acc' = fma(a1,b1, fma(a0,b0, acc)) (tools/probe_mfma_dot.hip: 0 of 2 M dots differ), so three
// issues with the k-slots
//        candidate side (A, rows)   cx    cy  |  cz    -xx_j |  1      0
//        query side     (B, cols)   2qx   2qy |  2qz    1    | -xx_i   0
// give   acc = 2 dot (scaling by 2 commutes with every rounding) -> rn(2 dot - xx_j) -> rn(that - xx_i) = pd, bit for
// bit what knn.hip's eval() computes with 5 VALU instructions per pair -- here it costs none: the matrix pipe is
// otherwise idle and a lane receives 16 ranking values of ONE query per tile (column = query, rows = candidates).
//
// Work split: a workgroup = 32 queries x the whole cloud, 4 waves.  Candidate c goes to "lane" c % 8 of the query
// (wave (c%8)/2, accumulator half (c%8)%2) and slot c / 8 of that lane (tile slot/16, accumulator register slot%16):
// consecutive indices rotate over the 8 lanes that serve one query, so a spatially ordered cloud still spreads a
// query's neighbourhood evenly, and within a lane slots ascend with the index.
//
//   pass 0  per lane, the maximum over tiles for each of its 16 accumulator registers (v_max3_f32, half an
//           instruction per pair); the 3 largest of those 16 group maxima are 3 distinct candidates, so the k-th
//           largest of the 8 x 3 = 24 values a query's lanes hold is a lower bound thr0 of its k-th best ranking
//           value that >= k candidates reach -- and a tight one: ~27 candidates reach it on average (max ~50).
//   pass 1  recompute the tiles (bit-identical), one v_sub + v_alignbit per value builds a 32-bit mask of the
//           candidates >= thr0 of two tiles; the rare hits are popped, turned into 64-bit keys
//           (order-preserving value bits << 32 | ~index: distinct, larger = nearer, lower index first under exact
//           ties) and appended to the query's list in LDS through an LDS atomic counter.
//   rank    each of the query's 8 lanes takes every 8th key of the list and counts the keys above it; a key of rank
//           r < k writes its index to idx[q][r].  No sorting network, no per-lane sorted lists.
// Lists hold 64 keys.  When a list overflows -- the tail of the bound (one query in 30 000 of a random cloud), or a point
// duplicated dozens of times -- the block ranks what it did collect, takes the k-th best of those keys as the query's new
// threshold (real candidates reach it, so it is a valid, tighter bound) and collects again.  If the bound cannot rise
// (>= 45 of the 64 collected keys sit AT it: exact ties) the query switches to collecting strictly above it; what is then
// missing from its k are the lowest-index candidates AT the bound, which a last phase picks one per round.  Every step
// either raises a threshold past at least one value level or ends, so the loop terminates; there is no second kernel.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

#define KM_MCAP 64                         // keys per query list
#define KM_STRIDE 65                       // u64 row stride of the lists (bank spread)
#define KM_WPAD 16                         // floats between the waves' candidate blocks (see the kernel)
#define KM_T3S 25                          // float row stride of the 24 per-query bound values
#define KM_PAD (-3.0e38f)                  // -xx of a padding candidate: its ranking value stays finite and below any real one

__device__ __forceinline__ f32x16 km_tile(const float *__restrict__ cxy, const float *__restrict__ czw, int at,
                                          float b1, float b2, float b3, float a3)
{
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cxy[at], b1, acc, 0, 0, 0);     // k = (cx * 2qx, cy * 2qy)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(czw[at], b2, acc, 0, 0, 0);     // k = (cz * 2qz, -xx_j * 1)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc, 0, 0, 0);          // k = (1 * -xx_i, 0 * 0)
    return acc;
}

// the next float above x: value > x  <=>  value >= km_nextup(x)
__device__ __forceinline__ float km_nextup(float x)
{
    const unsigned b = __float_as_uint(x + 0.0f);
    return __uint_as_float((int)b >= 0 ? b + 1u : b - 1u);
}

#ifdef KM_TIMING       // tools/probe_knn_mfma.hip: s_memtime stamps per wave instead of results
#define KMT(n) if (lane == 0) ((long long *)idx_out)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 16 + (n)] = __builtin_amdgcn_s_memtime();
#else
#define KMT(n)
#endif

// CT > 0: the cloud has exactly CT tiles per wave and pass 0 keeps them in registers for pass 1 (16 CT VGPRs)
template <int CT>
__global__ __launch_bounds__(256, 2) void knn_mfma_kernel(const float *__restrict__ xyz, int N, int k, int T_,
                                                          int64_t *__restrict__ idx_out)
{
    const int T = CT > 0 ? CT : T_;
    extern __shared__ __attribute__((aligned(16))) unsigned char km_smem[];
    // candidate slots (>= N), [wave][tile][row]; every wave's block is followed by KM_WPAD floats: the staging scatter sends the
    // four lanes that differ only in (c & 7) >> 1 to the same offset of four DIFFERENT blocks, and with a block of T x 32
    // floats (a multiple of the 64 banks) they all hit one bank -- the 24.5 % LDS conflict rate of round 2's PMC pass
    const int NS = 4 * T * 32, WS = T * 32 + KM_WPAD, P = 4 * WS;
    float *cx = (float *)km_smem;                              // cx | cy | cz | cw, P floats each
    float *cz = cx + 2 * P;
    float *dump = cx + 4 * P;                                  // [4 waves][2 tiles][4 quads][64 lanes][4]
    u64 *qlist = (u64 *)(dump + 4 * 2048);                     // [32][KM_STRIDE]
    float *t3 = (float *)(qlist + 32 * KM_STRIDE);             // [32][KM_T3S]
    float *thr0s = t3 + 32 * KM_T3S;                           // [32][9]: the 8 lanes' offers per query
    int *qcnt = (int *)(thr0s + 32 * 9);                       // [32]
    int *ovf = qcnt + 32;                                      // [1]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5, lane8 = 2 * wave + h;
    const int b = blockIdx.y, q0 = blockIdx.x * 32;
    const float *cloud = xyz + (size_t)b * N * 3;
    KMT(7)

    // ---- stage the cloud in slot order (coalesced reads issued four deep, scattered LDS writes), padding included
    for (int c0 = tid; c0 < NS; c0 += 1024) {
        float x[4], y[4], z[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            // unconditional loads from a clamped index (`if (c < N) load` is an exec-masked branch with a wait at its join, and the
            // four loads then leave one round trip behind the other); padding slots are zeroed by a select further down
            const int cc = min(c0 + 256 * u, N - 1);
            x[u] = cloud[cc * 3 + 0];
            y[u] = cloud[cc * 3 + 1];
            z[u] = cloud[cc * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (c0 + 256 * u >= N) { x[u] = 0.f; y[u] = 0.f; z[u] = 0.f; }      // v_cndmask on the loaded values (Inf * 0 would be NaN)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int c = c0 + 256 * u;
            if (c < NS) {
                const float w = c < N ? -((x[u] * x[u] + y[u] * y[u]) + z[u] * z[u]) : KM_PAD;
                const int l8 = c & 7, s = c >> 3, j = s >> 4, r = s & 15;
                const int p = (l8 >> 1) * WS + j * 32 + ((r >> 2) << 3) + ((l8 & 1) << 2) + (r & 3);
                cx[p] = x[u];
                cx[P + p] = y[u];
                cz[p] = z[u];
                cz[P + p] = w;
            }
        }
    }
    for (int e = tid; e < 32 * KM_STRIDE; e += 256) qlist[e] = 0ull;
    if (tid < 32) qcnt[tid] = 0;
    if (tid == 0) *ovf = 0;

    const int q = q0 + i;
    const bool valid = q < N;
    const float *qp = cloud + (size_t)(valid ? q : N - 1) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float qxx = (qx * qx + qy * qy) + qz * qz;
    const float b1 = h ? 2.0f * qy : 2.0f * qx;
    const float b2 = h ? 1.0f : 2.0f * qz;
    const float b3 = h ? 0.0f : -qxx;
    const float a3 = h ? 0.0f : 1.0f;
    const float *cxy = cx + h * P;                             // lane (i, h) supplies k-slot h of row i
    const float *czw = cz + h * P;
    const int at0 = wave * WS + i;
    KMT(0)
    __syncthreads();
    KMT(1)

    // ------------------------------------------------------------------ pass 0: group maxima -> thr0
    float mr[16];
#pragma unroll
    for (int r = 0; r < 16; r++) mr[r] = -INFINITY;
    f32x16 kept[CT > 0 ? CT : 1];
    int j = 0;
    if constexpr (CT > 0) {
        static_assert(CT % 2 == 0, "tiles are consumed in pairs");
#pragma unroll
        for (int t = 0; t < CT; t++) kept[t] = km_tile(cxy, czw, at0 + t * 32, b1, b2, b3, a3);
#pragma unroll
        for (int t = 0; t < CT; t += 2)
#pragma unroll
            for (int r = 0; r < 16; r++) mr[r] = fmaxf(fmaxf(mr[r], kept[t][r]), kept[t + 1][r]);
    } else {
        for (; j + 2 <= T; j += 2) {
            const f32x16 aA = km_tile(cxy, czw, at0 + j * 32, b1, b2, b3, a3);
            const f32x16 aB = km_tile(cxy, czw, at0 + j * 32 + 32, b1, b2, b3, a3);
#pragma unroll
            for (int r = 0; r < 16; r++) mr[r] = fmaxf(fmaxf(mr[r], aA[r]), aB[r]);
        }
        if (j < T) {
            const f32x16 aA = km_tile(cxy, czw, at0 + j * 32, b1, b2, b3, a3);
#pragma unroll
            for (int r = 0; r < 16; r++) mr[r] = fmaxf(mr[r], aA[r]);
        }
    }
    KMT(2)
    float m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;     // the lane's 3 largest group maxima, m1 >= m2 >= m3
#pragma unroll
    for (int r = 0; r < 16; r++) {                             // sorted insert: one v_med3 per slot (common.h, TopKV)
        const float v = mr[r];
        m3 = __builtin_amdgcn_fmed3f(m2, m3, v);
        m2 = __builtin_amdgcn_fmed3f(m1, m2, v);
        m1 = fmaxf(m1, v);
    }
    t3[i * KM_T3S + lane8 * 3 + 0] = m1;
    t3[i * KM_T3S + lane8 * 3 + 1] = m2;
    t3[i * KM_T3S + lane8 * 3 + 2] = m3;
    __syncthreads();
    {
        // the k-th largest of the 24 = the smallest value with fewer than k values above it: each lane counts for its own
        // three (m1 >= m2 >= m3, so the counts ascend) and offers the smallest that qualifies; the query's 8 lanes then
        // take the minimum of the offers
        int g1 = 0, g2 = 0, g3 = 0;
#pragma unroll 8
        for (int f = 0; f < 24; f++) {
            const float v = t3[i * KM_T3S + f];
            g1 += v > m1;
            g2 += v > m2;
            g3 += v > m3;
        }
        const float offer = g3 < k ? m3 : g2 < k ? m2 : g1 < k ? m1 : INFINITY;
        thr0s[i * 9 + lane8] = offer;
    }
    __syncthreads();
    float thr0_ = thr0s[i * 9];
#pragma unroll
    for (int l = 1; l < 8; l++) thr0_ = fminf(thr0_, thr0s[i * 9 + l]);
#ifdef KM_TIMING
    float thr0 = (k & 0x100) ? INFINITY : thr0_;             // probe: k + 256 = nothing reaches the threshold
    k &= 0xff;
#else
    float thr0 = thr0_;
#endif
    KMT(3)

    // ------------------------------------------------------------------ pass 1: collect the candidates >= thr (or > thr)
    float *mydump = dump + wave * 2048;
    u64 *mylist = qlist + i * KM_STRIDE;
    float thr = thr0;                  // this query's bound: a value that >= k of its candidates reach
    bool strict = false;               // collect the candidates strictly above thr (its ties are filled in at the end)
    for (int attempt = 0;; attempt++) {
        const float thrc = strict ? km_nextup(thr) : thr;
        bool over = false;
    #pragma unroll
        for (j = 0; j < T; j += 2) {
            const bool two = j + 1 < T;
            f32x16 aA, aB;
            if constexpr (CT > 0) {
                aA = kept[j];
                aB = kept[j + 1];
            } else {
                aA = km_tile(cxy, czw, at0 + j * 32, b1, b2, b3, a3);
                aB = km_tile(cxy, czw, at0 + (two ? j * 32 + 32 : j * 32), b1, b2, b3, a3);
            }
            unsigned sgn = 0;                                      // sign bits of (value - thrc): set = below the threshold
    #pragma unroll
            for (int r = 0; r < 16; r++) sgn = __builtin_amdgcn_alignbit(sgn, __float_as_uint(aA[r] - thrc), 31);
    #pragma unroll
            for (int r = 0; r < 16; r++) sgn = __builtin_amdgcn_alignbit(sgn, __float_as_uint(aB[r] - thrc), 31);
            unsigned m = ~sgn;                                     // bit 31 - e: accumulator register e & 15 of tile e >> 4 (A, B)
            if (!two) m &= 0xFFFF0000u;
            if (__builtin_amdgcn_ballot_w64(m != 0) != 0) {
    #pragma unroll
                for (int qd = 0; qd < 4; qd++) {
                    *(float4 *)(mydump + (qd * 64 + lane) * 4) = make_float4(aA[4 * qd], aA[4 * qd + 1], aA[4 * qd + 2], aA[4 * qd + 3]);
                    *(float4 *)(mydump + 1024 + (qd * 64 + lane) * 4) = make_float4(aB[4 * qd], aB[4 * qd + 1], aB[4 * qd + 2], aB[4 * qd + 3]);
                }
                // Four hits per trip, straight-line: an exec-mask region costs a scalar round trip (LABLOG 4.3b), so a lane
                // without a hit goes through the same motions with an increment of 0 and a write to the spare slot of its
                // row.  The dump reads and slot atomics of a trip are in flight together.
                const int cbase = 128 * j + lane8;
    #pragma unroll 1
                do {
                    int e[4], slot[4];
                    bool has[4];
                    float vv[4];
    #pragma unroll
                    for (int u = 0; u < 4; u++) {
                        has[u] = m != 0;
                        e[u] = has[u] ? __builtin_clz(m) : 31;
                        m &= ~(0x80000000u >> e[u]);
                        vv[u] = mydump[((e[u] >> 2) << 8) + lane * 4 + (e[u] & 3)];
                        slot[u] = atomicAdd(&qcnt[i], has[u] ? 1 : 0);
                    }
    #pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const unsigned bits = __float_as_uint(vv[u] + 0.0f);                         // -0 -> +0
                        const unsigned sk = bits ^ ((unsigned)((int)bits >> 31) | 0x80000000u);       // unsigned order == float order
                        const u64 key = ((u64)sk << 32) | (unsigned)(~(cbase + 8 * e[u]));            // candidate 128 (j + tile) + 8 r + lane8
                        const bool fits = slot[u] < KM_MCAP;
                        over |= has[u] && !fits;
                        mylist[has[u] && fits ? slot[u] : KM_MCAP] = key;                             // row slot 64: scratch
                    }
                } while (__builtin_amdgcn_ballot_w64(m != 0) != 0);
            }
        }
        if (over) *ovf = 1;
        KMT(4)
        __syncthreads();
        KMT(5)
        if (*ovf == 0) break;
        if (attempt == 64) {                                   // tripwire: every pass raises a threshold or ends (see the header)
            if (tid == 0) idx_out[((size_t)b * N + q0) * k] = -1;
            return;
        }
        // Some list of the block overflowed.  For EVERY query: the k-th best of the keys it did collect (any subset of its
        // candidates >= thr) is a bound that >= k candidates reach and that is >= thr.
        if (lane8 == 0) thr0s[i * 9 + 8] = thr;                // default: fewer than k keys collected (strict mode)
        __syncthreads();
        {
            u64 own[8];
            int rank[8];
#pragma unroll
            for (int o = 0; o < 8; o++) {
                own[o] = mylist[lane8 + 8 * o];
                rank[o] = 0;
            }
            for (int f = 0; f < KM_MCAP; f++) {
                const u64 kf = mylist[f];
#pragma unroll
                for (int o = 0; o < 8; o++) rank[o] += kf > own[o] ? 1 : 0;
            }
#pragma unroll
            for (int o = 0; o < 8; o++)
                if (own[o] != 0ull && rank[o] == k - 1) {      // at most one key per query: keys are distinct
                    const unsigned sk = (unsigned)(own[o] >> 32);
                    thr0s[i * 9 + 8] = __uint_as_float((sk & 0x80000000u) ? sk ^ 0x80000000u : ~sk);
                }
        }
        __syncthreads();
        {
            const float thr_new = thr0s[i * 9 + 8];
            if (thr_new > thr) {                               // a tighter bound: collect >= it
                thr = thr_new;
                strict = false;
            } else {
                strict = true;                                 // the bound cannot rise (ties at it): collect what lies above
            }
        }
        __syncthreads();                                       // everybody has read the lists and the new bounds
        for (int e = tid; e < 32 * KM_STRIDE; e += 256) qlist[e] = 0ull;
        if (tid < 32) qcnt[tid] = 0;
        if (tid == 0) *ovf = 0;
        __syncthreads();
    }

    // ------------------------------------------------------------------ rank: count the keys above each key
    int Mi = qcnt[i];
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) Mi = max(Mi, __shfl_xor(Mi, d, 64));
    const int Mmax = __builtin_amdgcn_readfirstlane(Mi);
    // a lane ranks every 8th key of its query's list: NO = ceil(longest list of the wave / 8) keys per lane
    u64 own[8];
    int rank[8];
#pragma unroll
    for (int o = 0; o < 8; o++) {
        own[o] = mylist[lane8 + 8 * o];                        // 0 past the end of the list
        rank[o] = 0;
    }
    if (Mmax <= 32) {
#pragma unroll 2
        for (int f = 0; f < Mmax; f++) {
            const u64 kf = mylist[f];
#pragma unroll
            for (int o = 0; o < 4; o++) rank[o] += kf > own[o] ? 1 : 0;
        }
    } else if (Mmax <= 48) {
#pragma unroll 2
        for (int f = 0; f < Mmax; f++) {
            const u64 kf = mylist[f];
#pragma unroll
            for (int o = 0; o < 6; o++) rank[o] += kf > own[o] ? 1 : 0;
        }
    } else {
#pragma unroll 2
        for (int f = 0; f < Mmax; f++) {
            const u64 kf = mylist[f];
#pragma unroll
            for (int o = 0; o < 8; o++) rank[o] += kf > own[o] ? 1 : 0;
        }
    }
    KMT(6)
#ifdef KM_TIMING
    if (lane == 0) ((long long *)idx_out)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 16 + 8] =
        rank[0] + rank[1] + rank[2] + rank[3] + rank[4] + rank[5] + rank[6] + rank[7];
    return;
#endif
    int64_t *dst = idx_out + ((size_t)b * N + (valid ? q : 0)) * k;
    if (valid) {
#pragma unroll
        for (int o = 0; o < 8; o++)
            if (own[o] != 0ull && rank[o] < k) dst[rank[o]] = (int64_t)(unsigned)(~(unsigned)own[o]);
    }

    // ------------------------------------------------------------------ ties at the bound (strict mode, fewer than k above it)
    // ranks Mq .. k-1 go to the lowest-index candidates whose value EQUALS thr, one per round: every lane offers its lowest
    // such slot beyond the one it gave last, the query's 8 lanes take the minimum.  Only blocks with heavily duplicated
    // points ever get here.
    const int Mq = qcnt[i];
    const int need = (strict && Mq < k) ? k - Mq : 0;
    __syncthreads();                                           // all ranking reads of qcnt / lists are done
    if (tid == 0) *ovf = 0;
    __syncthreads();
    if (need > 0) atomicMax(ovf, need);
    __syncthreads();
    const int rounds = *ovf;
    int last = -1;
    int *props = (int *)thr0s;
    for (int round = 0; round < rounds; round++) {
        int prop = 0x7fffffff;
        for (int t = T - 1; t >= 0; t--) {                     // descending: the lowest slot is the one that sticks
            f32x16 av;
            if constexpr (CT > 0) {
                av = kept[0];
#pragma unroll
                for (int u = 1; u < CT; u++) av = t == u ? kept[u] : av;
            } else {
                av = km_tile(cxy, czw, at0 + t * 32, b1, b2, b3, a3);
            }
#pragma unroll
            for (int r = 15; r >= 0; r--) {
                const int sl = 16 * t + r;
                prop = (av[r] == thr && sl > last) ? sl : prop;
            }
        }
        const int c = prop != 0x7fffffff ? 8 * prop + lane8 : 0x7fffffff;
        props[i * 9 + lane8] = c;
        __syncthreads();
        int cmin = props[i * 9];
#pragma unroll
        for (int l = 1; l < 8; l++) cmin = min(cmin, props[i * 9 + l]);
        if (c == cmin && c != 0x7fffffff) {
            last = prop;
            if (round < need && valid) dst[Mq + round] = c;
        }
        __syncthreads();
    }
}

size_t l3d_knn_mfma_lds_bytes(int N)
{
    const int T = l3d_divup(N, 128);
    return (size_t)4 * (4 * (T * 32 + KM_WPAD)) * 4 + 4 * 2048 * 4 + 32 * KM_STRIDE * 8 + 32 * KM_T3S * 4 + 32 * 9 * 4 + 32 * 4 + 16;
}

bool l3d_knn_mfma_supported(int N, int k) { return k <= 24 && N >= 256 && N <= 2048; }

int l3d_launch_knn_mfma(const float *xyz, int B, int N, int k, int64_t *idx, hipStream_t st)
{
    const int T = l3d_divup(N, 128);
    const dim3 grid(l3d_divup(N, 32), B);
    if (T == 8)                                                // 896 < N <= 1024: pass-0 tiles stay in registers
        hipLaunchKernelGGL(knn_mfma_kernel<8>, grid, dim3(256), l3d_knn_mfma_lds_bytes(N), st, xyz, N, k, T, idx);
    else
        hipLaunchKernelGGL(knn_mfma_kernel<0>, grid, dim3(256), l3d_knn_mfma_lds_bytes(N), st, xyz, N, k, T, idx);
    return l3d_check_launch();
}
