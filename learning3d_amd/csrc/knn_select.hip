// knn_select.hip -- k nearest of a query among up to 8192 candidates for LARGE k (33 … 200), direct metric.
//
// Replaces, for k > 32:
//   K13 knn_kernel_fast   utils/lib/src/interpolate_gpu.cu:9-57   (FlowEmbedding's nsample = 64, flownet3d.py:93-123)
//   T8  knn_point()       utils/model_common_utils.py:84-100
//
// knn.hip keeps a query per LANE with its sorted best-K list in registers; at K = 64 that is 64+ VALU operations per
// insertion, run whenever ANY of the 64 lanes has a hit, on one wave per SIMD (268 registers): 1.05 ms for 32 × 1024
// queries against 8192 candidates.  Here a query belongs to a WAVE:
//
//   * the cloud's candidates sit in LDS once per workgroup as three coordinate arrays; lane l reads candidates
//     256 g + 4 l … + 3 of row group g with one ds_read_b128 per coordinate (conflict free) and keeps all R = M / 64
//     squared distances of the query in registers — computed with the reference's rounding sequence
//     ((dx·dx + dy·dy) + dz·dz, no contraction), compared as unsigned integers (non-negative floats order like their bits);
//   * an upper bound T0 of the k-th smallest distance comes from 256 BUCKET MINIMA (bucket = candidate index mod 256, so a
//     spatially sorted cloud still spreads over all buckets): k buckets whose minimum is <= T0 are k distinct candidates
//     <= T0.  T0 is found by bisection on the bit pattern with wave-wide ballot counts (4 compares per step) and stops as
//     soon as the count lies in [k, k + 8]: about 1.2 k candidates of 8192 survive;
//   * survivors (d <= T0) are compacted into a per-wave LDS list with ballot prefix positions and ranked by counting:
//     rank = number of survivors with a smaller (distance, index) key; rank < k writes slot `rank` — ascending distance,
//     lowest index first on ties, the reference's order;
//   * if more than KS_CAP candidates survive (heavy duplication), the exact k-th key is found by two bisections over the
//     registers (distance bits, then index among the ties), which leaves exactly k survivors.
#include "common.h"

#define KS_WAVES 8
#define KS_CAP 512
enum { KS_OUT_PAIR = 1, KS_OUT_POINT = 2 };          // == OUT_KNN_PAIR / OUT_KNN_POINT of knn.hip

typedef float ks_f4 __attribute__((ext_vector_type(4)));

template <int R>
__global__ __launch_bounds__(64 * KS_WAVES) void knn_select_kernel(
    const float *__restrict__ qxyz, const float *__restrict__ cxyz, int Nq, int Nc, int k, int qpw, int out_mode,
    void *__restrict__ idx_out, float *__restrict__ val_out)
{
    constexpr int MP = R * 64;
    __shared__ __attribute__((aligned(16))) float sc[3][MP];
    __shared__ __attribute__((aligned(16))) uint2 surv[KS_WAVES][KS_CAP + 4];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const float *cbase = cxyz + (size_t)b * Nc * 3;
    for (int e = threadIdx.x; e < MP * 3; e += 64 * KS_WAVES) {
        const int j = e / 3, c = e - 3 * j;
        sc[c][j] = j < Nc ? cbase[e] : INFINITY;             // padding: distance +inf, ranks behind every real candidate
    }
    __syncthreads();

    uint2 *sv = surv[wave];
    const int q0 = (blockIdx.x * KS_WAVES + wave) * qpw;
#pragma unroll 1
    for (int qi = 0; qi < qpw; qi++) {
        const int q = q0 + qi;
        if (q >= Nq) break;
        const float *qp = qxyz + ((size_t)b * Nq + q) * 3;
        const float qx = qp[0], qy = qp[1], qz = qp[2];

        // ---- all distances of the query, in registers; bucket minima on the way
        unsigned d[R];
        unsigned bm[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        // (reads of group g + 1 are issued before group g's arithmetic; the scheduling fence keeps the compiler from
        // hoisting ALL the reads, which cost R = 128 its 256-register budget)
        ks_f4 Xn = *(const ks_f4 *)&sc[0][lane * 4], Yn = *(const ks_f4 *)&sc[1][lane * 4], Zn = *(const ks_f4 *)&sc[2][lane * 4];
#pragma unroll
        for (int g = 0; g < R / 4; g++) {
            const ks_f4 X = Xn, Y = Yn, Z = Zn;
            if (g + 1 < R / 4) {
                Xn = *(const ks_f4 *)&sc[0][(g + 1) * 256 + lane * 4];
                Yn = *(const ks_f4 *)&sc[1][(g + 1) * 256 + lane * 4];
                Zn = *(const ks_f4 *)&sc[2][(g + 1) * 256 + lane * 4];
            }
            __builtin_amdgcn_sched_barrier(0);
            const ks_f4 dx = qx - X, dy = qy - Y, dz = qz - Z;
            const ks_f4 dd = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                d[g * 4 + u] = __float_as_uint(dd[u]);
                bm[u] = min(bm[u], d[g * 4 + u]);
            }
        }
        // (opaque per query: otherwise all R candidate indices are hoisted out of the query loop into R registers)
        int l4 = lane * 4;
        asm volatile("" : "+v"(l4));
        auto cand_index = [&](int r) { return (r >> 2) * 256 + l4 + (r & 3); };

        // ---- T0: some value with  k <= #(bucket minima <= T0)  (<= k + 8 when the bisection gets there)
        unsigned T0;
        {
            unsigned lo = 0, hi = 0xffffffffu;
#pragma unroll 1
            while (lo < hi) {
                const unsigned mid = lo + ((hi - lo) >> 1);
                int c = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) c += __popcll(__ballot(bm[u] <= mid));
                if (c >= k) {
                    hi = mid;
                    if (c <= k + 8) break;
                } else {
                    lo = mid + 1;
                }
            }
            T0 = hi;
        }

        // ---- survivors -> LDS list.  pred(r): d < T || (d == T && index <= I)
        int S = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool p = d[r] <= T0;
            const unsigned long long m = __ballot(p);
            if (m != 0) {
                const int pos = S + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (p && pos < KS_CAP) sv[pos] = make_uint2(d[r], (unsigned)cand_index(r));
                S += __popcll(m);
            }
        }
        if (S > KS_CAP) {
            // exact k-th smallest key: distance bits T, then the index bound I among the candidates at distance T.
            // (the scheduling fences keep at most eight ballots alive; left alone, the scheduler gathered all R compares
            // first and spilled scalar registers into vector lanes)
            auto count = [&](auto pred) {
                int c = 0;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    c += __popcll(__ballot(pred(r)));
                    if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
                return c;
            };
            unsigned lo = 0, hi = 0xffffffffu;
#pragma unroll 1
            while (lo < hi) {
                const unsigned mid = lo + ((hi - lo) >> 1);
                if (count([&](int r) { return d[r] <= mid; }) >= k) hi = mid;
                else lo = mid + 1;
            }
            const unsigned T = hi;
            const int need = k - count([&](int r) { return d[r] < T; });        // >= 1 ties to take, lowest indices first
            int ilo = 0, ihi = MP - 1;
#pragma unroll 1
            while (ilo < ihi) {
                const int mid = ilo + ((ihi - ilo) >> 1);
                if (count([&](int r) { return d[r] == T && l4 <= mid - ((r >> 2) * 256 + (r & 3)); }) >= need) ihi = mid;
                else ilo = mid + 1;
            }
            S = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const bool p = d[r] < T || (d[r] == T && l4 <= ihi - ((r >> 2) * 256 + (r & 3)));
                const unsigned long long m = __ballot(p);
                if (m != 0) {
                    const int pos = S + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (p) sv[pos] = make_uint2(d[r], (unsigned)cand_index(r));      // exactly k entries
                    S += __popcll(m);
                }
            }
        }
        if (lane < 4) sv[S + lane] = make_uint2(0xffffffffu, 0xffffffffu);           // the rank loop reads in fours
        __builtin_amdgcn_wave_barrier();

        // ---- rank by counting; two of the lane's entries share every broadcast read
        const size_t o = ((size_t)b * Nq + q) * k;
#pragma unroll 1
        for (int t0 = 0; t0 < S; t0 += 128) {
            const int e0 = t0 + lane, e1 = t0 + 64 + lane;
            const uint2 m0 = sv[min(e0, S)], m1 = sv[min(e1, S)];                    // entry S is a sentinel
            const unsigned long long k0 = ((unsigned long long)m0.x << 32) | m0.y;
            const unsigned long long k1 = ((unsigned long long)m1.x << 32) | m1.y;
            int r0 = 0, r1 = 0;
#pragma unroll 1
            for (int s = 0; s < S; s += 4) {
                const uint4 a = *(const uint4 *)&sv[s], c = *(const uint4 *)&sv[s + 2];
                const unsigned long long o0 = ((unsigned long long)a.x << 32) | a.y, o1 = ((unsigned long long)a.z << 32) | a.w;
                const unsigned long long o2 = ((unsigned long long)c.x << 32) | c.y, o3 = ((unsigned long long)c.z << 32) | c.w;
                r0 += (o0 < k0) + (o1 < k0) + (o2 < k0) + (o3 < k0);
                r1 += (o0 < k1) + (o1 < k1) + (o2 < k1) + (o3 < k1);
            }
            if (out_mode == KS_OUT_PAIR) {
                int32_t *dst = (int32_t *)idx_out + o;
                if (e0 < S && r0 < k) { dst[r0] = (int32_t)m0.y; val_out[o + r0] = __uint_as_float(m0.x); }
                if (e1 < S && r1 < k) { dst[r1] = (int32_t)m1.y; val_out[o + r1] = __uint_as_float(m1.x); }
            } else {
                int64_t *dst = (int64_t *)idx_out + o;
                if (e0 < S && r0 < k) { dst[r0] = (int64_t)m0.y; val_out[o + r0] = sqrtf(__uint_as_float(m0.x)); }
                if (e1 < S && r1 < k) { dst[r1] = (int64_t)m1.y; val_out[o + r1] = sqrtf(__uint_as_float(m1.x)); }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

bool l3d_knn_select_supported(int Nc, int k) { return k > 32 && k <= L3D_KNN_MAX_K && k <= Nc && Nc <= 8192; }

int l3d_launch_knn_select(const float *q, const float *c, int B, int Nq, int Nc, int k, int out_mode, void *idx, float *val,
                          hipStream_t st)
{
    if (!l3d_knn_select_supported(Nc, k)) return L3D_ERR_UNSUPPORTED;
    const long total = (long)B * Nq;
    int qpw = (int)((total + 256L * KS_WAVES - 1) / (256L * KS_WAVES));              // about one workgroup per CU
    if (qpw < 1) qpw = 1;
    dim3 grid(l3d_divup(Nq, qpw * KS_WAVES), B), block(64 * KS_WAVES);
#define KS_CASE(RR)                                                                                                      \
    if (Nc <= RR * 64) {                                                                                                 \
        hipLaunchKernelGGL((knn_select_kernel<RR>), grid, block, 0, st, q, c, Nq, Nc, k, qpw, out_mode, idx, val);       \
        return l3d_check_launch();                                                                                       \
    }
    KS_CASE(4)
    KS_CASE(8)
    KS_CASE(16)
    KS_CASE(32)
    KS_CASE(64)
    KS_CASE(128)
#undef KS_CASE
    return L3D_ERR_UNSUPPORTED;
}
