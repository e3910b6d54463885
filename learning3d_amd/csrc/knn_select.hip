// knn_select.hip -- k nearest of a query among up to 8192 candidates, a WAVE per query, direct metric (k <= 200).
//
// Replaces, for k > 32 or 1024+ candidates (k <= 4 with very many queries goes to knn_small.hip):
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
//     soon as the count lies in [k, k + 8]: about 1.2 x k candidates survive (79 of 8192 at k = 64);
//   * survivors (d <= T0) are marked per lane (v_sub + v_alignbit), popped four per trip, recomputed from LDS, appended to a
//     per-wave LDS list at ballot prefix positions and ranked by counting:
//     rank = number of survivors with a smaller (distance, index) key; rank < k writes slot `rank` — ascending distance,
//     lowest index first on ties, the reference's order;
//   * if more than KS_CAP candidates survive (heavy duplication), the exact k-th key is found by two bisections over the
//     registers (distance bits, then index among the ties), which leaves exactly k survivors.
#include "common.h"

#define KS_WAVES 8
#define KS_CAP 512
enum { KS_OUT_PAIR = 1, KS_OUT_POINT = 2 };          // == OUT_KNN_PAIR / OUT_KNN_POINT of knn.hip

typedef float ks_f4 __attribute__((ext_vector_type(4)));

// tools/probe_knn_select.hip: shader-clock cycles per phase, summed over a wave's queries (compiled out otherwise)
#ifdef KS_TIMING
__device__ long long *g_ks_time;
#define KS_T(i) { const long long t_ = __builtin_amdgcn_s_memtime(); ks_acc[i] += t_ - ks_last; ks_last = t_; }
#else
#define KS_T(i)
#endif

template <int R>
__global__ __launch_bounds__(64 * KS_WAVES) void knn_select_kernel(
    const float *__restrict__ qxyz, const float *__restrict__ cxyz, int Nq, int Nc, int k, int qpw, int out_mode,
    void *__restrict__ idx_out, float *__restrict__ val_out)
{
    constexpr int MP = R * 64;
    __shared__ __attribute__((aligned(16))) float sc[3][MP];
    __shared__ __attribute__((aligned(16))) uint2 surv[KS_WAVES][KS_CAP + 8];

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
#ifdef KS_TIMING
    long long ks_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long ks_last = __builtin_amdgcn_s_memtime();
#endif
    const int q0 = (blockIdx.x * KS_WAVES + wave) * qpw;
#pragma unroll 1
    for (int qi = 0; qi < qpw; qi++) {
        const int q = q0 + qi;
        if (q >= Nq) break;
        const float *qp = qxyz + ((size_t)b * Nq + q) * 3;
        const float qx = qp[0], qy = qp[1], qz = qp[2];

        // ---- all distances of the query, in registers; bucket minima on the way
        // the query splat in VECTOR registers: with a scalar operand the compiler emits four v_sub_f32 instead of two v_pk_add_f32
        // (and as an ADDITION of -q: a subtraction is not packed either)
        ks_f4 q4x = {-qx, -qx, -qx, -qx}, q4y = {-qy, -qy, -qy, -qy}, q4z = {-qz, -qz, -qz, -qz};
        asm volatile("" : "+v"(q4x), "+v"(q4y), "+v"(q4z));
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
            const ks_f4 dx = X + q4x, dy = Y + q4y, dz = Z + q4z;     // c + (-q): (c - q)^2 == (q - c)^2 bit for bit
            const ks_f4 dd = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
            for (int u = 0; u < 4; u++) d[g * 4 + u] = __float_as_uint(dd[u]);
        }
        if constexpr (R >= 8) {
#pragma unroll
            for (int r = 0; r < R; r += 8)                             // v_min3_u32: two rows per operation
#pragma unroll
                for (int u = 0; u < 4; u++) bm[u] = min(bm[u], min(d[r + u], d[r + 4 + u]));
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) bm[u] = d[u];
        }
        // (opaque per query: otherwise all R candidate indices are hoisted out of the query loop into R registers)
        int l4 = lane * 4;
        asm volatile("" : "+v"(l4));
        auto cand_index = [&](int r) { return (r >> 2) * 256 + l4 + (r & 3); };

        KS_T(0)
        // ---- T0: some value with  k <= #(bucket minima <= T0)  (<= k + 8 when the bisection gets there)
        unsigned T0;
        {
            // the four compares of a step are issued back to back into four scalar pairs and counted afterwards: left to the
            // compiler each v_cmp -> s_bcnt1 pair went through vcc, one after the other (165 cycles per step)
            unsigned lo = 0, hi = 0x7fffffffu;
#pragma unroll 1
            while (lo < hi) {
                const unsigned mid = lo + ((hi - lo) >> 1);
                unsigned long long m0, m1, m2, m3;
                asm volatile("v_cmp_le_u32_e64 %0, %4, %8\n\tv_cmp_le_u32_e64 %1, %5, %8\n\t"
                             "v_cmp_le_u32_e64 %2, %6, %8\n\tv_cmp_le_u32_e64 %3, %7, %8"
                             : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
                             : "v"(bm[0]), "v"(bm[1]), "v"(bm[2]), "v"(bm[3]), "s"(mid));
                const int c = (__popcll(m0) + __popcll(m1)) + (__popcll(m2) + __popcll(m3));
                if (c >= k) {
                    hi = mid;
                    if (c <= k + 8) break;
                } else {
                    lo = mid + 1;
                }
            }
            T0 = hi;
        }

        KS_T(1)
        // ---- survivors -> LDS list.  Per lane a bit per row first (two VALU operations per row, no scalar round trip:
        // the sign of T0 - d is the "not a survivor" bit, shifted into the word), then the lanes pop their few bits together,
        // four per trip, and RECOMPUTE those candidates' distances from LDS (registers cannot be indexed by a popped bit;
        // the same operations give the same bits).  A branch per row on a ballot cost 86 cycles per row.
        constexpr int NW = (R + 31) / 32, RW = R < 32 ? R : 32;
        unsigned hm[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            unsigned nm = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < RW; i++) nm = __builtin_amdgcn_alignbit(nm, T0 - d[w * 32 + i], 31);   // row 32 w + i -> bit RW-1-i
            hm[w] = ~nm;
        }
        int S = 0;
#pragma unroll 1
        for (;;) {
            unsigned left = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) left |= hm[w];
            if (__ballot(left != 0) == 0 || S > KS_CAP) break;
            int jj[4];
            bool has[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int r = 0;
                bool done = false;
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    const bool sel = !done && hm[w] != 0;
                    r = sel ? w * 32 + RW - 1 - __builtin_ctz(hm[w] | 0x80000000u) : r;
                    hm[w] = sel ? hm[w] & (hm[w] - 1) : hm[w];
                    done = done || sel;
                }
                has[e] = done;
                jj[e] = (r >> 2) * 256 + l4 + (r & 3);
            }
            float cx[4], cy[4], cz[4];
#pragma unroll
            for (int e = 0; e < 4; e++) { cx[e] = sc[0][jj[e]]; cy[e] = sc[1][jj[e]]; cz[e] = sc[2][jj[e]]; }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float dx = cx[e] - qx, dy = cy[e] - qy, dz = cz[e] - qz;
                const float dd = (dx * dx + dy * dy) + dz * dz;
                const unsigned long long m = __ballot(has[e]);
                const int pos = S + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (has[e] && pos < KS_CAP) sv[pos] = make_uint2((unsigned)jj[e], __float_as_uint(dd));
                S += __popcll(m);
            }
        }
        KS_T(2)
#ifdef KS_TIMING
        ks_acc[6] += S;
        ks_acc[7] += S > KS_CAP;
#endif
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
                    if (p) sv[pos] = make_uint2((unsigned)cand_index(r), d[r]);      // exactly k entries
                    S += __popcll(m);
                }
            }
        }
        KS_T(3)
        if (lane < 8) sv[S + lane] = make_uint2(0xffffffffu, 0xffffffffu);           // the rank loop reads eight entries per trip
        __builtin_amdgcn_wave_barrier();

        // ---- rank by counting.  An entry is the 64-bit key (distance bits : index) in memory order; two of the lane's entries
        // share every broadcast read, eight keys are read per trip (four ds_read_b128 in flight before the compares)
        typedef unsigned long long ks_u64x2 __attribute__((ext_vector_type(2)));
        const unsigned long long *kv = (const unsigned long long *)sv;
        const size_t o = ((size_t)b * Nq + q) * k;
#pragma unroll 1
        for (int t0 = 0; t0 < S; t0 += 128) {
            const int e0 = t0 + lane, e1 = t0 + 64 + lane;
            const unsigned long long k0 = kv[min(e0, S)], k1 = kv[min(e1, S)];      // entry S is a sentinel
            int r0 = 0, r1 = 0;
#pragma unroll 1
            for (int s = 0; s < S; s += 8) {
                ks_u64x2 a[4];
#pragma unroll
                for (int i = 0; i < 4; i++) a[i] = *(const ks_u64x2 *)&kv[s + 2 * i];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    r0 += (a[i][0] < k0) + (a[i][1] < k0);
                    r1 += (a[i][0] < k1) + (a[i][1] < k1);
                }
            }
            const unsigned i0 = (unsigned)k0, i1 = (unsigned)k1;
            const float v0 = __uint_as_float((unsigned)(k0 >> 32)), v1 = __uint_as_float((unsigned)(k1 >> 32));
            if (out_mode == KS_OUT_PAIR) {
                int32_t *dst = (int32_t *)idx_out + o;
                if (e0 < S && r0 < k) { dst[r0] = (int32_t)i0; val_out[o + r0] = v0; }
                if (e1 < S && r1 < k) { dst[r1] = (int32_t)i1; val_out[o + r1] = v1; }
            } else {
                int64_t *dst = (int64_t *)idx_out + o;
                if (e0 < S && r0 < k) { dst[r0] = (int64_t)i0; val_out[o + r0] = sqrtf(v0); }
                if (e1 < S && r1 < k) { dst[r1] = (int64_t)i1; val_out[o + r1] = sqrtf(v1); }
            }
        }
        __builtin_amdgcn_wave_barrier();
        KS_T(4)
    }
#ifdef KS_TIMING
    if (lane == 0)
        for (int i = 0; i < 8; i++) g_ks_time[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * KS_WAVES + wave) * 8 + i] = ks_acc[i];
#endif
}

bool l3d_knn_select_supported(int Nc, int k) { return k >= 1 && k <= L3D_KNN_MAX_K && k <= Nc && Nc <= 8192; }
// where it is the faster kernel (tools/knn_select_bench.py, LABLOG R3.19): large k, where the lane-per-query kernels pay K
// operations per insertion, and any k from 1024 candidates up (k = 16 over 8192: 115 us against 291; at 1024 candidates the two
// meet for k <= 8).  Below that the per-query bisection and ranking outweigh a lane's short insertion list.
bool l3d_knn_select_preferred(int Nc, int k) { return l3d_knn_select_supported(Nc, k) && (k > 32 || Nc >= 1024); }

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
