// knn.hip -- fused pairwise-distance + top-k selection for gfx950 (wave64).
//
// Replaces, without ever materialising an [N,M] distance matrix:
//   T1  knn()          utils/model_common_utils.py:3-9     (expanded metric, self cloud, int64)
//   K13 knn_kernel_fast   utils/lib/src/interpolate_gpu.cu:9-57   (direct metric, int32, dist2)
//   K14 three_nn_kernel_fast  interpolate_gpu.cu:81-124            (k = 3)
//   T8  knn_point()    utils/model_common_utils.py:84-100  (direct metric, int64, sqrt)
//
// Kernel shape (MI355X-first, not the reference's thread-per-query + scratch arrays):
//   * one wave64 per workgroup, one query per lane; >= 512 workgroups at B=32,N=1024 so all
//     256 CUs get work;
//   * candidates are staged as float4 (x,y,z,w) tiles in LDS with one coalesced pass over the
//     [M,3] cloud, then read back as wave-uniform ds_read_b128 broadcasts (one LDS access per
//     candidate per wave, no bank conflicts);
//   * each lane keeps its sorted top-K in VGPRs (TopK<K>, common.h).  To avoid running the
//     K-slot insertion network for every candidate just because one of 64 lanes needs it,
//     candidates that beat the lane's current K-th best are appended to a small per-lane LDS
//     queue; the insertion network runs only when some lane's queue is full (and once at the
//     end), on all queued entries at once.
//   * arithmetic replays the reference's fp32 rounding sequence (file is built with
//     -ffp-contract=off; the only fused ops are the explicit fmaf()s of the MKL dot product).
#include "common.h"

#define TILE 1024   // candidates per LDS tile (16 KiB)
#define QCAP 16     // per-lane queue depth (8 KiB per wave)
#define CHUNK 8     // candidates scanned between queue-occupancy checks

// METRIC_EXPANDED_SQ: pointconv_util.knn_point (utils/pointconv_util.py:107-118) ranks square_distance()'s
// expanded form, dist = ((-2 * dot) + |q|^2) + |c|^2 with that rounding sequence, smallest first.
enum { METRIC_EXPANDED = 0, METRIC_DIRECT = 1, METRIC_EXPANDED_SQ = 2 };
enum { OUT_KNN_GRAPH = 0, OUT_KNN_PAIR = 1, OUT_KNN_POINT = 2 };

// 64 < k <= 200 where the wave-per-query selection kernel (knn_select.hip) does not reach (more than 8192 candidates, or its
// metric): one query per lane, k ROUNDS -- round r re-scans the candidates for the nearest one that ranks behind round r - 1's
// (smaller value, or equal value and higher index).  O(k Nc) evaluations per query and five registers of state: no caller in the
// reference's configurations comes here, and the sorted 128 / 200-slot register lists this replaces (rounds 1-5: topk_scan_kernel<128>,
// <200>) spilled 588 - 1635 registers.
template <int METRIC>
__global__ __launch_bounds__(64) void topk_rounds_kernel(
    const float *__restrict__ qxyz, const float *__restrict__ cxyz, int Nq, int Nc, int k,
    int out_mode, void *__restrict__ idx_out, float *__restrict__ val_out)
{
    __shared__ float4 cand[TILE];
    const int lane = threadIdx.x;
    const int b = blockIdx.y;
    const int q = blockIdx.x * 64 + lane;
    const bool valid = q < Nq;
    const int qc = valid ? q : Nq - 1;
    const float *qp = qxyz + ((size_t)b * Nq + qc) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float qxx = (qx * qx + qy * qy) + qz * qz;       // torch.sum(x**2): sequential rounding
    const float *cbase = cxyz + (size_t)b * Nc * 3;
    auto eval = [&](const float4 c) -> float {              // topk2_kernel's ranking values: larger = nearer
        if (METRIC == METRIC_EXPANDED) {
            const float dot = fmaf(qz, c.z, fmaf(qy, c.y, qx * c.x));
            return fmaf(2.0f, dot, c.w) - qxx;
        } else if (METRIC == METRIC_EXPANDED_SQ) {
            const float dot = fmaf(qz, c.z, fmaf(qy, c.y, qx * c.x));
            return -(fmaf(-2.0f, dot, qxx) + c.w);
        } else {
            const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
            return -((dx * dx + dy * dy) + dz * dz);
        }
    };
    float last_key = INFINITY;
    int last_idx = -1;
    const size_t o = ((size_t)b * Nq + q) * k;
    for (int r = 0; r < k; r++) {
        float best_key = -INFINITY;
        int best_idx = 0x7fffffff;
        for (int c0 = 0; c0 < Nc; c0 += TILE) {
            const int tn = min(TILE, Nc - c0);
            __syncthreads();
            l3d_stage_points<8>(cbase + (size_t)c0 * 3, tn, lane, 64, [&](int t, float x, float y, float z) {
                float w = 0.f;
                if (METRIC == METRIC_EXPANDED) w = -((x * x + y * y) + z * z);
                if (METRIC == METRIC_EXPANDED_SQ) w = (x * x + y * y) + z * z;
                cand[t] = make_float4(x, y, z, w);
            });
            __syncthreads();
            for (int t = 0; t < tn; t++) {
                const float key = eval(cand[t]);
                const int j = c0 + t;
                const bool behind = key < last_key || (key == last_key && j > last_idx);
                const bool take = behind && key > best_key;           // ascending j: the first of equal values stays
                best_key = take ? key : best_key;
                best_idx = take ? j : best_idx;
            }
        }
        if (best_idx == 0x7fffffff) { best_idx = min(last_idx + 1, Nc - 1); best_key = -INFINITY; }   // nothing left that compares (NaN rows)
        if (valid) {
            if (out_mode == OUT_KNN_GRAPH) {
                ((int64_t *)idx_out)[o + r] = best_idx;
            } else if (out_mode == OUT_KNN_PAIR) {
                ((int32_t *)idx_out)[o + r] = best_idx;
                val_out[o + r] = -best_key;
            } else {
                ((int64_t *)idx_out)[o + r] = best_idx;
                val_out[o + r] = sqrtf(-best_key);
            }
        }
        last_key = best_key;
        last_idx = best_idx;
    }
}

// ---------------------------------------------------------------------------------------------
// Two-pass variant (K <= 64): what makes the single-pass kernel above slow is the index half of
// the insertion network (cmp + 2 cndmask per slot, plus spurious moves) running on every queue
// entry of the busiest lane.  Here
//   pass 1 keeps VALUES ONLY (one v_med3_f32 per slot) and ends with the exact K-th best key;
//   pass 2 rescans (the cloud is L2/LDS resident) and collects the (< K) candidates strictly above
//          that key plus the first K candidates equal to it, in index order;
//   the full (value, index) network then runs on ~K entries once.
// W waves per workgroup split the candidate range into W contiguous slices for the SAME 64
// queries (all SIMDs busy at B=32, N=1024; ties still resolve to the lower index because slice
// w's indices all precede slice w+1's), and wave 0 merges the W sorted lists through LDS.
// ---------------------------------------------------------------------------------------------
#define T2 256          // candidates per per-wave LDS tile (4 KiB); multiple of 32

template <int K, int METRIC, int W>
__global__ __launch_bounds__(64 * W) void topk2_kernel(
    const float *__restrict__ qxyz, const float *__restrict__ cxyz, int Nq, int Nc, int k,
    int out_mode, void *__restrict__ idx_out, float *__restrict__ val_out)
{
    // per-wave LDS: a candidate tile and one scratch area that is the pass-1 key queue
    // ([QCAP][64]), the value-merge mailbox ([K][64]), then the pass-2 collection (akey | aidx | bidx,
    // [K+1][64] each).  20 KiB per wave at K=20 -> two 4-wave workgroups per CU.
    constexpr int KR = K + 1;                          // list rows: K entries + one dummy row the branch-free
                                                       // append may scribble on when a list is full
    constexpr int SCR = (2 * K + KR > QCAP ? 2 * K + KR : QCAP) * 64;   // A lists never fill (< K keys beat the K-th best)
    __shared__ float4 cand[W][T2 + 1];                 // slot T2: a sentinel whose key is -inf
    __shared__ int cnts[W][2][64];                     // pass 2's per-slice entry counts; during pass 1 its first W*64
                                                       // words hold each wave's current K-th best (sthr), shared
                                                       // between the slices -- the workgroup sits 960 B under the
                                                       // 80 KiB that let two of them share a CU, so nothing is added
    volatile float *sthr = (volatile float *)&cnts[0][0][0];
    __shared__ float scratch[W][SCR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int q = blockIdx.x * 64 + lane;
    const bool valid = q < Nq;
    const int qc = valid ? q : Nq - 1;
    const float *qp = qxyz + ((size_t)b * Nq + qc) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float qxx = (qx * qx + qy * qy) + qz * qz;
    const float *cbase = cxyz + (size_t)b * Nc * 3;

    auto sentinel = []() -> float4 {                         // a candidate whose key is -inf for every query
        return METRIC == METRIC_EXPANDED      ? make_float4(0.f, 0.f, 0.f, -INFINITY)
               : METRIC == METRIC_EXPANDED_SQ ? make_float4(0.f, 0.f, 0.f, INFINITY)
                                              : make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.f);
    };
    sthr[wave * 64 + lane] = -INFINITY;                      // visible to the other waves after stage()'s first barrier
    if (lane == 0)
        cand[wave][T2] = sentinel();

    float *akey = scratch[wave];                       // [K][64]
    int *aidx = (int *)scratch[wave] + K * 64;         // [K][64]
    int *bidx = (int *)scratch[wave] + 2 * K * 64;     // [KR][64]

    const int per = (Nc + W - 1) / W;                 // slice length (uniform)
    const int lo = wave * per, hi = min(Nc, lo + per);
    const int ntiles = (per + T2 - 1) / T2;           // uniform trip count -> barriers are safe

    auto eval = [&](const float4 c) -> float {
        if (METRIC == METRIC_EXPANDED) {
            const float dot = fmaf(qz, c.z, fmaf(qy, c.y, qx * c.x));
            const float tt = fmaf(2.0f, dot, c.w);
            return tt - qxx;
        } else if (METRIC == METRIC_EXPANDED_SQ) {
            const float dot = fmaf(qz, c.z, fmaf(qy, c.y, qx * c.x));
            const float tt = fmaf(-2.0f, dot, qxx);
            return -(tt + c.w);
        } else {
            const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
            return -((dx * dx + dy * dy) + dz * dz);
        }
    };
    auto stage = [&](int c0, int tn) {
        __syncthreads();
        const int tp = (tn + 31) & ~31;               // padded with sentinels whose key is -inf
        for (int t = lane; t < tp; t += 64) {
            float4 v;
            if (t < tn) {
                const float *cp = cbase + (size_t)(c0 + t) * 3;
                const float x = cp[0], y = cp[1], z = cp[2];
                const float cc = (x * x + y * y) + z * z;
                v = make_float4(x, y, z, METRIC == METRIC_EXPANDED ? -cc : METRIC == METRIC_EXPANDED_SQ ? cc : 0.f);
            } else {
                v = sentinel();
            }
            cand[wave][t] = v;
        }
        __syncthreads();
    };

#ifdef KNN_TIMING
#define KT(i) if (lane == 0) ((long long *)val_out)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * W + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define KT(i)
#endif
    KT(0)
    // ------------------------------------------------------------------ pass 1: K-th best key
    float thrF;
    {
        TopKV<K> tv;
        tv.init();
        float thr = -INFINITY;
        // Scan 32 candidates at a time; a candidate that beats the lane's current K-th best only sets
        // a bit in a VGPR mask (no LDS write, no exec-mask branch per candidate -- with 8 waves per CU
        // hammering the LDS, per-candidate queue writes made this loop LDS-bound).  After the 32, the
        // lanes pop their bits together and re-evaluate just those candidates for the value network.
        for (int tile = 0; tile < ntiles; tile++) {
            const int c0 = lo + tile * T2;
            const int tn = max(0, min(T2, hi - c0));
            stage(c0, tn);
            for (int g0 = 0; g0 < tn; g0 += 32) {
                unsigned mask = 0;
#pragma unroll
                for (int ch = 0; ch < 32; ch += CHUNK) {
                    float4 c[CHUNK];
#pragma unroll
                    for (int u = 0; u < CHUNK; u++) c[u] = cand[wave][g0 + ch + u];
#pragma unroll
                    for (int u = 0; u < CHUNK; u++) mask |= eval(c[u]) > thr ? (1u << (ch + u)) : 0u;
                }
                // pop 4 bits per trip: the 4 per-lane LDS reads are independent, so one LDS round trip
                // is paid per 4 insertions instead of per insertion.  A lane that has run out of bits reads the
                // sentinel slot (key -inf = a no-op insertion): no per-lane boolean lives across the loop, and
                // the loop is rotated by hand (if + do-while) because the compiler may not rotate a loop whose
                // condition is a convergent ballot -- unrotated, every trip copied the whole value list twice
                // and took four exec-mask branches (featknn.hip met the same problem).
                if (__builtin_amdgcn_ballot_w64(mask != 0) != 0) {
#pragma unroll 1
                    do {
                        float4 cc[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int bpos = mask != 0 ? g0 + __builtin_ctz(mask) : T2;
                            mask &= mask - 1;
                            cc[e] = cand[wave][bpos];
                        }
#pragma unroll
                        for (int e = 0; e < 4; e++) tv.insert(eval(cc[e]));
                    } while (__builtin_amdgcn_ballot_w64(mask != 0) != 0);
                }
                // The K-th best of ANY slice is a lower bound of the global K-th best, and it only ever rises: publish
                // this wave's, read the other waves' (whatever they have published so far -- no barrier needed, a
                // stale value is still a valid bound) and filter with the maximum.  A candidate <= that bound cannot
                // change the K-th largest value of the union, which is all pass 1 is after; it cuts the insertions
                // of a quarter-cloud slice from ~K ln(N/4K)+K to roughly a quarter of the single-stream count.
                thr = tv.worst();
                if (W > 1) {
                    sthr[wave * 64 + lane] = thr;
#pragma unroll
                    for (int w = 0; w < W; w++) thr = fmaxf(thr, sthr[w * 64 + lane]);
                }
            }
        }
        KT(1)
        // ---- global K-th best key: tree-merge the W value lists (one v_med3 per slot and entry),
        //      then wave 0 broadcasts its K-th value.  With the GLOBAL threshold pass 2 collects only
        //      the ~K winners (plus exact ties) over all slices, and the expensive (value, index)
        //      network runs once, on wave 0, instead of once per slice plus once per merge.
        float *mail = scratch[wave];
#pragma unroll 1
        for (int step = 1; step < W; step <<= 1) {
            const bool sender = (wave & (2 * step - 1)) == step;
            const bool receiver = (wave & (2 * step - 1)) == 0 && wave + step < W;
            __syncthreads();
            if (sender) {
#pragma unroll
                for (int i = 0; i < K; i++) mail[i * 64 + lane] = tv.v[i];
            }
            __syncthreads();
            if (receiver) {
                const float *mk = scratch[wave + step];
                // sorted lists: once no lane's entry can enter, nothing further can.  Rotated by hand (see pass 1).
                int i = 0;
                float kv = mk[lane];
                if (__builtin_amdgcn_ballot_w64(kv > tv.worst()) != 0) {
#pragma unroll 1
                    do {
                        const float kn = mk[min(i + 1, K - 1) * 64 + lane];
                        tv.insert(kv);
                        kv = kn;
                        i++;
                    } while (i < K && __builtin_amdgcn_ballot_w64(kv > tv.worst()) != 0);
                }
            }
        }
        __syncthreads();
        if (wave == 0) scratch[0][lane] = tv.worst();
        __syncthreads();
        thrF = scratch[0][lane];
        __syncthreads();                                     // everyone has read it before scratch is reused
    }

    KT(2)
    // ------------------ pass 2: collect (key > thrF) and the first K (key == thrF), thrF GLOBAL now
    int cntA = 0, cntB = 0;
    for (int tile = 0; tile < ntiles; tile++) {
        const int c0 = lo + tile * T2;
        const int tn = max(0, min(T2, hi - c0));
        if (ntiles > 1) stage(c0, tn);            // single-tile slices are still resident from pass 1
        for (int g0 = 0; g0 < tn; g0 += 32) {
            unsigned ma = 0, mb = 0;
#pragma unroll
            for (int ch = 0; ch < 32; ch += CHUNK) {
                float4 c[CHUNK];
#pragma unroll
                for (int u = 0; u < CHUNK; u++) c[u] = cand[wave][g0 + ch + u];
#pragma unroll
                for (int u = 0; u < CHUNK; u++) {
                    const float key = eval(c[u]);
                    ma |= key > thrF ? (1u << (ch + u)) : 0u;
                    mb |= key == thrF ? (1u << (ch + u)) : 0u;
                }
            }
            // hits are rare (~K in total over all slices): pop them in index order
#pragma unroll 1
            while (__any(ma != 0)) {
                const bool has = ma != 0 && cntA < K;
                const int bpos = ma != 0 ? __builtin_ctz(ma) : 0;
                ma &= ma - 1;
                const int pa = min(cntA, K - 1);
                akey[pa * 64 + lane] = eval(cand[wave][g0 + bpos]);
                aidx[pa * 64 + lane] = c0 + g0 + bpos;
                cntA += has ? 1 : 0;
            }
#pragma unroll 1
            while (__any(mb != 0)) {
                const bool has = mb != 0 && cntB < K;
                const int bpos = mb != 0 ? __builtin_ctz(mb) : 0;
                mb &= mb - 1;
                bidx[min(cntB, K) * 64 + lane] = c0 + g0 + bpos;      // row K is the dummy row
                cntB += has ? 1 : 0;
            }
        }
    }

    KT(3)
    // ---------------- wave 0 builds the (value, index) list from every slice's collected entries,
    // slices in index order, entries in index order within a slice: strict '>' insertion therefore
    // keeps lowest-index-first under exact ties.
    cnts[wave][0][lane] = cntA;
    cnts[wave][1][lane] = cntB;
    __syncthreads();
    TopK<K> top;
    top.init();
    if (wave == 0) {
        // walk each lane's entries as ONE list (slice 0's, then slice 1's, ...): the trip count is the
        // busiest lane's TOTAL (about K), not the sum over slices of the busiest lane per slice
#pragma unroll
        for (int which = 0; which < 2; which++) {            // 0: keys > thrF (akey/aidx), 1: keys == thrF (bidx)
            // cumulative entry counts of the slices, in registers: entry `it` of the lane's virtual list sits in
            // slice w = #(cum[.] <= it), slot it - cum[w-1] -- no LDS round trips to find it
            int cum[W];
            int tot = 0;
#pragma unroll
            for (int w = 0; w < W; w++) { tot += cnts[w][which][lane]; cum[w] = tot; }
            int tmax = tot;                                   // uniform trip count = the busiest lane's total
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) tmax = max(tmax, __shfl_xor(tmax, d, 64));
            tmax = __builtin_amdgcn_readfirstlane(tmax);
            const int list_off = which == 0 ? K * 64 : 2 * K * 64;
            auto fetch = [&](int it, float &key, int &id) {
                int w = 0, base = 0;
#pragma unroll
                for (int u = 0; u < W - 1; u++) {
                    const bool past = it >= cum[u];
                    w += past ? 1 : 0;
                    base = past ? cum[u] : base;
                }
                const int slot = min(it - base, K - 1);       // lanes past their total read a valid (ignored) slot
                const float *ak = scratch[0] + (size_t)w * SCR;
                const float kv = which == 0 ? ak[slot * 64 + lane] : thrF;
                id = ((const int *)ak)[list_off + slot * 64 + lane];
                key = it < tot ? kv : -INFINITY;               // a -inf key is a no-op insertion
            };
            float key_c;
            int id_c;
            fetch(0, key_c, id_c);
#pragma unroll 1
            for (int it = 0; it < tmax; it++) {
                float key_n;
                int id_n;
                fetch(it + 1, key_n, id_n);                    // the next entry's LDS reads overlap this insertion
                if constexpr (K == 20) topk20_insert(top, key_c, id_c);
                else top.insert(key_c, id_c);
                key_c = key_n;
                id_c = id_n;
            }
        }
    }

    KT(4)
    if (wave != 0 || !valid) return;
    const size_t o = ((size_t)b * Nq + q) * k;
    if (out_mode == OUT_KNN_GRAPH) {
        int64_t *dst = (int64_t *)idx_out + o;
#pragma unroll
        for (int i = 0; i < K; i++)
            if (i < k) dst[i] = top.id[i];
    } else if (out_mode == OUT_KNN_PAIR) {
        int32_t *dst = (int32_t *)idx_out + o;
#pragma unroll
        for (int i = 0; i < K; i++)
            if (i < k) { dst[i] = top.id[i]; val_out[o + i] = -top.v[i]; }
    } else {
        int64_t *dst = (int64_t *)idx_out + o;
#pragma unroll
        for (int i = 0; i < K; i++)
            if (i < k) { dst[i] = top.id[i]; val_out[o + i] = sqrtf(-top.v[i]); }
    }
}

template <int METRIC>
static int launch_topk(const float *q, const float *c, int B, int Nq, int Nc, int k, int out_mode,
                       void *idx, float *val, hipStream_t st)
{
    dim3 grid(l3d_divup(Nq, 64), B), block(64);
#define L3D_TOPK2_CASE(KK, WW)                                                                   \
    if (k <= KK) {                                                                               \
        hipLaunchKernelGGL((topk2_kernel<KK, METRIC, WW>), grid, dim3(64 * WW), 0, st, q, c, Nq, \
                           Nc, k, out_mode, idx, val);                                           \
        return l3d_check_launch();                                                               \
    }
    L3D_TOPK2_CASE(4, 4)
    L3D_TOPK2_CASE(8, 4)
    L3D_TOPK2_CASE(16, 4)
    L3D_TOPK2_CASE(20, 4)
    L3D_TOPK2_CASE(32, 4)
    L3D_TOPK2_CASE(64, 2)
#undef L3D_TOPK2_CASE
    if (k <= L3D_KNN_MAX_K) {
        hipLaunchKernelGGL((topk_rounds_kernel<METRIC>), grid, block, 0, st, q, c, Nq, Nc, k, out_mode, idx, val);
        return l3d_check_launch();
    }
    return L3D_ERR_UNSUPPORTED;
}

// knn_mfma.hip
bool l3d_knn_mfma_supported(int N, int k);
int l3d_launch_knn_mfma(const float *xyz, int B, int N, int k, int64_t *idx, hipStream_t st);

extern "C" int l3d_knn_graph_variant(const float *xyz, int B, int N, int k, int64_t *idx, int variant,
                                     l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && idx && B > 0 && N > 0 && k > 0 && k <= N && k <= L3D_KNN_MAX_K && variant >= 0 && variant <= 2);
    hipStream_t st = (hipStream_t)stream;
    const bool mfma = l3d_knn_mfma_supported(N, k);
    if (variant == 2 && !mfma) return L3D_ERR_UNSUPPORTED;
    if (variant != 1 && mfma)          // ranking values on the matrix cores + selection by rank counting (knn_mfma.hip)
        return l3d_launch_knn_mfma(xyz, B, N, k, idx, st);
    return launch_topk<METRIC_EXPANDED>(xyz, xyz, B, N, N, k, OUT_KNN_GRAPH, idx, nullptr, st);
}

extern "C" int l3d_knn_graph(const float *xyz, int B, int N, int k, int64_t *idx, l3d_stream_t stream)
{
    return l3d_knn_graph_variant(xyz, B, N, k, idx, 0, stream);
}

// knn_select.hip: a wave per query, k-th distance from bucket minima + rank counting (33 <= k <= 200, m <= 8192)
bool l3d_knn_select_supported(int Nc, int k);
bool l3d_knn_select_preferred(int Nc, int k);
// knn_small.hip: k <= 4 (three_nn), a query per lane with four sorted slots and a 32-candidate hit mask
bool l3d_knn_small_supported(int Nc, int k);
bool l3d_knn_small_preferred(long queries, int Nc, int k);
int l3d_launch_knn_small(const float *q, const float *c, int B, int Nq, int Nc, int k, int out_mode, void *idx, float *val, hipStream_t st);
int l3d_launch_knn_select(const float *q, const float *c, int B, int Nq, int Nc, int k, int out_mode, void *idx, float *val,
                          hipStream_t st);

extern "C" int l3d_knn_variant(int b, int n, int m, int k, const float *unknown, const float *known,
                               float *dist2, int32_t *idx, int variant, l3d_stream_t stream)
{
    L3D_REQUIRE(unknown && known && dist2 && idx && b > 0 && n > 0 && m > 0 && k > 0 &&
                k <= L3D_KNN_MAX_K && variant >= 0 && variant <= 3);
    if (variant == 2 && !l3d_knn_select_supported(m, k)) return L3D_ERR_UNSUPPORTED;
    if (variant == 3 && !l3d_knn_small_supported(m, k)) return L3D_ERR_UNSUPPORTED;
    if (variant == 3 || (variant == 0 && l3d_knn_small_preferred((long)b * n, m, k)))
        return l3d_launch_knn_small(unknown, known, b, n, m, k, OUT_KNN_PAIR, idx, dist2, (hipStream_t)stream);
    if (variant == 2 || (variant == 0 && l3d_knn_select_preferred(m, k)))
        return l3d_launch_knn_select(unknown, known, b, n, m, k, OUT_KNN_PAIR, idx, dist2, (hipStream_t)stream);
    return launch_topk<METRIC_DIRECT>(unknown, known, b, n, m, k, OUT_KNN_PAIR, idx, dist2,
                                      (hipStream_t)stream);
}

extern "C" int l3d_knn(int b, int n, int m, int k, const float *unknown, const float *known,
                       float *dist2, int32_t *idx, l3d_stream_t stream)
{
    return l3d_knn_variant(b, n, m, k, unknown, known, dist2, idx, 0, stream);
}

// pointconv_util.knn_point (utils/pointconv_util.py:107-118): nsample smallest square_distance(new_xyz, xyz)
// entries per query, indices only.  The reference asks torch.topk for sorted=False (order unspecified); the
// rows here come out nearest first, lowest index first under exact ties.
extern "C" int l3d_knn_point_expanded(int nsample, const float *xyz, const float *new_xyz, int B, int N, int S,
                                      int64_t *idx, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && new_xyz && idx && B > 0 && N > 0 && S > 0 && nsample > 0 && nsample <= N &&
                nsample <= L3D_KNN_MAX_K);
    return launch_topk<METRIC_EXPANDED_SQ>(new_xyz, xyz, B, S, N, nsample, OUT_KNN_GRAPH, idx, nullptr,
                                           (hipStream_t)stream);
}

extern "C" int l3d_three_nn(int b, int n, int m, const float *unknown, const float *known,
                            float *dist2, int32_t *idx, l3d_stream_t stream)
{
    return l3d_knn(b, n, m, 3, unknown, known, dist2, idx, stream);
}

extern "C" int l3d_knn_point(int k, const float *pos1, const float *pos2, int B, int N, int M,
                             float *val, int64_t *idx, l3d_stream_t stream)
{
    L3D_REQUIRE(pos1 && pos2 && val && idx && B > 0 && N > 0 && M > 0 && k > 0 && k <= N &&
                k <= L3D_KNN_MAX_K);
    if (l3d_knn_small_preferred((long)B * M, N, k))
        return l3d_launch_knn_small(pos2, pos1, B, M, N, k, OUT_KNN_POINT, idx, val, (hipStream_t)stream);
    if (l3d_knn_select_preferred(N, k))
        return l3d_launch_knn_select(pos2, pos1, B, M, N, k, OUT_KNN_POINT, idx, val, (hipStream_t)stream);
    return launch_topk<METRIC_DIRECT>(pos2, pos1, B, M, N, k, OUT_KNN_POINT, idx, val,
                                      (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// get_graph_feature gather: out[b][n][j] = (x[b][idx[b][n][j]], x[b][n])   [B,N,k,2C]
// one thread per (b,n,j); 2C consecutive floats written per thread (C=3: 24 B).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void graph_feature_kernel(const float *__restrict__ x,
                                                            const int64_t *__restrict__ idx, int N,
                                                            int C, int k, size_t total,
                                                            float *__restrict__ out)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    size_t bn = e / k;
    size_t b = bn / N;
    int64_t j = idx[e];
    const float *nb = x + ((size_t)b * N + j) * C;
    const float *ct = x + bn * C;
    float *o = out + e * (size_t)(2 * C);
    for (int c = 0; c < C; c++) o[c] = nb[c];
    for (int c = 0; c < C; c++) o[C + c] = ct[c];
}

extern "C" int l3d_graph_feature(const float *x, const int64_t *idx, int B, int N, int C, int k,
                                 float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(x && idx && out && B > 0 && N > 0 && C > 0 && k > 0);
    size_t total = (size_t)B * N * k;
    hipLaunchKernelGGL(graph_feature_kernel, dim3(l3d_divup(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, idx, N, C, k, total, out);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// CurveNet LPFA grouping (utils/curvenet_util.py:260-291), one pass:
//   geo  [B,9,N,k]  = (centre xyz, neighbour xyz, neighbour - centre)            (:273-275)
//   diff [B,C,N,k]  = x[:, :, idx] - x[:, :, n]          (only when x != NULL)   (:280-285)
// xyz [B,N,3]; x [B,C,N] channel-first as the reference holds it; idx [B,N,k] int64 (from l3d_knn_graph with k+1,
// first k kept, :264).  HBM-bound gathers; one thread per (b, n, j), channel loop with coalesced writes over j.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lpfa_group_kernel(const float *__restrict__ xyz, const float *__restrict__ x,
                                                         const int64_t *__restrict__ idx, int N, int C, int k, size_t total,
                                                         float *__restrict__ geo, float *__restrict__ diff)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;            // (b, n, j)
    if (e >= total) return;
    const size_t bn = e / k, b = bn / N, n = bn % N;
    const int jj = (int)(e % k);
    const int64_t j = idx[e];
    const float *pc = xyz + bn * 3, *pn = xyz + ((size_t)b * N + j) * 3;
    const size_t plane = (size_t)N * k, o = n * k + jj;
    float *g = geo + b * 9 * plane + o;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        g[(size_t)c * plane] = pc[c];
        g[(size_t)(3 + c) * plane] = pn[c];
        g[(size_t)(6 + c) * plane] = pn[c] - pc[c];
    }
    if (x) {
        const float *xb = x + b * (size_t)C * N;
        float *d = diff + b * (size_t)C * plane + o;
        for (int c = 0; c < C; c++) d[(size_t)c * plane] = xb[(size_t)c * N + j] - xb[(size_t)c * N + n];
    }
}

extern "C" int l3d_lpfa_group(const float *xyz, const float *x, const int64_t *idx, int B, int N, int C, int k, float *geo,
                              float *diff, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && idx && geo && B > 0 && N > 0 && k > 0 && (!x || (diff && C > 0)));
    const size_t total = (size_t)B * N * k;
    hipLaunchKernelGGL(lpfa_group_kernel, dim3((unsigned)l3d_divup((long)total, 256)), dim3(256), 0, (hipStream_t)stream, xyz, x, idx, N,
                       x ? C : 0, k, total, geo, diff);
    return l3d_check_launch();
}
