// chamfer_mfma.hip -- Chamfer nearest-neighbour search for LARGE clouds: candidates ranked on the fp16 matrix cores,
// the answer fixed by an exact re-evaluation, so distances and lowest-index argmins keep the bits of the reference
// (losses/cuda/chamfer_distance/chamfer_distance.cu:6-137, CPU twin chamfer_distance.cpp:59-87).
//
// chamfer.hip evaluates every (query, candidate) pair on the VALU: 4 packed + ~1 issue slots per pair, 0.34 of the packed
// fp32 peak at config 4 (64 x 16384 x 16384 pairs, both directions: 4.8 ms).  Here a pair costs 1/1024 of a
// v_mfma_f32_32x32x16_f16 (32 cycles per SIMD) plus 1/2 of a v_min3_f32:
//
//   * coordinates are centred and scaled per workgroup to |X| <= 2^9 (a power-of-two scale from the bounding box the workgroup
//     measures itself: no workspace, no extra launch, and every workgroup is self-contained) and split f16x2-style,
//     X ~ h + m 2^-12.  Twelve of the MFMA's sixteen k slots carry
//         s~(q, c) = |C|^2 - 2 Q.C = (256 p0 + p1 + 2^-12 p2) - 2 (hq.hc + 2^-12 mq.hc + 2^-12 hq.mc)
//     (|C|^2 in three fp16 pieces, exact; only the mq.mc 2^-24 term is dropped): |s~ - true| < 3 in these units whatever the
//     input (DESIGN.md 4.3 adds the terms up); |Q|^2 is constant per query and irrelevant to its argmin.
//   * queries are the MFMA's COLUMNS: a lane then holds 16 candidates of ONE query, min3-reduced in registers, no cross-lane
//     traffic.  A wave owns 128 queries (4 column blocks), a workgroup 512; candidates stream through LDS as A fragments.
//   * pass 1 takes the minimum of s~ per query (over every 4th group of 32 candidates when there are >= 8192 of them: an upper
//     bound of the true minimum is all that is needed); pass 2 recomputes the products and records every (query, 16-candidate
//     group) whose minimum lies within CM_BAND of the lane's threshold, which tightens with every record -- the true minimisers
//     are all among the records (band >= 2 x error bound).  1 - 2.5 records per query on uniform clouds; lattices with many
//     exact ties just record more.
//   * the records are evaluated EXACTLY, 16 candidates per record by 16 lanes, in the reference's arithmetic
//     ((dx*dx + dy*dy) + dz*dz, dx = c - q, no contraction), smaller distance then lower index winning: what one sequential
//     strict-'<' scan over all candidates returns.
//   * clouds with non-finite coordinates (or an extent no power of two scales) take an in-kernel exact scan instead.
#include "common.h"

#define CM_QW 512            // queries per workgroup: 4 waves x 4 column blocks of 32
#define CM_CHUNK 512         // candidates per LDS chunk = 16 MFMA steps of 32
#define CM_LIST 1024         // refinement records per wave
#define CM_BAND 16.0f        // in scaled units^2: > 2 x (error bound 3) + the reference's own rounding of d (< 1.5 at |X| <= 2^9)

typedef _Float16 cm_f16x8 __attribute__((ext_vector_type(8)));
typedef float cm_f32x16 __attribute__((ext_vector_type(16)));

struct CmXform { float cx, cy, cz, s; };

__device__ __forceinline__ void cm_split(float X, _Float16 &h, _Float16 &m)
{
    h = (_Float16)X;
    m = (_Float16)((X - (float)h) * 4096.0f);
}

__device__ __forceinline__ float cm_min16(const cm_f32x16 &a, float run)
{
    float m0 = __builtin_fminf(__builtin_fminf(a[0], a[1]), run);
    float m1 = __builtin_fminf(__builtin_fminf(a[2], a[3]), a[4]);
    float m2 = __builtin_fminf(__builtin_fminf(a[5], a[6]), a[7]);
    float m3 = __builtin_fminf(__builtin_fminf(a[8], a[9]), a[10]);
    float m4 = __builtin_fminf(__builtin_fminf(a[11], a[12]), a[13]);
    m0 = __builtin_fminf(__builtin_fminf(m0, a[14]), a[15]);
    m1 = __builtin_fminf(__builtin_fminf(m1, m2), m3);
    return __builtin_fminf(__builtin_fminf(m0, m1), m4);
}

template <int SUB>
__global__ __launch_bounds__(256, 2) void chamfer_mfma_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2, int N,
                                                           int M, float *__restrict__ dist1, float *__restrict__ dist2,
                                                           int32_t *__restrict__ idx1, int32_t *__restrict__ idx2)
{
    __shared__ __attribute__((aligned(16))) unsigned char abuf2[2][CM_CHUNK * 32];  // two chunks; per 32-candidate step: [k 0-7][32 rows] | [k 8-15][32 rows]
    __shared__ uint32_t list[4][CM_LIST];
    __shared__ float bestd[CM_QW];
    __shared__ int besti[CM_QW];
    __shared__ float red[4][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
    const int b = blockIdx.y, dir = blockIdx.z;
    const float *qs = dir == 0 ? xyz1 : xyz2, *cs = dir == 0 ? xyz2 : xyz1;
    const int Nq = dir == 0 ? N : M, Nc = dir == 0 ? M : N;
    float *dout = dir == 0 ? dist1 : dist2;
    int32_t *iout = dir == 0 ? idx1 : idx2;
    const int q0 = blockIdx.x * CM_QW;
    if (q0 >= Nq) return;
    const float *qb = qs + (size_t)b * Nq * 3, *cb = cs + (size_t)b * Nc * 3;

    // ---- sweep 0: bounding box of the candidates and of this workgroup's queries; non-finite check
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, nf = 0.f;
    for (int j = tid; j < Nc; j += 256)
#pragma unroll
        for (int c = 0; c < 3; c++) { const float v = cb[(size_t)j * 3 + c]; mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); nf += v * 0.f; }
    for (int j = tid; j < CM_QW; j += 256)
#pragma unroll
        for (int c = 0; c < 3; c++) { const float v = qb[(size_t)min(q0 + j, Nq - 1) * 3 + c]; mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); nf += v * 0.f; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int c = 0; c < 3; c++) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], off, 64)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off, 64)); }
        nf += __shfl_xor(nf, off, 64);
    }
    if (lane == 0) { red[wave][0] = mn[0]; red[wave][1] = mn[1]; red[wave][2] = mn[2]; red[wave][3] = mx[0]; red[wave][4] = mx[1]; red[wave][5] = mx[2]; red[wave][6] = nf; }
    for (int i = tid; i < CM_QW; i += 256) { bestd[i] = INFINITY; besti[i] = 0x7fffffff; }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; c++) {
        mn[c] = fminf(fminf(red[0][c], red[1][c]), fminf(red[2][c], red[3][c]));
        mx[c] = fmaxf(fmaxf(red[0][3 + c], red[1][3 + c]), fmaxf(red[2][3 + c], red[3][3 + c]));
    }
    nf = (red[0][6] + red[1][6]) + (red[2][6] + red[3][6]);
    CmXform xf;
    xf.cx = 0.5f * mn[0] + 0.5f * mx[0]; xf.cy = 0.5f * mn[1] + 0.5f * mx[1]; xf.cz = 0.5f * mn[2] + 0.5f * mx[2];
    const float R = fmaxf(fmaxf(mx[0] - xf.cx, mx[1] - xf.cy), fmaxf(mx[2] - xf.cz, fmaxf(xf.cx - mn[0], fmaxf(xf.cy - mn[1], xf.cz - mn[2]))));
    int e = 0;
    if (R > 0.f) (void)frexpf(R, &e);                      // R = f 2^e, f in [0.5, 1): |x - c| 2^(9-e) < 2^9 (+ one rounding)
    const bool exact_only = !(nf == 0.f) || !(R < INFINITY) || e > 100 || e < -100;
    xf.s = ldexpf(1.0f, 9 - e);

    const int qw0 = q0 + wave * 128;                       // this wave's 128 queries
    if (exact_only) {
        // garbage-in path: plain scan, two queries per lane (NaN / inf rows keep index 0 like the reference: '<' never fires)
        for (int u = 0; u < 2; u++) {
            const int q = qw0 + u * 64 + lane;
            if (q >= Nq) continue;
            const float x = qb[(size_t)q * 3], y = qb[(size_t)q * 3 + 1], z = qb[(size_t)q * 3 + 2];
            float bd = INFINITY; int bi = 0;
            for (int j = 0; j < Nc; j++) {
                const float dx = cb[(size_t)j * 3] - x, dy = cb[(size_t)j * 3 + 1] - y, dz = cb[(size_t)j * 3 + 2] - z;
                const float d = (dx * dx + dy * dy) + dz * dz;
                if (d < bd) { bd = d; bi = j; }
            }
            dout[(size_t)b * Nq + q] = bd; iout[(size_t)b * Nq + q] = bi;
        }
        return;
    }

    // ---- this wave's query fragments (B operand: column = query, k 8 half .. +7)
    cm_f16x8 Bq[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = min(qw0 + k * 32 + col, Nq - 1);
        _Float16 hx, hy, hz, mxx, myy, mzz;
        cm_split((qb[(size_t)q * 3] - xf.cx) * xf.s, hx, mxx);
        cm_split((qb[(size_t)q * 3 + 1] - xf.cy) * xf.s, hy, myy);
        cm_split((qb[(size_t)q * 3 + 2] - xf.cz) * xf.s, hz, mzz);
        const cm_f16x8 lo = {hx, hy, hz, mxx, myy, mzz, hx, hy};
        const cm_f16x8 hi = {hz, (_Float16)256.0f, (_Float16)1.0f, (_Float16)0.000244140625f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        Bq[k] = half ? hi : lo;
    }

    // candidate chunk -> A fragments in LDS (row = candidate): k 0-7 = -2hx -2hy -2hz | -2^-11 (hx hy hz) | -2^-11 (mx my); k 8-15 = -2^-11 mz | p0 p1 p2 | 0
    // The chunk's coordinates are REQUESTED a chunk ahead (fetch) and converted + written to the other LDS buffer after the current
    // chunk's MFMAs (commit): a 512-candidate chunk is ~1 us of matrix work, the same as one trip to memory.
    constexpr int PER = CM_CHUNK / 256;
    float sx[PER], sy[PER], sz[PER];
    // iteration it of the pipeline: it < nch1 = pass 1's virtual chunk it (every sub-th 32-candidate group: group g of the chunk
    // is real group (16 it + g) sub), else pass 2's chunk it - nch1 (all groups in order)
    // pass 1 looks at every sub-th group of 32 candidates.  Sampling loosens pass 2's first threshold: ~1 + ln(sub) more records per
    // query, whatever the cloud size -- per MFMA step that is nothing at 512 steps (config 4: 2.0 -> 1.6 ms) and a slow-path visit
    // in every step at 32 (N = 1024: 43 -> 86 us), hence SUB = 4 only when both clouds have >= 8192 points (the launcher)
    constexpr int sub = SUB;               // compile-time: as a kernel argument it costs 880 spilled registers (hipcc 7.2)
    const int ngroups = (Nc + 31) >> 5;
    const int nch1 = ((ngroups + sub - 1) / sub + 15) / 16, nch2 = (Nc + CM_CHUNK - 1) / CM_CHUNK, total = nch1 + nch2;
    auto cand_of = [&](int it, int i) {
        return it < nch1 ? ((it * 16 + (i >> 5)) * sub) * 32 + (i & 31) : (it - nch1) * CM_CHUNK + i;
    };
    auto fetch = [&](int it) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const size_t j = (size_t)min(cand_of(it, tid + u * 256), Nc - 1) * 3;       // unconditional loads from clamped rows
            sx[u] = cb[j]; sy[u] = cb[j + 1]; sz[u] = cb[j + 2];
        }
    };
    auto commit = [&](int it, unsigned char *abuf) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int i = tid + u * 256, j = cand_of(it, i);
            const float X = (sx[u] - xf.cx) * xf.s, Y = (sy[u] - xf.cy) * xf.s, Z = (sz[u] - xf.cz) * xf.s;
            _Float16 hx, hy, hz, mxx, myy, mzz;
            cm_split(X, hx, mxx); cm_split(Y, hy, myy); cm_split(Z, hz, mzz);
            const float c2 = (X * X + Y * Y) + Z * Z;
            const _Float16 p0 = (_Float16)(c2 * 0.00390625f);
            const float r1 = c2 - (float)p0 * 256.0f;
            const _Float16 p1 = (_Float16)r1;
            const _Float16 p2 = (_Float16)((r1 - (float)p1) * 4096.0f);
            const _Float16 k2 = (_Float16)-2.0f, k11 = (_Float16)-0.00048828125f, z16 = (_Float16)0.f;
            cm_f16x8 lo = {k2 * hx, k2 * hy, k2 * hz, k11 * hx, k11 * hy, k11 * hz, k11 * mxx, k11 * myy};
            cm_f16x8 hi = {k11 * mzz, p0, p1, p2, z16, z16, z16, z16};
            if (j >= Nc) {                                  // padding row: s~ = 256 * 60000 = 1.5e7, never a minimum
                lo = (cm_f16x8){z16, z16, z16, z16, z16, z16, z16, z16};
                hi = (cm_f16x8){z16, (_Float16)60000.0f, z16, z16, z16, z16, z16, z16};
            }
            unsigned char *row = abuf + (i >> 5) * 1024 + (i & 31) * 16;
            *(cm_f16x8 *)row = lo;
            *(cm_f16x8 *)(row + 512) = hi;
        }
    };
    const int frag_off = half * 512 + col * 16;
    const cm_f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int buf = 0;
    fetch(0);
    commit(0, abuf2[0]);
    fetch(1);                                              // total >= 2: pass 2 has at least one chunk

    // ---- pass 1: per query (lane: its half of every 32-candidate step) the minimum of s~ over a 1-in-sub sample of the
    // candidate groups: an UPPER bound of the query's true minimum, which is all pass 2's first threshold has to be
    float run[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    for (int it = 0; it < nch1; it++) {
        __syncthreads();                                    // abuf2[buf] written by every wave; abuf2[buf ^ 1] free (its readers passed this barrier's predecessor)
        const unsigned char *afrag = abuf2[buf] + frag_off;
        // Every chunk runs its 16 steps (rows past Nc are padding rows): a fixed trip count, fully unrolled.  Inside a wave the
        // stream is software-pipelined -- each MFMA of step s+1 is followed by one min3 tree of step s, pinned by sched_barriers (the
        // scheduler otherwise sinks every MFMA to its own tree and pads the dependency with s_nop 11): an in-order wave that issues four MFMAs back to back sits on the
        // matrix pipe for 96 cycles with its trees waiting behind, then runs 32 min3 with the pipe idle (first version: 0.30 busy).
        cm_f32x16 acc[2][4];
        cm_f16x8 Af[2];                                    // fragment of step s in Af[s & 1], read two steps ahead of its MFMAs
        {
            Af[0] = *(const cm_f16x8 *)afrag;
            Af[1] = *(const cm_f16x8 *)(afrag + 1024);
#pragma unroll
            for (int k = 0; k < 4; k++) acc[0][k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[0], Bq[k], zero, 0, 0, 0);
        }
#pragma unroll
        for (int st = 0; st < CM_CHUNK / 32; st++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {                  // source order IS the schedule: MFMA (next step, block k), then the tree of (this step, block k)
                if (st + 1 < CM_CHUNK / 32) acc[(st + 1) & 1][k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[(st + 1) & 1], Bq[k], zero, 0, 0, 0);
                if (k == 0 && st + 2 < CM_CHUNK / 32) Af[st & 1] = *(const cm_f16x8 *)(afrag + (st + 2) * 1024);
                __builtin_amdgcn_sched_barrier(0);
                run[k] = cm_min16(acc[st & 1][k], run[k]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        commit(it + 1, abuf2[buf ^ 1]);                      // it + 1 <= nch1 < total: the pipeline runs through the pass boundary
        fetch(min(it + 2, total - 1));                      // unconditional (a fetch behind a branch costs the kernel 900 spilled registers: hipcc 7.2)
        buf ^= 1;
    }
    float thr[4];
#pragma unroll
    for (int k = 0; k < 4; k++) thr[k] = fminf(run[k], __shfl_xor(run[k], 32, 64)) + CM_BAND;

    // ---- exact evaluation of the recorded (query, 16-candidate group) pairs, four records at a time
    int cnt = 0;
    auto flush = [&]() {
        const int g = lane >> 4, el = lane & 15;
        for (int i = 0; i < cnt; i += 4) {
            const bool valid = i + g < cnt;
            const uint32_t ent = list[wave][min(i + g, cnt - 1)];
            const int ql = ent & 127, hf = (ent >> 7) & 1, tile = ent >> 8;
            int cand = tile * 32 + (el >> 2) * 8 + hf * 4 + (el & 3);
            const bool live = valid && cand < Nc;
            cand = min(cand, Nc - 1);
            const int q = min(qw0 + ql, Nq - 1);
            const float dx = cb[(size_t)cand * 3] - qb[(size_t)q * 3], dy = cb[(size_t)cand * 3 + 1] - qb[(size_t)q * 3 + 1],
                        dz = cb[(size_t)cand * 3 + 2] - qb[(size_t)q * 3 + 2];
            float d = live ? (dx * dx + dy * dy) + dz * dz : INFINITY;
            int ci = live ? cand : 0x7fffffff;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const float od = __shfl_xor(d, off, 64);
                const int oi = __shfl_xor(ci, off, 64);
                const bool take = od < d || (od == d && oi < ci);
                d = take ? od : d; ci = take ? oi : ci;
            }
#pragma unroll
            for (int gg = 0; gg < 4; gg++)                  // one group after the other: two records of a batch may name the same query
                if (g == gg && el == 0 && valid) {
                    const int slot = wave * 128 + ql;
                    const float bd = bestd[slot];
                    const int bi = besti[slot];
                    if (d < bd || (d == bd && ci < bi)) { bestd[slot] = d; besti[slot] = ci; }
                }
        }
        cnt = 0;
    };

    // ---- pass 2: the same products again; record every lane-group whose minimum is inside the band
    for (int it = nch1; it < total; it++) {
        __syncthreads();
        const unsigned char *afrag = abuf2[buf] + frag_off;
        const int c0 = (it - nch1) * CM_CHUNK;
        cm_f32x16 acc[2][4];
        cm_f16x8 Af[2];                                    // fragment of step s in Af[s & 1], read two steps ahead of its MFMAs
        {
            Af[0] = *(const cm_f16x8 *)afrag;
            Af[1] = *(const cm_f16x8 *)(afrag + 1024);
#pragma unroll
            for (int k = 0; k < 4; k++) acc[0][k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[0], Bq[k], zero, 0, 0, 0);
        }
#pragma unroll
        for (int st = 0; st < CM_CHUNK / 32; st++) {
            const uint32_t tile = (uint32_t)((c0 >> 5) + st) << 8;
            float mk[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (st + 1 < CM_CHUNK / 32) acc[(st + 1) & 1][k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[(st + 1) & 1], Bq[k], zero, 0, 0, 0);
                if (k == 0 && st + 2 < CM_CHUNK / 32) Af[st & 1] = *(const cm_f16x8 *)(afrag + (st + 2) * 1024);
                __builtin_amdgcn_sched_barrier(0);
                mk[k] = cm_min16(acc[st & 1][k], INFINITY);
                __builtin_amdgcn_sched_barrier(0);
            }
            const bool any = (mk[0] <= thr[0]) | (mk[1] <= thr[1]) | (mk[2] <= thr[2]) | (mk[3] <= thr[3]);
            if (__builtin_amdgcn_ballot_w64(any)) {         // one branch per step; ~6 % of the steps on uniform clouds
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const bool hit = mk[k] <= thr[k];
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
                    const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
                    if (hit) {
                        list[wave][pos] = tile | (half << 7) | (k * 32 + col);
                        thr[k] = fminf(thr[k], mk[k] + CM_BAND);          // a lane only learns of a lower minimum through a hit: tighten here, nowhere else
                    }
                    cnt += __builtin_popcountll(bal);
                }
            }
            if (cnt > CM_LIST - 256) flush();               // a step appends at most 4 x 64
        }
        if (it + 1 < total) {
            commit(it + 1, abuf2[buf ^ 1]);
            fetch(min(it + 2, total - 1));
        }
        buf ^= 1;
    }
    flush();
    // wave-private slots: LDS operations of one wave complete in order, no barrier needed
    for (int u = 0; u < 2; u++) {
        const int ql = u * 64 + lane, q = qw0 + ql;
        if (q < Nq) {
            const int bi = besti[wave * 128 + ql];
            dout[(size_t)b * Nq + q] = bestd[wave * 128 + ql];
            iout[(size_t)b * Nq + q] = bi == 0x7fffffff ? 0 : bi;
        }
    }
}

int l3d_chamfer_forward_mfma(const float *xyz1, const float *xyz2, int B, int N, int M, float *dist1, float *dist2,
                             int32_t *idx1, int32_t *idx2, hipStream_t stream)
{
    const int mx = N > M ? N : M;
    const dim3 grid(l3d_divup(mx, CM_QW), B, 2);
    if ((N < M ? N : M) >= 8192) hipLaunchKernelGGL(chamfer_mfma_kernel<4>, grid, dim3(256), 0, stream, xyz1, xyz2, N, M, dist1, dist2, idx1, idx2);
    else hipLaunchKernelGGL(chamfer_mfma_kernel<1>, grid, dim3(256), 0, stream, xyz1, xyz2, N, M, dist1, dist2, idx1, idx2);
    return l3d_check_launch();
}
