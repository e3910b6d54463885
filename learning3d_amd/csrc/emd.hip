// emd.hip -- approximate Earth Mover's Distance (auction-style soft matching) for gfx950.
//
// Replaces the pybind module `_emd_ext._emd` (losses/cuda/emd_torch/pkg/include/emd.h:47-50):
//   K3 approxmatch      pkg/include/cuda/emd.cuh:7-185    K4 matchcost        emd.cuh:202-244
//   K5 matchcostgrad1   emd.cuh:302-323                   K6 matchcostgrad2   emd.cuh:259-299
//
// The reference runs ONE workgroup per cloud through 10 temperature levels x 3 O(n*m) passes and read-modify-writes
// match [B,m,n] once per level.  Here the same arithmetic -- every per-row sum accumulated in the reference's own
// index order, every product in its own grouping, exp through v_exp_f32 like __expf -- is laid out for a 256-CU part:
//
//   * a pass is a SWEEP kernel over (cloud, row chunk): the rows of one side against the whole partner cloud, staged
//     through LDS as records of TWO consecutive partner points {x0,x1,y0,y1 | z0,z1,a0,a1 | b0,b1}, so each lane
//     evaluates two pairs per packed-fp32 instruction (v_pk_add/mul_f32).  S lanes (1, 2 or 4) share one row: lane q
//     takes records q, q+S, ...; the S lanes then add the 2S products into the row sum IN INDEX ORDER through DPP
//     quad_perm operands (v_add_f32_dpp), every lane of the group redundantly, so the sum's bits are the sequential
//     sum's bits while B*n rows fill 2-4x as many lanes.  Sweeps are stream-ordered launches (a dependent kernel
//     boundary is ~1.5 us on this part; a software grid barrier is 4-7 us and can hang).
//   * pass 3 of level j and pass 1 of level j+1 walk the same (k, all l) pairs: ONE fused sweep evaluates d^2 once and
//     both exponentials.  Pass 3 of the last level only feeds `match`, so it is not swept at all.
//   * match is never read-modify-written: the sweeps keep each level's ratioL_j[k], ratioR_j[l] (80 KB per cloud) and
//     emd_match_kernel writes match[l][k] = sum_j (exp(level_j d^2) * ratioL_j[k]) * ratioR_j[l] ONCE, added in level
//     order (the bits of the reference's `+=` chain), accumulating the transport cost from the same registers.
//
// 1 + 9*2 + 1 sweeps + match + costsum = 22 launches, no host synchronisation (the reference has a
// cudaDeviceSynchronize() inside forward, emd.cuh:197).
#include "common.h"

#define EMD_LEVELS 10
#define EMD_TILE 1024                       // partner points per LDS tile = 512 two-point records
#define EMD_MATCH_LT 64                     // partner points per emd_match_kernel workgroup
#define EMD_LOG2E 1.44269504088896340736f   // 0x3fb8aa3b, the constant __expf multiplies by before v_exp_f32

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float emd_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// sum += the 2S products of one row's S lanes, in ascending partner index: lane q of the group holds w = products of
// partner points (2(tS+q), 2(tS+q)+1).  quad_perm selectors: S=2 -> lanes {0,1},{2,3} of a quad are the two groups.
template <int S>
__device__ __forceinline__ void emd_row_add(float &sum, f32x2 w)
{
    if constexpr (S == 1) {
        sum += w.x; sum += w.y;
    } else if constexpr (S == 2) {
        sum += emd_dpp<0xA0>(w.x); sum += emd_dpp<0xA0>(w.y);      // quad_perm [0,0,2,2]
        sum += emd_dpp<0xF5>(w.x); sum += emd_dpp<0xF5>(w.y);      // quad_perm [1,1,3,3]
    } else {
        sum += emd_dpp<0x00>(w.x); sum += emd_dpp<0x00>(w.y);
        sum += emd_dpp<0x55>(w.x); sum += emd_dpp<0x55>(w.y);
        sum += emd_dpp<0xAA>(w.x); sum += emd_dpp<0xAA>(w.y);
        sum += emd_dpp<0xFF>(w.x); sum += emd_dpp<0xFF>(w.y);
    }
    asm("" : "+v"(sum));               // keeps the SLP vectoriser from pairing two rows' chains into v_mov_dpp + v_pk_add (12 issues for 8)
}

__device__ __forceinline__ float emd_level(int j)            // j = 0..9 <-> the reference's loop variable 7..-2 (emd.cuh:27-31)
{
    return j == EMD_LEVELS - 1 ? 0.0f : -powf(4.0f, (float)(7 - j));
}

struct EmdSweep {
    int nrows, ncols;                  // rows: the side this sweep produces values for; cols: the partner cloud
    const float *rows_xyz, *cols_xyz;  // [B][nrows][3], [B][ncols][3]
    int kind;                          // 0: pass 1 of level 0   1: pass 2 of level j   2: pass 3 of level j + pass 1 of level j+1
    int j;
    float multiL, multiR;
    float *remL, *remR;                // [B][n], [B][m]
    float *ratL, *ratR;                // [B][10][n], [B][10][m]
    int n, m;
};

// rows x partner sweep.  FUSED = two weighted sums per pair (kind 2), else one (kinds 0 and 1).
template <int S, bool FUSED>
__global__ __launch_bounds__(256) void emd_sweep_kernel(const EmdSweep A)
{
    constexpr int R = 256 / S, U = 4, SLACK = 2 * U * S;       // SLACK: records the read-ahead may touch beyond the tile
    __shared__ float4 r0[EMD_TILE / 2 + SLACK];            // x0 x1 y0 y1
    __shared__ float4 r1[EMD_TILE / 2 + SLACK];            // z0 z1 a0 a1
    __shared__ float2 r2[FUSED ? EMD_TILE / 2 + SLACK : 1];   // b0 b1
    const int b = blockIdx.y, tid = threadIdx.x, q = tid % S;
    const int row = blockIdx.x * R + tid / S;
    const bool live = row < A.nrows;
    const float *pr = A.rows_xyz + (size_t)b * A.nrows * 3, *pc = A.cols_xyz + (size_t)b * A.ncols * 3;
    const int n = A.n, m = A.m, j = A.j;

    // partner weights of this sweep
    const float *wa = nullptr, *wb = nullptr;
    float consta = 0.f;
    if (A.kind == 0) consta = A.multiR;                                            // remainR before any consumption
    else if (A.kind == 1) wa = A.ratL + ((size_t)b * EMD_LEVELS + j) * n;          // ratioL_j[k]
    else { wa = A.ratR + ((size_t)b * EMD_LEVELS + j) * m; wb = A.remR + (size_t)b * m; }

    float X = 0.f, Y = 0.f, Z = 0.f, rl = 0.f;
    if (live) {
        X = pr[row * 3]; Y = pr[row * 3 + 1]; Z = pr[row * 3 + 2];
        if (A.kind == 2) rl = A.ratL[((size_t)b * EMD_LEVELS + j) * n + row];
    }
    const f32x2 X2 = {X, X}, Y2 = {Y, Y}, Z2 = {Z, Z}, rl2 = {rl, rl};
    const float level1 = emd_level(j), level2 = emd_level(j + 1 < EMD_LEVELS ? j + 1 : j);

    float s1 = (A.kind == 0) ? 1e-9f : 0.0f;               // pass 1 starts at 1e-9 (emd.cuh:41), passes 2 and 3 at 0
    float s2 = 1e-9f;                                      // FUSED: pass 1 of the next level
    for (int l0 = 0; l0 < A.ncols; l0 += EMD_TILE) {
        const int lend = min(A.ncols, l0 + EMD_TILE) - l0;
        const int lpad = (lend + 4 * S * U - 1) / (4 * S * U) * (4 * S * U);     // zero-weight phantom points: + 0.0f changes no bit
        __syncthreads();
        {
            // the tile's records: every thread's EMD_TILE / 256 points are loaded from clamped indices with NO condition around the
            // loads -- all of them in flight at once, one trip to memory per tile instead of one per point (n = 1024 is one tile per
            // sweep, and with one or two waves per SIMD nobody else covers the trip) -- then zeroed where they are padding
            constexpr int PT = EMD_TILE / 256;
            float x[PT], y[PT], z[PT], a[PT], bw[PT];
#pragma unroll
            for (int it = 0; it < PT; it++) {
                const int gl = l0 + min(tid + 256 * it, lend - 1);
                x[it] = pc[gl * 3]; y[it] = pc[gl * 3 + 1]; z[it] = pc[gl * 3 + 2];
                a[it] = wa ? wa[gl] : consta;
                bw[it] = FUSED ? wb[gl] : 0.f;
            }
#pragma unroll
            for (int it = 0; it < PT; it++) {
                const int l = tid + 256 * it;
                if (l < lpad) {
                    const bool in = l < lend;
                    float *f0 = (float *)r0 + (l >> 1) * 4 + (l & 1), *f1 = (float *)r1 + (l >> 1) * 4 + (l & 1);
                    f0[0] = in ? x[it] : 0.f; f0[2] = in ? y[it] : 0.f; f1[0] = in ? z[it] : 0.f; f1[2] = in ? a[it] : 0.f;
                    if (FUSED) ((float *)r2)[l] = in ? bw[it] : 0.f;
                }
            }
        }
        __syncthreads();
        // uniform trip count (scalar loop control, DPP sees every lane) in whole groups of 2U records per lane: a loop
        // holding convergent DPP operations is not runtime-unrolled by the compiler.  Two register sets ping-pong: the
        // next group's records are read while this group computes (with one wave per SIMD nobody else hides the ds_read
        // latency); the sched_barrier keeps the scheduler from sinking the reads to their first use.
        const int nt = lpad / (2 * S);
        float4 a0[U], a1[U], b0[U], b1[U];
        float2 a2[U], b2[U];
        auto fetch = [&](float4 (&c0)[U], float4 (&c1)[U], float2 (&c2)[U], int t) {      // past the tile's end: slack records, unused
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int p = (t + u) * S + q;
                c0[u] = r0[p]; c1[u] = r1[p];
                if (FUSED) c2[u] = r2[p];
            }
        };
        auto compute = [&](const float4 (&c0)[U], const float4 (&c1)[U], const float2 (&c2)[U]) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const f32x2 dx = (f32x2){c0[u].x, c0[u].y} - X2, dy = (f32x2){c0[u].z, c0[u].w} - Y2,
                            dz = (f32x2){c1[u].x, c1[u].y} - Z2;
                const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;
                const f32x2 t1 = (d2 * level1) * EMD_LOG2E;
                f32x2 w1 = {__builtin_amdgcn_exp2f(t1.x), __builtin_amdgcn_exp2f(t1.y)};
                if (FUSED) w1 = (w1 * rl2) * (f32x2){c1[u].z, c1[u].w};          // emd.cuh:157  __expf(..)*rl*ratioR[l]
                else w1 = w1 * (f32x2){c1[u].z, c1[u].w};                         // emd.cuh:56, 107
                emd_row_add<S>(s1, w1);
                if (FUSED) {
                    const f32x2 t2 = (d2 * level2) * EMD_LOG2E;
                    f32x2 w2 = {__builtin_amdgcn_exp2f(t2.x), __builtin_amdgcn_exp2f(t2.y)};
                    w2 = w2 * (f32x2){c2[u].x, c2[u].y};
                    emd_row_add<S>(s2, w2);
                }
            }
        };
        fetch(a0, a1, a2, 0);
        for (int t0 = 0; t0 < nt; t0 += 2 * U) {
            fetch(b0, b1, b2, t0 + U);
            __builtin_amdgcn_sched_barrier(0);
            compute(a0, a1, a2);
            fetch(a0, a1, a2, t0 + 2 * U);
            __builtin_amdgcn_sched_barrier(0);
            compute(b0, b1, b2);
        }
    }
    if (!live || q != 0) return;
    if (A.kind == 0) {                                     // emd.cuh:62-63 with remainL = multiL
        A.remL[(size_t)b * n + row] = A.multiL;
        A.ratL[((size_t)b * EMD_LEVELS) * n + row] = A.multiL / s1;
    } else if (A.kind == 1) {                              // emd.cuh:113-118
        const float rr = (j == 0) ? A.multiR : A.remR[(size_t)b * m + row];
        const float sumr = s1 * rr;
        const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
        A.ratR[((size_t)b * EMD_LEVELS + j) * m + row] = consumption * rr;
        A.remR[(size_t)b * m + row] = fmaxf(0.0f, rr - sumr);
    } else {                                               // emd.cuh:164-165, then :62-63 of the next level
        const float rem = fmaxf(0.0f, A.remL[(size_t)b * n + row] - s1);
        A.remL[(size_t)b * n + row] = rem;
        A.ratL[((size_t)b * EMD_LEVELS + j + 1) * n + row] = rem / s2;
    }
}

// match[b][l][k] = sum over the 10 levels, in level order, of (exp(level_j d2) * ratioL_j[k]) * ratioR_j[l]   (emd.cuh:157-158)
// and the workgroup's share of cost[b] = sum sqrt(d2) * match (emd.cuh:227-228).  Two k per lane (packed), EMD_MATCH_LT l per
// workgroup; grid (ceil(n/512), ceil(m/LT), B).
__global__ __launch_bounds__(256) void emd_match_kernel(int n, int m, const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2, const float *__restrict__ ratL,
                                                        const float *__restrict__ ratR, float *__restrict__ match,
                                                        float *__restrict__ costpart)
{
    __shared__ float lv[EMD_LEVELS];
    __shared__ float4 pl[EMD_MATCH_LT];
    __shared__ float rr[EMD_MATCH_LT * EMD_LEVELS];
    __shared__ float part[4];
    const int b = blockIdx.z, tid = threadIdx.x;
    const int k0 = blockIdx.x * 512 + tid * 2, l0 = blockIdx.y * EMD_MATCH_LT;
    const int lt = min(EMD_MATCH_LT, m - l0);
    if (tid < EMD_LEVELS) lv[tid] = emd_level(tid);
    if (tid < lt) {
        const float *p2 = xyz2 + ((size_t)b * m + l0 + tid) * 3;
        pl[tid] = make_float4(p2[0], p2[1], p2[2], 0.f);
    }
    for (int i = tid; i < lt * EMD_LEVELS; i += 256) {
        const int li = i / EMD_LEVELS, j = i - li * EMD_LEVELS;
        rr[i] = ratR[((size_t)b * EMD_LEVELS + j) * m + l0 + li];
    }
    const bool v0 = k0 < n, v1 = k0 + 1 < n;
    f32x2 X2 = {0.f, 0.f}, Y2 = X2, Z2 = X2, rl[EMD_LEVELS];
    const float *p1 = xyz1 + (size_t)b * n * 3;
    if (v0) { X2.x = p1[k0 * 3]; Y2.x = p1[k0 * 3 + 1]; Z2.x = p1[k0 * 3 + 2]; }
    if (v1) { X2.y = p1[k0 * 3 + 3]; Y2.y = p1[k0 * 3 + 4]; Z2.y = p1[k0 * 3 + 5]; }
#pragma unroll
    for (int j = 0; j < EMD_LEVELS; j++) {
        const float *r = ratL + ((size_t)b * EMD_LEVELS + j) * n;
        rl[j].x = v0 ? r[k0] : 0.f;
        rl[j].y = v1 ? r[k0 + 1] : 0.f;
    }
    __syncthreads();
    float cost = 0.f;
    float *mrow = match + ((size_t)b * m + l0) * n;
    const bool pair_store = v1 && (n & 1) == 0;            // 8-byte aligned when n is even (k0 is)
    for (int li = 0; li < lt; li++) {
        const float4 c = pl[li];
        const f32x2 dx = (f32x2){c.x, c.x} - X2, dy = (f32x2){c.y, c.y} - Y2, dz = (f32x2){c.z, c.z} - Z2;
        const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;
        f32x2 acc;
#pragma unroll
        for (int j = 0; j < EMD_LEVELS; j++) {
            const f32x2 t = (d2 * lv[j]) * EMD_LOG2E;
            f32x2 w = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
            const float r = rr[li * EMD_LEVELS + j];
            w = (w * rl[j]) * (f32x2){r, r};
            acc = (j == 0) ? w : acc + w;                  // the reference's 0 + w_7 + w_6 + ... chain
        }
        float *o = mrow + (size_t)li * n + k0;
        if (pair_store) *(float2 *)o = make_float2(acc.x, acc.y);
        else { if (v0) o[0] = acc.x; if (v1) o[1] = acc.y; }
        if (v0) cost += sqrtf(d2.x) * acc.x;
        if (v1) cost += sqrtf(d2.y) * acc.y;
    }
    for (int off = 32; off > 0; off >>= 1) cost += __shfl_down(cost, off, 64);
    if ((tid & 63) == 0) part[tid >> 6] = cost;
    __syncthreads();
    if (tid == 0)
        costpart[(size_t)b * gridDim.x * gridDim.y + blockIdx.y * gridDim.x + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// cost[b] = the workgroups' partial sums in a fixed order: deterministic, where the reference (emd.cuh:236-243) reduces a
// 512-entry tree per cloud and an earlier revision here raced fp32 atomics.
__global__ __launch_bounds__(256) void emd_costsum_kernel(int parts, const float *__restrict__ costpart, float *__restrict__ cost)
{
    __shared__ float part[4];
    const int b = blockIdx.x, t = threadIdx.x;
    const float *r = costpart + (size_t)b * parts;
    float s = 0.f;
    for (int l = t; l < parts; l += 256) s += r[l];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((t & 63) == 0) part[t >> 6] = s;
    __syncthreads();
    if (t == 0) cost[b] = (part[0] + part[1]) + (part[2] + part[3]);
}

// grad2[l] = sum_k (p2_l - p1_k) * match[l*n+k] * rsqrt(max(d2, 1e-20))        (K6, emd.cuh:259-299)
// The reference gives thread t of 256 the k = t, t+256, ... of one l, then folds the 256 partials with a stride-doubling tree
// (sum[t] += sum[t+j], j = 1, 2, 4 ..): the same 256 partials and the same pairing here (shfl_down inside a wave is that tree,
// the four waves' results combine as (w0+w1)+(w2+w3)), so the result carries the reference's bits.  EMD_G2_ROWS l per workgroup.
#define EMD_G2_ROWS 8
__global__ __launch_bounds__(256) void emd_grad2_kernel(int n, int m, const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2,
                                                        const float *__restrict__ match,
                                                        float *__restrict__ grad2)
{
    __shared__ float part[EMD_G2_ROWS][4][3];
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l0 = blockIdx.x * EMD_G2_ROWS, rows = min(EMD_G2_ROWS, m - l0);
    const float *p1 = xyz1 + (size_t)b * n * 3;
    for (int r = 0; r < rows; r++) {
        const int l = l0 + r;
        const float *p2 = xyz2 + ((size_t)b * m + l) * 3;
        const float *mt = match + ((size_t)b * m + l) * n;
        const float x2 = p2[0], y2 = p2[1], z2 = p2[2];
        float gx = 0, gy = 0, gz = 0;
        for (int k = t; k < n; k += 256) {
            const float dx = x2 - p1[k * 3], dy = y2 - p1[k * 3 + 1], dz = z2 - p1[k * 3 + 2];
            const float d = mt[k] * rsqrtf(fmaxf((dx * dx + dy * dy) + dz * dz, 1e-20f));
            gx += dx * d; gy += dy * d; gz += dz * d;
        }
        for (int off = 1; off < 64; off <<= 1) {
            gx += __shfl_down(gx, off, 64); gy += __shfl_down(gy, off, 64); gz += __shfl_down(gz, off, 64);
        }
        if (lane == 0) { part[r][wave][0] = gx; part[r][wave][1] = gy; part[r][wave][2] = gz; }
    }
    __syncthreads();
    if (t < rows * 3) {
        const int r = t / 3, c = t - r * 3;
        grad2[((size_t)b * m + l0 + r) * 3 + c] = (part[r][0][c] + part[r][1][c]) + (part[r][2][c] + part[r][3][c]);
    }
}

// grad1[k] = sum_l (p1_k - p2_l) * match[l*n+k] * rsqrt(max(d2, 1e-20))        (K5, emd.cuh:302-323)
// The reference adds the m terms of one k sequentially in one thread.  Four lanes share a k here (lane q takes l = 4i + q), the
// terms are added in l order through DPP operands by all four (as in the forward sweeps): the sequential sum's bits, four times
// the lanes, and EMD_G1_U match values per lane in flight while the previous EMD_G1_U are consumed.
#define EMD_G1_U 8
__global__ __launch_bounds__(256) void emd_grad1_kernel(int n, int m, const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2,
                                                        const float *__restrict__ match,
                                                        float *__restrict__ grad1)
{
    extern __shared__ float4 g1_p2[];                      // partner cloud, padded to a multiple of 8 * EMD_G1_U points
    constexpr int U = EMD_G1_U;
    const int b = blockIdx.y, t = threadIdx.x, q = t & 3;
    const int k = blockIdx.x * 64 + (t >> 2);
    const bool live = k < n;
    const int mpad = (m + 8 * U - 1) / (8 * U) * (8 * U);
    const float *p2 = xyz2 + (size_t)b * m * 3;
    for (int l = t; l < mpad + 4 * U; l += 256)
        g1_p2[l] = l < m ? make_float4(p2[l * 3], p2[l * 3 + 1], p2[l * 3 + 2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    float x1 = 0, y1 = 0, z1 = 0;
    if (live) { const float *p1 = xyz1 + ((size_t)b * n + k) * 3; x1 = p1[0]; y1 = p1[1]; z1 = p1[2]; }
    const float *mt = match + (size_t)b * n * m + (live ? k : 0);
    __syncthreads();
    float gx = 0, gy = 0, gz = 0;
    float ma[U], mb[U];
    auto fetch = [&](float (&v)[U], int l0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int l = l0 + 4 * u + q;
            v[u] = mt[(size_t)min(l, m - 1) * n];          // unconditional (no branch per row, the waits stay counted); consume() zeroes rows past m
        }
    };
    auto consume = [&](const float (&v)[U], int l0) {      // rows past m: zero terms, + 0.0f changes no bit of a sum that started at +0
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float4 c = g1_p2[l0 + 4 * u + q];
            const float dx = x1 - c.x, dy = y1 - c.y, dz = z1 - c.z;
            const float mv = l0 + 4 * u + q < m ? v[u] : 0.f;
            const float d = mv * rsqrtf(fmaxf((dx * dx + dy * dy) + dz * dz, 1e-20f));
            const float tx = dx * d, ty = dy * d, tz = dz * d;
            gx += emd_dpp<0x00>(tx); gy += emd_dpp<0x00>(ty); gz += emd_dpp<0x00>(tz);
            gx += emd_dpp<0x55>(tx); gy += emd_dpp<0x55>(ty); gz += emd_dpp<0x55>(tz);
            gx += emd_dpp<0xAA>(tx); gy += emd_dpp<0xAA>(ty); gz += emd_dpp<0xAA>(tz);
            gx += emd_dpp<0xFF>(tx); gy += emd_dpp<0xFF>(ty); gz += emd_dpp<0xFF>(tz);
        }
    };
    fetch(ma, 0);
    for (int l0 = 0; l0 < mpad; l0 += 8 * U) {
        fetch(mb, l0 + 4 * U);
        __builtin_amdgcn_sched_barrier(0);
        consume(ma, l0);
        fetch(ma, l0 + 8 * U);
        __builtin_amdgcn_sched_barrier(0);
        consume(mb, l0 + 4 * U);
    }
    if (live && q == 0) {
        float *o = grad1 + ((size_t)b * n + k) * 3;
        o[0] = gx; o[1] = gy; o[2] = gz;
    }
}

static inline int emd_cost_parts(int n, int m) { return l3d_divup(n, 512) * l3d_divup(m, EMD_MATCH_LT); }

extern "C" size_t l3d_emd_workspace_bytes(int B, int n, int m)
{
    if (B <= 0 || n <= 0 || m <= 0) return 0;
    return sizeof(float) * (size_t)B * ((size_t)(EMD_LEVELS + 1) * ((size_t)n + m) + emd_cost_parts(n, m));
}

template <int S>
static void emd_launch_sweep(const EmdSweep &a, int B, hipStream_t st)
{
    const dim3 grid(l3d_divup(a.nrows, 256 / S), B);
    if (a.kind == 2) hipLaunchKernelGGL((emd_sweep_kernel<S, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((emd_sweep_kernel<S, false>), grid, dim3(256), 0, st, a);
}

extern "C" int l3d_emd_forward(const float *xyz1, const float *xyz2, int B, int n, int m, float *match,
                               float *cost, void *workspace, int split, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz1 && xyz2 && match && cost && workspace && B > 0 && n > 0 && m > 0);
    L3D_REQUIRE(split == 0 || split == 1 || split == 2 || split == 4);
    L3D_REQUIRE(B <= 65535 && (size_t)n * 3 < (1u << 31) && (size_t)m * 3 < (1u << 31));
    hipStream_t st = (hipStream_t)stream;
    if (split == 0) {                                      // lanes per row: enough waves for every SIMD of the part (1024)
        const long rows = (long)B * (n < m ? n : m);
        split = rows >= 65536 ? 1 : rows >= 32768 ? 2 : 4;         // measured: profiles/round5_emd_bench.txt
    }
    float *ws = (float *)workspace;
    EmdSweep a;
    a.n = n; a.m = m;
    a.remL = ws;                          ws += (size_t)B * n;
    a.remR = ws;                          ws += (size_t)B * m;
    a.ratL = ws;                          ws += (size_t)B * EMD_LEVELS * n;
    a.ratR = ws;                          ws += (size_t)B * EMD_LEVELS * m;
    float *costpart = ws;
    if (n >= m) { a.multiL = 1; a.multiR = (float)(n / m); } else { a.multiL = (float)(m / n); a.multiR = 1; }   // emd.cuh:9-16
    auto sweep = [&](int kind, int j) {
        a.kind = kind; a.j = j;
        if (kind == 1) { a.nrows = m; a.ncols = n; a.rows_xyz = xyz2; a.cols_xyz = xyz1; }
        else           { a.nrows = n; a.ncols = m; a.rows_xyz = xyz1; a.cols_xyz = xyz2; }
        if (split == 1) emd_launch_sweep<1>(a, B, st);
        else if (split == 2) emd_launch_sweep<2>(a, B, st);
        else emd_launch_sweep<4>(a, B, st);
    };
    sweep(0, 0);
    for (int j = 0; j < EMD_LEVELS; j++) {
        sweep(1, j);
        if (j + 1 < EMD_LEVELS) sweep(2, j);
    }
    int rc = l3d_check_launch();
    if (rc) return rc;
    const int parts = emd_cost_parts(n, m);
    hipLaunchKernelGGL(emd_match_kernel, dim3(l3d_divup(n, 512), l3d_divup(m, EMD_MATCH_LT), B), dim3(256), 0, st, n, m, xyz1,
                       xyz2, (const float *)a.ratL, (const float *)a.ratR, match, costpart);
    hipLaunchKernelGGL(emd_costsum_kernel, dim3(B), dim3(256), 0, st, parts, (const float *)costpart, cost);
    return l3d_check_launch();
}

extern "C" int l3d_emd_backward(const float *xyz1, const float *xyz2, const float *match, int B, int n,
                                int m, float *grad1, float *grad2, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz1 && xyz2 && match && grad1 && grad2 && B > 0 && n > 0 && m > 0);
    hipStream_t st = (hipStream_t)stream;
    L3D_REQUIRE(B <= 65535);
    const int mpad = l3d_divup(m, 8 * EMD_G1_U) * 8 * EMD_G1_U + 4 * EMD_G1_U;
    const size_t lds = (size_t)mpad * sizeof(float4);
    if (lds > 160 * 1024) return L3D_ERR_UNSUPPORTED;     // m <= 10 176 partner points (the reference's own kernels stop at what fits their grid)
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)emd_grad1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return L3D_ERR_LAUNCH;
    hipLaunchKernelGGL(emd_grad1_kernel, dim3(l3d_divup(n, 64), B), dim3(256), lds, st, n, m, xyz1, xyz2, match, grad1);
    hipLaunchKernelGGL(emd_grad2_kernel, dim3(l3d_divup(m, EMD_G2_ROWS), B), dim3(256), 0, st, n, m, xyz1, xyz2, match, grad2);
    return l3d_check_launch();
}
