// chamfer.hip -- Chamfer nearest-neighbour search (fwd) and gradient (bwd) for gfx950.
//
// Replaces the pybind module `cd` of losses/cuda/chamfer_distance/ :
//   K1  ChamferDistanceKernel      chamfer_distance.cu:6-137   (<<<dim3(32,16),512>>> x2)
//   K2  ChamferDistanceGradKernel  chamfer_distance.cu:158-187 (fp32 atomicAdd scatter)
// and matches the CPU twin `nnsearch` (chamfer_distance.cpp:59-87) bit for bit:
//   d = (dx*dx + dy*dy) + dz*dz with dx = p2 - p1, no contraction, strict '<'.
//
// Design: both directions in ONE launch (blockIdx.z); a 4-wave workgroup owns 64 (or 128) queries,
// one (or two) per lane, and the 4 waves split every LDS tile of the other cloud between them
// (4 waves/SIMD hide the LDS latency; the (min, argmin) pairs are merged once through LDS).
// Candidates are float4 tiles read back as wave-uniform ds_read_b128 broadcasts, 8 in flight.  No running-min merge through global memory (the reference merges
// 512-point chunks through `result[]`, .cu:129-132): the running (min, argmin) lives in VGPRs.
// Backward is deterministic: every output point owns its sum (direct term + an index-ordered
// scan for the points that selected it), reproducing the CPU reference's accumulation order
// (chamfer_distance.cpp:138-176) instead of racing fp32 atomics.
#include "common.h"

#define CTILE 2048      // candidates per LDS tile, shared by the 4 waves (32 KiB)
#define CWAVES 4        // waves per workgroup: each scans 1/4 of every tile for the SAME 64*QPL queries
#define CUNROLL 8
#ifndef CH_SMALL_TILE
#define CH_SMALL_TILE 0      // 1: 1024-candidate tiles when neither cloud is longer (LABLOG R6.2: co-resident with the kNN kernel, and slower)
#endif
#define CHAMFER_LL_BLOCKS 64                                  // l3d_chamfer_loss_local_mb's workgroups
#define CHAMFER_LL_WS_BYTES (16 + CHAMFER_LL_BLOCKS * 16)

template <int QPL>   // queries per lane
__global__ __launch_bounds__(64 * CWAVES) void chamfer_fwd_kernel(const float *__restrict__ xyz1,
                                                                  const float *__restrict__ xyz2, int N,
                                                                  int M, float *__restrict__ dist1,
                                                                  float *__restrict__ dist2,
                                                                  int32_t *__restrict__ idx1,
                                                                  int32_t *__restrict__ idx2)
{
    __shared__ float4 cand[CTILE];
    __shared__ float rbest[CWAVES][QPL][64];
    __shared__ int rbesti[CWAVES][QPL][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int dir = blockIdx.z;
    const float *qs = dir == 0 ? xyz1 : xyz2;
    const float *cs = dir == 0 ? xyz2 : xyz1;
    const int Nq = dir == 0 ? N : M;
    const int Nc = dir == 0 ? M : N;
    float *dout = dir == 0 ? dist1 : dist2;
    int32_t *iout = dir == 0 ? idx1 : idx2;
    const int q0 = blockIdx.x * (64 * QPL);
    if (q0 >= Nq) return;            // whole block out of range (grid is sized for max(N,M))

    float qx[QPL], qy[QPL], qz[QPL], best[QPL];
    int besti[QPL];
#pragma unroll
    for (int u = 0; u < QPL; u++) {
        int q = min(q0 + u * 64 + lane, Nq - 1);
        const float *p = qs + ((size_t)b * Nq + q) * 3;
        qx[u] = p[0]; qy[u] = p[1]; qz[u] = p[2];
        best[u] = INFINITY; besti[u] = 0x7fffffff;
    }
    const float *cbase = cs + (size_t)b * Nc * 3;
    for (int c0 = 0; c0 < Nc; c0 += CTILE) {
        const int tn = min(CTILE, Nc - c0);
        __syncthreads();
        l3d_stage_points<8>(cbase + (size_t)c0 * 3, tn, tid, 64 * CWAVES, [&](int t, float x, float y, float z) { cand[t] = make_float4(x, y, z, 0.f); });
        __syncthreads();
        // this wave's slice of the tile
        const int per = (tn + CWAVES - 1) / CWAVES;
        const int t0 = wave * per, t1 = min(tn, t0 + per);
        int t = t0;
        for (; t + CUNROLL <= t1; t += CUNROLL) {
            float4 c[CUNROLL];
#pragma unroll
            for (int v = 0; v < CUNROLL; v++) c[v] = cand[t + v];
#pragma unroll
            for (int v = 0; v < CUNROLL; v++)
#pragma unroll
                for (int u = 0; u < QPL; u++) {
                    const float dx = c[v].x - qx[u], dy = c[v].y - qy[u], dz = c[v].z - qz[u];
                    const float d = (dx * dx + dy * dy) + dz * dz;
                    const bool lt = d < best[u];        // ascending index inside a slice: strict '<'
                    best[u] = lt ? d : best[u];
                    besti[u] = lt ? c0 + t + v : besti[u];
                }
        }
        for (; t < t1; t++) {
            const float4 c = cand[t];
#pragma unroll
            for (int u = 0; u < QPL; u++) {
                const float dx = c.x - qx[u], dy = c.y - qy[u], dz = c.z - qz[u];
                const float d = (dx * dx + dy * dy) + dz * dz;
                const bool lt = d < best[u];
                best[u] = lt ? d : best[u];
                besti[u] = lt ? c0 + t : besti[u];
            }
        }
    }
    // combine the 4 waves: smaller d wins, equal d -> lower index (what one sequential strict-'<'
    // scan over all candidates returns).  NaN/inf-only rows keep index 0 like the reference.
#pragma unroll
    for (int u = 0; u < QPL; u++) { rbest[wave][u][lane] = best[u]; rbesti[wave][u][lane] = besti[u]; }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int u = 0; u < QPL; u++) {
            float bv = rbest[0][u][lane];
            int bi = rbesti[0][u][lane];
#pragma unroll
            for (int w = 1; w < CWAVES; w++) {
                const float ov = rbest[w][u][lane];
                const int oi = rbesti[w][u][lane];
                const bool take = ov < bv || (ov == bv && oi < bi);
                bv = take ? ov : bv;
                bi = take ? oi : bi;
            }
            const int q = q0 + u * 64 + lane;
            if (q < Nq) {
                dout[(size_t)b * Nq + q] = bv;
                iout[(size_t)b * Nq + q] = bi == 0x7fffffff ? 0 : bi;
            }
        }
    }
}

// The loss tail of chamfer_fwd_packed_kernel<true>, called by wave 0 of every workgroup (grid (x, B, 2): direction z's slots are
// [z * gridDim.x * gridDim.y, ...)).  ws: 16 bytes (ticket, zero before the first launch) + one double per workgroup.
__device__ __forceinline__ void chamfer_loss_tail(double sq, unsigned *__restrict__ ws, double *__restrict__ partial,
                                                  float *__restrict__ loss, int N, int M)
{
    const int lane = threadIdx.x & 63;
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
    double *slots = (double *)(ws + 4);
    const unsigned per_dir = gridDim.x * gridDim.y, total = 2 * per_dir;
    const unsigned slot = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    unsigned ticket = 0;
    if (lane == 0) {
        __hip_atomic_store(&slots[slot], sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       // ... and acknowledged
        ticket = __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != total - 1) return;
    double t[2];
#pragma unroll
    for (int z = 0; z < 2; z++) {                      // lane l adds slots l, l + 64, ... of the direction, then a fixed shuffle tree
        double a = 0.0;
        for (unsigned i = lane; i < per_dir; i += 64) a += __hip_atomic_load(&slots[z * per_dir + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
        t[z] = a;
    }
    if (lane == 0) {
        const double n1 = (double)gridDim.y * N, n2 = (double)gridDim.y * M;
        partial[0] = t[0]; partial[1] = t[1]; partial[2] = n1; partial[3] = n2;
        loss[0] = (float)((t[0] / n1 + t[1] / n2) / 2.0);
        __hip_atomic_store(ws, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                 // re-armed for the next launch on this stream
    }
}

// kernel choice is an ARGUMENT of l3d_chamfer_forward_variant (0 = per-candidate kernels only, 1 = auto, 2 = packed
// kernel always); there is no process-wide state.  l3d_chamfer_forward == variant 1.

// ---------------------------------------------------------------------------------------------
// Packed variant: two queries per lane evaluated with ONE packed-fp32 instruction each step (v_pk_add_f32 /
// v_pk_mul_f32: the 157 TFLOP/s vector peak of this part exists only in packed form), and the argmin kept
// per CHUNK of 8 candidates instead of per candidate: min3 tree over the chunk, one strict '<' against the
// running best, remember the chunk's first index.  Which of the 8 it was is resolved once at the end by
// re-evaluating that chunk (same arithmetic -> the same bits) and taking the first candidate whose distance
// equals the best: exactly what one sequential strict-'<' scan returns.  Per (query, candidate): 4 packed + ~1
// instructions instead of 11.  Same tiling / wave split / merge as chamfer_fwd_kernel.
// ---------------------------------------------------------------------------------------------
// LOSS (its own instantiation): the loss tail of losses/chamfer_distance.py:38-40 in the same launch.  Wave 0 of every workgroup adds
// the square roots of its 128 final distances in fp64 and publishes the sum in its slot of `ws` with a WRITE-THROUGH store (sc1),
// waits for it (s_waitcnt vmcnt(0)) and draws a ticket (agent-scope atomic); the workgroup that draws the last ticket adds the
// slots in a fixed order and writes partial[4] = (sum sqrt d1, sum sqrt d2, n1, n2) and the loss, and re-arms the ticket.  Round 3
// tried this with __threadfence() in front of the ticket and lost 4 us per step (LABLOG R3.10): an agent-scope release fence
// writes the XCD's whole L2 back; a write-through store of the 8 bytes that matter does not.  The separate loss kernel was 7 us
// on the Chamfer branch's critical path in front of EdgeConv (launch + L2 turn-around for ~1 us of work).
// PTILE: candidates per LDS tile, 2048 in the product.  With 1024 (20 KB with the merge arrays; -DCH_SMALL_TILE=1) a workgroup fits on
// a CU BESIDE two of knn_mfma_kernel's (2 x 69.6 KB at N = 1024), which the 36 KB of a 2048-candidate tile does not -- the two kernels
// of the bench step's front, launched side by side on two streams, take turns on a CU (24.5 + 16.7 us alone, 39.5 together).  Measured
// (LABLOG R6.2): co-resident, the step is 1.3 % SLOWER (281.5 vs 277.9 us, four interleaved pairs on one box) -- the kNN kernel is the
// front's critical path and the Chamfer waves beside it take its issue slots.  Not the default.
template <bool LOSS, int PTILE>
__global__ __launch_bounds__(64 * CWAVES) void chamfer_fwd_packed_kernel(const float *__restrict__ xyz1,
                                                                         const float *__restrict__ xyz2, int N, int M,
                                                                         float *__restrict__ dist1,
                                                                         float *__restrict__ dist2,
                                                                         int32_t *__restrict__ idx1,
                                                                         int32_t *__restrict__ idx2,
                                                                         unsigned *__restrict__ ws, double *__restrict__ partial,
                                                                         float *__restrict__ loss)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ float4 cand[PTILE];
    __shared__ float rbest[CWAVES][2][64];
    __shared__ int rbesti[CWAVES][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int dir = blockIdx.z;
    const float *qs = dir == 0 ? xyz1 : xyz2;
    const float *cs = dir == 0 ? xyz2 : xyz1;
    const int Nq = dir == 0 ? N : M;
    const int Nc = dir == 0 ? M : N;
    float *dout = dir == 0 ? dist1 : dist2;
    int32_t *iout = dir == 0 ? idx1 : idx2;
    const int q0 = blockIdx.x * 128;
    if (!LOSS && q0 >= Nq) return;
    if (LOSS && q0 >= Nq) {                                // a workgroup beyond the shorter cloud still owns a slot and a ticket
        if (wave == 0) chamfer_loss_tail(0.0, ws, partial, loss, N, M);
        return;
    }

    f32x2 qx, qy, qz;
    {
        const float *p0 = qs + ((size_t)b * Nq + min(q0 + lane, Nq - 1)) * 3;
        const float *p1 = qs + ((size_t)b * Nq + min(q0 + 64 + lane, Nq - 1)) * 3;
        qx = f32x2{p0[0], p1[0]}; qy = f32x2{p0[1], p1[1]}; qz = f32x2{p0[2], p1[2]};
    }
    float best0 = INFINITY, best1 = INFINITY;
    int base0 = 0x7fffffff, base1 = 0x7fffffff;        // first candidate index of the chunk that holds the best
    const float *cbase = cs + (size_t)b * Nc * 3;
    for (int c0 = 0; c0 < Nc; c0 += PTILE) {
        const int tn = min(PTILE, Nc - c0);
        __syncthreads();
        {
            // the tile's loads all in flight before the first LDS write (a rolled load / wait / write loop is one round trip to
            // memory per 256 points: four of them in a row at N = 1024, a third of this kernel's 16 us): unconditional loads from
            // clamped indices (a uniform test around a load is still a branch with a wait at its join); only the LDS writes are predicated
            constexpr int NS = PTILE / (64 * CWAVES);
            float sx[NS], sy[NS], sz[NS];
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const float *cp = cbase + (size_t)(c0 + min(tid + i * 64 * CWAVES, tn - 1)) * 3;
                sx[i] = cp[0]; sy[i] = cp[1]; sz[i] = cp[2];
            }
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const int t = tid + i * 64 * CWAVES;
                if (t < tn) cand[t] = make_float4(sx[i], sy[i], sz[i], 0.f);
            }
        }
        __syncthreads();
        const int per = (tn + CWAVES - 1) / CWAVES;
        const int t0 = wave * per, t1 = min(tn, t0 + per);
        int t = t0;
        for (; t + 8 <= t1; t += 8) {
            float4 c[8];
#pragma unroll
            for (int v = 0; v < 8; v++) c[v] = cand[t + v];
            f32x2 d[8];
#pragma unroll
            for (int v = 0; v < 8; v++) {
                const f32x2 dx = f32x2{c[v].x, c[v].x} - qx, dy = f32x2{c[v].y, c[v].y} - qy, dz = f32x2{c[v].z, c[v].z} - qz;
                d[v] = (dx * dx + dy * dy) + dz * dz;
            }
            float m0 = fminf(fminf(d[0][0], d[1][0]), d[2][0]), m1 = fminf(fminf(d[0][1], d[1][1]), d[2][1]);
            m0 = fminf(fminf(m0, d[3][0]), d[4][0]); m1 = fminf(fminf(m1, d[3][1]), d[4][1]);
            m0 = fminf(fminf(m0, d[5][0]), d[6][0]); m1 = fminf(fminf(m1, d[5][1]), d[6][1]);
            m0 = fminf(m0, d[7][0]); m1 = fminf(m1, d[7][1]);
            const bool lt0 = m0 < best0, lt1 = m1 < best1;           // strict: an equal later chunk does not replace
            best0 = lt0 ? m0 : best0; base0 = lt0 ? c0 + t : base0;
            best1 = lt1 ? m1 : best1; base1 = lt1 ? c0 + t : base1;
        }
        for (; t < t1; t++) {                                       // slice tail: chunks of one
            const float4 c = cand[t];
            const f32x2 dx = f32x2{c.x, c.x} - qx, dy = f32x2{c.y, c.y} - qy, dz = f32x2{c.z, c.z} - qz;
            const f32x2 d = (dx * dx + dy * dy) + dz * dz;
            const bool lt0 = d[0] < best0, lt1 = d[1] < best1;
            best0 = lt0 ? d[0] : best0; base0 = lt0 ? c0 + t : base0;
            best1 = lt1 ? d[1] : best1; base1 = lt1 ? c0 + t : base1;
        }
    }
    // which candidate of the winning chunk: the first whose distance equals the best (a chunk of one matches at once)
    int bi0 = base0, bi1 = base1;
    {
        bool f0 = base0 == 0x7fffffff, f1 = base1 == 0x7fffffff;
#pragma unroll
        for (int v = 0; v < 8; v++) {
            const int j0 = min(base0 == 0x7fffffff ? 0 : base0 + v, Nc - 1), j1 = min(base1 == 0x7fffffff ? 0 : base1 + v, Nc - 1);
            const float *pa = cbase + (size_t)j0 * 3, *pb = cbase + (size_t)j1 * 3;
            const f32x2 dx = f32x2{pa[0], pb[0]} - qx, dy = f32x2{pa[1], pb[1]} - qy, dz = f32x2{pa[2], pb[2]} - qz;
            const f32x2 d = (dx * dx + dy * dy) + dz * dz;
            if (!f0 && d[0] == best0) { bi0 = j0; f0 = true; }
            if (!f1 && d[1] == best1) { bi1 = j1; f1 = true; }
        }
    }
    rbest[wave][0][lane] = best0; rbesti[wave][0][lane] = bi0;
    rbest[wave][1][lane] = best1; rbesti[wave][1][lane] = bi1;
    __syncthreads();
    if (wave == 0) {
        double sq = 0.0;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            float bv = rbest[0][u][lane];
            int bi = rbesti[0][u][lane];
#pragma unroll
            for (int w = 1; w < CWAVES; w++) {
                const float ov = rbest[w][u][lane];
                const int oi = rbesti[w][u][lane];
                const bool take = ov < bv || (ov == bv && oi < bi);
                bv = take ? ov : bv;
                bi = take ? oi : bi;
            }
            const int q = q0 + u * 64 + lane;
            if (q < Nq) {
                dout[(size_t)b * Nq + q] = bv;
                iout[(size_t)b * Nq + q] = bi == 0x7fffffff ? 0 : bi;
                if (LOSS) sq += (double)sqrtf(bv);
            }
        }
        if (LOSS) chamfer_loss_tail(sq, ws, partial, loss, N, M);
    }
}

// chamfer_mfma.hip: candidates ranked on the fp16 matrix cores, exact by refinement (same bits out)
int l3d_chamfer_forward_mfma(const float *xyz1, const float *xyz2, int B, int N, int M, float *dist1, float *dist2,
                             int32_t *idx1, int32_t *idx2, hipStream_t stream);

extern "C" int l3d_chamfer_forward_variant(const float *xyz1, const float *xyz2, int B, int N, int M,
                                           float *dist1, float *dist2, int32_t *idx1, int32_t *idx2,
                                           int variant, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz1 && xyz2 && dist1 && dist2 && idx1 && idx2 && B > 0 && N > 0 && M > 0 && variant >= 0 && variant <= 3);
    L3D_REQUIRE(B <= 65535);
    const int l3d_chamfer_forward_mode = variant;
    const int mx = N > M ? N : M;
    // large clouds (from 4096 x 4096 pairs per cloud: 81 vs 96 us at B 16, 2.0 vs 4.9 ms at config 4; below that the per-pair kernel
    // wins, profiles/round5_chamfer_bench.txt): the matrix-core ranking + exact refinement; variant 3 forces it, 0 and 2 never use it
    // ... and only with enough workgroups for the part: 2 ceil(max / 512) B of them, each re-scanning every candidate for its bounding
    // box -- a single 4096 x 4096 pair is 16 workgroups on 256 CUs, where the per-pair kernel's 64 x B x 2 do better
    if (variant == 3 || (variant == 1 && (long)N * M >= (1L << 24) && (long)2 * l3d_divup(mx, 512) * B >= 128))
        return l3d_chamfer_forward_mfma(xyz1, xyz2, B, N, M, dist1, dist2, idx1, idx2, (hipStream_t)stream);
    // two queries per lane, packed fp32 (chamfer_fwd_packed_kernel) once that still gives every SIMD a wave;
    // tiny problems keep one query per lane for the workgroup count
    const long wgs2 = (long)l3d_divup(mx, 128) * B * 2;
    if (l3d_chamfer_forward_mode == 2 || (l3d_chamfer_forward_mode == 1 && wgs2 * CWAVES >= 1024)) {
        dim3 grid(l3d_divup(mx, 128), B, 2);
        if (mx <= 1024 && CH_SMALL_TILE)
            hipLaunchKernelGGL((chamfer_fwd_packed_kernel<false, 1024>), grid, dim3(64 * CWAVES), 0, (hipStream_t)stream, xyz1, xyz2, N, M,
                               dist1, dist2, idx1, idx2, (unsigned *)nullptr, (double *)nullptr, (float *)nullptr);
        else
            hipLaunchKernelGGL((chamfer_fwd_packed_kernel<false, CTILE>), grid, dim3(64 * CWAVES), 0, (hipStream_t)stream, xyz1, xyz2, N, M,
                               dist1, dist2, idx1, idx2, (unsigned *)nullptr, (double *)nullptr, (float *)nullptr);
    } else if (wgs2 * 2 >= 8192) {
        dim3 grid(l3d_divup(mx, 128), B, 2);
        hipLaunchKernelGGL(chamfer_fwd_kernel<2>, grid, dim3(64 * CWAVES), 0, (hipStream_t)stream, xyz1,
                           xyz2, N, M, dist1, dist2, idx1, idx2);
    } else {
        dim3 grid(l3d_divup(mx, 64), B, 2);
        hipLaunchKernelGGL(chamfer_fwd_kernel<1>, grid, dim3(64 * CWAVES), 0, (hipStream_t)stream, xyz1,
                           xyz2, N, M, dist1, dist2, idx1, idx2);
    }
    return l3d_check_launch();
}

extern "C" int l3d_chamfer_forward(const float *xyz1, const float *xyz2, int B, int N, int M,
                                   float *dist1, float *dist2, int32_t *idx1, int32_t *idx2,
                                   l3d_stream_t stream)
{
    return l3d_chamfer_forward_variant(xyz1, xyz2, B, N, M, dist1, dist2, idx1, idx2, 1, stream);
}

extern "C" int l3d_chamfer_loss_local_mb(const float *dist1, const float *dist2, int B, int N, int M, void *ws, double *partial, float *loss,
                                         l3d_stream_t stream);

// Search + loss tail (losses/chamfer_distance.py:34-43 for one rank: the NN search of both directions, then
// (mean sqrt d1 + mean sqrt d2) / 2) -- ONE launch where the packed kernel serves the search (the shapes of configs 2 / 3), the
// search and l3d_chamfer_loss_local_mb otherwise.  partial [4] fp64 = (sum sqrt d1, sum sqrt d2, B N, B M): what a multi-GPU run
// all-gathers.  ws: l3d_chamfer_forward_loss_ws_bytes(B, N, M) bytes, the first 16 zero before the first call (the kernel re-arms it).
extern "C" size_t l3d_chamfer_forward_loss_ws_bytes(int B, int N, int M)
{
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    const size_t fused = 16 + (size_t)2 * B * l3d_divup(N > M ? N : M, 128) * sizeof(double);
    return fused > CHAMFER_LL_WS_BYTES ? fused : CHAMFER_LL_WS_BYTES;
}

extern "C" int l3d_chamfer_forward_loss(const float *xyz1, const float *xyz2, int B, int N, int M, float *dist1, float *dist2,
                                        int32_t *idx1, int32_t *idx2, void *ws, double *partial, float *loss, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz1 && xyz2 && dist1 && dist2 && idx1 && idx2 && ws && partial && loss && B > 0 && N > 0 && M > 0);
    L3D_REQUIRE(B <= 65535);
    const int mx = N > M ? N : M;
    const long wgs2 = (long)l3d_divup(mx, 128) * B * 2;
    if ((long)N * M < (1L << 24) && wgs2 * CWAVES >= 1024) {
        dim3 grid(l3d_divup(mx, 128), B, 2);
        if (mx <= 1024 && CH_SMALL_TILE)
            hipLaunchKernelGGL((chamfer_fwd_packed_kernel<true, 1024>), grid, dim3(64 * CWAVES), 0, (hipStream_t)stream, xyz1, xyz2, N, M, dist1,
                               dist2, idx1, idx2, (unsigned *)ws, partial, loss);
        else
            hipLaunchKernelGGL((chamfer_fwd_packed_kernel<true, CTILE>), grid, dim3(64 * CWAVES), 0, (hipStream_t)stream, xyz1, xyz2, N, M, dist1,
                               dist2, idx1, idx2, (unsigned *)ws, partial, loss);
        return l3d_check_launch();
    }
    const int rc = l3d_chamfer_forward_variant(xyz1, xyz2, B, N, M, dist1, dist2, idx1, idx2, 1, stream);
    return rc ? rc : l3d_chamfer_loss_local_mb(dist1, dist2, B, N, M, ws, partial, loss, stream);
}

// ---------------------------------------------------------------------------------------------
// backward.  For output point i of cloud A (partner cloud C, A's own nn index ia, C's nn index ic):
//   which == 0 (A = xyz1):  g = 0 + 2*gdA[i]*(a_i - c_ia[i])           (loop 1 of the CPU code)
//                           then for j ascending with ic[j] == i:  g -= 2*gdC[j]*(c_j - a_i)
//   which == 1 (A = xyz2):  g = 0; for j ascending with ic[j] == i: g -= 2*gdC[j]*(c_j - a_i)
//                           then g += 2*gdA[i]*(a_i - c_ia[i])          (loop 2 comes second)
// which is the exact accumulation order of chamfer_distance.cpp:138-176.
// ---------------------------------------------------------------------------------------------
#define BTILE 4096
__global__ __launch_bounds__(64) void chamfer_bwd_kernel(const float *__restrict__ xyz1,
                                                         const float *__restrict__ xyz2, int N,
                                                         int M, const float *__restrict__ gd1,
                                                         const float *__restrict__ gd2,
                                                         const int32_t *__restrict__ idx1,
                                                         const int32_t *__restrict__ idx2,
                                                         float *__restrict__ g1,
                                                         float *__restrict__ g2)
{
    __shared__ int32_t sel[BTILE];
    const int lane = threadIdx.x;
    const int b = blockIdx.y;
    const int which = blockIdx.z;
    const float *A = which == 0 ? xyz1 : xyz2;
    const float *C = which == 0 ? xyz2 : xyz1;
    const int NA = which == 0 ? N : M;
    const int NC = which == 0 ? M : N;
    const float *gdA = which == 0 ? gd1 : gd2;
    const float *gdC = which == 0 ? gd2 : gd1;
    const int32_t *ia = which == 0 ? idx1 : idx2;
    const int32_t *ic = which == 0 ? idx2 : idx1;
    float *gout = which == 0 ? g1 : g2;
    if (blockIdx.x * 64 >= NA) return;
    const int i = blockIdx.x * 64 + lane;
    const bool valid = i < NA;
    const int iq = valid ? i : NA - 1;
    const float *ap = A + ((size_t)b * NA + iq) * 3;
    const float ax = ap[0], ay = ap[1], az = ap[2];
    // direct term
    const int j2 = ia[(size_t)b * NA + iq];
    const float *cp = C + ((size_t)b * NC + j2) * 3;
    const float g = gdA[(size_t)b * NA + iq] * 2;
    const float dxd = g * (ax - cp[0]), dyd = g * (ay - cp[1]), dzd = g * (az - cp[2]);
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (which == 0) { gx += dxd; gy += dyd; gz += dzd; }
    // scatter term, gathered: scan the partner cloud's selections in index order
    const int32_t *icb = ic + (size_t)b * NC;
    for (int c0 = 0; c0 < NC; c0 += BTILE) {
        const int tn = min(BTILE, NC - c0);
        __syncthreads();
        for (int t = lane; t < tn; t += 64) sel[t] = icb[c0 + t];
        __syncthreads();
        // 32 selections per trip into a bit mask (a per-candidate branch costs an exec-mask round trip
        // even when no lane takes it); the rare matches are popped in index order afterwards
        for (int g0 = 0; g0 < tn; g0 += 32) {
            unsigned mask = 0;
            const int gn = min(32, tn - g0);
            if (gn == 32) {
#pragma unroll
                for (int u = 0; u < 32; u++) mask |= sel[g0 + u] == iq ? (1u << u) : 0u;
            } else {
                for (int u = 0; u < gn; u++) mask |= sel[g0 + u] == iq ? (1u << u) : 0u;
            }
#pragma unroll 1
            while (__any(mask != 0)) {
                if (mask != 0) {
                    const int j = c0 + g0 + __builtin_ctz(mask);
                    mask &= mask - 1;
                    const float *pj = C + ((size_t)b * NC + j) * 3;
                    const float gj = gdC[(size_t)b * NC + j] * 2;
                    gx -= gj * (pj[0] - ax);
                    gy -= gj * (pj[1] - ay);
                    gz -= gj * (pj[2] - az);
                }
            }
        }
    }
    if (which == 1) { gx += dxd; gy += dyd; gz += dzd; }
    if (valid) {
        float *o = gout + ((size_t)b * NA + i) * 3;
        o[0] = gx; o[1] = gy; o[2] = gz;
    }
}

// ---------------------------------------------------------------------------------------------
// The same gradients from a SORTED selection list.  The scan above does N*M index comparisons per cloud and direction (as
// many as the forward search does distance evaluations, at a fraction of its rate: 155 us against 18 us at B = 32,
// N = M = 1024) to find, for every point i, the partner points j with ic[j] == i in ascending j.  Here one 1024-thread
// workgroup per (cloud, direction) sorts the keys (ic[j] << shift | j) in LDS (bitonic; <= 32768 points = 128 KB), which
// puts every point's partners next to each other in ascending j; a point then finds its run with one binary search and
// adds the terms in that order -- the accumulation order of chamfer_distance.cpp:138-176, hence the scan kernel's bits.
// ---------------------------------------------------------------------------------------------
#define CB_SORT_MAX 32768
__global__ __launch_bounds__(1024) void chamfer_bwd_sorted_kernel(const float *__restrict__ xyz1,
                                                                  const float *__restrict__ xyz2, int N, int M,
                                                                  const float *__restrict__ gd1,
                                                                  const float *__restrict__ gd2,
                                                                  const int32_t *__restrict__ idx1,
                                                                  const int32_t *__restrict__ idx2,
                                                                  float *__restrict__ g1, float *__restrict__ g2,
                                                                  int P1, int P2, int shift1, int shift2)
{
    extern __shared__ uint32_t keys[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int which = blockIdx.y;
    const float *A = which == 0 ? xyz1 : xyz2;
    const float *C = which == 0 ? xyz2 : xyz1;
    const int NA = which == 0 ? N : M;
    const int NC = which == 0 ? M : N;
    const int P = which == 0 ? P2 : P1;                     // power of two >= NC
    const int shift = which == 0 ? shift2 : shift1;         // 2^shift >= NC
    const float *gdA = (which == 0 ? gd1 : gd2) + (size_t)b * NA;
    const float *gdC = (which == 0 ? gd2 : gd1) + (size_t)b * NC;
    const int32_t *ia = (which == 0 ? idx1 : idx2) + (size_t)b * NA;
    const int32_t *ic = (which == 0 ? idx2 : idx1) + (size_t)b * NC;
    float *gout = (which == 0 ? g1 : g2) + (size_t)b * NA * 3;
    const float *Ab = A + (size_t)b * NA * 3, *Cb = C + (size_t)b * NC * 3;

    for (int t = tid; t < P; t += 1024) keys[t] = t < NC ? ((uint32_t)ic[t] << shift) | (uint32_t)t : 0xffffffffu;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += 1024) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const uint32_t x = keys[lo], y = keys[hi];
                const bool up = (lo & k) == 0;              // ascending block
                if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
            }
            __syncthreads();
        }
    }
    const uint32_t jmask = (1u << shift) - 1u;
    for (int i = tid; i < NA; i += 1024) {
        const float ax = Ab[i * 3], ay = Ab[i * 3 + 1], az = Ab[i * 3 + 2];
        const int j2 = ia[i];
        const float g = gdA[i] * 2;
        const float dxd = g * (ax - Cb[j2 * 3]), dyd = g * (ay - Cb[j2 * 3 + 1]), dzd = g * (az - Cb[j2 * 3 + 2]);
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (which == 0) { gx += dxd; gy += dyd; gz += dzd; }
        // first position whose key is >= (i << shift)
        const uint32_t want = (uint32_t)i << shift;
        int lo = 0, hi = NC;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] < want) lo = mid + 1; else hi = mid;
        }
        for (int pos = lo; pos < NC; pos++) {
            const uint32_t key = keys[pos];
            if ((key >> shift) != (uint32_t)i) break;
            const int j = (int)(key & jmask);
            const float gj = gdC[j] * 2;
            gx -= gj * (Cb[j * 3] - ax);
            gy -= gj * (Cb[j * 3 + 1] - ay);
            gz -= gj * (Cb[j * 3 + 2] - az);
        }
        if (which == 1) { gx += dxd; gy += dyd; gz += dzd; }
        gout[i * 3] = gx; gout[i * 3 + 1] = gy; gout[i * 3 + 2] = gz;
    }
}

// variant 0 = the scan kernel, 2 = the sorted-list kernel (N, M <= 32768), 1 = auto (what l3d_chamfer_backward does)
extern "C" int l3d_chamfer_backward_variant(const float *xyz1, const float *xyz2, int B, int N, int M,
                                            const float *graddist1, const float *graddist2,
                                            const int32_t *idx1, const int32_t *idx2, float *gradxyz1,
                                            float *gradxyz2, int variant, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz1 && xyz2 && graddist1 && graddist2 && idx1 && idx2 && gradxyz1 && gradxyz2 &&
                B > 0 && N > 0 && M > 0 && variant >= 0 && variant <= 2);
    const int mx = N > M ? N : M;
    const bool sortable = mx <= CB_SORT_MAX;
    if (variant == 2 && !sortable) return L3D_ERR_UNSUPPORTED;
    if (variant != 0 && sortable) {
        int P1 = 2, P2 = 2, s1 = 1, s2 = 1;                  // P1 >= N (keys of idx1: direction 1 sorts the N selections of cloud 1)
        while (P1 < N) { P1 <<= 1; s1++; }
        while (P2 < M) { P2 <<= 1; s2++; }
        const size_t lds = sizeof(uint32_t) * (size_t)(P1 > P2 ? P1 : P2);
        hipLaunchKernelGGL(chamfer_bwd_sorted_kernel, dim3(B, 2), dim3(1024), lds, (hipStream_t)stream, xyz1, xyz2, N, M,
                           graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, P1, P2, s1, s2);
        const int rc = l3d_check_launch();
        // auto mode: a launch that the device refuses (up to 128 KB of dynamic LDS at 32 768 points) falls back to the scan
        // kernel -- same bits, slower -- instead of failing clouds the scan kernel has always taken
        if (rc == L3D_OK || variant == 2) return rc;
    }
    dim3 grid(l3d_divup(mx, 64), B, 2);
    hipLaunchKernelGGL(chamfer_bwd_kernel, grid, dim3(64), 0, (hipStream_t)stream, xyz1, xyz2, N, M,
                       graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2);
    return l3d_check_launch();
}

extern "C" int l3d_chamfer_backward(const float *xyz1, const float *xyz2, int B, int N, int M,
                                    const float *graddist1, const float *graddist2,
                                    const int32_t *idx1, const int32_t *idx2, float *gradxyz1,
                                    float *gradxyz2, l3d_stream_t stream)
{
    return l3d_chamfer_backward_variant(xyz1, xyz2, B, N, M, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, 1, stream);
}

// ---------------------------------------------------------------------------------------------
// Loss tail, kept on the device so a step never synchronises with the host:
//   partial[0..3] = (sum sqrt(dist1), sum sqrt(dist2), #dist1, #dist2) of this rank's shard  (fp64)
//   loss = (sum_r p[r][0] / sum_r p[r][2] + sum_r p[r][1] / sum_r p[r][3]) / 2   over the gathered
//          per-rank partials  == losses/chamfer_distance.py:38-40 on the whole batch.
// One 1024-thread workgroup per direction, no atomics, no pre-zeroing (deterministic).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sqrt_sum_kernel(const float *__restrict__ d1, size_t n1,
                                                        const float *__restrict__ d2, size_t n2,
                                                        double *__restrict__ partial)
{
    __shared__ double part[16];
    const int which = blockIdx.x;
    const float *d = which == 0 ? d1 : d2;
    const size_t n = which == 0 ? n1 : n2;
    double acc = 0.0;
    const size_t n4 = n >> 2;                          // 16-byte loads, 4 independent sqrt per trip
    for (size_t i = threadIdx.x; i < n4; i += blockDim.x) {
        const float4 v = ((const float4 *)d)[i];
        acc += ((double)sqrtf(v.x) + (double)sqrtf(v.y)) + ((double)sqrtf(v.z) + (double)sqrtf(v.w));
    }
    for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) acc += (double)sqrtf(d[i]);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; w++) t += part[w];
        partial[which] = t;
        partial[2 + which] = (double)n;
    }
}

extern "C" int l3d_chamfer_partials(const float *dist1, const float *dist2, int B, int N, int M,
                                    double *partial, l3d_stream_t stream)
{
    L3D_REQUIRE(dist1 && dist2 && partial && B > 0 && N > 0 && M > 0);
    hipLaunchKernelGGL(sqrt_sum_kernel, dim3(2), dim3(1024), 0, (hipStream_t)stream, dist1,
                       (size_t)B * N, dist2, (size_t)B * M, partial);
    return l3d_check_launch();
}

// Single-rank tail in ONE launch over up to 64 workgroups (a one-workgroup kernel was a chain of 16 dependent-latency loop trips
// per thread, 12 us for 2 x 32 K values): both sqrt-sums and the combine; every thread loads its one or two float4 at once, each
// workgroup writes its two fp64 partial sums to ws, and the workgroup that draws the last ticket adds them with a
// fixed shuffle tree (deterministic) and re-arms the ticket for the next call on the stream.
//   ws: CHAMFER_LL_WS_BYTES bytes, the first 8 of them zero before the first call.
__global__ __launch_bounds__(256) void chamfer_loss_local_mb_kernel(const float *__restrict__ d1, size_t n1,
                                                                    const float *__restrict__ d2, size_t n2,
                                                                    unsigned *__restrict__ ws, double *__restrict__ partial,
                                                                    float *__restrict__ loss)
{
    __shared__ double part[2][4];
    __shared__ bool last;
    double acc1 = 0.0, acc2 = 0.0;
    const size_t a4 = n1 >> 2, b4 = n2 >> 2, m4 = a4 > b4 ? a4 : b4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m4; i += stride) {
        if (i < a4) {
            const float4 v = ((const float4 *)d1)[i];
            acc1 += ((double)sqrtf(v.x) + (double)sqrtf(v.y)) + ((double)sqrtf(v.z) + (double)sqrtf(v.w));
        }
        if (i < b4) {
            const float4 v = ((const float4 *)d2)[i];
            acc2 += ((double)sqrtf(v.x) + (double)sqrtf(v.y)) + ((double)sqrtf(v.z) + (double)sqrtf(v.w));
        }
    }
    if (blockIdx.x == 0) {
        for (size_t i = (a4 << 2) + threadIdx.x; i < n1; i += 256) acc1 += (double)sqrtf(d1[i]);
        for (size_t i = (b4 << 2) + threadIdx.x; i < n2; i += 256) acc2 += (double)sqrtf(d2[i]);
    }
    for (int off = 32; off > 0; off >>= 1) { acc1 += __shfl_down(acc1, off, 64); acc2 += __shfl_down(acc2, off, 64); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = acc1; part[1][threadIdx.x >> 6] = acc2; }
    __syncthreads();
    double *slots = (double *)(ws + 4);
    if (threadIdx.x == 0) {
        slots[2 * blockIdx.x] = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
        slots[2 * blockIdx.x + 1] = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
        __threadfence();
        last = atomicAdd(ws, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {                 // one wave: a slot pair per lane, then a fixed shuffle tree
        __threadfence();
        double t1 = 0.0, t2 = 0.0;
        if (threadIdx.x < gridDim.x) {
            t1 = __builtin_nontemporal_load(&slots[2 * threadIdx.x]);
            t2 = __builtin_nontemporal_load(&slots[2 * threadIdx.x + 1]);
        }
        for (int off = 32; off > 0; off >>= 1) { t1 += __shfl_down(t1, off, 64); t2 += __shfl_down(t2, off, 64); }
        if (threadIdx.x == 0) {
            partial[0] = t1; partial[1] = t2; partial[2] = (double)n1; partial[3] = (double)n2;
            loss[0] = (float)((t1 / (double)n1 + t2 / (double)n2) / 2.0);
            ws[0] = 0;                              // re-armed for the next launch on this stream
        }
    }
}

extern "C" size_t l3d_chamfer_loss_local_ws_bytes(void) { return CHAMFER_LL_WS_BYTES; }

extern "C" int l3d_chamfer_loss_local_mb(const float *dist1, const float *dist2, int B, int N, int M, void *ws, double *partial,
                                         float *loss, l3d_stream_t stream)
{
    L3D_REQUIRE(dist1 && dist2 && ws && partial && loss && B > 0 && N > 0 && M > 0);
    const size_t n1 = (size_t)B * N, n2 = (size_t)B * M, m4 = (n1 > n2 ? n1 : n2) >> 2;
    long blocks = l3d_divup((long)m4, 256);
    blocks = blocks < 1 ? 1 : (blocks > CHAMFER_LL_BLOCKS ? CHAMFER_LL_BLOCKS : blocks);
    // the 4-byte ticket in ws is re-armed by the kernel's last block; a caller whose launch failed or was aborted zeroes ws before
    // the next call (learning3d_amd/losses/chamfer_distance.py does, on any failed status) -- a memset per launch is a 4.8 us
    // fill kernel in front of a 4.6 us kernel (measured: profiles/round3, first collection)
    hipLaunchKernelGGL(chamfer_loss_local_mb_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dist1, n1, dist2, n2,
                       (unsigned *)ws, partial, loss);
    return l3d_check_launch();
}


__global__ void chamfer_combine_kernel(const double *__restrict__ partials, int world,
                                       float *__restrict__ loss)
{
    if (threadIdx.x != 0) return;
    double s1 = 0, s2 = 0, n1 = 0, n2 = 0;
    for (int r = 0; r < world; r++) {
        s1 += partials[r * 4 + 0]; s2 += partials[r * 4 + 1];
        n1 += partials[r * 4 + 2]; n2 += partials[r * 4 + 3];
    }
    loss[0] = (float)((s1 / n1 + s2 / n2) / 2.0);
}

extern "C" int l3d_chamfer_combine(const double *partials, int world, float *loss, l3d_stream_t stream)
{
    L3D_REQUIRE(partials && loss && world > 0);
    hipLaunchKernelGGL(chamfer_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, world, loss);
    return l3d_check_launch();
}
