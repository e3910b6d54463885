// grouping.hip -- ball query, grouping / gather, farthest point sampling, 3-NN interpolation.
//
// Replaces the pybind module `pointnet2_cuda` (utils/lib/src/pointnet2_api.cpp:10-25):
//   K7  ball_query_kernel_fast            ball_query_gpu.cu:9-45
//   K8  group_points_kernel_fast          group_points_gpu.cu:47-66     K9  grad  :8-25
//   K10 gather_points_kernel_fast         sampling_gpu.cu:8-24          K11 grad  :46-63
//   K12 furthest_point_sampling_kernel    sampling_gpu.cu:93-209
//   K15 three_interpolate_kernel_fast     interpolate_gpu.cu:149-169    K16 grad  :192-214
// and the fused torch-level twins of utils/model_common_utils.py:
//   T4  query_ball_point :102-130 (+ ppfnet_util.py:96-131 itself_indices, pointconv_util.py:85-105)
//       -- the reference sorts an int64 [B,S,N] tensor (17 GB at B=256,N=8192,S=1024); here it is
//          one streaming scan with early exit
//   a3  square_distance :19-38,  a5 index_points :40-56,  T7 farthest_point_sample :58-82
//
// (K13/K14 knn / three_nn live in knn.hip.)
#include "common.h"
#include "split_bf16.h"          // f32x4

#define BQ_TILE 2048

// ---------------------------------------------------------------------------------------------
// Ball query.  One wave64 per workgroup, one centroid per lane, cloud streamed through LDS.
// MODE 0 = native K7 : direct-difference d2, strict '<', int32 out, empty ball -> 0
// MODE 1 = torch  T4 : expanded d2 (reference rounding order), '<=' (mask is d2 > r2), int64 out,
//                      empty ball -> N, optional hit count, optional itself_indices
// The scan stops as soon as every lane of the wave has nsample hits (unless counting).
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64) void ball_query_kernel(int n, int m, float r2, int nsample,
                                                        const float *__restrict__ new_xyz,
                                                        const float *__restrict__ xyz,
                                                        const int64_t *__restrict__ itself,
                                                        void *__restrict__ idx_out,
                                                        int64_t *__restrict__ cnt_out)
{
    __shared__ float4 cand[BQ_TILE];
    const int lane = threadIdx.x;
    const int b = blockIdx.y;
    const int s = blockIdx.x * 64 + lane;
    const bool valid = s < m;
    const int sc = valid ? s : m - 1;
    const float *qp = new_xyz + ((size_t)b * m + sc) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float qss = (qx * qx + qy * qy) + qz * qz;
    const int self = (MODE == 1 && itself) ? (int)itself[(size_t)b * m + sc] : -1;
    int32_t *o32 = (int32_t *)idx_out + ((size_t)b * m + sc) * nsample;
    int64_t *o64 = (int64_t *)idx_out + ((size_t)b * m + sc) * nsample;
    const float *cbase = xyz + (size_t)b * n * 3;
    const bool counting = (MODE == 1) && cnt_out != nullptr;

    int cnt = 0, first = -1;
    long total = 0;
    for (int c0 = 0; c0 < n; c0 += BQ_TILE) {
        const int tn = min(BQ_TILE, n - c0);
        const int tp = (tn + 31) & ~31;              // padded with far-away sentinels (never inside a ball)
        __syncthreads();
        // eight points per lane in flight before the first LDS write (rolled, this loop was one round trip to memory per 64
        // points); unconditional loads from clamped indices -- a conditional load is a branch with a wait at its join
        for (int tb = 0; tb < tp; tb += 512) {
            float sx[8], sy[8], sz[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float *cp = cbase + (size_t)(c0 + min(tb + 64 * u + lane, tn - 1)) * 3;
                sx[u] = cp[0]; sy[u] = cp[1]; sz[u] = cp[2];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int t = tb + 64 * u + lane;
                if (t < tp)
                    cand[t] = t < tn ? make_float4(sx[u], sy[u], sz[u], (sx[u] * sx[u] + sy[u] * sy[u]) + sz[u] * sz[u])
                                     : make_float4(3.0e18f, 3.0e18f, 3.0e18f, 2.7e37f);
            }
        }
        __syncthreads();
        if (!counting && __all(cnt >= nsample || !valid)) break;
        // 32 candidates per trip: the in-ball test only sets a bit (no exec-mask branch, no store per
        // candidate); hits are popped afterwards in index order.  The wave leaves the scan as soon as
        // every lane holds nsample hits.
        for (int g0 = 0; g0 < tn; g0 += 32) {
            unsigned mask = 0;
#pragma unroll
            for (int ch = 0; ch < 32; ch += 8) {
                float4 c[8];
#pragma unroll
                for (int u = 0; u < 8; u++) c[u] = cand[g0 + ch + u];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    bool hit;
                    if (MODE == 0) {
                        const float dx = qx - c[u].x, dy = qy - c[u].y, dz = qz - c[u].z;
                        const float d2 = (dx * dx + dy * dy) + dz * dz;
                        hit = d2 < r2;
                    } else {
                        // square_distance(new_xyz, xyz): (-2*dot + |q|^2) + |p|^2
                        const float dot = fmaf(qz, c[u].z, fmaf(qy, c[u].y, qx * c[u].x));
                        const float d2 = (-2.0f * dot + qss) + c[u].w;
                        hit = !(d2 > r2) && (c0 + g0 + ch + u != self);
                    }
                    mask |= hit ? (1u << (ch + u)) : 0u;
                }
            }
            if (counting) total += __builtin_popcount(mask);
            if (cnt >= nsample || !valid) mask = 0;
#pragma unroll 1
            while (__any(mask != 0)) {
                if (mask != 0) {
                    const int hit_idx = c0 + g0 + __builtin_ctz(mask);
                    mask &= mask - 1;
                    if (MODE == 0) o32[cnt] = hit_idx; else o64[cnt] = hit_idx;
                    if (cnt == 0) first = hit_idx;
                    cnt++;
                    if (cnt >= nsample) mask = 0;
                }
            }
            if (!counting && __all(cnt >= nsample || !valid)) break;
        }
    }
    if (!valid) return;
    // pad: native -> first hit (or 0 when the ball is empty, the pre-zeroed idx of
    // pointnet2_utils.py:246); torch -> first hit / itself_indices / N when empty
    long fill;
    if (MODE == 0) fill = first < 0 ? 0 : first;
    else fill = self >= 0 ? self : (first < 0 ? n : first);
    for (int l = cnt; l < nsample; l++) {
        if (MODE == 0) o32[l] = (int32_t)fill; else o64[l] = fill;
    }
    if (counting) cnt_out[(size_t)b * m + s] = total;
}

// ---------------------------------------------------------------------------------------------
// Sliced ball query (n >= 1024, nsample <= 64): the single-wave kernel above leaves half the SIMDs idle at
// FlowNet3D's sa1 (512 waves on 1024 SIMDs) and each wave walks all n candidates.  Here 4 waves serve the SAME
// 64 centroids, wave w scanning the w-th quarter of the cloud through its own LDS tile (no workgroup barrier
// in the scan, so a wave can leave early) and collecting its first nsample hits in index order.  A wave may
// stop once its own hits plus the hits of the EARLIER quarters reach nsample for every lane; the earlier waves'
// counts are read from LDS without synchronisation -- they only grow, so a stale value just delays the exit.
// Wave 0 then concatenates the four lists in slice order (= index order) and pads like the kernel above.
// ---------------------------------------------------------------------------------------------
#define BQ_W 4
#define BQ_T4 512
template <int MODE>
__global__ __launch_bounds__(64 * BQ_W) void ball_query_sliced_kernel(int n, int m, float r2, int nsample,
                                                                      const float *__restrict__ new_xyz,
                                                                      const float *__restrict__ xyz,
                                                                      const int64_t *__restrict__ itself,
                                                                      void *__restrict__ idx_out,
                                                                      int64_t *__restrict__ cnt_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char bq_lds[];
    float4 *cand = (float4 *)bq_lds + (threadIdx.x >> 6) * BQ_T4;                       // this wave's tile
    int *hits = (int *)(bq_lds + BQ_W * BQ_T4 * 16);                                    // [BQ_W][nsample][64]
    volatile int *cnts = (volatile int *)(hits + BQ_W * nsample * 64);                  // [BQ_W][64]
    long *totals = (long *)(hits + BQ_W * nsample * 64 + BQ_W * 64);                    // [BQ_W][64] (counting)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int s = blockIdx.x * 64 + lane;
    const bool valid = s < m;
    const int sc = valid ? s : m - 1;
    const float *qp = new_xyz + ((size_t)b * m + sc) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float qss = (qx * qx + qy * qy) + qz * qz;
    const int self = (MODE == 1 && itself) ? (int)itself[(size_t)b * m + sc] : -1;
    const float *cbase = xyz + (size_t)b * n * 3;
    const bool counting = (MODE == 1) && cnt_out != nullptr;
    int *myhits = hits + wave * nsample * 64;

    cnts[wave * 64 + lane] = 0;
    __syncthreads();
    const int per = ((n + BQ_W - 1) / BQ_W + 31) & ~31;         // slice length, multiple of 32
    const int lo = wave * per, hi = min(n, lo + per);
    int cnt = 0;
    long total = 0;
    for (int c0 = lo; c0 < hi; c0 += BQ_T4) {
        const int tn = min(BQ_T4, hi - c0);
        const int tp = (tn + 31) & ~31;
        {   // the tile's 8 points per lane all in flight before the first LDS write (see ball_query_kernel)
            float sx[BQ_T4 / 64], sy[BQ_T4 / 64], sz[BQ_T4 / 64];
#pragma unroll
            for (int u = 0; u < BQ_T4 / 64; u++) {
                const float *cp = cbase + (size_t)(c0 + min(64 * u + lane, tn - 1)) * 3;
                sx[u] = cp[0]; sy[u] = cp[1]; sz[u] = cp[2];
            }
#pragma unroll
            for (int u = 0; u < BQ_T4 / 64; u++) {
                const int t = 64 * u + lane;
                if (t < tp)
                    cand[t] = t < tn ? make_float4(sx[u], sy[u], sz[u], (sx[u] * sx[u] + sy[u] * sy[u]) + sz[u] * sz[u])
                                     : make_float4(3.0e18f, 3.0e18f, 3.0e18f, 2.7e37f);   // far-away sentinel, never inside a ball
            }
        }
        bool stop = false;
        for (int g0 = 0; g0 < tn; g0 += 32) {
            unsigned mask = 0;
#pragma unroll
            for (int ch = 0; ch < 32; ch += 8) {
                float4 c[8];
#pragma unroll
                for (int u = 0; u < 8; u++) c[u] = cand[g0 + ch + u];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    bool hit;
                    if (MODE == 0) {
                        const float dx = qx - c[u].x, dy = qy - c[u].y, dz = qz - c[u].z;
                        const float d2 = (dx * dx + dy * dy) + dz * dz;
                        hit = d2 < r2;
                    } else {
                        const float dot = fmaf(qz, c[u].z, fmaf(qy, c[u].y, qx * c[u].x));
                        const float d2 = (-2.0f * dot + qss) + c[u].w;
                        hit = !(d2 > r2) && (c0 + g0 + ch + u != self);
                    }
                    mask |= hit ? (1u << (ch + u)) : 0u;
                }
            }
            if (counting) total += __builtin_popcount(mask);
            if (cnt >= nsample || !valid) mask = 0;
            if (__builtin_amdgcn_ballot_w64(mask != 0) != 0) {
#pragma unroll 1
                do {
                    if (mask != 0) {
                        myhits[cnt * 64 + lane] = c0 + g0 + __builtin_ctz(mask);
                        mask &= mask - 1;
                        cnt++;
                        if (cnt >= nsample) mask = 0;
                    }
                } while (__builtin_amdgcn_ballot_w64(mask != 0) != 0);
            }
            cnts[wave * 64 + lane] = cnt;
            if (!counting) {
                int before = 0;
                for (int w = 0; w < wave; w++) before += cnts[w * 64 + lane];
                if (__all(cnt + before >= nsample || !valid)) { stop = true; break; }
            }
        }
        if (stop) break;
    }
    cnts[wave * 64 + lane] = cnt;
    if (counting) totals[wave * 64 + lane] = total;
    __syncthreads();
    if (wave != 0 || !valid) return;
    // concatenate the slices' lists in slice order
    int cum[BQ_W], tot = 0;
#pragma unroll
    for (int w = 0; w < BQ_W; w++) { tot += cnts[w * 64 + lane]; cum[w] = tot; }
    const int got = min(tot, nsample);
    int32_t *o32 = (int32_t *)idx_out + ((size_t)b * m + s) * nsample;
    int64_t *o64 = (int64_t *)idx_out + ((size_t)b * m + s) * nsample;
    int first = -1;
    for (int l = 0; l < got; l++) {
        int w = 0, base = 0;
#pragma unroll
        for (int u = 0; u < BQ_W - 1; u++) {
            const bool past = l >= cum[u];
            w += past ? 1 : 0;
            base = past ? cum[u] : base;
        }
        const int h = hits[(w * nsample + (l - base)) * 64 + lane];
        if (l == 0) first = h;
        if (MODE == 0) o32[l] = h; else o64[l] = h;
    }
    long fill;
    if (MODE == 0) fill = first < 0 ? 0 : first;
    else fill = self >= 0 ? self : (first < 0 ? n : first);
    for (int l = got; l < nsample; l++) {
        if (MODE == 0) o32[l] = (int32_t)fill; else o64[l] = fill;
    }
    if (counting) {
        long t = 0;
#pragma unroll
        for (int w = 0; w < BQ_W; w++) t += totals[w * 64 + lane];
        cnt_out[(size_t)b * m + s] = t;
    }
}

static size_t bq_sliced_lds(int nsample)
{
    return (size_t)BQ_W * BQ_T4 * 16 + (size_t)BQ_W * nsample * 64 * 4 + (size_t)BQ_W * 64 * 4 + (size_t)BQ_W * 64 * 8 + 16;
}

// ---------------------------------------------------------------------------------------------
// Ball query through a cell list (native K7 semantics, n >= 2048, nsample <= 64, caller-provided workspace).  The scanning kernels
// above evaluate all n candidates per centroid unless every lane of a wave fills up early -- and furthest-point-sampled centroids
// are scattered by construction, so a wave almost always holds a centroid of a sparse region: 268 M pair evaluations per batch at
// FlowNet3D's sa1 (32 clouds x 1024 centroids x 8192 points), 116 us.  Here:
//   bq_cells_build_kernel   one workgroup per cloud: bounding box, a grid of cells of edge >= 1.001 r (at most 16 per axis),
//                           counting sort of the points by cell -> (x, y, z, index) records in cell order + cell starts
//   bq_cells_query_kernel   one WAVE per centroid: the 3 x 3 x 3 cells around it are nine contiguous runs of records (the three
//                           x-neighbours are adjacent in cell order); a lane per record, the same d2 = (dx dx + dy dy) + dz dz < r^2
//                           test as ball_query_kernel<0>, and of the hits the nsample SMALLEST INDICES are kept -- which is what
//                           "the first nsample hits of a scan in index order" are -- as a sorted list, one entry per lane; a hit
//                           enters only if it beats the list's current last entry (a few dozen insertions per centroid)
// Every in-ball point lies in those 27 cells: |dx| < r <= edge / 1.001 moves the (monotone, rounded) cell coordinate by at most one.
// Same indices, same order, same padding as the scanning kernels (tests compare them on every shape); ~20x fewer pair evaluations.
// ---------------------------------------------------------------------------------------------
#define BQC_G 16                                   // cells per axis, at most
#define BQC_NC (BQC_G * BQC_G * BQC_G)
struct BqGrid { float minx, miny, minz, inv; int gx, gy, gz, pad; };

static size_t bq_cells_ws_per_cloud(int n) { return (size_t)16 * n + 4 * (BQC_NC + 8) + sizeof(BqGrid); }

__global__ __launch_bounds__(1024) void bq_cells_build_kernel(int n, float radius, const float *__restrict__ xyz, unsigned char *__restrict__ ws,
                                                              size_t ws_per_cloud)
{
    __shared__ int counts[BQC_NC];
    __shared__ float red[6][16];
    __shared__ int wsum[17];
    __shared__ BqGrid grid;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
    const float *cloud = xyz + (size_t)b * n * 3;
    unsigned char *w = ws + (size_t)b * ws_per_cloud;
    float4 *rec = (float4 *)w;
    int *start = (int *)(w + (size_t)16 * n);
    BqGrid *gout = (BqGrid *)(w + (size_t)16 * n + 4 * (BQC_NC + 8));
    // bounding box
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = t; i < n; i += 1024)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = cloud[(size_t)i * 3 + c];
            lo[c] = fminf(lo[c], v);
            hi[c] = fmaxf(hi[c], v);
        }
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], d, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], d, 64));
        }
        if (lane == 0) { red[c][wave] = lo[c]; red[3 + c][wave] = hi[c]; }
    }
    for (int i = t; i < BQC_NC; i += 1024) counts[i] = 0;
    __syncthreads();
    if (t == 0) {
        float mn[3], mx[3], ext = 0.f;
        for (int c = 0; c < 3; c++) {
            mn[c] = red[c][0]; mx[c] = red[3 + c][0];
            for (int k = 1; k < 16; k++) { mn[c] = fminf(mn[c], red[c][k]); mx[c] = fmaxf(mx[c], red[3 + c][k]); }
            ext = fmaxf(ext, mx[c] - mn[c]);
        }
        // cell edge: at least 1.001 r (so that rounding of the cell coordinate cannot make an in-ball point skip a cell) and large
        // enough for 16 cells to span the longest axis; a degenerate or non-finite box collapses to one cell
        float edge = fmaxf(radius * 1.001f, ext * (1.0f / BQC_G) * 1.001f);
        if (!(edge > 0.f) || !(edge < 3.0e38f)) edge = 1.f;
        grid.minx = mn[0]; grid.miny = mn[1]; grid.minz = mn[2];
        grid.inv = 1.0f / edge;
        int g[3];
        for (int c = 0; c < 3; c++) {
            const float cells = (mx[c] - mn[c]) * grid.inv;
            g[c] = (cells >= 0.f && cells < (float)BQC_G) ? (int)cells + 1 : (cells >= (float)BQC_G ? BQC_G : 1);
            if (g[c] > BQC_G) g[c] = BQC_G;
        }
        grid.gx = g[0]; grid.gy = g[1]; grid.gz = g[2]; grid.pad = 0;
        *gout = grid;
    }
    __syncthreads();
    const BqGrid G = grid;
    auto cell_of = [&](float x, float y, float z) {
        int cx = (int)floorf((x - G.minx) * G.inv), cy = (int)floorf((y - G.miny) * G.inv), cz = (int)floorf((z - G.minz) * G.inv);
        cx = min(max(cx, 0), G.gx - 1); cy = min(max(cy, 0), G.gy - 1); cz = min(max(cz, 0), G.gz - 1);
        return (cz * G.gy + cy) * G.gx + cx;
    };
    for (int i = t; i < n; i += 1024) atomicAdd(&counts[cell_of(cloud[(size_t)i * 3], cloud[(size_t)i * 3 + 1], cloud[(size_t)i * 3 + 2])], 1);
    __syncthreads();
    // exclusive scan of the 4096 counts: four per thread, wave scan, wave totals
    {
        int v[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = counts[t * 4 + k]; s += v[k]; }
        int inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave + 1] = inc;
        __syncthreads();
        if (t == 0) {
            wsum[0] = 0;
            for (int k = 1; k <= 16; k++) wsum[k] += wsum[k - 1];
        }
        __syncthreads();
        int ex = wsum[wave] + inc - s;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            start[t * 4 + k] = ex;
            counts[t * 4 + k] = ex;                  // becomes the scatter cursor
            ex += v[k];
        }
        if (t == 1023) start[BQC_NC] = ex;
    }
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        const float x = cloud[(size_t)i * 3], y = cloud[(size_t)i * 3 + 1], z = cloud[(size_t)i * 3 + 2];
        const int pos = atomicAdd(&counts[cell_of(x, y, z)], 1);       // order within a cell is irrelevant: the query selects by index
        rec[pos] = make_float4(x, y, z, __int_as_float(i));
    }
}

__global__ __launch_bounds__(256, 8) void bq_cells_query_kernel(int n, int m, float r2, int nsample, const float *__restrict__ new_xyz,
                                                             const unsigned char *__restrict__ ws, size_t ws_per_cloud,
                                                             int32_t *__restrict__ idx_out)
{
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    // the wave's centroid, as a value the compiler KNOWS is wave-uniform: everything derived from it (the cell coordinate, the runs'
    // bounds, every loop and branch below) then lives in scalar registers -- left as threadIdx.x >> 6 the cell-start reads became
    // per-lane loads behind exec-masked branches with a full wait behind each one (65.7 us; LABLOG R4.11)
    const int s = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (s >= m) return;                                                 // whole waves leave: no barrier below
    const unsigned char *w = ws + (size_t)b * ws_per_cloud;
    const float4 *rec = (const float4 *)w;
    const int *start = (const int *)(w + (size_t)16 * n);
    const BqGrid G = *(const BqGrid *)(w + (size_t)16 * n + 4 * (BQC_NC + 8));
    const float *qp = new_xyz + ((size_t)b * m + s) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    // the centroid's cell coordinate, NOT clamped to the grid (a query point outside the cloud's box sees only the cells that exist)
    const float fx = floorf((qx - G.minx) * G.inv), fy = floorf((qy - G.miny) * G.inv), fz = floorf((qz - G.minz) * G.inv);
    const int cx = (fx >= -2.f && fx <= (float)(BQC_G + 1)) ? (int)fx : (fx < 0.f ? -2 : BQC_G + 1);
    const int cy = (fy >= -2.f && fy <= (float)(BQC_G + 1)) ? (int)fy : (fy < 0.f ? -2 : BQC_G + 1);
    const int cz = (fz >= -2.f && fz <= (float)(BQC_G + 1)) ? (int)fz : (fz < 0.f ? -2 : BQC_G + 1);
    int L = 0x7fffffff;                                                 // lane l: the l-th smallest hit index so far
    int thr = 0x7fffffff;                                               // the list's entry nsample - 1 (wave-uniform)
    // a lane's record -> its hit index or "none"; hits that beat the list's last entry are inserted one at a time (sorted insert:
    // everything behind the position moves up a lane -- a DPP wave shift, no LDS trip)
    auto consume = [&](bool live, const float4 &c) {
        int cand = 0x7fffffff;
        if (live) {
            const float dx = qx - c.x, dyy = qy - c.y, dzz = qz - c.z;
            const float d2 = (dx * dx + dyy * dyy) + dzz * dzz;         // ball_query_gpu.cu:32-34's evaluation order
            if (d2 < r2) cand = __float_as_int(c.w);
        }
        unsigned long long mask = __builtin_amdgcn_ballot_w64(cand < thr);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int v = __builtin_amdgcn_readlane(cand, j);
            if (v >= thr) continue;                                     // the list moved on since the ballot
            const int pos = __builtin_popcountll(__builtin_amdgcn_ballot_w64(L < v));
            const int up = __builtin_amdgcn_update_dpp(L, L, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            L = lane > pos ? up : (lane == pos ? v : L);
            thr = __builtin_amdgcn_readlane(L, nsample - 1);
        }
    };
    // the nine runs of records (the three x-neighbours of a (y, z) row are adjacent in cell order); their first 64 records are
    // requested together, so that a centroid pays two memory round trips instead of nine pairs of them
    int st[9], en[9];
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G.gx - 1);
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
        const bool ok = z >= 0 && z < G.gz && y >= 0 && y < G.gy && x0 <= x1;
        const int row = ok ? (z * G.gy + y) * G.gx : 0;
        st[r] = ok ? start[row + x0] : 0;
        en[r] = ok ? start[row + x1 + 1] : 0;
    }
    float4 head[9];
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const int i = st[r] + lane;
        head[r] = rec[min(i, n - 1)];                                   // unconditional (clamped): nine loads in flight, one wait
    }
#pragma unroll
    for (int r = 0; r < 9; r++) {
        if (st[r] >= en[r]) continue;
        consume(st[r] + lane < en[r], head[r]);
        for (int i0 = st[r] + 64; i0 < en[r]; i0 += 64) {
            const int i = i0 + lane;
            const float4 c = rec[min(i, n - 1)];
            consume(i < en[r], c);
        }
    }
    // lanes 0 .. cnt - 1 hold the hits in index order; pad with the first hit, or 0 for an empty ball (pointnet2_utils.py:246)
    const int first = __builtin_amdgcn_readfirstlane(L);
    const int fill = first == 0x7fffffff ? 0 : first;
    if (lane < nsample) idx_out[((size_t)b * m + s) * nsample + lane] = L == 0x7fffffff ? fill : L;
}

// workspace: NULL, or b * l3d-documented bytes (16 n + 16 448 per cloud) for the cell-list kernels (taken when n >= 2048 and nsample <= 64)
extern "C" int l3d_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                              const float *xyz, int32_t *idx, void *workspace, l3d_stream_t stream)
{
    L3D_REQUIRE(new_xyz && xyz && idx && b > 0 && n > 0 && m > 0 && nsample > 0);
    const float r2 = radius * radius;                       // ball_query_gpu.cu:24
    if (workspace && n >= 2048 && nsample <= 64 && b <= 65535 && radius > 0.f && ((((size_t)workspace) & 15) == 0)) {
        const size_t per = (bq_cells_ws_per_cloud(n) + 15) & ~(size_t)15;
        hipLaunchKernelGGL(bq_cells_build_kernel, dim3((unsigned)b), dim3(1024), 0, (hipStream_t)stream, n, radius, xyz,
                           (unsigned char *)workspace, per);
        hipLaunchKernelGGL(bq_cells_query_kernel, dim3((unsigned)l3d_divup(m, 4), (unsigned)b), dim3(256), 0, (hipStream_t)stream, n, m, r2,
                           nsample, new_xyz, (const unsigned char *)workspace, per, idx);
        return l3d_check_launch();
    }
    if (n >= 1024 && nsample <= 64) {
        hipLaunchKernelGGL(ball_query_sliced_kernel<0>, dim3(l3d_divup(m, 64), b), dim3(64 * BQ_W), bq_sliced_lds(nsample),
                           (hipStream_t)stream, n, m, r2, nsample, new_xyz, xyz, (const int64_t *)nullptr, (void *)idx,
                           (int64_t *)nullptr);
        return l3d_check_launch();
    }
    hipLaunchKernelGGL(ball_query_kernel<0>, dim3(l3d_divup(m, 64), b), dim3(64), 0,
                       (hipStream_t)stream, n, m, r2, nsample, new_xyz, xyz,
                       (const int64_t *)nullptr, (void *)idx, (int64_t *)nullptr);
    return l3d_check_launch();
}

extern "C" int l3d_query_ball_point(float radius, int nsample, const float *xyz,
                                    const float *new_xyz, int B, int N, int S,
                                    const int64_t *itself_indices, int64_t *idx, int64_t *cnt,
                                    l3d_stream_t stream)
{
    L3D_REQUIRE(new_xyz && xyz && idx && B > 0 && N > 0 && S > 0 && nsample > 0);
    // python `radius ** 2` is evaluated in double, then compared against an fp32 tensor
    const float r2 = (float)((double)radius * (double)radius);
    if (N >= 1024 && nsample <= 64) {
        hipLaunchKernelGGL(ball_query_sliced_kernel<1>, dim3(l3d_divup(S, 64), B), dim3(64 * BQ_W), bq_sliced_lds(nsample),
                           (hipStream_t)stream, N, S, r2, nsample, new_xyz, xyz, itself_indices, (void *)idx, cnt);
        return l3d_check_launch();
    }
    hipLaunchKernelGGL(ball_query_kernel<1>, dim3(l3d_divup(S, 64), B), dim3(64), 0,
                       (hipStream_t)stream, N, S, r2, nsample, new_xyz, xyz, itself_indices,
                       (void *)idx, cnt);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// square_distance: dist[b][i][j] = (-2*dot(s_i,d_j) + |s_i|^2) + |d_j|^2     (a3)
// HBM-write-bound: one thread per 4 consecutive j (float4 store when M % 4 == 0).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void square_distance_kernel(const float *__restrict__ src,
                                                              const float *__restrict__ dst, int N,
                                                              int M, float *__restrict__ out)
{
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const float *s = src + ((size_t)b * N + i) * 3;
    const float *d = dst + ((size_t)b * M + j) * 3;
    const float sx = s[0], sy = s[1], sz = s[2];
    const float dx = d[0], dy = d[1], dz = d[2];
    const float dot = fmaf(sz, dz, fmaf(sy, dy, sx * dx));
    const float ss = (sx * sx + sy * sy) + sz * sz;
    const float dd = (dx * dx + dy * dy) + dz * dz;
    out[((size_t)b * N + i) * M + j] = (-2.0f * dot + ss) + dd;
}

// square_distance for C != 3 (the reference body is generic in C, model_common_utils.py:34-37): the dot product is an
// fma chain over the channels in ascending order (MKL's blocking for K > 3 is not restated: ~1 ulp of the dot).
__global__ __launch_bounds__(256) void square_distance_c_kernel(const float *__restrict__ src,
                                                                const float *__restrict__ dst, int N, int M, int C,
                                                                float *__restrict__ out)
{
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const float *s = src + ((size_t)b * N + i) * C;
    const float *d = dst + ((size_t)b * M + j) * C;
    float dot = 0.f, ss = 0.f, dd = 0.f;
    for (int c = 0; c < C; c++) {
        const float a = s[c], e = d[c];
        dot = c == 0 ? a * e : fmaf(a, e, dot);
        ss = c == 0 ? a * a : ss + a * a;
        dd = c == 0 ? e * e : dd + e * e;
    }
    out[((size_t)b * N + i) * M + j] = (-2.0f * dot + ss) + dd;
}

extern "C" int l3d_square_distance(const float *src, const float *dst, int B, int N, int M, int C,
                                   float *dist, l3d_stream_t stream)
{
    L3D_REQUIRE(src && dst && dist && B > 0 && N > 0 && M > 0 && C > 0 && N <= 65535 && B <= 65535);
    if (C == 3)                                     // the xyz form: the reference's rounding sequence for three coordinates
        hipLaunchKernelGGL(square_distance_kernel, dim3(l3d_divup(M, 256), N, B), dim3(256), 0, (hipStream_t)stream, src, dst, N, M, dist);
    else
        hipLaunchKernelGGL(square_distance_c_kernel, dim3(l3d_divup(M, 256), N, B), dim3(256), 0,
                           (hipStream_t)stream, src, dst, N, M, C, dist);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// compute_density (utils/pointconv_util.py:194-203), fused: density[b][i] = mean_j exp(-d2_ij / (2 bw^2)) / (2.5 bw)
// with d2 = square_distance(xyz, xyz) in the reference's expanded rounding order.  One query per thread, the cloud
// streamed through LDS as (x, y, z, |p|^2); the [B,N,N] matrix of the reference never exists.  VALU/transcendental
// bound: N^2 exps per cloud.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gaussian_density_kernel(const float *__restrict__ xyz, int N, float two_bw2,
                                                               float norm, float *__restrict__ out)
{
    __shared__ float4 tile[1024];
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const float *base = xyz + (size_t)b * N * 3;
    const int ic = min(i, N - 1);
    const float qx = base[ic * 3], qy = base[ic * 3 + 1], qz = base[ic * 3 + 2];
    const float qq = (qx * qx + qy * qy) + qz * qz;
    float acc = 0.f;
    for (int j0 = 0; j0 < N; j0 += 1024) {
        const int tn = min(1024, N - j0);
        __syncthreads();
        l3d_stage_points<4>(base + (size_t)j0 * 3, tn, threadIdx.x, 256,
                            [&](int t, float x, float y, float z) { tile[t] = make_float4(x, y, z, (x * x + y * y) + z * z); });
        __syncthreads();
        for (int t = 0; t < tn; t++) {
            const float4 c = tile[t];
            const float dot = fmaf(qz, c.z, fmaf(qy, c.y, qx * c.x));
            const float d2 = (-2.0f * dot + qq) + c.w;
            acc += expf(-d2 / two_bw2) / norm;
        }
    }
    if (i < N) out[(size_t)b * N + i] = acc / (float)N;
}

extern "C" int l3d_gaussian_density(const float *xyz, int B, int N, float bandwidth, float *density, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && density && B > 0 && N > 0 && bandwidth > 0.f && B <= 65535);
    // the reference divides by the Python doubles 2.0*bw*bw and 2.5*bw, which torch converts to fp32 scalars
    const float two_bw2 = (float)(2.0 * (double)bandwidth * (double)bandwidth), norm = (float)(2.5 * (double)bandwidth);
    hipLaunchKernelGGL(gaussian_density_kernel, dim3(l3d_divup(N, 256), B), dim3(256), 0, (hipStream_t)stream, xyz, N,
                       two_bw2, norm, density);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Grouping / gather (pure data movement: the HBM-bound ops of the path)
// group : out[b][c][s][k] = points[b][c][idx[b][s][k]]     thread per (s,k), loop over a channel
// gather: out[b][c][s]    = points[b][c][idx[b][s]]        slab; idx read once, writes coalesced
// ---------------------------------------------------------------------------------------------
#define GP_CCHUNK 16
__global__ __launch_bounds__(256) void group_points_kernel(int c, int n, int total /*S*K*/,
                                                           const float *__restrict__ points,
                                                           const int32_t *__restrict__ idx,
                                                           float *__restrict__ out)
{
    const int b = blockIdx.z;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c0 = blockIdx.y * GP_CCHUNK;
    const int c1 = min(c, c0 + GP_CCHUNK);
    const int j = idx[(size_t)b * total + e];
    const float *p = points + ((size_t)b * c + c0) * n + j;
    float *o = out + ((size_t)b * c + c0) * total + e;
    for (int cc = c0; cc < c1; cc++, p += n, o += total) *o = *p;
}

__global__ __launch_bounds__(256) void group_points_grad_kernel(int c, int n, int total,
                                                                const float *__restrict__ grad_out,
                                                                const int32_t *__restrict__ idx,
                                                                float *__restrict__ grad_points)
{
    const int b = blockIdx.z;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c0 = blockIdx.y * GP_CCHUNK;
    const int c1 = min(c, c0 + GP_CCHUNK);
    const int j = idx[(size_t)b * total + e];
    float *p = grad_points + ((size_t)b * c + c0) * n + j;
    const float *o = grad_out + ((size_t)b * c + c0) * total + e;
    for (int cc = c0; cc < c1; cc++, p += n, o += total) atomicAdd(p, *o);
}

static int launch_group(bool grad, int b, int c, int n, int total, const float *src,
                        const int32_t *idx, float *dst, hipStream_t st)
{
    dim3 grid(l3d_divup(total, 256), l3d_divup(c, GP_CCHUNK), b);
    if (grad) {
        hipError_t e = hipMemsetAsync(dst, 0, sizeof(float) * (size_t)b * c * n, st);
        if (e != hipSuccess) { g_l3d_last_hip_error = (int)e; return L3D_ERR_LAUNCH; }
        hipLaunchKernelGGL(group_points_grad_kernel, grid, dim3(256), 0, st, c, n, total, src, idx, dst);
    } else {
        hipLaunchKernelGGL(group_points_kernel, grid, dim3(256), 0, st, c, n, total, src, idx, dst);
    }
    return l3d_check_launch();
}

extern "C" int l3d_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                                const int32_t *idx, float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(points && idx && out && b > 0 && c > 0 && n > 0 && npoints > 0 && nsample > 0);
    return launch_group(false, b, c, n, npoints * nsample, points, idx, out, (hipStream_t)stream);
}
extern "C" int l3d_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                     const float *grad_out, const int32_t *idx, float *grad_points,
                                     l3d_stream_t stream)
{
    L3D_REQUIRE(grad_out && idx && grad_points && b > 0 && c > 0 && n > 0 && npoints > 0 && nsample > 0);
    return launch_group(true, b, c, n, npoints * nsample, grad_out, idx, grad_points, (hipStream_t)stream);
}
extern "C" int l3d_gather_points(int b, int c, int n, int npoints, const float *points,
                                 const int32_t *idx, float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(points && idx && out && b > 0 && c > 0 && n > 0 && npoints > 0);
    return launch_group(false, b, c, n, npoints, points, idx, out, (hipStream_t)stream);
}
extern "C" int l3d_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                      const int32_t *idx, float *grad_points, l3d_stream_t stream)
{
    L3D_REQUIRE(grad_out && idx && grad_points && b > 0 && c > 0 && n > 0 && npoints > 0);
    return launch_group(true, b, c, n, npoints, grad_out, idx, grad_points, (hipStream_t)stream);
}

// index_points: points [B,N,C] (channel-last), idx [B,S] int64 -> out [B,S,C]     (a5)
__global__ __launch_bounds__(256) void index_points_kernel(const float *__restrict__ points,
                                                           const int64_t *__restrict__ idx, int N,
                                                           int C, int S, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over S*C
    if (e >= (size_t)S * C) return;
    const int s = (int)(e / C), cc = (int)(e % C);
    const int64_t j = idx[(size_t)b * S + s];
    out[(size_t)b * S * C + e] = points[((size_t)b * N + j) * C + cc];
}

extern "C" int l3d_index_points(const float *points, const int64_t *idx, int B, int N, int C, int S,
                                float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(points && idx && out && B > 0 && N > 0 && C > 0 && S > 0);
    hipLaunchKernelGGL(index_points_kernel, dim3(l3d_divup((long)S * C, 256), B), dim3(256), 0,
                       (hipStream_t)stream, points, idx, N, C, S, out);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Farthest point sampling.  One workgroup (up to 1024 threads = 16 waves) per cloud; the cloud
// and the running min-distance live in VGPRs (PPT points per thread), so a round is
//   PPT x (3 sub, 3 mul, 2 add, min, cmp, 2 cndmask)  +  wave arg-max (6 DPP/shuffle steps)
//   +  one LDS exchange between the <=16 waves  +  2 barriers
// instead of the reference's global-memory `temp` round trip and 10-level LDS tree
// (sampling_gpu.cu:139-203).  Arg-max ties resolve to the lowest index.
// OUT64: int64 centroids + optional start index (torch twin T7), else int32, start 0 (K12).
// ---------------------------------------------------------------------------------------------
// 16-lane (DPP row) reductions on integer keys: non-negative fp32 values order like their bit
// patterns, so the arg-max runs on the integer pipe with quad_perm / row_half_mirror / row_mirror
// DPP modifiers -- no ds_bpermute, no LDS round trip.
__device__ __forceinline__ int row16_max_i(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));    // quad_perm(1,0,3,2)
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));    // quad_perm(2,3,0,1)
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));   // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));   // row_mirror
    return v;
}
__device__ __forceinline__ int row16_min_i(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
    v = row16_max_i(v);
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_min_i(int v)
{
    v = row16_min_i(v);
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// LDS_XYZ: the cloud is also kept in LDS (12 B/point, n <= 12288) so the coordinates of the point
// picked in the previous round come from a wave-uniform LDS read instead of a dependent global load.
// Launch bound: PPT >= 8 is only ever launched with 512 threads (launch_fps: n >= 4096), and under a 1024-thread bound (128 VGPRs)
// the 32-points-per-thread instantiation spilled 118-186 VGPRs (profiles/round2_kernel_resources.txt).
template <int PPT, bool OUT64, bool LDS_XYZ>
__global__ __launch_bounds__(PPT >= 8 ? 512 : 1024) void fps_kernel(int n, int m, const float *__restrict__ xyz,
                                                   const int64_t *__restrict__ start,
                                                   float *__restrict__ temp,
                                                   void *__restrict__ out)
{
    extern __shared__ float sxyz[];              // [3][n] when LDS_XYZ
    __shared__ int wv[2][16];
    __shared__ int wi[2][16];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int b = blockIdx.x;
    const float *p = xyz + (size_t)b * n * 3;
    float px[PPT], py[PPT], pz[PPT], dmin[PPT];
#pragma unroll
    for (int u = 0; u < PPT; u++) {              // unconditional loads from clamped indices: all PPT points of a thread in flight
        const int kc = min(tid + u * nthr, n - 1);  // (`ok ? p[..] : 0` is a branch with a wait at its join: PPT round trips in a row)
        px[u] = p[kc * 3];
        py[u] = p[kc * 3 + 1];
        pz[u] = p[kc * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const int k = tid + u * nthr;
        const bool ok = k < n;
        const float in = ok ? 1.f : 0.f;
        px[u] *= in; py[u] *= in; pz[u] *= in;
        dmin[u] = ok ? 1e10f : -1.f;          // out-of-range slots can never win the arg-max
        if (LDS_XYZ && ok) { sxyz[k] = px[u]; sxyz[n + k] = py[u]; sxyz[2 * n + k] = pz[u]; }
    }
    if (tid < 32) { wv[tid >> 4][tid & 15] = (int)0x80000000; wi[tid >> 4][tid & 15] = 0x7fffffff; }
    int old = (OUT64 && start) ? (int)start[b] : 0;
    if (tid == 0) {
        if (OUT64) ((int64_t *)out)[(size_t)b * m] = old; else ((int32_t *)out)[(size_t)b * m] = old;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    // Equal maxima.  OUT64 (the torch twin, model_common_utils.py:58-82: torch.max): the lowest index.  !OUT64 (pointnet2's
    // kernel, sampling_gpu.cu:86-205): its thread `tid` scans k = tid, tid + T, ... with a strict '>', and its tree merges slot s
    // with slot s + h (h = T/2 ... 1) keeping the lower slot's candidate unless the other is strictly larger -- two tied
    // candidates meet where their thread ids first agree modulo h and the one with bit h clear survives, so the winner is the
    // smallest (bit-reversed (k mod T), k), T = opt_n_threads(n) = min(2^floor(log2 n), 1024) (cuda_utils.h:6-14).  Clipped clouds
    // (config 5) put duplicate points on the corners of the box, exactly where sampling goes first.  tkey() orders candidates
    // accordingly; it is only evaluated on the (rare) tie paths, and the per-thread scan visits its slots in tkey order (for 512
    // threads and T = 1024: k mod 1024 = tid + 512 (u & 1), bit 9 decides last -- even slots, then odd ones).
    const int lt = OUT64 ? 0 : min(31 - __builtin_clz(n | 1), 10);
    auto brev = [&](int x) { return lt ? (int)(__builtin_bitreverse32((unsigned)x) >> (32 - lt)) : 0; };
    auto tkey = [&](int k) { return OUT64 ? k : ((brev(k & ((1 << lt) - 1)) << 15) | (k >> lt)); };
    auto tinv = [&](int p) { return OUT64 ? p : (((p & 0x7fff) << lt) | brev(p >> 15)); };
    for (int j = 1; j < m; j++) {
        float x1, y1, z1;                       // wave-uniform
        if (LDS_XYZ) { x1 = sxyz[old]; y1 = sxyz[n + old]; z1 = sxyz[2 * n + old]; }
        else { x1 = p[old * 3]; y1 = p[old * 3 + 1]; z1 = p[old * 3 + 2]; }
        int best = (int)0x80000000;             // bit pattern of the running max (d2 >= 0, or -1 for padding)
        int besti = 0;
        if constexpr (PPT >= 2) {
            // two points per packed-fp32 instruction (v_pk_add_f32 / v_pk_mul_f32; same IEEE results): the
            // update of 8192 points is VALU-throughput-bound inside the one CU a cloud owns
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 x2 = {x1, x1}, y2 = {y1, y1}, z2 = {z1, z1};
#pragma unroll
            for (int u = 0; u < PPT; u += 2) {
                const f32x2 dx = f32x2{px[u], px[u + 1]} - x2, dy = f32x2{py[u], py[u + 1]} - y2, dz = f32x2{pz[u], pz[u + 1]} - z2;
                const f32x2 d = (dx * dx + dy * dy) + dz * dz;
                dmin[u] = fminf(d[0], dmin[u]);
                dmin[u + 1] = fminf(d[1], dmin[u + 1]);
            }
            // arg-max over the thread's slots with a strict '>', visited in the tie order above: ascending k, except for the
            // pointnet2 op at 512 threads (n >= 4096: the reference runs 1024 threads), where k mod 1024 = tid + 512 (u & 1)
            constexpr bool EVEN_FIRST = !OUT64 && PPT >= 8;
#pragma unroll
            for (int s = 0; s < PPT; s++) {
                const int u = EVEN_FIRST ? (s < PPT / 2 ? 2 * s : 2 * (s - PPT / 2) + 1) : s;
                const int key = __builtin_bit_cast(int, dmin[u]);
                const bool gt = key > best;
                best = gt ? key : best;
                besti = gt ? tid + u * nthr : besti;
            }
        } else {
#pragma unroll
            for (int u = 0; u < PPT; u++) {
                const float dx = px[u] - x1, dy = py[u] - y1, dz = pz[u] - z1;
                const float d = (dx * dx + dy * dy) + dz * dz;
                const float d2 = fminf(d, dmin[u]);
                dmin[u] = d2;
                const int key = __builtin_bit_cast(int, d2);
                const bool gt = key > best;         // ascending k within a thread: lowest index wins
                best = gt ? key : best;
                besti = gt ? tid + u * nthr : besti;
            }
        }
        // wave arg-max (lowest index among equal maxima), then the <= 16 waves through LDS
        const int wmax = wave_max_i(best);
        // index of the maximum: almost always ONE lane holds it -- then its index is a ballot + find-first
        // + readlane (3 instructions) instead of a second 6-step DPP reduction; only an exact tie between
        // lanes takes the reduction (lowest index among equal maxima, as before)
        const unsigned long long hit = __ballot(best == wmax);
        int widx;
        if (__builtin_popcountll(hit) == 1)
            widx = __builtin_amdgcn_readlane(besti, __builtin_ctzll(hit));
        else
            widx = tinv(wave_min_i(best == wmax ? tkey(besti) : 0x7fffffff));
        const int buf = j & 1;
        if (lane == 0) { wv[buf][wave] = wmax; wi[buf][wave] = widx; }
        __syncthreads();
        const int ev = wv[buf][lane & 15], ei = wi[buf][lane & 15];      // every 16-lane row sees all waves
        const int bmax = row16_max_i(ev);
        const unsigned hit16 = (unsigned)(__ballot(ev == bmax) & 0xffffull);      // row 0 sees all 16 waves
        if (__builtin_popcount(hit16) == 1)
            old = __builtin_amdgcn_readlane(ei, __builtin_ctz(hit16));
        else
            old = __builtin_amdgcn_readfirstlane(tinv(row16_min_i(ev == bmax ? tkey(ei) : 0x7fffffff)));
        if (tid == 0) {
            if (OUT64) ((int64_t *)out)[(size_t)b * m + j] = old; else ((int32_t *)out)[(size_t)b * m + j] = old;
        }
        // double-buffered wv/wi: the next round writes the other buffer, so one barrier per round
    }
    if (temp) {
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const int k = tid + u * nthr;
            if (k < n) temp[(size_t)b * n + k] = dmin[u];
        }
    }
}

template <bool OUT64>
static int launch_fps(int b, int n, int m, const float *xyz, const int64_t *start, float *temp,
                      void *out, hipStream_t st)
{
    // threads: multiple of 64, <= 1024; points per thread: 1,2,4,8,16,32
    // large clouds: 512 threads x 16 points -- the per-wave arg-max reduction (~40 instructions) is amortised over
    // twice the points and the second level sees 8 waves (the round is VALU-throughput-bound inside one CU)
    int nthr = n >= 4096 ? 512 : (n >= 1024 ? 1024 : ((n + 63) / 64) * 64);      // (256 x 32 points: 2.2 ms -- one wave per SIMD)
    int ppt = (n + nthr - 1) / nthr;
#define L3D_FPS_CASE(P)                                                                           \
    if (ppt <= P) {                                                                               \
        if (n <= 12288)                                                                           \
            hipLaunchKernelGGL((fps_kernel<P, OUT64, true>), dim3(b), dim3(nthr),                 \
                               sizeof(float) * 3 * (size_t)n, st, n, m, xyz, start, temp, out);   \
        else                                                                                      \
            hipLaunchKernelGGL((fps_kernel<P, OUT64, false>), dim3(b), dim3(nthr), 0, st, n, m,   \
                               xyz, start, temp, out);                                            \
        return l3d_check_launch();                                                                \
    }
    L3D_FPS_CASE(1)
    L3D_FPS_CASE(2)
    L3D_FPS_CASE(4)
    L3D_FPS_CASE(8)
    L3D_FPS_CASE(16)
    L3D_FPS_CASE(32)
#undef L3D_FPS_CASE
    return L3D_ERR_UNSUPPORTED;       // > 32768 points per cloud
}

extern "C" int l3d_furthest_point_sampling(int b, int n, int m, const float *points, float *temp,
                                           int32_t *idx, l3d_stream_t stream)
{
    L3D_REQUIRE(points && idx && b > 0 && n > 0 && m > 0);
    return launch_fps<false>(b, n, m, points, nullptr, temp, idx, (hipStream_t)stream);
}

extern "C" int l3d_farthest_point_sample(const float *xyz, int B, int N, int npoint,
                                         const int64_t *start, float *temp, int64_t *centroids,
                                         l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && centroids && B > 0 && N > 0 && npoint > 0);
    return launch_fps<true>(B, N, npoint, xyz, start, temp, centroids, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// three_interpolate: out[b][c][n] = (w0*p[i0] + w1*p[i1]) + w2*p[i2]     (K15)  + grad (K16)
// thread per n, loop over a channel slab (idx/weight read once, coalesced writes)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void three_interpolate_kernel(int c, int m, int n,
                                                                const float *__restrict__ points,
                                                                const int32_t *__restrict__ idx,
                                                                const float *__restrict__ weight,
                                                                float *__restrict__ out, long out_bstride)
{
    const int b = blockIdx.z;
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n) return;
    const int c0 = blockIdx.y * GP_CCHUNK, c1 = min(c, c0 + GP_CCHUNK);
    const int32_t *ix = idx + ((size_t)b * n + pt) * 3;
    const float *w = weight + ((size_t)b * n + pt) * 3;
    const int i0 = ix[0], i1 = ix[1], i2 = ix[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    for (int cc = c0; cc < c1; cc++) {
        const float *p = points + ((size_t)b * c + cc) * m;
        out[(size_t)b * out_bstride + (size_t)cc * n + pt] = (w0 * p[i0] + w1 * p[i1]) + w2 * p[i2];
    }
}

// The same with the channel rows staged in LDS: a workgroup owns CG channels of one cloud (CG * m floats, <= 64 KB) and walks
// ALL n points, so the 3 random 4-byte reads per output are LDS reads instead of three trips through the texture path
// (FlowNet3D's feature propagation, models/flownet3d.py:268: 256 channels x 8192 points from 1024 -- 208 -> 75 us).  idx and
// weights of a point stay in registers across the CG channels; stores are coalesced over points.  Same operation order.
__global__ __launch_bounds__(256) void three_interpolate_lds_kernel(int c, int m, int n, int CG,
                                                                    const float *__restrict__ points,
                                                                    const int32_t *__restrict__ idx,
                                                                    const float *__restrict__ weight,
                                                                    float *__restrict__ out, long out_bstride)
{
    extern __shared__ float tirow[];                             // [CG][m]
    const int b = blockIdx.y, c0 = blockIdx.x * CG;
    const int cg = min(CG, c - c0);
    const float *pb = points + ((size_t)b * c + c0) * m;
    for (int e = threadIdx.x; e < cg * m; e += 256) tirow[e] = pb[e];
    __syncthreads();
    float *ob = out + (size_t)b * out_bstride + (size_t)c0 * n;
    for (int pt = threadIdx.x; pt < n; pt += 256) {
        const int32_t *ix = idx + ((size_t)b * n + pt) * 3;
        const float *w = weight + ((size_t)b * n + pt) * 3;
        const int i0 = ix[0], i1 = ix[1], i2 = ix[2];
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        for (int cc = 0; cc < cg; cc++) {
            const float *p = tirow + cc * m;
            ob[(size_t)cc * n + pt] = (w0 * p[i0] + w1 * p[i1]) + w2 * p[i2];
        }
    }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points)
{
    const int b = blockIdx.z;
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n) return;
    const int c0 = blockIdx.y * GP_CCHUNK, c1 = min(c, c0 + GP_CCHUNK);
    const int32_t *ix = idx + ((size_t)b * n + pt) * 3;
    const float *w = weight + ((size_t)b * n + pt) * 3;
    const int i0 = ix[0], i1 = ix[1], i2 = ix[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    for (int cc = c0; cc < c1; cc++) {
        const float g = grad_out[((size_t)b * c + cc) * n + pt];
        float *gp = grad_points + ((size_t)b * c + cc) * m;
        atomicAdd(gp + i0, g * w0);
        atomicAdd(gp + i1, g * w1);
        atomicAdd(gp + i2, g * w2);
    }
}

// LDS-staged kernel when a useful number of channel rows fits (CG >= 4 rows of m floats in 64 KB) and there are enough points to
// amortise the staging; else one thread per point with global gathers
static void ti_launch(int b, int c, int m, int n, const float *points, const int32_t *idx, const float *weight, float *out,
                      long out_bstride, hipStream_t st)
{
    int cg = 16384 / m;
    if (cg > 16) cg = 16;
    if (cg >= 4 && n >= 4 * m && b <= 65535)
        hipLaunchKernelGGL(three_interpolate_lds_kernel, dim3(l3d_divup(c, cg), b), dim3(256), (size_t)cg * m * 4, st, c, m, n, cg,
                           points, idx, weight, out, out_bstride);
    else
        hipLaunchKernelGGL(three_interpolate_kernel, dim3(l3d_divup(n, 256), l3d_divup(c, GP_CCHUNK), b), dim3(256), 0, st, c, m, n,
                           points, idx, weight, out, out_bstride);
}

extern "C" int l3d_three_interpolate(int b, int c, int m, int n, const float *points,
                                     const int32_t *idx, const float *weight, float *out,
                                     l3d_stream_t stream)
{
    L3D_REQUIRE(points && idx && weight && out && b > 0 && c > 0 && m > 0 && n > 0);
    ti_launch(b, c, m, n, points, idx, weight, out, (long)c * n, (hipStream_t)stream);
    return l3d_check_launch();
}

// three_interpolate followed by torch.cat([interpolated, skip], dim=1) (PointNetFeaturePropogation,
// models/flownet3d.py:268-272): the interpolation writes its channels of the [B, c + c1, n] result directly,
// the skip features are one strided device copy.
extern "C" int l3d_three_interpolate_concat(int b, int c, int m, int n, const float *points, const int32_t *idx,
                                            const float *weight, const float *skip, int c1, float *out,
                                            l3d_stream_t stream)
{
    L3D_REQUIRE(points && idx && weight && out && b > 0 && c > 0 && m > 0 && n > 0 && c1 >= 0 && (c1 == 0 || skip));
    hipStream_t st = (hipStream_t)stream;
    const long bs = (long)(c + c1) * n;
    ti_launch(b, c, m, n, points, idx, weight, out, bs, st);
    if (c1 > 0) {
        hipError_t e = hipMemcpy2DAsync(out + (size_t)c * n, (size_t)bs * 4, skip, (size_t)c1 * n * 4, (size_t)c1 * n * 4, b,
                                        hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { g_l3d_last_hip_error = (int)e; return L3D_ERR_LAUNCH; }
    }
    return l3d_check_launch();
}

extern "C" int l3d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                          const int32_t *idx, const float *weight,
                                          float *grad_points, l3d_stream_t stream)
{
    L3D_REQUIRE(grad_out && idx && weight && grad_points && b > 0 && c > 0 && m > 0 && n > 0);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * m, st);
    if (e != hipSuccess) { g_l3d_last_hip_error = (int)e; return L3D_ERR_LAUNCH; }
    hipLaunchKernelGGL(three_interpolate_grad_kernel,
                       dim3(l3d_divup(n, 256), l3d_divup(c, GP_CCHUNK), b), dim3(256), 0, st, c, n, m,
                       grad_out, idx, weight, grad_points);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// QueryAndGroup's tail in one pass (pointnet2_utils.py:274-292): out [B, 3+C, S, K] with
//   out[b][c][s][k] = xyz[b][idx[b][s][k]][c] - new_xyz[b][s][c]          (c < 3, use_xyz)
//   out[b][3+c][s][k] = features[b][c][idx[b][s][k]]
// instead of two grouping launches, a broadcast subtraction and a torch.cat copy of the result.
// One thread per (s, k); channels in a loop: every store is coalesced over (s, k).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_concat_kernel(const float *__restrict__ xyz /*[B,N,3]*/,
                                                           const float *__restrict__ new_xyz /*[B,S,3]*/,
                                                           const float *__restrict__ feat /*[B,C,N] or null*/,
                                                           const int32_t *__restrict__ idx /*[B,S,K]*/, int N, int S,
                                                           int K, int C, int use_xyz, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;                 // s*K + k
    if (e >= S * K) return;
    const int s_ = e / K;
    const int j = idx[((size_t)b * S) * K + e];
    const size_t SK = (size_t)S * K;
    const int c0 = use_xyz ? 3 : 0;
    float *ob = out + (size_t)b * (c0 + C) * SK + e;
    if (use_xyz) {
        const float *p = xyz + ((size_t)b * N + j) * 3, *q = new_xyz + ((size_t)b * S + s_) * 3;
        ob[0] = p[0] - q[0];
        ob[SK] = p[1] - q[1];
        ob[2 * SK] = p[2] - q[2];
    }
    const float *fb = feat + (size_t)b * C * N + j;
    for (int c = 0; c < C; c++) ob[(size_t)(c0 + c) * SK] = fb[(size_t)c * N];
}

extern "C" int l3d_group_concat(const float *xyz, const float *new_xyz, const float *features, const int32_t *idx,
                                int B, int N, int S, int K, int C, int use_xyz, float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && new_xyz && idx && out && B > 0 && N > 0 && S > 0 && K > 0 && C >= 0 && (C == 0 || features) &&
                (use_xyz || C > 0));
    if (B > 65535) return L3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(group_concat_kernel, dim3(l3d_divup((long)S * K, 256), B), dim3(256), 0, (hipStream_t)stream, xyz,
                       new_xyz, features, idx, N, S, K, C, use_xyz, out);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// The grouped conv input of FlowNet3D's FlowEmbedding / PointNetSetUpConv (models/flownet3d.py:125-180,
// :182-242) in one pass: two grouping_operations, a broadcast subtraction, a repeat and one or two torch.cat
// copies become
//   order 0:  out = [ xyz[idx] - new_xyz | features[idx] | centre (broadcast over K) ]     (FlowEmbedding, concat)
//   order 1:  out = [ features[idx] | xyz[idx] - new_xyz | centre ]                        (PointNetSetUpConv)
// out [B, 3 + C + C1, S, K]; one thread per (s, k), channels in a loop, every store coalesced over (s, k).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_concat2_kernel(const float *__restrict__ xyz /*[B,N,3]*/,
                                                            const float *__restrict__ new_xyz /*[B,S,3]*/,
                                                            const float *__restrict__ feat /*[B,C,N]*/,
                                                            const float *__restrict__ centre /*[B,C1,S] or null*/,
                                                            const int32_t *__restrict__ idx /*[B,S,K]*/, int N, int S,
                                                            int K, int C, int C1, int order, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;                 // s*K + k
    if (e >= S * K) return;
    const int s_ = e / K;
    const int j = idx[((size_t)b * S) * K + e];
    const size_t SK = (size_t)S * K;
    float *ob = out + (size_t)b * (3 + C + C1) * SK + e;
    const float *p = xyz + ((size_t)b * N + j) * 3, *q = new_xyz + ((size_t)b * S + s_) * 3;
    float *oxyz = ob + (order == 0 ? 0 : (size_t)C * SK);
    oxyz[0] = p[0] - q[0];
    oxyz[SK] = p[1] - q[1];
    oxyz[2 * SK] = p[2] - q[2];
    float *of = ob + (order == 0 ? 3 * SK : 0);
    const float *fb = feat + (size_t)b * C * N + j;
    for (int c = 0; c < C; c++) of[(size_t)c * SK] = fb[(size_t)c * N];
    if (C1 > 0) {
        float *oc = ob + (size_t)(3 + C) * SK;
        const float *cb = centre + (size_t)b * C1 * S + s_;
        for (int c = 0; c < C1; c++) oc[(size_t)c * SK] = cb[(size_t)c * S];
    }
}

extern "C" int l3d_group_concat2(const float *xyz, const float *new_xyz, const float *features, const float *centre,
                                 const int32_t *idx, int B, int N, int S, int K, int C, int C1, int order, float *out,
                                 l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && new_xyz && features && idx && out && B > 0 && N > 0 && S > 0 && K > 0 && C > 0 && C1 >= 0 &&
                (C1 == 0 || centre) && (order == 0 || order == 1));
    if (B > 65535) return L3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(group_concat2_kernel, dim3(l3d_divup((long)S * K, 256), B), dim3(256), 0, (hipStream_t)stream, xyz,
                       new_xyz, features, centre, idx, N, S, K, C, C1, order, out);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// First layer of a grouped shared MLP WITHOUT the grouped tensor (models/flownet3d.py:125-180 FlowEmbedding, :182-242
// PointNetSetUpConv, :73-123 PointNetSetAbstraction).  The 1x1 conv acts on [xyz[idx] - centre | feat[idx] | centre_feat],
// so -- like the PRNet layer below -- it splits into per-POINT products computed before the grouping,
//     U = (s W_feat) feat  over the source points,   V = (s W_centre) centre_feat + t  over the centres,   Wx = s W_xyz,
// and the layer's output for neighbour k of centre i is  act(U[idx_ik] + V_i + Wx (xyz[idx_ik] - centre_i)):
// K times fewer conv flops and no [B, 3+C+C1, S, K] tensor (2.2 GB for FlowNet3D's su3 at B = 32, read once more by conv1).
// Channel-last on both sides: a gathered row of U and an output row are C1 contiguous floats (16 bytes per lane); the
// output [B, S K, C1] is what the next layer's kernels take as channel_last input.  A thread owns 4 channels and walks rows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_first_layer_kernel(const float *__restrict__ U /*[B,N,C1]*/,
                                                                const float *__restrict__ V /*[B,S,C1] or null*/,
                                                                const float *__restrict__ shift /*[C1] or null*/,
                                                                const float *__restrict__ wx /*[C1][3]*/,
                                                                const float *__restrict__ xyz /*[B,N,3]*/,
                                                                const float *__restrict__ new_xyz /*[B,S,3]*/,
                                                                const int32_t *__restrict__ idx /*[B,S,K]*/, int N, int S, int K,
                                                                int C1, int act, float *__restrict__ out /*[B,S*K,C1]*/)
{
    const int lpr = C1 >> 2, rpb = 256 / lpr;                    // lanes per row, rows per pass of the workgroup
    const int lc = threadIdx.x % lpr, rsub = threadIdx.x / lpr;
    if (rsub >= rpb) return;
    const int b = blockIdx.y;
    f32x4 w0, w1, w2, sh;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const float *w = wx + (size_t)(4 * lc + u) * 3;
        w0[u] = w[0]; w1[u] = w[1]; w2[u] = w[2];
        sh[u] = shift ? shift[4 * lc + u] : 0.f;
    }
    const long SK = (long)S * K;
    const int32_t *ib = idx + (size_t)b * SK;
    for (long e = (long)blockIdx.x * rpb + rsub; e < SK; e += (long)gridDim.x * rpb) {
        const int s_ = (int)(e / K), j = ib[e];
        const float *p = xyz + ((size_t)b * N + j) * 3, *q = new_xyz + ((size_t)b * S + s_) * 3;
        const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
        f32x4 r = *(const f32x4 *)(U + ((size_t)b * N + j) * C1 + 4 * lc) + sh;
        if (V) r += *(const f32x4 *)(V + ((size_t)b * S + s_) * C1 + 4 * lc);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float v = fmaf(w2[u], dz, fmaf(w1[u], dy, fmaf(w0[u], dx, r[u])));
            r[u] = act ? l3d_act(v, act) : v;
        }
        *(f32x4 *)(out + ((size_t)b * SK + e) * C1 + 4 * lc) = r;
    }
}

extern "C" int l3d_group_first_layer(const float *U, const float *V, const float *shift, const float *wx, const float *xyz,
                                     const float *new_xyz, const int32_t *idx, int B, int N, int S, int K, int C1, int relu,
                                     float *out, l3d_stream_t stream)
{
    L3D_REQUIRE(U && wx && xyz && new_xyz && idx && out && B > 0 && N > 0 && S > 0 && K > 0 && C1 > 0);
    if (B > 65535 || (C1 & 3) || C1 > 1024 || (((size_t)U | (size_t)V | (size_t)out) & 15)) return L3D_ERR_UNSUPPORTED;
    const int rpb = 256 / (C1 >> 2);
    long nblk = l3d_divup((long)S * K, rpb);
    if (nblk > 2048) nblk = 2048;
    hipLaunchKernelGGL(group_first_layer_kernel, dim3((unsigned)nblk, B), dim3(256), 0, (hipStream_t)stream, U, V, shift, wx, xyz,
                       new_xyz, idx, N, S, K, C1, relu, out);
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// One dynamic-graph EdgeConv layer on LINEAR pre-activations (PRNet's DGCNN, models/prnet.py:76-97:
// get_graph_feature -> conv2d 1x1 (2C -> Cout, no bias) -> BN -> leaky_relu -> max over k).
// The conv acts on (neighbour ; centre), so it splits into two per-POINT products
//     P = (s * W[:, :C]) x,     Q = (s * W[:, C:]) x + t          (BN scale s, shift t folded in),
// and  max_j lrelu(P[idx_ij] + Q_i)  ==  lrelu(max_j P[idx_ij] + Q_i)  exactly (x + c and lrelu are monotone,
// rounding included): k times fewer conv flops and no [B,2C,N,k] tensor.  This kernel is the second half:
//     out[b][co][i] = act(max_j pq[b][co][idx[b][i][j]] + pq[b][Cout + co][i]).
// A workgroup stages CG channel rows of P (N floats each) in LDS and gathers from there (random 4-byte
// LDS reads); a thread owns a point, its k indices stay in registers across the CG channels; stores are
// coalesced over points.  `out` takes a batch stride so a layer can write its slice of the cat buffer.
// (Tried, slower end to end: four channels interleaved per point for 16-byte LDS gathers; staging the
// index tile through LDS -- both cut occupancy more than they saved.)
// ---------------------------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(256) void edge_gather_max_kernel(const float *__restrict__ pq, const int64_t *__restrict__ idx,
                                                              int Cout, int N, int k, int CG, int act,
                                                              float *__restrict__ out, long out_bstride)
{
    extern __shared__ float prow[];                              // [CG][N]
    const int b = blockIdx.y, co0 = blockIdx.x * CG;
    const int cg = min(CG, Cout - co0);
    const float *pb = pq + ((size_t)b * 2 * Cout + co0) * N;
    for (int e = threadIdx.x; e < cg * N; e += 256) prow[e] = pb[e];
    __syncthreads();
    const float *qb = pq + ((size_t)b * 2 * Cout + Cout + co0) * N;
    float *ob = out + (size_t)b * out_bstride + (size_t)co0 * N;
    for (int i = threadIdx.x; i < N; i += 256) {
        const int64_t *ip = idx + ((size_t)b * N + i) * k;
        int nb[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; j++) nb[j] = (int)ip[j < k ? j : 0];
        for (int c = 0; c < cg; c++) {
            const float *pr = prow + c * N;
            float m = pr[nb[0]];
#pragma unroll
            for (int j = 1; j < KMAX; j++) m = fmaxf(m, pr[nb[j]]);   // j >= k repeats neighbour 0
            float v = m + qb[(size_t)c * N + i];
            if (act) v = l3d_act(v, act);
            ob[(size_t)c * N + i] = v;
        }
    }
}

extern "C" int l3d_edge_gather_max(const float *pq, const int64_t *idx, int B, int Cout, int N, int k, int act,
                                   float *out, long out_bstride, l3d_stream_t stream)
{
    L3D_REQUIRE(pq && idx && out && B > 0 && Cout > 0 && N > 0 && k > 0 && out_bstride >= (long)Cout * N);
    if (B > 65535 || k > 40 || (size_t)N * 4 > 128 * 1024) return L3D_ERR_UNSUPPORTED;
    // channel rows per workgroup: as many as fit 64 KiB of LDS, fewer while the grid would leave CUs idle
    int CG = 16;
    while (CG > 1 && ((size_t)CG * N * 4 > 64 * 1024 || (long)l3d_divup(Cout, CG) * B < 1024)) CG >>= 1;
    dim3 grid(l3d_divup(Cout, CG), B), block(256);
    const size_t lds = (size_t)CG * N * 4;
    hipStream_t st = (hipStream_t)stream;
    if (k <= 20) {
        hipLaunchKernelGGL(edge_gather_max_kernel<20>, grid, block, lds, st, pq, idx, Cout, N, k, CG, act, out, out_bstride);
    } else {
        hipLaunchKernelGGL(edge_gather_max_kernel<40>, grid, block, lds, st, pq, idx, Cout, N, k, CG, act, out, out_bstride);
    }
    return l3d_check_launch();
}
