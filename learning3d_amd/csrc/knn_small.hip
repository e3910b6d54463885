// knn_small.hip -- k <= 4 nearest candidates of MANY queries, direct metric: three_nn (k = 3) and knn with k <= 4.
//
// Replaces, for k <= 4:
//   K14 three_nn_kernel_fast  utils/lib/src/interpolate_gpu.cu:81-124   (PointNetFeaturePropogation, flownet3d.py:245-266:
//                                                                         8192 queries x 1024 candidates per cloud at config 5)
//   K13 knn_kernel_fast       interpolate_gpu.cu:9-57  with k <= 4
//
// Both other kernels are built for long lists: knn.hip's two-pass lane kernel took 293 us for 32 x 8192 queries against 1024
// candidates at k = 3, the wave-per-query selection kernel 284 us (a bound search and a rank count per QUERY).  With three
// slots the list is cheaper than either: a query per lane, the candidates as coordinate arrays in LDS read as wave-uniform
// ds_read_b128 (four candidates per read), distances in packed fp32 with the reference's rounding sequence
// ((dx dx + dy dy) + dz dz, no contraction).  A candidate is compared with the lane's current k-th best by v_sub + v_alignbit
// into a 32-candidate bit mask (no scalar round trip, no branch per candidate); after 32 candidates the lanes pop their hits
// together, highest bit = lowest index first, recompute those distances from LDS (same operations, same bits) and insert
// them branch-free into four sorted slots with a strict '<' -- ascending distance, lowest index first under ties.
#include "common.h"

#define KSM_T 2048                      // candidates per LDS tile (24 KB)
enum { KSM_OUT_PAIR = 1, KSM_OUT_POINT = 2 };          // == OUT_KNN_PAIR / OUT_KNN_POINT of knn.hip
typedef float ksm_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void knn_small_kernel(const float *__restrict__ qxyz, const float *__restrict__ cxyz, int Nq, int Nc,
                                                        int k, int out_mode, void *__restrict__ idx_out, float *__restrict__ val_out)
{
    __shared__ __attribute__((aligned(16))) float sc[3][KSM_T];
    const int b = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const bool valid = q < Nq;
    const float *qp = qxyz + ((size_t)b * Nq + (valid ? q : Nq - 1)) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const ksm_f4 nqx = {-qx, -qx, -qx, -qx}, nqy = {-qy, -qy, -qy, -qy}, nqz = {-qz, -qz, -qz, -qz};   // c + (-q): packed adds
    const float *cbase = cxyz + (size_t)b * Nc * 3;

    float bd[4] = {INFINITY, INFINITY, INFINITY, INFINITY};        // ascending; slots >= k are scratch
    int bi[4] = {0, 0, 0, 0};
    unsigned thr = 0x7f800000u;                                    // bits of the k-th best so far

    for (int t0 = 0; t0 < Nc; t0 += KSM_T) {
        const int tn = min(KSM_T, Nc - t0), tp = (tn + 31) & ~31;
        __syncthreads();                                           // the previous tile's reads are done
        for (int e = threadIdx.x; e < tp * 3; e += 256) {
            const int j = e / 3, c = e - 3 * j;
            sc[c][j] = j < tn ? cbase[(size_t)t0 * 3 + e] : INFINITY;       // padding: distance +inf, never below a threshold
        }
        __syncthreads();
#pragma unroll 1
        for (int g0 = 0; g0 < tp; g0 += 32) {
            unsigned mask = 0;                                     // candidate g0 + u -> bit 31 - u
#pragma unroll
            for (int u = 0; u < 32; u += 4) {
                const ksm_f4 X = *(const ksm_f4 *)&sc[0][g0 + u], Y = *(const ksm_f4 *)&sc[1][g0 + u], Z = *(const ksm_f4 *)&sc[2][g0 + u];
                const ksm_f4 dx = X + nqx, dy = Y + nqy, dz = Z + nqz;
                const ksm_f4 dd = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
                for (int e = 0; e < 4; e++) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(dd[e]) - thr, 31);   // sign: d < k-th best
            }
            if (__ballot(mask != 0) != 0) {
#pragma unroll 1
                do {
                    const bool has = mask != 0;
                    const int u = has ? __builtin_clz(mask) : 0;
                    mask &= ~(0x80000000u >> u);
                    const int jl = g0 + u;
                    const float dx = sc[0][jl] - qx, dy = sc[1][jl] - qy, dz = sc[2][jl] - qz;
                    const float d = (dx * dx + dy * dy) + dz * dz;
                    const int j = t0 + jl;
                    // the mask was built against the threshold at the top of the block: test again (strict: an equal distance
                    // at a higher index stays out)
                    const bool c3 = has && d < bd[3], c2 = has && d < bd[2], c1 = has && d < bd[1], c0 = has && d < bd[0];
                    bd[3] = c2 ? bd[2] : (c3 ? d : bd[3]);  bi[3] = c2 ? bi[2] : (c3 ? j : bi[3]);
                    bd[2] = c1 ? bd[1] : (c2 ? d : bd[2]);  bi[2] = c1 ? bi[1] : (c2 ? j : bi[2]);
                    bd[1] = c0 ? bd[0] : (c1 ? d : bd[1]);  bi[1] = c0 ? bi[0] : (c1 ? j : bi[1]);
                    bd[0] = c0 ? d : bd[0];                 bi[0] = c0 ? j : bi[0];
                } while (__ballot(mask != 0) != 0);
                const float kth = k == 1 ? bd[0] : (k == 2 ? bd[1] : (k == 3 ? bd[2] : bd[3]));
                thr = __float_as_uint(kth);
            }
        }
    }
    if (!valid) return;
    const size_t o = ((size_t)b * Nq + q) * k;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (s < k) {
            if (out_mode == KSM_OUT_PAIR) { ((int32_t *)idx_out)[o + s] = bi[s]; val_out[o + s] = bd[s]; }
            else { ((int64_t *)idx_out)[o + s] = bi[s]; val_out[o + s] = sqrtf(bd[s]); }
        }
    }
}

bool l3d_knn_small_supported(int Nc, int k) { return k >= 1 && k <= 4 && Nc >= 1; }
// where it is the faster kernel (tools/knn_select_bench.py small): from 65 536 queries up -- a query per lane needs a thousand waves
// to fill the chip (32 x 8192 queries x 1024 candidates, k = 3: 88 us against 289 / 280; 32 x 1024 queries x 8192: 217 against 108)
bool l3d_knn_small_preferred(long queries, int Nc, int k) { return l3d_knn_small_supported(Nc, k) && queries >= 65536; }

int l3d_launch_knn_small(const float *q, const float *c, int B, int Nq, int Nc, int k, int out_mode, void *idx, float *val, hipStream_t st)
{
    if (!l3d_knn_small_supported(Nc, k)) return L3D_ERR_UNSUPPORTED;
    dim3 grid(l3d_divup(Nq, 256), B), block(256);
    hipLaunchKernelGGL(knn_small_kernel, grid, block, 0, st, q, c, Nq, Nc, k, out_mode, idx, val);
    return l3d_check_launch();
}
