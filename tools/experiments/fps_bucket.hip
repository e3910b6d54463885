// fps_bucket.hip -- an experiment, not part of the library (LABLOG R3.9): farthest point sampling that visits only the
// spatial buckets a new sample can change.  Bit-identical to fps_kernel (grouping.hip) on every cloud tried, and not faster
// at 32 clouds per GPU: a round of fps_kernel is ~0.42 us of reduce / exchange / barrier latency plus ~0.49 us of VALU work on
// 8192 points, and the bookkeeping of the buckets costs the busy wave what the skipped updates save.
// Included by tools/probe_fps_bucket.hip after learning3d_amd/csrc/grouping.hip (it uses that file's wave reductions).
// ---------------------------------------------------------------------------------------------
// Farthest point sampling with spatial buckets (4096 <= n <= 8192: FlowNet3D's first set-abstraction layer,
// models/flownet3d.py:293 -- 72 % of config 5's layer time with the kernel above, whose every round updates the
// running minimum of ALL n points although a new sample can only lower it inside its own neighbourhood).
//
// The cloud is sorted once along a Morton curve (keys in LDS, bitonic) and cut into buckets of 64 consecutive points,
// one (wave, register slot) each; a wave owns 16 consecutive buckets, i.e. one region of space, and lane u of the wave
// holds slot u's bounding box.  A round then
//   * tests the new sample against the wave's 16 boxes in one VALU pass (lane u: slot u): if the squared distance to the
//     box, lowered by 2e-6 relative (the fp32 evaluation of a point's distance is within 7e-7 of the exact one, which is
//     >= the exact box distance), is not below G = the largest running minimum of the WHOLE cloud (the value the previous
//     round's arg-max returned; running minima only fall), no point of the bucket can change -- fminf(d, dmin) would
//     return dmin for every one of them -- and the slot is skipped.  sqrt(G) is the covering radius of the samples so
//     far: after a few dozen samples it is a fraction of the cloud and most waves skip all 16 slots;
//   * a wave with no surviving slot re-sends its arg-max of the previous round; the others update their surviving slots
//     and take their arg-max again (values by a max3 tree, the index only in the lanes that hold the maximum);
//   * the waves' candidates are merged through LDS as in the kernel above.
// Skipped updates are no-ops, ties resolve to the lowest ORIGINAL index at every level, and the distance is
// evaluated with the same expression: the samples (and `temp`) are the bits the kernel above produces.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_min_f(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

template <int PPT, bool OUT64>
__global__ __launch_bounds__(512) void fps_bucket_kernel(int n, int m, const float *__restrict__ xyz,
                                                         const int64_t *__restrict__ start,
                                                         float *__restrict__ temp, void *__restrict__ out, int idx_bits)
{
    constexpr int W = 8, NTHR = 512, P = NTHR * PPT;
    static_assert(PPT <= 16, "lane u of a wave holds slot u's bucket record");
    extern __shared__ float sxyz[];              // [3][n]; before that the P sort keys
    uint32_t *keys = (uint32_t *)sxyz;
    __shared__ int wv[2][16];
    __shared__ int wi[2][16];
    __shared__ float red[6][W];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float *p = xyz + (size_t)b * n * 3;

    // ---- cloud bounding box -> Morton keys (code << idx_bits | index) -> sorted
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = tid; k < n; k += NTHR)
#pragma unroll
            for (int a = 0; a < 3; a++) { const float v = p[k * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[a] = wave_min_f(lo[a]); hi[a] = wave_max_f(hi[a]); }
        if (lane == 0)
#pragma unroll
            for (int a = 0; a < 3; a++) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
        __syncthreads();
        const int mbits = (32 - idx_bits) / 3;
        float sc[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo[a] = red[a][0]; hi[a] = red[3 + a][0];
            for (int w = 1; w < W; w++) { lo[a] = fminf(lo[a], red[a][w]); hi[a] = fmaxf(hi[a], red[3 + a][w]); }
            const float ext = hi[a] - lo[a];
            sc[a] = ext > 0.f && ext < INFINITY ? (float)((1 << mbits) - 1) / ext : 0.f;
        }
        for (int k = tid; k < P; k += NTHR) {
            uint32_t key = 0xffffffffu;
            if (k < n) {
                uint32_t q[3], code = 0;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const float f = (p[k * 3 + a] - lo[a]) * sc[a];
                    q[a] = (uint32_t)min(max((int)f, 0), (1 << mbits) - 1);        // a NaN coordinate lands in cell 0
                }
                for (int bit = mbits - 1; bit >= 0; bit--)
                    code = (code << 3) | (((q[0] >> bit) & 1u) << 2) | (((q[1] >> bit) & 1u) << 1) | ((q[2] >> bit) & 1u);
                key = (code << idx_bits) | (uint32_t)k;
            }
            keys[k] = key;
        }
        __syncthreads();
        for (int kk = 2; kk <= P; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (P >> 1); t += NTHR) {
                    const int l = ((t & ~(j - 1)) << 1) | (t & (j - 1)), h = l | j;
                    const uint32_t x = keys[l], y = keys[h];
                    if ((x > y) == ((l & kk) == 0)) { keys[l] = y; keys[h] = x; }
                }
                __syncthreads();
            }
    }
    // slot u of this thread: sorted position ((wave PPT + u) 64 + lane) -- a wave owns PPT consecutive buckets, one spatial region
    int oi[PPT];
    const uint32_t imask = (1u << idx_bits) - 1u;
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const int pos = (wave * PPT + u) * 64 + lane;
        oi[u] = pos < n ? (int)(keys[pos] & imask) : 0x7fffffff;        // sentinels sort behind the n real keys
    }
    __syncthreads();
    for (int k = tid; k < n; k += NTHR) { sxyz[k] = p[k * 3]; sxyz[n + k] = p[k * 3 + 1]; sxyz[2 * n + k] = p[k * 3 + 2]; }
    if (tid < 32) { wv[tid >> 4][tid & 15] = (int)0x80000000; wi[tid >> 4][tid & 15] = 0x7fffffff; }
    __syncthreads();
    float px[PPT], py[PPT], pz[PPT];
    int dmin[PPT];                                                       // bit patterns: non-negative floats order like their bits; -1.f for padding
    float blx = INFINITY, bly = INFINITY, blz = INFINITY, bhx = -INFINITY, bhy = -INFINITY, bhz = -INFINITY;   // lane u: slot u's box
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const bool ok = oi[u] != 0x7fffffff;
        px[u] = ok ? sxyz[oi[u]] : 0.f;
        py[u] = ok ? sxyz[n + oi[u]] : 0.f;
        pz[u] = ok ? sxyz[2 * n + oi[u]] : 0.f;
        dmin[u] = __builtin_bit_cast(int, ok ? 1e10f : -1.f);            // out-of-range slots can never win the arg-max
        const float lx = wave_min_f(ok ? px[u] : INFINITY), ly = wave_min_f(ok ? py[u] : INFINITY), lz = wave_min_f(ok ? pz[u] : INFINITY);
        const float hx = wave_max_f(ok ? px[u] : -INFINITY), hy = wave_max_f(ok ? py[u] : -INFINITY), hz = wave_max_f(ok ? pz[u] : -INFINITY);
        if (lane == u) { blx = lx; bly = ly; blz = lz; bhx = hx; bhy = hy; bhz = hz; }
    }
    int old = (OUT64 && start) ? (int)start[b] : 0;
    if (tid == 0) {
        if (OUT64) ((int64_t *)out)[(size_t)b * m] = old; else ((int32_t *)out)[(size_t)b * m] = old;
    }
    int wmax = (int)0x80000000, widx = 0x7fffffff;                       // this wave's arg-max, kept across rounds that leave it alone
    float G = 1e10f;                                                     // the cloud's largest running minimum before this round
    for (int j = 1; j < m; j++) {
        const float x1 = sxyz[old], y1 = sxyz[n + old], z1 = sxyz[2 * n + old];          // wave-uniform
        // which of the wave's buckets can change (lane u: slot u's box): a bucket whose box is farther from the sample than
        // EVERY point's running minimum (G) keeps all of its minima
        unsigned act;
        {
            const float ex = fmaxf(fmaxf(blx - x1, x1 - bhx), 0.f), ey = fmaxf(fmaxf(bly - y1, y1 - bhy), 0.f),
                        ez = fmaxf(fmaxf(blz - z1, z1 - bhz), 0.f);
            const float bound = ((ex * ex + ey * ey) + ez * ez) * (1.0f - 2e-6f);
            const bool may = j == 1 || bound < G + 1e-37f;               // an empty bucket: NaN < G is false
            act = (unsigned)(__ballot(may && lane < PPT) & 0xffffull);
        }
        if (act) {                                                       // wave-uniform
#pragma unroll
            for (int u = 0; u < PPT; u++) {
                if (act & (1u << u)) {
                    const float dx = px[u] - x1, dy = py[u] - y1, dz = pz[u] - z1;
                    const float d = (dx * dx + dy * dy) + dz * dz;
                    dmin[u] = __builtin_bit_cast(int, fminf(d, __builtin_bit_cast(float, dmin[u])));
                }
            }
            // this wave's arg-max again: values first (a max3 tree), the index only in the lane(s) that hold the maximum
            int vm = dmin[0];
#pragma unroll
            for (int u = 1; u + 1 < PPT; u += 2) vm = max(vm, max(dmin[u], dmin[u + 1]));
            vm = max(vm, dmin[PPT - 1]);
            wmax = wave_max_i(vm);
            int cand = 0x7fffffff;
#pragma unroll
            for (int u = 0; u < PPT; u++) cand = min(cand, dmin[u] == wmax ? oi[u] : 0x7fffffff);
            const unsigned long long hit = __ballot(vm == wmax);
            if (__builtin_popcountll(hit) == 1) widx = __builtin_amdgcn_readlane(cand, __builtin_ctzll(hit));
            else widx = wave_min_i(cand);
        }
        const int buf = j & 1;
        if (lane == 0) { wv[buf][wave] = wmax; wi[buf][wave] = widx; }
        __syncthreads();
        const int ev = wv[buf][lane & 15], ei = wi[buf][lane & 15];      // every 16-lane row sees all waves
        const int bmax = row16_max_i(ev);
        const unsigned h16 = (unsigned)(__ballot(ev == bmax) & 0xffffull);
        if (__builtin_popcount(h16) == 1)
            old = __builtin_amdgcn_readlane(ei, __builtin_ctz(h16));
        else
            old = __builtin_amdgcn_readfirstlane(row16_min_i(ev == bmax ? ei : 0x7fffffff));
        G = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(bmax));
        if (tid == 0) {
            if (OUT64) ((int64_t *)out)[(size_t)b * m + j] = old; else ((int32_t *)out)[(size_t)b * m + j] = old;
        }
    }
    if (temp) {
#pragma unroll
        for (int u = 0; u < PPT; u++)
            if (oi[u] != 0x7fffffff) temp[(size_t)b * n + oi[u]] = __builtin_bit_cast(float, dmin[u]);
    }
}

