#define KNN_TIMING
#include "../learning3d_amd/csrc/knn.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
thread_local int g_l3d_last_hip_error = 0;
int main()
{
    const int B = 32, N = 1024, K = 20;
    float *xyz; int64_t *idx; long long *tbuf;
    hipMalloc(&xyz, 4 * B * N * 3); hipMalloc(&idx, 8 * B * N * K); hipMalloc(&tbuf, 8 * 8 * 4 * 512 * 2);
    std::vector<float> h(B * N * 3);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    hipMemcpy(xyz, h.data(), 4 * h.size(), hipMemcpyHostToDevice);
    dim3 grid(N / 64, B);
    for (int it = 0; it < 3; it++)
        hipLaunchKernelGGL((topk2_kernel<20, METRIC_EXPANDED, 4>), grid, dim3(256), 0, 0, xyz, xyz, N, N, K, OUT_KNN_GRAPH, (void *)idx, (float *)tbuf);
    hipDeviceSynchronize();
    std::vector<long long> t(512 * 4 * 8);
    hipMemcpy(t.data(), tbuf, 8 * t.size(), hipMemcpyDeviceToHost);
    const char *names[4] = {"pass1 scan+flush", "value merge + broadcast", "pass2 rescan", "final (wave0) "};
    for (int w = 0; w < 4; w++) {
        printf("wave %d:", w);
        for (int ph = 0; ph < 4; ph++) {
            std::vector<long long> d;
            for (int blk = 0; blk < 512; blk++) { const long long *p = &t[(blk * 4 + w) * 8]; d.push_back(p[ph + 1] - p[ph]); }
            std::sort(d.begin(), d.end());
            printf("  %s med %lld max %lld |", names[ph], d[256], d[511]);
        }
        printf("\n");
    }
    long long mn = 1LL << 62, mx = 0;
    for (int blk = 0; blk < 512; blk++) for (int w = 0; w < 4; w++) { mn = std::min(mn, t[(blk * 4 + w) * 8]); mx = std::max(mx, t[(blk * 4 + w) * 8 + 4]); }
    printf("first start -> last end: %lld ticks (s_memtime, 100 MHz?)\n", mx - mn);
    return 0;
}
