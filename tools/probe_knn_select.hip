// Where knn_select_kernel's time goes: shader-clock cycles per phase (distances / T0 bisection / compaction / exact path /
// ranking + output), summed per wave, and the mean number of survivors.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude tools/probe_knn_select.hip -o tools/bin/probe_knn_select
#define KS_TIMING
#include "../learning3d_amd/csrc/knn_select.hip"
#include <cstdio>
#include <vector>
thread_local int g_l3d_last_hip_error = 0;
int main(int argc, char **argv)
{
    const int B = 32, n = 1024, m = argc > 1 ? atoi(argv[1]) : 8192, k = argc > 2 ? atoi(argv[2]) : 64;
    unsigned s = 7;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    std::vector<float> hc((size_t)B * m * 3), hq((size_t)B * n * 3);
    for (auto &v : hc) v = rnd() * 4 - 2;
    for (auto &v : hq) v = rnd() * 4 - 2;
    float *c, *q, *val; int *idx; long long *tm;
    hipMalloc(&c, 4 * hc.size()); hipMalloc(&q, 4 * hq.size()); hipMalloc(&val, 4ul * B * n * k); hipMalloc(&idx, 4ul * B * n * k);
    const int nw = 4096 * KS_WAVES;
    hipMalloc(&tm, 8ul * nw * 8); hipMemset(tm, 0, 8ul * nw * 8);
    hipMemcpy(c, hc.data(), 4 * hc.size(), hipMemcpyHostToDevice); hipMemcpy(q, hq.data(), 4 * hq.size(), hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_ks_time), &tm, sizeof(tm));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) l3d_launch_knn_select(q, c, B, n, m, k, KS_OUT_PAIR, idx, val, nullptr);
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < 10; i++) l3d_launch_knn_select(q, c, B, n, m, k, KS_OUT_PAIR, idx, val, nullptr);
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> t((size_t)nw * 8);
    hipMemcpy(t.data(), tm, t.size() * 8, hipMemcpyDeviceToHost);
    double acc[8] = {0}; int waves = 0;
    for (int w = 0; w < nw; w++) { if (t[w * 8] == 0) continue; waves++; for (int i = 0; i < 8; i++) acc[i] += t[w * 8 + i]; }
    const double nq = (double)B * n;
    printf("m %d k %d: %.1f us per launch (%s); %d waves; cycles per query: distances %.0f, T0 %.0f, compaction %.0f, exact path %.0f, rank+write %.0f; survivors %.1f, exact-path queries %.4f\n",
           m, k, ms * 100, hipGetErrorString(hipGetLastError()), waves, acc[0] / nq, acc[1] / nq, acc[2] / nq, acc[3] / nq, acc[4] / nq, acc[6] / nq, acc[7] / nq);
    return 0;
}
