// Stage timing of knn_mfma_kernel (s_memtime per wave): staging | barrier | pass 0 | thr0 | pass 1 | barrier | rank
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_knn_mfma.hip -o tools/bin/probe_knn_mfma
#define KM_TIMING
#include "../learning3d_amd/csrc/knn_mfma.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
thread_local int g_l3d_last_hip_error = 0;
int main()
{
    const int B = 32, N = 1024, k = 20;
    std::vector<float> h((size_t)B * N * 3);
    srand(1);
    for (auto &v : h) v = rand() / (float)RAND_MAX;
    float *x; long long *out;
    hipMalloc(&x, h.size() * 4); hipMalloc(&out, (size_t)B * N * k * 8);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int dbg = 0; dbg < 2; dbg++) {
    printf("---- %s\n", dbg ? "no candidate reaches the threshold (pass 1 = tiles + masks only, empty lists)" : "normal");
    for (int it = 0; it < 3; it++) l3d_launch_knn_mfma(x, B, N, k + 256 * dbg, (int64_t *)out, 0);
    hipDeviceSynchronize();
    const int nw = B * (N / 32) * 4;
    std::vector<long long> t((size_t)nw * 16);
    hipMemcpy(t.data(), out, t.size() * 8, hipMemcpyDeviceToHost);
    const char *names[6] = {"stage barrier", "pass 0 tiles", "top3 + thr0", "pass 1", "barrier", "rank"};
    double tot = 0;
    { double a = 0; for (int w = 0; w < nw; w++) a += (double)(t[(size_t)w * 16] - t[(size_t)w * 16 + 7]); printf("%-16s %8.0f\n", "staging", a / nw); tot += a / nw; }
    for (int s = 0; s < 6; s++) {
        double a = 0;
        for (int w = 0; w < nw; w++) a += (double)(t[(size_t)w * 16 + s + 1] - t[(size_t)w * 16 + s]);
        a /= nw;
        tot += a;
        printf("%-16s %8.0f cycles (100 MHz ticks x 24 if s_memtime is the 100 MHz counter)\n", names[s], a);
    }
    long long lo = t[7], hi = t[6];
    for (int w = 0; w < nw; w++) { if (t[(size_t)w * 16 + 7] < lo) lo = t[(size_t)w * 16 + 7]; if (t[(size_t)w * 16 + 6] > hi) hi = t[(size_t)w * 16 + 6]; }
    printf("sum %8.0f per wave; first start -> last end %lld\n", tot, hi - lo);
  }
    return 0;
}
