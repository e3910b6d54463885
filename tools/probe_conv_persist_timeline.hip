// Timeline of the persistent conv5 kernel's workgroups (B = 32, N = 1024, 512 -> 1024; 256 workgroups, two tiles each):
// s_memrealtime marks (100 MHz) per workgroup:  0 entry | 1 chunks 0, 1 of the first tile landed | per tile i: 2+3i main loop done,
// 3+3i next tile's first chunks landed (barrier), 4+3i epilogue issued (thread 0) | 8 thread 0's stores acknowledged
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_conv_persist_timeline.hip -o tools/bin/probe_conv_persist_timeline
#define CF_TIMELINE
#define CF_PERSIST
#include "../learning3d_amd/csrc/conv_f16.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
thread_local int g_l3d_last_hip_error = 0;
int main()
{
    const int B = 32, N = 1024, Cin = 512, Cout = 1024;
    const size_t xb = l3d_f16_act_bytes((long)B * N, Cin), wb = l3d_conv_f16_weight_bytes(Cout, Cin);
    void *x, *w; float *y; long long *tl;
    const int nwg = 256;
    hipMalloc(&x, xb); hipMalloc(&w, wb); hipMalloc(&y, (size_t)B * Cout * N * 4); hipMalloc(&tl, (size_t)nwg * 128);
    hipMemset(x, 0x11, xb); hipMemset(w, 0x11, wb);
    hipMemcpyToSymbol(HIP_SYMBOL(g_cf_timeline), &tl, sizeof(tl));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&]() { return l3d_pointwise_conv_f16(x, w, nullptr, nullptr, 0, B, Cin, Cout, N, 1, 1, y, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr); };
    for (int it = 0; it < 20; it++) if (run()) { printf("launch failed\n"); return 1; }
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < 20; it++) run();
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("kernel (with marks): %.1f us per launch\n", ms * 1000 / 20);
    std::vector<long long> t((size_t)nwg * 16);
    hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = t[0];
    for (int g = 0; g < nwg; g++) t0 = std::min(t0, t[(size_t)g * 16]);
    const char *names[9] = {"entry", "first chunks landed", "tile 0 main loop done", "tile 1 chunks landed", "tile 0 epilogue issued",
                            "tile 1 main loop done", "(no next tile)", "tile 1 epilogue issued", "stores acknowledged"};
    for (int i = 0; i < 9; i++) {
        double s = 0, mn = 1e30, mx = 0;
        for (int g = 0; g < nwg; g++) { const double v = (double)(t[(size_t)g * 16 + i] - t0) / 100.0; s += v; mn = std::min(mn, v); mx = std::max(mx, v); }
        printf("   %-26s mean %8.2f us   min %8.2f   max %8.2f\n", names[i], s / nwg, mn, mx);
    }
    return 0;
}
