// conv_f16b_kernel against conv_f16_kernel<false,false,false,2> at conv5's shape: same bits?  time per launch, and the
// workgroup timeline (s_memrealtime marks; see probe_conv_timeline.hip).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_conv_f16b.hip -o tools/bin/probe_conv_f16b
#define CB_TIMELINE
#include "../learning3d_amd/csrc/conv_f16.hip"
#include "experiments/conv_f16b.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
thread_local int g_l3d_last_hip_error = 0;
int main()
{
    const int B = 32, N = 1024, Cin = 512, Cout = 1024;
    const size_t xb = l3d_f16_act_bytes((long)B * N, Cin), wb = l3d_conv_f16_weight_bytes(Cout, Cin);
    const size_t xpb = l3d_f16_plane_bytes((long)B * N, Cin), wpb = l3d_f16_plane_bytes(Cout, Cin);
    void *x, *w; float *y0, *y1, *sc, *sh; long long *tl;
    const size_t ny = (size_t)B * Cout * N;
    hipMalloc(&x, xb); hipMalloc(&w, wb); hipMalloc(&y0, ny * 4); hipMalloc(&y1, ny * 4); hipMalloc(&tl, 512 * 64);
    hipMalloc(&sc, Cout * 4); hipMalloc(&sh, Cout * 4);
    {
        std::vector<_Float16> hx(xb / 2), hw(wb / 2);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 32768.f - 1.f; };
        for (auto &v : hx) v = (_Float16)(rnd() * 64.f);
        for (auto &v : hw) v = (_Float16)(rnd() * 4.f);
        float one = 1.f;
        memcpy((char *)hx.data() + 2 * xpb, &one, 4);
        for (int i = 0; i < 3; i++) memcpy((char *)hw.data() + 3 * wpb + 4 * i, &one, 4);
        hipMemcpy(x, hx.data(), xb, hipMemcpyHostToDevice);
        hipMemcpy(w, hw.data(), wb, hipMemcpyHostToDevice);
        std::vector<float> hs(Cout), hh(Cout);
        for (int i = 0; i < Cout; i++) { hs[i] = 0.5f + 0.001f * i; hh[i] = -3.f + 0.01f * i; }
        hipMemcpy(sc, hs.data(), Cout * 4, hipMemcpyHostToDevice);
        hipMemcpy(sh, hh.data(), Cout * 4, hipMemcpyHostToDevice);
    }
    hipMemcpyToSymbol(HIP_SYMBOL(g_cb_timeline), &tl, sizeof(tl));
    hipMemset(y0, 0xff, ny * 4); hipMemset(y1, 0xee, ny * 4);
    int rc0 = l3d_pointwise_conv_f16(x, w, sc, sh, 0, B, Cin, Cout, N, 1, 1, y0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
    int rc1 = l3d_pointwise_conv_f16b(x, w, sc, sh, 0, B, Cin, Cout, N, 1, y1, nullptr);
    hipDeviceSynchronize();
    printf("rc %d %d, hip error %s\n", rc0, rc1, hipGetErrorString(hipGetLastError()));
    {
        std::vector<float> a(ny), c(ny);
        hipMemcpy(a.data(), y0, ny * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), y1, ny * 4, hipMemcpyDeviceToHost);
        size_t bad = 0, nz = 0; double big = 0;
        for (size_t i = 0; i < ny; i++) {
            if (memcmp(&a[i], &c[i], 4)) { if (bad < 5) printf("  diff at %zu: %g vs %g\n", i, a[i], c[i]); bad++; }
            nz += a[i] != 0.f; big = std::max(big, (double)fabsf(a[i]));
        }
        printf("values that differ: %zu of %zu (nonzero %zu, max |y| %.3g)\n", bad, ny, nz, big);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 3; rep++) {
            for (int it = 0; it < 20; it++) which ? l3d_pointwise_conv_f16b(x, w, sc, sh, 0, B, Cin, Cout, N, 1, y1, nullptr)
                                                  : l3d_pointwise_conv_f16(x, w, sc, sh, 0, B, Cin, Cout, N, 1, 1, y0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
            hipEventRecord(e0, nullptr);
            for (int it = 0; it < 50; it++) which ? l3d_pointwise_conv_f16b(x, w, sc, sh, 0, B, Cin, Cout, N, 1, y1, nullptr)
                                                  : l3d_pointwise_conv_f16(x, w, sc, sh, 0, B, Cin, Cout, N, 1, 1, y0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us per launch\n", which ? "conv_f16b_kernel           " : "conv_f16_kernel<..,2> (old)", ms * 1000 / 50);
        }
    }
    const int nwg = 512;
    std::vector<long long> t((size_t)nwg * 8);
    hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = t[0];
    for (int g = 0; g < nwg; g++) t0 = std::min(t0, t[(size_t)g * 8]);
    auto us = [&](long long v) { return (double)(v - t0) / 100.0; };
    std::map<unsigned, std::vector<int>> bycu;
    for (int g = 0; g < nwg; g++) {
        const unsigned id = (unsigned)t[(size_t)g * 8 + 7];
        bycu[((id >> 8) & 0xff) | ((unsigned)(g & 7) << 16)].push_back(g);
    }
    printf("distinct (xcd, se, sh, cu): %zu\n", bycu.size());
    double sum[4][6] = {{0}}; int cnt[4] = {0, 0, 0, 0};
    for (auto &kv : bycu) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end(), [&](int a, int b) { return t[(size_t)a * 8] < t[(size_t)b * 8]; });
        for (size_t r = 0; r < v.size() && r < 4; r++) {
            for (int i = 0; i < 6; i++) sum[r][i] += us(t[(size_t)v[r] * 8 + i]);
            cnt[r]++;
        }
    }
    const char *names[6] = {"entry", "chunk 0 landed", "main loop done", "stores issued (t0)", "t0 stores acked", "all waves acked"};
    for (int r = 0; r < 4; r++) {
        if (!cnt[r]) continue;
        printf("round %d (%d workgroups), mean us since the first entry:\n", r, cnt[r]);
        for (int i = 0; i < 6; i++) printf("   %-20s %8.2f\n", names[i], sum[r][i] / cnt[r]);
    }
    return 0;
}
