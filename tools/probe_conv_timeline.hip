// Timeline of conv5's workgroups (two-plane instantiation, B = 32, N = 1024, 512 -> 1024): s_memrealtime marks per workgroup
//   0 entry | 1 first barrier passed (chunk 0 landed) | 2 main loop done | 3 epilogue stores issued (thread 0) |
//   4 thread 0's stores acknowledged (vmcnt 0) | 5 all waves' stores acknowledged | 7 HW_ID
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_conv_timeline.hip -o tools/bin/probe_conv_timeline
#define CF_TIMELINE
#include "../learning3d_amd/csrc/conv_f16.hip"
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
thread_local int g_l3d_last_hip_error = 0;
int main()
{
    const int B = 32, N = 1024, Cin = 512, Cout = 1024;
    const size_t xb = l3d_f16_act_bytes((long)B * N, Cin), wb = l3d_conv_f16_weight_bytes(Cout, Cin);
    void *x, *w; float *y; long long *tl;
    const int nwg = 512;
    hipMalloc(&x, xb); hipMalloc(&w, wb); hipMalloc(&y, (size_t)B * Cout * N * 4); hipMalloc(&tl, (size_t)nwg * 64);
    hipMemset(x, 0x11, xb); hipMemset(w, 0x11, wb);
    hipMemcpyToSymbol(HIP_SYMBOL(g_cf_timeline), &tl, sizeof(tl));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 20; it++) l3d_pointwise_conv_f16(x, w, nullptr, nullptr, 0, B, Cin, Cout, N, 1, 1, y, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < 20; it++) l3d_pointwise_conv_f16(x, w, nullptr, nullptr, 0, B, Cin, Cout, N, 1, 1, y, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("kernel (with marks): %.1f us per launch\n", ms * 1000 / 20);
    std::vector<long long> t((size_t)nwg * 8);
    hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = t[0];
    for (int g = 0; g < nwg; g++) t0 = std::min(t0, t[(size_t)g * 8]);
    auto us = [&](long long v) { return (double)(v - t0) / 100.0; };
    // per-phase averages by round (round = rank of the workgroup's entry time on its CU)
    std::map<unsigned, std::vector<int>> bycu;
    for (int g = 0; g < nwg; g++) {
        const unsigned id = (unsigned)t[(size_t)g * 8 + 7];
        // HW_ID: [7:4]? keep the CU-identifying bits: cu_id [11:8], sh_id [12], se_id [15:13] on gfx9; XCC from blockIdx % 8
        bycu[((id >> 8) & 0xff) | ((unsigned)(g & 7) << 16)].push_back(g);
    }
    printf("distinct (xcd, se, sh, cu): %zu\n", bycu.size());
    double sum[2][6] = {{0}}; int cnt[2] = {0, 0};
    double gap = 0; int ngap = 0;
    for (auto &kv : bycu) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end(), [&](int a, int b) { return t[(size_t)a * 8] < t[(size_t)b * 8]; });
        for (size_t r = 0; r < v.size() && r < 2; r++) {
            for (int i = 0; i < 6; i++) sum[r][i] += us(t[(size_t)v[r] * 8 + i]);
            cnt[r]++;
        }
        if (v.size() >= 2) { gap += us(t[(size_t)v[1] * 8]) - us(t[(size_t)v[0] * 8 + 5]); ngap++; }
    }
    const char *names[6] = {"entry", "chunk 0 landed", "main loop done", "stores issued (t0)", "t0 stores acked", "all waves acked"};
    for (int r = 0; r < 2; r++) {
        printf("round %d (%d workgroups), mean us since the first entry:\n", r, cnt[r]);
        for (int i = 0; i < 6; i++) printf("   %-20s %8.2f\n", names[i], sum[r][i] / std::max(1, cnt[r]));
    }
    printf("mean gap between round 0's last ack and round 1's entry on the same CU: %.2f us (%d CUs)\n", gap / std::max(1, ngap), ngap);
    // a few raw rows
    for (int g = 0; g < 4; g++) {
        printf("wg %d:", g);
        for (int i = 0; i < 6; i++) printf(" %7.2f", us(t[(size_t)g * 8 + i]));
        printf("  hwid %08llx\n", t[(size_t)g * 8 + 7]);
    }
    return 0;
}
