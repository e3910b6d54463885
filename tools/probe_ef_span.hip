// When do the 256 persistent workgroups of edgeconv_f16b_kernel<5, true> start and end, and on which XCD?  (s_memrealtime, 100 MHz.)
// The tile -> workgroup assignment is static (tile += gridDim.x): if the XCDs of a box do not run at one speed the kernel lasts as long
// as the slowest of them.  Prints the spread of the workgroups' durations per XCD.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -Iinclude -c tools/probe_ef_span.hip -o /tmp/p.o && hipcc --offload-arch=gfx950 /tmp/p.o learning3d_amd/csrc/build/mlp.o -o tools/bin/probe_ef_span
#define EF_WGSPAN
#include "../learning3d_amd/csrc/edgeconv_f16b.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
thread_local int g_l3d_last_hip_error = 0;
extern "C" size_t l3d_edgeconv_packed_floats(int c1, int c2, int c3, int c4);
extern "C" int l3d_edgeconv_pack_mag(const float *const *w, const float *const *scale, const float *const *shift, const float *mag, int c1,
                                     int c2, int c3, int c4, float *dst);
int main()
{
    const int B = 32, N = 1024, K = 20;
    const int cs[4] = {64, 64, 128, 256}, cin[4] = {6, 64, 64, 128};
    std::vector<std::vector<float>> w(4), sc(4), sh(4);
    unsigned s = 99;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f - 0.5f; };
    const float *wp[4], *scp[4], *shp[4];
    float mag[4] = {4.f, 4.f, 4.f, 4.f};
    for (int l = 0; l < 4; l++) {
        w[l].resize((size_t)cs[l] * cin[l]); sc[l].assign(cs[l], 1.f); sh[l].assign(cs[l], 0.f);
        for (auto &v : w[l]) v = rnd() * 2.f / sqrtf((float)cin[l]);
        wp[l] = w[l].data(); scp[l] = sc[l].data(); shp[l] = sh[l].data();
    }
    const size_t nfl = l3d_edgeconv_packed_floats(64, 64, 128, 256);
    std::vector<float> hp(nfl);
    l3d_edgeconv_pack_mag(wp, scp, shp, mag, 64, 64, 128, 256, hp.data());
    float *xyz, *packed; void *img; int64_t *idx; long long *span; int *flag;
    hipMalloc(&xyz, 4 * B * N * 3); hipMalloc(&idx, 8ul * B * N * K); hipMalloc(&packed, 4 * nfl); hipMalloc(&img, 2ul * B * N * 512 * 2 + 64);
    hipMalloc(&span, 256 * 32); hipMalloc(&flag, 4); hipMemset(flag, 0, 4);
    hipMemcpy(packed, hp.data(), 4 * nfl, hipMemcpyHostToDevice);
    std::vector<float> hv((size_t)B * N * 3);
    for (auto &v : hv) v = rnd() + 0.5f;
    hipMemcpy(xyz, hv.data(), 4 * hv.size(), hipMemcpyHostToDevice);
    std::vector<int64_t> hi((size_t)B * N * K);
    for (size_t i = 0; i < hi.size(); i++) { s = s * 1664525u + 1013904223u; hi[i] = (s >> 8) % N; }
    hipMemcpy(idx, hi.data(), 8 * hi.size(), hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_ef_span), &span, sizeof(span));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 30; i++) l3d_edgeconv_forward_f16b(xyz, idx, B, N, K, packed, img, 2, flag, nullptr);
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < 50; i++) l3d_edgeconv_forward_f16b(xyz, idx, B, N, K, packed, img, 2, flag, nullptr);
        hipEventRecord(e1, nullptr); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> t(256 * 4);
        hipMemcpy(t.data(), span, t.size() * 8, hipMemcpyDeviceToHost);
        long long t0 = t[0], t1 = 0;
        for (int g = 0; g < 256; g++) { t0 = std::min(t0, t[g * 4]); t1 = std::max(t1, t[g * 4 + 1]); }
        printf("rep %d: %.1f us per launch; last launch: first start -> last end %.2f us (hip: %s)\n", rep, ms * 1000 / 50, (t1 - t0) / 100.0,
               hipGetErrorString(hipGetLastError()));
        for (int x = 0; x < 8; x++) {
            double dmin = 1e9, dmax = 0, dsum = 0, emax = 0; int n = 0;
            for (int g = 0; g < 256; g++)
                if ((int)(t[g * 4 + 2] & 0xf) == x) {
                    const double d = (t[g * 4 + 1] - t[g * 4]) / 100.0;
                    dmin = std::min(dmin, d); dmax = std::max(dmax, d); dsum += d; n++;
                    emax = std::max(emax, (t[g * 4 + 1] - t0) / 100.0);
                }
            if (n) printf("   XCC %d: %3d workgroups, duration min %.2f mean %.2f max %.2f us, last end at %.2f us\n", x, n, dmin, dsum / n, dmax, emax);
        }
    }
    return 0;
}
