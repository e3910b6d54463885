// Per-layer cycle breakdown (s_memtime) of (edgeconv_f16b_kernel<5, false>) at the BASELINE shape; weights are
// zeros (timing does not depend on values), neighbours pseudo-random.  Not a product path.
#define EF_TIMING
#include "../learning3d_amd/csrc/edgeconv_f16b.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
thread_local int g_l3d_last_hip_error = 0;
int main()
{
    const int B = 32, N = 1024, K = 20;
    float *xyz, *packed, *pooled; int64_t *idx; unsigned long long *tdbg;
    hipMalloc(&xyz, 4 * B * N * 3); hipMalloc(&idx, 8 * B * N * K); hipMalloc(&packed, 4 * EC_PACKED_FLOATS); hipMalloc(&pooled, 4ul * B * N * 512);
    const int nblk = (N / 16) * B;
    hipMalloc(&tdbg, 8ul * nblk * 6);
    hipMemset(packed, 0, 4 * EC_PACKED_FLOATS);
    std::vector<float> hv(B * N * 3);
    for (size_t i = 0; i < hv.size(); i++) hv[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    hipMemcpy(xyz, hv.data(), 4 * hv.size(), hipMemcpyHostToDevice);
    std::vector<int64_t> hi((size_t)B * N * K);
    for (size_t i = 0; i < hi.size(); i++) hi[i] = (i * 40503u) % N;
    hipMemcpy(idx, hi.data(), 8 * hi.size(), hipMemcpyHostToDevice);
    dim3 grid((N / 16) * B), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((edgeconv_f16b_kernel<5, false>), grid, block, 0, 0, xyz, idx, B, N, K, packed, pooled, (int*)nullptr, 4096.0f, tdbg);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((edgeconv_f16b_kernel<5, false>), grid, block, 0, 0, xyz, idx, B, N, K, packed, pooled, (int*)nullptr, 4096.0f, tdbg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("kernel: %.1f us\n", ms / 10 * 1e3);
    std::vector<unsigned long long> t((size_t)nblk * 6);
    hipMemcpy(t.data(), tdbg, 8 * t.size(), hipMemcpyDeviceToHost);
    const char *names[5] = {"setup+gather", "layer1(fp32)+finish", "layer2", "layer3", "layer4"};
    for (int s = 0; s < 5; s++) {
        std::vector<double> d;
        for (int b = 0; b < nblk; b++) d.push_back((double)(t[b * 6 + s + 1] - t[b * 6 + s]));
        std::sort(d.begin(), d.end());
        printf("%-22s median %8.0f  p10 %8.0f  p90 %8.0f  (s_memtime ticks)\n", names[s], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
    }
    std::vector<double> tot, st;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < nblk; b++) { tot.push_back((double)(t[b * 6 + 5] - t[b * 6])); tmin = std::min(tmin, t[b * 6]); tmax = std::max(tmax, t[b * 6 + 5]); }
    std::sort(tot.begin(), tot.end());
    printf("total per WG           median %8.0f ; first start -> last end %llu ticks for %d WGs\n", tot[tot.size() / 2], tmax - tmin, nblk);
    return 0;
}
