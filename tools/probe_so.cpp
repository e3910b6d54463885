// Times the kernels of the SHIPPED libl3d_hip.so (dlopen) with plain HIP events, outside Python.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef int (*chained_t)(const float *, const int64_t *, int, int, int, const float *, float *, void *);
typedef int (*lds_t)(const float *, const int64_t *, int, int, int, const float *, int, int, int, int, float *, void *);
int main(int argc, char **argv)
{
    void *h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen failed %s\n", dlerror()); return 1; }
    chained_t chained = (chained_t)dlsym(h, "l3d_edgeconv_forward_chained");
    lds_t lds = (lds_t)dlsym(h, "l3d_edgeconv_forward");
    const int B = 32, N = 1024, K = 20, NP = 46080 + 45568 + 67584;
    float *xyz, *packed, *pooled; int64_t *idx;
    hipMalloc(&xyz, 4 * B * N * 3); hipMalloc(&idx, 8 * B * N * K); hipMalloc(&packed, 4 * NP); hipMalloc(&pooled, 4 * (size_t)B * N * 512);
    std::vector<float> hv(B * N * 3 + NP);
    for (size_t i = 0; i < hv.size(); i++) hv[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(xyz, hv.data(), 4 * B * N * 3, hipMemcpyHostToDevice);
    hipMemcpy(packed, hv.data(), 4 * NP, hipMemcpyHostToDevice);
    std::vector<int64_t> hi((size_t)B * N * K);
    for (size_t i = 0; i < hi.size(); i++) hi[i] = (i * 40503u) % N;
    hipMemcpy(idx, hi.data(), 8 * hi.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        auto run = [&]() { if (which) chained(xyz, idx, B, N, K, packed, pooled, nullptr); else lds(xyz, idx, B, N, K, packed, 64, 64, 128, 256, pooled, nullptr); };
        for (int i = 0; i < 5; i++) run();
        hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 50; i++) run(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us\n", which ? "chained(.so)" : "lds(.so)", ms / 50 * 1e3);
    }
    return 0;
}
