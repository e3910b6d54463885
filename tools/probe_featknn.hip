// Timing probe for featknn.hip: build with -DFK_PROBE=0/1/2 (full / no insertions / GEMM only) and
// optionally -DFK_COUNT (insertion-loop trips per wave).   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include "../learning3d_amd/csrc/featknn.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
thread_local int g_l3d_last_hip_error = 0;
int main(int argc, char **argv)
{
    const int B = argc > 3 ? atoi(argv[3]) : 32, N = argc > 2 ? atoi(argv[2]) : 1024, C = argc > 1 ? atoi(argv[1]) : 64, K = 20;
    float *x; int64_t *idx; void *ws;
    hipMalloc(&x, 4ull * B * C * N); hipMalloc(&idx, 8ull * B * N * K);
    hipMalloc(&ws, l3d_knn_feature_workspace_bytes(B, C, N));
    std::vector<float> h((size_t)B * C * N);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; float u1 = ((s >> 8) + 1) / 16777217.f; s = s * 1664525u + 1013904223u; float u2 = (s >> 8) / 16777216.f; v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
    hipMemcpy(x, h.data(), 4 * h.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 20; it++) l3d_knn_feature(x, B, C, N, K, ws, idx, 0);
    hipDeviceSynchronize();
#ifdef FK_COUNT
    unsigned long long zero = 0; hipMemcpyToSymbol(HIP_SYMBOL(fk_trip_counter), &zero, 8);
    l3d_knn_feature(x, B, C, N, K, ws, idx, 0); hipDeviceSynchronize();
    unsigned long long tr; hipMemcpyFromSymbol(&tr, HIP_SYMBOL(fk_trip_counter), 8);
    printf("trips per wave: %.1f\n", (double)tr / (B * ((N + 127) / 128) * 4));
#endif
    const int R = 50;
    hipEventRecord(e0, 0);
    for (int it = 0; it < R; it++) l3d_knn_feature(x, B, C, N, K, ws, idx, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("FK_PROBE=%d C=%d N=%d: %.1f us per call (split + main)\n", FK_PROBE, C, N, ms * 1000 / R);
    return 0;
}
