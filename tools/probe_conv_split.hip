// Ablation timing of conv_split_kernel at the conv5 shape: build with -DCS_PROBE=<bits> (see
// conv_split.hip) and compare.  Not a product path.
#include "../learning3d_amd/csrc/conv_split.hip"
#include <cstdio>
#include <vector>
thread_local int g_l3d_last_hip_error = 0;
int main()
{
    const int B = 32, N = 1024, Cin = 512, Cout = 1024;
    float *x, *w, *y; void *ws, *xs;
    hipMalloc(&x, 4ul * B * N * Cin); hipMalloc(&w, 4ul * Cout * Cin); hipMalloc(&y, 4ul * B * N * Cout);
    hipMalloc(&ws, l3d_split_bytes(Cout, Cin)); hipMalloc(&xs, l3d_split_bytes(B * N, Cin));
    std::vector<float> h((size_t)B * N * Cin);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    hipMemcpy(x, h.data(), 4 * h.size(), hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), 4ul * Cout * Cin, hipMemcpyHostToDevice);
    l3d_split_rows(w, Cout, Cin, ws, nullptr);
    l3d_split_rows(x, B * N, Cin, xs, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 1; mode <= 2; mode++) {
        auto run = [&]() { l3d_pointwise_conv_split(mode == 2 ? xs : (void *)x, mode, ws, nullptr, nullptr, 0, B, Cin, Cout, N, 1, y, nullptr); };
        for (int i = 0; i < 3; i++) run();
        hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 20; i++) run(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("probe=%d mode=%d: %.1f us\n", CS_PROBE, mode, ms / 20 * 1e3);
    }
    return 0;
}
