// Stand-alone timing probe for the two MFMA kernels of mlp.hip (not part of the product).
// Build variants with -DPW_TK=32, -DPW_PROBE_NO_STORE, -DPW_PROBE_NO_GLOBAL ... and compare.
#include "../learning3d_amd/csrc/mlp.hip"
#include "../learning3d_amd/csrc/edgeconv2.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
thread_local int g_l3d_last_hip_error = 0;
int main(int argc, char **argv)
{
    const int B = 32, N = 1024, CIN = 512, COUT = 1024, K = 20;
    float *x, *w, *y, *sc, *sh, *xyz, *packed, *pooled; int64_t *idx;
    hipMalloc(&x, sizeof(float) * B * N * CIN); hipMalloc(&w, sizeof(float) * COUT * CIN);
    hipMalloc(&y, sizeof(float) * (size_t)B * COUT * N); hipMalloc(&sc, 4 * COUT); hipMalloc(&sh, 4 * COUT);
    hipMalloc(&xyz, 4 * B * N * 3); hipMalloc(&idx, 8 * B * N * K); hipMalloc(&packed, 4 * EC_PACKED_FLOATS);
    hipMalloc(&pooled, 4 * (size_t)B * N * 512);
    std::vector<float> h((size_t)B * N * CIN);
    const bool zero = getenv("ZERO") != nullptr;
    for (size_t i = 0; i < h.size(); i++) h[i] = zero ? 0.f : (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(x, h.data(), 4 * h.size(), hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), 4 * COUT * CIN, hipMemcpyHostToDevice);
    hipMemcpy(sc, h.data(), 4 * COUT, hipMemcpyHostToDevice); hipMemcpy(sh, h.data(), 4 * COUT, hipMemcpyHostToDevice);
    hipMemcpy(xyz, h.data(), 4 * B * N * 3, hipMemcpyHostToDevice);
    hipMemcpy(packed, h.data(), 4 * EC_PACKED_FLOATS, hipMemcpyHostToDevice);
    std::vector<int64_t> hi((size_t)B * N * K);
    for (size_t i = 0; i < hi.size(); i++) hi[i] = (i * 40503u) % N;
    hipMemcpy(idx, hi.data(), 8 * hi.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 3; which++) {
        auto run = [&]() {
            if (which == 0) l3d_pointwise_conv(x, 1, w, sc, sh, 0, B, CIN, COUT, N, 1, y, nullptr);
            else if (which == 1) l3d_edgeconv_forward(xyz, idx, B, N, K, packed, 64, 64, 128, 256, pooled, nullptr);
            else l3d_edgeconv_forward_chained(xyz, idx, B, N, K, packed, pooled, nullptr);
        };
        for (int i = 0; i < 5; i++) run();
        hipDeviceSynchronize();
        const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 20;
        hipEventRecord(e0);
        for (int i = 0; i < reps; i++) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        const double flop = which == 0 ? 2.0 * B * N * CIN * COUT : 2.0 * B * N * K * 45440.0;
        printf("%s %s: %.1f us  %.1f TFLOP/s\n", argv[0], which == 0 ? "conv5" : which == 1 ? "edgeconv" : "edgeconv_chained", ms * 1e3, flop / ms / 1e9);
    }
    return 0;
}
