// attention_f16_kernel at DCP's shape (B=32, H=4, D=128, N=M=1024) stand-alone: microseconds per launch, for ablation builds
// (-DAF_NOLOAD: no global loads; -DAF_NOSPLIT: operand splits replaced by moves) that show where its time goes.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-DAF_NOLOAD] [-DAF_NOSPLIT] [-DAF_NOEXP] tools/probe_attention_f16.hip -o tools/bin/probe_att
#include "../learning3d_amd/csrc/attention_f16.hip"
#include <cstdio>
#include <cstring>
#include <vector>
thread_local int g_l3d_last_hip_error = 0;
int main()
{
    const int B = 32, H = 4, D = 128, N = 1024;
    const size_t n = (size_t)B * H * D * N;
    std::vector<float> hq(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; hq[i] = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
    float *q, *k, *v, *ctx; unsigned *ws; void *img;
    hipMalloc(&q, n * 4); hipMalloc(&k, n * 4); hipMalloc(&v, n * 4); hipMalloc(&ctx, n * 4); hipMalloc(&ws, 16);
    hipMalloc(&img, (size_t)(H * D / 8) * B * N * 16 * 2 + 16);
    hipMemcpy(q, hq.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(k, hq.data() + 7, (n - 7) * 4, hipMemcpyHostToDevice);
    hipMemcpy(v, hq.data() + 13, (n - 13) * 4, hipMemcpyHostToDevice);
    const float one = 1.0f; unsigned mx[4];
    memcpy(&mx[0], &one, 4); mx[1] = mx[2] = mx[0]; mx[3] = 0;
    hipMemcpy(ws, mx, 16, hipMemcpyHostToDevice);
    const long bs = (long)H * D * N;
    for (int mode = 0; mode < 2; mode++) {           // 0: fp32 context, 1: plane image (DCP's route)
        for (int it = 0; it < 3; it++)
            l3d_attention_forward_f16_maxima(q, k, v, B, H, D, N, N, bs, bs, bs, 0.0884f, ws, mode ? nullptr : ctx, mode ? img : nullptr, nullptr);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0, nullptr);
            for (int it = 0; it < 10; it++)
                l3d_attention_forward_f16_maxima(q, k, v, B, H, D, N, N, bs, bs, bs, 0.0884f, ws, mode ? nullptr : ctx, mode ? img : nullptr, nullptr);
            hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%s: %.1f us per launch (%.0f TFLOP/s fp32-equivalent)\n", mode ? "planes out" : "fp32 out  ", best * 100.f,
               4.0 * B * H * (double)N * N * D / (best * 100e-6) / 1e12);
    }
    return 0;
}
