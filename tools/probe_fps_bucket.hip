// fps_bucket_kernel (tools/experiments/fps_bucket.hip) against fps_kernel (grouping.hip): same samples and running minima?
// time per launch at config 5's shape.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude tools/probe_fps_bucket.hip -o tools/bin/probe_fps_bucket
#include "../learning3d_amd/csrc/grouping.hip"
#include "experiments/fps_bucket.hip"
#include <cstdio>
#include <cstring>
#include <vector>
thread_local int g_l3d_last_hip_error = 0;

static void launch_bucket(int b, int n, int m, const float *xyz, float *temp, int32_t *idx)
{
    const int ppt = (n + 511) / 512;
    int idx_bits = 12;
    while ((1 << idx_bits) < n) idx_bits++;
    const size_t keys = sizeof(float) * (size_t)(ppt <= 8 ? 4096 : 8192), cloud = sizeof(float) * 3 * (size_t)n;
    const size_t lds = keys > cloud ? keys : cloud;
    if (ppt <= 8) hipLaunchKernelGGL((fps_bucket_kernel<8, false>), dim3(b), dim3(512), lds, 0, n, m, xyz, nullptr, temp, idx, idx_bits);
    else hipLaunchKernelGGL((fps_bucket_kernel<16, false>), dim3(b), dim3(512), lds, 0, n, m, xyz, nullptr, temp, idx, idx_bits);
}

int main()
{
    struct Case { const char *name; int B, n, m, kind; };
    const Case cases[] = {{"uniform", 32, 8192, 1024, 0}, {"shell", 32, 8192, 1024, 1}, {"lattice (exact ties)", 2, 5120, 2000, 2},
                          {"4 x duplicated", 2, 4800, 1500, 3}, {"collapsed", 1, 4500, 40, 4}, {"ragged", 3, 5000, 700, 0},
                          {"uniform, m = n", 2, 4096, 4096, 0}};
    for (const Case &c : cases) {
        std::vector<float> h((size_t)c.B * c.n * 3);
        unsigned s = 777u + c.kind;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
        for (int b = 0; b < c.B; b++)
            for (int k = 0; k < c.n; k++) {
                float *q = &h[((size_t)b * c.n + k) * 3];
                if (c.kind == 0) { q[0] = rnd(); q[1] = rnd(); q[2] = rnd(); }
                else if (c.kind == 1) {
                    float x, y, z, r2;
                    do { x = 2 * rnd() - 1; y = 2 * rnd() - 1; z = 2 * rnd() - 1; r2 = x * x + y * y + z * z; } while (r2 < 1e-3f || r2 > 1.f);
                    const float r = sqrtf(r2); q[0] = x / r; q[1] = y / r; q[2] = z / r;
                } else if (c.kind == 2) { const int p = (k * 2654435761u) % c.n; q[0] = (p % 16) / 8.f; q[1] = ((p / 16) % 16) / 8.f; q[2] = (p / 256) / 8.f; }
                else if (c.kind == 3) { if (k < c.n / 4) { q[0] = rnd(); q[1] = rnd(); q[2] = rnd(); } else memcpy(q, q - (size_t)(c.n / 4) * 3, 12); }
                else { q[0] = q[1] = q[2] = 0.25f; }
            }
        float *x, *t0, *t1; int32_t *i0, *i1;
        hipMalloc(&x, h.size() * 4); hipMalloc(&t0, (size_t)c.B * c.n * 4); hipMalloc(&t1, (size_t)c.B * c.n * 4);
        hipMalloc(&i0, (size_t)c.B * c.m * 4); hipMalloc(&i1, (size_t)c.B * c.m * 4);
        hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemset(i0, 0xff, (size_t)c.B * c.m * 4); hipMemset(i1, 0xee, (size_t)c.B * c.m * 4);
        l3d_furthest_point_sampling(c.B, c.n, c.m, x, t0, i0, nullptr);
        launch_bucket(c.B, c.n, c.m, x, t1, i1);
        hipDeviceSynchronize();
        std::vector<int32_t> a((size_t)c.B * c.m), d((size_t)c.B * c.m);
        std::vector<float> ta((size_t)c.B * c.n), td((size_t)c.B * c.n);
        hipMemcpy(a.data(), i0, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(d.data(), i1, d.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(ta.data(), t0, ta.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(td.data(), t1, td.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0, badt = 0;
        for (size_t k = 0; k < a.size(); k++) bad += a[k] != d[k];
        for (size_t k = 0; k < ta.size(); k++) badt += memcmp(&ta[k], &td[k], 4) != 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms[2];
        for (int which = 0; which < 2; which++) {
            hipEventRecord(e0, nullptr);
            for (int it = 0; it < 3; it++) which ? launch_bucket(c.B, c.n, c.m, x, nullptr, i1) : (void)l3d_furthest_point_sampling(c.B, c.n, c.m, x, nullptr, i0, nullptr);
            hipEventRecord(e1, nullptr); hipDeviceSynchronize();
            hipEventElapsedTime(&ms[which], e0, e1);
        }
        printf("%-22s B %2d n %5d m %4d: samples that differ %zu, minima that differ %zu | all points %7.1f us, buckets %7.1f us (hip: %s)\n",
               c.name, c.B, c.n, c.m, bad, badt, ms[0] * 1000 / 3, ms[1] * 1000 / 3, hipGetErrorString(hipGetLastError()));
        hipFree(x); hipFree(t0); hipFree(t1); hipFree(i0); hipFree(i1);
    }
    return 0;
}
