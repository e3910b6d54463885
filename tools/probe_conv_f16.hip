// Where does a wave of conv_f16_kernel spend its chunk time?  s_memtime sums per wave: waiting for its DMA pieces | at the
// barrier | issuing the next DMA | fragment reads + MFMA issue | loop overhead.  (MFMAs execute asynchronously: time the pipe
// is busy shows up as stall at the NEXT instruction that cannot issue.)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_conv_f16.hip -o tools/bin/probe_conv_f16
#define CF_TIMING
#include "../learning3d_amd/csrc/conv_f16.hip"
#include <cstdio>
#include <vector>
thread_local int g_l3d_last_hip_error = 0;
int main()
{
    const int B = 32, N = 1024, Cin = 512, Cout = 1024;
    const size_t xb = l3d_f16_act_bytes((long)B * N, Cin), wb = l3d_conv_f16_weight_bytes(Cout, Cin);
    void *x, *w; float *y;
    hipMalloc(&x, xb); hipMalloc(&w, wb); hipMalloc(&y, (size_t)B * Cout * N * 4);
    hipMemset(x, 0x11, xb); hipMemset(w, 0x11, wb);
    for (int it = 0; it < 3; it++) l3d_pointwise_conv_f16(x, w, nullptr, nullptr, 0, B, Cin, Cout, N, 1, 0, y, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
    hipDeviceSynchronize();
    const int nw = 512 * 8;
    std::vector<long long> t((size_t)nw * 8);
    hipMemcpy(t.data(), y, t.size() * 8, hipMemcpyDeviceToHost);
    const char *names[5] = {"wait own DMA pieces (vmcnt)", "barrier", "issue next DMA", "fragment reads + MFMA issue", "loop overhead"};
    double tot = 0;
    for (int s = 0; s < 5; s++) {
        double a = 0;
        for (int v = 0; v < nw; v++) a += (double)t[(size_t)v * 8 + s];
        a /= nw * 32.0;
        tot += a;
        printf("%-30s %8.0f cycles per chunk\n", names[s], a);
    }
    printf("sum %.0f cycles per chunk per wave (two waves per SIMD share the matrix pipe: 2 x 768 = 1536 of MFMA time per chunk)\n", tot);
    return 0;
}
