// Where a wave of conv_split_kernel<1> spends its cycles per K chunk (s_memtime deltas summed over the 32
// chunks): issue of the next chunk's global loads | LDS operand reads landed | MFMAs issued |
// wait-for-loads + split + LDS stores | barrier.  Not a product path.
#define CS_TIMING
#define CS_STAGES 2
#include "../learning3d_amd/csrc/conv_split.hip"
#include <cstdio>
#include <vector>
thread_local int g_l3d_last_hip_error = 0;
int main()
{
    const int B = 32, N = 1024, Cin = 512, Cout = 1024;
    float *x, *w, *y; void *ws;
    hipMalloc(&x, 4ul * B * N * Cin); hipMalloc(&w, 4ul * Cout * Cin); hipMalloc(&y, 4ul * B * N * Cout + 65536);
    hipMalloc(&ws, l3d_split_bytes(Cout, Cin));
    std::vector<float> h((size_t)B * N * Cin);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    hipMemcpy(x, h.data(), 4 * h.size(), hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), 4ul * Cout * Cin, hipMemcpyHostToDevice);
    l3d_split_rows(w, Cout, Cin, ws, nullptr);
    for (int i = 0; i < 3; i++) l3d_pointwise_conv_split(x, 1, ws, nullptr, nullptr, 0, B, Cin, Cout, N, 1, y, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(4 * 8 * 5);
    hipMemcpy(t.data(), y + (size_t)B * N * Cout, 8 * t.size(), hipMemcpyDeviceToHost);
    const char *names[5] = {"issue next loads", "LDS reads landed", "MFMAs issued", "wait loads+split+store", "barrier"};
    for (int blk = 0; blk < 2; blk++)
        for (int wv = 0; wv < 8; wv += 3) {
            printf("tile %d wave %d:", blk, wv);
            unsigned long long tot = 0;
            for (int i = 0; i < 5; i++) tot += t[(blk * 8 + wv) * 5 + i];
            for (int i = 0; i < 5; i++) printf("  %s %llu (%.0f%%)", names[i], t[(blk * 8 + wv) * 5 + i] / 32, 100.0 * t[(blk * 8 + wv) * 5 + i] / tot);
            printf("  | per chunk %llu\n", tot / 32);
        }
    return 0;
}
