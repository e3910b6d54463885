// Is the fp32 MFMA's K accumulation the fma chain fmaf(z,z',fmaf(y,y',x*x')) that the kNN ranking replays?
// D = A(32x2) B(2x32) + C with v_mfma_f32_32x32x2_f32, two issues for k = (x,y) and (z,0); compares all 1024 dots of a
// tile pair bitwise with the chain, for many random tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *q /*[32][3]*/, const float *c /*[32][3]*/, float *out /*[32 cand][32 query]*/)
{
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    // A: rows = candidates (M), k = h (0 or 1): lane (i, h) supplies A[i][h];  B: cols = queries (N): lane (i, h) supplies B[h][i]
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c[i * 3 + h], q[i * 3 + h], acc, 0, 0, 0);            // k = x, y
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? c[i * 3 + 2] : 0.f, h == 0 ? q[i * 3 + 2] : 0.f, acc, 0, 0, 0);   // k = z, 0
    for (int r = 0; r < 16; r++) out[(8 * (r >> 2) + 4 * h + (r & 3)) * 32 + i] = acc[r];
}
int main()
{
    float *dq, *dc, *dout; hipMalloc(&dq, 384); hipMalloc(&dc, 384); hipMalloc(&dout, 4096);
    long bad = 0, tot = 0;
    srand(1);
    for (int it = 0; it < 2000; it++) {
        float q[96], c[96], o[1024];
        for (int i = 0; i < 96; i++) { q[i] = rand() / (float)RAND_MAX; c[i] = rand() / (float)RAND_MAX; if (it & 1) { q[i] = q[i] * 4 - 2; c[i] = c[i] * 4 - 2; } }
        hipMemcpy(dq, q, 384, hipMemcpyHostToDevice); hipMemcpy(dc, c, 384, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dq, dc, dout);
        hipMemcpy(o, dout, 4096, hipMemcpyDeviceToHost);
        for (int j = 0; j < 32; j++) for (int i = 0; i < 32; i++) {
            const float want = fmaf(q[i * 3 + 2], c[j * 3 + 2], fmaf(q[i * 3 + 1], c[j * 3 + 1], q[i * 3] * c[j * 3]));
            tot++;
            if (memcmp(&want, &o[j * 32 + i], 4)) { if (bad < 5) printf("mismatch q%d c%d: chain %.9g mfma %.9g\n", i, j, want, o[j * 32 + i]); bad++; }
        }
    }
    printf("%ld / %ld dots differ from the fma chain\n", bad, tot);
    return 0;
}
