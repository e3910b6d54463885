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
// The whole ranking value of `knn` as a 3-issue chain: k-slots (cx*2qx, cy*2qy), (cz*2qz, -xx_j*1), (1*-xx_i, 0*0)
__global__ void k5(const float *q, const float *c, float *out)
{
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    const float qx = q[i * 3], qy = q[i * 3 + 1], qz = q[i * 3 + 2], cx = c[i * 3], cy = c[i * 3 + 1], cz = c[i * 3 + 2];
    const float qxx = (qx * qx + qy * qy) + qz * qz, cw = -((cx * cx + cy * cy) + cz * cz);
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? cy : cx, h ? 2.0f * qy : 2.0f * qx, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? cw : cz, h ? 1.0f : 2.0f * qz, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? 0.0f : 1.0f, h ? 0.0f : -qxx, acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) out[(8 * (r >> 2) + 4 * h + (r & 3)) * 32 + i] = acc[r];
}
static float ref_pd(const float *q, const float *c)
{
    volatile float qxx = q[0] * q[0]; qxx = qxx + q[1] * q[1]; volatile float t = q[2] * q[2]; qxx = qxx + t;
    volatile float cxx = c[0] * c[0]; cxx = cxx + c[1] * c[1]; volatile float u = c[2] * c[2]; cxx = cxx + u;
    const float dot = fmaf(q[2], c[2], fmaf(q[1], c[1], q[0] * c[0]));
    volatile float tt = fmaf(2.0f, dot, -cxx);
    volatile float pd = tt - qxx;
    return pd;
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
    bad = tot = 0;
    for (int it = 0; it < 2000; it++) {
        float q[96], c[96], o[1024];
        for (int i = 0; i < 96; i++) { q[i] = rand() / (float)RAND_MAX; c[i] = rand() / (float)RAND_MAX; if (it & 1) { q[i] = q[i] * 4 - 2; c[i] = c[i] * 4 - 2; } }
        if (it % 3 == 0) for (int i = 0; i < 48; i++) c[i] = q[i];           // self pairs / exact cancellation
        hipMemcpy(dq, q, 384, hipMemcpyHostToDevice); hipMemcpy(dc, c, 384, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k5, dim3(1), dim3(64), 0, 0, dq, dc, dout);
        hipMemcpy(o, dout, 4096, hipMemcpyDeviceToHost);
        for (int j = 0; j < 32; j++) for (int i = 0; i < 32; i++) {
            const float want = ref_pd(q + i * 3, c + j * 3);
            tot++;
            if (memcmp(&want, &o[j * 32 + i], 4)) { if (bad < 5) printf("pd mismatch q%d c%d: chain %.9g mfma %.9g\n", i, j, want, o[j * 32 + i]); bad++; }
        }
    }
    printf("%ld / %ld ranking values differ from (-xx_j + 2 dot) - xx_i\n", bad, tot);
    return 0;
}
