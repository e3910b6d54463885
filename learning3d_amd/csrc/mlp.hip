// mlp.hip -- the per-point shared-MLP (1x1 conv + folded BN + ReLU [+ max over k]) on gfx950's
// fp32 MFMA (v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32: exact fp32, 157 TFLOP/s dense).
//
// Replaces the chains of ATen ops in
//   models/dgcnn.py:32-46   get_graph_feature -> 4 x relu(bn(conv2d 1x1)) -> max over k -> cat
//   models/dgcnn.py:48      conv5 512 -> emb_dims
//   models/pointnet.py:22-49, models/pcn.py:111-125   Conv1d stacks
// which, un-fused, write and re-read the [B,C,N,k] activations through HBM (1.34 GB per forward
// at B=32, N=1024, k=20).
//
// Kernel 1: edgeconv_kernel.  One 256-thread workgroup owns 4 points x k neighbours = 4k rows of
// the [B*N*k, C] activation matrix and carries them through all four layers without leaving the
// CU: layer outputs go VGPR(acc) -> bias/ReLU -> LDS -> next layer's A operand.  The 4 waves split
// the OUTPUT channels of each layer (each wave streams only its own slice of the weights, as
// pre-packed MFMA B fragments, straight from L2 into VGPRs), and share the activation tile in LDS.
// The row order inside the tile is chosen so that a point's k neighbours are exactly the
// (M-tile, accumulator-register) pairs of ONE 16-lane group: max-over-k is 4*MT v_max in
// registers, no cross-lane traffic.        row = mt*16 + p*4 + r   <->   (point p, neighbour mt*4+r)
//
// Kernel 2: pointwise_conv_kernel.  Y[b] = act(scale * (W X[b]) + shift), a 128x128x16-tiled fp32
// GEMM on v_mfma_f32_32x32x2_f32 with the point index on the MFMA column axis so the [B,Cout,N]
// store is 128 B-coalesced, register-staged global->LDS prefetch of the next K chunk.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// packed parameter block (see l3d_edgeconv_pack)
//   layer L fragment order: [nt = Cout/16][kq][lane 64][KV]  holding  W'[k][n]  with
//     k = (kq*KV + s)*4 + (lane>>4),  n = nt*16 + (lane&15),   W'[k][n] = scale[n]*conv.weight[n][k]
//   layer 1: Cin 6 padded to 8, KV = 2;  layers 2-4: KV = 4
//   then the four bias (= BN shift) vectors.
// ---------------------------------------------------------------------------------------------
#include <string.h>
#include "edgeconv_layout.h"

extern "C" size_t l3d_edgeconv_packed_floats(int c1, int c2, int c3, int c4)
{
    if (c1 != EC_C1 || c2 != EC_C2 || c3 != EC_C3 || c4 != EC_C4) return 0;
    return EC_PACKED_FLOATS;
}

// host-side fp32 -> three bf16 (round-to-nearest-even), v = h + m + l exactly for finite v
static inline uint16_t l3d_f2bf_host(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float l3d_bf2f_host(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline void l3d_split3_host(float v, uint16_t (&pl)[3])
{
    pl[0] = l3d_f2bf_host(v);
    const float r = v - l3d_bf2f_host(pl[0]);
    pl[1] = l3d_f2bf_host(r);
    pl[2] = l3d_f2bf_host(r - l3d_bf2f_host(pl[1]));
}

static void pack_layer(const float *w /*[cout][cin]*/, const float *scale, int cin, int cin_pad,
                       int cout, int kv, float *dst)
{
    const int kq_n = cin_pad / (4 * kv);
    for (int nt = 0; nt < cout / 16; nt++)
        for (int kq = 0; kq < kq_n; kq++)
            for (int lane = 0; lane < 64; lane++)
                for (int s = 0; s < kv; s++) {
                    const int k = (kq * kv + s) * 4 + (lane >> 4);
                    const int n = nt * 16 + (lane & 15);
                    float v = k < cin ? w[(size_t)n * cin + k] : 0.f;
                    if (scale) v *= scale[n];
                    dst[(((size_t)nt * kq_n + kq) * 64 + lane) * kv + s] = v;
                }
}

extern "C" int l3d_edgeconv_pack(const float *const w[4], const float *const scale[4],
                                     const float *const shift[4], const float *act_mag, int c1, int c2, int c3, int c4,
                                     float *packed)
{
    L3D_REQUIRE(w && packed && w[0] && w[1] && w[2] && w[3]);
    if (c1 != EC_C1 || c2 != EC_C2 || c3 != EC_C3 || c4 != EC_C4) return L3D_ERR_UNSUPPORTED;
    pack_layer(w[0], scale ? scale[0] : nullptr, 6, 8, EC_C1, 2, packed + EC_OFF_W1);
    pack_layer(w[1], scale ? scale[1] : nullptr, EC_C1, EC_C1, EC_C2, 4, packed + EC_OFF_W2);
    pack_layer(w[2], scale ? scale[2] : nullptr, EC_C2, EC_C2, EC_C3, 4, packed + EC_OFF_W3);
    pack_layer(w[3], scale ? scale[3] : nullptr, EC_C3, EC_C3, EC_C4, 4, packed + EC_OFF_W4);
    const int cs[4] = {EC_C1, EC_C2, EC_C3, EC_C4};
    const int off[4] = {EC_OFF_B1, EC_OFF_B2, EC_OFF_B3, EC_OFF_B4};
    for (int l = 0; l < 4; l++)
        for (int c = 0; c < cs[l]; c++) packed[off[l] + c] = (shift && shift[l]) ? shift[l][c] : 0.f;
    // second copy of the weights in the layout of the register-chained kernel (edgeconv2.hip):
    //   layer 1: [m][lane][2]    = W'[16m + (lane&15)][4s + (lane>>4)],            s = 0,1 (Cin 6 -> 8)
    //   layer L: [m][q][lane][4] = W'[16m + (lane&15)][16q + 4(lane>>4) + e],      e = 0..3
    const int cin[4] = {6, EC_C1, EC_C2, EC_C3};
    const int o2[4] = {EC2_OFF_W1, EC2_OFF_W2, EC2_OFF_W3, EC2_OFF_W4};
    for (int l = 0; l < 4; l++) {
        const float *wl = w[l];
        const float *sc = scale ? scale[l] : nullptr;
        float *dst = packed + o2[l];
        for (int m = 0; m < cs[l] / 16; m++) {
            if (l == 0) {
                for (int lane = 0; lane < 64; lane++)
                    for (int sidx = 0; sidx < 2; sidx++) {
                        const int oc = 16 * m + (lane & 15), ic = 4 * sidx + (lane >> 4);
                        float v = ic < 6 ? wl[(size_t)oc * 6 + ic] : 0.f;
                        if (sc) v *= sc[oc];
                        dst[(m * 64 + lane) * 2 + sidx] = v;
                    }
            } else {
                const int nq = cin[l] / 16;
                for (int q = 0; q < nq; q++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int e = 0; e < 4; e++) {
                            const int oc = 16 * m + (lane & 15), ic = 16 * q + 4 * (lane >> 4) + e;
                            float v = wl[(size_t)oc * cin[l] + ic];
                            if (sc) v *= sc[oc];
                            dst[(((size_t)m * nq + q) * 64 + lane) * 4 + e] = v;
                        }
            }
        }
    }
    // third copy (layers 2-4) for the bf16x3 kernel (edgeconv_split.hip): W' = h + m + l exactly,
    //   block [step = ((m/2)*S + s)*2 + (m&1)][plane][lane][slot 8]  (M-tile pair outer, k-step, M-tile inner):
    //   slot e (0..3) = W'[16m + (lane&15)][32s + 4(lane>>4) + e],  slot 4+e = W'[..][32s + 16 + 4(lane>>4) + e]
    const int o3[4] = {0, EC3_OFF_W2, EC3_OFF_W3, EC3_OFF_W4};
    for (int l = 1; l < 4; l++) {
        const float *wl = w[l];
        const float *sc = scale ? scale[l] : nullptr;
        uint16_t *dst = (uint16_t *)(packed + o3[l]);
        const int S = cin[l] / 32;
        for (int m = 0; m < cs[l] / 16; m++)
            for (int sidx = 0; sidx < S; sidx++)
                for (int lane = 0; lane < 64; lane++)
                    for (int slot = 0; slot < 8; slot++) {
                        const int oc = 16 * m + (lane & 15);
                        const int ic = 32 * sidx + 16 * (slot >> 2) + 4 * (lane >> 4) + (slot & 3);
                        float v = wl[(size_t)oc * cin[l] + ic];
                        if (sc) v *= sc[oc];
                        uint16_t pl[3];
                        l3d_split3_host(v, pl);
                        for (int p = 0; p < 3; p++)
                            dst[(((((size_t)(m >> 1) * S + sidx) * 2 + (m & 1)) * 3 + p) * 64 + lane) * 8 + slot] = pl[p];
                    }
    }
    // fourth copy (layers 2-4) for the f16x2 kernel (edgeconv_f16.hip): W = w' 2^S, planes H = f16(W), Hs = f16(H 2^-12),
    //   M = f16(W - H) in the fragment order of the third copy; biases times 2^S; 2^-S per layer.
    const int o4[4] = {0, EC4_OFF_W2, EC4_OFF_W3, EC4_OFF_W4};
    const int ob4[4] = {0, EC4_OFF_B2, EC4_OFF_B3, EC4_OFF_B4};
    int Sl[4] = {0, 0, 0, 0};
    for (int l = 1; l < 4; l++) {
        const float *wl = w[l];
        const float *sc = scale ? scale[l] : nullptr;
        float wmax = 0.f;
        for (int oc = 0; oc < cs[l]; oc++)
            for (int ic = 0; ic < cin[l]; ic++) {
                float v = wl[(size_t)oc * cin[l] + ic];
                if (sc) v *= sc[oc];
                wmax = fmaxf(wmax, fabsf(v));
            }
        int S = 0;                                            // max|w| 2^S in [4, 8); all-zero weights: S = 0
        if (wmax > 0.f && wmax < INFINITY) {
            int e;
            frexpf(wmax, &e);                                 // wmax = f 2^e, f in [0.5, 1)
            S = 3 - e;
        }
        Sl[l] = S;
        const float up = ldexpf(1.0f, S);
        uint16_t *dst = (uint16_t *)(packed + o4[l]);
        const int St = cin[l] / 32;
        for (int m = 0; m < cs[l] / 16; m++)
            for (int sidx = 0; sidx < St; sidx++)
                for (int lane = 0; lane < 64; lane++)
                    for (int slot = 0; slot < 8; slot++) {
                        const int oc = 16 * m + (lane & 15);
                        const int ic = 32 * sidx + 16 * (slot >> 2) + 4 * (lane >> 4) + (slot & 3);
                        float v = wl[(size_t)oc * cin[l] + ic];
                        if (sc) v *= sc[oc];
                        v *= up;                               // exact
                        const _Float16 H = (_Float16)v;        // round-to-nearest-even
                        const _Float16 M = (_Float16)(v - (float)H);
                        const _Float16 Hs = (_Float16)((float)H * 0x1p-12f);
                        const _Float16 pl[3] = {H, Hs, M};
                        for (int p = 0; p < 3; p++) {
                            uint16_t bits;
                            memcpy(&bits, &pl[p], 2);
                            dst[(((((size_t)(m >> 1) * St + sidx) * 2 + (m & 1)) * 3 + p) * 64 + lane) * 8 + slot] = bits;
                        }
                    }
    }
    // plane exponents from the expected activation magnitudes (act_mag[l] 2^T_l in [2^11, 2^12); default magnitude 1)
    int Tl[4], Tout = 1 << 20;
    for (int l = 0; l < 4; l++) {
        const float mag = (act_mag && act_mag[l] > 0.f && act_mag[l] < INFINITY) ? act_mag[l] : 1.0f;
        int e;
        frexpf(mag, &e);
        Tl[l] = 12 - e;
        Tout = Tl[l] < Tout ? Tl[l] : Tout;
    }
    for (int i = 0; i < 16; i++) packed[EC4_OFF_SC + i] = 0.f;
    for (int l = 0; l < 4; l++) {
        const int A = l == 0 ? 0 : Sl[l] + Tl[l - 1];           // log2 of the accumulator's scale
        if (l < 3) packed[EC4_OFF_SC + l] = ldexpf(1.0f, Tl[l] - A);
        packed[EC4_OFF_SC + 4 + l] = ldexpf(1.0f, -A);
        packed[EC4_OFF_SC + 8 + l] = ldexpf(1.0f, Tout - A);
        if (l >= 1)
            for (int c = 0; c < cs[l]; c++) packed[ob4[l] + c] = ldexpf((shift && shift[l]) ? shift[l][c] : 0.f, A);
    }
    packed[EC4_OFF_SC + 12] = ldexpf(1.0f, -Tout);

    // fifth copy (edgeconv_layout.h): two weight planes, accumulators of layers 1-3 ARE the planes of the next layer.
    // Weights keep the fourth copy's scaling (max|W| in [4,8): an unscaled fp16 residual M = f16(W - H) is a normal number
    // for |W| >= 2^-3, and costs 2^-25 absolute below that -- against typical weights of order one).  The plane exponents
    // then follow from the weights, T_l = S_l + T_(l-1), and only T_1 is free: it is set so that the layer whose expected
    // magnitude lands highest sits in [2^11, 2^12) (fp16 tops out at 2^16).  The other layers sit lower; an activation's
    // unscaled residual costs 2^-25 absolute in plane units, harmless while a layer's expected magnitude is placed at
    // >= 2^4 -- otherwise the block is marked unusable and the host runs the three-plane kernel.
    {
        const int o5[4] = {0, EC5_OFF_W2, EC5_OFF_W3, EC5_OFF_W4};
        const int ob5[4] = {EC5_OFF_B1, EC5_OFF_B2, EC5_OFF_B3, EC5_OFF_B4};
        int el[4], A5[4], T5[4];
        for (int l = 0; l < 4; l++) {
            const float mag = (act_mag && act_mag[l] > 0.f && act_mag[l] < INFINITY) ? act_mag[l] : 1.0f;
            frexpf(mag, &el[l]);                                     // mag in [2^(e-1), 2^e)
        }
        // T5[l] = T1 + cum[l], cum = S_2 + .. + S_(l+1) (layers are 1-based in the comments, l is 0-based here).  With
        // max|W| in [4,8) every layer lifts the planes by S_l ~ 5 binades (default-initialised convs: max|w| ~ 1/8), more
        // than the 8 binades between "expected magnitude at 2^12" and "at 2^4" allow over two layers: layers 2 and 3 then
        // give up to two binades of weight scale each (max|W| in [1,2) at worst: M's 2^-25 floor against typical |W| ~ 0.3).
        int S5[4] = {0, Sl[1], Sl[2], Sl[3]};
        int T1 = 0;
        bool ok = true;
        for (int pass = 0; pass < 5; pass++) {
            const int cum[3] = {0, S5[1], S5[1] + S5[2]};
            T1 = 1 << 20;
            for (int l = 0; l < 3; l++) T1 = (12 - el[l] - cum[l]) < T1 ? (12 - el[l] - cum[l]) : T1;
            int lo = 1 << 20;
            for (int l = 0; l < 3; l++) {
                T5[l] = T1 + cum[l];
                lo = (el[l] + T5[l]) < lo ? (el[l] + T5[l]) : lo;
            }
            ok = lo >= 4;                                            // every layer's expected magnitude placed at >= 2^3
            if (ok || pass == 4) break;
            if (S5[2] > Sl[2] - 2) S5[2]--;                          // lower the later layer's weight scale first
            else if (S5[1] > Sl[1] - 2) S5[1]--;
            else break;
        }
        T5[3] = 0;
        A5[0] = T5[0];
        for (int l = 1; l < 4; l++) A5[l] = S5[l] + T5[l - 1];       // == T5[l] for l = 1, 2
        const int Tout5 = Tout;                                      // the pooled planes for conv5: placed by the expected magnitudes alone
        // layer 1 (fp32 MFMA): the second copy's layout, times 2^T_1 (exact)
        for (int i = 0; i < 8 * EC_C1; i++) packed[EC5_OFF_W1 + i] = ldexpf(packed[EC2_OFF_W1 + i], T5[0]);
        for (int c = 0; c < EC_C1; c++) packed[EC5_OFF_B1 + c] = ldexpf((shift && shift[0]) ? shift[0][c] : 0.f, T5[0]);
        for (int l = 1; l < 4; l++) {
            const float *wl = w[l];
            const float *sc = scale ? scale[l] : nullptr;
            const float up = ldexpf(1.0f, S5[l]);
            uint16_t *dst = (uint16_t *)(packed + o5[l]);
            const int St = cin[l] / 32;
            for (int m = 0; m < cs[l] / 16; m++)
                for (int sidx = 0; sidx < St; sidx++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int slot = 0; slot < 8; slot++) {
                            const int oc = 16 * m + (lane & 15);
                            const int ic = 32 * sidx + 16 * (slot >> 2) + 4 * (lane >> 4) + (slot & 3);
                            float v = wl[(size_t)oc * cin[l] + ic];
                            if (sc) v *= sc[oc];
                            v *= up;                               // exact (power of two)
                            const _Float16 H = (_Float16)v;
                            const _Float16 M = (_Float16)(v - (float)H);
                            const _Float16 pl[2] = {H, M};
                            for (int p = 0; p < 2; p++) {
                                uint16_t bits;
                                memcpy(&bits, &pl[p], 2);
                                dst[(((((size_t)(m >> 1) * St + sidx) * 2 + (m & 1)) * 2 + p) * 64 + lane) * 8 + slot] = bits;
                            }
                        }
            for (int c = 0; c < cs[l]; c++) packed[ob5[l] + c] = ldexpf((shift && shift[l]) ? shift[l][c] : 0.f, A5[l]);
        }
        for (int i = 0; i < 16; i++) packed[EC5_OFF_SC + i] = 0.f;
        for (int l = 0; l < 4; l++) {
            if (l < 3) packed[EC5_OFF_SC + l] = 1.0f;
            packed[EC5_OFF_SC + 4 + l] = ldexpf(1.0f, -A5[l]);
            packed[EC5_OFF_SC + 8 + l] = ldexpf(1.0f, Tout5 - A5[l]);
        }
        packed[EC5_OFF_SC + 12] = ldexpf(1.0f, -Tout5);
        packed[EC5_OFF_SC + 13] = ok ? 1.0f : 0.f;
    }
    return L3D_OK;
}

extern "C" int l3d_edgeconv_packed_v2_flag_index(void) { return EC5_OFF_SC + 13; }

// One layer for one wave: acc[mt][i] += A(act rows) x B(weight fragments of this wave's N-tiles)
//   act   : LDS, [4*MT*4 rows][CIN + 2] (the +2 skew makes the 16-row x 2-k ds_read_b32 pattern
//           hit 32 distinct banks)
//   wfrag : global, this layer's packed fragments
template <int MT, int CIN, int KV, int NT>
__device__ __forceinline__ void ec_layer_mma(const float *__restrict__ act,
                                             const float *__restrict__ wfrag, int wave, int lane,
                                             f32x4 (&acc)[MT][NT])
{
    constexpr int S = CIN + 2;
    constexpr int KQ = CIN / (4 * KV);
    typedef float vecT __attribute__((ext_vector_type(KV)));
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < NT; i++) acc[mt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const vecT *wp = (const vecT *)wfrag + ((size_t)(wave * NT) * KQ) * 64 + lane;
    // Software pipeline, two register sets each (no copies), order pinned with sched_barrier:
    //   A operands: the ds_reads of k-step pair p+1 are in flight while pair p's MFMAs run;
    //   B fragments: the global loads of weight block kq+1 are issued when block kq starts.
    constexpr int NP = CIN / 8;                 // pairs of k-steps
    vecT bf[2][NT];
    float av[2][MT][2];
#pragma unroll
    for (int i = 0; i < NT; i++) bf[0][i] = wp[(size_t)(i * KQ) * 64];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        av[0][mt][0] = act[(mt * 16 + r) * S + g];
        av[0][mt][1] = act[(mt * 16 + r) * S + 4 + g];
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int cur = p & 1, nxt = cur ^ 1;
        const int kq = (2 * p) / KV, s0 = (2 * p) % KV;
        if (s0 == 0 && kq + 1 < KQ) {
#pragma unroll
            for (int i = 0; i < NT; i++) bf[(kq + 1) & 1][i] = wp[(size_t)(i * KQ + kq + 1) * 64];
        }
        if (p + 1 < NP) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                av[nxt][mt][0] = act[(mt * 16 + r) * S + (2 * p + 2) * 4 + g];
                av[nxt][mt][1] = act[(mt * 16 + r) * S + (2 * p + 3) * 4 + g];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
            for (int i = 0; i < NT; i++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    acc[mt][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][mt][e], bf[kq & 1][i][s0 + e],
                                                                      acc[mt][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// bias + ReLU, max over the point's k neighbours (registers only), optional write of the
// activations for the next layer, and the pooled [B*N, CTOT] (channel-last) store.
template <int MT, int NT, int COUT_STRIDE, bool WRITE_NEXT>
__device__ __forceinline__ void ec_layer_epilogue(f32x4 (&acc)[MT][NT],
                                                  const float *__restrict__ bias, int wave, int lane,
                                                  float *__restrict__ act_out,
                                                  float *__restrict__ pooled_row, bool store_ok)
{
    const int c16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int ch = (wave * NT + i) * 16 + c16;
        const float bv = bias[ch];
        float mx = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const float v = fmaxf(acc[mt][i][rr] + bv, 0.f);
                mx = fmaxf(mx, v);
                if (WRITE_NEXT) act_out[(mt * 16 + g * 4 + rr) * COUT_STRIDE + ch] = v;
            }
        if (store_ok) pooled_row[ch] = mx;
    }
}

template <int MT>
__global__ __launch_bounds__(256, MT <= 5 ? 3 : 1) void edgeconv_kernel(const float *__restrict__ xyz,
                                                          const int64_t *__restrict__ idx, int N,
                                                          int k, const float *__restrict__ packed,
                                                          float *__restrict__ pooled /*[B*N][512]*/)
{
    constexpr int ROWS = MT * 16;
    constexpr int CTOT = EC_C1 + EC_C2 + EC_C3 + EC_C4;
    // One LDS arena of 2 x ROWS x 66 floats (42 KiB at MT=5 -> three workgroups per CU):
    //   X = first half : feature tile (8 ch, stride 10), later h2 (64 ch, stride 66)
    //   Y = second half: h1 (64 ch, stride 66)
    //   h3 (128 ch, stride 130 = ROWS x 130 <= 2 x ROWS x 66) overlays X and Y once h2 is dead.
    __shared__ float arena[2 * ROWS * (EC_C2 + 2)];
    float *const P0 = arena;
    float *const P1 = arena + ROWS * (EC_C2 + 2);
    float *const P3 = arena;
    static_assert(ROWS * (EC_C3 + 2) <= 2 * ROWS * (EC_C2 + 2), "h3 must fit the arena");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * 4;

    // ---- stage the graph feature rows: (neighbour xyz, centre xyz, 0, 0)   dgcnn.py:32 ----
    if (tid < ROWS) {
        const int row = tid;
        const int p = (row >> 2) & 3;
        const int j = (row >> 4) * 4 + (row & 3);
        const int n = min(n0 + p, N - 1);
        const int jj = j < k ? j : 0;                      // pad k up to 4*MT with a duplicate
        const int64_t nb = idx[((size_t)b * N + n) * k + jj];
        const float *pn = xyz + ((size_t)b * N + nb) * 3;
        const float *pc = xyz + ((size_t)b * N + n) * 3;
        float *f = P0 + row * 10;
        f[0] = pn[0]; f[1] = pn[1]; f[2] = pn[2];
        f[3] = pc[0]; f[4] = pc[1]; f[5] = pc[2];
        f[6] = 0.f; f[7] = 0.f;
    }
    __syncthreads();

    const int g = lane >> 4;
    const bool store_ok = (n0 + g) < N;
    float *prow = pooled + ((size_t)b * N + min(n0 + g, N - 1)) * CTOT;

    {   // layer 1: 6(8) -> 64
        f32x4 acc[MT][1];
        ec_layer_mma<MT, 8, 2, 1>(P0, packed + EC_OFF_W1, wave, lane, acc);
        ec_layer_epilogue<MT, 1, EC_C1 + 2, true>(acc, packed + EC_OFF_B1, wave, lane, P1, prow, store_ok);
    }
    __syncthreads();
    {   // layer 2: 64 -> 64
        f32x4 acc[MT][1];
        ec_layer_mma<MT, EC_C1, 4, 1>(P1, packed + EC_OFF_W2, wave, lane, acc);
        ec_layer_epilogue<MT, 1, EC_C2 + 2, true>(acc, packed + EC_OFF_B2, wave, lane, P0, prow + EC_C1, store_ok);
    }
    __syncthreads();
    {   // layer 3: 64 -> 128
        f32x4 acc[MT][2];
        ec_layer_mma<MT, EC_C2, 4, 2>(P0, packed + EC_OFF_W3, wave, lane, acc);
        __syncthreads();                 // every wave is done reading h2 before h3 overlays it
        ec_layer_epilogue<MT, 2, EC_C3 + 2, true>(acc, packed + EC_OFF_B3, wave, lane, P3, prow + EC_C1 + EC_C2, store_ok);
    }
    __syncthreads();
    {   // layer 4: 128 -> 256 (activations are only max-pooled, never stored)
        f32x4 acc[MT][4];
        ec_layer_mma<MT, EC_C3, 4, 4>(P3, packed + EC_OFF_W4, wave, lane, acc);
        ec_layer_epilogue<MT, 4, 0, false>(acc, packed + EC_OFF_B4, wave, lane, nullptr, prow + EC_C1 + EC_C2 + EC_C3, store_ok);
    }
}

extern "C" int l3d_edgeconv_forward(const float *xyz, const int64_t *idx, int B, int N, int k,
                                    const float *packed, int c1, int c2, int c3, int c4,
                                    float *pooled, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && idx && packed && pooled && B > 0 && N > 0 && k > 0);
    if (c1 != EC_C1 || c2 != EC_C2 || c3 != EC_C3 || c4 != EC_C4) return L3D_ERR_UNSUPPORTED;
    if (B > 65535) return L3D_ERR_UNSUPPORTED;
    dim3 grid(l3d_divup(N, 4), B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (k <= 8)       hipLaunchKernelGGL(edgeconv_kernel<2>, grid, block, 0, st, xyz, idx, N, k, packed, pooled);
    else if (k <= 16) hipLaunchKernelGGL(edgeconv_kernel<4>, grid, block, 0, st, xyz, idx, N, k, packed, pooled);
    else if (k <= 20) hipLaunchKernelGGL(edgeconv_kernel<5>, grid, block, 0, st, xyz, idx, N, k, packed, pooled);
    else if (k <= 32) hipLaunchKernelGGL(edgeconv_kernel<8>, grid, block, 0, st, xyz, idx, N, k, packed, pooled);
    else return L3D_ERR_UNSUPPORTED;
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// pointwise conv:  y[b][co][n] = act(scale[co] * sum_ci w[co][ci] * X[b](ci,n) + shift[co])
//   shift is indexed [b*shift_bstride + co] (shift_bstride = 0: shared by the batch; = Cout: one
//   vector per cloud -- how a broadcast global feature concatenated to every point is folded away)
//   X channel-first  [B,Cin,N]   (x_channel_last = 0, torch Conv1d layout)   or
//   X channel-last   [B,N,Cin]   (x_channel_last = 1, the EdgeConv kernel's pooled output)
// Workgroup tile 128 (co) x 128 (n), K chunk 16; 4 waves as 2x2, each 64x64 = 2x2 MFMA 32x32x2.
// ---------------------------------------------------------------------------------------------
#define PW_TM 128
#define PW_TN 128
#ifndef PW_TK
#define PW_TK 16       // K chunk per LDS stage (tools/probe_mlp.hip overrides it to explore)
#endif
#define PW_LD (128 + 4)      // LDS row stride (floats) of the k-major tiles; 16 B aligned

// FULL: every tile is complete and every row is 16-byte addressable (Cout, N multiples of 128, Cin of
// 16): the staging loads are then unconditional.  The guarded variant wraps each 16-byte load in an
// exec-mask branch -- 26 s_and_saveexec / s_cbranch_execz per K chunk, which sat in front of every
// chunk's MFMAs and cost ~11 % at the conv5 shape.
// POOL: the epilogue additionally takes the max over every `pool` (8, 16, 32 or 64) consecutive points
// and writes y [B, Cout, N / pool]: the "shared MLP over [B,C,S,K] then max over K" tail of a PointNet++
// set-abstraction / flow-embedding layer (models/flownet3d.py:118-122, :170-176) without materialising
// the [B,Cout,S,K] activation or launching a reduction over it.
template <bool XCL, bool FULL, bool POOL>
__global__ __launch_bounds__(256, 2) void pointwise_conv_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ scale,
    const float *__restrict__ shift, int shift_bstride, int Cin, int Cout, int N, int relu,
    float *__restrict__ y, int pool)
{
    __shared__ __attribute__((aligned(16))) float As[PW_TK][PW_LD];     // As[k][co]
    __shared__ __attribute__((aligned(16))) float Bs[PW_TK][PW_LD];     // Bs[k][n]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.z;
    const int co0 = blockIdx.y * PW_TM, n0 = blockIdx.x * PW_TN;
    const int wm = wave >> 1, wn = wave & 1;          // wave tile origin (64x64) inside the WG tile
    const float *xb = x + (size_t)b * Cin * N;

    // global->register staging assignment
    //   A: thread -> (co = tid/2, 8 consecutive k at (tid&1)*8)
    //   B (channel-first): thread -> (k = tid/16, 8 consecutive n at (tid&15)*8)
    //   B (channel-last) : thread -> (n = tid/2,  8 consecutive k at (tid&1)*8)
    float ra[8], rb[8];
    const bool cin_vec = (Cin & 3) == 0, n_vec = (N & 3) == 0;
    auto load8 = [](float (&r)[8], const float *src, int pos, int limit, bool row_ok, bool vec) {
        if (FULL || (row_ok && vec && pos + 8 <= limit)) {
            const float4 v0 = *(const float4 *)(src + pos), v1 = *(const float4 *)(src + pos + 4);
            r[0] = v0.x; r[1] = v0.y; r[2] = v0.z; r[3] = v0.w;
            r[4] = v1.x; r[5] = v1.y; r[6] = v1.z; r[7] = v1.w;
        } else if (row_ok && pos + 8 <= limit) {
            // in range but not 16-byte addressable (row length not a multiple of 4, e.g. Cin = 259): eight plain
            // loads, no per-element exec-mask branch -- only the last chunk of a row takes the guarded path below
#pragma unroll
            for (int e = 0; e < 8; e++) r[e] = src[pos + e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) r[e] = (row_ok && pos + e < limit) ? src[pos + e] : 0.f;
        }
    };
    auto load_chunk = [&](int k0) {
        {
            const int co = co0 + (tid >> 1);
            const bool rok = co < Cout;
            load8(ra, w + (size_t)(rok ? co : 0) * Cin, k0 + (tid & 1) * 8, Cin, rok, cin_vec);
        }
        if (XCL) {
            const int n = n0 + (tid >> 1);
            const bool nok = n < N;
            load8(rb, xb + (size_t)(nok ? n : 0) * Cin, k0 + (tid & 1) * 8, Cin, nok, cin_vec);
        } else {
            const int kk = k0 + (tid >> 4);
            const bool kok = kk < Cin;
            load8(rb, xb + (size_t)(kok ? kk : 0) * N, n0 + (tid & 15) * 8, N, kok, n_vec);
        }
    };
    auto store_chunk = [&]() {
        {
            const int m = tid >> 1, kb = (tid & 1) * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) As[kb + e][m] = ra[e];
        }
        if (XCL) {
            const int n = tid >> 1, kb = (tid & 1) * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) Bs[kb + e][n] = rb[e];
        } else {
            const int kk = tid >> 4, nb = (tid & 15) * 8;
            *(float4 *)&Bs[kk][nb] = make_float4(rb[0], rb[1], rb[2], rb[3]);
            *(float4 *)&Bs[kk][nb + 4] = make_float4(rb[4], rb[5], rb[6], rb[7]);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, kh = lane >> 5;
    load_chunk(0);
    for (int k0 = 0; k0 < Cin; k0 += PW_TK) {
        __syncthreads();                 // previous chunk's MFMA reads are done
        store_chunk();
        __syncthreads();
#ifndef PW_PROBE_NO_GLOBAL
        if (k0 + PW_TK < Cin) load_chunk(k0 + PW_TK);     // overlaps with the MFMAs below
#endif
        // software-pipelined operand fetch: the ds_reads of k-step s+1 are in flight while the four
        // 64-cycle MFMAs of k-step s occupy the matrix pipe (otherwise each k-step exposes one LDS
        // round trip: measured 73 % MFMA-busy before, see profiles/)
        float av[2][2], bv[2][2];          // [k-step parity][tile]: two register sets, no copies
#pragma unroll
        for (int i = 0; i < 2; i++) av[0][i] = As[kh][wm * 64 + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < 2; j++) bv[0][j] = Bs[kh][wn * 64 + j * 32 + l31];
#pragma unroll
        for (int st = 0; st < PW_TK / 2; st++) {
            const int cur = st & 1, nxt = cur ^ 1;
            if (st + 1 < PW_TK / 2) {
#pragma unroll
                for (int i = 0; i < 2; i++) av[nxt][i] = As[2 * (st + 1) + kh][wm * 64 + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < 2; j++) bv[nxt][j] = Bs[2 * (st + 1) + kh][wn * 64 + j * 32 + l31];
            }
            // pin the order "reads of step s+1, then MFMAs of step s" (hipcc otherwise sinks the reads
            // below the MFMAs and waits on them immediately)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // epilogue: D[row = co][col = n]; col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    if (POOL) {
        const int Np = N / pool;
        float *yb = y + (size_t)b * Cout * Np;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int co = co0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                const bool co_ok = FULL || co < Cout;
                const float sc = (scale && co_ok) ? scale[co] : 1.f;
                const float sh = (shift && co_ok) ? shift[(size_t)b * shift_bstride + co] : 0.f;
                float v[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int n = n0 + wn * 64 + j * 32 + l31;
                    v[j] = acc[i][j][e] * sc + sh;
                    if (relu) v[j] = l3d_act(v[j], relu);
                    if (!FULL && n >= N) v[j] = -INFINITY;          // N % pool == 0: a group is all in or all out
                }
                if (pool == 64) v[0] = fmaxf(v[0], v[1]);
                const int span = pool < 32 ? pool : 32;             // lanes to reduce over (xor masks stay inside the 32-lane half)
                v[0] = l3d_group_max(v[0], span);
                v[1] = l3d_group_max(v[1], span);
                if (co_ok && (l31 & (span - 1)) == 0) {
                    const int nb = n0 + wn * 64 + l31;
                    if (pool == 64) {
                        if (nb < N) yb[(size_t)co * Np + nb / 64] = v[0];
                    } else {
                        if (nb < N) yb[(size_t)co * Np + nb / pool] = v[0];
                        if (nb + 32 < N) yb[(size_t)co * Np + (nb + 32) / pool] = v[1];
                    }
                }
            }
        return;
    }
    float *yb = y + (size_t)b * Cout * N;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int co = co0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            if (!FULL && co >= Cout) continue;
            const float sc = scale ? scale[co] : 1.f;
            const float sh = shift ? shift[(size_t)b * shift_bstride + co] : 0.f;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int n = n0 + wn * 64 + j * 32 + l31;
                float v = acc[i][j][e] * sc + sh;
                if (relu) v = l3d_act(v, relu);
#ifdef PW_PROBE_NO_STORE
                if (v == 123.456f)
#endif
                if (FULL || n < N) yb[(size_t)co * N + n] = v;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Narrow heads (Cout <= 8, e.g. PCN's folding conv7 512 -> 3, FlowNet3D's conv2 128 -> 3): a 128-wide
// MFMA tile would be > 90 % padding.  HBM-bound instead: one thread per point reads its Cin inputs
// (coalesced over points for channel-first x, float4 runs for channel-last x), weights are
// wave-uniform (scalar loads), COUT accumulators in registers.
// ---------------------------------------------------------------------------------------------
template <int COUT, bool XCL>
__global__ __launch_bounds__(256) void pointwise_conv_narrow_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ scale,
    const float *__restrict__ shift, int shift_bstride, int Cin, int N, int relu, float *__restrict__ y)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float *xb = x + (size_t)b * Cin * N;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; c++) acc[c] = 0.f;
    if (XCL) {
        const float *xr = xb + (size_t)n * Cin;
        for (int k = 0; k < Cin; k++) {
            const float v = xr[k];
#pragma unroll
            for (int c = 0; c < COUT; c++) acc[c] = fmaf(w[c * Cin + k], v, acc[c]);
        }
    } else {
#pragma unroll 4
        for (int k = 0; k < Cin; k++) {
            const float v = xb[(size_t)k * N + n];
#pragma unroll
            for (int c = 0; c < COUT; c++) acc[c] = fmaf(w[c * Cin + k], v, acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < COUT; c++) {
        float v = acc[c] * (scale ? scale[c] : 1.f) + (shift ? shift[(size_t)b * shift_bstride + c] : 0.f);
        if (relu) v = l3d_act(v, relu);
        y[((size_t)b * COUT + c) * N + n] = v;
    }
}

template <bool XCL>
static int launch_narrow(const float *x, const float *w, const float *scale, const float *shift,
                         int shift_bstride, int B, int Cin, int Cout, int N, int relu, float *y, hipStream_t st)
{
    dim3 grid(l3d_divup(N, 256), B), block(256);
#define L3D_NARROW(CO)                                                                                  \
    case CO:                                                                                            \
        hipLaunchKernelGGL((pointwise_conv_narrow_kernel<CO, XCL>), grid, block, 0, st, x, w, scale,    \
                           shift, shift_bstride, Cin, N, relu, y);                                      \
        break;
    switch (Cout) {
        L3D_NARROW(1) L3D_NARROW(2) L3D_NARROW(3) L3D_NARROW(4) L3D_NARROW(5) L3D_NARROW(6) L3D_NARROW(7) L3D_NARROW(8)
        default: return L3D_ERR_UNSUPPORTED;
    }
#undef L3D_NARROW
    return l3d_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Streaming conv for the narrowest, memory-bound layers of a PointNet++ shared MLP (FlowNet3D sa1,
// models/flownet3d.py:73-123: 6 -> 32 -> 32 over 1 M columns): Cout <= 32, Cin <= 128, channel-first x.  The
// 128 x 128 MFMA tile above is 75 % padding there and its per-workgroup prologue / epilogue dominates.  Here one
// thread owns one column: x is read once, coalesced over columns; W^T sits in LDS and is read as wave-uniform
// broadcasts; the 32 accumulators live in registers and advance two channels per packed-fp32 FMA.
// POOL: max over `pool` consecutive columns (lanes) by shuffles, y [B, Cout, N / pool].
// Measured at B=64, N=16384: 6 -> 32 88 -> 35 us, 32 -> 32 118 -> 61 us.  Two wider variants were tried and lost
// to the tile kernel: this scheme at Cout = 64 (the broadcast LDS reads, 16 per input channel, saturate the LDS
// pipe: 32 -> 64 + max 264 -> 322 us) and an LDS-free fp32-MFMA version with the weights in registers (313 us).
// ---------------------------------------------------------------------------------------------
template <int COUT, bool POOL>
__global__ __launch_bounds__(256) void pointwise_conv_stream_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ scale,
    const float *__restrict__ shift, int shift_bstride, int Cin, int Cout, int N, int relu, float *__restrict__ y, int pool)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float wT[];          // [Cin][COUT], zero-padded beyond Cout
    const int b = blockIdx.y, tid = threadIdx.x;
    const int co_off = blockIdx.z * COUT;                               // Cout > COUT: one pass over x per COUT channels
    for (int e = tid; e < Cin * COUT; e += 256) {
        const int k = e / COUT, c = e - k * COUT;
        wT[e] = co_off + c < Cout ? w[(size_t)(co_off + c) * Cin + k] : 0.f;
    }
    __syncthreads();
    const int n = blockIdx.x * 256 + tid;
    const int nc = min(n, N - 1);
    const float *xb = x + (size_t)b * Cin * N + nc;
    f32x2 acc[COUT / 2];
#pragma unroll
    for (int c = 0; c < COUT / 2; c++) acc[c] = f32x2{0.f, 0.f};
#pragma unroll 2
    for (int k = 0; k < Cin; k++) {
        const float xv = xb[(size_t)k * N];
        const f32x2 x2 = {xv, xv};
        const float4 *wk = (const float4 *)(wT + k * COUT);
#pragma unroll
        for (int c4 = 0; c4 < COUT / 4; c4++) {
            const float4 wv = wk[c4];
            acc[2 * c4] = __builtin_elementwise_fma(f32x2{wv.x, wv.y}, x2, acc[2 * c4]);
            acc[2 * c4 + 1] = __builtin_elementwise_fma(f32x2{wv.z, wv.w}, x2, acc[2 * c4 + 1]);
        }
    }
    const int np = POOL ? N / pool : N;
#pragma unroll
    for (int c2 = 0; c2 < COUT / 2; c2++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int c = co_off + 2 * c2 + h;
            if (c < Cout) {                                            // Cout is uniform: no divergence around the shuffles
                float v = acc[c2][h] * (scale ? scale[c] : 1.f) + (shift ? shift[(size_t)b * shift_bstride + c] : 0.f);
                if (relu) v = l3d_act(v, relu);
                if (POOL) {
                    v = l3d_group_max(v, pool < 32 ? pool : 32);
                    if (pool == 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
                    if (n < N && (tid & (pool - 1)) == 0) y[((size_t)b * Cout + c) * np + n / pool] = v;
                } else if (n < N) {
                    y[((size_t)b * Cout + c) * N + n] = v;
                }
            }
        }
    }
}

template <bool POOL>
static int launch_stream(const float *x, const float *w, const float *scale, const float *shift, int shift_bstride,
                         int B, int Cin, int Cout, int N, int relu, float *y, int pool, hipStream_t st)
{
    dim3 grid(l3d_divup(N, 256), B, l3d_divup(Cout, 32)), block(256);
    hipLaunchKernelGGL((pointwise_conv_stream_kernel<32, POOL>), grid, block, (size_t)Cin * 32 * 4, st, x, w, scale, shift,
                       shift_bstride, Cin, Cout, N, relu, y, pool);
    return l3d_check_launch();
}

// shapes the streaming kernel takes: very narrow and long (memory-bound); the MFMA tile keeps everything else
static bool stream_shape(int x_channel_last, int Cin, int Cout, int N, int pool)
{
    (void)pool;
    if (x_channel_last || N < 4096) return false;
    if (Cout > 8 && Cout <= 32 && Cin <= 128) return true;
    return Cout <= 64 && Cin <= 32 && N >= 16384;               // two passes of 32 channels still beat the tile kernel here
}

template <bool POOL>
static int launch_pointwise_conv(const float *x, int x_channel_last, const float *w, const float *scale,
                                 const float *shift, int shift_bstride, int B, int Cin, int Cout, int N, int relu,
                                 float *y, int pool, hipStream_t st)
{
    dim3 grid(l3d_divup(N, PW_TN), l3d_divup(Cout, PW_TM), B), block(256);
    const bool full = (Cout % PW_TM) == 0 && (N % PW_TN) == 0 && (Cin % PW_TK) == 0 &&
                      ((((size_t)x) | ((size_t)w)) & 15) == 0;
#define L3D_PW(XCL_, FULL_)                                                                                  \
    hipLaunchKernelGGL((pointwise_conv_kernel<XCL_, FULL_, POOL>), grid, block, 0, st, x, w, scale, shift,   \
                       shift_bstride, Cin, Cout, N, relu, y, pool)
    if (x_channel_last) {
        if (full) L3D_PW(true, true); else L3D_PW(true, false);
    } else {
        if (full) L3D_PW(false, true); else L3D_PW(false, false);
    }
#undef L3D_PW
    return l3d_check_launch();
}

// pool = 0: y [B,Cout,N]; pool = 8, 16, 32, 64: the max over every `pool` consecutive points in the epilogue, y [B,Cout,N/pool]
extern "C" int l3d_pointwise_conv(const float *x, int x_channel_last, const float *w,
                                  const float *scale, const float *shift, int shift_bstride, int B,
                                  int Cin, int Cout, int N, int relu, int pool, float *y, l3d_stream_t stream)
{
    L3D_REQUIRE(x && w && y && B > 0 && Cin > 0 && Cout > 0 && N > 0 && pool >= 0);
    if (B > 65535) return L3D_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (pool) {
        if ((pool != 8 && pool != 16 && pool != 32 && pool != 64) || N % pool) return L3D_ERR_UNSUPPORTED;
        if (stream_shape(x_channel_last, Cin, Cout, N, pool))
            return launch_stream<true>(x, w, scale, shift, shift_bstride, B, Cin, Cout, N, relu, y, pool, st);
        return launch_pointwise_conv<true>(x, x_channel_last, w, scale, shift, shift_bstride, B, Cin, Cout, N, relu, y, pool, st);
    }
    if (Cout <= 8 && B <= 65535)
        return x_channel_last ? launch_narrow<true>(x, w, scale, shift, shift_bstride, B, Cin, Cout, N, relu, y, st)
                              : launch_narrow<false>(x, w, scale, shift, shift_bstride, B, Cin, Cout, N, relu, y, st);
    if (stream_shape(x_channel_last, Cin, Cout, N, 0))
        return launch_stream<false>(x, w, scale, shift, shift_bstride, B, Cin, Cout, N, relu, y, 0, st);
    return launch_pointwise_conv<false>(x, x_channel_last, w, scale, shift, shift_bstride, B, Cin, Cout, N, relu, y, 0, st);
}
