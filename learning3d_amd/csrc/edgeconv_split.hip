// edgeconv_split.hip -- the register-chained EdgeConv stack of edgeconv2.hip with layers 2-4 on the
// bf16 matrix cores ("bf16x3": every fp32 operand split exactly into three bf16 planes, six bf16
// MFMA products per fp32 product, fp32 accumulate -- see conv_split.hip for the error argument).
// models/dgcnn.py:32-46.
//
// Chaining with v_mfma_f32_16x16x32_bf16.  As in edgeconv2.hip every layer is computed transposed,
// D[ch][row] = sum_k W'[ch][k] act[row][k], weights = A operand, activations = B operand, one wave
// owns MT row tiles of 16 rows (4 points x 4*MT neighbours).  The accumulator layout is the same as
// for the fp32 MFMA: lane (j = row, g) register r holds channel 16m + 4g + r of M-tile m.  The bf16
// B operand of lane (j, g) is EIGHT k-slots for row j; k is only a summation index, so k-step s takes
//     slots 0..3 of lane group g  <->  input channel 16(2s)   + 4g + e   (M-tile 2s,   register e)
//     slots 4..7 of lane group g  <->  input channel 16(2s+1) + 4g + e   (M-tile 2s+1, register e)
// i.e. the B operand of k-step s is the pair of previous-layer accumulators (2s, 2s+1) of the same
// lane, after bias (initial accumulator value), ReLU and the exact three-way bf16 split: still no LDS,
// no barriers, no cross-lane traffic.  The A operand is pre-split and pre-permuted on the host
// (l3d_edgeconv_pack, third block) and streamed as 1 KB fragments, consumed strictly linearly.
//
// Loop order is M-tile-pair outer / k-step inner: only two M-tiles of accumulators (2 x MT x 4
// registers) are live, so the dominant register cost is the split input planes (layer 4: 128
// channels x 3 planes = 240 VGPRs for MT = 5) and the kernel fits one wave per SIMD.  Layer 1
// (6 -> 64, K padded to 8) stays on the fp32 MFMA: a K=32 bf16 step would be 3/4 padding.
#include <type_traits>
#include "common.h"
#include "edgeconv_layout.h"
#include "split_bf16.h"

__device__ __forceinline__ float es_quad_max(float v)
{
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    return v;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define ES_BF(u) __builtin_bit_cast(bf16x8, (u))
#ifndef ES_VPM
#define ES_VPM 4            // VALU instructions the scheduler may place after each MFMA of a group
#endif

// One output M-tile pair of a dense layer: 2 x S steps of {prefetch fragment step+2, 6 x MT MFMAs}.
// ---------------------------------------------------------------------------------------------
// The finish work of a completed M-tile pair (ReLU, max-pool, three-way split) cut into small units so
// that it can be issued BETWEEN the MFMAs of the next pair: with one wave per SIMD nothing else hides
// VALU work, and a VALU instruction issued in the shadow of a 16-cycle MFMA is free.
//   per row tile t:  [relu+max of h0[t]] [relu+max of h1[t]] ([split h0[t], h1[t]] unless LAST)
//   then [quad max + store of M-tile 2s] [same for 2s+1]
// ---------------------------------------------------------------------------------------------
template <bool LAST> struct EsUnits { static constexpr int PER_T = LAST ? 2 : 3; };

// Home a freshly split fragment word in the accumulation half of the register file: the layer-4
// input planes (240 registers for MT = 5) are only ever read as MFMA B operands, which may be AGPRs;
// left to itself the allocator keeps them in VGPRs, runs out, and reloads spilled words before every use.
__device__ __forceinline__ uint32_t es_to_agpr(uint32_t v)
{
    uint32_t r;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(r) : "v"(v));
    return r;
}

// v_max_f32 written out: fmaxf() on an MFMA result costs a second, canonicalising v_max x,x,x.  Only
// used where the accumulator was written hundreds of cycles earlier (the pipelined units): the
// compiler does not pad MFMA -> VALU hazards around inline asm.
template <bool RAW> __device__ __forceinline__ float es_max(float a, float b)
{
    if (!RAW) return fmaxf(a, b);
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int MT, bool LAST, bool AGPR_OUT, bool RAW, int U>
__device__ __forceinline__ void es_finish_unit_c(f32x4 (&h)[2][MT], bf16x8 (&pl)[3][MT], f32x4 (&mx)[2],
                                                 float *__restrict__ dst, bool writer)
{
    constexpr int PER_T = EsUnits<LAST>::PER_T;
    constexpr int t = U / PER_T, k = U % PER_T;
    if constexpr (t < MT && k < 2) {                         // relu + running max of M-tile k
#pragma unroll
        for (int r = 0; r < 4; r++) {
            h[k][t][r] = es_max<RAW>(h[k][t][r], 0.f);
            mx[k][r] = t == 0 ? h[k][t][r] : es_max<RAW>(mx[k][r], h[k][t][r]);
        }
    } else if constexpr (t < MT) {                           // split both M-tiles of row tile t
        // whole 16-byte fragments are written at once: component-wise stores into the plane arrays
        // defeat their promotion to registers (the MFMA reads them back as one 8 x bf16 vector)
        uint32_t q[4][3];
        split_pair(h[0][t][0], h[0][t][1], q[0][0], q[0][1], q[0][2]);
        split_pair(h[0][t][2], h[0][t][3], q[1][0], q[1][1], q[1][2]);
        split_pair(h[1][t][0], h[1][t][1], q[2][0], q[2][1], q[2][2]);
        split_pair(h[1][t][2], h[1][t][3], q[3][0], q[3][1], q[3][2]);
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const u32x4 v = {AGPR_OUT ? es_to_agpr(q[0][p]) : q[0][p], AGPR_OUT ? es_to_agpr(q[1][p]) : q[1][p],
                             AGPR_OUT ? es_to_agpr(q[2][p]) : q[2][p], AGPR_OUT ? es_to_agpr(q[3][p]) : q[3][p]};
            pl[p][t] = __builtin_bit_cast(bf16x8, v);        // one 16-byte value: the type the MFMA reads
        }
    } else {                                                 // U = MT*PER_T + k, k = 0, 1: pooled store
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = es_quad_max(mx[k][r]);
        if (writer) *(f32x4 *)(dst + 16 * k) = v;
    }
}
template <int MT, bool LAST> struct EsN { static constexpr int UNITS = MT * EsUnits<LAST>::PER_T + 2; };

// Compile-time loops: every register-array index in this file must be a constant, and `#pragma unroll`
// is only a request (bodies this large exceed the unroller's pragma threshold, the loop stays rolled,
// the index becomes dynamic and the accumulators / planes land in scratch memory).
template <int I0, int I1, class F>
__device__ __forceinline__ void es_static_for(F &&f)
{
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        es_static_for<I0 + 1, I1>(f);
    }
}

template <int MT, bool LAST, bool AGPR_OUT = false>
__device__ __forceinline__ void es_finish_all(f32x4 (&h)[2][MT], bf16x8 (&pl)[3][MT], float *__restrict__ dst, bool writer)
{
    f32x4 mx[2];
    es_static_for<0, EsN<MT, LAST>::UNITS>([&](auto u) {
        es_finish_unit_c<MT, LAST, AGPR_OUT, false, decltype(u)::value>(h, pl, mx, dst, writer);
    });
}

// ES_PIN: the compiler's IR passes sink a load towards its first use (two steps later) regardless of
// sched_barrier, which collapses the prefetch distance; a memory clobber right after the issue pins it
// (the fragment pointer is deliberately NOT __restrict__, or the clobber would not order the load).
#define ES_PIN() asm volatile("" ::: "memory")

// One output M-tile pair of a dense layer: 2 x S steps (k-step outer, M-tile inner -- the order the
// fragments are packed in) of {prefetch fragment step+2, 6 groups of MT MFMAs}, with the finish units
// of a PREVIOUS pair (hp, if HAS_PREV) spread over the first NGU groups.  That previous pair is the
// preceding pair of this layer (NGU = all groups) or, for a layer's first pair, the LAST pair of the
// previous layer, whose planes are this layer's k-step S-1: its units then ride on the groups of
// k-steps 0 .. S-2 only (NGU = (S-1)*12) and are complete before the first MFMA that reads them.
// mp = this pair, mp_next = the pair executed after it (fragment and bias prefetches cross the pair
// boundary); pairs may be executed in any order.  bv: this pair's bias, loaded during the previous
// pair; replaced by the next pair's on return.
template <int MT, int S, bool HAS_PREV, bool PREV_LAST, bool PREV_AGPR, int NGU>
__device__ __forceinline__ void es_pair(int mp, int mp_next, const bf16x8 (&pin)[S][3][MT], const bf16x8 (&pin_last)[3][MT],
                                        const uint4 *wp,
                                        uint4 (&a0)[3], uint4 (&a1)[3], f32x4 (&bv)[2], const float *__restrict__ bias,
                                        f32x4 (&acc)[2][MT], f32x4 (&hp)[2][MT], bf16x8 (&po_prev)[3][MT],
                                        float *__restrict__ dst_prev, bool writer_prev, int g)
{
    constexpr int NU = HAS_PREV ? EsN<MT, PREV_LAST>::UNITS : 0;    // finish units to hide
#pragma unroll
    for (int mm = 0; mm < 2; mm++)
#pragma unroll
        for (int t = 0; t < MT; t++) acc[mm][t] = bv[mm];
    bv[0] = *(const f32x4 *)(bias + 32 * mp_next + 4 * g);
    bv[1] = *(const f32x4 *)(bias + 32 * mp_next + 16 + 4 * g);
    f32x4 mx[2];
    es_static_for<0, 2 * S>([&](auto rc) {
        // execution step r = 2 s + mm; fragment two steps ahead: inside this pair, or the first two of the next
        constexpr int r = decltype(rc)::value, s = r >> 1, mm = r & 1;
        const int nxt = r + 2 < 2 * S ? mp * 2 * S + r + 2 : mp_next * 2 * S + (r + 2 - 2 * S);
        uint4 a2[3];
#pragma unroll
        for (int p = 0; p < 3; p++) a2[p] = wp[(size_t)(nxt * 3 + p) * 64];
        ES_PIN();
        __builtin_amdgcn_sched_barrier(0);
        // six products, smallest first; MT independent accumulators between dependent MFMAs
        es_static_for<0, 6>([&](auto pc) {
            constexpr int prod = decltype(pc)::value;
            constexpr int pa = prod == 0 ? 2 : (prod == 2 || prod == 3) ? 1 : 0;      // W plane: l h m m h h
            constexpr int pb = prod == 1 ? 2 : (prod == 2 || prod == 4) ? 1 : 0;      // x plane: h l m h m h
#pragma unroll
            for (int t = 0; t < MT; t++)
                acc[mm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ES_BF(a0[pa]), (s == S - 1 ? pin_last[pb][t] : pin[s][pb][t]),
                                                                     acc[mm][t], 0, 0, 0);
            constexpr int gi = r * 6 + prod;
            if constexpr (HAS_PREV && gi < NGU) {
                es_static_for<gi * NU / NGU, (gi + 1) * NU / NGU>([&](auto u) {
                    es_finish_unit_c<MT, PREV_LAST, PREV_AGPR, true, decltype(u)::value>(hp, po_prev, mx, dst_prev, writer_prev);
                });
                // issue order inside the group: one MFMA, then a few of the unit's VALU instructions
#pragma unroll
                for (int t = 0; t < MT; t++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, ES_VPM, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int p = 0; p < 3; p++) { a0[p] = a1[p]; a1[p] = a2[p]; }
    });
}

// One dense layer: S input k-steps (32 channels each, planes in pin), NPAIR output M-tile pairs,
// software-pipelined over pairs (accumulators double-buffered: pair i's MFMAs hide pair i-1's finish).
// On entry accB holds the previous layer's last, unfinished pair (its planes are pin[S-1], its pooled
// output goes to dst_in); on return accB holds THIS layer's last unfinished pair (pair index *mp_out).
// wl: [step = (pair*S + s)*2 + mm][plane][lane] fragments, prefetched two steps ahead.  rot (only for
// the rolled LAST layer, where no register array is indexed by the pair): this workgroup starts at pair
// `rot`.  IN_AGPR: pin's planes are homed in AGPRs; OUT_AGPR: this layer's output planes are.
template <int MT, int S, int NPAIR, bool LAST, bool UNROLL, bool IN_AGPR, bool OUT_AGPR>
__device__ __forceinline__ void es_layer(const bf16x8 (&pin)[S][3][MT], bf16x8 (&pout)[LAST ? 1 : NPAIR][3][MT],
                                         const uint4 *wl, const float *__restrict__ bias, float *__restrict__ prow,
                                         f32x4 (&accA)[2][MT], f32x4 (&accB)[2][MT], float *__restrict__ dst_in,
                                         int *mp_out, bool writer, int lane, int g, int rot)
{
    static_assert(NPAIR % 2 == 0 && S >= 2, "pairs are processed two at a time; deferred finish needs S >= 2");
    constexpr int NG = 2 * S * 6, NGD = (S - 1) * 12;
    bf16x8 last[3][MT];             // planes of k-step S-1: produced here by the deferred finish (pin[S-1] is never written)
    const uint4 *wp = wl + lane;
    const int first = UNROLL ? 0 : rot;
    uint4 a0[3], a1[3];
    f32x4 bv[2];
#pragma unroll
    for (int p = 0; p < 3; p++) {
        a0[p] = wp[(size_t)((first * 2 * S) * 3 + p) * 64];
        a1[p] = wp[(size_t)((first * 2 * S + 1) * 3 + p) * 64];
    }
    bv[0] = *(const f32x4 *)(bias + 32 * first + 4 * g);
    bv[1] = *(const f32x4 *)(bias + 32 * first + 16 + 4 * g);
    ES_PIN();
    if constexpr (UNROLL) {
        // written out (NPAIR is 2 or 4): a `#pragma unroll` loop over this much code is not always
        // unrolled, and a rolled loop indexes pout dynamically, which sends the planes to scratch
        static_assert(NPAIR == 2 || NPAIR == 4, "unrolled layers have 2 or 4 output pairs");
        es_pair<MT, S, true, false, IN_AGPR, NGD>(0, 1, pin, last, wp, a0, a1, bv, bias, accA, accB, last, dst_in, writer, g);
        es_pair<MT, S, true, LAST, OUT_AGPR, NG>(1, NPAIR > 2 ? 2 : 1, pin, last, wp, a0, a1, bv, bias, accB, accA, pout[0], prow, writer, g);
        if constexpr (NPAIR == 4) {
            es_pair<MT, S, true, LAST, OUT_AGPR, NG>(2, 3, pin, last, wp, a0, a1, bv, bias, accA, accB, pout[1], prow + 32, writer, g);
            es_pair<MT, S, true, LAST, OUT_AGPR, NG>(3, 3, pin, last, wp, a0, a1, bv, bias, accB, accA, pout[2], prow + 64, writer, g);
        }
        *mp_out = NPAIR - 1;
    } else {
        const int q0 = rot % NPAIR, q1 = (1 + rot) % NPAIR, q2 = (2 + rot) % NPAIR;
        es_pair<MT, S, true, false, IN_AGPR, NGD>(q0, q1, pin, last, wp, a0, a1, bv, bias, accA, accB, last, dst_in, writer, g);
        es_pair<MT, S, true, LAST, OUT_AGPR, NG>(q1, q2, pin, last, wp, a0, a1, bv, bias, accB, accA, pout[0], prow + 32 * q0, writer, g);
        int mpB = q1;                                                   // pair whose results sit in accB
#pragma unroll 1
        for (int i = 2; i < NPAIR; i += 2) {
            const int m0 = (i + rot) % NPAIR, m1 = (i + 1 + rot) % NPAIR, m2 = (i + 2 + rot) % NPAIR;
            es_pair<MT, S, true, LAST, OUT_AGPR, NG>(m0, m1, pin, last, wp, a0, a1, bv, bias, accA, accB, pout[0], prow + 32 * mpB, writer, g);
            es_pair<MT, S, true, LAST, OUT_AGPR, NG>(m1, i + 2 < NPAIR ? m2 : m1, pin, last, wp, a0, a1, bv, bias, accB, accA, pout[0],
                                                    prow + 32 * m0, writer, g);
            mpB = m1;
        }
        *mp_out = mpB;
    }
}

template <int MT>
__global__ __launch_bounds__(256, 1) void edgeconv_split_kernel(const float *__restrict__ xyz,
                                                                const int64_t *__restrict__ idx, int N, int k,
                                                                const float *packed,
                                                                float *__restrict__ pooled /*[B*N][512]*/
#ifdef ES_TIMING
                                                                , unsigned long long *tdbg
#endif
)
{
#ifdef ES_TIMING
    unsigned long long tk[6];
#define ES_T(i) tk[i] = __builtin_amdgcn_s_memtime()
#else
#define ES_T(i)
#endif
    ES_T(0);
    constexpr int CTOT = EC_C1 + EC_C2 + EC_C3 + EC_C4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int n = (blockIdx.x * 4 + wave) * 4 + (j >> 2);          // this lane's point
    const int nc = min(n, N - 1);
    const bool writer = (n < N) && ((j & 3) == 0);
    float *prow = pooled + ((size_t)b * N + nc) * CTOT + 4 * g;

    // ---- layer 1 on the fp32 MFMA (as edgeconv2.hip): graph feature rows as B operands, k-step s,
    //      lane group g -> channel 4s + g of (neighbour xyz, centre xyz, 0, 0)          dgcnn.py:32
    const float *pc = xyz + ((size_t)b * N + nc) * 3;
    const float cx = pc[0], cy = pc[1], cz = pc[2];
    float b1[MT][2];
#pragma unroll
    for (int t = 0; t < MT; t++) {
        const int jj = 4 * t + (j & 3);
        const int64_t nb = idx[((size_t)b * N + nc) * k + (jj < k ? jj : 0)];   // pad k up to 4*MT with a duplicate
        const float *pn = xyz + ((size_t)b * N + nb) * 3;
        const float nx = pn[0], ny = pn[1], nz = pn[2];
        b1[t][0] = g == 0 ? nx : (g == 1 ? ny : (g == 2 ? nz : cx));
        b1[t][1] = g == 0 ? cy : (g == 1 ? cz : 0.f);
    }
    ES_T(1);
    bf16x8 p1[EC_C1 / 32][3][MT];
    f32x4 accA[2][MT], accB[2][MT];                    // accB: the pair whose finish is pending
    {
        const f32x2 *w1 = (const f32x2 *)(packed + EC2_OFF_W1);
#pragma unroll
        for (int mp = 0; mp < EC_C1 / 32; mp++) {
#pragma unroll
            for (int mm = 0; mm < 2; mm++) {
                const int m = 2 * mp + mm;
                const f32x4 bv = *(const f32x4 *)(packed + EC_OFF_B1 + 16 * m + 4 * g);
                const f32x2 a = w1[m * 64 + lane];
#pragma unroll
                for (int t = 0; t < MT; t++) accB[mm][t] = bv;
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int t = 0; t < MT; t++)
                        accB[mm][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[t][s], accB[mm][t], 0, 0, 0);
            }
            if (mp + 1 < EC_C1 / 32) es_finish_all<MT, false>(accB, p1[mp], prow + 32 * mp, writer);
        }
    }
    int mp_last;

    ES_T(2);
    // ---- layer 2: 64 -> 64   (its first pair hides the finish of layer 1's last pair, and so on down)
    bf16x8 p2[EC_C2 / 32][3][MT];
    es_layer<MT, EC_C1 / 32, EC_C2 / 32, false, true, false, false>(
        p1, p2, (const uint4 *)(packed + EC3_OFF_W2), packed + EC_OFF_B2, prow + EC_C1, accA, accB,
        prow + 32 * (EC_C1 / 32 - 1), &mp_last, writer, lane, g, 0);
    ES_T(3);
    // ---- layer 3: 64 -> 128
    bf16x8 p3[EC_C3 / 32][3][MT];
    es_layer<MT, EC_C2 / 32, EC_C3 / 32, false, true, false, (MT > 4)>(
        p2, p3, (const uint4 *)(packed + EC3_OFF_W3), packed + EC_OFF_B3, prow + EC_C1 + EC_C2, accA, accB,
        prow + EC_C1 + 32 * (EC_C2 / 32 - 1), &mp_last, writer, lane, g, 0);
    ES_T(4);
    // ---- layer 4: 128 -> 256, only max-pooled
    bf16x8 dummy[1][3][MT];
#ifndef ES_ROT
#define ES_ROT 1
#endif
    const int rot = ES_ROT ? (int)(((blockIdx.x + gridDim.x * blockIdx.y) >> 3) % (EC_C4 / 32)) : 0;   // >>3: ids = XCD mod 8
    es_layer<MT, EC_C3 / 32, EC_C4 / 32, true, false, (MT > 4), false>(
        p3, dummy, (const uint4 *)(packed + EC3_OFF_W4), packed + EC_OFF_B4, prow + EC_C1 + EC_C2 + EC_C3, accA, accB,
        prow + EC_C1 + EC_C2 + 32 * (EC_C3 / 32 - 1), &mp_last, writer, lane, g, rot);
    es_finish_all<MT, true>(accB, dummy[0], prow + EC_C1 + EC_C2 + EC_C3 + 32 * mp_last, writer);
    ES_T(5);
#ifdef ES_TIMING
    if (threadIdx.x == 0)
        for (int i = 0; i < 6; i++) tdbg[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 6 + i] = tk[i];
#endif
}

#ifndef ES_TIMING
extern "C" int l3d_edgeconv_forward_split(const float *xyz, const int64_t *idx, int B, int N, int k,
                                          const float *packed, float *pooled, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && idx && packed && pooled && B > 0 && N > 0 && k > 0);
    if (k > 20 || B > 65535 || (((size_t)packed) & 15)) return L3D_ERR_UNSUPPORTED;
    dim3 grid(l3d_divup(N, 16), B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (k <= 16) hipLaunchKernelGGL(edgeconv_split_kernel<4>, grid, block, 0, st, xyz, idx, N, k, packed, pooled);
    else              hipLaunchKernelGGL(edgeconv_split_kernel<5>, grid, block, 0, st, xyz, idx, N, k, packed, pooled);
    return l3d_check_launch();
}
#endif
