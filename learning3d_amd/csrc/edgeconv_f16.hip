// edgeconv_f16.hip -- the register-chained EdgeConv stack of edgeconv_split.hip with layers 2-4 as "f16x2":
// every fp32 operand is carried as an fp16 HIGH part and a 2^12-SCALED fp16 residual, and one fp32 product costs
// THREE fp16 MFMA products (half of bf16x3's six) at fp32-level accuracy.  models/dgcnn.py:32-46.
//
// Arithmetic (the error-corrected fp16 split of Ootomo & Yokota, IJHPCA 2022, adapted to the MFMA):
//   activation x (fp32):  h = f16(x),  m' = f16((x - h) * 2^12)            -> x = h + m' 2^-12 up to 2^-24 |x|
//   weight     w (fp32):  W = w 2^S (S per layer, static, so that max|W| is in [4,8)),
//                         H = f16(W),  M = f16(W - H),  Hs = f16(H 2^-12)  -> W = H + M up to 2^-24 |W|
//   one accumulator:      acc = b 2^S + sum_k ( M h  +  Hs m'  +  H h )     = 2^S (b + w.x) up to the dropped
//                         M m' 2^-12 term (2^-24 relative), every fp16 x fp16 product exact in the fp32 accumulator;
//   epilogue:             y = max(acc, 0) 2^-S  (a power of two: exact).
// Scaling the residual keeps it a NORMAL fp16 number whenever x is one (an unscaled residual of x < 0.25 is
// subnormal and loses the bits it exists to carry); scaling the weights does the same for M.  CPU emulation
// (K = 64..512, activations scaled 1e-3 .. 1e2): max error 0.5-1.1x, rms 0.7-1.2x of the fp32-MFMA kernel's own error
// against fp64 (bf16x3: 0.7-1.1x / 0.7-0.9x); tests/test_gpu_parity.py holds this kernel to the same <= 2x / 1.5x bar.
// Range: fp16 tops out at 65504.  Activations are post-ReLU, so the max-pooled outputs the kernel writes anyway ARE
// the largest activations: each wave tracks their maximum and raises *range_flag if a layer-1..3 output exceeds
// 60000 (results are then invalid; the host falls back to edgeconv_split.hip, whose bf16 planes have fp32's range).
//
// The split is 4 VALU instructions per value pair instead of bf16x3's 10: v_fma_mixlo/hi_f16 convert WITH the
// power-of-two scale, v_fma_mix_f32 forms the residual straight from the packed fp16 halves.
//
// Chaining with v_mfma_f32_16x16x32_f16 is that of edgeconv_split.hip: every layer transposed,
// D[ch][row] = sum_k W'[ch][k] act[row][k], weights = A operand, activations = B operand, one wave
// owns MT row tiles of 16 rows (4 points x 4*MT neighbours); lane (j = row, g) register r holds channel
// 16m + 4g + r of M-tile m; the B operand of k-step s is the pair of previous-layer accumulators (2s, 2s+1) of the
// same lane after ReLU and the split: no LDS, no barriers, no cross-lane traffic.  The A operand is pre-split and
// pre-permuted on the host (l3d_edgeconv_pack, fourth block) and streamed as 1 KB fragments.
#include <type_traits>
#include "common.h"
#include "edgeconv_layout.h"
#include "split_bf16.h"      // f32x2 / f32x4 typedefs

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float ef_quad_max(float v)
{
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    return v;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define EF_BF(u) __builtin_bit_cast(f16x8, (u))
#ifndef EF_VPM
#define EF_VPM 4            // VALU instructions the scheduler may place after each MFMA of a group
#endif

// One output M-tile pair of a dense layer: 2 x S steps of {prefetch fragment step+2, 6 x MT MFMAs}.
// ---------------------------------------------------------------------------------------------
// The finish work of a completed M-tile pair (ReLU, max-pool, three-way split) cut into small units so
// that it can be issued BETWEEN the MFMAs of the next pair: with one wave per SIMD nothing else hides
// VALU work, and a VALU instruction issued in the shadow of a 16-cycle MFMA is free.
//   per row tile t:  [relu+max of h0[t]] [relu+max of h1[t]] ([split h0[t], h1[t]] unless LAST)
//   then [quad max + store of M-tile 2s] [same for 2s+1]
// ---------------------------------------------------------------------------------------------
template <bool LAST> struct EfUnits { static constexpr int PER_T = LAST ? 2 : 3; };

// Home a freshly split fragment word in the accumulation half of the register file: the layer-4
// input planes (240 registers for MT = 5) are only ever read as MFMA B operands, which may be AGPRs;
// left to itself the allocator keeps them in VGPRs, runs out, and reloads spilled words before every use.
__device__ __forceinline__ uint32_t ef_to_agpr(uint32_t v)
{
    uint32_t r;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(r) : "v"(v));
    return r;
}

// v_max_f32 written out: fmaxf() on an MFMA result costs a second, canonicalising v_max x,x,x.  Only
// used where the accumulator was written hundreds of cycles earlier (the pipelined units): the
// compiler does not pad MFMA -> VALU hazards around inline asm.
template <bool RAW> __device__ __forceinline__ float ef_max(float a, float b)
{
    if (!RAW) return fmaxf(a, b);
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// relu'd accumulators (scaled domain: a = 2^S y) of one value pair -> packed fp16 (h, m') words of y = a c:
//   h = f16(a c) (c = 2^-S: the product is exact, one rounding), r = a c - h (exact), m' = f16(r 2^12)
__device__ __forceinline__ void ef_split_pair(float a0, float a1, float c, uint32_t &h, uint32_t &m)
{
    float r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a0), "v"(c));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a0), "v"(c), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(a1), "v"(c), "v"(h));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(m) : "v"(r0), "s"(4096.0f));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(m) : "v"(r1), "s"(4096.0f));
}

// c: 2^-S of the layer whose accumulators h holds (1 for layer 1); ovf: running maximum of the pooled outputs
template <int MT, bool LAST, bool AGPR_OUT, bool RAW, int U>
__device__ __forceinline__ void ef_finish_unit_c(f32x4 (&h)[2][MT], f16x8 (&pl)[2][MT], f32x4 (&mx)[2],
                                                 float *__restrict__ dst, bool writer, float c, float &ovf)
{
    constexpr int PER_T = EfUnits<LAST>::PER_T;
    constexpr int t = U / PER_T, k = U % PER_T;
    if constexpr (t < MT && k < 2) {                         // relu + running max of M-tile k
#pragma unroll
        for (int r = 0; r < 4; r++) {
            h[k][t][r] = ef_max<RAW>(h[k][t][r], 0.f);
            mx[k][r] = t == 0 ? h[k][t][r] : ef_max<RAW>(mx[k][r], h[k][t][r]);
        }
    } else if constexpr (t < MT) {                           // split both M-tiles of row tile t
        // whole 16-byte fragments are written at once: component-wise stores into the plane arrays
        // defeat their promotion to registers (the MFMA reads them back as one 8 x f16 vector)
        uint32_t q[4][2];
        ef_split_pair(h[0][t][0], h[0][t][1], c, q[0][0], q[0][1]);
        ef_split_pair(h[0][t][2], h[0][t][3], c, q[1][0], q[1][1]);
        ef_split_pair(h[1][t][0], h[1][t][1], c, q[2][0], q[2][1]);
        ef_split_pair(h[1][t][2], h[1][t][3], c, q[3][0], q[3][1]);
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const u32x4 v = {AGPR_OUT ? ef_to_agpr(q[0][p]) : q[0][p], AGPR_OUT ? ef_to_agpr(q[1][p]) : q[1][p],
                             AGPR_OUT ? ef_to_agpr(q[2][p]) : q[2][p], AGPR_OUT ? ef_to_agpr(q[3][p]) : q[3][p]};
            pl[p][t] = __builtin_bit_cast(f16x8, v);         // one 16-byte value: the type the MFMA reads
        }
    } else {                                                 // U = MT*PER_T + k, k = 0, 1: pooled store
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = ef_quad_max(mx[k][r]) * c;
        if (!LAST) ovf = fmaxf(fmaxf(ovf, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));    // only split layers matter
        if (writer) *(f32x4 *)(dst + 16 * k) = v;
    }
}
template <int MT, bool LAST> struct EfN { static constexpr int UNITS = MT * EfUnits<LAST>::PER_T + 2; };

// Compile-time loops: every register-array index in this file must be a constant, and `#pragma unroll`
// is only a request (bodies this large exceed the unroller's pragma threshold, the loop stays rolled,
// the index becomes dynamic and the accumulators / planes land in scratch memory).
template <int I0, int I1, class F>
__device__ __forceinline__ void ef_static_for(F &&f)
{
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        ef_static_for<I0 + 1, I1>(f);
    }
}

template <int MT, bool LAST, bool AGPR_OUT = false>
__device__ __forceinline__ void ef_finish_all(f32x4 (&h)[2][MT], f16x8 (&pl)[2][MT], float *__restrict__ dst, bool writer,
                                              float c, float &ovf)
{
    f32x4 mx[2];
    ef_static_for<0, EfN<MT, LAST>::UNITS>([&](auto u) {
        ef_finish_unit_c<MT, LAST, AGPR_OUT, false, decltype(u)::value>(h, pl, mx, dst, writer, c, ovf);
    });
}

// EF_PIN: the compiler's IR passes sink a load towards its first use (two steps later) regardless of
// sched_barrier, which collapses the prefetch distance; a memory clobber right after the issue pins it
// (the fragment pointer is deliberately NOT __restrict__, or the clobber would not order the load).
#define EF_PIN() asm volatile("" ::: "memory")

// One output M-tile pair of a dense layer: 2 x S steps (k-step outer, M-tile inner -- the order the
// fragments are packed in) of {prefetch fragment step+2, 3 groups of MT MFMAs}, with the finish units
// of a PREVIOUS pair (hp, if HAS_PREV) spread over the first NGU groups.  That previous pair is the
// preceding pair of this layer (NGU = all groups) or, for a layer's first pair, the LAST pair of the
// previous layer, whose planes are this layer's k-step S-1: its units then ride on the groups of
// k-steps 0 .. S-2 only (NGU = (S-1)*6) and are complete before the first MFMA that reads them.
// mp = this pair, mp_next = the pair executed after it (fragment and bias prefetches cross the pair
// boundary); pairs may be executed in any order.  bv: this pair's bias, loaded during the previous
// pair; replaced by the next pair's on return.  c_prev: 2^-S of the layer the previous pair belongs to.
template <int MT, int S, bool HAS_PREV, bool PREV_LAST, bool PREV_AGPR, int NGU>
__device__ __forceinline__ void ef_pair(int mp, int mp_next, const f16x8 (&pin)[S][2][MT], const f16x8 (&pin_last)[2][MT],
                                        const uint4 *wp,
                                        uint4 (&a0)[3], uint4 (&a1)[3], f32x4 (&bv)[2], const float *__restrict__ bias,
                                        f32x4 (&acc)[2][MT], f32x4 (&hp)[2][MT], f16x8 (&po_prev)[2][MT],
                                        float *__restrict__ dst_prev, bool writer_prev, int g, float c_prev, float &ovf)
{
    constexpr int NU = HAS_PREV ? EfN<MT, PREV_LAST>::UNITS : 0;    // finish units to hide
#pragma unroll
    for (int mm = 0; mm < 2; mm++)
#pragma unroll
        for (int t = 0; t < MT; t++) acc[mm][t] = bv[mm];
    bv[0] = *(const f32x4 *)(bias + 32 * mp_next + 4 * g);
    bv[1] = *(const f32x4 *)(bias + 32 * mp_next + 16 + 4 * g);
    f32x4 mx[2];
    ef_static_for<0, 2 * S>([&](auto rc) {
        // execution step r = 2 s + mm; fragment two steps ahead: inside this pair, or the first two of the next
        constexpr int r = decltype(rc)::value, s = r >> 1, mm = r & 1;
        const int nxt = r + 2 < 2 * S ? mp * 2 * S + r + 2 : mp_next * 2 * S + (r + 2 - 2 * S);
        uint4 a2[3];
#pragma unroll
        for (int p = 0; p < 3; p++) a2[p] = wp[(size_t)(nxt * 3 + p) * 64];
        EF_PIN();
        __builtin_amdgcn_sched_barrier(0);
        // three products, smallest first (M h, Hs m', H h); MT independent accumulators between dependent MFMAs
        ef_static_for<0, 3>([&](auto pc) {
            constexpr int prod = decltype(pc)::value;
            constexpr int pa = prod == 0 ? 2 : (prod == 1 ? 1 : 0);                   // W plane: M  Hs H   (packed H, Hs, M)
            constexpr int pb = prod == 1 ? 1 : 0;                                     // x plane: h  m' h
#pragma unroll
            for (int t = 0; t < MT; t++)
                acc[mm][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(EF_BF(a0[pa]), (s == S - 1 ? pin_last[pb][t] : pin[s][pb][t]),
                                                                    acc[mm][t], 0, 0, 0);
            constexpr int gi = r * 3 + prod;
            if constexpr (HAS_PREV && gi < NGU) {
                ef_static_for<gi * NU / NGU, (gi + 1) * NU / NGU>([&](auto u) {
                    ef_finish_unit_c<MT, PREV_LAST, PREV_AGPR, true, decltype(u)::value>(hp, po_prev, mx, dst_prev, writer_prev,
                                                                                          c_prev, ovf);
                });
                // issue order inside the group: one MFMA, then a few of the unit's VALU instructions
#pragma unroll
                for (int t = 0; t < MT; t++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, EF_VPM, 0);
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
// c_in / c_own: 2^-S of the previous layer (whose last pair is finished here) and of this layer; bias is pre-scaled.
template <int MT, int S, int NPAIR, bool LAST, bool UNROLL, bool IN_AGPR, bool OUT_AGPR>
__device__ __forceinline__ void ef_layer(const f16x8 (&pin)[S][2][MT], f16x8 (&pout)[LAST ? 1 : NPAIR][2][MT],
                                         const uint4 *wl, const float *__restrict__ bias, float *__restrict__ prow,
                                         f32x4 (&accA)[2][MT], f32x4 (&accB)[2][MT], float *__restrict__ dst_in,
                                         int *mp_out, bool writer, int lane, int g, int rot, float c_in, float c_own, float &ovf)
{
    static_assert(NPAIR % 2 == 0 && S >= 2, "pairs are processed two at a time; deferred finish needs S >= 2");
    constexpr int NG = 2 * S * 3, NGD = (S - 1) * 6;
    f16x8 last[2][MT];             // planes of k-step S-1: produced here by the deferred finish (pin[S-1] is never written)
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
    EF_PIN();
    if constexpr (UNROLL) {
        // written out (NPAIR is 2 or 4): a `#pragma unroll` loop over this much code is not always
        // unrolled, and a rolled loop indexes pout dynamically, which sends the planes to scratch
        static_assert(NPAIR == 2 || NPAIR == 4, "unrolled layers have 2 or 4 output pairs");
        ef_pair<MT, S, true, false, IN_AGPR, NGD>(0, 1, pin, last, wp, a0, a1, bv, bias, accA, accB, last, dst_in, writer, g, c_in, ovf);
        ef_pair<MT, S, true, LAST, OUT_AGPR, NG>(1, NPAIR > 2 ? 2 : 1, pin, last, wp, a0, a1, bv, bias, accB, accA, pout[0], prow, writer, g, c_own, ovf);
        if constexpr (NPAIR == 4) {
            ef_pair<MT, S, true, LAST, OUT_AGPR, NG>(2, 3, pin, last, wp, a0, a1, bv, bias, accA, accB, pout[1], prow + 32, writer, g, c_own, ovf);
            ef_pair<MT, S, true, LAST, OUT_AGPR, NG>(3, 3, pin, last, wp, a0, a1, bv, bias, accB, accA, pout[2], prow + 64, writer, g, c_own, ovf);
        }
        *mp_out = NPAIR - 1;
    } else {
        const int q0 = rot % NPAIR, q1 = (1 + rot) % NPAIR, q2 = (2 + rot) % NPAIR;
        ef_pair<MT, S, true, false, IN_AGPR, NGD>(q0, q1, pin, last, wp, a0, a1, bv, bias, accA, accB, last, dst_in, writer, g, c_in, ovf);
        ef_pair<MT, S, true, LAST, OUT_AGPR, NG>(q1, q2, pin, last, wp, a0, a1, bv, bias, accB, accA, pout[0], prow + 32 * q0, writer, g, c_own, ovf);
        int mpB = q1;                                                   // pair whose results sit in accB
#pragma unroll 1
        for (int i = 2; i < NPAIR; i += 2) {
            const int m0 = (i + rot) % NPAIR, m1 = (i + 1 + rot) % NPAIR, m2 = (i + 2 + rot) % NPAIR;
            ef_pair<MT, S, true, LAST, OUT_AGPR, NG>(m0, m1, pin, last, wp, a0, a1, bv, bias, accA, accB, pout[0], prow + 32 * mpB, writer, g, c_own, ovf);
            ef_pair<MT, S, true, LAST, OUT_AGPR, NG>(m1, i + 2 < NPAIR ? m2 : m1, pin, last, wp, a0, a1, bv, bias, accB, accA, pout[0],
                                                    prow + 32 * m0, writer, g, c_own, ovf);
            mpB = m1;
        }
        *mp_out = mpB;
    }
}

template <int MT>
__global__ __launch_bounds__(256, 1) void edgeconv_f16_kernel(const float *__restrict__ xyz,
                                                              const int64_t *__restrict__ idx, int N, int k,
                                                              const float *packed,
                                                              float *__restrict__ pooled /*[B*N][512]*/,
                                                              int *__restrict__ range_flag
#ifdef EF_TIMING
                                                              , unsigned long long *tdbg
#endif
)
{
#ifdef EF_TIMING
    unsigned long long tk[6];
#define EF_T(i) tk[i] = __builtin_amdgcn_s_memtime()
#else
#define EF_T(i)
#endif
    EF_T(0);
    constexpr int CTOT = EC_C1 + EC_C2 + EC_C3 + EC_C4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int n = (blockIdx.x * 4 + wave) * 4 + (j >> 2);          // this lane's point
    const int nc = min(n, N - 1);
    const bool writer = (n < N) && ((j & 3) == 0);
    float *prow = pooled + ((size_t)b * N + nc) * CTOT + 4 * g;
    // 2^-S of layers 2..4 (uniform: scalar loads)
    const float c2 = packed[EC4_OFF_SC], c3 = packed[EC4_OFF_SC + 1], c4 = packed[EC4_OFF_SC + 2];
    float ovf = 0.f;

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
    EF_T(1);
    f16x8 p1[EC_C1 / 32][2][MT];
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
            if (mp + 1 < EC_C1 / 32) ef_finish_all<MT, false>(accB, p1[mp], prow + 32 * mp, writer, 1.0f, ovf);
        }
    }
    int mp_last;

    EF_T(2);
    // ---- layer 2: 64 -> 64   (its first pair hides the finish of layer 1's last pair, and so on down)
    f16x8 p2[EC_C2 / 32][2][MT];
    ef_layer<MT, EC_C1 / 32, EC_C2 / 32, false, true, false, false>(
        p1, p2, (const uint4 *)(packed + EC4_OFF_W2), packed + EC4_OFF_B2, prow + EC_C1, accA, accB,
        prow + 32 * (EC_C1 / 32 - 1), &mp_last, writer, lane, g, 0, 1.0f, c2, ovf);
    EF_T(3);
    // ---- layer 3: 64 -> 128
    f16x8 p3[EC_C3 / 32][2][MT];
    ef_layer<MT, EC_C2 / 32, EC_C3 / 32, false, true, false, (MT > 4)>(
        p2, p3, (const uint4 *)(packed + EC4_OFF_W3), packed + EC4_OFF_B3, prow + EC_C1 + EC_C2, accA, accB,
        prow + EC_C1 + 32 * (EC_C2 / 32 - 1), &mp_last, writer, lane, g, 0, c2, c3, ovf);
    EF_T(4);
    // ---- layer 4: 128 -> 256, only max-pooled
    f16x8 dummy[1][2][MT];
#ifndef EF_ROT
#define EF_ROT 1
#endif
    const int rot = EF_ROT ? (int)(((blockIdx.x + gridDim.x * blockIdx.y) >> 3) % (EC_C4 / 32)) : 0;   // >>3: ids = XCD mod 8
    ef_layer<MT, EC_C3 / 32, EC_C4 / 32, true, false, (MT > 4), false>(
        p3, dummy, (const uint4 *)(packed + EC4_OFF_W4), packed + EC4_OFF_B4, prow + EC_C1 + EC_C2 + EC_C3, accA, accB,
        prow + EC_C1 + EC_C2 + 32 * (EC_C3 / 32 - 1), &mp_last, writer, lane, g, rot, c3, c4, ovf);
    ef_finish_all<MT, true>(accB, dummy[0], prow + EC_C1 + EC_C2 + EC_C3 + 32 * mp_last, writer, c4, ovf);
    EF_T(5);
    // fp16 range guard: ovf = the largest layer-1..3 activation this lane pooled (post-ReLU, so the pooled maxima
    // are the maxima).  Never taken for BatchNorm'd networks; the host re-runs on the bf16x3 kernel if it is.
    if (ovf > 60000.f && range_flag) *(volatile int *)range_flag = 1;     // may live in mapped host memory: plain store
#ifdef EF_TIMING
    if (threadIdx.x == 0)
        for (int i = 0; i < 6; i++) tdbg[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 6 + i] = tk[i];
#endif
}

#ifndef EF_TIMING
extern "C" int l3d_edgeconv_forward_f16(const float *xyz, const int64_t *idx, int B, int N, int k,
                                        const float *packed, float *pooled, int *range_flag, l3d_stream_t stream)
{
    L3D_REQUIRE(xyz && idx && packed && pooled && B > 0 && N > 0 && k > 0);
    if (k > 20 || B > 65535 || (((size_t)packed) & 15)) return L3D_ERR_UNSUPPORTED;
    dim3 grid(l3d_divup(N, 16), B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (k <= 16) hipLaunchKernelGGL(edgeconv_f16_kernel<4>, grid, block, 0, st, xyz, idx, N, k, packed, pooled, range_flag);
    else              hipLaunchKernelGGL(edgeconv_f16_kernel<5>, grid, block, 0, st, xyz, idx, N, k, packed, pooled, range_flag);
    return l3d_check_launch();
}
#endif
