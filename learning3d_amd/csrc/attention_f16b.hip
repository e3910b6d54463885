// attention_f16b.hip -- flash-style multi-head attention (utils/transformer.py:17-25, MultiHeadedAttention :120-147) as f16x2 on
// the fp16 matrix cores, restructured around ONE barrier per key tile.  Same entry-point contract as attention_f16.hip
// (l3d_attention_forward_f16b: channel-first fp32 q, k, v with batch strides, the three operand maxima in a 16-byte
// workspace, context as fp32 [B, H D, N] and / or as the fp16 activation image of conv_f16.hip); attention_f16.hip's kernel took
// 352 us at DCP's shape (B 32, H 4, D 128, N = M = 1024) against 82 us of matrix-pipe time, 213 us of it in the loop's structure
// (16 barriers per key tile with 12 MFMAs behind each, Q re-staged for every key tile, LABLOG R2.4e).
//
// Arithmetic: every operand is carried as two fp16 planes with an UNSCALED residual (x 2^T -> h = f16(X), m = f16(X - h); T from
// the tensor's maximum so that it sits in [2^11, 2^12): a subnormal residual costs 2^-25 absolute in plane units), three
// products per fp32 product: M h + H m + H h.  Probabilities are carried times 2^12 (folded into the exponent of exp2).
//
// Workgroup = 256 queries of one (cloud, head) = 8 waves x 32 queries, two waves per SIMD.  v_mfma_f32_32x32x16_f16 throughout:
//   S^T[key][query] = K[key][ch] Q^T[ch][query]   A = K cells from LDS, B = this wave's Q fragments (registers, loaded once)
//   O^T[d][query]  += V^T[d][key] P[key][query]    A = V^T cells from LDS, B = the probabilities, straight from S^T's accumulators
// A lane of the 32x32 accumulator holds 16 rows of ONE column (query): row max / sum are in-lane reductions plus one exchange
// with lane ^ 32.  MFMA row i of S^T carries key pi(i) = 16 (a >> 1) + 8 h + 4 (a & 1) + b  (i = 8 a + 4 h + b), so that
// accumulator registers 8 s .. 8 s + 7 of half h ARE the B fragment (keys 16 s + 8 h .. + 7) of PV k-step s: no shuffle, no LDS
// round trip for P.  K is [B, C, M] channel-first: a thread stages (key, 8 channels) -> one 16-byte cell per plane; V is already
// "V^T" in that layout: a thread stages (channel, 8 consecutive keys).  Key tiles of 32 in a ring of three 32 KB stages (D = 128):
// the next tile's global loads are issued at the top of an iteration, split and written to LDS behind its first MFMA phase, one
// __syncthreads per tile; the two waves of a SIMD run half a tile apart (see the main loop).  The running maximum rarely moves after the first tiles: O is rescaled only when some lane's maximum did.
#include "common.h"
#include "split_bf16.h"          // f32x4 / f32x16 typedefs

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define AB_TQ 256
#define AB_TK 32
#define AB_NEG (-1.0e30f)

__device__ __forceinline__ int ab_exponent(float mx, int hi)
{
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);                 // mx = f 2^e, f in [0.5, 1)
    return hi - e;
}

// (a0, a1) c -> packed fp16 (h, m) with m the UNSCALED residual: h = f16(a c), m = f16(a c - h)
__device__ __forceinline__ void ab_split(float a0, float a1, float c, uint32_t &h, uint32_t &m)
{
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const float x0 = a0 * c, x1 = a1 * c;
    const f16x2 hh = {(_Float16)x0, (_Float16)x1};
    const f16x2 mm = {(_Float16)(x0 - (float)hh[0]), (_Float16)(x1 - (float)hh[1])};
    h = __builtin_bit_cast(uint32_t, hh);
    m = __builtin_bit_cast(uint32_t, mm);
}

// (a0, a1) c -> packed fp16 (h, m rs): rs = 2^12 is ab_split_scaled, rs = 1 ab_split
__device__ __forceinline__ void ab_split_rs(float a0, float a1, float c, float rs, uint32_t &h, uint32_t &m)
{
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const float x0 = a0 * c, x1 = a1 * c;
    const f16x2 hh = {(_Float16)x0, (_Float16)x1};
    const f16x2 mm = {(_Float16)((x0 - (float)hh[0]) * rs), (_Float16)((x1 - (float)hh[1]) * rs)};
    h = __builtin_bit_cast(uint32_t, hh);
    m = __builtin_bit_cast(uint32_t, mm);
}

template <int ND>
__global__ __launch_bounds__(512) void attention_f16b_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                             const float *__restrict__ v, int H, int N, int M, float scale,
                                                             float *__restrict__ ctx, long q_bs, long k_bs, long v_bs,
                                                             const unsigned *__restrict__ amax, uint2 *__restrict__ cph,
                                                             uint2 *__restrict__ cpm, float *__restrict__ cinv, float img_rs)
{
    constexpr int D = 32 * ND, KS = D / 16;                       // channels per head, k-steps of the S^T product
    constexpr int KPL = KS * 2 * 32 * 16;                         // bytes of one K plane of a tile: [k-step][g][key][16 B]
    // bytes of one V plane of a tile: [PV k-step][g][d][16 B], each of the four (k-step, g) blocks followed by 32 bytes: a staging
    // thread is (key octet t & 3, channel t >> 2), so four consecutive lanes write the four blocks' cells of one channel -- D 16 =
    // 2048 bytes apart they shared their banks (4-way conflict: 23 % of this kernel's LDS cycles, profiles/round4_pmc_attention.txt)
    constexpr int VBLK = D * 16 + 32, VPL = 2 * 2 * VBLK;
    constexpr int STAGE = 2 * KPL + 2 * VPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char ab_lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 31, g = lane >> 5;
    // Workgroup order (1-D grid).  Workgroup L runs on XCD L % 8, each with its own L2: the query blocks of one (cloud,
    // head) take consecutive slots of ONE XCD, so that head's K / V come from HBM once instead of once per XCD hosting
    // one of its query blocks (the plain 3-D grid moved 671 MB for 268 MB of operands, profiles/round4_pmc_attention.txt).
    int b, h, i0, nB;
    {
        const int nq = (N + AB_TQ - 1) / AB_TQ, L = blockIdx.x;
        int grp, qb_;
        if (((gridDim.x / nq) & 7) == 0) {
            const int xcd = L & 7, slot = L >> 3;
            qb_ = slot % nq;
            grp = (slot / nq) * 8 + xcd;
        } else {
            qb_ = L % nq;
            grp = L / nq;
        }
        b = grp / H; h = grp - b * H; i0 = qb_ * AB_TQ; nB = gridDim.x / nq / H;
    }
    const float *qb = q + (size_t)b * q_bs + (size_t)h * D * N;
    const float *kb = k + (size_t)b * k_bs + (size_t)h * D * M;
    const float *vb = v + (size_t)b * v_bs + (size_t)h * D * M;

    const int Tq = ab_exponent(__uint_as_float(amax[0]), 12), Tk = ab_exponent(__uint_as_float(amax[1]), 12),
              Tv = ab_exponent(__uint_as_float(amax[2]), 12);
    const float cq = ldexpf(1.f, Tq), ck = ldexpf(1.f, Tk), cv = ldexpf(1.f, Tv);
    const float sc2 = ldexpf(scale * 1.4426950408889634f, -(Tq + Tk));     // accumulator -> log2 domain

    // ---- this wave's Q fragments: query i0 + 32 wave + col, k-group g: channels 16 ks + 8 g .. + 7, both planes
    f16x8 Qh[KS], Qm[KS];
    {
        const int qi = min(i0 + wave * 32 + col, N - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            u32x4 hv, mv;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float a0 = qb[(size_t)(16 * ks + 8 * g + 2 * e) * N + qi], a1 = qb[(size_t)(16 * ks + 8 * g + 2 * e + 1) * N + qi];
                uint32_t hh, mm;
                ab_split(a0, a1, cq, hh, mm);
                hv[e] = hh;
                mv[e] = mm;
            }
            Qh[ks] = __builtin_bit_cast(f16x8, hv);
            Qm[ks] = __builtin_bit_cast(f16x8, mv);
        }
    }

    // ---- staging roles: K item = (key = t & 31, octet = t >> 5), V item = (key octet = t & 3, channel = t >> 2)
    const int k_key = t & 31, k_oct = t >> 5;
    const bool k_item = k_oct < D / 8;
    const int v_oct = t & 3, v_d = t >> 2;
    const bool v_item = v_d < D;
    const bool v_vec8 = (M & 7) == 0 && (v_bs & 3) == 0 && ((((size_t)v) & 15) == 0);     // every 8-key run of a V row: two aligned 16-byte loads
    float kr[8], vr[8];
    // Ablation builds of tools/probe_attention_f16b.hip (timing only, results are garbage): AB_NOLOAD skips the tiles' global loads,
    // AB_NOSTAGE also their splits and LDS writes, AB_NOSOFTMAX the exp2 / split work on the probabilities, AB_NOLDSREAD reads
    // one operand cell per phase instead of all of them.
    auto load_tile = [&](int key0) {
#if defined(AB_NOLOAD) || defined(AB_NOSTAGE)
        return;
#endif
        // Nothing here is conditional per lane: a conditional load is an exec-masked branch with a wait at its join, and the tile's
        // loads then queue up one round trip behind the other instead of flying together under the previous tile's MFMAs (the form
        // this replaced: `key < M ? src[..] : 0` and a per-lane choice between the 16-byte and the scalar V loads).  Indices beyond
        // the operand are clamped to valid ones; on the last, partial tile (a wave-uniform branch) the values are zeroed by a
        // 0 / 1 factor.  Threads without an item (D < 128) load their clamped neighbour's and store nothing.
        {
            const int key = key0 + k_key, kc = min(key, M - 1);
            const float *src = kb + (size_t)(8 * min(k_oct, D / 8 - 1)) * M + kc;
#pragma unroll
            for (int e = 0; e < 8; e++) kr[e] = src[(size_t)e * M];
            if (key0 + AB_TK > M) {
                const float in = key < M ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) kr[e] *= in;
            }
        }
        {
            const int key = key0 + 8 * v_oct;
            const float *row = vb + (size_t)min(v_d, D - 1) * M;
            if (v_vec8) {                                         // uniform: 16-byte aligned rows and M % 8 == 0
                const float *src = row + min(key, M - 8);
                const f32x4 x0 = *(const f32x4 *)src, x1 = *(const f32x4 *)(src + 4);
                vr[0] = x0[0]; vr[1] = x0[1]; vr[2] = x0[2]; vr[3] = x0[3];
                vr[4] = x1[0]; vr[5] = x1[1]; vr[6] = x1[2]; vr[7] = x1[3];
                if (key0 + AB_TK > M) {
                    const float in = key < M ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; e++) vr[e] *= in;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) vr[e] = row[min(key + e, M - 1)];
                if (key0 + AB_TK > M) {
#pragma unroll
                    for (int e = 0; e < 8; e++) vr[e] *= key + e < M ? 1.f : 0.f;
                }
            }
        }
    };
    auto store_tile = [&](int stage) {
#ifdef AB_NOSTAGE
        return;
#endif
        unsigned char *base = ab_lds + stage * STAGE;
        if (k_item) {
            u32x4 hv, mv;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                uint32_t hh, mm;
                ab_split(kr[2 * e], kr[2 * e + 1], ck, hh, mm);
                hv[e] = hh;
                mv[e] = mm;
            }
            const int off = (((k_oct >> 1) * 2 + (k_oct & 1)) * 32 + k_key) * 16;
            *(u32x4 *)(base + off) = hv;
            *(u32x4 *)(base + KPL + off) = mv;
        }
        if (v_item) {
            u32x4 hv, mv;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                uint32_t hh, mm;
                ab_split(vr[2 * e], vr[2 * e + 1], cv, hh, mm);
                hv[e] = hh;
                mv[e] = mm;
            }
            const int off = ((v_oct >> 1) * 2 + (v_oct & 1)) * VBLK + v_d * 16;
            *(u32x4 *)(base + 2 * KPL + off) = hv;
            *(u32x4 *)(base + 2 * KPL + VPL + off) = mv;
        }
    };

    f32x16 O[ND];
#pragma unroll
    for (int dt = 0; dt < ND; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) O[dt][r] = 0.f;
    float m_run = AB_NEG, l_run = 0.f;

    // MFMA row `col` of S^T carries key pi(col)
    const int a_ = col >> 3, hb_ = (col >> 2) & 1, b_ = col & 3;
    const int pik = 16 * (a_ >> 1) + 8 * hb_ + 4 * (a_ & 1) + b_;
    const int ka_off = (g * 32 + pik) * 16;                       // + ks * 1024 (+ KPL for the m plane)
    const int va_off = 2 * KPL + g * VBLK + col * 16;             // + s * 2 VBLK + dt * 512 (+ VPL for the m plane)

    const int ntile = (M + AB_TK - 1) / AB_TK;

    // ---- the three phases of a key tile
    // S^T = K Q^T (rows = keys, permuted; column = this lane's query): 3 KS MFMAs
    // Consecutive MFMAs never share an accumulator: a dependent 32x32x16 MFMA waits for its predecessor's result (the bare loop with
    // one S accumulator and O[dt] updated three times in a row ran at 0.6 of the matrix-pipe rate, tools/probe_attention_f16b.hip):
    // S is accumulated in two halves (even / odd k-steps), the PV products walk the d-tiles innermost.
    auto score = [&](f32x16 &S, int stage) {
        const unsigned char *base = ab_lds + stage * STAGE;
        f32x16 S1;
#pragma unroll
        for (int r = 0; r < 16; r++) { S[r] = 0.f; S1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
#ifdef AB_NOLDSREAD
            const f16x8 Kh = Qm[ks], Km = Qh[ks], Kh1 = Qm[ks + 1], Km1 = Qh[ks + 1];
#else
            const f16x8 Kh = *(const f16x8 *)(base + ka_off + ks * 1024);
            const f16x8 Km = *(const f16x8 *)(base + KPL + ka_off + ks * 1024);
            const f16x8 Kh1 = *(const f16x8 *)(base + ka_off + (ks + 1) * 1024);
            const f16x8 Km1 = *(const f16x8 *)(base + KPL + ka_off + (ks + 1) * 1024);
#endif
            S = __builtin_amdgcn_mfma_f32_32x32x16_f16(Km, Qh[ks], S, 0, 0, 0);
            S1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Km1, Qh[ks + 1], S1, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_32x32x16_f16(Kh, Qm[ks], S, 0, 0, 0);
            S1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Kh1, Qm[ks + 1], S1, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_32x32x16_f16(Kh, Qh[ks], S, 0, 0, 0);
            S1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Kh1, Qh[ks + 1], S1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) S[r] += S1[r];
    };
    // online softmax in the log2 domain on the raw accumulators (sc2 > 0: the maximum commutes with the scale); accumulator
    // register r = 4 a + b of half g is key 16 (a >> 1) + 8 g + 4 (a & 1) + b.  Leaves the probabilities times 2^12 as the PV
    // product's B fragments.
    auto softmax = [&](f32x16 &S, int kt, f16x8 (&Ph)[2], f16x8 (&Pm)[2]) {
        const int key0 = kt * AB_TK;
        if (key0 + AB_TK > M) {                                   // the last, ragged tile: keys past M never win and weigh nothing
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int key = key0 + 16 * (r >> 3) + 8 * g + 4 * ((r >> 2) & 1) + (r & 3);
                S[r] = key < M ? S[r] : AB_NEG / sc2;
            }
        }
        float mx = fmaxf(fmaxf(S[0], S[1]), fmaxf(S[2], S[3]));
#pragma unroll
        for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(S[r], S[r + 1]), fmaxf(S[r + 2], S[r + 3])));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc2;
        const float m_new = fmaxf(m_run, mx);
        if (__any(m_new > m_run)) {                               // some query's running maximum moved: rescale what was accumulated
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < ND; dt++)
#pragma unroll
                for (int r = 0; r < 16; r++) O[dt][r] *= alpha;
            m_run = m_new;
        }
        const float shift = 12.0f - m_run;                        // p 2^12 = exp2(S sc2 - m + 12): one fma + v_exp_f32 per value
        float lsum = 0.f;
#ifdef AB_NOSOFTMAX
        for (int s2 = 0; s2 < 2; s2++) {
            u32x4 hv, mv;
            for (int e = 0; e < 4; e++) { hv[e] = __float_as_uint(S[8 * s2 + 2 * e]); mv[e] = __float_as_uint(S[8 * s2 + 2 * e + 1]); lsum += S[8 * s2 + e]; }
            Ph[s2] = __builtin_bit_cast(f16x8, hv);
            Pm[s2] = __builtin_bit_cast(f16x8, mv);
        }
        l_run += lsum + shift;
        return;
#endif
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            u32x4 hv, mv;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float p0 = __builtin_amdgcn_exp2f(fmaf(S[8 * s2 + 2 * e], sc2, shift));
                const float p1 = __builtin_amdgcn_exp2f(fmaf(S[8 * s2 + 2 * e + 1], sc2, shift));
                lsum += p0 + p1;
                uint32_t hh, mm;
                ab_split(p0, p1, 1.0f, hh, mm);
                hv[e] = hh;
                mv[e] = mm;
            }
            Ph[s2] = __builtin_bit_cast(f16x8, hv);
            Pm[s2] = __builtin_bit_cast(f16x8, mv);
        }
        l_run += lsum;
    };
    // O^T += V^T P: 6 ND MFMAs
    auto pv = [&](int stage, const f16x8 (&Ph)[2], const f16x8 (&Pm)[2]) {
        const unsigned char *base = ab_lds + stage * STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            f16x8 Vh[ND], Vm[ND];
#pragma unroll
            for (int dt = 0; dt < ND; dt++) {
#ifdef AB_NOLDSREAD
                Vh[dt] = Qh[dt]; Vm[dt] = Qm[dt];
#else
                Vh[dt] = *(const f16x8 *)(base + va_off + s2 * (2 * VBLK) + dt * 512);
                Vm[dt] = *(const f16x8 *)(base + VPL + va_off + s2 * (2 * VBLK) + dt * 512);
#endif
            }
#pragma unroll
            for (int dt = 0; dt < ND; dt++) O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Vm[dt], Ph[s2], O[dt], 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < ND; dt++) O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Vh[dt], Pm[s2], O[dt], 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < ND; dt++) O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Vh[dt], Ph[s2], O[dt], 0, 0, 0);
        }
    };

    // ---- main loop.  A SIMD hosts wave w and wave w + 4.  Left to themselves both would run the same phase at the same time
    // (one barrier per tile keeps them in step): matrix-pipe phases collide, VALU phases collide.  Waves 4-7 therefore run half a
    // tile behind: waves 0-3  [score(kt)  ][softmax(kt)  ][pv(kt)   ]
    //              waves 4-7  [softmax(kt-1)][pv(kt-1)   ][score(kt)]      -- MFMA under VALU twice, MFMA beside MFMA once:
    // the matrix pipe sees an unbroken stream.  Tile kt's V is then read one iteration after its K: a ring of THREE stages
    // (tile kt+1 is written while tiles kt and kt-1 are live).
    const bool lag = wave >= 4;
    f32x16 S;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int st_cur = 0, st_prev = 2, st_next = 1;                     // stages of tiles kt, kt-1, kt+1
    for (int kt = 0; kt < ntile; kt++) {
        const bool more = kt + 1 < ntile;
        if (more) load_tile((kt + 1) * AB_TK);                    // global loads in flight behind the first phase
        f16x8 Ph[2], Pm[2];
        if (!lag) {
            score(S, st_cur);
            if (more) store_tile(st_next);
            softmax(S, kt, Ph, Pm);
            pv(st_cur, Ph, Pm);
        } else {
            if (kt > 0) {
                softmax(S, kt - 1, Ph, Pm);
                pv(st_prev, Ph, Pm);
            }
            if (more) store_tile(st_next);
            score(S, st_cur);
        }
#ifndef AB_NOBARRIER
        __syncthreads();                                          // tile kt+1 is in place; tile kt-1's stage is free
#endif
        const int tmp = st_prev;
        st_prev = st_cur;
        st_cur = st_next;
        st_next = tmp;
    }
    if (lag) {                                                    // the trailing half tile of waves 4-7
        f16x8 Ph[2], Pm[2];
        softmax(S, ntile - 1, Ph, Pm);
        pv(st_prev, Ph, Pm);
    }

    // ---- epilogue: O^T[d = 32 dt + 8 a + 4 g + b][query]; l in units of 2^12 like P, V in units of 2^Tv
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = ldexpf(1.f / l_tot, -Tv);
    const int i = i0 + wave * 32 + col;
    if (i < N && ctx) {
        float *cb = ctx + ((size_t)b * H + h) * D * N + i;
#pragma unroll
        for (int dt = 0; dt < ND; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * g;
                cb[(size_t)d * N] = O[dt][r] * inv;
            }
    }
    // The context as the fp16 plane image conv_f16.hip consumes ([H D / 8][B N][8], h | m' 2^12 of ctx 2^T): a context vector is a
    // convex combination of value vectors, so |ctx| <= max|v| and T = Tv puts it below 2^12.  A lane holds 4 consecutive channels
    // of an octet (its partner lane ^ 32 the other 4): each writes its 8-byte half of the 16-byte cell.
    if (cph) {
        if (blockIdx.x == 0 && t == 0) *cinv = ldexpf(1.f, -Tv);
        if (i < N) {
            const float invp = 1.f / l_tot;                       // O / l = ctx 2^Tv: already in plane units
            const size_t rows = (size_t)nB * N, row = (size_t)b * N + i;
#pragma unroll
            for (int dt = 0; dt < ND; dt++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int oc = (h * D + 32 * dt + 8 * gq) >> 3;
                    uint32_t h0, h1, m0, m1;
                    // rs = 2^12: m' = f16((X - h) 2^12), conv_f16.hip's three-plane convention; rs = 1: the unscaled residual its two-plane
                    // form reads.  One code path with the factor as data (a run-time branch between ab_split and ab_split_scaled
                    // produced an image that was one fp16 ulp off in 5 of 262144 values -- not run to ground)
                    ab_split_rs(O[dt][4 * gq], O[dt][4 * gq + 1], invp, img_rs, h0, m0);
                    ab_split_rs(O[dt][4 * gq + 2], O[dt][4 * gq + 3], invp, img_rs, h1, m1);
                    cph[((size_t)oc * rows + row) * 2 + g] = make_uint2(h0, h1);
                    cpm[((size_t)oc * rows + row) * 2 + g] = make_uint2(m0, m1);
                }
        }
    }
}

// max|x| of q, k, v (blockIdx.y picks the tensor) as float bits, into out[0..2] (zeroed by the caller)
__global__ __launch_bounds__(256) void at_absmax3_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                         const float *__restrict__ v, long q_bs, long k_bs, long v_bs,
                                                         long q_span, long kv_span, int B, unsigned *__restrict__ out)
{
    const int which = blockIdx.y;
    const float *p = which == 0 ? q : which == 1 ? k : v;
    const long bs = which == 0 ? q_bs : which == 1 ? k_bs : v_bs, span = which == 0 ? q_span : kv_span;
    float m = 0.f;
    const bool vec = (span & 3) == 0 && (bs & 3) == 0 && (((size_t)p) & 15) == 0;
    for (int b = 0; b < B; b++) {
        const float *pb = p + (size_t)b * bs;
        if (vec) {
            const long n4 = span / 4, step = (long)gridDim.x * 256;
            long i = (long)blockIdx.x * 256 + threadIdx.x;
            for (; i + 3 * step < n4; i += 4 * step) {                       // four loads in flight per thread
                const f32x4 x0 = *(const f32x4 *)(pb + 4 * i), x1 = *(const f32x4 *)(pb + 4 * (i + step)),
                            x2 = *(const f32x4 *)(pb + 4 * (i + 2 * step)), x3 = *(const f32x4 *)(pb + 4 * (i + 3 * step));
#pragma unroll
                for (int e = 0; e < 4; e++) m = fmaxf(fmaxf(m, fmaxf(fabsf(x0[e]), fabsf(x1[e]))), fmaxf(fabsf(x2[e]), fabsf(x3[e])));
            }
            for (; i < n4; i += step) {
                const f32x4 x = *(const f32x4 *)(pb + 4 * i);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(x[0]), fabsf(x[1]))), fmaxf(fabsf(x[2]), fabsf(x[3])));
            }
        } else {
            for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < span; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(pb[i]));
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    __shared__ float wmax[4];                        // ONE atomic per workgroup: thousands of atomics on three words serialise
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out + which, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

static int l3d_attention_absmax3(const float *q, const float *k, const float *v, long q_bs, long k_bs, long v_bs, long q_span, long kv_span,
                                 int B, unsigned *amax, hipStream_t st)
{
    if (hipMemsetAsync(amax, 0, 16, st) != hipSuccess) return L3D_ERR_LAUNCH;
    hipLaunchKernelGGL(at_absmax3_kernel, dim3(512, 3), dim3(256), 0, st, q, k, v, q_bs, k_bs, v_bs, q_span, kv_span, B, amax);
    return l3d_check_launch();
}

// workspace: 16 bytes of device memory (the three maxima; maxima_ready bit 0: already there, e.g. from
// l3d_pointwise_conv_f16 with its amax argument; bit 1: ctx_img with an UNSCALED residual plane, for the two-plane form of
// l3d_pointwise_conv_f16); ctx [B, H D, N] fp32 and / or ctx_img = the context as an fp16 activation image
// (l3d_f16_act_bytes(B N, H D) bytes) for l3d_pointwise_conv_f16; everything else as l3d_attention_forward_strided
extern "C" int l3d_attention_forward_f16b(const float *q, const float *k, const float *v, int B, int H, int D, int N, int M,
                                          long q_bstride, long k_bstride, long v_bstride, float scale, void *workspace,
                                          int maxima_ready, float *ctx, void *ctx_img, l3d_stream_t stream)
{
    L3D_REQUIRE(q && k && v && (ctx || ctx_img) && workspace && B > 0 && H > 0 && D > 0 && N > 0 && M > 0 && (maxima_ready & ~3) == 0);
    if (ctx_img && (((size_t)ctx_img) & 15)) return L3D_ERR_UNSUPPORTED;
    const size_t cpb = (size_t)(H * D / 8) * ((size_t)B * N) * 16;
    uint2 *cph = (uint2 *)ctx_img, *cpm = ctx_img ? (uint2 *)((unsigned char *)ctx_img + cpb) : nullptr;
    float *cinv = ctx_img ? (float *)((unsigned char *)ctx_img + 2 * cpb) : nullptr;
    if ((D != 32 && D != 64 && D != 128) || B > 65535 || H > 65535) return L3D_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    unsigned *amax = (unsigned *)workspace;
    const float img_rs = (maxima_ready & 2) ? 1.0f : 4096.0f;
    if (!(maxima_ready & 1)) {
        const int rc = l3d_attention_absmax3(q, k, v, q_bstride, k_bstride, v_bstride, (long)H * D * N, (long)H * D * M, B, amax, st);
        if (rc != L3D_OK) return rc;
    }
    dim3 grid(l3d_divup(N, AB_TQ) * H * B), block(512);
#define AB_LDS(ND_) (3 * (2 * ((32 * ND_) / 16) * 2 * 32 * 16 + 2 * 2 * 2 * ((32 * ND_) * 16 + 32)))
    if (D == 32)      hipLaunchKernelGGL(attention_f16b_kernel<1>, grid, block, AB_LDS(1), st, q, k, v, H, N, M, scale, ctx, q_bstride, k_bstride, v_bstride, amax, cph, cpm, cinv, img_rs);
    else if (D == 64) hipLaunchKernelGGL(attention_f16b_kernel<2>, grid, block, AB_LDS(2), st, q, k, v, H, N, M, scale, ctx, q_bstride, k_bstride, v_bstride, amax, cph, cpm, cinv, img_rs);
    else              hipLaunchKernelGGL(attention_f16b_kernel<4>, grid, block, AB_LDS(4), st, q, k, v, H, N, M, scale, ctx, q_bstride, k_bstride, v_bstride, amax, cph, cpm, cinv, img_rs);
#undef AB_LDS
    return l3d_check_launch();
}
