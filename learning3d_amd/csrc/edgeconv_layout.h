// edgeconv_layout.h -- offsets inside the packed EdgeConv parameter block (l3d_edgeconv_pack).
#pragma once
#define EC_C1 64
#define EC_C2 64
#define EC_C3 128
#define EC_C4 256
#define EC_OFF_W1 0
#define EC_OFF_W2 (EC_OFF_W1 + 8 * EC_C1)
#define EC_OFF_W3 (EC_OFF_W2 + EC_C1 * EC_C2)
#define EC_OFF_W4 (EC_OFF_W3 + EC_C2 * EC_C3)
#define EC_OFF_B1 (EC_OFF_W4 + EC_C3 * EC_C4)
#define EC_OFF_B2 (EC_OFF_B1 + EC_C1)
#define EC_OFF_B3 (EC_OFF_B2 + EC_C2)
#define EC_OFF_B4 (EC_OFF_B3 + EC_C3)
#define EC_PACKED_V1_FLOATS (EC_OFF_B4 + EC_C4)

// second weight copy for the register-chained kernel (edgeconv2.hip), see l3d_edgeconv_pack
#define EC2_OFF_W1 EC_PACKED_V1_FLOATS
#define EC2_OFF_W2 (EC2_OFF_W1 + 8 * EC_C1)
#define EC2_OFF_W3 (EC2_OFF_W2 + EC_C1 * EC_C2)
#define EC2_OFF_W4 (EC2_OFF_W3 + EC_C2 * EC_C3)
#define EC2_END (EC2_OFF_W4 + EC_C3 * EC_C4)

// third copy, layers 2-4 only, for the bf16x3 kernel (edgeconv_split.hip): every weight split into
// three bf16 planes, fragment-ordered for v_mfma_f32_16x16x32_bf16 as the A operand.  Offsets and
// sizes in floats (one 16-byte fragment = 4 floats); block [step][plane 3][lane 64][8 bf16] with
// step = ((m/2)*S + s)*2 + (m&1), S = Cin/32 k-steps, m = output M-tile of 16 channels (the order the
// kernel consumes them: M-tile pair, k-step, M-tile of the pair).
#define EC3_OFF_W2 EC2_END
#define EC3_OFF_W3 (EC3_OFF_W2 + (EC_C2 / 16) * (EC_C1 / 32) * 3 * 64 * 4)
#define EC3_OFF_W4 (EC3_OFF_W3 + (EC_C3 / 16) * (EC_C2 / 32) * 3 * 64 * 4)
#define EC3_END (EC3_OFF_W4 + (EC_C4 / 16) * (EC_C3 / 32) * 3 * 64 * 4)

// fourth copy, layers 2-4, for the f16x2 kernel (edgeconv_f16.hip): W = w' 2^S (S per layer so that max|W| is in
// [4,8)) as three fp16 planes H = f16(W), Hs = f16(H 2^-12), M = f16(W - H), same fragment order as the third copy
// ([step][plane 3: H, Hs, M][lane 64][8 f16]); then the biases of layers 2-4 in accumulator units, then 16 scale
// constants.  Activation planes are stored times 2^T (T_l per layer from the expected activation magnitude, so that it
// sits near 2^12 in fp16's 2^-14 .. 2^16; T_out = min_l T_l for the pooled planes handed to conv5); layer l >= 2
// accumulates A_l = 2^(S_l + T_(l-1)) times the true value (A_1 = 1).  EC4_OFF_SC + :
//   0..2  cs_l = 2^T_l / A_l, l = 1..3 (accumulator -> this layer's planes)      4..7  cp_l = 1 / A_l (-> fp32 pooled)
//   8..11 co_l = 2^T_out / A_l (-> pooled planes for conv5)                      12    2^-T_out
#define EC4_OFF_W2 EC3_END
#define EC4_OFF_W3 (EC4_OFF_W2 + (EC_C2 / 16) * (EC_C1 / 32) * 3 * 64 * 4)
#define EC4_OFF_W4 (EC4_OFF_W3 + (EC_C3 / 16) * (EC_C2 / 32) * 3 * 64 * 4)
#define EC4_OFF_B2 (EC4_OFF_W4 + (EC_C4 / 16) * (EC_C3 / 32) * 3 * 64 * 4)
#define EC4_OFF_B3 (EC4_OFF_B2 + EC_C2)
#define EC4_OFF_B4 (EC4_OFF_B3 + EC_C3)
#define EC4_OFF_SC (EC4_OFF_B4 + EC_C4)
#define EC4_END (EC4_OFF_SC + 16)

// fifth copy, for the two-plane f16x2 kernel (edgeconv_f16b.hip = edgeconv_f16.hip with EF_V2): the activation residual is
// carried UNSCALED (m = f16(x - h)), so the Hs plane is gone -- two fp16 weight planes H = f16(W), M = f16(W - H) per
// fragment step ([step][plane 2: H, M][lane 64][8 f16], same step order) and the products are M h + H m + H h.  Weights are
// scaled as in the fourth copy (W = w' 2^S_l, max|W| in [4,8)).  The accumulators of layers 1-3 ARE the next layer's planes
// (no multiply in the split: h = cvt_pk_f16(a), m = cvt_pk_f16(a - h)): layer 1's fp32 weights and bias are stored times
// 2^T_1 (EC5_OFF_W1 / _B1, layout of the second copy) and T_l = S_l + T_(l-1) for l = 2, 3; T_1 is chosen so that the
// highest-placed layer has its expected magnitude in [2^11, 2^12).  A layer placed below 2^4 clears EC5_OFF_SC + 13 and
// the host uses the three-plane kernel.  Layer 4: A_4 = 2^(S_4 + T_3).
// EC5_OFF_SC + :  4..7 cp_l = 1 / A_l (-> fp32 pooled)   8..11 co_l = 2^T_out / A_l (-> pooled planes)   12  2^-T_out
//                 13  1.0 if the block is usable
#define EC5_NPL 2
#define EC5_OFF_W2 EC4_END
#define EC5_OFF_W3 (EC5_OFF_W2 + (EC_C2 / 16) * (EC_C1 / 32) * EC5_NPL * 64 * 4)
#define EC5_OFF_W4 (EC5_OFF_W3 + (EC_C3 / 16) * (EC_C2 / 32) * EC5_NPL * 64 * 4)
#define EC5_OFF_B2 (EC5_OFF_W4 + (EC_C4 / 16) * (EC_C3 / 32) * EC5_NPL * 64 * 4)
#define EC5_OFF_B3 (EC5_OFF_B2 + EC_C2)
#define EC5_OFF_B4 (EC5_OFF_B3 + EC_C3)
#define EC5_OFF_W1 (EC5_OFF_B4 + EC_C4)
#define EC5_OFF_B1 (EC5_OFF_W1 + 8 * EC_C1)
#define EC5_OFF_SC (EC5_OFF_B1 + EC_C1)
#define EC_PACKED_FLOATS (EC5_OFF_SC + 16)
