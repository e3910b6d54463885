"""CPU model of the "f16x2" arithmetic the GEMM kernels run on the fp16 matrix cores (edgeconv_f16b.hip, conv_f16.hip, fold_mlp_f16.hip,
attention_f16b.hip): every fp32 operand as two fp16 planes, three fp16 x fp16 products per fp32 product, fp32 accumulation.  numpy float16
rounds to nearest even like v_cvt_f16_f32 / v_cvt_pk_f16_f32, and an fp16 x fp16 product is exact in fp32, so the model below is the
kernels' arithmetic up to the order of the fp32 accumulation.  What the GPU tests measure against fp64 on the device is bounded here
without one: the error of a K-term dot product stays at the level of a plain fp32 dot product for both residual conventions
 * scaled residual     x 2^T = h + m' 2^-12,  products  M h + (H 2^-12) m' + H h     (three weight planes, any operand scale)
 * unscaled residual   x 2^T = h + m,         products  M h + H m + H h              (two weight planes; operands placed near 2^11)."""
import numpy as np


def _planes(v, hi_exp, scaled):
    """v (fp32) -> (h, m, 2^-T) with max|v| 2^T in [2^(hi_exp-1), 2^hi_exp)"""
    mx = float(np.abs(v).max())
    T = hi_exp - int(np.frexp(mx)[1]) if mx > 0 else 0
    X = (v * np.float32(2.0 ** T)).astype(np.float32)                      # a power of two: exact
    h = X.astype(np.float16)
    r = (X - h.astype(np.float32)).astype(np.float32)                      # exact in fp32
    m = (r * np.float32(4096.0)).astype(np.float16) if scaled else r.astype(np.float16)
    return h, m, 2.0 ** -T


def _dot_f16x2(w, x, scaled):
    """sum_k w_k x_k: weights placed with max|W| in [4, 8), activations with max|X| in [2^11, 2^12)"""
    H, M, cw = _planes(w, 3, False)                                        # W = H + M (the weight residual is never scaled)
    h, m, cx = _planes(x, 12, scaled)
    f32 = lambda a: a.astype(np.float32)
    Hs = (f32(H) * np.float32(2.0 ** -12)).astype(np.float16) if scaled else H
    acc = np.float32(0.0)
    for p, q in ((M, h), (Hs, m), (H, h)):                                 # smallest first, as the kernels issue them
        for k in range(len(w)):
            acc = np.float32(acc + f32(p[k]) * f32(q[k]))                  # fp16 x fp16 is exact in fp32; the sum rounds
    return float(acc) * cw * cx


def test_f16x2_dot_products_are_fp32_accurate():
    rng = np.random.default_rng(7)
    for scaled in (True, False):
        worst = 0.0
        for K in (64, 128, 512):
            for scale in (1e-3, 1.0, 100.0):
                for _ in range(6):
                    w = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32)
                    x = np.maximum(rng.standard_normal(K) * scale, 0).astype(np.float32)       # post-ReLU activations
                    exact = float(np.dot(w.astype(np.float64), x.astype(np.float64)))
                    acc32 = np.float32(0.0)
                    for k in range(K):
                        acc32 = np.float32(acc32 + w[k] * x[k])
                    bound = float(np.dot(np.abs(w).astype(np.float64), np.abs(x).astype(np.float64)))   # sum |w_k x_k|
                    e_f16x2 = abs(_dot_f16x2(w, x, scaled) - exact) / bound
                    e_fp32 = abs(float(acc32) - exact) / bound
                    worst = max(worst, e_f16x2)
                    # a handful of fp32 roundings of the running sum + 2^-22 per operand: the same order as the fp32 dot product's own
                    # error (K 2^-24 worst case, far less on average)
                    assert e_f16x2 <= max(4 * e_fp32, 3 * K ** 0.5 * 2.0 ** -24), (scaled, K, scale, e_f16x2, e_fp32)
        assert worst <= 2.0 ** -17


def test_unscaled_residual_needs_the_operand_near_2_to_11():
    """why the two-plane kernels place their activations with the maximum in [2^11, 2^12): further down the fp16 residual goes subnormal
    and the split loses bits; the scaled residual does not care"""
    x = np.float32(1.2345678e-3)                                            # 2^-10: the residual of h is ~2^-21, subnormal in fp16 (< 2^-14)
    for scaled, ok in ((True, True), (False, False)):
        h = np.float16(x)
        r = np.float32(x - np.float32(h))
        m = np.float16(r * np.float32(4096.0)) if scaled else np.float16(r)
        back = np.float32(h) + (np.float32(m) * np.float32(2.0 ** -12) if scaled else np.float32(m))
        rel = abs(float(back) - float(x)) / float(x)
        assert (rel <= 2.0 ** -21) == ok, (scaled, rel)
    X = np.float32(x * 2.0 ** 21)                                           # placed near 2^11: the unscaled residual is a normal fp16 number
    h = np.float16(X)
    m = np.float16(np.float32(X - np.float32(h)))
    assert abs(float(np.float32(h) + np.float32(m)) - float(X)) / float(X) <= 2.0 ** -21
