"""CPU model of edgeconv_kernel's data flow (mlp.hip): replays, lane by lane, the packed-fragment
layout produced by l3d_edgeconv_pack (a HOST function of libl3d_hip.so, callable without a GPU),
the MFMA 16x16x4 operand/accumulator lane maps, the row <-> (point, neighbour) mapping and the
epilogue, and checks the pooled output against a direct numpy evaluation of
models/dgcnn.py:32-46 (eval mode, BN folded).  Guards the index algebra that cannot be executed
here for lack of a GPU."""
import ctypes as C

import numpy as np
import pytest

from learning3d_amd import _lib

C1, C2, C3, C4 = 64, 64, 128, 256
# third weight copy (layers 2-4) for the bf16x3 kernel: [m][s][plane 3][lane 64][8 bf16] = 4 floats per fragment
SPLIT_FLOATS = ((C2 // 16) * (C1 // 32) + (C3 // 16) * (C2 // 32) + (C4 // 16) * (C3 // 32)) * 3 * 64 * 4
# fifth copy (two-plane f16x2 kernel): two-plane fragments + scaled biases of layers 2-4 + layer 1's scaled weights / bias + 16 constants
V2_FLOATS = SPLIT_FLOATS // 3 * 2 + (C2 + C3 + C4) + 8 * C1 + C1 + 16
MT = 5


def mfma_16x16x4(a, b, acc):
    """a, b: [64] per-lane operands; acc: [64,4].  A[i][k]=a[16k+i], B[k][j]=b[16k+j],
    D[i][j] -> lane 16*(i//4)+j, reg i%4 (cdna_hip_programming.md section 3)."""
    A = a.reshape(4, 16).T          # [i, k]
    Bm = b.reshape(4, 16)           # [k, j]
    D = A.astype(np.float64) @ Bm.astype(np.float64)
    out = acc.copy()
    for i in range(16):
        for j in range(16):
            out[16 * (i // 4) + j, i % 4] += D[i, j]
    return out


def run_layer(act, S, wfrag, cin, kv, nt_per_wave, cout_total, bias):
    """act: flat LDS image with row stride S.  Returns (next_act rows x cout, pooled [4, cout])."""
    kq_n = cin // (4 * kv)
    rows = MT * 16
    nxt = np.zeros((rows, cout_total), np.float64)
    pooled = np.zeros((4, cout_total), np.float64)
    lanes = np.arange(64)
    r, g = lanes & 15, lanes >> 4
    for wave in range(4):
        for i in range(nt_per_wave):
            nt = wave * nt_per_wave + i
            accs = [np.zeros((64, 4)) for _ in range(MT)]
            for kq in range(kq_n):
                frag = wfrag[((nt * kq_n + kq) * 64) * kv:((nt * kq_n + kq) * 64 + 64) * kv].reshape(64, kv)
                for s in range(kv):
                    k = (kq * kv + s) * 4 + g
                    for mt in range(MT):
                        a = act[(mt * 16 + r) * S + k]
                        accs[mt] = mfma_16x16x4(a, frag[:, s], accs[mt])
            ch = nt * 16 + (lanes & 15)
            for mt in range(MT):
                v = np.maximum(accs[mt] + bias[ch][:, None], 0.0)       # [64 lanes, 4 regs]
                for rr in range(4):
                    nxt[mt * 16 + g * 4 + rr, ch] = v[:, rr]
                    np.maximum.at(pooled, (g, ch), v[:, rr])
    return nxt, pooled


def to_lds(mat, S):
    rows, c = mat.shape
    img = np.zeros(rows * S)
    for row in range(rows):
        img[row * S:row * S + c] = mat[row]
    return img


def test_edgeconv_fragment_layout_and_row_mapping():
    lib = _lib.lib()
    rng = np.random.default_rng(0)
    ws = [rng.standard_normal((C1, 6)).astype(np.float32), rng.standard_normal((C2, C1)).astype(np.float32) * 0.2,
          rng.standard_normal((C3, C2)).astype(np.float32) * 0.2, rng.standard_normal((C4, C3)).astype(np.float32) * 0.1]
    scs = [rng.uniform(0.5, 1.5, c).astype(np.float32) for c in (C1, C2, C3, C4)]
    shs = [rng.uniform(-0.2, 0.2, c).astype(np.float32) for c in (C1, C2, C3, C4)]
    n = lib.l3d_edgeconv_packed_floats(C1, C2, C3, C4)
    # v1 + chained + biases + bf16x3 planes + f16x2 planes (same size) + their scaled biases + 16 scale constants
    assert n == 2 * (8 * C1 + C1 * C2 + C2 * C3 + C3 * C4) + C1 + C2 + C3 + C4 + 2 * SPLIT_FLOATS + (C2 + C3 + C4) + 16 + V2_FLOATS
    packed = np.zeros(n, np.float32)
    arr = lambda xs: (C.c_void_p * 4)(*[x.ctypes.data for x in xs])
    assert lib.l3d_edgeconv_pack(arr(ws), arr(scs), arr(shs), None, C1, C2, C3, C4, packed.ctypes.data) == 0
    assert lib.l3d_edgeconv_pack(arr(ws), arr(scs), arr(shs), None, 32, 32, 64, 128, packed.ctypes.data) == -2

    # one tile: 4 points, 20 neighbours each
    N, k = 16, 20
    xyz = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    idx = rng.integers(0, N, (N, k))
    n0 = 4
    feat = np.zeros((MT * 16, 8))
    for row in range(MT * 16):
        p, j = (row >> 2) & 3, (row >> 4) * 4 + (row & 3)
        feat[row, 0:3] = xyz[idx[n0 + p, j]]
        feat[row, 3:6] = xyz[n0 + p]
    o_w1, o_w2 = 0, 8 * C1
    o_w3 = o_w2 + C1 * C2
    o_w4 = o_w3 + C2 * C3
    o_b1 = o_w4 + C3 * C4
    o_b2, o_b3, o_b4 = o_b1 + C1, o_b1 + C1 + C2, o_b1 + C1 + C2 + C3
    pk = packed.astype(np.float64)
    h1, p1 = run_layer(to_lds(feat, 10), 10, pk[o_w1:o_w2], 8, 2, 1, C1, pk[o_b1:o_b2])
    h2, p2 = run_layer(to_lds(h1, C1 + 2), C1 + 2, pk[o_w2:o_w3], C1, 4, 1, C2, pk[o_b2:o_b3])
    h3, p3 = run_layer(to_lds(h2, C2 + 2), C2 + 2, pk[o_w3:o_w4], C2, 4, 2, C3, pk[o_b3:o_b4])
    _, p4 = run_layer(to_lds(h3, C3 + 2), C3 + 2, pk[o_w4:o_b1], C3, 4, 4, C4, pk[o_b4:o_b4 + C4])
    got = np.concatenate([p1, p2, p3, p4], axis=1)                 # [4 points, 512]

    # direct evaluation
    want = []
    for p in range(4):
        f = np.concatenate([xyz[idx[n0 + p]], np.repeat(xyz[n0 + p][None], k, 0)], axis=1).astype(np.float64)  # [k,6]
        outs = []
        h = f
        for w, sc, sh in zip(ws, scs, shs):
            h = np.maximum((h @ (w.astype(np.float64) * sc[:, None].astype(np.float64)).T) + sh, 0.0)
            outs.append(h.max(axis=0))
        want.append(np.concatenate(outs))
    want = np.stack(want)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)


def test_edgeconv_chained_register_layout():
    """CPU model of edgeconv2_kernel (edgeconv2.hip): weights are the MFMA A operand, activations the
    B operand taken DIRECTLY from the previous layer's accumulator registers; checks the k-order
    permutation (k-step (q,e), lane group g <-> channel 16q+4g+e), the second packed weight copy,
    the row <-> (point, neighbour) map and the quad max."""
    lib = _lib.lib()
    rng = np.random.default_rng(1)
    ws = [rng.standard_normal((C1, 6)).astype(np.float32), rng.standard_normal((C2, C1)).astype(np.float32) * 0.2,
          rng.standard_normal((C3, C2)).astype(np.float32) * 0.2, rng.standard_normal((C4, C3)).astype(np.float32) * 0.1]
    scs = [rng.uniform(0.5, 1.5, c).astype(np.float32) for c in (C1, C2, C3, C4)]
    shs = [rng.uniform(-0.2, 0.2, c).astype(np.float32) for c in (C1, C2, C3, C4)]
    n = lib.l3d_edgeconv_packed_floats(C1, C2, C3, C4)
    v1 = 8 * C1 + C1 * C2 + C2 * C3 + C3 * C4 + C1 + C2 + C3 + C4
    assert n == v1 + 8 * C1 + C1 * C2 + C2 * C3 + C3 * C4 + 2 * SPLIT_FLOATS + (C2 + C3 + C4) + 16 + V2_FLOATS
    packed = np.zeros(n, np.float32)
    arr = lambda xs: (C.c_void_p * 4)(*[x.ctypes.data for x in xs])
    assert lib.l3d_edgeconv_pack(arr(ws), arr(scs), arr(shs), None, C1, C2, C3, C4, packed.ctypes.data) == 0
    pk = packed.astype(np.float64)
    o_b = [8 * C1 + C1 * C2 + C2 * C3 + C3 * C4]
    o_b += [o_b[0] + C1, o_b[0] + C1 + C2, o_b[0] + C1 + C2 + C3]
    o2 = [v1, v1 + 8 * C1, v1 + 8 * C1 + C1 * C2, v1 + 8 * C1 + C1 * C2 + C2 * C3]

    N, k = 16, 20
    xyz = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    idx = rng.integers(0, N, (N, k))
    n0 = 8                                          # this wave's 4 points
    lanes = np.arange(64)
    j, g = lanes & 15, lanes >> 4

    def mfma(a, b, acc):                            # A[i][k]=a[16k+i] (weights), B[k][j]=b[16k+j] (acts)
        A = a.reshape(4, 16).T
        Bm = b.reshape(4, 16)
        D = A @ Bm                                  # [i=out ch in tile][j=row in tile]
        out = acc.copy()
        for i in range(16):
            for jj in range(16):
                out[16 * (i // 4) + jj, i % 4] += D[i, jj]
        return out

    # layer-1 B operands
    b1 = np.zeros((MT, 2, 64))
    for t in range(MT):
        for l in range(64):
            p, nbr = n0 + (j[l] >> 2), 4 * t + (j[l] & 3)
            f = np.concatenate([xyz[idx[p, nbr]], xyz[p], [0, 0]])
            b1[t, 0, l] = f[g[l]]
            b1[t, 1, l] = f[4 + g[l]]
    def bias_init(off, m):
        acc = np.zeros((64, 4))
        for r in range(4):
            acc[:, r] = pk[off + 16 * m + 4 * g + r]
        return acc
    pooled = {}
    def pool(h_m, choff, m):                       # h_m: [MT][64,4] after relu
        mx = np.max(np.stack(h_m), axis=0)         # over row tiles
        for l in range(64):
            q0 = l & ~3
            val = mx[q0:q0 + 4].max(axis=0)        # quad max
            p = n0 + (j[l] >> 2)
            for r in range(4):
                pooled[(p, choff + 16 * m + 4 * g[l] + r)] = val[r]
    # layer 1
    h = []
    w1 = pk[o2[0]:o2[0] + 8 * C1].reshape(C1 // 16, 64, 2)
    for m in range(C1 // 16):
        row = []
        for t in range(MT):
            acc = bias_init(o_b[0], m)
            for s in range(2):
                acc = mfma(w1[m, :, s], b1[t, s], acc)
            row.append(np.maximum(acc, 0))
        pool(row, 0, m)
        h.append(row)
    choff = C1
    for li, (cin, cout) in enumerate([(C1, C2), (C2, C3), (C3, C4)], start=1):
        nq = cin // 16
        wl = pk[o2[li]:o2[li] + cin * cout].reshape(cout // 16, nq, 64, 4)
        hn = []
        for m in range(cout // 16):
            row = []
            for t in range(MT):
                acc = bias_init(o_b[li], m)
                for q in range(nq):
                    for e in range(4):
                        acc = mfma(wl[m, q, :, e], h[q][t][:, e], acc)
                row.append(np.maximum(acc, 0))
            pool(row, choff, m)
            hn.append(row)
        h = hn
        choff += cout
    got = np.array([[pooled[(n0 + p, c)] for c in range(C1 + C2 + C3 + C4)] for p in range(4)])
    want = []
    for p in range(4):
        f = np.concatenate([xyz[idx[n0 + p]], np.repeat(xyz[n0 + p][None], k, 0)], axis=1).astype(np.float64)
        outs, hh = [], f
        for w, sc, sh in zip(ws, scs, shs):
            hh = np.maximum((hh @ (w.astype(np.float64) * sc[:, None].astype(np.float64)).T) + sh, 0.0)
            outs.append(hh.max(axis=0))
        want.append(np.concatenate(outs))
    np.testing.assert_allclose(got, np.stack(want), rtol=1e-6, atol=1e-6)



def _bf16_round(x):
    """fp32 -> nearest-even bf16, returned as float32 (numpy model of v_cvt_pk_bf16_f32)"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def _split3(x):
    x = np.asarray(x, np.float32)
    h = _bf16_round(x)
    r = x - h
    m = _bf16_round(r)
    return h, m, _bf16_round(r - m)


def test_edgeconv_split_bf16x3_layout():
    """CPU model of edgeconv_split_kernel (edgeconv_split.hip): layers 2-4 as v_mfma_f32_16x16x32_bf16
    with weights (A) from the third packed copy and activations (B) built in-lane from PAIRS of
    previous-layer accumulators: k-step s, lane group g, slot 4u+e <-> channel 16(2s+u) + 4g + e.
    Checks that packing, the host-side split (planes sum EXACTLY to the folded fp32 weight), the slot
    permutation and the six-product sum reproduce the layer stack."""
    lib = _lib.lib()
    rng = np.random.default_rng(2)
    ws = [rng.standard_normal((C1, 6)).astype(np.float32), rng.standard_normal((C2, C1)).astype(np.float32) * 0.2,
          rng.standard_normal((C3, C2)).astype(np.float32) * 0.2, rng.standard_normal((C4, C3)).astype(np.float32) * 0.1]
    scs = [rng.uniform(0.5, 1.5, c).astype(np.float32) for c in (C1, C2, C3, C4)]
    shs = [rng.uniform(-0.2, 0.2, c).astype(np.float32) for c in (C1, C2, C3, C4)]
    n = lib.l3d_edgeconv_packed_floats(C1, C2, C3, C4)
    packed = np.zeros(n, np.float32)
    arr = lambda xs: (C.c_void_p * 4)(*[x.ctypes.data for x in xs])
    assert lib.l3d_edgeconv_pack(arr(ws), arr(scs), arr(shs), None, C1, C2, C3, C4, packed.ctypes.data) == 0
    v1 = 8 * C1 + C1 * C2 + C2 * C3 + C3 * C4 + C1 + C2 + C3 + C4
    o_b = [8 * C1 + C1 * C2 + C2 * C3 + C3 * C4]
    o_b += [o_b[0] + C1, o_b[0] + C1 + C2, o_b[0] + C1 + C2 + C3]
    o2_w1 = v1
    o3 = v1 + 8 * C1 + C1 * C2 + C2 * C3 + C3 * C4
    lanes = np.arange(64)
    j, g = lanes & 15, lanes >> 4

    # ---- the split weight copy: decode, check exactness and the slot permutation
    planes = {}
    off = o3
    for li, (cin, cout) in enumerate([(C1, C2), (C2, C3), (C3, C4)], start=1):
        S, M = cin // 32, cout // 16
        cnt = M * S * 3 * 64 * 4
        raw = packed[off:off + cnt].view(np.uint16).reshape(M // 2, S, 2, 3, 64, 8)     # [pair][k-step][m&1][plane][lane][slot]
        raw = raw.transpose(0, 2, 1, 3, 4, 5).reshape(M, S, 3, 64, 8)                  # -> [m][k-step][plane][lane][slot]
        f = (np.ascontiguousarray(raw).astype(np.uint32) << 16).view(np.float32)       # bf16 -> fp32
        planes[li] = f
        folded = (ws[li] * scs[li][:, None]).astype(np.float32)
        for m in range(M):
            for s_ in range(S):
                for slot in range(8):
                    oc = 16 * m + j
                    ic = 32 * s_ + 16 * (slot >> 2) + 4 * g + (slot & 3)
                    tot = f[m, s_, 0, :, slot].astype(np.float64) + f[m, s_, 1, :, slot] + f[m, s_, 2, :, slot]
                    np.testing.assert_array_equal(tot.astype(np.float32), folded[oc, ic])
        off += cnt
    assert off == o3 + SPLIT_FLOATS                    # the f16x2 copy follows (test_edgeconv_f16x2_pack)

    # ---- lane-level forward for one wave (4 points x 20 neighbours)
    N, k = 16, 20
    xyz = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    idx = rng.integers(0, N, (N, k))
    n0 = 4
    pk = packed.astype(np.float64)

    def mfma_f32(a, b, acc):                       # 16x16x4 fp32, as in the chained model above
        D = a.reshape(4, 16).T @ b.reshape(4, 16)
        out = acc.copy()
        for i in range(16):
            out[16 * (i // 4) + np.arange(16), i % 4] += D[i]
        return out

    def mfma_bf16(a, b, acc):                      # a, b: [64 lanes][8 slots]; k index = 8*(lane>>4) + slot
        A = np.zeros((16, 32)); Bm = np.zeros((32, 16))
        for l in range(64):
            A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a[l]
            Bm[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b[l]
        D = A @ Bm
        out = acc.copy()
        for i in range(16):
            out[16 * (i // 4) + np.arange(16), i % 4] += D[i]
        return out

    def bias_init(off_b, m):
        acc = np.zeros((64, 4))
        for r in range(4):
            acc[:, r] = pk[off_b + 16 * m + 4 * g + r]
        return acc

    pooled = {}
    def pool(h_m, choff, m):
        mx = np.max(np.stack(h_m), axis=0)
        for l in range(64):
            q0 = l & ~3
            val = mx[q0:q0 + 4].max(axis=0)
            for r in range(4):
                pooled[(n0 + (j[l] >> 2), choff + 16 * m + 4 * g[l] + r)] = val[r]

    # layer 1 (fp32 MFMA)
    w1 = pk[o2_w1:o2_w1 + 8 * C1].reshape(C1 // 16, 64, 2)
    h = []
    for m in range(C1 // 16):
        row = []
        for t in range(MT):
            b1 = np.zeros((2, 64))
            for l in range(64):
                p, nbr = n0 + (j[l] >> 2), 4 * t + (j[l] & 3)
                f = np.concatenate([xyz[idx[p, nbr]], xyz[p], [0, 0]])
                b1[0, l], b1[1, l] = f[g[l]], f[4 + g[l]]
            acc = bias_init(o_b[0], m)
            for s_ in range(2):
                acc = mfma_f32(w1[m, :, s_], b1[s_], acc)
            row.append(np.maximum(acc, 0).astype(np.float32))
        pool(row, 0, m)
        h.append(row)
    choff = C1
    for li, (cin, cout) in enumerate([(C1, C2), (C2, C3), (C3, C4)], start=1):
        S, M = cin // 32, cout // 16
        # B planes per k-step: slots 0..3 = M-tile 2s regs, 4..7 = M-tile 2s+1 regs (in-lane)
        bpl = [[None] * MT for _ in range(S)]
        for s_ in range(S):
            for t in range(MT):
                v = np.concatenate([h[2 * s_][t], h[2 * s_ + 1][t]], axis=1)       # [64, 8]
                bpl[s_][t] = _split3(v)
        hn = []
        for m in range(M):
            row = []
            for t in range(MT):
                acc = bias_init(o_b[li], m)
                for s_ in range(S):
                    a = planes[li][m, s_]                                          # [3][64][8]
                    xh, xm, xl = bpl[s_][t]
                    for (pa, xb) in ((2, xh), (0, xl), (1, xm), (1, xh), (0, xm), (0, xh)):
                        acc = mfma_bf16(a[pa].astype(np.float64), xb.astype(np.float64), acc)
                row.append(np.maximum(acc, 0).astype(np.float32))
            pool(row, choff, m)
            hn.append(row)
        h = hn
        choff += cout
    got = np.array([[pooled[(n0 + p, c)] for c in range(C1 + C2 + C3 + C4)] for p in range(4)])
    want = []
    for p in range(4):
        f = np.concatenate([xyz[idx[n0 + p]], np.repeat(xyz[n0 + p][None], k, 0)], axis=1).astype(np.float64)
        outs, hh = [], f
        for w, sc, sh in zip(ws, scs, shs):
            hh = np.maximum((hh @ (w * sc[:, None]).astype(np.float32).astype(np.float64).T) + sh, 0.0)
            outs.append(hh.max(axis=0))
        want.append(np.concatenate(outs))
    np.testing.assert_allclose(got, np.stack(want), rtol=2e-6, atol=2e-6)


def test_edgeconv_f16x2_pack():
    """The fourth packed copy (edgeconv_f16.hip): per layer W = w' 2^S with max|W| in [4,8); planes H = f16(W),
    Hs = f16(H 2^-12), M = f16(W - H) in the fragment order of the bf16x3 copy, H + M = W to 2^-22 |W| worst case
    (two 11-bit roundings; + fp16's subnormal floor); biases in accumulator units; the 16 power-of-two scale constants tie the layers together
    (cs_l = 2^T_l / A_l, cp_l = 1 / A_l, co_l = 2^T_out / A_l, A_l = 2^(S_l + T_(l-1)), A_1 = 1) with T_l taken from the
    expected activation magnitudes handed to l3d_edgeconv_pack."""
    lib = _lib.lib()
    rng = np.random.default_rng(3)
    ws = [rng.standard_normal((C1, 6)).astype(np.float32), rng.standard_normal((C2, C1)).astype(np.float32) * 0.2,
          rng.standard_normal((C3, C2)).astype(np.float32) * 0.02, rng.standard_normal((C4, C3)).astype(np.float32) * 3.0]
    scs = [rng.uniform(0.5, 1.5, c).astype(np.float32) for c in (C1, C2, C3, C4)]
    shs = [rng.uniform(-0.2, 0.2, c).astype(np.float32) for c in (C1, C2, C3, C4)]
    mags = np.array([1.0, 5.0, 0.03, 700.0], np.float32)
    n = lib.l3d_edgeconv_packed_floats(C1, C2, C3, C4)
    packed = np.zeros(n, np.float32)
    arr = lambda xs: (C.c_void_p * 4)(*[x.ctypes.data for x in xs])
    assert lib.l3d_edgeconv_pack(arr(ws), arr(scs), arr(shs), mags.ctypes.data, C1, C2, C3, C4, packed.ctypes.data) == 0
    v1 = 8 * C1 + C1 * C2 + C2 * C3 + C3 * C4 + C1 + C2 + C3 + C4
    o4 = v1 + 8 * C1 + C1 * C2 + C2 * C3 + C3 * C4 + SPLIT_FLOATS
    o_b4 = o4 + SPLIT_FLOATS
    o_sc = o_b4 + C2 + C3 + C4
    assert o_sc + 16 + V2_FLOATS == n
    sc = packed[o_sc:o_sc + 16].astype(np.float64)
    T = [12 - (int(np.floor(np.log2(m))) + 1) for m in mags]           # mag 2^T in [2^11, 2^12)
    assert all(2 ** 11 <= m * 2.0 ** t < 2 ** 12 for m, t in zip(mags, T))
    Tout = min(T)
    lanes = np.arange(64)
    j, g = lanes & 15, lanes >> 4
    off, boff = o4, o_b4
    A = [0.0]
    for li, (cin, cout) in enumerate([(C1, C2), (C2, C3), (C3, C4)], start=1):
        S_, M_ = cin // 32, cout // 16
        cnt = M_ * S_ * 3 * 64 * 4
        raw = packed[off:off + cnt].view(np.float16).reshape(M_ // 2, S_, 2, 3, 64, 8).transpose(0, 2, 1, 3, 4, 5).reshape(M_, S_, 3, 64, 8)
        folded = (ws[li] * scs[li][:, None]).astype(np.float32)
        Sexp = 3 - (int(np.floor(np.log2(np.abs(folded).max()))) + 1)
        assert 4 <= np.abs(folded).max() * 2.0 ** Sexp < 8
        Wsc = folded.astype(np.float64) * 2.0 ** Sexp
        for m in range(0, M_, max(1, M_ // 4)):
            for s_ in range(S_):
                for slot in range(8):
                    oc = 16 * m + j
                    ic = 32 * s_ + 16 * (slot >> 2) + 4 * g + (slot & 3)
                    H, Hs, Mm = (raw[m, s_, p, :, slot].astype(np.float64) for p in range(3))
                    np.testing.assert_array_equal(H, Wsc[oc, ic].astype(np.float16).astype(np.float64))
                    np.testing.assert_array_equal(Hs, (H * 2.0 ** -12).astype(np.float16).astype(np.float64))
                    assert np.all(np.abs(H + Mm - Wsc[oc, ic]) <= 2.0 ** -22 * np.abs(Wsc[oc, ic]) + 2.0 ** -25)
        a_l = Sexp + T[li - 1]                                           # log2 of layer li+1's accumulator scale
        A.append(float(a_l))
        np.testing.assert_array_equal(packed[boff:boff + cout].astype(np.float64), shs[li].astype(np.float64) * 2.0 ** a_l)
        off += cnt
        boff += cout
    for l in range(4):
        if l < 3:
            assert sc[l] == 2.0 ** (T[l] - A[l])
        assert sc[4 + l] == 2.0 ** (-A[l]) and sc[8 + l] == 2.0 ** (Tout - A[l])
    assert sc[12] == 2.0 ** (-Tout)


def test_edgeconv_f16x2_two_plane_pack():
    """The fifth packed copy (edgeconv_f16b.hip): two fp16 weight planes H = f16(W), M = f16(W - H) per fragment step with the
    fourth copy's weight scaling (max|W| in [4,8)); the accumulators of layers 1-3 are the next layer's planes -- layer 1's fp32
    weights / bias times 2^T_1, T_l = S_l + T_(l-1), T_1 such that the highest-placed layer's expected magnitude sits in
    [2^11, 2^12) -- biases in accumulator units, the pooled-output constants, and the usability flag (a layer whose expected
    magnitude would be placed below 2^4 clears it)."""
    lib = _lib.lib()
    rng = np.random.default_rng(5)
    ws = [rng.standard_normal((C1, 6)).astype(np.float32), rng.standard_normal((C2, C1)).astype(np.float32) * 0.2,
          rng.standard_normal((C3, C2)).astype(np.float32) * 0.05, rng.standard_normal((C4, C3)).astype(np.float32) * 3.0]
    scs = [rng.uniform(0.5, 1.5, c).astype(np.float32) for c in (C1, C2, C3, C4)]
    shs = [rng.uniform(-0.2, 0.2, c).astype(np.float32) for c in (C1, C2, C3, C4)]
    arr = lambda xs: (C.c_void_p * 4)(*[x.ctypes.data for x in xs])
    n = lib.l3d_edgeconv_packed_floats(C1, C2, C3, C4)
    flag = lib.l3d_edgeconv_packed_v2_flag_index()
    o5 = n - V2_FLOATS
    assert flag == n - 16 + 13

    def pack(mags):
        packed = np.zeros(n, np.float32)
        m = np.asarray(mags, np.float32)
        assert lib.l3d_edgeconv_pack(arr(ws), arr(scs), arr(shs), m.ctypes.data, C1, C2, C3, C4, packed.ctypes.data) == 0
        return packed
    mags = [3.0, 5.0, 0.7, 40.0]
    packed = pack(mags)
    assert packed[flag] == 1.0
    e = [int(np.floor(np.log2(m))) + 1 for m in mags]                     # mag in [2^(e-1), 2^e)
    folded = [(w * sc[:, None]).astype(np.float32) for w, sc in zip(ws, scs)]
    S0 = [0] + [3 - (int(np.floor(np.log2(np.abs(f).max()))) + 1) for f in folded[1:]]
    assert all(4 <= np.abs(f).max() * 2.0 ** s_ < 8 for f, s_ in zip(folded[1:], S0[1:]))

    def place(S0, e):
        """the packer's policy: T_1 puts the highest layer at 2^12; layers 2, 3 give up to two binades of weight scale each
        until every layer's expected magnitude sits at >= 2^4"""
        S = list(S0)
        for _ in range(5):
            cum = [0, S[1], S[1] + S[2]]
            T1 = min(12 - e[l] - cum[l] for l in range(3))
            T = [T1 + cum[l] for l in range(3)]
            ok = min(e[l] + T[l] for l in range(3)) >= 4
            if ok:
                break
            if S[2] > S0[2] - 2:
                S[2] -= 1
            elif S[1] > S0[1] - 2:
                S[1] -= 1
            else:
                break
        return S, T, ok
    S, T, ok = place(S0, e)
    assert ok and max(e[l] + T[l] for l in range(3)) == 12 and min(e[l] + T[l] for l in range(3)) >= 4
    assert all(1 <= np.abs(f).max() * 2.0 ** s_ < 8 for f, s_ in zip(folded[1:], S[1:]))
    A = [T[0], S[1] + T[0], S[2] + T[1], S[3] + T[2]]
    assert A[1] == T[1] and A[2] == T[2]                                   # accumulator units = the next layer's plane units
    lanes = np.arange(64)
    j, g = lanes & 15, lanes >> 4
    off = o5
    frag_floats = [(C2 // 16) * (C1 // 32) * 2 * 64 * 4, (C3 // 16) * (C2 // 32) * 2 * 64 * 4, (C4 // 16) * (C3 // 32) * 2 * 64 * 4]
    o_b = o5 + sum(frag_floats)
    o_w1 = o_b + C2 + C3 + C4
    o_b1 = o_w1 + 8 * C1
    o_sc = o_b1 + C1
    assert o_sc + 16 == n
    boff = o_b
    for li, (cin, cout) in enumerate([(C1, C2), (C2, C3), (C3, C4)], start=1):
        S_, M_ = cin // 32, cout // 16
        raw = packed[off:off + frag_floats[li - 1]].view(np.float16).reshape(M_ // 2, S_, 2, 2, 64, 8).transpose(0, 2, 1, 3, 4, 5).reshape(M_, S_, 2, 64, 8)
        Wsc = folded[li].astype(np.float64) * 2.0 ** S[li]
        for m in range(0, M_, max(1, M_ // 4)):
            for s_ in range(S_):
                for slot in range(8):
                    oc = 16 * m + j
                    ic = 32 * s_ + 16 * (slot >> 2) + 4 * g + (slot & 3)
                    H, Mm = (raw[m, s_, p, :, slot].astype(np.float64) for p in range(2))
                    np.testing.assert_array_equal(H, Wsc[oc, ic].astype(np.float16).astype(np.float64))
                    assert np.all(np.abs(H + Mm - Wsc[oc, ic]) <= 2.0 ** -22 * np.abs(Wsc[oc, ic]) + 2.0 ** -25)
        np.testing.assert_array_equal(packed[boff:boff + cout].astype(np.float64), shs[li].astype(np.float64) * 2.0 ** A[li])
        off += frag_floats[li - 1]
        boff += cout
    # layer 1: the second copy's fragments and the bias, times 2^T_1
    v1 = 8 * C1 + C1 * C2 + C2 * C3 + C3 * C4 + C1 + C2 + C3 + C4
    np.testing.assert_array_equal(packed[o_w1:o_w1 + 8 * C1], packed[v1:v1 + 8 * C1] * np.float32(2.0 ** T[0]))
    np.testing.assert_array_equal(packed[o_b1:o_b1 + C1], shs[0] * np.float32(2.0 ** T[0]))
    sc = packed[o_sc:o_sc + 16].astype(np.float64)
    Tout = min(12 - x for x in e)
    for l in range(4):
        assert sc[4 + l] == 2.0 ** (-A[l]) and sc[8 + l] == 2.0 ** (Tout - A[l])
    assert sc[12] == 2.0 ** (-Tout) and sc[0] == sc[1] == sc[2] == 1.0
    # expected magnitudes so far apart that one layer would sit below 2^4: the flag drops (the host runs the three-plane kernel)
    for bad in ([3.0, 3.0e6, 0.7, 40.0], [3.0, 5.0, 1.0e5, 40.0], [4e-3, 4.0, 4.0, 4.0]):
        assert pack(bad)[flag] == 0.0 and not place(S0, [int(np.floor(np.log2(m))) + 1 for m in bad])[2]
    # default-initialised DGCNN (max|w| ~ 1/8 in every layer, magnitudes 4): usable after lowering the weight scales
    ws_d = [rng.uniform(-0.4, 0.4, (C1, 6)).astype(np.float32), rng.uniform(-0.125, 0.125, (C2, C1)).astype(np.float32),
            rng.uniform(-0.125, 0.125, (C3, C2)).astype(np.float32), rng.uniform(-0.088, 0.088, (C4, C3)).astype(np.float32)]
    ones = [np.ones(c, np.float32) for c in (C1, C2, C3, C4)]
    packed = np.zeros(n, np.float32)
    m4 = np.full(4, 4.0, np.float32)
    assert lib.l3d_edgeconv_pack(arr(ws_d), arr(ones), arr(shs), m4.ctypes.data, C1, C2, C3, C4, packed.ctypes.data) == 0
    assert packed[flag] == 1.0
