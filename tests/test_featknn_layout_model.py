"""CPU model of featknn.hip's bookkeeping (round-6 kernel; no GPU): the plane image the split kernel writes and the DMA pieces / fragment
addresses the kernel reads it with, the key <-> (wave, lane, accumulator, register) mapping of the 32x32x16 MFMA output the selection
relies on, the validity of the bound taken from the lanes' largest group maxima (and of the one-product sweep's margin), and the LDS
budget."""
import numpy as np


def test_plane_image_pieces_and_fragment_addresses():
    # planes [plane p][octet o][Np rows][16 B]; unit (kt, ch) = 128 keys x 64 channels; piece z = wave * 4 + i -> (half z >> 4,
    # plane (z >> 3) & 1, octet z & 7) lands at stage + z * 1024 (64 rows x 16 B); wave (qb, kh), lane (col, hf) reads row block a,
    # k-step s, plane p at kh * 16384 + p * 8192 + (2 s + hf) * 1024 + (32 a + col) * 16
    Np, Cp = 384, 128
    noct = Cp // 8
    for kt in range(Np // 128):
        for ch in range(Cp // 64):
            stage = {}
            for z in range(32):
                half, p, o = z >> 4, (z >> 3) & 1, z & 7
                for lane in range(64):
                    cell = (p * noct + ch * 8 + o) * Np + kt * 128 + half * 64 + lane           # uint4 index in the cloud's image
                    stage[z * 1024 + lane * 16] = (p, ch * 8 + o, kt * 128 + half * 64 + lane)
                    assert cell < 2 * noct * Np
            assert len(stage) == 2048                                                           # 32 KB of distinct 16-byte cells
            for kh in range(2):
                for hf in range(2):
                    for col in range(32):
                        for a in range(2):
                            for s in range(4):
                                for p in range(2):
                                    off = kh * 16384 + p * 8192 + (2 * s + hf) * 1024 + (32 * a + col) * 16
                                    plane, octet, row = stage[off]
                                    # the fragment a lane of v_mfma_f32_32x32x16_f16 wants: row 32 a + col of the wave's 64 keys,
                                    # channels 16 s + 8 hf .. + 7 of the chunk
                                    assert plane == p and octet == ch * 8 + 2 * s + hf
                                    assert row == kt * 128 + kh * 64 + 32 * a + col


def test_key_mapping_covers_every_key_once_per_query():
    # a query is served by four lanes: waves kh = 0, 1 (key halves of every 128-key tile) x half-waves hf = 0, 1; lane (kh, hf) holds
    # keys kt * 128 + kh * 64 + 32 a + (r & 3) + 8 (r >> 2) + 4 hf
    Np = 512
    keys = []
    for kt in range(Np // 128):
        for kh in range(2):
            for hf in range(2):
                for a in range(2):
                    for r in range(16):
                        keys.append(kt * 128 + kh * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hf)
    assert sorted(keys) == list(range(Np))
    # the aux rows a lane reads for row block a, register group g (4 consecutive keys): kh * 64 + 4 hf + 32 a + 8 g .. + 3
    for kh in range(2):
        for hf in range(2):
            for a in range(2):
                for g in range(4):
                    rows = [kh * 64 + 4 * hf + 32 * a + 8 * g + e for e in range(4)]
                    regs = [kh * 64 + 32 * a + ((4 * g + e) & 3) + 8 * ((4 * g + e) >> 2) + 4 * hf for e in range(4)]
                    assert rows == regs


def _bound(values, lanes_of, KC):
    """the kernel's bound for one query: every lane keeps group maxima (16 registers x 2 row blocks over all its tiles), offers its
    T = KC // 4 + 1 largest, and the KC-th largest of the 4 T offers is the bound"""
    T = KC // 4 + 1
    offers = []
    for lane in range(4):
        groups = {}
        for key, grp in lanes_of[lane]:
            groups[grp] = max(groups.get(grp, -np.inf), values[key])
        top = sorted(groups.values(), reverse=True)[:T]
        offers += top + [-np.inf] * (T - len(top))
    return sorted(offers, reverse=True)[KC - 1]


def test_bound_is_reached_by_at_least_k_candidates_and_few_more():
    rng = np.random.default_rng(3)
    N = 1024
    lanes_of = [[] for _ in range(4)]
    for kt in range(N // 128):
        for kh in range(2):
            for hf in range(2):
                for a in range(2):
                    for r in range(16):
                        lanes_of[kh * 2 + hf].append((kt * 128 + kh * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hf, (a, r)))
    counts = []
    for KC in (20, 32, 64):
        for trial in range(200):
            v = rng.standard_normal(N) if trial % 2 else np.round(rng.standard_normal(N), 1)       # with and without exact ties
            thr = _bound(v, lanes_of, KC)
            reach = int((v >= thr).sum())
            assert reach >= KC                                 # the group maxima are distinct candidates
            if KC == 20 and trial % 2:
                counts.append(reach)
    assert np.mean(counts) < 30 and max(counts) <= 64          # the list capacity of the K <= 20 class is 64


def test_one_product_margin_bounds_the_dropped_products():
    # sweep 0 uses h h' alone; the dropped m h' + h m' + m m' is below 2 (2^-11 + 2^-11 + 2^-22) |x_i| |x_j| < 0x1.1p-9 |x_i| |x_j|
    rng = np.random.default_rng(5)
    worst = 0.0
    for _ in range(200):
        C = int(rng.choice([64, 128, 256]))
        xi, xj = rng.standard_normal(C) * rng.choice([1e-3, 1.0, 50.0]), rng.standard_normal(C) * rng.choice([1e-3, 1.0, 50.0])

        def split(x):
            T = 12 - int(np.frexp(np.abs(x).max())[1])
            X = np.ldexp(x, T)
            h = X.astype(np.float16).astype(np.float64)
            return h, X - h, T
        hi, mi, Ti = split(xi)
        hj, mj, Tj = split(xj)
        exact = 2 * float(xi @ xj)
        one = 2 * float(hi @ hj) * 2.0 ** (-Ti - Tj)
        margin = float.fromhex("0x1.1p-9") * np.linalg.norm(xi) * np.linalg.norm(xj)
        worst = max(worst, abs(exact - one) / margin)
    assert worst < 1.0


def test_lds_budget():
    stage, dump = 33792, 8 * 17 * 64 * 4
    small = 128 * 4 + 132 * 4
    lists = 128 * 65 * 4 + 128 * 66 * 2
    assert 2 * stage + dump + small + lists <= 160 * 1024          # the K <= 20 class (lists in LDS)
    for T in (6, 9, 17):                                              # the bound exchange lives in the idle stages
        assert (4 * T + 1) * 128 * 4 <= 2 * stage
    assert 32 * 192 * 8 <= 2 * stage                                  # a batch of 32 workspace lists (K = 64) staged for ranking


def test_rows_as_weight_operand_layout():
    # models/_rows.py, round 6: y [rows, Cout] = x W^T through l3d_pointwise_conv_f16 with the batch's rows as the kernel's WEIGHT operand.
    # The kernel writes y[b][co][n] at (b Cout + co) N + n for weight row co and activation row n; with B = 1, "Cout" = rows and "N" = the
    # layer's Cout that offset is row * Cout + c: row-major [rows, Cout].  The two-plane form reads the weight image's planes 0 (H) and
    # 2 (M) and the scale behind plane 2: where l3d_split_f16_operand (kind 1) puts h, m and 2^-T.
    rows, Cout, Cin = 512, 256, 64
    rng = np.random.default_rng(0)
    x, W = rng.standard_normal((rows, Cin)), rng.standard_normal((Cout, Cin))
    y_kernel = np.empty(rows * Cout)
    for co in range(0, rows, 37):                       # the kernel's indices: weight row co (a batch row), activation row n (an output channel)
        for n in range(0, Cout, 11):
            y_kernel[(0 * rows + co) * Cout + n] = x[co] @ W[n]
            assert abs(y_kernel[co * Cout + n] - (x @ W.T)[co, n]) < 1e-12
    plane = lambda r, c: ((c + 7) // 8) * r * 16           # common.h: l3d_f16_plane_bytes
    pb = plane(rows, Cin)
    weight_image_bytes, act_image_bytes = 3 * pb + 16, 2 * pb + 16
    h_off, m_off, inv_off = 0, 2 * pb, 3 * pb              # kind 1: the slots the two-plane kernel reads (wH, wM = wp + 2 wpb, winv = wp + 3 wpb)
    assert inv_off + 16 == weight_image_bytes and m_off + pb == inv_off
    assert (0, pb, 2 * pb) == (0, pb, act_image_bytes - 16)   # kind 0: h | m | 2^-T, an ordinary activation image
    # a cell: 8 consecutive k of one row, octet-major: [k / 8][row][8 fp16]
    cell = lambda row, k: ((k // 8) * rows + row) * 16 + (k % 8) * 2
    seen = {cell(r, k) for r in range(rows) for k in range(Cin)}
    assert len(seen) == rows * Cin and max(seen) + 2 == pb
