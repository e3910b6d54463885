"""CPU model of featknn.hip's index bookkeeping (no GPU): the split buffer's region addressing, the
key <-> (lane, accumulator tile, register) mapping of the 32x32x16 MFMA output the top-k epilogue relies on,
the sentinel row of the candidate loop, and argument validation of the new entry points."""
import numpy as np


def test_split_buffer_region_addressing():
    # featknn_split_kernel writes octet c8 = 2*kc16 + kg of plane p at ((kc16*3 + p)*2 + kg)*Np + n;
    # featknn_kernel reads region r of a 32-channel chunk ch at (ch*12 + r)*Np + key with r = (s*3 + p)*2 + kg
    Np = 384
    for C in (32, 64, 96, 256):
        seen = set()
        for ch in range(C // 32):
            for s in range(2):
                for p in range(3):
                    for kg in range(2):
                        r = (s * 3 + p) * 2 + kg
                        kc16 = 2 * ch + s
                        kernel_off = (ch * 12 + r) * Np
                        split_off = ((kc16 * 3 + p) * 2 + kg) * Np
                        assert kernel_off == split_off
                        seen.add(kernel_off)
        assert len(seen) == (C // 16) * 6                                  # every (kc16, plane, kg) region exactly once
        assert max(seen) + Np == (C // 16) * 6 * Np                        # and they tile the per-cloud image densely


def test_key_mapping_covers_every_key_once_per_query():
    # lane l owns query column l & 31 and, of each 128-key tile, the keys 32 a + (r & 3) + 8 (r >> 2) + 4 (l >> 5)
    for i in range(32):
        keys = []
        for h in range(2):
            for a in range(4):
                for r in range(16):
                    keys.append(32 * a + (r & 3) + 8 * (r >> 2) + 4 * h)
        assert sorted(keys) == list(range(128))
    # within a lane the keys ascend with (a, r): strict '>' insertion keeps the lower index under ties
    for h in range(2):
        seq = [32 * a + (r & 3) + 8 * (r >> 2) + 4 * h for a in range(4) for r in range(16)]
        assert seq == sorted(seq)


def test_sentinel_row_pop_order():
    # candidate loop: bit positions are popped lowest first, exhausted lanes read row 16 (= -inf, a no-op insertion)
    rng = np.random.default_rng(0)
    for _ in range(100):
        mask = int(rng.integers(0, 1 << 16))
        m, order = mask, []
        while True:
            bp = min(((m & -m).bit_length() - 1) if m else 16, 16)
            m &= m - 1
            if bp == 16:
                break
            order.append(bp)
        assert order == [b for b in range(16) if mask >> b & 1]


def test_new_entry_points_validate_arguments():
    from learning3d_amd import _lib
    l = _lib.lib()
    assert l.l3d_knn_feature(None, 1, 64, 128, 8, None, None, None) == -1
    assert l.l3d_edge_gather_max(None, None, 1, 64, 128, 20, 0, None, 0, None) == -1
    assert l.l3d_scatter_add_det(None, None, None, 1, 1, 1, 1, 1, None, None, None) == -1
    assert l.l3d_group_concat2(None, None, None, None, None, 1, 1, 1, 1, 1, 0, 0, None, None) == -1
    assert l.l3d_add_transposed(None, None, 1, 1, 1, None, None) == -1
    assert l.l3d_three_interpolate_concat(1, 1, 1, 1, None, None, None, None, 0, None, None) == -1
    # split planes + -|x|^2, and (round 5) the sorted K-lists of the key-range parts where a CU would otherwise hold one workgroup:
    # 2 x 3 query tiles -> 3 parts (one per key tile), lists at the longest length (64), 8 bytes per entry
    assert l.l3d_knn_feature_workspace_bytes(2, 64, 300) == 2 * 64 * 384 * 6 + 2 * 384 * 4 + 2 * 300 * 3 * 64 * 8
    assert l.l3d_knn_feature_workspace_bytes(32, 64, 1024) == 32 * 64 * 1024 * 6 + 32 * 1024 * 4          # C < 128, 256 query tiles: no split
