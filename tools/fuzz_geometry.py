#!/usr/bin/env python3
"""Random-shape sweep of the geometry ops against the CPU oracle, bit for bit where the fixed-shape
tests are (indices, distances), including the awkward shapes: fewer points than lanes, K larger than a ball's population, empty
balls, ragged clouds, duplicated points.  Test infrastructure (it imports oracle/): run on a GPU box, `python tools/fuzz_geometry.py
[seed] [rounds]`."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
import learning3d_amd.utils as U  # noqa: E402
from learning3d_amd.losses.chamfer_distance import ChamferDistanceFunction  # noqa: E402
from learning3d_amd.utils import pointnet2_utils as P  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(seed)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
count = {}


def ok(name):
    count[name] = count.get(name, 0) + 1


def cloud(B, N, kind):
    if kind == 0:
        x = rng.uniform(-1, 1, (B, N, 3))
    elif kind == 1:
        x = np.clip(rng.standard_normal((B, N, 3)), -2, 2)
    elif kind == 2:                                                   # duplicated points: exact distance ties
        base = rng.uniform(-1, 1, (B, max(1, N // 3), 3))
        x = np.concatenate([base] * 3 + [rng.uniform(-1, 1, (B, N, 3))], 1)[:, :N]
    else:                                                             # lattice: many equal distances
        x = rng.integers(0, 5, (B, N, 3)) / 4.0
    return x.astype(np.float32)


with torch.no_grad():
    for it in range(rounds):
        B = int(rng.integers(1, 4)); N = int(rng.choice([1, 2, 7, 33, 64, 65, 200, 777, 1500, 3000])); kind = int(rng.integers(0, 4))
        xyz = cloud(B, N, kind)
        # kNN graph (a1): ties compared modulo equal ranking values
        k = int(rng.integers(1, min(N, 24) + 1))
        idx = U.knn(dev(xyz.transpose(0, 2, 1)), k).cpu().numpy()
        oracle.assert_knn_equal_modulo_ties(idx, oracle.knn(xyz, k), xyz); ok("knn")
        # FPS (a6 / K12)
        S = int(rng.integers(1, N + 1))
        got = P.furthest_point_sample(dev(xyz), S).cpu().numpy()
        assert np.array_equal(got, oracle.furthest_point_sampling(xyz, S)), ("fps", B, N, S, kind); ok("fps")
        new_xyz = np.ascontiguousarray(np.take_along_axis(xyz, got[:, :, None].astype(np.int64).repeat(3, 2), 1))
        # ball query (K7): first K in index order, padded with the first hit, empty -> zeros
        r = float(rng.choice([0.05, 0.3, 1.0, 5.0])); K = int(rng.choice([1, 4, 16, 33, 64]))
        got = P.ball_query(r, K, dev(xyz), dev(new_xyz)).cpu().numpy()
        assert np.array_equal(got, oracle.ball_query(r, K, xyz, new_xyz)), ("ball_query", B, N, S, r, K, kind); ok("ball_query")
        # torch-twin query_ball_point (a4): empty ball -> N
        got = U.query_ball_point(r, K, dev(xyz), dev(new_xyz)).cpu().numpy()
        assert np.array_equal(got, oracle.query_ball_point(r, K, xyz, new_xyz)), ("query_ball_point", B, N, S, r, K, kind); ok("query_ball_point")
        # grouping / gather (K9, K11)
        C = int(rng.integers(1, 40)); feat = rng.standard_normal((B, C, N)).astype(np.float32)
        bidx = oracle.ball_query(r, K, xyz, new_xyz)
        assert np.array_equal(P.grouping_operation(dev(feat), dev(bidx)).cpu().numpy(), oracle.group_points(feat, bidx)); ok("group_points")
        fidx = oracle.furthest_point_sampling(xyz, S)
        assert np.array_equal(P.gather_operation(dev(feat), dev(fidx)).cpu().numpy(), oracle.gather_points(feat, fidx)); ok("gather_points")
        # kNN between two clouds (K13) and 3-NN (K14): ties by lowest index
        M = int(rng.choice([3, 5, 64, 300, 1100])); other = cloud(B, M, int(rng.integers(0, 4)))
        kk = int(rng.integers(1, min(M, 64) + 1))
        d2, ii = P.knn(kk, dev(xyz), dev(other))
        od2, oi = oracle.knn_pair(kk, xyz, other)
        assert np.array_equal(ii.cpu().numpy(), oi) and np.array_equal(d2.cpu().numpy(), od2), ("knn_pair", B, N, M, kk); ok("knn_pair")
        # the wave-per-query selection kernel's range (knn_select.hip: >= 1024 candidates or k > 32, k up to 200), few queries
        Mb = int(rng.choice([1024, 1100, 2500, 4097, 8192])); big = cloud(B, Mb, int(rng.integers(0, 4)))
        kb = int(rng.choice([1, 3, 16, 33, 64, 100, 200])); qn = xyz[:, :min(N, 150)]
        d2b, iib = P.knn(kb, dev(qn), dev(big))
        od2b, oib = oracle.knn_pair(kb, np.ascontiguousarray(qn), big)
        assert np.array_equal(iib.cpu().numpy(), oib) and np.array_equal(d2b.cpu().numpy(), od2b), ("knn_pair big", B, N, Mb, kb); ok("knn_pair_select")
        # the four-slot kernel (knn_small.hip, k <= 4) named explicitly: the automatic choice takes it only from 65 536 queries up
        from learning3d_amd._lib import check, lib, ptr, stream_ptr
        ks = int(rng.integers(1, 5)); qd, cd = dev(xyz), dev(other)
        sd = torch.empty((B, N, ks), device="cuda"); si = torch.empty((B, N, ks), dtype=torch.int32, device="cuda")
        check(lib().l3d_knn_variant(B, N, M, ks, ptr(qd), ptr(cd), ptr(sd), ptr(si), 3, stream_ptr()), "l3d_knn_variant")
        osd, osi = oracle.knn_pair(ks, xyz, other)
        if M >= ks:                                       # (slots beyond the candidate count: (+inf, 0) here, unspecified in the oracle)
            assert np.array_equal(si.cpu().numpy(), osi) and np.array_equal(np.sqrt(sd.cpu().numpy()), osd), ("knn_small", B, N, M, ks); ok("knn_small")
        d3, i3 = P.three_nn(dev(xyz), dev(other))
        od3, oi3 = oracle.three_nn(xyz, other)
        assert np.array_equal(i3.cpu().numpy(), oi3) and np.allclose(d3.cpu().numpy(), od3, rtol=0, atol=0), ("three_nn", B, N, M); ok("three_nn")
        w = rng.uniform(0, 1, (B, N, 3)).astype(np.float32); w /= w.sum(-1, keepdims=True)
        fm = rng.standard_normal((B, C, M)).astype(np.float32)
        assert np.array_equal(P.three_interpolate(dev(fm), dev(oi3), dev(w)).cpu().numpy(), oracle.three_interpolate(fm, oi3, w)); ok("three_interpolate")
        # Chamfer (a9): distances bit-exact
        d1, d2_ = ChamferDistanceFunction.apply(dev(xyz), dev(other))
        o1, o2, _, _ = oracle.chamfer_forward(xyz, other)
        assert np.array_equal(d1.cpu().numpy(), o1) and np.array_equal(d2_.cpu().numpy(), o2), ("chamfer", B, N, M); ok("chamfer")
        # ... and the matrix-core-ranked kernel (chamfer_mfma.hip, l3d_chamfer_forward_variant 3; the default from 2^24 pairs per cloud):
        # distances AND indices against the oracle's strict-'<' scan (lowest index on ties)
        from learning3d_amd._lib import check as _chk, lib as _lib, ptr as _ptr, stream_ptr as _sp
        ta, tb = dev(xyz), dev(other)
        m1, m2 = torch.empty((B, N), device="cuda"), torch.empty((B, M), device="cuda")
        j1, j2 = torch.empty((B, N), dtype=torch.int32, device="cuda"), torch.empty((B, M), dtype=torch.int32, device="cuda")
        _chk(_lib().l3d_chamfer_forward_variant(_ptr(ta), _ptr(tb), B, N, M, _ptr(m1), _ptr(m2), _ptr(j1), _ptr(j2), 3, _sp()), "chamfer mfma")
        _, _, oi1, oi2 = oracle.chamfer_forward(xyz, other)
        assert np.array_equal(m1.cpu().numpy(), o1) and np.array_equal(m2.cpu().numpy(), o2), ("chamfer_mfma dist", B, N, M)
        assert np.array_equal(j1.cpu().numpy(), oi1) and np.array_equal(j2.cpu().numpy(), oi2), ("chamfer_mfma idx", B, N, M); ok("chamfer_mfma")
        # square_distance / index_points / knn_point (a3, a5, a7)
        if N * M <= 400000:
            assert np.array_equal(U.square_distance(dev(xyz), dev(other)).cpu().numpy(), oracle.square_distance(xyz, other)); ok("square_distance")
            v, i_ = U.knn_point(min(kk, N), dev(xyz), dev(other))
            ov, oi_ = oracle.knn_point(min(kk, N), xyz, other)
            assert np.array_equal(i_.cpu().numpy(), oi_) and np.allclose(v.cpu().numpy(), ov, rtol=0, atol=1e-7), ("knn_point", B, N, M); ok("knn_point")
        print(f"round {it}: B {B} N {N} S {S} M {M} kind {kind} ok", flush=True)
print("fuzz_geometry OK:", ", ".join(f"{k} x{v}" for k, v in count.items()))
