"""GPU parity tests: the HIP kernels (through the C ABI and the reference-shaped Python API) against
the CPU oracle and the golden vectors generated from the real reference.  Run with -m gpu on MI355X.

Bars (BASELINE.json): kNN / ball-query / FPS / grouping indices bit-identical (modulo exact fp32
ties, where torch.topk's order is unspecified); Chamfer distances bit-exact (integer idx exact);
SVD rotations and shared-MLP features within 1e-5 / rtol 1e-4 fp32."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def rand(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * (hi - lo) + lo).numpy()


@pytest.fixture(scope="module")
def U():
    import learning3d_amd.utils as U
    return U


# ------------------------------------------------------------------------------------------- kNN
def test_knn_golden_c2_distribution(U, golden):
    g = golden("knn_n1024_k20")
    x = dev(g["xyz"])                                     # [B,N,3]
    idx = U.knn(x.permute(0, 2, 1), 20).cpu().numpy()
    assert idx.dtype == np.int64 and idx.shape == (2, 1024, 20)
    assert np.array_equal(idx, oracle.knn(g["xyz"], 20))              # bit-identical to the oracle
    oracle.assert_knn_equal_modulo_ties(idx, g["idx"], g["xyz"])      # and to the reference modulo ties


def test_knn_golden_small_and_add_one(U, golden):
    g = golden("knn_n200_k7")
    x = dev(g["xyz"]).permute(0, 2, 1)
    assert np.array_equal(U.knn(x, 7).cpu().numpy(), g["idx"].astype(np.int64))
    assert np.array_equal(U.knn(x, 7, add_one_to_k=True).cpu().numpy(), g["idx_plus1"].astype(np.int64))


@pytest.mark.parametrize("B,N,k", [(1, 20, 20), (3, 65, 5), (2, 333, 33), (2, 1500, 64), (1, 300, 130)])
def test_knn_vs_oracle_ragged_sizes(U, B, N, k):
    xyz = rand((B, N, 3), 100 + N, -1, 1)
    idx = U.knn(dev(xyz).permute(0, 2, 1), k).cpu().numpy()
    assert np.array_equal(idx, oracle.knn(xyz, k))


def test_knn_full_baseline_size_properties(U):
    """B=32, N=1024, k=20 (BASELINE config 2, the bench shape): ALL 32 clouds bit-identical to the oracle
    (which is pinned to the reference's golden on this very distribution), plus the structural properties."""
    xyz = rand((32, 1024, 3), 0)
    idx = U.knn(dev(xyz).permute(0, 2, 1), 20).cpu().numpy()
    assert (idx[:, :, 0] == np.arange(1024)[None]).mean() > 0.999
    assert all(len(np.unique(r)) == 20 for r in idx[0])
    assert np.array_equal(idx, oracle.knn(xyz, 20))


def test_knn_rejects_bad_k(U):
    x = dev(rand((1, 8, 3), 1)).permute(0, 2, 1)
    with pytest.raises(RuntimeError):
        U.knn(x, 9)


def test_graph_feature_golden_and_layout(U, golden):
    g = golden("graph_feature_n96")
    f = U.get_graph_feature(dev(g["xyz"]).permute(0, 2, 1), k=20)
    assert f.shape == (2, 6, 96, 20)
    assert f.stride() == (96 * 20 * 6, 1, 120, 6)          # same non-contiguous view as the reference
    assert np.array_equal(f.cpu().numpy(), g["feat"])


# -------------------------------------------------------------------------- torch-level primitives
def test_square_distance_bit_exact(U, golden):
    g = golden("square_distance")
    d = U.square_distance(dev(g["src"]), dev(g["dst"])).cpu().numpy()
    assert np.array_equal(d, g["dist"])


def test_query_ball_point_golden(U, golden):
    g = golden("query_ball_point")
    idx, cnt = U.query_ball_point(float(g["radius"]), int(g["nsample"]), dev(g["xyz"]), dev(g["new_xyz"]), get_cnt=True)
    assert idx.dtype == torch.int64
    assert np.array_equal(idx.cpu().numpy(), g["idx"]) and np.array_equal(cnt.cpu().numpy(), g["cnt"])
    far = U.query_ball_point(float(g["radius"]), int(g["nsample"]), dev(g["xyz"]), dev(g["far"])).cpu().numpy()
    assert (far == g["xyz"].shape[1]).all()                 # empty ball -> N, like the reference


def test_query_ball_point_vs_oracle_large(U):
    xyz = np.clip(np.random.default_rng(0).standard_normal((2, 3000, 3)), -2, 2).astype(np.float32)
    new = xyz[:, ::7][:, :300].copy()
    idx, cnt = U.query_ball_point(0.5, 16, dev(xyz), dev(new), get_cnt=True)
    oi, oc = oracle.query_ball_point(0.5, 16, xyz, new, get_cnt=True)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(cnt.cpu().numpy(), oc)


def test_index_points_golden(U, golden):
    g = golden("index_points")
    o2 = U.index_points(dev(g["points"]), dev(g["idx2"], torch.int64)).cpu().numpy()
    o1 = U.index_points(dev(g["points"]), dev(g["idx1"], torch.int64)).cpu().numpy()
    assert np.array_equal(o2, g["out2"]) and np.array_equal(o1, g["out1"])


def test_farthest_point_sample_golden(U, golden):
    g = golden("farthest_point_sample")
    idx = U.farthest_point_sample(dev(g["xyz"]), g["idx"].shape[1], start_with_first_point=True)
    assert idx.dtype == torch.int64 and np.array_equal(idx.cpu().numpy(), g["idx"])
    r = U.farthest_point_sample(dev(g["xyz"]), 16).cpu().numpy()          # random start: valid + distinct
    assert r.min() >= 0 and r.max() < 512 and all(len(np.unique(x)) == 16 for x in r)


def test_knn_point_golden(U, golden):
    g = golden("knn_point")
    val, idx = U.knn_point(g["idx"].shape[2], dev(g["pos1"]), dev(g["pos2"]))
    assert np.array_equal(idx.cpu().numpy(), g["idx"])
    np.testing.assert_allclose(val.cpu().numpy(), g["val"], rtol=0, atol=1e-7)


# ----------------------------------------------------------------------------------------- Chamfer
def test_chamfer_golden_forward_backward(golden):
    from learning3d_amd.losses.chamfer_distance import ChamferDistanceFunction, ChamferDistanceLoss, chamfer_distance
    g = golden("chamfer")
    a, b = dev(g["xyz1"]).requires_grad_(), dev(g["xyz2"]).requires_grad_()
    d1, d2 = ChamferDistanceFunction.apply(a, b)
    assert np.array_equal(d1.detach().cpu().numpy(), g["dist1"])          # bit-exact vs the reference C++
    assert np.array_equal(d2.detach().cpu().numpy(), g["dist2"])
    (d1 * dev(g["graddist1"])).sum().add((d2 * dev(g["graddist2"])).sum()).backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), g["gradxyz1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(b.grad.cpu().numpy(), g["gradxyz2"], rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        loss = ChamferDistanceLoss()(dev(g["xyz1"]), dev(g["xyz2"]))
    assert abs(float(loss) - float(g["loss_ext"])) < 1e-6
    loss_g = chamfer_distance(a, b)                                        # autograd branch
    assert abs(float(loss_g) - float(g["loss_ext"])) < 1e-6


def test_chamfer_packed_kernel_bit_identical():
    """The packed-fp32 forward (two queries per lane, argmin per chunk of 8 resolved at the end) against the
    per-candidate kernels and the oracle: identical distances AND indices, including exact ties (duplicated
    points -> the first index must win), ragged sizes and a collapsed cloud."""
    import ctypes
    from learning3d_amd._lib import lib, check, ptr, stream_ptr
    rng = np.random.default_rng(33)
    cases = [(rng.uniform(0, 1, (2, 77, 3)), rng.uniform(0, 1, (2, 130, 3))),
             (rng.uniform(0, 1, (3, 1024, 3)), rng.uniform(0, 1, (3, 1000, 3))),
             (rng.uniform(0, 1, (1, 2500, 3)), rng.uniform(0, 1, (1, 4099, 3)))]
    dup = rng.uniform(0, 1, (1, 300, 3)); cases.append((rng.uniform(0, 1, (1, 200, 3)), np.concatenate([dup, dup, dup], 1)))
    cases.append((rng.uniform(0, 1, (2, 500, 3)), np.full((2, 700, 3), 0.25)))
    for a, b in cases:
        a, b = a.astype(np.float32), b.astype(np.float32)
        B, N, M = a.shape[0], a.shape[1], b.shape[1]
        ta, tb = dev(a), dev(b)
        res = {}
        for m in (0, 2):
            d1 = torch.empty(B, N, device="cuda"); d2 = torch.empty(B, M, device="cuda")
            i1 = torch.empty(B, N, dtype=torch.int32, device="cuda"); i2 = torch.empty(B, M, dtype=torch.int32, device="cuda")
            check(lib().l3d_chamfer_forward_variant(ptr(ta), ptr(tb), B, N, M, ptr(d1), ptr(d2), ptr(i1), ptr(i2), m,
                                                    stream_ptr()), "cd")
            res[m] = [x.cpu().numpy() for x in (d1, d2, i1, i2)]
        for x, y in zip(res[0], res[2]):
            assert np.array_equal(x, y), (B, N, M)
        o = oracle.chamfer_forward(a[:1], b[:1])
        for x, y in zip(res[2], o):
            assert np.array_equal(x[:1], y), (B, N, M)


def test_chamfer_backward_sorted_kernel_bit_identical():
    """The sorted-list backward (partner selections sorted by target in LDS, one binary search per point) against the scan
    kernel and the oracle's restatement of chamfer_distance.cpp:138-176: the same bits, including many partners per point
    (duplicated and collapsed clouds), ragged and non-power-of-two sizes, and PCN's 16384-point clouds."""
    from learning3d_amd._lib import lib, check, ptr, stream_ptr
    rng = np.random.default_rng(34)
    cases = [(rng.uniform(0, 1, (2, 77, 3)), rng.uniform(0, 1, (2, 130, 3))),
             (rng.uniform(0, 1, (3, 1024, 3)), rng.uniform(0, 1, (3, 1000, 3))),
             (rng.uniform(0, 1, (1, 2500, 3)), rng.uniform(0, 1, (1, 4099, 3))),
             (rng.uniform(0, 1, (2, 16384, 3)), rng.uniform(0, 1, (2, 1024, 3))),
             (rng.uniform(0, 1, (1, 32768, 3)), rng.uniform(0, 1, (1, 20000, 3))),      # the sorted kernel's largest clouds: 128 KB of LDS
             (rng.uniform(0, 1, (1, 1, 3)), rng.uniform(0, 1, (1, 5, 3)))]
    dup = rng.uniform(0, 1, (1, 300, 3)); cases.append((rng.uniform(0, 1, (1, 200, 3)), np.concatenate([dup, dup, dup], 1)))
    cases.append((rng.uniform(0, 1, (2, 500, 3)), np.full((2, 700, 3), 0.25)))
    for a, b in cases:
        a, b = a.astype(np.float32), b.astype(np.float32)
        B, N, M = a.shape[0], a.shape[1], b.shape[1]
        ta, tb = dev(a), dev(b)
        d1 = torch.empty(B, N, device="cuda"); d2 = torch.empty(B, M, device="cuda")
        i1 = torch.empty(B, N, dtype=torch.int32, device="cuda"); i2 = torch.empty(B, M, dtype=torch.int32, device="cuda")
        check(lib().l3d_chamfer_forward(ptr(ta), ptr(tb), B, N, M, ptr(d1), ptr(d2), ptr(i1), ptr(i2), stream_ptr()), "cd")
        gd1 = dev(rng.normal(size=(B, N)).astype(np.float32)); gd2 = dev(rng.normal(size=(B, M)).astype(np.float32))
        res = {}
        for v in (0, 2, 1):
            g1 = torch.full((B, N, 3), float("nan"), device="cuda"); g2 = torch.full((B, M, 3), float("nan"), device="cuda")
            check(lib().l3d_chamfer_backward_variant(ptr(ta), ptr(tb), B, N, M, ptr(gd1), ptr(gd2), ptr(i1), ptr(i2), ptr(g1), ptr(g2),
                                                     v, stream_ptr()), "cd bwd")
            res[v] = [g1.cpu().numpy(), g2.cpu().numpy()]
        for v in (2, 1):
            for x, y in zip(res[0], res[v]):
                assert np.array_equal(x, y), (B, N, M, v)
        if N * M <= 4099 * 2500:
            o1, o2 = oracle.chamfer_backward(a[:1], b[:1], gd1.cpu().numpy()[:1], gd2.cpu().numpy()[:1],
                                             i1.cpu().numpy()[:1], i2.cpu().numpy()[:1])
            np.testing.assert_allclose(res[2][0][:1], o1, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(res[2][1][:1], o2, rtol=1e-5, atol=1e-6)
    # beyond the LDS-resident list: variant 2 declines, auto takes the scan kernel
    a, b = dev(rand((1, 40000, 3), 5)), dev(rand((1, 64, 3), 6))
    d1 = torch.empty(1, 40000, device="cuda"); d2 = torch.empty(1, 64, device="cuda")
    i1 = torch.empty(1, 40000, dtype=torch.int32, device="cuda"); i2 = torch.empty(1, 64, dtype=torch.int32, device="cuda")
    check(lib().l3d_chamfer_forward(ptr(a), ptr(b), 1, 40000, 64, ptr(d1), ptr(d2), ptr(i1), ptr(i2), stream_ptr()), "cd")
    g1 = torch.empty(1, 40000, 3, device="cuda"); g2 = torch.empty(1, 64, 3, device="cuda")
    assert lib().l3d_chamfer_backward_variant(ptr(a), ptr(b), 1, 40000, 64, ptr(d1), ptr(d2), ptr(i1), ptr(i2), ptr(g1), ptr(g2),
                                              2, stream_ptr()) != 0
    check(lib().l3d_chamfer_backward(ptr(a), ptr(b), 1, 40000, 64, ptr(d1), ptr(d2), ptr(i1), ptr(i2), ptr(g1), ptr(g2), stream_ptr()), "cd bwd")
    assert torch.isfinite(g1).all() and torch.isfinite(g2).all()


def test_chamfer_loss_local_equals_partials_plus_combine():
    from learning3d_amd.losses.chamfer_distance import chamfer_loss_local, chamfer_partials, chamfer_combine
    rng = np.random.default_rng(44)
    for (B, N, M) in [(32, 1024, 1024), (3, 77, 130), (1, 5, 2)]:
        d1, d2 = dev(rng.uniform(0, 2, (B, N)).astype(np.float32)), dev(rng.uniform(0, 2, (B, M)).astype(np.float32))
        b = chamfer_combine(chamfer_partials(d1, d2))
        for _ in range(3):                                   # the multi-workgroup kernel re-arms its ticket: call it repeatedly
            a = chamfer_loss_local(d1, d2)
            assert abs(a.item() - b.item()) <= 1.2e-7 * max(1.0, abs(b.item()))   # fp64 sums in another order, rounded to fp32
        want = (np.sqrt(d1.cpu().numpy().astype(np.float64)).mean() + np.sqrt(d2.cpu().numpy().astype(np.float64)).mean()) / 2
        assert abs(a.item() - want) < 1e-6


def test_chamfer_forward_loss_one_launch_equals_search_plus_tail():
    """l3d_chamfer_forward_loss (round 6: the loss tail inside the search kernel's launch -- write-through slots, a ticket, the last
    workgroup adds them): distances and indices bit for bit those of l3d_chamfer_forward, partial sums and loss those of the two-step
    route (fp64 sums in another order), against the oracle's loss; config 2's shape, ragged and unequal clouds (workgroups beyond the
    shorter cloud still draw a ticket), the two-launch fallback for tiny and for large clouds, repeated calls (the ticket re-arms)."""
    from learning3d_amd.losses.chamfer_distance import (ChamferDistanceLoss, chamfer_forward_loss, chamfer_partials, chamfer_combine)
    from learning3d_amd._lib import lib, check, ptr, stream_ptr
    for (B, N, M, seed) in [(32, 1024, 1024, 3), (32, 1024, 700, 4), (16, 300, 2000, 5), (2, 77, 130, 1), (1, 5, 2, 6), (1, 4096, 4100, 7)]:
        a, b = rand((B, N, 3), seed), rand((B, M, 3), seed + 50)
        ta, tb = dev(a), dev(b)
        d1 = torch.empty(B, N, device="cuda"); d2 = torch.empty(B, M, device="cuda")
        i1 = torch.empty(B, N, dtype=torch.int32, device="cuda"); i2 = torch.empty(B, M, dtype=torch.int32, device="cuda")
        check(lib().l3d_chamfer_forward(ptr(ta), ptr(tb), B, N, M, ptr(d1), ptr(d2), ptr(i1), ptr(i2), stream_ptr()), "cd")
        part_ref = chamfer_partials(d1, d2)
        loss_ref = chamfer_combine(part_ref)
        for rep in range(3):
            loss, part, e1, e2, j1, j2 = chamfer_forward_loss(ta, tb, want="all")
            assert torch.equal(e1, d1) and torch.equal(e2, d2) and torch.equal(j1, i1) and torch.equal(j2, i2), (B, N, M, rep)
            pr, pg = part_ref.cpu().numpy(), part.cpu().numpy()
            assert pg[2] == B * N and pg[3] == B * M
            np.testing.assert_allclose(pg[:2], pr[:2], rtol=1e-13, atol=0)
            assert abs(loss.item() - loss_ref.item()) <= 1.2e-7 * max(1.0, abs(loss_ref.item())), (B, N, M, rep)
        want = (np.sqrt(d1.cpu().numpy().astype(np.float64)).mean() + np.sqrt(d2.cpu().numpy().astype(np.float64)).mean()) / 2
        assert abs(loss.item() - want) < 1e-6
        with torch.no_grad():
            assert ChamferDistanceLoss()(ta, tb).item() == loss.item()          # the module's no-grad forward IS this entry point
    o = oracle.chamfer_forward(a[:1], b[:1])
    assert np.array_equal(e1.cpu().numpy()[:1], o[0]) and np.array_equal(j2.cpu().numpy()[:1], o[3])


def test_chamfer_idx_and_ragged_vs_oracle():
    from learning3d_amd._lib import lib, check, ptr, stream_ptr
    for (B, N, M, seed) in [(2, 77, 130, 1), (1, 2500, 64, 2), (32, 1024, 1024, 3)]:
        a, b = rand((B, N, 3), seed), rand((B, M, 3), seed + 50)
        ta, tb = dev(a), dev(b)
        d1 = torch.empty(B, N, device="cuda"); d2 = torch.empty(B, M, device="cuda")
        i1 = torch.empty(B, N, dtype=torch.int32, device="cuda"); i2 = torch.empty(B, M, dtype=torch.int32, device="cuda")
        check(lib().l3d_chamfer_forward(ptr(ta), ptr(tb), B, N, M, ptr(d1), ptr(d2), ptr(i1), ptr(i2), stream_ptr()), "cd")
        sel = [0, B - 1]
        o1, o2, oi1, oi2 = oracle.chamfer_forward(a[sel], b[sel])
        assert np.array_equal(d1.cpu().numpy()[sel], o1) and np.array_equal(d2.cpu().numpy()[sel], o2)
        assert np.array_equal(i1.cpu().numpy()[sel], oi1) and np.array_equal(i2.cpu().numpy()[sel], oi2)


def test_chamfer_large_properties():
    """PCN-sized Chamfer slice (2048 x 16384): symmetric consistency d1[i] == |a_i - b_idx1[i]|^2."""
    from learning3d_amd.losses.chamfer_distance import ChamferDistanceFunction
    a, b = dev(rand((2, 2048, 3), 7, -0.5, 0.5)), dev(rand((2, 16384, 3), 8, -0.5, 0.5))
    d1, d2 = ChamferDistanceFunction.apply(a, b)
    ref1 = torch.cdist(a.double(), b.double()).min(dim=2)[0] ** 2
    ref2 = torch.cdist(b.double(), a.double()).min(dim=2)[0] ** 2
    np.testing.assert_allclose(d1.cpu().numpy(), ref1.cpu().numpy(), rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(d2.cpu().numpy(), ref2.cpu().numpy(), rtol=1e-4, atol=1e-9)


# ---------------------------------------------------------------------------- PointNet++ native ops
def test_gather_type_backward_is_deterministic_and_correct():
    """grouping / gather / three_interpolate backward through l3d_scatter_add_det (stable
    sort of the indices + per-target sums in ascending entry order) instead of fp32 atomics: the same bits on
    every run, equal to an fp64 index_add within fp32 rounding, and to the atomic kernels within tolerance."""
    from learning3d_amd.utils import pointnet2_utils as P
    rng = np.random.default_rng(123)
    B, C, N, S, K = 3, 20, 500, 64, 16
    feat = dev(rng.standard_normal((B, C, N)).astype(np.float32))
    idx_g = dev(rng.integers(0, 40, (B, S, K)).astype(np.int32))                 # heavy collisions: 40 targets
    idx_s = dev(rng.integers(0, N, (B, S)).astype(np.int32))
    idx_3 = dev(rng.integers(0, S, (B, N, 3)).astype(np.int32))
    w3 = dev(rng.uniform(0, 1, (B, N, 3)).astype(np.float32))
    known = dev(rng.standard_normal((B, C, S)).astype(np.float32))

    def run(det):
        P.DETERMINISTIC_BACKWARD = det
        outs = []
        f = feat.clone().requires_grad_()
        go = dev(np.random.default_rng(5).standard_normal((B, C, S, K)).astype(np.float32))
        (P.grouping_operation(f, idx_g) * go).sum().backward(); outs.append(f.grad.clone())
        f = feat.clone().requires_grad_()
        go = dev(np.random.default_rng(6).standard_normal((B, C, S)).astype(np.float32))
        (P.gather_operation(f, idx_s) * go).sum().backward(); outs.append(f.grad.clone())
        f = known.clone().requires_grad_()
        go = dev(np.random.default_rng(7).standard_normal((B, C, N)).astype(np.float32))
        (P.three_interpolate(f, idx_3, w3) * go).sum().backward(); outs.append(f.grad.clone())
        return [o.cpu().numpy() for o in outs]

    try:
        d1, d2, atomic = run(True), run(True), run(False)
    finally:
        P.DETERMINISTIC_BACKWARD = True
    for a, b_ in zip(d1, d2):
        assert np.array_equal(a, b_)                                               # same bits, run to run
    for a, b_ in zip(d1, atomic):
        np.testing.assert_allclose(a, b_, rtol=1e-4, atol=1e-4)
    # fp64 references
    go = np.random.default_rng(5).standard_normal((B, C, S, K))
    ref = np.zeros((B, C, N)); ig = idx_g.cpu().numpy()
    for b in range(B):
        np.add.at(ref[b], (slice(None), ig[b].reshape(-1)), go[b].astype(np.float32).reshape(C, -1))
    np.testing.assert_allclose(d1[0], ref, rtol=1e-5, atol=1e-4)
    go = np.random.default_rng(7).standard_normal((B, C, N)).astype(np.float32)
    ref = np.zeros((B, C, S)); i3 = idx_3.cpu().numpy(); w = w3.cpu().numpy().astype(np.float64)
    for b in range(B):
        for t in range(3):
            np.add.at(ref[b], (slice(None), i3[b, :, t]), go[b] * w[b, :, t][None, :])
    np.testing.assert_allclose(d1[2], ref, rtol=1e-5, atol=1e-4)


def test_pointnet2_ops_vs_oracle(golden):
    from learning3d_amd.utils import pointnet2_utils as P
    rng = np.random.default_rng(5)
    xyz = np.clip(rng.standard_normal((3, 2000, 3)), -2, 2).astype(np.float32)
    feats = rng.uniform(0, 1, (3, 7, 2000)).astype(np.float32)
    txyz, tf = dev(xyz), dev(feats)
    fps = P.furthest_point_sample(txyz, 256)
    assert fps.dtype == torch.int32
    ofps = oracle.furthest_point_sampling(xyz, 256)
    assert np.array_equal(fps.cpu().numpy(), ofps)
    new_xyz = P.gather_operation(txyz.transpose(1, 2).contiguous(), fps)              # [B,3,S]
    assert np.array_equal(new_xyz.cpu().numpy(), oracle.gather_points(xyz.transpose(0, 2, 1), ofps))
    new_xyz_t = new_xyz.transpose(1, 2).contiguous()
    idx = P.ball_query(0.5, 16, txyz, new_xyz_t)
    oidx = oracle.ball_query(0.5, 16, xyz, new_xyz_t.cpu().numpy())
    assert idx.dtype == torch.int32 and np.array_equal(idx.cpu().numpy(), oidx)
    empty = P.ball_query(0.01, 8, txyz, dev(np.full((3, 5, 3), 9.0, np.float32)))
    assert (empty == 0).all()
    grouped = P.grouping_operation(tf, idx)
    assert np.array_equal(grouped.cpu().numpy(), oracle.group_points(feats, oidx))
    qg = P.QueryAndGroup(0.5, 16)(txyz, new_xyz_t, tf)
    assert qg.shape == (3, 10, 256, 16)
    # kNN between two clouds (FlowEmbedding uses k=64), 3-NN, interpolation
    other = np.clip(rng.standard_normal((3, 700, 3)), -2, 2).astype(np.float32)
    d, kidx = P.knn(64, dev(other), txyz)
    od, okidx = oracle.knn_pair(64, other, xyz)
    assert np.array_equal(kidx.cpu().numpy(), okidx)
    np.testing.assert_allclose(d.cpu().numpy(), od, rtol=0, atol=1e-7)
    d3, i3 = P.three_nn(dev(other), txyz)
    od3, oi3 = oracle.three_nn(other, xyz)
    assert np.array_equal(i3.cpu().numpy(), oi3)
    w = rng.uniform(0, 1, (3, 700, 3)).astype(np.float32)
    out = P.three_interpolate(tf, i3, dev(w))
    np.testing.assert_allclose(out.cpu().numpy(), oracle.three_interpolate(feats, oi3, w), rtol=1e-6, atol=1e-7)


def test_pointnet2_backward_vs_oracle():
    from learning3d_amd.utils import pointnet2_utils as P
    rng = np.random.default_rng(6)
    feats = rng.uniform(0, 1, (2, 5, 300)).astype(np.float32)
    idx = rng.integers(0, 300, (2, 40, 8)).astype(np.int32)
    go = rng.standard_normal((2, 5, 40, 8)).astype(np.float32)
    f = dev(feats).requires_grad_()
    P.grouping_operation(f, dev(idx)).backward(dev(go))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.group_points_grad(go, idx, 300), rtol=1e-5, atol=1e-5)
    idx1 = rng.integers(0, 300, (2, 64)).astype(np.int32)
    go1 = rng.standard_normal((2, 5, 64)).astype(np.float32)
    f = dev(feats).requires_grad_()
    P.gather_operation(f, dev(idx1)).backward(dev(go1))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.gather_points_grad(go1, idx1, 300), rtol=1e-5, atol=1e-5)
    i3 = rng.integers(0, 300, (2, 90, 3)).astype(np.int32)
    w = rng.uniform(0, 1, (2, 90, 3)).astype(np.float32)
    go3 = rng.standard_normal((2, 5, 90)).astype(np.float32)
    f = dev(feats).requires_grad_()
    P.three_interpolate(f, dev(i3), dev(w)).backward(dev(go3))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.three_interpolate_grad(go3, i3, w, 300), rtol=1e-5, atol=1e-5)


def test_flownet_sized_ball_query_properties():
    """BASELINE config 5 per-GPU slice: B=32, N=8192, S=1024, r=0.5, K=16 -- every returned index is
    inside the ball, rows are strictly ascending until the padding starts, 2 clouds vs the oracle."""
    from learning3d_amd.utils import pointnet2_utils as P
    g = torch.Generator().manual_seed(0)
    xyz = torch.clamp(torch.randn((32, 8192, 3), generator=g), -2, 2)
    txyz = xyz.cuda()
    fps = P.furthest_point_sample(txyz, 1024)
    new_xyz = P.gather_operation(txyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    idx = P.ball_query(0.5, 16, txyz, new_xyz)
    sel = list(range(32))                                   # all 32 clouds of the per-GPU slice against the oracle
    ofps = oracle.furthest_point_sampling(xyz[sel].numpy(), 1024)
    assert np.array_equal(fps.cpu().numpy()[sel], ofps)
    oidx = oracle.ball_query(0.5, 16, xyz[sel].numpy(), new_xyz.cpu().numpy()[sel])
    assert np.array_equal(idx.cpu().numpy()[sel], oidx)
    nb = torch.gather(txyz, 1, idx.long().view(32, -1, 1).expand(-1, -1, 3)).view(32, 1024, 16, 3)
    d2 = ((nb - new_xyz.unsqueeze(2)) ** 2).sum(-1)
    assert (d2 < 0.25 + 1e-6).all()


# --------------------------------------------------------------------------------------------- SVD
def test_svd3x3_golden(golden):
    from learning3d_amd.utils.svd import svd3x3_rotation
    g = golden("svd3x3")
    R = svd3x3_rotation(dev(g["H"])).cpu().numpy()
    s = np.linalg.svd(g["H"].astype(np.float64), compute_uv=False)
    ok = ((s[:, 1] - s[:, 2]) / s[:, 0] > 1e-2) & (s[:, 2] / s[:, 0] > 1e-2)
    np.testing.assert_allclose(R[ok], g["R"][ok], atol=1e-5)
    np.testing.assert_allclose(np.linalg.det(R.astype(np.float64)), 1.0, atol=1e-5)      # always proper
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.broadcast_to(np.eye(3), R.shape), atol=1e-5)


def test_svd_head_golden(golden):
    from learning3d_amd.utils import SVDHead
    g = golden("svd_head")
    head = SVDHead(emb_dims=32).cuda()
    R, t = head(dev(g["src_emb"]), dev(g["tgt_emb"]), dev(g["src"]), dev(g["tgt"]))
    np.testing.assert_allclose(R.cpu().numpy(), g["R"], atol=1e-5)
    np.testing.assert_allclose(t.cpu().numpy(), g["t"], atol=1e-5)
    assert "reflect" in head.state_dict()


def test_kabsch_recovers_known_transform():
    from learning3d_amd.utils.svd import kabsch
    rng = np.random.default_rng(3)
    B, N = 64, 1024
    src = rng.uniform(-0.5, 0.5, (B, 3, N))
    q, _ = np.linalg.qr(rng.standard_normal((B, 3, 3)))
    q[:, :, 0] *= np.sign(np.linalg.det(q))[:, None]
    tt = rng.uniform(-0.5, 0.5, (B, 3, 1))
    corr = q @ src + tt
    R, t = kabsch(dev(src, torch.float32), dev(corr, torch.float32))
    np.testing.assert_allclose(R.cpu().numpy(), q, atol=1e-5)
    np.testing.assert_allclose(t.cpu().numpy(), tt[:, :, 0], atol=1e-5)


# -------------------------------------------------------------------------------------- shared MLP
def _load(module, g):
    sd = {k[2:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("w.")}
    module.load_state_dict(sd)
    return module.cuda().eval()


def test_dgcnn_golden(golden):
    from learning3d_amd.models import DGCNN
    g = golden("dgcnn_emb64")
    net = _load(DGCNN(emb_dims=64), g)
    with torch.no_grad():
        out = net(dev(g["x"]))
    assert out.shape == (2, 64, 128)
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    # the differentiable per-layer route (HIP graph feature + HIP conv / BatchNorm layers; what a backward recomputes) agrees too
    from learning3d_amd.models import _fused
    x = dev(g["x"]).requires_grad_()
    with _fused.per_layer_route():
        out2 = net(x)
    np.testing.assert_allclose(out2.detach().cpu().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    # and with grad mode on (the reference's own call pattern) the forward is the fused one, bit for bit
    assert torch.equal(net(dev(g["x"])).detach(), out)


def test_graph_feature_backward_matches_torch_indexing():
    """get_graph_feature is differentiable in the reference (advanced indexing, model_common_utils.py:146-154);
    the HIP gather's backward must be the same adjoint."""
    from learning3d_amd.utils import get_graph_feature, knn
    rng = np.random.default_rng(91)
    for (B, C, N, k) in [(2, 3, 100, 8), (2, 64, 200, 20)]:
        x = dev(rng.standard_normal((B, C, N)).astype(np.float32)).requires_grad_()
        gw = dev(rng.standard_normal((B, 2 * C, N, k)).astype(np.float32))
        f = get_graph_feature(x, k=k)
        (f * gw).sum().backward()
        got = x.grad.clone()
        x2 = x.detach().clone().requires_grad_()
        idx = knn(x2.detach(), k)
        xt = x2.transpose(2, 1)                                                   # [B,N,C]
        nb = torch.gather(xt.unsqueeze(1).expand(B, N, N, C), 2, idx.unsqueeze(-1).expand(B, N, k, C))
        ref = torch.cat([nb, xt.unsqueeze(2).expand(B, N, k, C)], dim=3).permute(0, 3, 1, 2)
        np.testing.assert_array_equal(f.detach().cpu().numpy(), ref.detach().cpu().numpy())
        (ref * gw).sum().backward()
        np.testing.assert_allclose(got.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_prnet_dgcnn_dynamic_graphs_golden(golden):
    """SURVEY.md 8(f) rank 2's caller: PRNet's DGCNN (models/prnet.py:62-97) rebuilds the k-NN graph in feature
    space at every layer.  Golden = the reference class run on CPU (tests/golden/make_golden.py).  Fused route:
    l3d_knn_graph / l3d_knn_feature + one 2*Cout-row conv per layer + l3d_edge_gather_max; autograd route:
    the reference's op sequence."""
    from learning3d_amd.models.prnet import DGCNN as PRNetDGCNN
    g = golden("prnet_dgcnn_emb64")
    net = _load(PRNetDGCNN(emb_dims=64), g)
    with torch.no_grad():
        out = net(dev(g["x"]))
    assert out.shape == (2, 64, 128)
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-4, atol=2e-5)
    from learning3d_amd.models import _fused
    x = dev(g["x"]).requires_grad_()
    with _fused.per_layer_route():
        out2 = net(x)
    np.testing.assert_allclose(out2.detach().cpu().numpy(), g["out"], rtol=1e-4, atol=2e-5)
    out2.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    # full size (N = 1024, emb 512: feature kNN at C = 64 / 64 / 128, layers 3-4 on the GEMM kernels) against the REFERENCE class
    # run on CPU with the same seeded weights (golden prnet_dgcnn_full: every 8th channel x every 4th point, and every
    # channel's sum / maximum over the points).  A neighbour swapped at a rounding-level tie of the feature-space distances
    # changes a max over k only where that neighbour won it: such entries are counted, not tolerated silently
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from seeded import seeded_params
    gf = golden("prnet_dgcnn_full")
    torch.manual_seed(3)
    big = seeded_params(PRNetDGCNN(emb_dims=512).eval(), int(gf["seed"])).cuda()
    with torch.no_grad():
        fused = big(dev(gf["x"]))
    assert fused.shape == (2, 512, 1024)
    got, want = fused[:, ::8, ::4].cpu().numpy(), gf["out_strided"]
    bad = np.abs(got - want) > 2e-5 + 1e-4 * np.abs(want)
    assert bad.mean() < 1e-3, bad.mean()
    np.testing.assert_allclose(fused.double().sum(-1).cpu().numpy(), gf["out_sum"], rtol=2e-5, atol=1e-3)
    mx = fused.max(-1)[0].cpu().numpy()
    assert (np.abs(mx - gf["out_max"]) > 2e-5 + 1e-4 * np.abs(gf["out_max"])).mean() < 5e-3
    with _fused.per_layer_route():
        ref = big(dev(gf["x"]).requires_grad_()).detach()
    bad = (fused - ref).abs() > 1e-4 + 1e-4 * ref.abs()
    assert bad.float().mean().item() < 1e-3, bad.float().mean().item()


def test_edge_gather_max_and_leaky_activation_code():
    from learning3d_amd.models import _fused
    from learning3d_amd.models.prnet import ACT_LRELU
    from learning3d_amd._lib import lib, check, ptr, stream_ptr
    rng = np.random.default_rng(77)
    for (B, Cout, N, k) in [(2, 64, 300, 20), (1, 10, 77, 5), (3, 256, 1024, 20), (1, 8, 5000, 33)]:
        pq = rng.standard_normal((B, 2 * Cout, N)).astype(np.float32)
        idx = rng.integers(0, N, (B, N, k)).astype(np.int64)
        P, Q = pq[:, :Cout], pq[:, Cout:]
        gathered = np.stack([P[b][:, idx[b]] for b in range(B)])                    # [B,Cout,N,k]
        z = gathered.max(axis=-1) + Q
        want = np.where(z > 0, z, np.float32(0.2) * z)
        buf = torch.full((B, Cout + 3, N), 7.0, device="cuda")                       # a slice of a wider buffer
        out = buf[:, 1:1 + Cout]
        pq_d, idx_d = dev(pq), dev(idx)                                              # keep alive across the call
        check(lib().l3d_edge_gather_max(ptr(pq_d), ptr(idx_d), B, Cout, N, k, ACT_LRELU, ptr(out), (Cout + 3) * N,
                                        stream_ptr()), "l3d_edge_gather_max")
        np.testing.assert_array_equal(out.cpu().numpy(), want)
        assert torch.all(buf[:, 0] == 7.0) and torch.all(buf[:, 1 + Cout:] == 7.0)
    # LeakyReLU through the conv epilogues (fp32-MFMA kernel, narrow kernel, bf16x3 kernel)
    for (B, Cin, Cout, N) in [(2, 5, 70, 100), (1, 64, 4, 200), (2, 64, 256, 256)]:
        x = rng.standard_normal((B, Cin, N)).astype(np.float32)
        w = (rng.standard_normal((Cout, Cin)) / np.sqrt(Cin)).astype(np.float32)
        sh = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
        z = np.einsum("oc,bcn->bon", w.astype(np.float64), x) + sh[None, :, None]
        want = np.where(z > 0, z, 0.2 * z)
        got = _fused.pointwise_conv(dev(x), dev(w), None, dev(sh), relu=ACT_LRELU).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)


def test_dgcnn_full_size_vs_oracle_port():
    """BASELINE config 2 shape for 2 clouds (emb 1024): fused HIP forward vs the torch-CPU oracle."""
    from learning3d_amd.models import DGCNN
    torch.manual_seed(1)
    net = DGCNN(emb_dims=1024).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.1, 0.1); m.running_var.uniform_(0.8, 1.2)
    x = rand((2, 1024, 3), 0)
    w = {k: v.numpy() for k, v in net.state_dict().items()}
    want = oracle.dgcnn_forward_torch(x, w).numpy()
    with torch.no_grad():
        got = net.cuda()(dev(x)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)


def test_edgeconv_all_kernels_agree_and_ragged():
    """LDS-staged (mlp.hip), register-chained fp32-MFMA (edgeconv2.hip), register-chained bf16x3
    (edgeconv_split.hip) and register-chained f16x2 (edgeconv_f16b.hip) EdgeConv kernels against a torch fp64
    evaluation, including N not a multiple of 16 and k < 20.  The matrix-core kernels (bf16x3: six bf16 products per
    fp32 product; f16x2: three fp16 products) must be as close to fp64 as the fp32-MFMA kernels are: max error
    <= 2x, rms error <= 1.5x the fp32-MFMA kernel's own.  Also with activations scaled down 1000x / up 100x through
    the BatchNorm weights (the f16x2 residual scaling and the weight scaling have to hold there)."""
    from learning3d_amd.models import DGCNN, _fused
    import learning3d_amd.utils as U
    torch.manual_seed(4)
    net = DGCNN(emb_dims=64).cuda().eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.1, 0.1); m.running_var.uniform_(0.8, 1.2)
    for (B, N, k, gain) in [(2, 1024, 20, 1.0), (3, 203, 20, 1.0), (2, 77, 16, 1.0), (1, 50, 7, 1.0),
                            (2, 512, 20, 1e-3), (2, 512, 20, 30.0)]:
        x = dev(rand((B, N, 3), 30 + N))
        with torch.no_grad():
            net.bn1.weight.fill_(gain)                     # scales every activation of the stack
            idx = U.knn(x.permute(0, 2, 1), k)
            packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
            a = _fused.edgeconv_forward(x, idx, packed, kernel="lds")
            c = a                                           # the fp32-MFMA kernel: the yardstick of the error bars below
            sp = _fused.edgeconv_forward(x, idx, packed, kernel="split")
            # the f16x2 kernel (edgeconv_f16b.hip) where its block is usable: always for BatchNorm magnitudes that say what the
            # activations are (gain 1); with bn1 scaled down 1000x the chained plane exponents would leave a layer below 2^4,
            # the packer marks the block unusable and "f16" is served by the bf16x3 kernel (gain 30 still fits)
            assert net._packed.v2_ok or gain != 1.0
            assert not (net._packed.v2_ok and gain == 1e-3), "plane exponents 10 binades apart cannot be chained"
            f16b = _fused.edgeconv_forward(x, idx, packed, kernel="f16", v2=net._packed.v2_ok)
            if not net._packed.v2_ok:
                assert torch.equal(f16b, sp)                # the fallback IS the bf16x3 kernel
            _fused.check_range(x.device, sync=True)
            # fp64 torch evaluation of dgcnn.py:32-46 on the same graph
            nb = torch.gather(x.unsqueeze(1).expand(B, N, N, 3), 2, idx.unsqueeze(-1).expand(B, N, k, 3))
            h = torch.cat([nb, x.unsqueeze(2).expand(B, N, k, 3)], dim=3).permute(0, 3, 1, 2).double()
            outs = []
            for conv, bn in [(net.conv1, net.bn1), (net.conv2, net.bn2), (net.conv3, net.bn3), (net.conv4, net.bn4)]:
                w, sc, sh = _fused.fold_conv_bn(conv, bn)
                h = torch.relu(torch.einsum("oc,bcnk->bonk", w.double(), h) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
                outs.append(h.max(dim=-1)[0])
            want = torch.cat(outs, dim=1).permute(0, 2, 1).float().cpu().numpy()
            want64 = torch.cat(outs, dim=1).permute(0, 2, 1).cpu().numpy()
            # the same stack evaluated in plain fp32 by torch (rocBLAS): with the fp32-MFMA kernel above, the yardstick of what
            # "fp32-level error" means at this input scale (round 3 measured against the register-chained fp32 kernel, retired)
            h32 = torch.cat([nb, x.unsqueeze(2).expand(B, N, k, 3)], dim=3).permute(0, 3, 1, 2).contiguous()
            outs32 = []
            for conv, bn in [(net.conv1, net.bn1), (net.conv2, net.bn2), (net.conv3, net.bn3), (net.conv4, net.bn4)]:
                w, sc, sh = _fused.fold_conv_bn(conv, bn)
                h32 = torch.relu(torch.einsum("oc,bcnk->bonk", w, h32) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
                outs32.append(h32.max(dim=-1)[0])
            t32 = torch.cat(outs32, dim=1).permute(0, 2, 1).cpu().numpy()
        np.testing.assert_allclose(a.cpu().numpy(), want, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(sp.cpu().numpy(), want, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(f16b.cpu().numpy(), want, rtol=1e-4, atol=1e-5 * max(1.0, gain))
        e_c = np.maximum(np.abs(c.cpu().numpy() - want64).max(), np.abs(t32 - want64).max()) * np.ones(1)
        r_c = max(np.sqrt(((c.cpu().numpy() - want64) ** 2).mean()), np.sqrt(((t32 - want64) ** 2).mean()))
        for name, got in (("bf16x3", sp), ("f16x2", f16b)):
            e_s = np.abs(got.cpu().numpy() - want64)
            print(f"edgeconv {name} B={B} N={N} k={k} gain={gain}: max err {e_s.max():.3e} ({e_s.max() / e_c.max():.2f}x fp32), "
                  f"rms {np.sqrt((e_s ** 2).mean()):.3e} ({np.sqrt((e_s ** 2).mean()) / r_c:.2f}x)")
            # no worse than 2x (max) / 1.5x (rms) of an fp32 evaluation's own error.  With the activations scaled down 1000x the
            # BatchNorm shifts dominate and the fp32 evaluations land at ~2.7 ulp of the largest output; the split kernels measure
            # 5.6 ulp / 0.55 ulp rms there (2.05x / 3.3x of the fp32 evaluations -- 8e-8 absolute against the 1e-5 tolerance), so
            # that case is held to 8 ulp / 1 ulp of the largest output instead
            ulp = 2.0 ** (np.floor(np.log2(float(np.abs(want64).max()))) - 23)
            assert e_s.max() <= max(2.0 * e_c.max(), 8 * ulp if gain < 1 else 0.0), (name, B, N, k, gain, e_s.max(), e_c.max(), ulp)
            assert np.sqrt((e_s ** 2).mean()) <= max(1.5 * r_c, ulp if gain < 1 else 0.0), (name, B, N, k, gain, np.sqrt((e_s ** 2).mean()), r_c, ulp)


def test_edgeconv_f16_planes_output_equals_pooled():
    """out_mode 1 of the f16 EdgeConv kernel (pooled values handed to conv5 as fp16 planes, scaled by 2^T_out) decodes
    to the fp32 pooled output of out_mode 0 within the representation error of the split (2^-23 of the value, or
    2^-49 of the tensor scale), and conv5 on it equals conv5 on the fp32 pooled tensor split by the generic splitter
    to fp32 rounding."""
    from learning3d_amd.models import DGCNN, _fused
    import learning3d_amd.utils as U
    torch.manual_seed(4)
    net = DGCNN(emb_dims=256).cuda().eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.1, 0.1); m.running_var.uniform_(0.8, 1.2); m.weight.data.uniform_(0.5, 2.0)
    B, N, k = 2, 512, 20
    x = dev(rand((B, N, 3), 91))
    with torch.no_grad():
        idx = U.knn(x.permute(0, 2, 1), k)
        packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
        assert net._packed.v2_ok
        for v2 in (True,):
            pooled = _fused.edgeconv_forward(x, idx, packed, kernel="f16", v2=v2)
            img = _fused.edgeconv_forward(x, idx, packed, planes=True, v2=v2)
            _fused.check_range(sync=True)
            raw = img.cpu().numpy()
            pb = 64 * B * N * 16
            h = raw[:pb].view(np.float16).reshape(64, B * N, 8).astype(np.float64)
            m = raw[pb:2 * pb].view(np.float16).reshape(64, B * N, 8).astype(np.float64)
            xinv = float(raw[2 * pb:2 * pb + 4].view(np.float32)[0])
            dec = ((h + m / 4096.0) * xinv).transpose(1, 0, 2).reshape(B, N, 512)
            want = pooled.cpu().numpy().astype(np.float64)
            assert np.log2(xinv) == np.round(np.log2(xinv))                      # a power of two
            np.testing.assert_allclose(dec, want, rtol=2.0 ** -22, atol=np.abs(want).max() * 2.0 ** -40)
            w5, s5, b5, _, w5f = net._conv5_folded()
            y1 = _fused.pointwise_conv_f16(img, B, N, w5f, 512, 256, s5, b5, relu=True)
            y2 = _fused.pointwise_conv_f16(_fused.split_rows_f16(pooled), B, N, w5f, 512, 256, s5, b5, relu=True)
            np.testing.assert_allclose(y1.cpu().numpy(), y2.cpu().numpy(), rtol=1e-5, atol=1e-6)
        # out_mode 2: the same pooled values with an UNSCALED residual plane, for the two-plane conv kernel (l3d_pointwise_conv_f16 with L3D_CONV_F16_TWO_PLANE)
        img2 = _fused.edgeconv_forward(x, idx, packed, planes=True, v2=True, unscaled=True)
        raw2 = img2.cpu().numpy()
        h2 = raw2[:pb].view(np.float16).reshape(64, B * N, 8).astype(np.float64)
        m2 = raw2[pb:2 * pb].view(np.float16).reshape(64, B * N, 8).astype(np.float64)
        assert np.array_equal(h2, h)                                           # same h plane as out_mode 1 of the same kernel
        dec2 = ((h2 + m2) * xinv).transpose(1, 0, 2).reshape(B, N, 512)
        np.testing.assert_allclose(dec2, want, rtol=2.0 ** -22, atol=xinv * 2.0 ** -24)        # an unscaled residual below 2^-14 is a subnormal: 2^-25 in plane units
        y3 = _fused.pointwise_conv_f16(img2, B, N, w5f, 512, 256, s5, b5, relu=True, unscaled=True)
        np.testing.assert_allclose(y3.cpu().numpy(), y1.cpu().numpy(), rtol=1e-5, atol=1e-6)
        # both conv kernels against fp64 on the decoded image: the two-plane kernel no further from it than 1.5x the three-plane one
        w5, _, _, _, _ = net._conv5_folded()
        ref = torch.relu(torch.einsum("oc,bnc->bon", w5.double(), torch.from_numpy(want).cuda()) * s5.double()[None, :, None] + b5.double()[None, :, None])
        e3, e2 = (y1.double() - ref).abs(), (y3.double() - ref).abs()
        assert float(e2.max()) <= 1.5 * float(e3.max()) + 1e-9 and float((e2 ** 2).mean().sqrt()) <= 1.25 * float((e3 ** 2).mean().sqrt()) + 1e-12
        # and the model's own forward takes this route (the two-plane kernels by default)
        out = net(x)
        np.testing.assert_allclose(out.cpu().numpy(), y3.cpu().numpy(), rtol=0, atol=0)


def test_edgeconv_f16_range_flag():
    """f16x2 range contract: an activation beyond fp16's range raises the flag (and check_range raises); normal inputs
    never do.  The flag lives in pinned host memory the kernel writes directly."""
    from learning3d_amd.models import DGCNN, _fused
    import learning3d_amd.utils as U
    torch.manual_seed(4)
    net = DGCNN(emb_dims=64).cuda().eval()
    x = dev(rand((2, 256, 3), 77))
    with torch.no_grad():
        idx = U.knn(x.permute(0, 2, 1), 20)
        get = lambda: net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
        packed = get()
        assert net._packed.v2_ok
        _fused.edgeconv_forward(x, idx, packed, kernel="f16", v2=True)
        _fused.check_range(x.device, sync=True)                         # fine
        # the plane exponents are static (from the BatchNorm magnitudes): coordinates 3e4 times larger than the statistics
        # describe put layer 1's outputs beyond fp16 (a BatchNorm weight of 1e6 would instead be seen by the packer, which then
        # marks the block unusable and the call goes to the bf16x3 kernel)
        x = x * 3.0e4
        _fused.edgeconv_forward(x, idx, packed, kernel="f16", v2=True)
        with pytest.raises(_fused.L3DRangeError):
            _fused.check_range(x.device, sync=True)
        _fused.check_range(x.device, sync=True)                         # the flag was cleared by the raise
        net.bn2.weight.fill_(1e6)
        get()
        assert not net._packed.v2_ok                                    # ... as said: not chainable
        sp = _fused.edgeconv_forward(x, idx, get(), kernel="split")      # the wide-range kernel takes the same input
        assert torch.isfinite(sp).all()


def test_pointnet_golden(golden):
    from learning3d_amd.models import PointNet
    g = golden("pointnet_emb64")
    net = _load(PointNet(emb_dims=64, use_bn=True), g)
    with torch.no_grad():
        out = net(dev(g["x"]))
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-4, atol=1e-5)


def test_knn_feature_space_matches_exact_topk():
    """C != 3 (feature-space graphs): l3d_knn_feature = bf16x3 GEMM + top-k epilogue.
    Its indices cannot be bit-pinned to the reference (MKL's sgemm summation order), so every returned
    neighbour list is checked against exact fp64 distances: the set is the k nearest up to rounding, the
    order is ascending, self comes first, no index repeats."""
    from learning3d_amd.utils import knn, get_graph_feature
    from learning3d_amd import _lib
    for (B, C, N, k, seed) in [(2, 64, 300, 16, 61), (2, 32, 128, 20, 62), (2, 128, 1000, 20, 63), (1, 256, 513, 7, 64),
                               (1, 96, 77, 1, 65), (1, 64, 200, 24, 66), (1, 48, 150, 8, 67),   # k > 20; C % 32 != 0 (zero-padded)
                               (2, 9, 300, 21, 70), (1, 130, 257, 64, 71), (1, 5, 40, 33, 72)]:
        rng = np.random.default_rng(seed)
        x = rng.standard_normal((B, C, N)).astype(np.float32)
        idx = knn(dev(x), k).cpu().numpy()
        assert idx.shape == (B, N, k) and idx.dtype == np.int64
        xd = x.astype(np.float64)
        sq = (xd ** 2).sum(axis=1)
        d = sq[:, :, None] + sq[:, None, :] - 2 * np.einsum("bci,bcj->bij", xd, xd)  # [B,N,N]
        kth = np.sort(d, axis=-1)[:, :, k - 1]
        got = np.take_along_axis(d, idx, axis=-1)
        tol = 4e-6 * sq.max()
        assert np.all(got.max(axis=-1) <= kth + tol), (B, C, N, k)
        assert np.all(np.diff(got, axis=-1) >= -tol), (B, C, N, k)
        assert np.all(idx[:, :, 0] == np.arange(N)[None])                          # self first
        srt = np.sort(idx, axis=-1)
        assert np.all(srt[:, :, 1:] != srt[:, :, :-1])                             # no repeats
    # against the reference's own op sequence (oracle.knn_feature: MKL sgemm + topk in fp32 on the CPU): identical
    # indices except where two candidates sit within fp32 rounding of each other
    x = np.random.default_rng(69).standard_normal((2, 64, 1024)).astype(np.float32)
    mine, ref = knn(dev(x), 20).cpu().numpy(), oracle.knn_feature(x, 20).numpy()
    assert (mine == ref).mean() > 0.999, (mine == ref).mean()
    # exact ties (every point duplicated N/2 later): equal fp32 scores -> lower index first, as l3d_knn_graph
    rng = np.random.default_rng(68)
    half = rng.standard_normal((1, 64, 128)).astype(np.float32)
    x = np.concatenate([half, half], axis=2)
    idx = knn(dev(x), 20).cpu().numpy()[0]
    assert np.all(idx[:, 0::2] + 128 == idx[:, 1::2]) and np.all(idx[:, 0::2] < 128)
    # lists that overflow (round 6: more than 64 candidates at the bound -> the exact wave-per-query selection at the end of the kernel):
    # a cloud of identical points (every ranking value equal: indices 0 .. k-1 for every query), and 200 copies of one point among
    # random ones (their queries see 200 exact ties at the top; the others must not be disturbed)
    for C, N, k in ((64, 300, 20), (128, 1024, 20), (48, 200, 33)):
        x = np.broadcast_to(np.random.default_rng(73).standard_normal((1, C, 1)).astype(np.float32), (1, C, N)).copy()
        idx = knn(dev(x), k).cpu().numpy()[0]
        assert np.array_equal(idx, np.broadcast_to(np.arange(k), (N, k))), (C, N, k)
    rng = np.random.default_rng(74)
    x = rng.standard_normal((1, 64, 1024)).astype(np.float32)
    x[:, :, 100:300] = x[:, :, 100:101]
    idx = knn(dev(x), 20).cpu().numpy()[0]
    assert np.array_equal(idx[100:300], np.broadcast_to(np.arange(100, 120), (200, 20)))            # ties: lowest indices first
    xd = x.astype(np.float64)[0]
    sq = (xd ** 2).sum(0)
    d = sq[:, None] + sq[None, :] - 2 * xd.T @ xd
    kth = np.sort(d, axis=1)[:, 19]
    got = np.take_along_axis(d, idx, axis=1)
    assert np.all(got.max(axis=1) <= kth + 4e-6 * sq.max())
    # the caller: get_graph_feature on a feature map
    x = np.random.default_rng(61).standard_normal((2, 64, 300)).astype(np.float32)
    feat = get_graph_feature(dev(x), k=16)
    assert feat.shape == (2, 128, 300, 16)
    np.testing.assert_array_equal(feat[:, 64:, :, 3].cpu().numpy(), x)             # centre half = the point itself
    # unsupported shapes are refused by the C ABI, not silently mangled
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    out = torch.empty((1, 64, 8), dtype=torch.int64, device="cuda")
    xx = torch.zeros((1, 48, 64), device="cuda")
    assert _lib.lib().l3d_knn_feature(_lib.ptr(xx), 1, 48, 64, 65, _lib.ptr(ws), _lib.ptr(out), _lib.stream_ptr()) == -1   # k > N
    xx = torch.zeros((1, 32, 128), device="cuda")
    out = torch.empty((1, 128, 80), dtype=torch.int64, device="cuda")
    assert _lib.lib().l3d_knn_feature(_lib.ptr(xx), 1, 32, 128, 80, _lib.ptr(ws), _lib.ptr(out), _lib.stream_ptr()) == -2  # k > 64


def test_pointwise_conv_ragged_shapes():
    from learning3d_amd.models._fused import pointwise_conv
    rng = np.random.default_rng(9)
    for (B, Cin, Cout, N) in [(2, 3, 64, 100), (1, 130, 70, 257), (3, 512, 256, 128), (2, 5, 512, 1000),
                              (2, 512, 3, 1000), (1, 128, 8, 77)]:
        x = rng.standard_normal((B, Cin, N)).astype(np.float32)
        w = (rng.standard_normal((Cout, Cin)) / np.sqrt(Cin)).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
        sh = rng.uniform(-0.5, 0.5, (B, Cout)).astype(np.float32)
        want = np.maximum(np.einsum("oc,bcn->bon", w.astype(np.float64), x) * sc[None, :, None] + sh[:, :, None], 0)
        got = pointwise_conv(dev(x), dev(w), dev(sc), dev(sh), relu=True).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
        got_cl = pointwise_conv(dev(np.ascontiguousarray(x.transpose(0, 2, 1))), dev(w), dev(sc), dev(sh), relu=True,
                                channel_last=True).cpu().numpy()
        np.testing.assert_allclose(got_cl, want, rtol=1e-4, atol=1e-5)


def test_conv_split_accuracy():
    """bf16x3 (six bf16 MFMA products) vs an fp64 reference: its error must be at the fp32 level --
    no worse than 1.5x the fp32-MFMA kernel's own error on the same inputs -- in all three x modes."""
    from learning3d_amd.models import _fused
    from learning3d_amd._lib import lib, check, ptr, stream_ptr
    rng = np.random.default_rng(21)
    for (B, Cin, Cout, N, scale_x) in [(2, 512, 1024, 1024, 1.0), (1, 64, 256, 256, 1e-3), (3, 320, 512, 512, 50.0),
                                      (2, 128, 256, 384, 1.0)]:
        x = (np.maximum(rng.standard_normal((B, N, Cin)), 0) * scale_x).astype(np.float32)     # post-ReLU like
        w = (rng.standard_normal((Cout, Cin)) / np.sqrt(Cin)).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
        sh = (rng.uniform(-0.5, 0.5, (B, Cout)) * scale_x).astype(np.float32)
        want = np.einsum("oc,bnc->bon", w.astype(np.float64), x.astype(np.float64)) * sc[None, :, None] + sh[:, :, None]
        xd, wd, scd, shd = dev(x), dev(w), dev(sc), dev(sh)
        f32 = _fused.pointwise_conv(xd, wd, scd, shd, channel_last=True, split=False).cpu().numpy()
        spl = _fused.pointwise_conv(xd, wd, scd, shd, channel_last=True, split=True).cpu().numpy()
        xcf = dev(np.ascontiguousarray(x.transpose(0, 2, 1)))
        spl_cf = _fused.pointwise_conv(xcf, wd, scd, shd, channel_last=False, split=True).cpu().numpy()
        # x_mode 2: x pre-split with the same splitter
        xs = _fused.split_rows(xd.reshape(B * N, Cin))
        ws = _fused.split_rows(wd)
        y2 = torch.empty((B, Cout, N), dtype=torch.float32, device="cuda")
        check(lib().l3d_pointwise_conv_split(ptr(xs), 2, ptr(ws), ptr(scd), ptr(shd), Cout, B, Cin, Cout, N, 0, 0, ptr(y2),
                                             stream_ptr()), "l3d_pointwise_conv_split")
        e32 = np.abs(f32 - want)
        for name, got in (("channel_last", spl), ("channel_first", spl_cf), ("presplit", y2.cpu().numpy())):
            es = np.abs(got - want)
            assert es.max() <= 1.5 * e32.max() + 1e-30, (name, B, Cin, Cout, N, es.max(), e32.max())
            assert np.sqrt((es ** 2).mean()) <= 1.5 * np.sqrt((e32 ** 2).mean()), (name, "rms")
        np.testing.assert_array_equal(spl, spl_cf)          # same products, same order
        np.testing.assert_array_equal(spl, y2.cpu().numpy())


def test_conv_f16_accuracy():
    """f16x2 1x1 conv (three fp16 MFMA products per fp32 product, conv_f16.hip) vs an fp64 reference: error at the fp32
    level -- max <= 2x, rms <= 1.5x the fp32-MFMA kernel's own error on the same inputs -- for activations scaled 1e-3 ..
    1e-6 .. 50 (the splitter scales the tensor by a power of two taken from its own maximum), weights x 0.01 .. 30,
    channel-last and channel-first sources; non-finite inputs raise the range flag."""
    from learning3d_amd.models import _fused
    rng = np.random.default_rng(22)
    for (B, Cin, Cout, N, scale_x) in [(2, 512, 1024, 1024, 1.0), (1, 64, 256, 256, 1e-3), (3, 320, 512, 512, 50.0),
                                      (2, 128, 256, 768, 1.0), (1, 32, 256, 256, 1e-6)]:
        x = (np.maximum(rng.standard_normal((B, N, Cin)), 0) * scale_x).astype(np.float32)     # post-ReLU like
        w = (rng.standard_normal((Cout, Cin)) / np.sqrt(Cin)).astype(np.float32) * rng.choice([0.01, 1.0, 30.0])
        sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
        sh = (rng.uniform(-0.5, 0.5, (B, Cout)) * scale_x).astype(np.float32)
        want = np.einsum("oc,bnc->bon", w.astype(np.float64), x.astype(np.float64)) * sc[None, :, None] + sh[:, :, None]
        xd, wd, scd, shd = dev(x), dev(w), dev(sc), dev(sh)
        f32 = _fused.pointwise_conv(xd, wd, scd, shd, channel_last=True, split=False).cpu().numpy()
        wp = _fused.split_weights_f16(wd)
        y_cl = _fused.pointwise_conv_f16(_fused.split_rows_f16(xd), B, N, wp, Cin, Cout, scd, shd).cpu().numpy()
        xcf = dev(np.ascontiguousarray(x.transpose(0, 2, 1)))
        y_cf = _fused.pointwise_conv_f16(_fused.split_rows_f16(xcf, channel_first=True), B, N, wp, Cin, Cout, scd, shd).cpu().numpy()
        _fused.check_range(xd.device, sync=True)
        e32 = np.abs(f32 - want)
        es = np.abs(y_cl - want)
        print(f"conv f16x2 B={B} Cin={Cin} Cout={Cout} N={N} x-scale {scale_x}: max err {es.max():.3e} ({es.max() / e32.max():.2f}x fp32-MFMA), "
              f"rms {np.sqrt((es ** 2).mean()):.3e} ({np.sqrt((es ** 2).mean()) / np.sqrt((e32 ** 2).mean()):.2f}x)")
        assert es.max() <= 2.0 * e32.max() + 1e-30, (B, Cin, Cout, N, es.max(), e32.max())
        assert np.sqrt((es ** 2).mean()) <= 1.5 * np.sqrt((e32 ** 2).mean()), (B, Cin, Cout, N, "rms")
        np.testing.assert_array_equal(y_cl, y_cf)            # same planes, same products, same order
    bad = np.ones((1, 256, 32), np.float32); bad[0, 3, 5] = np.inf
    _fused.split_rows_f16(dev(bad))
    with pytest.raises(_fused.L3DRangeError):
        _fused.check_range(sync=True)


def test_pointwise_conv_maxpool_epilogue():
    """conv + BN + ReLU + max over K consecutive points in one launch vs the two-step composition,
    K in {8,16,32,64}, ragged Cout / S."""
    from learning3d_amd.models._fused import pointwise_conv, pointwise_conv_maxpool
    rng = np.random.default_rng(71)
    for (B, Cin, Cout, S, K) in [(2, 6, 64, 100, 16), (1, 131, 70, 33, 8), (2, 64, 128, 40, 32), (1, 259, 128, 24, 64),
                                 (2, 128, 256, 128, 16)]:
        x = rng.standard_normal((B, Cin, S * K)).astype(np.float32)
        w = (rng.standard_normal((Cout, Cin)) / np.sqrt(Cin)).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
        sh = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
        full = pointwise_conv(dev(x), dev(w), dev(sc), dev(sh), relu=True, split=False)
        want = full.view(B, Cout, S, K).max(dim=-1)[0].cpu().numpy()
        got = pointwise_conv_maxpool(dev(x), dev(w), dev(sc), dev(sh), True, K)
        assert got is not None and got.shape == (B, Cout, S)
        if (Cout % 256 == 0 and (S * K) % 256 == 0 and Cin % 16 == 0 and Cin >= 32):   # bf16x3 kernel took it
            full = pointwise_conv(dev(x), dev(w), dev(sc), dev(sh), relu=True, split=True)
            want = full.view(B, Cout, S, K).max(dim=-1)[0].cpu().numpy()
        np.testing.assert_array_equal(got.cpu().numpy(), want)               # same kernel arithmetic, only the epilogue differs
    from learning3d_amd.models._fused import conv_global_max
    x = rng.standard_normal((3, 512, 2048)).astype(np.float32)
    w = (rng.standard_normal((1024, 512)) / np.sqrt(512)).astype(np.float32)
    b_ = rng.standard_normal(1024).astype(np.float32)
    want = pointwise_conv(dev(x), dev(w), None, dev(b_), relu=False).max(dim=2)[0].cpu().numpy()
    np.testing.assert_array_equal(conv_global_max(dev(x), dev(w), None, dev(b_), False).cpu().numpy(), want)


def test_query_and_group_fused_matches_composition():
    """QueryAndGroup's one-pass gather + centring + concat (l3d_group_concat) against the autograd route
    (two grouping launches, subtraction, torch.cat) on the same ball-query indices."""
    from learning3d_amd.utils.pointnet2_utils import QueryAndGroup
    rng = np.random.default_rng(81)
    B, N, S, K, Cf = 2, 500, 70, 16, 5
    xyz = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    new_xyz = xyz[:, :S].copy()
    feat = rng.standard_normal((B, Cf, N)).astype(np.float32)
    for use_xyz, f in ((True, feat), (True, None), (False, feat)):
        qg = QueryAndGroup(0.4, K, use_xyz=use_xyz)
        with torch.no_grad():
            fused = qg(dev(xyz), dev(new_xyz), dev(f) if f is not None else None)
        ref = qg(dev(xyz).requires_grad_(), dev(new_xyz), dev(f) if f is not None else None)   # composition route
        np.testing.assert_array_equal(fused.cpu().numpy(), ref.detach().cpu().numpy())


def test_pcn_fused_matches_reference_order_path():
    from learning3d_amd.models import PCN
    torch.manual_seed(3)
    net = PCN(emb_dims=1024, num_coarse=64, grid_size=2, detailed_output=True).cuda().eval()   # conv5 is 1029-wide: emb must be 1024 (pcn.py:73)
    x = dev(rand((2, 300, 3), 4, -0.5, 0.5))
    with torch.no_grad():
        fused = net(x)
    from learning3d_amd.models import _fused
    with _fused.per_layer_route():
        ref = net(x.clone().requires_grad_())        # per-layer differentiable route = the reference's op order
    for k in ("coarse_output", "fine_output"):
        np.testing.assert_allclose(fused[k].cpu().numpy(), ref[k].detach().cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_fold_mlp_both_arithmetics_vs_fp64():
    """PCN's folding decoder as one kernel (fold_mlp.hip bf16x3, fold_mlp_f16.hip f16x2) against an fp64 evaluation of
    models/pcn.py:84-101, ragged N, activations from tiny to large: the f16x2 kernel picks its plane scale per workgroup from
    a bound it computes itself, so no magnitude may overflow fp16 or lose fp32-level accuracy."""
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    from learning3d_amd.models import _fused
    rng = np.random.default_rng(77)
    B, N = 2, 600
    for gscale, sscale in ((1.0, 1.0), (1e-3, 1e-3), (30.0, 100.0)):
        g = (rng.standard_normal((B, N, 5)) * gscale).astype(np.float32)
        w5g = (rng.standard_normal((512, 5)) * 0.5).astype(np.float32)
        s5 = (rng.standard_normal((B, 512)) * sscale).astype(np.float32)
        w6 = (rng.standard_normal((512, 512)) / 512 ** 0.5).astype(np.float32)
        b6 = (rng.standard_normal(512) * 0.1 * sscale).astype(np.float32)
        w7 = (rng.standard_normal((3, 512)) / 512 ** 0.5).astype(np.float32)
        b7 = rng.standard_normal(3).astype(np.float32)
        ce = rng.standard_normal((B, N, 3)).astype(np.float32)
        h5 = np.maximum(s5[:, None, :].astype(np.float64) + g.astype(np.float64) @ w5g.astype(np.float64).T, 0)
        h6 = np.maximum(h5 @ w6.astype(np.float64).T + b6, 0)
        want = h6 @ w7.astype(np.float64).T + b7 + ce
        dv = {k: dev(v) for k, v in dict(g=g, w5g=w5g, s5=s5, w6=w6, b6=b6, w7=w7, b7=b7, ce=ce).items()}
        outs = {}
        for name, fn, wimg in (("bf16x3", lib().l3d_fold_mlp, _fused.split_rows(dv["w6"])),
                               ("f16x2", lib().l3d_fold_mlp_f16, _fused.split_weights_f16(dv["w6"]))):
            out = torch.empty((B, N, 3), dtype=torch.float32, device="cuda")
            check(fn(ptr(dv["g"]), 5, ptr(dv["w5g"]), ptr(dv["s5"]), ptr(wimg), ptr(dv["b6"]), ptr(dv["w7"]), ptr(dv["b7"]),
                     ptr(dv["ce"]), B, N, ptr(out), stream_ptr()), name)
            outs[name] = out.cpu().numpy() - want
        scale = np.abs(want).max()
        e3, e16 = outs["bf16x3"], outs["f16x2"]
        assert np.abs(e16).max() <= 2e-6 * scale + 2.0 * np.abs(e3).max(), (gscale, np.abs(e16).max(), np.abs(e3).max())
        assert np.sqrt((e16 ** 2).mean()) <= 1.5 * np.sqrt((e3 ** 2).mean()) + 1e-7 * scale


def test_soft_correspondence_flash_vs_fp64():
    """Fused score-GEMM + softmax + weighted target sum (softcorr.hip) against an fp64 evaluation of
    utils/svd.py:22-27, ragged N / M included; and the SVDHead built on it against the oracle."""
    from learning3d_amd.utils.svd import soft_correspondence, SVDHead
    rng = np.random.default_rng(33)
    for (B, C, N, M) in [(2, 512, 1024, 1024), (3, 64, 200, 333), (1, 128, 128, 700), (2, 32, 77, 64)]:
        q = rng.standard_normal((B, C, N)).astype(np.float32)
        k = rng.standard_normal((B, C, M)).astype(np.float32)
        v = rng.uniform(-1, 1, (B, 3, M)).astype(np.float32)
        s = np.einsum("bcn,bcm->bnm", q.astype(np.float64), k.astype(np.float64)) / np.sqrt(C)
        s = np.exp(s - s.max(axis=2, keepdims=True))
        s /= s.sum(axis=2, keepdims=True)
        want = np.einsum("bdm,bnm->bdn", v.astype(np.float64), s)
        got = soft_correspondence(dev(q), dev(k), dev(v)).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6)
    # SVD head end to end (R, t within 1e-5 of the oracle's torch.svd-based restatement)
    B, C, N = 4, 64, 256
    src = rng.uniform(-0.5, 0.5, (B, N, 3)).astype(np.float32)
    tgt = rng.uniform(-0.5, 0.5, (B, N, 3)).astype(np.float32)
    se = rng.standard_normal((B, C, N)).astype(np.float32)
    te = rng.standard_normal((B, C, N)).astype(np.float32)
    R_want, t_want = oracle.svd_head(se, te, src, tgt)
    with torch.no_grad():
        R, t = SVDHead(C).cuda()(dev(se), dev(te), dev(src), dev(tgt))
    np.testing.assert_allclose(R.cpu().numpy(), R_want, rtol=0, atol=1e-5)
    np.testing.assert_allclose(t.cpu().numpy(), t_want, rtol=0, atol=1e-5)


def test_sample_and_group_multi_vs_oracle_composition():
    """utils/ppfnet_util.py:193-243 on the HIP FPS / ball query / gather kernels against the same
    composition of the oracle's restatements (indices exact, PPF features to 1e-5)."""
    from learning3d_amd.utils import sample_and_group, sample_and_group_multi
    rng = np.random.default_rng(51)
    B, N, S, K, r = 2, 300, 40, 12, 0.35
    xyz = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    nrm = rng.standard_normal((B, N, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
    out, grouped, fps = sample_and_group_multi(S, r, K, dev(xyz), dev(nrm), returnfps=True)
    # the reference's FPS starts from a random point (ppfnet_util.py:70-73), so the centres are checked
    # structurally and everything downstream against the oracle GIVEN those centres
    fps_np = fps.cpu().numpy()
    assert all(len(set(row)) == S for row in fps_np)
    new_xyz = oracle.index_points(xyz, fps_np)
    np.testing.assert_array_equal(out["xyz"].cpu().numpy(), new_xyz)
    # ball query with itself_indices (ppfnet_util.py:96-131): the centre is EXCLUDED from its own
    # neighbourhood and used as the padding value; restated here in numpy (expanded squared distance in
    # fp32 as square_distance does, first nsample hits in index order)
    d2 = oracle.square_distance(new_xyz, xyz)
    want_idx = np.empty((B, S, K), np.int64)
    for b_ in range(B):
        for s_ in range(S):
            hits = [j for j in np.nonzero(d2[b_, s_] <= np.float32(r * r))[0] if j != fps_np[b_, s_]][:K]
            want_idx[b_, s_] = hits + [fps_np[b_, s_]] * (K - len(hits))
    d = out["dxyz"].cpu().numpy()
    np.testing.assert_allclose(d, oracle.index_points(xyz, want_idx) - new_xyz[:, :, None, :], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["ppf"][..., 3].cpu().numpy(), np.linalg.norm(d, axis=-1), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(grouped.cpu().numpy() - new_xyz[:, :, None, :], d, rtol=0, atol=1e-6)
    # plain sample_and_group: indices exactly the oracle's query_ball_point
    nx, npts, gx, fps2 = sample_and_group(S, r, K, dev(xyz), dev(nrm), returnfps=True)
    new_xyz2 = oracle.index_points(xyz, fps2.cpu().numpy())
    qidx = oracle.query_ball_point(r, K, xyz, new_xyz2)
    want = np.concatenate([oracle.index_points(xyz, qidx) - new_xyz2[:, :, None, :], oracle.index_points(nrm, qidx)], axis=-1)
    np.testing.assert_allclose(npts.cpu().numpy(), want, rtol=0, atol=1e-6)


# --------------------------------------------------------------------------------------------- EMD
def test_emd_vs_oracle():
    from learning3d_amd.losses.emd import EMDFunction
    a, b = rand((2, 256, 3), 20), rand((2, 256, 3), 21)
    ta, tb = dev(a).requires_grad_(), dev(b).requires_grad_()
    cost = EMDFunction.apply(ta, tb)
    ocost, omatch = oracle.emd_forward(a, b)
    np.testing.assert_allclose(cost.detach().cpu().numpy(), ocost, rtol=1e-3)
    cost.sum().backward()
    g1, g2 = oracle.emd_backward(a, b, omatch)
    np.testing.assert_allclose(ta.grad.cpu().numpy(), g1, rtol=1e-2, atol=1e-3)
    np.testing.assert_allclose(tb.grad.cpu().numpy(), g2, rtol=1e-2, atol=1e-3)


def test_native_library_is_what_ran():
    """The driver records which .so files the test process loaded; assert it here too."""
    from learning3d_amd._lib import LIB_PATH
    with open("/proc/self/maps") as f:
        assert any(LIB_PATH in line for line in f)


# ------------------------------------------------------------------- model-level configs (3 and 5)
def test_dcp_golden(golden):
    """BASELINE config 3 (DCP-v2) on the reference-generated fixture: reference state_dict loads
    unchanged; features, rotation and translation match."""
    from learning3d_amd.models import DCP, DGCNN
    g = golden("dcp_emb64")
    net = _load(DCP(DGCNN(emb_dims=64)), g)
    with torch.no_grad():
        out = net(dev(g["template"]), dev(g["source"]))
    np.testing.assert_allclose(out["r"].cpu().numpy(), g["r"], rtol=1e-3, atol=2e-5)
    # north_star: "SVD rotations within 1e-5 fp32" -- held end to end (fp32 rounding alone moves R by 1.4e-6 on this
    # fixture: tests/test_oracle_golden.py::test_dcp_oracle_port_is_the_reference)
    np.testing.assert_allclose(out["est_R"].cpu().numpy(), g["est_R"], atol=1e-5)
    np.testing.assert_allclose(out["est_t"].cpu().numpy(), g["est_t"], atol=1e-5)
    np.testing.assert_allclose(out["est_T"].cpu().numpy(), g["est_T"], atol=1e-5)
    np.testing.assert_allclose(out["transformed_source"].cpu().numpy(), g["transformed_source"], atol=1e-5)


def test_flash_attention_vs_fp64():
    """l3d_attention_forward (attention.hip) against an fp64 evaluation of utils/transformer.py:17-25 on
    channel-first q, k, v, all three head widths, ragged N / M."""
    from learning3d_amd._lib import lib, check, ptr, stream_ptr
    rng = np.random.default_rng(41)
    for (B, H, D, N, M) in [(2, 4, 128, 256, 256), (1, 2, 64, 200, 333), (2, 1, 32, 128, 100), (1, 4, 128, 1024, 1024)]:
        q = rng.standard_normal((B, H, D, N)).astype(np.float32)
        k = rng.standard_normal((B, H, D, M)).astype(np.float32)
        v = rng.standard_normal((B, H, D, M)).astype(np.float32)
        s = np.einsum("bhdn,bhdm->bhnm", q.astype(np.float64), k.astype(np.float64)) / np.sqrt(D)
        s = np.exp(s - s.max(axis=-1, keepdims=True))
        s /= s.sum(axis=-1, keepdims=True)
        want = np.einsum("bhdm,bhnm->bhdn", v.astype(np.float64), s)
        qd, kd, vd = dev(q.reshape(B, H * D, N)), dev(k.reshape(B, H * D, M)), dev(v.reshape(B, H * D, M))
        out = torch.empty_like(qd)
        check(lib().l3d_attention_forward_strided(ptr(qd), ptr(kd), ptr(vd), B, H, D, N, M, H * D * N, H * D * M, H * D * M,
                                                  float(1 / np.sqrt(D)), ptr(out), stream_ptr()), "l3d_attention_forward_strided")
        np.testing.assert_allclose(out.cpu().numpy().reshape(B, H, D, N), want, rtol=1e-5, atol=2e-6)
        # the f16x2 kernel (attention_f16b.hip): same bar, its error against fp64 within 2x (max) / 1.5x (rms) of the bf16x3
        # kernel's own; fp32 context and the plane image of it
        ws = torch.zeros(4, dtype=torch.int32, device=qd.device)
        e3 = out.cpu().numpy().reshape(B, H, D, N) - want
        outb = torch.empty_like(qd)
        imgb = torch.empty(lib().l3d_f16_image_bytes(1, B * N, H * D), dtype=torch.uint8, device=qd.device)
        check(lib().l3d_attention_forward_f16b(ptr(qd), ptr(kd), ptr(vd), B, H, D, N, M, H * D * N, H * D * M, H * D * M,
                                               float(1 / np.sqrt(D)), ptr(ws), 0, ptr(outb), ptr(imgb), stream_ptr()), "l3d_attention_forward_f16b")
        gotb = outb.cpu().numpy().reshape(B, H, D, N)
        np.testing.assert_allclose(gotb, want, rtol=1e-5, atol=2e-6)
        eb = gotb - want
        assert np.abs(eb).max() <= 2.0 * np.abs(e3).max() + 1e-9 and np.sqrt((eb ** 2).mean()) <= 1.5 * np.sqrt((e3 ** 2).mean()) + 1e-10, \
            (B, H, D, N, M, np.abs(eb).max(), np.abs(e3).max())
        raw = imgb.cpu().numpy()
        pb = (H * D // 8) * B * N * 16
        ph = raw[:pb].view(np.float16).reshape(H * D // 8, B * N, 8).astype(np.float64)
        pm = raw[pb:2 * pb].view(np.float16).reshape(H * D // 8, B * N, 8).astype(np.float64)
        xinv = float(raw[2 * pb:2 * pb + 4].view(np.float32)[0])
        dec = ((ph + pm / 4096.0) * xinv).transpose(1, 0, 2).reshape(B, N, H * D).transpose(0, 2, 1).reshape(B, H, D, N)
        np.testing.assert_allclose(dec, gotb.astype(np.float64), rtol=2.0 ** -21, atol=np.abs(gotb).max() * 2.0 ** -30)
        # ... and with an unscaled residual plane (maxima_ready bit 1: what the two-plane form of l3d_pointwise_conv_f16 reads)
        check(lib().l3d_attention_forward_f16b(ptr(qd), ptr(kd), ptr(vd), B, H, D, N, M, H * D * N, H * D * M, H * D * M,
                                               float(1 / np.sqrt(D)), ptr(ws), 2, None, ptr(imgb), stream_ptr()), "l3d_attention_forward_f16b")
        raw2 = imgb.cpu().numpy()
        assert np.array_equal(raw2[:pb], raw[:pb]) and np.array_equal(raw2[2 * pb:2 * pb + 4], raw[2 * pb:2 * pb + 4])
        pm2 = raw2[pb:2 * pb].view(np.float16).reshape(H * D // 8, B * N, 8).astype(np.float64)
        dec2 = ((ph + pm2) * xinv).transpose(1, 0, 2).reshape(B, N, H * D).transpose(0, 2, 1).reshape(B, H, D, N)
        np.testing.assert_allclose(dec2, gotb.astype(np.float64), rtol=2.0 ** -21, atol=np.abs(gotb).max() * 2.0 ** -25)
    # operands far from unit scale: the per-tensor power-of-two scaling must keep fp32-level accuracy (and not overflow fp16)
    for sq, sk, sv in ((1e-4, 3e2, 1e3), (5e3, 1e-3, 1e-5)):
        B, H, D, N, M = 1, 2, 64, 256, 384
        q = (rng.standard_normal((B, H, D, N)) * sq).astype(np.float32)
        k = (rng.standard_normal((B, H, D, M)) * sk).astype(np.float32)
        v = (rng.standard_normal((B, H, D, M)) * sv).astype(np.float32)
        sc = 1.0 / (np.sqrt(D) * sq * sk)
        s = np.einsum("bhdn,bhdm->bhnm", q.astype(np.float64), k.astype(np.float64)) * sc
        s = np.exp(s - s.max(axis=-1, keepdims=True))
        s /= s.sum(axis=-1, keepdims=True)
        want = np.einsum("bhdm,bhnm->bhdn", v.astype(np.float64), s)
        qd, kd, vd = dev(q.reshape(B, H * D, N)), dev(k.reshape(B, H * D, M)), dev(v.reshape(B, H * D, M))
        ws = torch.zeros(4, dtype=torch.int32, device=qd.device)
        outb = torch.empty_like(qd)
        check(lib().l3d_attention_forward_f16b(ptr(qd), ptr(kd), ptr(vd), B, H, D, N, M, H * D * N, H * D * M, H * D * M,
                                               float(sc), ptr(ws), 0, ptr(outb), None, stream_ptr()), "l3d_attention_forward_f16b")
        np.testing.assert_allclose(outb.cpu().numpy().reshape(B, H, D, N), want, rtol=2e-5, atol=2e-6 * sv)


def test_add_transposed_residual():
    from learning3d_amd._lib import lib, check, ptr, stream_ptr
    rng = np.random.default_rng(55)
    for (B, N, C) in [(2, 1024, 512), (3, 77, 130), (1, 1, 1), (2, 33, 31)]:
        x = dev(rng.standard_normal((B, N, C)).astype(np.float32)); y = dev(rng.standard_normal((B, C, N)).astype(np.float32))
        out = torch.empty_like(x)
        check(lib().l3d_add_transposed(ptr(x), ptr(y), B, N, C, ptr(out), stream_ptr()), "l3d_add_transposed")
        assert torch.equal(out, x + y.transpose(1, 2))


def test_transformer_fast_linear_path_matches_torch_path():
    """DCP's pointer network: the no-grad GPU path (Linear / feed-forward layers on the bf16x3 conv kernel,
    channel-first projections) against the same module evaluated in fp64 on the CPU
    (utils/transformer.py:14-243 op order)."""
    import copy
    from learning3d_amd.utils.transformer import Transformer
    torch.manual_seed(12)
    net = Transformer(256, 1, 0.0, 512, 4).eval()
    ref = copy.deepcopy(net).double()
    rng = np.random.default_rng(13)
    a = rng.standard_normal((2, 256, 128)).astype(np.float32)
    b = rng.standard_normal((2, 256, 128)).astype(np.float32)
    with torch.no_grad():
        want = ref(torch.from_numpy(a).double(), torch.from_numpy(b).double())
        got = net.cuda()(dev(a), dev(b))
    att = net.model.encoder.layers[0].self_attn
    assert getattr(att, "_l3d_fused", None), "fused q|k|v projection not taken"
    assert getattr(att.linears[-1], "_l3d_split", None) is not None, "fast linear path not taken"
    for g_, w_ in zip(got, want):
        np.testing.assert_allclose(g_.cpu().numpy(), w_.numpy(), rtol=1e-4, atol=1e-5)


def test_plane_output_bound_cache_follows_parameter_updates():
    """_fused._plane_obs caches {max|shift|, max|scale|} per (tensor, version): an in-place parameter update (an optimizer step)
    must be seen -- a stale bound would place the output planes for the OLD magnitudes (overflow or lost precision)."""
    from learning3d_amd.models import _fused
    rng = np.random.default_rng(5)
    B, N, C0, C1 = 1, 256, 64, 256
    x = rng.standard_normal((B, N, C0)).astype(np.float32)
    w = (rng.standard_normal((C1, C0)) / C0 ** 0.5).astype(np.float32)
    ximg = _fused.split_rows_f16(dev(x)); wimg = _fused.split_weights_f16(dev(w))
    shift = dev(rng.standard_normal(C1).astype(np.float32))
    w2 = (rng.standard_normal((C1, C1)) / C1 ** 0.5).astype(np.float32); w2img = _fused.split_weights_f16(dev(w2))
    for scale_up in (1.0, 3000.0, 1e-3):
        shift.mul_(scale_up)                                           # in place: same storage, new version
        img = _fused.pointwise_conv_f16(ximg, B, N, wimg, C0, C1, None, shift, relu=True, out_planes=True)
        y = _fused.pointwise_conv_f16(img, B, N, w2img, C1, C1, None, None)
        h = np.maximum(x.astype(np.float64) @ w.astype(np.float64).T + shift.cpu().numpy().astype(np.float64), 0)
        want = (h @ w2.astype(np.float64).T).transpose(0, 2, 1)
        assert np.abs(y.cpu().numpy() - want).max() <= 2e-5 * np.abs(want).max(), scale_up
    _fused.check_range(sync=True)


def test_linear_rows_kernel():
    """l3d_linear_rows (PCN's fully connected decoder, models/pcn.py:132-137: a Linear over as many rows as there are clouds)
    against fp64: ragged row counts, Cout not a multiple of the workgroup's 16 channels, with and without bias / ReLU."""
    from learning3d_amd.models import _fused
    rng = np.random.default_rng(88)
    for (R, Cin, Cout, relu, bias) in [(64, 1024, 1024, True, True), (64, 1024, 3072, False, True), (1, 256, 5, False, False),
                                       (70, 512, 40, True, True), (200, 256, 33, False, True), (3, 128, 16, True, True)]:
        lin = torch.nn.Linear(Cin, Cout, bias=bias).cuda()
        x = dev(rng.standard_normal((R, Cin)).astype(np.float32))
        with torch.no_grad():
            got = _fused.linear_rows(x, lin, relu)
        want = x.double().cpu() @ lin.weight.detach().double().cpu().t()
        if bias:
            want = want + lin.bias.detach().double().cpu()
        if relu:
            want = want.clamp_min(0)
        assert got.shape == (R, Cout) and (got.is_contiguous() or Cin % 256)          # (3, 128, 16): the conv-kernel route
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5 * float(want.abs().max()))


def test_transformer_channel_first_pass():
    """utils/transformer.py, Transformer._pass_cf: a whole encoder-decoder pass in the [B,C,N] layout of the GEMMs (channel-first
    LayerNorm straight to planes, residual connections in the epilogues of the output projection and the feed-forward's second
    layer) against the module-by-module route and against the reference's op sequence in fp64 (utils/transformer.py:14-243);
    the pieces too: l3d_layernorm_planes_cf vs the row kernel's formula in fp64, l3d_pointwise_conv_f16's residual epilogue vs conv + add."""
    import copy
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    from learning3d_amd.models import _fused
    from learning3d_amd.utils import transformer as T
    rng = np.random.default_rng(77)
    # ---- LayerNorm over the channels of a channel-first tensor
    for (B, C, N) in [(2, 512, 256), (1, 256, 320), (3, 128, 70)]:
        x = (rng.standard_normal((B, C, N)) * rng.uniform(0.1, 3.0, (B, 1, N)) + rng.uniform(-2, 2, (B, 1, N))).astype(np.float32)
        a = rng.uniform(0.5, 1.5, C).astype(np.float32); b = rng.uniform(-0.5, 0.5, C).astype(np.float32)
        x64 = x.astype(np.float64)
        want = a[None, :, None] * (x64 - x64.mean(1, keepdims=True)) / (x64.std(1, ddof=1, keepdims=True) + 1e-6) + b[None, :, None]
        y = torch.empty((B, C, N), device="cuda"); img = torch.empty(lib().l3d_f16_image_bytes(1, B * N, C), dtype=torch.uint8, device="cuda")
        tx, ta, tb = dev(x), dev(a), dev(b)                       # held: a temporary's block would be handed to the next allocation
        check(lib().l3d_layernorm_planes_cf(ptr(tx), ptr(ta), ptr(tb), 1e-6, B, C, N, ptr(y), ptr(img), 0, stream_ptr()), "ln cf")
        np.testing.assert_allclose(y.cpu().numpy(), want, rtol=2e-6, atol=2e-6)
        # the image against the row kernel's image of the same values laid out [B,N,C]
        rows = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda()
        y2 = torch.empty_like(rows); img2 = torch.empty_like(img)
        check(lib().l3d_layernorm_planes(ptr(rows), ptr(ta), ptr(tb), 1e-6, B * N, C, ptr(y2), ptr(img2), stream_ptr()), "ln rows")
        pb = (C // 8) * B * N * 16
        h1 = img[:pb].view(torch.float16).float(); h2 = img2[:pb].view(torch.float16).float()
        assert torch.equal(img[2 * pb:2 * pb + 4], img2[2 * pb:2 * pb + 4])                      # the same plane scale
        assert (h1 - h2).abs().max().item() <= 2.0 ** -10 * h2.abs().max().item()               # high planes: at most one fp16 ulp apart
    # ---- residual epilogue
    B, N, C0, C1 = 2, 512, 512, 256
    x = rng.standard_normal((B, N, C0)).astype(np.float32)
    ximg = _fused.split_rows_f16(dev(x))
    w = (rng.standard_normal((C1, C0)) / C0 ** 0.5).astype(np.float32); sh = (rng.standard_normal(C1) * 0.3).astype(np.float32)
    wimg = _fused.split_weights_f16(dev(w))
    res = dev(rng.standard_normal((B, C1, N)).astype(np.float32))
    plain = _fused.pointwise_conv_f16(ximg, B, N, wimg, C0, C1, None, dev(sh))
    fused = _fused.pointwise_conv_f16(ximg, B, N, wimg, C0, C1, None, dev(sh), residual=res)
    assert torch.equal(fused, res + plain)
    # ---- the whole pass
    torch.manual_seed(12)
    net = T.Transformer(512, 1, 0.0, 1024, 4).eval()
    for prm in net.parameters():
        if prm.dim() == 1 and prm.numel() == 512:
            prm.data.add_(torch.randn(512) * 0.1)
    ref = copy.deepcopy(net).double()
    a_ = rng.standard_normal((2, 512, 256)).astype(np.float32); b_ = rng.standard_normal((2, 512, 512)).astype(np.float32)
    net = net.cuda()
    import learning3d_amd._lib as _lib
    with torch.no_grad():
        want = ref(torch.from_numpy(a_).double(), torch.from_numpy(b_).double())
        T.CHANNEL_FIRST_PASS = False
        try:
            mod = net(dev(a_), dev(b_))
        finally:
            T.CHANNEL_FIRST_PASS = True
        # the pass with its plane images' residual planes unscaled (every projection on the two-plane form of the kernel: the default) and
        # with the scaled images of rounds 3-5
        for two in (True, False):
            prev, T.TWO_PLANE_IMAGES = T.TWO_PLANE_IMAGES, two
            _lib.LAUNCH_LOG = log = []
            try:
                got = net(dev(a_), dev(b_))
            finally:
                _lib.LAUNCH_LOG = None
                T.TWO_PLANE_IMAGES = prev
            tag = "[two-plane]" if two else ""
            assert "l3d_layernorm_planes_cf" in log and "l3d_add_transposed" not in log, sorted(set(log))
            for name in ("l3d_pointwise_conv_f16[residual]", "l3d_pointwise_conv_f16[planes]"):
                assert name + tag in log and (name + "[two-plane]" in log) == two, sorted(set(log))
            assert ("l3d_pointwise_conv_f16[absmax][two-plane]" in log) == two and ("l3d_pointwise_conv_f16[absmax]" in log) != two, sorted(set(log))
            for g_, m_, w_ in zip(got, mod, want):
                assert g_.shape == w_.shape and g_.is_contiguous()
                np.testing.assert_allclose(g_.cpu().numpy(), w_.numpy(), rtol=1e-4, atol=2e-5)
                assert (g_ - m_).abs().max().item() <= 2e-5 * m_.abs().max().item()
    # ---- the two-plane form piece by piece: an unscaled LayerNorm image through the projection in its three output modes, against fp64
    B, C, N, C1 = 2, 512, 512, 512
    x = (rng.standard_normal((B, C, N)) * rng.uniform(0.1, 3.0, (B, 1, N))).astype(np.float32)
    a = rng.uniform(0.5, 1.5, C).astype(np.float32); b = rng.uniform(-0.5, 0.5, C).astype(np.float32)
    w = (rng.standard_normal((C1, C)) / C ** 0.5).astype(np.float32); sh = (rng.standard_normal(C1) * 0.3).astype(np.float32)
    w2 = (rng.standard_normal((C1, C1)) / C1 ** 0.5).astype(np.float32)
    x64 = x.astype(np.float64)
    ln64 = a[None, :, None] * (x64 - x64.mean(1, keepdims=True)) / (x64.std(1, ddof=1, keepdims=True) + 1e-6) + b[None, :, None]
    y64 = np.einsum("oc,bcn->bon", w.astype(np.float64), ln64) + sh[None, :, None]
    tx, ta, tb, tsh = dev(x), dev(a), dev(b), dev(sh)
    wimg, w2img = _fused.split_weights_f16(dev(w)), _fused.split_weights_f16(dev(w2))
    res = dev(rng.standard_normal((B, C1, N)).astype(np.float32))
    outs = {}
    for flags in (1, 0):
        img = torch.empty(lib().l3d_f16_image_bytes(1, B * N, C), dtype=torch.uint8, device="cuda")
        check(lib().l3d_layernorm_planes_cf(ptr(tx), ptr(ta), ptr(tb), 1e-6, B, C, N, None, ptr(img), flags, stream_ptr()), "ln cf")
        ws = torch.zeros(4, dtype=torch.int32, device="cuda")
        y_amax = _fused.pointwise_conv_f16(img, B, N, wimg, C, C1, None, tsh, amax=(ws, 256), unscaled=bool(flags))
        y_res = _fused.pointwise_conv_f16(img, B, N, wimg, C, C1, None, tsh, residual=res, unscaled=bool(flags))
        him = _fused.pointwise_conv_f16(img, B, N, wimg, C, C1, None, tsh, relu=True, out_planes=True, unscaled=bool(flags))
        y_chain = _fused.pointwise_conv_f16(him, B, N, w2img, C1, C1, None, None, unscaled=bool(flags))
        scale = np.abs(y64).max()
        assert np.abs(y_amax.cpu().numpy() - y64).max() <= 4e-6 * scale, flags
        assert torch.equal(y_res, res + y_amax), flags
        mx = ws.view(torch.float32).cpu().numpy()
        assert mx[0] == np.abs(y_amax[:, :256].cpu().numpy()).max() and mx[1] == np.abs(y_amax[:, 256:].cpu().numpy()).max(), flags
        c64 = np.einsum("oc,bcn->bon", w2.astype(np.float64), np.maximum(y64, 0))
        assert np.abs(y_chain.cpu().numpy() - c64).max() <= 4e-6 * np.abs(c64).max(), flags
        outs[flags] = (y_amax, y_chain)
    assert (outs[1][0] - outs[0][0]).abs().max().item() <= 2e-6 * scale                 # the two residual conventions agree to fp32 level
    _fused.check_range(sync=True)


def test_flownet3d_set_abstraction_vs_oracle():
    """BASELINE config 5's layer (sa1: npoint=1024 -> here 128, r=0.5, K=16, mlp [32,32,64]) against the
    oracle composition FPS -> gather -> ball query -> group -> torch-CPU conv stack."""
    from learning3d_amd.models import PointNetSetAbstraction
    rng = np.random.default_rng(11)
    B, N, S = 2, 1024, 128
    xyz = np.clip(rng.standard_normal((B, 3, N)), -2, 2).astype(np.float32)
    feat = rng.uniform(0, 1, (B, 3, N)).astype(np.float32)
    torch.manual_seed(7)
    sa = PointNetSetAbstraction(npoint=S, radius=0.5, nsample=16, in_channel=3, mlp=[32, 32, 64], group_all=False).eval()
    for m in sa.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.1, 0.1); m.running_var.uniform_(0.8, 1.2)
    # oracle composition on CPU
    xyz_t = np.ascontiguousarray(xyz.transpose(0, 2, 1))
    fps = oracle.furthest_point_sampling(xyz_t, S)
    new_xyz = oracle.gather_points(xyz, fps)                                       # [B,3,S]
    idx = oracle.ball_query(0.5, 16, xyz_t, np.ascontiguousarray(new_xyz.transpose(0, 2, 1)))
    g_xyz = oracle.group_points(xyz, idx) - new_xyz[:, :, :, None]
    h = torch.from_numpy(np.concatenate([g_xyz, oracle.group_points(feat, idx)], axis=1))
    with torch.no_grad():
        for conv, bn in zip(sa.mlp_convs, sa.mlp_bns):
            h = torch.relu(bn(conv(h)))
        want = h.max(-1)[0].numpy()
        sa = sa.cuda()
        got_xyz, got = sa(dev(xyz), dev(feat))
    assert np.array_equal(got_xyz.cpu().numpy(), new_xyz)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-5)


class _lib_log:
    """`with _lib_log() as log:` -- the names of the C-ABI calls made inside"""

    def __enter__(self):
        from learning3d_amd import _lib
        _lib.LAUNCH_LOG = []
        return _lib.LAUNCH_LOG

    def __exit__(self, *exc):
        from learning3d_amd import _lib
        _lib.LAUNCH_LOG = None


def test_ball_query_cell_list_equals_scanning_kernels():
    """l3d_ball_query with scratch (grouping.hip bq_cells_*: counting sort of the cloud into cells of edge >= r, a wave per centroid
    over its 27 cells keeping the nsample smallest hit indices) against the scanning kernels it replaces for n >= 2048: the same
    int32 indices in the same order with the same padding -- Gaussian / uniform / clipped (duplicate points) / flat (one cell
    thick) clouds, radii from "every ball empty" to "every ball holds the whole cloud", nsample 1 ... 64, centroids that are cloud
    points, arbitrary points and points far outside the cloud's box, a ragged centroid count."""
    from learning3d_amd.utils import pointnet2_utils as P
    rng = np.random.default_rng(123)
    cases = []
    for (B, N, S, K, r, kind) in [(2, 8192, 1024, 16, 0.5, "gauss"), (3, 2048, 333, 32, 0.2, "uniform"), (2, 4096, 257, 64, 0.35, "clip"),
                                  (1, 3000, 100, 1, 0.1, "uniform"), (2, 2500, 64, 8, 1e-4, "uniform"), (1, 2048, 50, 16, 50.0, "gauss"),
                                  (2, 6000, 200, 16, 0.3, "flat"), (1, 5000, 77, 24, 0.25, "outside")]:
        if kind == "gauss":
            xyz = rng.standard_normal((B, N, 3))
        elif kind == "clip":
            xyz = np.clip(rng.standard_normal((B, N, 3)), -0.7, 0.7)
        elif kind == "flat":
            xyz = rng.uniform(-1, 1, (B, N, 3)); xyz[:, :, 2] = 0.25
        else:
            xyz = rng.uniform(-1, 1, (B, N, 3))
        xyz = xyz.astype(np.float32)
        if kind == "outside":
            new = rng.uniform(-3, 3, (B, S, 3)).astype(np.float32)
            new[:, :5] = 1.0e6
        else:
            new = np.stack([xyz[b][rng.choice(N, S, replace=False)] for b in range(B)])
            new[:, ::7] += rng.uniform(-0.05, 0.05, new[:, ::7].shape).astype(np.float32)
        cases.append((dev(xyz), dev(new), K, r, kind))
    for xyz, new, K, r, kind in cases:
        old = P.BALL_QUERY_CELLS
        try:
            P.BALL_QUERY_CELLS = True
            a = P.ball_query(r, K, xyz, new)
            P.BALL_QUERY_CELLS = False
            b = P.ball_query(r, K, xyz, new)
        finally:
            P.BALL_QUERY_CELLS = old
        assert a.dtype == torch.int32 and torch.equal(a, b), (kind, tuple(xyz.shape), K, r, int((a != b).sum()))
    # and something was actually found / not found where it should be
    xyz, new, K, r, _ = cases[0]
    idx = P.ball_query(r, K, xyz, new).long()
    d = (xyz.gather(1, idx.reshape(2, -1, 1).expand(-1, -1, 3)).view(2, 1024, K, 3) - new.unsqueeze(2)).square().sum(-1)
    assert bool(((d < r * r) | (idx == 0)).all())


def test_fused_set_abstraction_kernel_every_instantiation_vs_fp64():
    """l3d_sa_mlp3_fused (sa_fused.hip): gather + three conv / BatchNorm / ReLU layers + max over K in one kernel, for D = 0, 3 and 9
    feature channels (input padded to 8 / 16), both width triples, K = 8, 16, 32 and 64 (two centroids per row tile, one, and a
    running maximum over two / four tiles), S not a multiple of the 64 centroids of a workgroup.  Against the grouped input pushed
    through the same layers in fp64: an fp32 fma chain per output, held to rtol 1e-5 / atol 1e-6 here (the path's bar is 1e-4 /
    1e-5); and the module's two routes (fused, group + conv launches) against each other."""
    from learning3d_amd.models import PointNetSetAbstraction, _fused
    from learning3d_amd.utils import pointnet2_utils as P
    rng = np.random.default_rng(77)
    B, N = 2, 700
    for D, widths, K, S in ((3, [32, 32, 64], 16, 100), (0, [32, 32, 64], 8, 64), (9, [32, 32, 64], 32, 70), (3, [64, 64, 128], 64, 65),
                            (12, [64, 64, 128], 16, 130), (3, [64, 64, 128], 8, 31)):
        torch.manual_seed(D + K)
        sa = PointNetSetAbstraction(npoint=S, radius=0.6, nsample=K, in_channel=D, mlp=list(widths), group_all=False).eval()
        for m in sa.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.3, 0.3)
        sa = sa.cuda()
        xyz = dev(np.clip(rng.standard_normal((B, 3, N)), -2, 2).astype(np.float32))
        feat = dev(rng.uniform(-1, 1, (B, D, N)).astype(np.float32)) if D else None
        with _lib_log() as log, torch.no_grad():
            new_xyz, got = sa(xyz, feat)
        assert "l3d_sa_mlp3_fused" in log and "l3d_group_concat" not in log, log
        with torch.no_grad():
            xyz_t = xyz.permute(0, 2, 1).contiguous()
            idx = P.ball_query(0.6, K, xyz_t, new_xyz.transpose(1, 2).contiguous())
            h = (P.grouping_operation(xyz, idx) - new_xyz.unsqueeze(-1)).double()
            if D:
                h = torch.cat([h, P.grouping_operation(feat, idx).double()], 1)
            for conv, bn in zip(sa.mlp_convs, sa.mlp_bns):
                w, sc, sh = _fused.fold_conv_bn(conv, bn)
                h = torch.relu(torch.einsum("oc,bcsk->bosk", w.double(), h) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
            want = h.max(-1)[0]
            np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=str((D, widths, K, S)))
            old = _fused.SA_FUSED
            _fused.SA_FUSED = False
            try:
                with _lib_log() as log2:
                    _, unfused = sa(xyz, feat)
            finally:
                _fused.SA_FUSED = old
            assert "l3d_sa_mlp3_fused" not in log2 and "l3d_group_concat" in log2, log2
            np.testing.assert_allclose(got.cpu().numpy(), unfused.cpu().numpy(), rtol=1e-4, atol=1e-5)
    # a stack the kernel does not take goes to the layer kernels
    sa = PointNetSetAbstraction(npoint=32, radius=0.6, nsample=16, in_channel=3, mlp=[32, 48, 64], group_all=False).eval().cuda()
    with _lib_log() as log, torch.no_grad():
        sa(xyz, dev(rng.uniform(-1, 1, (B, 3, N)).astype(np.float32)))
    assert "l3d_sa_mlp3_fused" not in log, log


def test_group_concat2_matches_composition():
    """l3d_group_concat2 == grouping_operation x2 + broadcast subtraction + repeat + torch.cat (flownet3d.py:125-180,
    :182-242), both channel orders, with and without the broadcast centre features."""
    from learning3d_amd.utils import pointnet2_utils as P
    from learning3d_amd._lib import lib, check, ptr, stream_ptr
    rng = np.random.default_rng(66)
    B, N, S, K, C, C1 = 2, 300, 77, 9, 13, 5
    xyz = dev(rng.standard_normal((B, N, 3)).astype(np.float32)); new = dev(rng.standard_normal((B, S, 3)).astype(np.float32))
    feat = dev(rng.standard_normal((B, C, N)).astype(np.float32)); cen = dev(rng.standard_normal((B, C1, S)).astype(np.float32))
    idx = dev(rng.integers(0, N, (B, S, K)).astype(np.int32))
    pos_diff = P.grouping_operation(xyz.transpose(1, 2).contiguous(), idx) - new.transpose(1, 2).reshape(B, 3, S, 1)
    fg = P.grouping_operation(feat, idx)
    cb = cen.view(B, C1, S, 1).repeat(1, 1, 1, K)
    for order, c1, want in [(0, C1, torch.cat([pos_diff, fg, cb], 1)), (1, 0, torch.cat([fg, pos_diff], 1)),
                            (1, C1, torch.cat([fg, pos_diff, cb], 1))]:
        out = torch.empty((B, 3 + C + c1, S, K), device="cuda")
        check(lib().l3d_group_concat2(ptr(xyz), ptr(new), ptr(feat), ptr(cen) if c1 else None, ptr(idx), B, N, S, K, C, c1,
                                      order, ptr(out), stream_ptr()), "l3d_group_concat2")
        assert torch.equal(out, want)


def test_flownet3d_forward_runs():
    from learning3d_amd.models import FlowNet3D
    torch.manual_seed(0)
    net = FlowNet3D().cuda().eval()
    g = torch.Generator().manual_seed(3)
    pc1 = torch.clamp(torch.randn((2, 3, 2048), generator=g), -2, 2).cuda()
    pc2 = (pc1 + 0.05 * torch.randn((2, 3, 2048), generator=g).cuda()).contiguous()
    f1 = torch.rand((2, 3, 2048), generator=g).cuda()
    f2 = torch.rand((2, 3, 2048), generator=g).cuda()
    with torch.no_grad():
        sf = net(pc1, pc2, f1, f2)
    assert sf.shape == (2, 3, 2048) and torch.isfinite(sf).all()
    # fused and per-layer differentiable routes agree
    from learning3d_amd.models import _fused
    with _fused.per_layer_route():
        sf2 = net(pc1, pc2, f1.clone().requires_grad_(), f2)
    np.testing.assert_allclose(sf.cpu().numpy(), sf2.detach().cpu().numpy(), rtol=2e-3, atol=2e-4)


def test_config4_pcn_chamfer_full_size():
    """BASELINE config 4: PCN(detailed) on partial [64,2048,3] -> fine [64,16384,3]; Chamfer loss against
    gt [64,16384,3] (17.2 G pair evaluations per direction).  Full size through properties + one cloud
    against the oracle (the reference's own fallback would need 3 x 206 GB here)."""
    from learning3d_amd.models import PCN
    from learning3d_amd.losses import ChamferDistanceLoss
    from learning3d_amd.losses.chamfer_distance import ChamferDistanceFunction
    torch.manual_seed(0)
    net = PCN(emb_dims=1024, num_coarse=1024, grid_size=4, detailed_output=True).cuda().eval()
    g = torch.Generator().manual_seed(0)
    partial = (torch.rand((64, 2048, 3), generator=g) - 0.5).cuda()
    gt = (torch.rand((64, 16384, 3), generator=g) - 0.5).cuda()
    with torch.no_grad():
        out = net(partial)
        fine = out["fine_output"]
        assert fine.shape == (64, 16384, 3) and torch.isfinite(fine).all()
        loss = ChamferDistanceLoss()(gt, fine)
        d1, d2 = ChamferDistanceFunction.apply(gt, fine.contiguous())
    assert loss.ndim == 0 and torch.isfinite(loss)
    o1, o2, _, _ = oracle.chamfer_forward(gt[:1].cpu().numpy(), fine[:1].contiguous().cpu().numpy())
    assert np.array_equal(d1[:1].cpu().numpy(), o1) and np.array_equal(d2[:1].cpu().numpy(), o2)
    want = (torch.sqrt(d1).double().mean() + torch.sqrt(d2).double().mean()) / 2
    assert abs(float(loss) - float(want)) < 1e-6


def test_edge_cases_small_and_degenerate():
    """Ragged / degenerate inputs: single point clouds, k == N, nsample > N, duplicate points (exact
    ties), M != N Chamfer with one-point clouds."""
    import learning3d_amd.utils as U
    from learning3d_amd.utils import pointnet2_utils as P
    from learning3d_amd.losses.chamfer_distance import ChamferDistanceFunction
    # k == N == 1 .. small
    x = dev(rand((2, 1, 3), 1))
    assert (U.knn(x.permute(0, 2, 1), 1) == 0).all()
    x = dev(rand((1, 5, 3), 2))
    idx = U.knn(x.permute(0, 2, 1), 5).cpu().numpy()
    assert sorted(idx[0, 0].tolist()) == [0, 1, 2, 3, 4] and np.array_equal(idx, oracle.knn(x.cpu().numpy(), 5))
    # duplicate points: exact ties resolve to the lower index, like the oracle's contract
    pts = rand((1, 40, 3), 3)
    pts[0, 20:] = pts[0, :20]
    idx = U.knn(dev(pts).permute(0, 2, 1), 6).cpu().numpy()
    assert np.array_equal(idx, oracle.knn(pts, 6))
    # ball query: nsample > N, radius covering everything / nothing
    xyz = dev(rand((1, 7, 3), 4))
    bq = P.ball_query(10.0, 16, xyz, xyz[:, :3].contiguous()).cpu().numpy()
    assert np.array_equal(bq, oracle.ball_query(10.0, 16, xyz.cpu().numpy(), xyz[:, :3].cpu().numpy()))
    q = U.query_ball_point(1e-6, 4, xyz, xyz[:, :2].contiguous()).cpu().numpy()      # only the point itself
    assert np.array_equal(q, oracle.query_ball_point(1e-6, 4, xyz.cpu().numpy(), xyz[:, :2].cpu().numpy()))
    # FPS with npoint == N
    f = P.furthest_point_sample(xyz, 7).cpu().numpy()
    assert sorted(f[0].tolist()) == list(range(7))
    # Chamfer with one-point clouds and M != N
    a, b = dev(rand((2, 1, 3), 5)), dev(rand((2, 9, 3), 6))
    d1, d2 = ChamferDistanceFunction.apply(a, b)
    o1, o2, _, _ = oracle.chamfer_forward(a.cpu().numpy(), b.cpu().numpy())
    assert np.array_equal(d1.cpu().numpy(), o1) and np.array_equal(d2.cpu().numpy(), o2)
    # K13 with k > m: slots beyond m hold (+inf -> sqrt inf, index 0) like best[]=1e40, besti[]=0
    d, i = P.knn(5, dev(rand((1, 4, 3), 7)), dev(rand((1, 3, 3), 8)))
    assert torch.isinf(d[..., 3:]).all() and (i[..., 3:] == 0).all()


# ------------------------------------------------ round 2: reference-run pins for PCN, config 1, pointconv_util
def test_pcn_reference_golden(golden):
    """PCN against the REFERENCE model's own output (tests/golden/make_golden.py runs models/pcn.py on the CPU with
    seeded_params weights; the same function rebuilds them here by key).  Both routes: the fused inference route
    (no_grad) and the reference-order torch route (grad enabled)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from seeded import seeded_params
    from learning3d_amd.models import PCN
    g = golden("pcn_seeded")
    net = PCN(emb_dims=1024, num_coarse=64, grid_size=2, detailed_output=True)
    assert sorted(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    net = seeded_params(net, int(g["seed"])).cuda().eval()
    x = dev(g["x"])
    with torch.no_grad():
        fused = net(x)
    from learning3d_amd.models import _fused
    with _fused.per_layer_route():
        ref_route = net(x.clone().requires_grad_())
    for k in ("coarse_output", "fine_output"):
        np.testing.assert_allclose(fused[k].cpu().numpy(), g[k], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ref_route[k].detach().cpu().numpy(), g[k], rtol=1e-4, atol=1e-5)


def test_config1_classifier_checkpoint_logits(golden):
    """BASELINE config 1 (examples/test_pointnet.py:98-118): PointNet classifier with the reference's trained
    checkpoint pretrained/exp_classifier/models/best_model.t7 (stored in the fixture), B=8, N=1024, x ~ U(-1,1)
    seed 0; logits against the reference model run on the CPU.  Logits reach |72|: the bar is 1e-5 relative + 1e-5
    absolute (SURVEY.md 8(d) c1 asks 1e-5), and identical predicted classes."""
    from learning3d_amd.models import Classifier, PointNet
    g = golden("classifier_best_model")
    model = Classifier(feature_model=PointNet(emb_dims=1024, use_bn=True))
    model.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w.")})   # strict: same keys
    model = model.cuda().eval()
    with torch.no_grad():
        logits = model(dev(g["x"])).cpu().numpy()
    np.testing.assert_allclose(logits, g["logits"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(logits.argmax(1), g["logits"].argmax(1))
    np.testing.assert_allclose(logits, oracle.pointnet_classifier_forward_torch(g["x"], {k[2:]: v for k, v in g.items() if k.startswith("w.")}),
                               rtol=1e-5, atol=1e-5)


def test_pointconv_util_golden(golden):
    """utils/pointconv_util.py mirror (BASELINE north_star names the file): FPS from index 0, smallest-k kNN on the
    expanded distance (indices only), fused density, and one PointConvDensitySetAbstraction layer with the reference's
    state_dict -- all against the reference functions run on the CPU."""
    from learning3d_amd.utils import pointconv_util as PC
    g = golden("pointconv_util")
    xyz = dev(g["xyz"])
    fps = PC.farthest_point_sample(xyz, g["fps"].shape[1])
    assert fps.dtype == torch.int64 and np.array_equal(fps.cpu().numpy(), g["fps"])
    new_xyz = PC.index_points(xyz, fps)
    idx = PC.knn_point(16, xyz, new_xyz)
    assert idx.dtype == torch.int64 and np.array_equal(idx.cpu().numpy(), g["knn_idx_sorted"])
    np.testing.assert_allclose(PC.compute_density(xyz, 0.1).cpu().numpy(), g["density"], rtol=1e-5, atol=0)
    sa = PC.PointConvDensitySetAbstraction(npoint=64, nsample=16, in_channel=8, mlp=[16, 32], bandwidth=0.1, group_all=False)
    sa.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w.")})
    sa = sa.cuda().eval()
    with torch.no_grad():
        sxyz, spts = sa(xyz.permute(0, 2, 1), dev(g["feats"]))
    np.testing.assert_allclose(sxyz.cpu().numpy(), g["sa_xyz"], rtol=0, atol=0)
    np.testing.assert_allclose(spts.cpu().numpy(), g["sa_points"], rtol=1e-4, atol=1e-5)
    # ragged sizes against the oracle restatement (rows are nearest-first on both sides)
    rng = np.random.default_rng(3)
    for B, N, S, K in ((1, 20, 7, 20), (3, 333, 65, 9), (2, 1500, 1500, 32)):
        a = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
        q = a[:, :S].copy() if S <= N else rng.uniform(-1, 1, (B, S, 3)).astype(np.float32)
        got = PC.knn_point(K, dev(a), dev(q)).cpu().numpy()
        want = oracle.knn_point_expanded(K, a, q)
        if not np.array_equal(got, want):                # only exact ties of the ranked value may be ordered differently
            d = oracle.square_distance(q, a)
            assert np.array_equal(np.take_along_axis(d, got, -1), np.take_along_axis(d, want, -1))
    with pytest.raises(RuntimeError):
        PC.knn_point(30, dev(a[:, :20]), dev(q))


def test_square_distance_generic_channels():
    from learning3d_amd.utils import square_distance
    rng = np.random.default_rng(4)
    for C in (1, 2, 5, 32):
        s = rng.uniform(-1, 1, (2, 70, C)).astype(np.float32)
        d = rng.uniform(-1, 1, (2, 130, C)).astype(np.float32)
        want = -2 * torch.matmul(torch.from_numpy(s), torch.from_numpy(d).permute(0, 2, 1))
        want += torch.sum(torch.from_numpy(s) ** 2, -1).view(2, 70, 1)
        want += torch.sum(torch.from_numpy(d) ** 2, -1).view(2, 1, 130)
        np.testing.assert_allclose(square_distance(dev(s), dev(d)).cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)


def test_svd_head_is_differentiable_like_the_reference():
    """R and t must carry gradients to the embeddings (DCP trains through the head).  Gradients of
    a scalar of (R, t) w.r.t. the embeddings against autograd through the reference's op sequence (utils/svd.py:13-59)
    evaluated in fp64 on the CPU."""
    from learning3d_amd.utils import SVDHead
    B, C, N = 3, 32, 64
    se, te = rand((B, C, N), 30, -1, 1) * 4, rand((B, C, N), 31, -1, 1) * 4
    src, tgt = rand((B, N, 3), 32, -0.5, 0.5), rand((B, N, 3), 33, -0.5, 0.5)
    gR, gt = rand((B, 3, 3), 34, -1, 1), rand((B, 3), 35, -1, 1)
    head = SVDHead(emb_dims=C, input_shape="bnc").cuda()
    a, b = dev(se).requires_grad_(), dev(te).requires_grad_()
    R, t = head(a, b, dev(src), dev(tgt))
    assert R.requires_grad and t.requires_grad
    ((R * dev(gR)).sum() + (t * dev(gt)).sum()).backward()
    assert torch.isfinite(a.grad).all() and torch.isfinite(b.grad).all() and float(a.grad.abs().max()) > 0
    # reference sequence, fp64 CPU
    a64 = torch.from_numpy(se).double().requires_grad_(); b64 = torch.from_numpy(te).double().requires_grad_()
    s64, t64 = torch.from_numpy(src).double().permute(0, 2, 1), torch.from_numpy(tgt).double().permute(0, 2, 1)
    scores = torch.softmax(torch.matmul(a64.transpose(2, 1), b64) / np.sqrt(C), dim=2)
    corr = torch.matmul(t64, scores.transpose(2, 1))
    sc, cc = s64 - s64.mean(2, keepdim=True), corr - corr.mean(2, keepdim=True)
    H = sc @ cc.transpose(2, 1)
    refl = torch.diag(torch.tensor([1.0, 1.0, -1.0], dtype=torch.float64))
    Rs = []
    for i in range(B):
        u, s_, v = torch.svd(H[i]); r = v @ u.t()
        if torch.det(r.detach()) < 0:
            r = (v @ refl) @ u.t()
        Rs.append(r)
    R64 = torch.stack(Rs)
    t_64 = (torch.matmul(-R64, s64.mean(2, keepdim=True)) + corr.mean(2, keepdim=True)).view(B, 3)
    ((R64 * torch.from_numpy(gR).double()).sum() + (t_64 * torch.from_numpy(gt).double()).sum()).backward()
    np.testing.assert_allclose(R.detach().cpu().numpy(), R64.detach().numpy(), atol=1e-5)
    for got, want in ((a.grad, a64.grad), (b.grad, b64.grad)):
        scale = float(want.abs().max())
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-3, atol=1e-4 * scale)


def test_config3_dcp_full_size_vs_oracle_port():
    """BASELINE config 3 at full size: DCP-v2 (DGCNN emb 512 + Transformer + SVD head), B=32, N=1024, inputs as
    SURVEY.md 8(d) c3 (template U(-0.5,0.5)^3 seed 0; source = R template + t, Euler angles in [0,45 deg]^3,
    t in U(-0.5,0.5)^3), against oracle.dcp_forward_torch -- the fp32 CPU restatement that reproduces the reference's
    golden bit for bit (tests/test_oracle_golden.py::test_dcp_oracle_port_is_the_reference).  R, t are compared where
    the SVD is well conditioned ((s2 +- s3)/s1 >= 1e-2, the survey's criterion)."""
    from learning3d_amd.models import DCP, DGCNN
    torch.manual_seed(5)
    net = DCP(feature_model=DGCNN(emb_dims=512), cycle=False).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
    w = {k: v.numpy() for k, v in net.state_dict().items()}
    B, N = 32, 1024
    template = rand((B, N, 3), 0, -0.5, 0.5)
    ang = rand((B, 3), 1, 0, np.pi / 4)
    cx, cy, cz = np.cos(ang).T; sx, sy, sz = np.sin(ang).T
    Rg = np.zeros((B, 3, 3), np.float32)
    for i in range(B):
        Rx = np.array([[1, 0, 0], [0, cx[i], -sx[i]], [0, sx[i], cx[i]]]); Ry = np.array([[cy[i], 0, sy[i]], [0, 1, 0], [-sy[i], 0, cy[i]]])
        Rz = np.array([[cz[i], -sz[i], 0], [sz[i], cz[i], 0], [0, 0, 1]])
        Rg[i] = (Rz @ Ry @ Rx).astype(np.float32)
    source = (template @ Rg.transpose(0, 2, 1) + rand((B, 1, 3), 2, -0.5, 0.5)).astype(np.float32)
    with torch.no_grad():
        out = net.cuda()(dev(template), dev(source))
    want = {k: [] for k in ("est_R", "est_t", "r", "H")}
    for c in range(0, B, 8):                                 # bound the CPU side's [8,4,1024,1024] attention maps
        o = oracle.dcp_forward_torch(template[c:c + 8], source[c:c + 8], w)
        for k in want:
            want[k].append(o[k])
    want = {k: np.concatenate(v) for k, v in want.items()}
    np.testing.assert_allclose(out["r"].cpu().numpy(), want["r"], rtol=1e-3, atol=2e-5)
    s = np.linalg.svd(want["H"].astype(np.float64), compute_uv=False)
    ok = ((s[:, 1] - s[:, 2]) / s[:, 0] >= 1e-2) & ((s[:, 1] + s[:, 2]) / s[:, 0] >= 1e-2)
    assert ok.sum() >= B // 2, ok.sum()
    eR = np.abs(out["est_R"].cpu().numpy() - want["est_R"])[ok].max()
    et = np.abs(out["est_t"].cpu().numpy() - want["est_t"])[ok].max()
    print(f"config 3 full size: max |dR| {eR:.2e}, max |dt| {et:.2e} over {ok.sum()} well-conditioned clouds")
    assert eR <= 1e-5 and et <= 1e-5, (eR, et)


def test_device_feed_dcp_transform_golden(golden):
    """the batched on-device DCPTransform (l3d_euler_transform) against the reference's own
    (scipy, one cloud at a time) for fixed angles; the drawing wrapper's ranges; the seeded on-device cloud generator."""
    from learning3d_amd.ops.transform_functions import DCPTransform, euler_transform
    from learning3d_amd.data_utils import RegistrationFeed, uniform_clouds
    g = golden("dcp_transform")
    euler = np.stack([g["anglez"], g["angley"], g["anglex"]], axis=1).astype(np.float32)
    src, igt = euler_transform(dev(g["template"]), dev(euler), dev(g["translation"].astype(np.float32)))
    # the fixture's angles are fp64; the device entry point takes fp32 angles: 1e-6 covers their rounding
    np.testing.assert_allclose(src.cpu().numpy(), g["source"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(igt.cpu().numpy(), g["igt"], rtol=0, atol=2e-6)
    # against the oracle at the SAME fp32 angles: one rounding apart at most
    osrc, oigt = oracle.dcp_transform(g["template"], euler[:, 2].astype(np.float64), euler[:, 1].astype(np.float64),
                                      euler[:, 0].astype(np.float64), g["translation"].astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(src.cpu().numpy(), osrc, rtol=0, atol=1.2e-7)
    np.testing.assert_allclose(igt.cpu().numpy(), oigt, rtol=0, atol=6e-8)
    tf = DCPTransform(angle_range=45, translation_range=1, generator=torch.Generator(device="cuda").manual_seed(3))
    t = dev(rand((8, 64, 3), 1, -0.5, 0.5))
    s = tf(t)
    R = tf.igt[:, :3, :3].transpose(1, 2)
    np.testing.assert_allclose((torch.matmul(t, R.transpose(1, 2)) + tf.igt[:, :3, 3].unsqueeze(1)).cpu().numpy(), s.cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(torch.linalg.det(R.double()).cpu().numpy(), 1.0, atol=1e-6)
    assert float(tf.anglex.max()) <= np.pi / 4 and float(tf.anglex.min()) >= 0 and float(tf.translation.abs().max()) <= 1
    a, b = uniform_clouds(4, 1000, -0.5, 0.5, seed=7), uniform_clouds(4, 1000, -0.5, 0.5, seed=7)
    assert torch.equal(a, b) and float(a.min()) >= -0.5 and float(a.max()) < 0.5 and abs(float(a.mean())) < 0.02
    assert not torch.equal(a, uniform_clouds(4, 1000, -0.5, 0.5, seed=8))
    feed = RegistrationFeed(16, 256, seed=5, length=2)
    batches = list(feed)
    assert len(batches) == 2 and batches[0][0].shape == (16, 256, 3) and batches[0][2].shape == (16, 4, 4)
    assert not torch.equal(batches[0][0], batches[1][0])


# ------------------------------------------------------------------------------- training path (8(f) rank 3)
def test_train_conv_bn_relu_matches_torch():
    """_train.conv_bn_act (HIP GEMMs for conv / dgrad / wgrad, HIP BatchNorm statistics and normalisation) against
    torch's Conv2d -> BatchNorm2d (batch statistics) -> ReLU: outputs, the gradients of x, W, gamma, beta, and the
    running-statistics update.  Also the per-cloud fp64 partial sums against numpy."""
    from learning3d_amd.models import _train
    torch.manual_seed(11)
    for (B, Cin, Cout, N, K) in [(4, 6, 64, 128, 20), (3, 64, 128, 96, 20), (2, 512, 256, 256, 1)]:
        conv = torch.nn.Conv2d(Cin, Cout, 1, bias=False).cuda()
        bn = torch.nn.BatchNorm2d(Cout).cuda().train()
        bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.uniform_(-0.3, 0.3)
        conv_r, bn_r = torch.nn.Conv2d(Cin, Cout, 1, bias=False).cuda(), torch.nn.BatchNorm2d(Cout).cuda().train()
        conv_r.load_state_dict(conv.state_dict()); bn_r.load_state_dict(bn.state_dict())
        x = torch.randn(B, Cin, N, K, device="cuda") * 0.7 + 0.1
        go = torch.randn(B, Cout, N, K, device="cuda")
        xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
        ya = _train.conv_bn_act(xa, conv, bn, relu=True, sync=False)
        yb = torch.relu(bn_r(conv_r(xb)))
        np.testing.assert_allclose(ya.detach().cpu().numpy(), yb.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
        ya.backward(go); yb.backward(go)
        for name, a, b in (("x", xa.grad, xb.grad), ("W", conv.weight.grad, conv_r.weight.grad),
                           ("gamma", bn.weight.grad, bn_r.weight.grad), ("beta", bn.bias.grad, bn_r.bias.grad)):
            scale = float(b.abs().max())
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-5 * max(1.0, scale), err_msg=name)
        np.testing.assert_allclose(bn.running_mean.cpu().numpy(), bn_r.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), bn_r.running_var.cpu().numpy(), rtol=1e-5, atol=1e-6)
        assert int(bn.num_batches_tracked) == 1
    z = torch.randn(3, 5, 777, device="cuda")
    part = _train.channel_stats(z).cpu().numpy()
    z64 = z.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(part[..., 0], z64.sum(-1), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(part[..., 1], (z64 ** 2).sum(-1), rtol=1e-12)


# test_dgcnn_training_step_*: tests/test_gpu_grad_routes.py (gradients against an fp64 evaluation on the fp32 run's own ReLU /
# max-pool branches)


def test_curvenet_lpfa_golden(golden):
    """CurveNet's LPFA (utils/curvenet_util.py:229-291): kNN on xyz with add_one_to_k + the one-pass grouping kernel, both
    variants (initial: geometry only; deep: feature differences + xyz2feature), against the reference module on the CPU,
    and the autograd route against the fused one."""
    from learning3d_amd.utils.curvenet_util import LPFA
    g = golden("lpfa")
    xyz, feats = dev(g["xyz"]), dev(g["feats"])
    for name, initial in (("init", True), ("deep", False)):
        m = LPFA(9 if initial else 16, 24, k=12, mlp_num=1 if initial else 2, initial=initial)
        m.load_state_dict({k[len(f"w_{name}."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(f"w_{name}.")})
        m = m.cuda().eval()
        with torch.no_grad():
            out = m(xyz if initial else feats, xyz)
        np.testing.assert_allclose(out.cpu().numpy(), g["out_" + name], rtol=1e-4, atol=1e-5)
        x_in = (xyz if initial else feats).clone().requires_grad_()
        out2 = m(x_in, xyz)
        np.testing.assert_allclose(out2.detach().cpu().numpy(), g["out_" + name], rtol=1e-4, atol=1e-5)
        if not initial:
            out2.sum().backward()
            assert torch.isfinite(x_in.grad).all() and float(x_in.grad.abs().max()) > 0


# ------------------------------------------------ kNN on the matrix cores (knn_mfma.hip) vs the insertion kernel
def _knn_variant(x_bn3, k, variant):
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    x = x_bn3.contiguous()
    B, N, _ = x.shape
    idx = torch.full((B, N, k), -7, dtype=torch.int64, device=x.device)
    check(lib().l3d_knn_graph_variant(ptr(x), B, N, k, ptr(idx), variant, stream_ptr()), "l3d_knn_graph_variant")
    return idx


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,k", [(4, 1024, 20), (2, 1000, 24), (3, 256, 1), (2, 2048, 20), (2, 1500, 7), (1, 257, 16)])
def test_knn_mfma_equals_insertion_kernel(B, N, k):
    """The matrix-core kernel (variant 2) must return the insertion kernel's (variant 1) indices exactly: same ranking
    bits (the MFMA k-slot chain IS the reference's fma chain), same tie rule.  Uniform, spatially sorted and clustered
    clouds; the first case is also pinned to the oracle."""
    g = torch.Generator().manual_seed(N + k)
    uni = torch.rand((B, N, 3), generator=g)
    srt = torch.stack([c[torch.argsort(c[:, 0])] for c in torch.rand((B, N, 3), generator=g)])
    clu = torch.randn((B, N, 3), generator=g) * 0.05 + torch.randn((B, 1, 3), generator=g)
    for name, pts in (("uniform", uni), ("sorted", srt), ("clustered", clu)):
        a = _knn_variant(dev(pts), k, 2).cpu().numpy()
        b = _knn_variant(dev(pts), k, 1).cpu().numpy()
        assert (a >= 0).all() and (a < N).all(), name
        assert np.array_equal(a, b), f"{name}: {np.argwhere(a != b)[:5]}"
    if (B, N, k) == (4, 1024, 20):
        oracle.assert_knn_equal_modulo_ties(_knn_variant(dev(uni), k, 2).cpu().numpy(), oracle.knn(uni.numpy(), k), uni.numpy())


@pytest.mark.gpu
def test_knn_mfma_ties_duplicates_and_list_overflow():
    """Exact ties resolve to the lower index like the insertion kernel.  Clouds with hundreds of copies of one point, a
    dense cluster of near-duplicates, and an all-equal cloud overflow the 64-key candidate lists: the kernel tightens
    its thresholds itself, switches to strict collection at a tie and fills the remaining ranks with the lowest-index
    ties -- there is no second kernel, and no block may be left marked."""
    g = torch.Generator().manual_seed(5)
    N, k = 1024, 20
    pairs = torch.rand((2, N, 3), generator=g)
    pairs[:, N // 2:] = pairs[:, :N // 2]                                   # every point twice: ties everywhere
    heavy = torch.rand((2, N, 3), generator=g)
    heavy[0, 100:420] = heavy[0, 7]                                          # 321 copies of one point
    heavy[1, 600:700] = heavy[1, 3]
    heavy[1, 900:1000] = heavy[1, 3]
    near = torch.rand((2, N, 3), generator=g)
    near[:, 200:700] = near[:, 11:12] + 1e-4 * torch.rand((2, 500, 3), generator=g)      # 500 distinct points in a 1e-4 box
    mixed = torch.rand((1, N, 3), generator=g)
    mixed[0, 300:330] = mixed[0, 5] + 1e-5 * torch.rand((30, 3), generator=g)             # 30 closer than the duplicates ...
    mixed[0, 400:900] = mixed[0, 5] + 0.01                                                # ... 500 copies of a point just behind them
    same = torch.zeros((1, N, 3)) + 0.25                                     # all points identical
    big = torch.rand((1, 2048, 3), generator=g)                              # N = 2048: the variant that recomputes its tiles
    big[0, 50:600] = big[0, 9]
    for name, pts, kk in (("pairs", pairs, k), ("heavy", heavy, k), ("near", near, k), ("mixed", mixed, k), ("same", same, k),
                          ("big", big, k), ("heavy k=24", heavy, 24), ("heavy k=1", heavy, 1)):
        a = _knn_variant(dev(pts), kk, 2).cpu().numpy()
        b = _knn_variant(dev(pts), kk, 1).cpu().numpy()
        assert (a >= 0).all() and (a < pts.shape[1]).all(), f"{name}: an index was left unwritten"
        assert np.array_equal(a, b), f"{name}: {np.argwhere(a != b)[:5]}"
    a = _knn_variant(dev(same), k, 2).cpu().numpy()
    assert np.array_equal(a[0, 0], np.arange(k))                             # all-equal cloud: first k indices


@pytest.mark.gpu
def test_knn_mfma_random_clouds():
    """About one query in 30 000 of a random cloud lets more than 64 candidates through the group-maxima threshold (the
    batch below contains such queries); the kernel settles those itself with a tighter threshold."""
    g = torch.Generator().manual_seed(0)
    for name, pts in (("U(0,1)", torch.rand((32, 1024, 3), generator=g)), ("U(-.5,.5)", torch.rand((32, 1024, 3), generator=g) - 0.5),
                      ("N(0,1)", torch.randn((32, 1024, 3), generator=g)), ("U(-.5,.5) N=2048", torch.rand((8, 2048, 3), generator=g) - 0.5)):
        a = _knn_variant(dev(pts), 20, 2).cpu().numpy()
        assert (a >= 0).all(), name
        assert np.array_equal(a, _knn_variant(dev(pts), 20, 1).cpu().numpy()), name


@pytest.mark.gpu
def test_knn_mfma_fuzz_against_insertion_kernel():
    """60 random shapes and duplicate patterns (N 256..2048, k 1..24; duplicate runs, near-duplicate clumps, repeated
    coordinates on a coarse grid): the two kernels must agree index for index."""
    rng = np.random.default_rng(2024)
    for it in range(60):
        N = int(rng.integers(256, 2049))
        k = int(rng.integers(1, 25))
        B = int(rng.integers(1, 4))
        pts = rng.random((B, N, 3)).astype(np.float32) * float(rng.choice([1.0, 10.0, 0.01])) - float(rng.choice([0.0, 0.5]))
        mode = it % 4
        if mode == 1:                                        # runs of exact duplicates
            for _ in range(int(rng.integers(1, 6))):
                lo = int(rng.integers(0, N - 1)); n = int(rng.integers(2, min(400, N - lo)))
                pts[:, lo:lo + n] = pts[:, int(rng.integers(0, N))][:, None]
        elif mode == 2:                                      # tight clumps of distinct points
            for _ in range(int(rng.integers(1, 4))):
                lo = int(rng.integers(0, N - 1)); n = int(rng.integers(2, min(300, N - lo)))
                pts[:, lo:lo + n] = pts[:, lo:lo + 1] + (1e-5 * rng.random((B, n, 3))).astype(np.float32)
        elif mode == 3:                                      # coarse grid: many exactly equal distances
            pts = np.round(pts * 4) / 4
        a = _knn_variant(dev(pts), k, 2).cpu().numpy()
        b = _knn_variant(dev(pts), k, 1).cpu().numpy()
        assert np.array_equal(a, b), f"case {it} (N={N}, k={k}, mode {mode}): {np.argwhere(a != b)[:5]}"


@pytest.mark.gpu
def test_knn_variant_argument_checks():
    from learning3d_amd._lib import L3DError
    x = dev(rand((1, 200, 3), 1))
    with pytest.raises(L3DError):
        _knn_variant(x, 7, 2)                                                # N < 256: the matrix-core kernel is not built for it
    with pytest.raises(L3DError):
        _knn_variant(dev(rand((1, 512, 3), 1)), 32, 2)                       # k > 24
    assert np.array_equal(_knn_variant(x, 7, 0).cpu().numpy(), _knn_variant(x, 7, 1).cpu().numpy())


@pytest.mark.gpu
def test_device_pose_transforms_golden(golden):
    """l3d_twist_transform / l3d_quat_transform and their class mirrors against the reference's PNLKTransform /
    RPMNetTransform / PCRNetTransform outputs (golden generated by running the reference)."""
    from learning3d_amd.ops import transform_functions as T
    g = golden("pose_transforms")
    src, igt, gt = T.twist_transform(dev(g["template"]), dev(g["twist"]))
    np.testing.assert_allclose(src.cpu().numpy(), g["source"], atol=2e-6)
    np.testing.assert_allclose(igt.cpu().numpy(), g["igt"], atol=1e-6)
    np.testing.assert_allclose(gt.cpu().numpy(), g["gt"], atol=1e-6)
    tf = T.RPMNetTransform(mag=1)
    p6 = tf.apply_transform(dev(np.concatenate([g["template"], g["normals"]], axis=2)), dev(g["twist"]))
    np.testing.assert_allclose(p6.cpu().numpy(), g["source6"], atol=2e-6)
    np.testing.assert_allclose(tf.igt.cpu().numpy(), g["igt"], atol=1e-6)
    np.testing.assert_allclose(T.quat_transform(dev(g["template"]), dev(g["pose7"])).cpu().numpy(), g["pcr_source"], atol=1e-6)
    # the drawing side: unit twists of the requested magnitude, proper rotations, gt = igt^-1
    gen = torch.Generator(device="cuda").manual_seed(3)
    t1 = T.PNLKTransform(mag=0.8, generator=gen)
    x = t1.generate_transform(16, "cuda")
    np.testing.assert_allclose(x.norm(dim=1).cpu().numpy(), 0.8, rtol=1e-5)
    out = t1(dev(g["template"]))
    assert out.shape == g["template"].shape
    eye = torch.matmul(t1.gt, t1.igt).cpu().numpy()
    np.testing.assert_allclose(eye, np.broadcast_to(np.eye(4), eye.shape), atol=1e-5)
    t3 = T.PCRNetTransform(angle_range=45, translation_range=1, generator=gen)
    s3 = t3(dev(g["template"]))
    assert t3.igt.shape == (g["template"].shape[0], 7) and torch.isfinite(s3).all()
    np.testing.assert_allclose(t3.igt[:, :4].norm(dim=1).cpu().numpy(), 1.0, rtol=1e-5)
    np.testing.assert_allclose(s3.cpu().numpy(), oracle.quat_transform(g["template"], t3.igt.cpu().numpy()), atol=1e-6)


@pytest.mark.gpu
def test_resident_registration_feed():
    """A dataset resident in HBM served as registration batches: every cloud once per epoch, same batches for the same
    (seed, epoch), template rows are dataset rows, and (source, igt) are consistent for each device transform."""
    from learning3d_amd.data_utils.device_feed import ResidentRegistrationFeed
    from learning3d_amd.ops import transform_functions as T
    g = torch.Generator().manual_seed(0)
    data = torch.rand((70, 300, 3), generator=g).cuda() - 0.5
    labels = torch.arange(70).cuda()
    feed = ResidentRegistrationFeed(data, labels, batch_size=16, num_points=128, seed=5)
    batches = list(feed)
    assert len(batches) == len(feed) == 4
    seen = torch.cat([b[3] for b in batches])
    assert seen.unique().numel() == 64                                      # a permutation: no cloud twice
    for tmpl, src, igt, lab in batches:
        assert torch.equal(tmpl, data[lab, :128])
        R, t = igt[:, :3, :3].transpose(1, 2), igt[:, :3, 3]                # DCP's igt holds R^T | t
        np.testing.assert_allclose(src.cpu().numpy(), (torch.matmul(tmpl, R.transpose(1, 2)) + t[:, None]).cpu().numpy(), atol=1e-5)
    again = list(ResidentRegistrationFeed(data, labels, batch_size=16, num_points=128, seed=5))
    assert all(torch.equal(a[1], b[1]) for a, b in zip(batches, again))
    # a twist transform instead, with the point shuffle
    feed2 = ResidentRegistrationFeed(data, None, batch_size=10, num_points=64, transform=T.PNLKTransform(mag=0.5), randomize_points=True, seed=1)
    tmpl, src, igt, lab = next(iter(feed2))
    assert lab is None and tmpl.shape == (10, 64, 3)
    want = torch.matmul(tmpl, igt[:, :3, :3].transpose(1, 2)) + igt[:, None, :3, 3]
    np.testing.assert_allclose(src.cpu().numpy(), want.cpu().numpy(), atol=1e-5)


@pytest.mark.gpu
def test_layernorm_planes_matches_layernorm_and_feeds_conv_f16():
    """l3d_layernorm_planes: y equals the values-only call (img NULL) bit for bit, and its fp16 plane image drives l3d_pointwise_conv_f16
    to the same result as the f16x2 conv on a split of y (the image is y 2^T with T from the layer's parameters)."""
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    from learning3d_amd.models import _fused
    rng = np.random.default_rng(17)
    for (B, N, C, amp) in [(2, 256, 512, 1.0), (1, 512, 128, 40.0), (3, 256, 64, 1e-3)]:
        x = dev((rng.standard_normal((B, N, C)) * amp + amp).astype(np.float32))
        a = dev((rng.standard_normal(C) * 0.5 + 1).astype(np.float32))
        b = dev((rng.standard_normal(C) * 0.2).astype(np.float32))
        y0 = torch.empty_like(x)
        check(lib().l3d_layernorm_planes(ptr(x), ptr(a), ptr(b), 1e-6, B * N, C, ptr(y0), None, stream_ptr()), "ln")
        y1 = torch.empty_like(x)
        img = torch.empty(lib().l3d_f16_image_bytes(1, B * N, C), dtype=torch.uint8, device="cuda")
        check(lib().l3d_layernorm_planes(ptr(x), ptr(a), ptr(b), 1e-6, B * N, C, ptr(y1), ptr(img), stream_ptr()), "lnp")
        assert torch.equal(y0, y1)
        w = dev((rng.standard_normal((256, C)) / C ** 0.5).astype(np.float32))
        wimg = _fused.split_weights_f16(w)
        got = _fused.pointwise_conv_f16(img, B, N, wimg, C, 256).cpu().numpy()
        want = np.einsum("oc,bnc->bon", w.cpu().numpy().astype(np.float64), y0.cpu().numpy().astype(np.float64))
        ref = _fused.pointwise_conv_f16(_fused.split_rows_f16(y0), B, N, wimg, C, 256).cpu().numpy()
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 2.0 * np.abs(ref - want).max() + 1e-6 * scale
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-6 * scale)
    _fused.check_range(sync=True)


@pytest.mark.gpu
def test_attention_f16_context_planes_feed_conv_f16():
    """l3d_attention_forward_f16b can hand its context to the output projection as an fp16 plane image (scale from max|v|):
    the f16x2 conv on that image must equal the projection of the fp32 context to fp32-level accuracy."""
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    from learning3d_amd.models import _fused
    rng = np.random.default_rng(23)
    for (B, H, D, N, M, vs) in [(2, 4, 128, 256, 256, 1.0), (1, 4, 64, 512, 300, 250.0), (2, 8, 32, 256, 256, 1e-3)]:
        C = H * D
        q, k = (dev(rng.standard_normal((B, C, n)).astype(np.float32)) for n in (N, M))
        v = dev((rng.standard_normal((B, C, M)) * vs).astype(np.float32))
        ws = torch.zeros(4, dtype=torch.int32, device="cuda")
        ctx = torch.empty((B, C, N), dtype=torch.float32, device="cuda")
        img = torch.empty(lib().l3d_f16_image_bytes(1, B * N, C), dtype=torch.uint8, device="cuda")
        check(lib().l3d_attention_forward_f16b(ptr(q), ptr(k), ptr(v), B, H, D, N, M, C * N, C * M, C * M, float(1 / np.sqrt(D)),
                                               ptr(ws), 0, ptr(ctx), ptr(img), stream_ptr()), "att")
        w = dev((rng.standard_normal((256, C)) / C ** 0.5).astype(np.float32))
        got = _fused.pointwise_conv_f16(img, B, N, _fused.split_weights_f16(w), C, 256).cpu().numpy()
        want = np.einsum("oc,bcn->bon", w.cpu().numpy().astype(np.float64), ctx.cpu().numpy().astype(np.float64))
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=3e-6 * np.abs(want).max())
    _fused.check_range(sync=True)


@pytest.mark.gpu
def test_conv_f16_plane_output_chains():
    """Two f16x2 layers chained through an fp16 plane image (l3d_pointwise_conv_f16 with out_img -> l3d_pointwise_conv_f16): the
    hidden layer is never written in fp32; its plane scale comes from a bound the kernel derives from the weights' row
    sums and the input image's scale.  Against fp64, with hidden activations from tiny to large."""
    from learning3d_amd.models import _fused
    rng = np.random.default_rng(29)
    B, N, C0, C1, C2 = 2, 512, 256, 512, 256
    for xs, w1s, b1s in ((1.0, 1.0, 0.1), (1e-3, 1.0, 1e-4), (20.0, 5.0, 50.0)):
        x = (rng.standard_normal((B, N, C0)) * xs).astype(np.float32)
        w1 = (rng.standard_normal((C1, C0)) * w1s / C0 ** 0.5).astype(np.float32)
        b1 = (rng.standard_normal(C1) * b1s).astype(np.float32)
        w2 = (rng.standard_normal((C2, C1)) / C1 ** 0.5).astype(np.float32)
        b2 = rng.standard_normal(C2).astype(np.float32)
        h = np.maximum(x.astype(np.float64) @ w1.astype(np.float64).T + b1, 0)
        want = (h @ w2.astype(np.float64).T + b2).transpose(0, 2, 1)
        ximg = _fused.split_rows_f16(dev(x))
        himg = _fused.pointwise_conv_f16(ximg, B, N, _fused.split_weights_f16(dev(w1)), C0, C1, None, dev(b1), relu=True, out_planes=True)
        got = _fused.pointwise_conv_f16(himg, B, N, _fused.split_weights_f16(dev(w2)), C1, C2, None, dev(b2)).cpu().numpy()
        hf = _fused.pointwise_conv_f16(ximg, B, N, _fused.split_weights_f16(dev(w1)), C0, C1, None, dev(b1), relu=True)       # fp32 hidden
        ref = _fused.pointwise_conv_f16(_fused.split_rows_f16(hf, channel_first=True), B, N, _fused.split_weights_f16(dev(w2)), C1, C2, None, dev(b2)).cpu().numpy()
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 2.0 * np.abs(ref - want).max() + 2e-6 * scale, (xs, np.abs(got - want).max(), np.abs(ref - want).max())
    _fused.check_range(sync=True)


def test_conv_f16_pool_epilogue_and_first_layer():
    """conv_f16.hip's pooled epilogue and the first-layer plane writer, the pieces of PCN's f16x2 encoder chain (pcn.py:110-124):
    (1) l3d_first_layer_f16_planes (3 -> 128, ReLU) feeding a 128 -> 256 layer, against fp64, both input layouts;
    (2) ypool == the maximum of the same layer's fp32 output, alone and together with a plane image; (3) a per-cloud shift in
    plane mode (conv3 with the pooled half of W3 folded into the shift).  Activations from tiny to large."""
    from learning3d_amd.models import _fused
    rng = np.random.default_rng(31)
    B, N = 3, 512
    for xs, bs in ((1.0, 0.1), (1e-3, 1e-4), (40.0, 30.0)):
        x = (rng.uniform(-1, 1, (B, N, 3)) * xs).astype(np.float32)
        w1 = rng.standard_normal((128, 3)).astype(np.float32)
        b1 = (rng.standard_normal(128) * bs).astype(np.float32)
        w2 = (rng.standard_normal((256, 128)) / 128 ** 0.5).astype(np.float32)
        b2 = (rng.standard_normal(256) * bs).astype(np.float32)
        w3 = (rng.standard_normal((256, 256)) / 16).astype(np.float32)
        s3 = (rng.standard_normal((B, 256)) * bs * 3).astype(np.float32)           # per-cloud shift
        h1 = np.maximum(x.astype(np.float64) @ w1.astype(np.float64).T + b1, 0)
        h2 = h1 @ w2.astype(np.float64).T + b2                                     # [B,N,256]
        h3 = np.maximum(h2 @ w3.astype(np.float64).T + s3[:, None, :], 0)
        w2i, w3i = _fused.split_weights_f16(dev(w2)), _fused.split_weights_f16(dev(w3))
        for cl in (True, False):
            xin = dev(x) if cl else dev(np.ascontiguousarray(x.transpose(0, 2, 1)))
            img1 = _fused.first_layer_f16_planes(xin, dev(w1), dev(b1), True, cl)
            y2 = _fused.pointwise_conv_f16(img1, B, N, w2i, 128, 256, None, dev(b2))                       # fp32 [B,256,N]
            tol = 1e-5 * np.abs(h2).max()
            assert np.abs(y2.cpu().numpy() - h2.transpose(0, 2, 1)).max() <= tol, (xs, cl)
        img2, g = _fused.pointwise_conv_f16_pool(img1, B, N, w2i, 128, 256, None, dev(b2), relu=False, out_planes=True)
        assert torch.equal(g, y2.max(dim=2)[0])
        _, g_only = _fused.pointwise_conv_f16_pool(img1, B, N, w2i, 128, 256, None, dev(b2), relu=False)
        assert torch.equal(g_only, g)
        y3 = _fused.pointwise_conv_f16(img2, B, N, w3i, 256, 256, None, dev(s3), relu=True)                # per-cloud shift, fp32 out
        assert np.abs(y3.cpu().numpy() - h3.transpose(0, 2, 1)).max() <= 1e-5 * np.abs(h3).max() + 1e-5 * np.abs(h2).max()
        img3, g3 = _fused.pointwise_conv_f16_pool(img2, B, N, w3i, 256, 256, None, dev(s3), relu=True, out_planes=True)
        assert torch.equal(g3, y3.max(dim=2)[0])
        ident = np.eye(256, dtype=np.float32)
        back = _fused.pointwise_conv_f16(img3, B, N, _fused.split_weights_f16(dev(ident)), 256, 256).cpu().numpy()   # read the image back
        assert np.abs(back - y3.cpu().numpy()).max() <= 1e-6 * np.abs(h3).max()
    _fused.check_range(sync=True)


def test_pcn_encoder_f16_chain_matches_reference_order_path():
    """PCN with N % 256 == 0 takes the f16x2 encoder chain (models/pcn.py::_encode_f16); against the reference-order torch
    route (autograd path) and an fp64 evaluation of pcn.py:110-124, both input layouts."""
    from learning3d_amd.models import PCN, _fused
    assert _fused.gemm_arith() == "f16x2"
    for shape in ("bnc", "bcn"):
        torch.manual_seed(5)
        net = PCN(emb_dims=1024, input_shape=shape, num_coarse=64, grid_size=2, detailed_output=True).cuda().eval()
        x = dev(rand((3, 512, 3), 6, -0.5, 0.5))
        if shape == "bcn":
            x = x.permute(0, 2, 1).contiguous()
        with torch.no_grad():
            fused = net(x)
            gf = net.global_feature_v.clone()
        assert "conv4" in net.__dict__.get("_w_f16_cache", {}), "the f16x2 chain did not run"
        with _fused.per_layer_route():
            ref = net(x.clone().requires_grad_())
        gr = net.global_feature_v.detach()
        xd = (x if shape == "bcn" else x.permute(0, 2, 1)).double().cpu()
        p = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
        conv = lambda n, t: torch.einsum("oc,bcn->bon", p[n + ".weight"][:, :, 0], t) + p[n + ".bias"][None, :, None]
        h = conv("conv2", torch.relu(conv("conv1", xd)))
        h = torch.cat([h, h.max(dim=2, keepdim=True)[0].expand(-1, -1, h.shape[2])], dim=1)
        want = conv("conv4", torch.relu(conv("conv3", h))).max(dim=2)[0].numpy()
        e_f16, e_ref = np.abs(gf.cpu().numpy() - want).max(), np.abs(gr.cpu().numpy() - want).max()
        assert e_f16 <= max(2 * e_ref, 1e-5 * np.abs(want).max()), (e_f16, e_ref)      # (torch's own error moves with the solver it picks)
        for k in ("coarse_output", "fine_output"):
            np.testing.assert_allclose(fused[k].cpu().numpy(), ref[k].detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    _fused.check_range(sync=True)


def test_conv_f16_absmax_feeds_attention_maxima():
    """The fused q|k|v projection reports max|q|, |k|, |v| from its own epilogue (l3d_pointwise_conv_f16 with amax_out) and
    l3d_attention_forward_f16b (maxima_ready = 1) takes them instead of a pass over the three tensors: the maxima equal torch's,
    and the attention output is bit-identical to the call that measures them itself (utils/transformer.py:183-189)."""
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    from learning3d_amd.models import _fused
    rng = np.random.default_rng(41)
    B, N, C, H = 2, 512, 512, 4
    x = rng.standard_normal((B, N, C)).astype(np.float32)
    w = (rng.standard_normal((3 * C, C)) / C ** 0.5 * np.repeat([1.0, 3.0, 0.2], C)[:, None]).astype(np.float32)
    b = rng.standard_normal(3 * C).astype(np.float32) * 0.1
    ws = torch.zeros(4, dtype=torch.int32, device="cuda")
    qkv = _fused.pointwise_conv_f16(_fused.split_rows_f16(dev(x)), B, N, _fused.split_weights_f16(dev(w)), C, 3 * C, None, dev(b), amax=(ws, C))
    plain = _fused.pointwise_conv_f16(_fused.split_rows_f16(dev(x)), B, N, _fused.split_weights_f16(dev(w)), C, 3 * C, None, dev(b))
    assert torch.equal(qkv, plain)
    got = ws[:3].view(torch.float32).cpu().numpy()
    want = np.array([float(qkv[:, i * C:(i + 1) * C].abs().max()) for i in range(3)], dtype=np.float32)
    assert np.array_equal(got, want), (got, want)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    outs = []
    for ready, wsp in ((1, ws), (0, torch.zeros(4, dtype=torch.int32, device="cuda"))):
        ctx = torch.empty((B, C, N), dtype=torch.float32, device="cuda")
        check(lib().l3d_attention_forward_f16b(ptr(q), ptr(k), ptr(v), B, H, C // H, N, N, q.stride(0), k.stride(0), v.stride(0),
                                               1.0 / (C // H) ** 0.5, ptr(wsp), ready, ptr(ctx), None, stream_ptr()), "attention")
        outs.append(ctx)
    assert torch.equal(outs[0], outs[1])
    _fused.check_range(sync=True)


def test_flownet3d_factored_first_layer_matches_grouped_route():
    """FlowEmbedding / PointNetSetUpConv (reference models/flownet3d.py:125-180, :182-242): the first grouped layer as
    per-point products + l3d_group_first_layer against (a) the grouped-tensor route of this package and (b) the reference-order
    torch ops (autograd route), random BN statistics, kNN and ball-query grouping."""
    import learning3d_amd.models.flownet3d as F3
    rng = np.random.default_rng(53)
    B, N1, N2, C = 2, 256, 128, 64
    pos1 = dev(rng.uniform(-1, 1, (B, 3, N1)).astype(np.float32))
    pos2 = dev(rng.uniform(-1, 1, (B, 3, N2)).astype(np.float32))
    f1 = dev(rng.standard_normal((B, C, N1)).astype(np.float32))
    f2 = dev(rng.standard_normal((B, C, N2)).astype(np.float32))
    torch.manual_seed(9)
    mods = [F3.FlowEmbedding(radius=10.0, nsample=16, in_channel=C, mlp=[128, 128, 128], pooling='max', corr_func='concat'),
            F3.PointNetSetUpConv(nsample=8, radius=2.4, f1_channel=C, f2_channel=C, mlp=[128, 64, 256], mlp2=[128]),
            F3.PointNetSetUpConv(nsample=8, radius=0.9, f1_channel=C, f2_channel=C, mlp=[64], mlp2=[], knn=False),
            F3.PointNetSetUpConv(nsample=8, radius=2.4, f1_channel=C, f2_channel=C, mlp=[128, 128, 256], mlp2=[128])]
    for m in mods:
        m.cuda().eval()
        for sub in m.modules():
            if isinstance(sub, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                sub.running_mean.uniform_(-0.2, 0.2); sub.running_var.uniform_(0.5, 1.5)
                sub.weight.data.uniform_(0.5, 1.5); sub.bias.data.uniform_(-0.3, 0.3)
        outs = {}
        for flag in ((True, True), (True, False), (False, False)):               # (factored first layer, f16x2 chain behind it)
            F3.FACTOR_FIRST_LAYER, F3.F16_GROUPED_STACK = flag
            try:
                with torch.no_grad():
                    r = m(pos1, pos2, f1, f2)
            finally:
                F3.FACTOR_FIRST_LAYER, F3.F16_GROUPED_STACK = True, False
            outs[flag] = (r[1] if isinstance(r, tuple) else r)
        ref = m(pos1, pos2, f1.clone().requires_grad_(), f2)                      # torch conv / BN ops in the reference's order
        ref = (ref[1] if isinstance(ref, tuple) else ref).detach()
        scale = float(ref.abs().max())
        for flag, out in outs.items():
            assert out.shape == ref.shape
            assert float((out - ref).abs().max()) <= 3e-5 * scale + 1e-6, (type(m).__name__, flag, float((out - ref).abs().max()), scale)
    from learning3d_amd.models import _fused
    _fused.check_range(sync=True)


def test_layernorm_deferred_values_materialise_on_demand():
    """utils/transformer.py: inside SublayerConnection the norm output of a sublayer marked _planes_ok is written as the fp16
    plane image only (reference :82-88, :109-119); whoever reads its fp32 values gets them through _ln_values.  Unmarked
    sublayers (user callables) always see real values; a marked sublayer that falls to its slow route (mask given) must too."""
    from learning3d_amd.utils import transformer as T
    torch.manual_seed(21)
    ln = T.LayerNorm(512).cuda()
    ln.a_2.data.uniform_(0.5, 1.5); ln.b_2.data.uniform_(-0.5, 0.5)
    x = dev(rand((2, 256, 512), 22, -2, 2))
    with torch.no_grad():
        full = ln(x)
        lazy = ln(x, values=False)
        assert getattr(lazy, "_l3d_pending", None) is not None and getattr(full, "_l3d_pending", None) is None
        assert torch.equal(lazy._l3d_planes[:-12], full._l3d_planes[:-12])          # planes + 2^-T (the last 12 bytes are scratch)
        assert T._ln_values(lazy) is lazy and lazy._l3d_pending is None
        np.testing.assert_allclose(lazy.cpu().numpy(), full.cpu().numpy(), rtol=1e-6, atol=1e-6)
        sc = T.SublayerConnection(512).cuda().eval()
        sc.norm.a_2.data.copy_(ln.a_2.data); sc.norm.b_2.data.copy_(ln.b_2.data)
        out = sc(x, lambda y: y * 2.0)                                        # unmarked callable
        np.testing.assert_allclose(out.cpu().numpy(), (x + 2.0 * full).cpu().numpy(), rtol=1e-6, atol=1e-5)
        mha = T.MultiHeadedAttention(4, 512).cuda().eval()
        mask = torch.ones((1, 256, 256), device="cuda")
        got = sc(x, T._planes_ok(lambda y: mha(y, y, y, mask)))               # marked, but the mask forces the slow route
        want = x + mha(full, full, full, mask)
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)
        ffn = T.PositionwiseFeedForward(512, 1000).cuda().eval()              # d_ff not a multiple of 256: bf16x3 / torch route
        got = sc(x, ffn)
        want = x + ffn(full)
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_conv_f16_narrow_tile_and_group_maxima():
    """conv_f16.hip's 128 x 512 tile (Cout % 128 == 0, not % 256) and the grouped-maximum epilogue (max over every K = 8, 16, 32,
    64, 128 consecutive points): against fp64 and against the maximum of the kernel's own fp32 output, both tiles, plane output
    chained into a second narrow layer (FlowNet3D's 128 -> 128 -> 128 stacks, models/flownet3d.py:163-179)."""
    from learning3d_amd.models import _fused
    rng = np.random.default_rng(61)
    B, N, C0 = 2, 1024, 128
    x = rng.standard_normal((B, N, C0)).astype(np.float32)
    ximg = _fused.split_rows_f16(dev(x))
    for C1 in (128, 384, 256):                                  # narrow, narrow (3 tiles), wide
        w = (rng.standard_normal((C1, C0)) / C0 ** 0.5).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, C1).astype(np.float32)
        sh = (rng.standard_normal(C1) * 0.3).astype(np.float32)
        want = np.maximum((x.astype(np.float64) @ w.astype(np.float64).T) * sc + sh, 0).transpose(0, 2, 1)       # [B,C1,N]
        wimg = _fused.split_weights_f16(dev(w))
        y = _fused.pointwise_conv_f16(ximg, B, N, wimg, C0, C1, dev(sc), dev(sh), relu=True)
        assert np.abs(y.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max(), C1
        for K in (8, 16, 32, 64, 128):
            _, part = _fused.pointwise_conv_f16_pool(ximg, B, N, wimg, C0, C1, dev(sc), dev(sh), relu=True, group=K)
            assert part.shape == (B, C1, N // K)
            assert torch.equal(part, y.view(B, C1, N // K, K).max(dim=3)[0]), (C1, K)
        himg, g = _fused.pointwise_conv_f16_pool(ximg, B, N, wimg, C0, C1, dev(sc), dev(sh), relu=True, out_planes=True, group=16)
        assert torch.equal(g, y.view(B, C1, N // 16, 16).max(dim=3)[0])
        w2 = (rng.standard_normal((128, C1)) / C1 ** 0.5).astype(np.float32)
        want2 = (want.transpose(0, 2, 1) @ w2.astype(np.float64).T).transpose(0, 2, 1)
        y2 = _fused.pointwise_conv_f16(himg, B, N, _fused.split_weights_f16(dev(w2)), C1, 128)
        assert np.abs(y2.cpu().numpy() - want2).max() <= 1e-5 * np.abs(want2).max() + 1e-5 * np.abs(want).max(), C1
    _fused.check_range(sync=True)


def test_group_first_layer_planes_equals_fp32_rows():
    """l3d_group_first_layer_planes_auto == l3d_group_first_layer read back through an identity f16x2 layer (fp32-level), for
    C1 = 64 / 128 / 256, with and without the per-centre term and a ragged row count; the bound comes from the block maxima
    l3d_absmax4_partials writes, as in the product's call."""
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    from learning3d_amd.models import _fused
    rng = np.random.default_rng(67)
    B, N, S, K = 2, 200, 64, 8                                   # S*K = 512 rows per cloud
    xyz = dev(rng.uniform(-1, 1, (B, N, 3)).astype(np.float32))
    cen = dev(rng.uniform(-1, 1, (B, S, 3)).astype(np.float32))
    idx = dev(rng.integers(0, N, (B, S, K)).astype(np.int32))
    for C1, with_v in ((64, True), (128, False), (256, True)):
        U = dev(rng.standard_normal((B, N, C1)).astype(np.float32))
        V = dev(rng.standard_normal((B, S, C1)).astype(np.float32)) if with_v else None
        sh = None if with_v else dev(rng.standard_normal(C1).astype(np.float32))
        wx = dev(rng.standard_normal((C1, 3)).astype(np.float32))
        rows = torch.empty((B, S * K, C1), dtype=torch.float32, device="cuda")
        check(lib().l3d_group_first_layer(ptr(U), ptr(V), ptr(sh), ptr(wx), ptr(xyz), ptr(cen), ptr(idx), B, N, S, K, C1, 1, ptr(rows),
                                          stream_ptr()), "l3d_group_first_layer")
        g = torch.gather(U, 1, idx.view(B, -1, 1).long().expand(-1, -1, C1)).view(B, S, K, C1)
        d = torch.gather(xyz, 1, idx.view(B, -1, 1).long().expand(-1, -1, 3)).view(B, S, K, 3) - cen[:, :, None, :]
        want = g + (V[:, :, None, :] if with_v else sh) + d @ wx.t()
        np.testing.assert_allclose(rows.view(B, S, K, C1).cpu().numpy(), torch.relu(want).cpu().numpy(), rtol=1e-5, atol=1e-5)
        bound = (U.abs().max() + (V.abs().max() if with_v else sh.abs().max()) + wx.abs().sum(1).max() * (xyz.abs().max() + cen.abs().max())).reshape(1)
        img = torch.empty(lib().l3d_f16_image_bytes(1, B * S * K, C1), dtype=torch.uint8, device="cuda")
        part = torch.empty(256, dtype=torch.float32, device="cuda")
        check(lib().l3d_absmax4_partials(ptr(U), U.numel(), ptr(V), V.numel() if with_v else 0, ptr(xyz), xyz.numel(), ptr(cen), cen.numel(),
                                         ptr(part), stream_ptr()), "absmax4")
        check(lib().l3d_group_first_layer_planes_auto(ptr(U), ptr(V), ptr(sh), ptr(wx), ptr(xyz), ptr(cen), ptr(idx), B, N, S, K, C1, 1,
                                                      ptr(part), float(wx.abs().sum(1).max()), 0.0 if with_v else float(sh.abs().max()),
                                                      ptr(img), ptr(_fused.range_flag(U.device)), stream_ptr()), "planes")
        ident = _fused.split_weights_f16(torch.eye(C1, device="cuda"))
        back = _fused.pointwise_conv_f16(img, B, S * K, ident, C1, C1) if C1 != 64 else None
        if back is not None:                                    # (64 output channels are not a conv_f16 tile: compare 128 / 256)
            assert (back.transpose(1, 2) - rows).abs().max().item() <= 2e-6 * float(bound)
    _fused.check_range(sync=True)


def test_three_interpolate_lds_staged_kernel():
    """three_interpolate at shapes that take the LDS-staged kernel (n >= 4 m, channel rows in LDS; FlowNet3D's feature
    propagation, models/flownet3d.py:268): against the definition (w0 p[i0] + w1 p[i1]) + w2 p[i2] evaluated in fp32 with the same
    operation order, ragged channel count, and through the concat entry point with skip channels."""
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    from learning3d_amd.utils import pointnet2_utils as P
    rng = np.random.default_rng(71)
    for B, c, m, n, c1 in ((2, 40, 256, 2048, 0), (2, 24, 1024, 4096, 5), (1, 7, 3000, 12288, 3)):
        pts = dev(rng.standard_normal((B, c, m)).astype(np.float32))
        idx = dev(rng.integers(0, m, (B, n, 3)).astype(np.int32))
        w = rng.uniform(0.05, 1, (B, n, 3)).astype(np.float32)
        w = dev(w / w.sum(-1, keepdims=True))
        g = [torch.gather(pts, 2, idx[:, :, j].long().unsqueeze(1).expand(-1, c, -1)) for j in range(3)]
        want = (w[:, :, 0].unsqueeze(1) * g[0] + w[:, :, 1].unsqueeze(1) * g[1]) + w[:, :, 2].unsqueeze(1) * g[2]
        got = P.three_interpolate(pts, idx, w)
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-7)
        if c1:
            skip = dev(rng.standard_normal((B, c1, n)).astype(np.float32))
            out = torch.empty((B, c + c1, n), dtype=torch.float32, device="cuda")
            check(lib().l3d_three_interpolate_concat(B, c, m, n, ptr(pts), ptr(idx), ptr(w), ptr(skip), c1, ptr(out), stream_ptr()), "concat")
            assert torch.equal(out[:, :c], got) and torch.equal(out[:, c:], skip)


def test_classifier_pools_inside_the_feature_models_last_conv():
    """models/classifier.py:23 max-pools the feature model's [B,emb,N] output; at inference PointNet / DGCNN hand the Classifier
    the maximum taken in conv5's epilogue (forward_pooled) -- the same values, the feature map never written."""
    from learning3d_amd.models import Classifier, DGCNN, PointNet, Pooling
    torch.manual_seed(13)
    x = dev(rand((4, 1024, 3), 14, -1, 1))
    for fm in (PointNet(emb_dims=1024, use_bn=True), PointNet(emb_dims=512), DGCNN(emb_dims=1024)):
        model = Classifier(feature_model=fm).cuda().eval()
        with torch.no_grad():
            pooled = fm.forward_pooled(x)
            assert pooled is not None and pooled.shape == (4, fm.emb_dims)
            want = Pooling('max')(fm(x))
            if isinstance(fm, DGCNN):
                # forward(): two-plane conv5 on an image with an unscaled residual; forward_pooled(): the pooled epilogue of the
                # three-plane kernel on the scaled-residual image -- the same values to fp32 rounding, not the same bits
                np.testing.assert_allclose(pooled.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-6)
                want = pooled
            assert torch.equal(pooled, want), type(fm).__name__
            logits = model(x)
            ref = model.linear3(torch.relu(model.bn2(model.linear2(torch.relu(model.bn1(model.linear1(want)))))))
            assert torch.equal(logits, ref)
        # grad mode on (the reference's scripts never use no_grad): the same fused route, and it is differentiable
        xg = x.clone().requires_grad_()
        pooled_g = fm.forward_pooled(xg)
        assert pooled_g is not None and torch.equal(pooled_g.detach(), pooled)
        pooled_g.sum().backward()
        assert torch.isfinite(xg.grad).all() and float(xg.grad.abs().max()) > 0


# ------------------------------------------- kNN for large k (knn_select.hip: a wave per query) vs the lane-per-query kernel
def _knn_pair_variant(q, c, k, variant):
    from learning3d_amd._lib import check, lib, ptr, stream_ptr
    q, c = q.contiguous(), c.contiguous()
    B, n, _ = q.shape
    m = c.shape[1]
    d2 = torch.full((B, n, k), -1.0, device=q.device)
    idx = torch.full((B, n, k), -7, dtype=torch.int32, device=q.device)
    check(lib().l3d_knn_variant(B, n, m, k, ptr(q), ptr(c), ptr(d2), ptr(idx), variant, stream_ptr()), "l3d_knn_variant")
    return d2, idx


@pytest.mark.gpu
@pytest.mark.parametrize("B,n,m,k", [(2, 300, 1000, 64), (1, 70, 8192, 64), (2, 128, 256, 64), (1, 40, 5000, 200),
                                     (1, 50, 100, 33), (1, 33, 4097, 100)])
def test_knn_select_kernel_equals_lane_kernel(B, n, m, k):
    """The wave-per-query selection kernel (variant 2) returns the lane-per-query kernel's (variant 1) squared distances
    and indices bit for bit, in the reference's order (ascending, lowest index first on ties): uniform clouds, clouds
    sorted along x, clouds on a coarse grid (most distances tie — the exact two-bisection path), one repeated point."""
    g = torch.Generator().manual_seed(m + k)
    uni = torch.rand((B, m, 3), generator=g)
    srt = torch.stack([c[torch.argsort(c[:, 0])] for c in torch.rand((B, m, 3), generator=g)])
    grid = torch.round(torch.rand((B, m, 3), generator=g) * 3) / 3
    same = torch.full((B, m, 3), 0.25)
    for name, cl in (("uniform", uni), ("sorted", srt), ("grid", grid), ("same", same)):
        cl = cl.cuda()
        q = (torch.rand((B, n, 3), generator=g) if name != "grid" else torch.round(torch.rand((B, n, 3), generator=g) * 3) / 3).cuda()
        d1, i1 = _knn_pair_variant(q, cl, k, 1)
        d2, i2 = _knn_pair_variant(q, cl, k, 2)
        assert torch.equal(i1, i2), name
        assert torch.equal(d1, d2), name
        d0, i0 = _knn_pair_variant(q, cl, k, 0)                # the automatic choice is the selection kernel here
        assert torch.equal(i0, i2) and torch.equal(d0, d2), name
    # against the definition on one cloud: sorted squared distances with a stable argsort
    cl, q = uni[:1].cuda(), torch.rand((1, n, 3), generator=g).cuda()
    d2, i2 = _knn_pair_variant(q, cl, k, 2)
    dx = q[:, :, None, :] - cl[:, None, :, :]
    full = (dx[..., 0] * dx[..., 0] + dx[..., 1] * dx[..., 1]) + dx[..., 2] * dx[..., 2]
    order = torch.argsort(full, dim=-1, stable=True)[..., :k]
    assert torch.equal(i2.long(), order)


@pytest.mark.gpu
@pytest.mark.parametrize("B,n,m,k", [(2, 300, 1000, 3), (1, 8192, 1024, 3), (3, 257, 64, 1), (1, 100, 4100, 4), (2, 33, 2049, 2),
                                     (2, 70, 5, 3), (1, 10, 2, 3), (9, 8192, 300, 3)])
def test_knn_small_kernel_equals_lane_kernel(B, n, m, k):
    """The four-slot kernel (variant 3: three_nn and knn with k <= 4) returns the lane-per-query kernel's (variant 1) squared
    distances and indices bit for bit -- ascending, lowest index first on ties, (+inf, 0) in slots beyond the candidate count --
    on uniform, sorted, gridded and single-point clouds; so does the automatic choice (this kernel from 65 536 queries up)."""
    g = torch.Generator().manual_seed(m + k)
    uni = torch.rand((B, m, 3), generator=g)
    srt = torch.stack([c[torch.argsort(c[:, 0])] for c in torch.rand((B, m, 3), generator=g)])
    grid = torch.round(torch.rand((B, m, 3), generator=g) * 3) / 3
    same = torch.full((B, m, 3), 0.25)
    for name, cl in (("uniform", uni), ("sorted", srt), ("grid", grid), ("same", same)):
        cl = cl.cuda()
        q = (torch.rand((B, n, 3), generator=g) if name != "grid" else torch.round(torch.rand((B, n, 3), generator=g) * 3) / 3).cuda()
        d1, i1 = _knn_pair_variant(q, cl, k, 1)
        d3, i3 = _knn_pair_variant(q, cl, k, 3)
        assert torch.equal(i1, i3), name
        assert torch.equal(d1, d3), name
        d0, i0 = _knn_pair_variant(q, cl, k, 0)
        assert torch.equal(i0, i3) and torch.equal(d0, d3), name
    if m >= k:
        cl, q = uni[:1].cuda(), torch.rand((1, n, 3), generator=g).cuda()
        d3, i3 = _knn_pair_variant(q, cl, k, 3)
        dx = q[:, :, None, :] - cl[:, None, :, :]
        full = (dx[..., 0] * dx[..., 0] + dx[..., 1] * dx[..., 1]) + dx[..., 2] * dx[..., 2]
        order = torch.argsort(full, dim=-1, stable=True)[..., :k]
        assert torch.equal(i3.long(), order)


# ------------------------------------------------------------------ round 4: fixtures held by the reference
def test_ppfnet_query_ball_and_sample_and_group_multi_reference_golden(golden, monkeypatch):
    """utils/ppfnet_util.py:96-131 (query_ball_point with itself_indices) and :193-243 (sample_and_group_multi) against
    fixtures generated from the imported reference (tests/golden/make_golden_r4.py).  The reference samples its centres from a
    random start (:84), so the fixture's centres take the place of farthest_point_sample on both sides."""
    import learning3d_amd.utils.ppfnet_util as PP
    from learning3d_amd.utils import query_ball_point
    g = golden("ppfnet_util")
    xyz, nrm = dev(g["xyz"]), dev(g["normals"])
    fps = dev(g["fps"]).long()
    r, K = float(g["radius"]), int(g["nsample"])
    new_xyz = PP.index_points(xyz, fps)
    assert np.array_equal(query_ball_point(r, K, xyz, new_xyz, itself_indices=fps).cpu().numpy(), g["idx_itself"])
    assert np.array_equal(query_ball_point(r, K, xyz, new_xyz).cpu().numpy(), g["idx_plain"])
    assert np.array_equal(query_ball_point(0.02, K, xyz, new_xyz, itself_indices=fps).cpu().numpy(), g["idx_tiny"])
    monkeypatch.setattr(PP, "farthest_point_sample", lambda x, n: fps)
    S = fps.shape[1]
    out, grouped, fps_ret = PP.sample_and_group_multi(S, r, K, xyz, nrm, returnfps=True)
    assert torch.equal(fps_ret, fps)
    np.testing.assert_array_equal(out["xyz"].cpu().numpy(), g["multi_xyz"])
    np.testing.assert_array_equal(out["dxyz"].cpu().numpy(), g["multi_dxyz"])
    np.testing.assert_array_equal(grouped.cpu().numpy(), g["multi_grouped"])
    np.testing.assert_allclose(out["ppf"].cpu().numpy(), g["multi_ppf"], rtol=1e-5, atol=2e-6)     # atan2 / norm: libm vs device
    nx, npts = PP.sample_and_group(S, r, K, xyz, nrm)
    np.testing.assert_array_equal(nx.cpu().numpy(), g["sg_new_xyz"])
    np.testing.assert_array_equal(npts.cpu().numpy(), g["sg_new_points"])
    out_all = PP.sample_and_group_multi(-1, r, K, xyz[:, :64].contiguous(), nrm[:, :64].contiguous())
    np.testing.assert_array_equal(out_all["dxyz"].cpu().numpy(), g["all_dxyz"])
    np.testing.assert_allclose(out_all["ppf"].cpu().numpy(), g["all_ppf"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("tag,use_bn", [("ipcrnet", False), ("pnlk", True)])
def test_pointnet_trained_checkpoints_f16x2(golden, tag, use_bn):
    """The reference's two other trained PointNet feature extractors (pretrained/exp_ipcrnet and exp_pnlk, best_ptnet_model.t7)
    through the f16x2 route (and bf16x3): trained weight magnitudes beyond config 1's classifier checkpoint.  Features against
    the reference's own forward, rtol 1e-4 / atol 1e-5 of the feature scale."""
    from learning3d_amd.models import PointNet, _fused
    g = golden("ptnet_checkpoints")
    sd = {k[len(tag) + 3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(tag + ".w.")}
    net = PointNet(emb_dims=1024, use_bn=use_bn)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    want = g[tag + ".out"]
    scale = float(np.abs(want).max())
    for arith in ("f16x2", "bf16x3"):
        with _fused.arith(arith):
            got = net(dev(g["x"])).detach().cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5 * max(1.0, scale), err_msg=f"{tag} {arith}")
    _fused.check_range(sync=True)
