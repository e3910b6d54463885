"""INTEGRATION.md section B, executed: the ctypes shims a learning3d maintainer would paste in place of the reference's three
native extensions (`cd`, `pointnet2_cuda`, `_emd_ext._emd`) are taken from the document itself, exec'd, and driven the way
the reference's own Python wrappers drive their extension -- same argument order, outputs allocated where and how the
reference allocates them (losses/cuda/chamfer_distance/chamfer_distance.py:14-61, utils/lib/pointnet2_utils.py:10-256,
losses/cuda/emd_torch/pkg/layer/emd_loss_layer.py:7-19).  Results must equal this package's own Python API on the same
inputs.  /root/reference is not read: the call patterns are restated here, the shim code comes from INTEGRATION.md."""
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shims():
    from learning3d_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "_l3d." in b]
    src = "\n".join(blocks).replace("/path/to/learning3d_amd/libl3d_hip.so", _lib.LIB_PATH)
    ns = {"torch": torch}
    exec(compile(src, "INTEGRATION.md", "exec"), ns)
    for name in ("cd", "pointnet2", "emd", "knn"):
        assert name in ns, f"INTEGRATION.md no longer defines the `{name}` shim"
    return ns


def rand(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * (hi - lo) + lo).cuda()


def test_cd_shim_as_chamfer_distance_function_drives_it(shims):
    from learning3d_amd.losses.chamfer_distance import ChamferDistanceFunction
    cd = shims["cd"]
    xyz1, xyz2 = rand((3, 500, 3), 1), rand((3, 700, 3), 2)
    B, n, m = 3, 500, 700
    # chamfer_distance.py:21-25: zeros, int32 indices, then cd.forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2)
    dist1, dist2 = torch.zeros(B, n).cuda(), torch.zeros(B, m).cuda()
    idx1, idx2 = torch.zeros(B, n, dtype=torch.int).cuda(), torch.zeros(B, m, dtype=torch.int).cuda()
    cd.forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2)
    a, b = xyz1.clone().requires_grad_(), xyz2.clone().requires_grad_()
    d1, d2 = ChamferDistanceFunction.apply(a, b)
    assert torch.equal(dist1, d1) and torch.equal(dist2, d2)
    # :47-57: gradient buffers allocated by the wrapper, cd.backward_cuda(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
    g1, g2 = rand((B, n), 3), rand((B, m), 4)
    gx1, gx2 = torch.zeros_like(xyz1), torch.zeros_like(xyz2)
    cd.backward_cuda(xyz1, xyz2, gx1, gx2, g1, g2, idx1, idx2)
    torch.autograd.backward([d1, d2], [g1, g2])
    assert torch.equal(gx1, a.grad) and torch.equal(gx2, b.grad)


def test_pointnet2_shim_all_ten_wrappers(shims):
    from learning3d_amd.utils import pointnet2_utils as P
    pn = shims["pointnet2"]
    B, N, S, K, Cc = 2, 1024, 128, 16, 8
    xyz = rand((B, N, 3), 5, -1, 1)
    feat = rand((B, Cc, N), 6)
    # FurthestPointSampling.forward (pointnet2_utils.py:12-29): IntTensor output, temp filled with 1e10
    fps = torch.cuda.IntTensor(B, S)
    temp = torch.cuda.FloatTensor(B, N).fill_(1e10)
    pn.furthest_point_sampling_wrapper(B, N, S, xyz, temp, fps)
    assert torch.equal(fps, P.furthest_point_sample(xyz, S))
    # GatherOperation.forward (:42-58) / backward (:63-70)
    out = torch.cuda.FloatTensor(B, Cc, S)
    pn.gather_points_wrapper(B, Cc, N, S, feat, fps, out)
    assert torch.equal(out, P.gather_operation(feat, fps))
    gout = rand((B, Cc, S), 7)
    gfeat = torch.cuda.FloatTensor(B, Cc, N).zero_()
    pn.gather_points_grad_wrapper(B, Cc, N, S, gout, fps, gfeat)
    f2 = feat.clone().requires_grad_()
    (P.gather_operation(f2, fps) * gout).sum().backward()
    np.testing.assert_allclose(gfeat.cpu().numpy(), f2.grad.cpu().numpy(), rtol=1e-6, atol=1e-6)
    new_xyz = out.new_empty((B, S, 3))
    new_xyz.copy_(P.gather_operation(xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2))
    # BallQuery.forward (:231-249): idx pre-zeroed, ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
    idx = torch.cuda.IntTensor(B, S, K).zero_()
    pn.ball_query_wrapper(B, N, S, 0.3, K, new_xyz, xyz, idx)
    assert torch.equal(idx, P.ball_query(0.3, K, xyz, new_xyz))
    # GroupingOperation.forward (:187-205) / backward (:208-222)
    grouped = torch.cuda.FloatTensor(B, Cc, S, K)
    pn.group_points_wrapper(B, Cc, N, S, K, feat, idx, grouped)
    assert torch.equal(grouped, P.grouping_operation(feat, idx))
    ggr = rand((B, Cc, S, K), 8)
    gfeat2 = torch.cuda.FloatTensor(B, Cc, N).zero_()
    pn.group_points_grad_wrapper(B, Cc, N, S, K, ggr, idx, gfeat2)
    f3 = feat.clone().requires_grad_()
    (P.grouping_operation(f3, idx) * ggr).sum().backward()
    np.testing.assert_allclose(gfeat2.cpu().numpy(), f3.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
    # KNN.forward (:78-97): knn_wrapper(B, N, m, k, unknown, known, dist2, idx) then sqrt
    k = 5
    d2 = torch.cuda.FloatTensor(B, S, k)
    kidx = torch.cuda.IntTensor(B, S, k)
    pn.knn_wrapper(B, S, N, k, new_xyz, xyz, d2, kidx)
    dd, ii = P.knn(k, new_xyz, xyz)
    assert torch.equal(kidx, ii) and torch.equal(torch.sqrt(d2), dd)
    # ThreeNN.forward (:107-126), ThreeInterpolate.forward (:139-159) / backward (:162-178)
    d3 = torch.cuda.FloatTensor(B, N, 3)
    i3 = torch.cuda.IntTensor(B, N, 3)
    pn.three_nn_wrapper(B, N, S, xyz, new_xyz, d3, i3)
    dd3, ii3 = P.three_nn(xyz, new_xyz)
    assert torch.equal(i3, ii3) and torch.equal(torch.sqrt(d3), dd3)
    w = 1.0 / (dd3 + 1e-8)
    w = (w / w.sum(dim=2, keepdim=True)).contiguous()
    sfeat = rand((B, Cc, S), 9)
    interp = torch.cuda.FloatTensor(B, Cc, N)
    pn.three_interpolate_wrapper(B, Cc, S, N, sfeat, i3, w, interp)
    assert torch.equal(interp, P.three_interpolate(sfeat, i3, w))
    gint = rand((B, Cc, N), 10)
    gs = torch.cuda.FloatTensor(B, Cc, S).zero_()
    pn.three_interpolate_grad_wrapper(B, Cc, N, S, gint, i3, w, gs)
    s2 = sfeat.clone().requires_grad_()
    (P.three_interpolate(s2, i3, w) * gint).sum().backward()
    np.testing.assert_allclose(gs.cpu().numpy(), s2.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_emd_shim(shims):
    from learning3d_amd.losses.emd import EMDFunction
    emd = shims["emd"]
    a, b = rand((2, 256, 3), 11), rand((2, 256, 3), 12)
    cost, match = emd.emd_forward(a, b)                       # emd_loss_layer.py:10-13
    a2, b2 = a.clone().requires_grad_(), b.clone().requires_grad_()
    c2 = EMDFunction.apply(a2, b2)
    assert torch.equal(cost, c2)
    g1, g2 = emd.emd_backward(a, b, match)                    # :16-19
    c2.sum().backward()
    assert torch.equal(g1, a2.grad) and torch.equal(g2, b2.grad)


def test_knn_body_replacement(shims):
    import learning3d_amd.utils as U
    x = rand((2, 3, 700), 13)
    assert torch.equal(shims["knn"](x, 20), U.knn(x, 20))
    assert torch.equal(shims["knn"](x, 20, add_one_to_k=True), U.knn(x, 20, add_one_to_k=True))
