"""BASELINE.json's configs at their own sizes against reference-derived goldens and the oracle ("configs tested
below their size"): FlowNet3D as a model (config 5), its sa1 layer at N = 8192 / S = 1024, PCN at num_coarse 1024 / grid 4
(config 4), the DGCNN features of all 32 clouds of config 2.  Index outputs bit-exact, features rtol 1e-4 / atol 1e-5."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from seeded import seeded_params  # noqa: E402

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def test_flownet3d_reference_golden(golden):
    """FlowNet3D (models/flownet3d.py:289-328) against the reference model's own output (make_golden.py: the reference's
    flownet3d.py + pointnet2_utils.py on the CPU stand-in for pointnet2_cuda), seeded weights by state_dict key, eval mode,
    grad mode ON as the reference's scripts run it.  Sampled coordinates bit-exact; features and flow rtol 1e-4 / atol 1e-5."""
    from learning3d_amd.models import FlowNet3D, _fused
    g = golden("flownet3d_seeded")
    net = FlowNet3D()
    assert sorted(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    net = seeded_params(net, int(g["seed"])).cuda().eval()
    pc1, pc2, f1, f2 = (dev(g[k]) for k in ("pc1", "pc2", "f1", "f2"))
    sf = net(pc1, pc2, f1, f2)
    np.testing.assert_allclose(sf.detach().cpu().numpy(), g["sf"], rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        l1_pc1, l1_f1 = net.sa1(pc1, f1)
        l2_pc1, l2_f1 = net.sa2(l1_pc1, l1_f1)
        l1_pc2, l1_f2 = net.sa1(pc2, f2)
        l2_pc2, l2_f2 = net.sa2(l1_pc2, l1_f2)
        _, l2_new = net.fe_layer(l2_pc1, l2_pc2, l2_f1, l2_f2)
    assert np.array_equal(l1_pc1.cpu().numpy(), g["l1_pc1"])
    np.testing.assert_allclose(l1_f1.cpu().numpy(), g["l1_feature1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(l2_f1.cpu().numpy(), g["l2_feature1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(l2_new.cpu().numpy(), g["l2_feature1_new"], rtol=1e-4, atol=1e-5)
    # the per-layer differentiable route (what a backward recomputes) is the same function
    with _fused.per_layer_route():
        sf2 = net(pc1, pc2, f1.clone().requires_grad_(), f2)
    np.testing.assert_allclose(sf2.detach().cpu().numpy(), g["sf"], rtol=1e-4, atol=1e-5)
    sf.sum().backward()                                                    # checkpointed backward runs end to end
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_flownet3d_whole_model_at_config5_size_vs_oracle():
    """The WHOLE FlowNet3D forward at config 5's point count, N = 8192 (bench.py's `c5_flownet3d_forward` inputs: clipped normal
    coordinates, second frame = first + 0.05 noise, uniform features), against oracle.flownet3d_forward_torch (the K7-K16
    restatements + torch-CPU conv / BatchNorm, itself pinned to the reference model at N = 2048 by test_oracle_golden.py): the
    routes that only N = 8192 selects -- four-slot 3-NN, cell-list ball query, knn_select, the fused set-abstraction kernel --
    are compared end to end here, not just timed.  Sampled coordinates bit-exact, features and flow rtol 1e-4 / atol 1e-5."""
    from learning3d_amd.models import FlowNet3D
    net = seeded_params(FlowNet3D(), 4).cuda().eval()
    g = torch.Generator().manual_seed(7)
    B, N = 2, 8192
    pc1 = torch.clamp(torch.randn((B, 3, N), generator=g), -2, 2)
    pc2 = (pc1 + 0.05 * torch.randn((B, 3, N), generator=g)).contiguous()
    f1, f2 = torch.rand((B, 3, N), generator=g), torch.rand((B, 3, N), generator=g)
    with torch.no_grad():
        sf = net(pc1.cuda(), pc2.cuda(), f1.cuda(), f2.cuda())
        l1_pc1, l1_f1 = net.sa1(pc1.cuda(), f1.cuda())
    w = {k: v.cpu().numpy() for k, v in net.state_dict().items()}
    want, inter = oracle.flownet3d_forward_torch(pc1.numpy(), pc2.numpy(), f1.numpy(), f2.numpy(), w, return_intermediates=True)
    assert np.array_equal(l1_pc1.cpu().numpy(), inter["l1_pc1"])
    np.testing.assert_allclose(l1_f1.cpu().numpy(), inter["l1_feature1"], rtol=1e-4, atol=1e-5)
    assert sf.shape == (B, 3, N)
    np.testing.assert_allclose(sf.cpu().numpy(), want, rtol=1e-4, atol=1e-5)


def test_flownet3d_sa1_config5_shape_golden(golden):
    """Config 5's layer at config size: sa1 (npoint 1024 of N 8192, r 0.5, K 16, mlp 32/32/64) against the reference's
    PointNetSetAbstraction.forward on the 4 golden clouds -- alone, and as clouds 0..3 of a 32-cloud batch (the per-GPU shard
    of B = 256): centroids bit-exact, features rtol 1e-4 / atol 1e-5, and a cloud's result does not depend on its batch."""
    from learning3d_amd.models import PointNetSetAbstraction
    g = golden("flownet3d_sa1_c5")
    sa = PointNetSetAbstraction(npoint=1024, radius=0.5, nsample=16, in_channel=3, mlp=[32, 32, 64], group_all=False)
    assert sorted(sa.state_dict().keys()) == [str(k) for k in g["keys"]]
    sa = seeded_params(sa, int(g["seed"])).cuda().eval()
    xyz4 = torch.clamp(torch.randn((4, 3, 8192), generator=torch.Generator().manual_seed(int(g["seed_xyz"]))), -2, 2)
    feat4 = torch.rand((4, 3, 8192), generator=torch.Generator().manual_seed(int(g["seed_feat"])))
    rest_x = torch.clamp(torch.randn((28, 3, 8192), generator=torch.Generator().manual_seed(5)), -2, 2)
    rest_f = torch.rand((28, 3, 8192), generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        nx4, nf4 = sa(xyz4.cuda(), feat4.cuda())
        nx32, nf32 = sa(torch.cat([xyz4, rest_x]).cuda(), torch.cat([feat4, rest_f]).cuda())
    assert np.array_equal(nx4.cpu().numpy(), g["new_xyz"])
    np.testing.assert_allclose(nf4.cpu().numpy(), g["new_feat"], rtol=1e-4, atol=1e-5)
    assert nf32.shape == (32, 64, 1024) and torch.equal(nx32[:4], nx4) and torch.equal(nf32[:4], nf4)
    # the other 28 clouds against the oracle composition (indices bit-exact through new_xyz, features 1e-4 / 1e-5) on 4 of them
    w = {k: v.cpu().numpy() for k, v in sa.state_dict().items()}
    ox, of = oracle.set_abstraction_forward_torch(rest_x[:4].numpy(), rest_f[:4].numpy(), w, "", 1024, 0.5, 16)
    assert np.array_equal(nx32[4:8].cpu().numpy(), ox)
    np.testing.assert_allclose(nf32[4:8].cpu().numpy(), of, rtol=1e-4, atol=1e-5)


def test_pcn_reference_golden_config4_shape(golden):
    """PCN at config 4's decoder shape (num_coarse 1024, grid_size 4 -> 16384 fine points, partial clouds of 2048 points)
    against the reference model's own output; fold_mlp_f16's per-workgroup scale bound is exercised at config size against
    the reference, not only against itself.  Both arithmetics."""
    from learning3d_amd.models import PCN, _fused
    g = golden("pcn_seeded_c4")
    net = PCN(emb_dims=1024, num_coarse=1024, grid_size=4, detailed_output=True)
    assert sorted(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    net = seeded_params(net, int(g["seed"])).cuda().eval()
    gx = torch.Generator().manual_seed(int(g["seed_x"]))
    x = (torch.rand((2, 2048, 3), generator=gx) - 0.5).cuda()
    for arith in ("f16x2", "bf16x3"):
        with _fused.arith(arith):
            out = net(x)
        for k in ("coarse_output", "fine_output"):
            np.testing.assert_allclose(out[k].detach().cpu().numpy(), g[k], rtol=1e-4, atol=1e-5, err_msg=f"{arith} {k}")
    _fused.check_range(sync=True)


def test_config4_chamfer_8_of_64_clouds():
    """Config 4's Chamfer stress (64 x 16384 x 16384 pairs per direction): distances of 8 of the 64 clouds bit-exact against
    the oracle's nnsearch restatement, argmins exact, loss against fp64 from the kernel's own distances."""
    from learning3d_amd.losses import ChamferDistanceLoss
    from learning3d_amd.losses.chamfer_distance import ChamferDistanceFunction
    g = torch.Generator().manual_seed(0)
    fine = (torch.rand((64, 16384, 3), generator=g) - 0.5).cuda()
    gt = (torch.rand((64, 16384, 3), generator=g) - 0.5).cuda()
    with torch.no_grad():
        loss = ChamferDistanceLoss()(gt, fine)
        d1, d2 = ChamferDistanceFunction.apply(gt, fine)
    sel = list(range(0, 64, 8))
    o1, o2, _, _ = oracle.chamfer_forward(gt[sel].cpu().numpy(), fine[sel].cpu().numpy())
    assert np.array_equal(d1[sel].cpu().numpy(), o1) and np.array_equal(d2[sel].cpu().numpy(), o2)
    want = (torch.sqrt(d1).double().mean() + torch.sqrt(d2).double().mean()) / 2
    assert abs(float(loss) - float(want)) < 1e-6


def test_config2_features_all_32_clouds_vs_oracle():
    """Config 2: DGCNN(emb 1024) features of ALL 32 clouds of the bench input (U(0,1)^3, seed 0) against the oracle port of
    models/dgcnn.py (torch-CPU convs on the oracle's kNN graph), rtol 1e-4 / atol 1e-5; grad mode on."""
    from learning3d_amd.models import DGCNN
    torch.manual_seed(1)
    net = DGCNN(emb_dims=1024).eval()
    x = torch.rand((32, 1024, 3), generator=torch.Generator().manual_seed(0)).numpy()
    w = {k: v.numpy() for k, v in net.state_dict().items()}
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    want = np.concatenate([oracle.dgcnn_forward_torch(x[i:i + 8], w).numpy() for i in range(0, 32, 8)])
    got = net.cuda()(dev(x)).detach().cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
