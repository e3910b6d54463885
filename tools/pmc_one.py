"""Run one kernel a few times (for rocprofv3 --pmc passes).
usage: pmc_one.py featknn|emd|emd_sweep|knn|chamfer|edgeconv|edgeconv_split|edgeconv_f16b|conv5|conv5_split|conv5_f16|conv5_f16_2p|group_c5|sa_mlp3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import learning3d_amd.utils as U
from learning3d_amd.losses.chamfer_distance import ChamferDistance
from learning3d_amd.models import DGCNN, _fused
what = sys.argv[1]
g = torch.Generator().manual_seed(0)
x = torch.rand((32, 1024, 3), generator=g).cuda(); a = torch.rand((32, 1024, 3), generator=g).cuda(); b = torch.rand((32, 1024, 3), generator=g).cuda()
net = DGCNN(emb_dims=1024).cuda().eval()
with torch.no_grad():
    idx = U.knn(x.permute(0, 2, 1), 20)
    packed = net._packed.get([net.conv1, net.conv2, net.conv3, net.conv4], [net.bn1, net.bn2, net.bn3, net.bn4], x.device)
    pooled = _fused.edgeconv_forward(x, idx, packed, kernel="split")
    w5, s5, b5, w5s, w5f = net._conv5_folded()
    img = _fused.edgeconv_forward(x, idx, packed, planes=True, v2=True)
    img2 = _fused.edgeconv_forward(x, idx, packed, planes=True, v2=True, unscaled=True)
    torch.cuda.synchronize()
    if what == "featknn":                        # feature-space kNN at B 32, C 64, N 1024, k 20 (featknn.hip: split + featknn_kernel<20, 1>)
        xf = torch.randn((32, 64, 1024), generator=g).cuda()
        for _ in range(5):
            U.knn(xf, 20)
        torch.cuda.synchronize()
        sys.exit(0)
    if what in ("bmm11", "bmm12", "bmm22"):     # l3d_bmm_f32 on a DCP training step's Linear shapes (R = 32768 rows, 512 -> 512): forward / dgrad / wgrad
        from learning3d_amd.models import _rows
        xx, ww, gg = torch.randn(32768, 512, device="cuda"), torch.randn(512, 512, device="cuda"), torch.randn(32768, 512, device="cuda")
        for _ in range(5):
            if what == "bmm11":
                _rows.bmm(xx, ww.t())
            elif what == "bmm12":
                _rows.bmm(gg, ww)
            else:
                _rows.bmm(gg.t(), xx, parts=_rows._split_parts(512, 512, 32768))
        torch.cuda.synchronize()
        sys.exit(0)
    if what == "group_c5":                       # config 5's grouping gather (bench.py --workload c5: the HBM-bound op)
        from learning3d_amd.utils import pointnet2_utils as P
        gq = torch.Generator().manual_seed(0)
        xyz5 = torch.clamp(torch.randn((32, 8192, 3), generator=gq), -2, 2).cuda()
        feat5 = torch.rand((32, 3, 8192), generator=gq).cuda()
        new5 = xyz5[:, :1024].contiguous()
        qg = P.QueryAndGroup(0.5, 16)
        for _ in range(5):
            qg(xyz5, new5, feat5)
        torch.cuda.synchronize()
        sys.exit(0)
    if what in ("emd", "emd_sweep", "emd_match"):                            # EMD forward at B 32, n = m = 1024 (emd.hip: 20 sweeps + match + costsum)
        from learning3d_amd._lib import lib, check, ptr, stream_ptr
        B_, n_ = 32, 1024
        ws_ = torch.empty(lib().l3d_emd_workspace_bytes(B_, n_, n_), dtype=torch.uint8, device="cuda")
        mt_ = torch.empty((B_, n_, n_), device="cuda"); c_ = torch.empty(B_, device="cuda")
        for _ in range(3):
            check(lib().l3d_emd_forward(ptr(a), ptr(b), B_, n_, n_, ptr(mt_), ptr(c_), ptr(ws_), 0, stream_ptr()), "emd")
        torch.cuda.synchronize()
        sys.exit(0)
    if what == "chamfer_c4":                     # config 4's Chamfer search at B 8 (l3d_chamfer_forward picks chamfer_mfma_kernel at 16384 x 16384)
        from learning3d_amd._lib import lib, check, ptr, stream_ptr
        a_, b_ = torch.rand(8, 16384, 3, device="cuda") - 0.5, torch.rand(8, 16384, 3, device="cuda") - 0.5
        d1_, d2_ = torch.empty(8, 16384, device="cuda"), torch.empty(8, 16384, device="cuda")
        i1_, i2_ = torch.empty(8, 16384, dtype=torch.int32, device="cuda"), torch.empty(8, 16384, dtype=torch.int32, device="cuda")
        for _ in range(3):
            check(lib().l3d_chamfer_forward(ptr(a_), ptr(b_), 8, 16384, 16384, ptr(d1_), ptr(d2_), ptr(i1_), ptr(i2_), stream_ptr()), "cd")
        torch.cuda.synchronize()
        sys.exit(0)
    if what == "attention":                      # DCP's attention call: B 32, 4 heads x 128, N = M = 1024, maxima ready, plane image out
        from learning3d_amd._lib import lib, check, ptr, stream_ptr
        B_, H_, D_, N_ = 32, 4, 128, 1024
        q_, k_, v_ = (torch.randn(B_, H_ * D_, N_, device="cuda") for _ in range(3))
        ws_ = torch.zeros(4, dtype=torch.int32, device="cuda")
        ctx_ = torch.empty_like(q_)
        img_ = torch.empty(lib().l3d_f16_image_bytes(1, B_ * N_, H_ * D_), dtype=torch.uint8, device="cuda")
        check(lib().l3d_attention_forward_f16b(ptr(q_), ptr(k_), ptr(v_), B_, H_, D_, N_, N_, H_ * D_ * N_, H_ * D_ * N_, H_ * D_ * N_, 1.0 / D_ ** 0.5,
                                               ptr(ws_), 0, ptr(ctx_), None, stream_ptr()), "att")          # leaves the maxima in ws_
        for _ in range(5):
            check(lib().l3d_attention_forward_f16b(ptr(q_), ptr(k_), ptr(v_), B_, H_, D_, N_, N_, H_ * D_ * N_, H_ * D_ * N_, H_ * D_ * N_, 1.0 / D_ ** 0.5,
                                                   ptr(ws_), 1, None, ptr(img_), stream_ptr()), "att")
        torch.cuda.synchronize()
        sys.exit(0)
    if what == "bq_cells":                       # config 5's ball query through the cell list (grouping.hip bq_cells_*)
        from learning3d_amd.utils import pointnet2_utils as P
        gq = torch.Generator().manual_seed(0)
        xyz5 = torch.clamp(torch.randn((32, 8192, 3), generator=gq), -2, 2).cuda()
        new5 = P.gather_operation(xyz5.transpose(1, 2).contiguous(), P.furthest_point_sample(xyz5, 1024)).transpose(1, 2).contiguous()
        for _ in range(5):
            P.ball_query(0.5, 16, xyz5, new5)
        torch.cuda.synchronize()
        sys.exit(0)
    if what == "sa_mlp3":                        # config 5's fused set-abstraction layer (sa_fused.hip), behind a ball query
        from learning3d_amd.models import PointNetSetAbstraction
        gq = torch.Generator().manual_seed(0)
        xyz5 = torch.clamp(torch.randn((32, 3, 8192), generator=gq), -2, 2).cuda()
        feat5 = torch.rand((32, 3, 8192), generator=gq).cuda()
        sa = PointNetSetAbstraction(npoint=1024, radius=0.5, nsample=16, in_channel=3, mlp=[32, 32, 64], group_all=False).cuda().eval()
        fps = sa.sample(xyz5)
        for _ in range(5):
            sa(xyz5, feat5, fps_idx=fps)
        torch.cuda.synchronize()
        sys.exit(0)
    for _ in range(5):
        if what == "knn": U.knn(x.permute(0, 2, 1), 20)
        elif what == "chamfer": ChamferDistance()(a, b)
        elif what == "edgeconv": _fused.edgeconv_forward(x, idx, packed, kernel="lds")            # fp32 MFMA
        elif what == "edgeconv_split": _fused.edgeconv_forward(x, idx, packed, kernel="split")        # bf16x3
        elif what == "edgeconv_f16b": _fused.edgeconv_forward(x, idx, packed, planes=True, v2=True, unscaled=True)      # f16x2 two-plane persistent kernel, out_mode 2 (the step's launch)
        elif what == "conv5_f16": _fused.pointwise_conv_f16(img, 32, 1024, w5f, 512, 1024, s5, b5, relu=True)
        elif what == "conv5_f16_2p": _fused.pointwise_conv_f16(img2, 32, 1024, w5f, 512, 1024, s5, b5, relu=True, unscaled=True)   # the step's conv5
        elif what == "conv5": _fused.pointwise_conv(pooled, w5, s5, b5, relu=True, channel_last=True, split=False)
        elif what == "conv5_split": _fused.pointwise_conv(pooled, w5, s5, b5, relu=True, channel_last=True, w_split=w5s)
    torch.cuda.synchronize()
