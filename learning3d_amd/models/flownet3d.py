"""Drop-in for learning3d/models/flownet3d.py (reference: models/flownet3d.py:73-328) on MI355X.

The reference needs the separately built `pointnet2_cuda` extension (and raises NameError without
it); here the same layers run on libl3d_hip.so: FPS, gather, ball query / kNN, grouping are the HIP
kernels of grouping.hip / knn.hip, and every Conv2d/Conv1d(k=1)+BN+ReLU stack is the fp32-MFMA GEMM
of mlp.hip with BN folded (inference).  Parameter names match the reference."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils import pointnet2_utils as pointutils
from ..utils.model_common_utils import query_ball_point
from . import _fused


FACTOR_FIRST_LAYER = True    # grouped first layers as per-point products + a gather (l3d_group_first_layer); False: grouped tensor + conv
# ... and the layers behind them as an f16x2 chain on fp16 plane images (conv_f16.hip's 128 x 512 tile, max over K in the last
# layer's epilogue).  Correct and tested, but OFF: at config 5's shapes these 128-channel layers have 8 K-chunks per tile and
# move 0.5 GB each -- 118 us on the narrow f16x2 tile against 165 us on the fp32 MFMA, which the extra bound / scale
# reductions (21 tiny launches per forward) more than eat: 5.12 -> 5.28 ms (tools/flownet_bench.py, LABLOG R2.4f).  Round 3: the
# bound's four data maxima in one launch (l3d_absmax4_partials, finished inside the layer's kernel) and the parameter-only terms
# cached: 5.00 -> 5.13 ms -- closer, still not a gain; stays off.
F16_GROUPED_STACK = False
F16_GROUPED_MIN_ROWS = 0      # rows (S * K) per cloud from which a grouped stack takes the f16x2 chain when F16_GROUPED_STACK is on
# A per-point (ungrouped) stack over at least F16_PLAIN_MIN_ROWS points of the whole batch as an f16x2 chain (one split pass, then
# conv_f16.hip with plane images between the layers) instead of bf16x3 / fp32-MFMA layers: feature propagation's 272 -> 256 -> 256 and
# the head's 256 -> 128 at N = 8192 (reference models/flownet3d.py:271-286, :325-327).  Correct (1.5e-8 from the other route) and OFF:
# the three GEMMs drop from 625 to 378 us, but the two split passes over 285 / 268 MB of fp32 input (maximum + split: 195 us each with
# the 16-byte absmax kernel, 545 before it) take more back -- forward 4.92 ms against 4.50 (tools/flownet_plain_ab.py, three interleaved
# pairs).  It would pay with the producers (l3d_three_interpolate_concat, the stack's own last layer) writing plane images; not built.
F16_PLAIN_STACK = os.environ.get("L3D_F16_PLAIN_STACK", "0") != "0"
F16_PLAIN_MIN_ROWS = 65536


def _mlp_stack(x, convs, bns, module, pool=False, cl_shape=None):
    """x [B,C,S,K] (or [B,C,N]) through [conv1x1 + BN + ReLU]*; fused MFMA path at inference.
    pool=True (x 4-D): also take the max over K, i.e. return [B,C',S] -- at inference the last layer's
    kernel does it in its epilogue (no [B,C',S,K] activation, no reduction launch).
    cl_shape = (S, K): x is [B, S*K, C] channel-last (the factored first layer's output, inference only)."""
    if len(convs) == 0:
        if cl_shape is not None:
            x = x.view(x.shape[0], cl_shape[0], cl_shape[1], x.shape[2]).permute(0, 3, 1, 2)
        return torch.max(x, -1)[0] if pool else x
    if cl_shape is not None or _fused.can_fuse(module, x):
        if cl_shape is not None:
            shp = (x.shape[0], x.shape[2], cl_shape[0], cl_shape[1])
            h = x
        else:
            shp = x.shape
            h = x.reshape(shp[0], shp[1], -1)
        last = len(convs) - 1
        if (F16_PLAIN_STACK and cl_shape is None and not pool and len(shp) == 3 and _fused.gemm_arith() == "f16x2"
                and shp[0] * shp[2] >= F16_PLAIN_MIN_ROWS
                and all(_fused.f16_eligible(h.shape[1] if i == 0 else convs[i - 1].out_channels, c.out_channels, shp[2])
                        for i, c in enumerate(convs))):
            # a per-point stack over many points (feature propagation and the head's conv1 at N = 8192, reference :271-286, :325-327):
            # one split pass over the input, then the f16x2 kernel layer by layer, plane images in between
            img = _fused.split_rows_f16(h, channel_first=True)
            cin = h.shape[1]
            for i, (conv, bn) in enumerate(zip(convs, bns)):
                w, sc, sh = _fused.fold_conv_bn(conv, bn)
                hit = conv.__dict__.get("_l3d_w_f16p")
                if hit is None or hit[0] != (w.data_ptr(), w._version, cin):
                    wp = F.pad(w, (0, cin - w.shape[1])).contiguous() if cin > w.shape[1] else w     # zero columns for zero channels
                    hit = ((w.data_ptr(), w._version, cin), _fused.split_weights_f16(wp))
                    conv.__dict__["_l3d_w_f16p"] = hit
                if i == last:
                    return _fused.pointwise_conv_f16(img, shp[0], shp[2], hit[1], cin, w.shape[0], sc, sh, relu=True)
                img = _fused.pointwise_conv_f16(img, shp[0], shp[2], hit[1], cin, w.shape[0], sc, sh, relu=True, out_planes=True)
                cin = w.shape[0]
        for i, (conv, bn) in enumerate(zip(convs, bns)):
            w, sc, sh = _fused.fold_conv_bn(conv, bn)
            if i == 0 and cl_shape is None and h.shape[1] > w.shape[1]:
                # the producer padded the input with zero channels to a multiple of 16 (PointNetFeaturePropogation: 259 -> 272,
                # which moves the layer from the fp32-MFMA kernel to the matrix-core split kernels): zero weight columns to match
                hit = conv.__dict__.get("_l3d_wpad")
                if hit is None or hit[0] != (w.data_ptr(), w._version, h.shape[1]):
                    hit = ((w.data_ptr(), w._version, h.shape[1]), F.pad(w, (0, h.shape[1] - w.shape[1])).contiguous())
                    conv.__dict__["_l3d_wpad"] = hit
                w = hit[1]
            cl = cl_shape is not None and i == 0
            n_pts = h.shape[1] if cl else h.shape[2]
            ws = None
            if _fused.split_eligible(w.shape[1], w.shape[0], n_pts):     # bf16x3 kernel: its weight planes are cached per layer
                hit = conv.__dict__.get("_l3d_wsplit")
                if hit is None or hit[0] != (w.data_ptr(), w._version):
                    hit = ((w.data_ptr(), w._version), _fused.split_rows(w))
                    conv.__dict__["_l3d_wsplit"] = hit
                ws = hit[1]
            if pool and i == last and len(shp) == 4:
                y = _fused.pointwise_conv_maxpool(h, w, sc, sh, True, shp[3], w_split=ws, channel_last=cl)
                if y is not None:
                    return y
            h = _fused.pointwise_conv(h, w, sc, sh, relu=True, channel_last=cl, w_split=ws)
        h = h.view(shp[0], h.shape[1], *shp[2:])
        return torch.max(h, -1)[0] if pool else h
    from ._train import conv_bn_act, hip_layers_ok
    for conv, bn in zip(convs, bns):
        # autograd is live (train-mode BatchNorm, or the backward recomputation of _fused.checkpointed): HIP conv / dgrad /
        # wgrad + BatchNorm kernels per layer (_train.py)
        x = conv_bn_act(x.contiguous(), conv, bn) if hip_layers_ok(x) else F.relu(bn(conv(x)))
    return torch.max(x, -1)[0] if pool else x


def _factored_first_layer(src_xyz_t, centre_xyz_t, src_feat, centre_feat, idx, order, conv, bn, module, planes=False):
    """conv1 + BN + ReLU of a grouped MLP (reference models/flownet3d.py:163-172, :226-232) without forming its input: the
    conv is linear in [xyz[idx] - centre | feat[idx] | centre_feat] (order 0; order 1: [feat[idx] | xyz[idx] - centre]), so
    U = (s W_feat) feat over the SOURCE points and V = (s W_centre) centre_feat + t over the centres are 1x1 convs on
    ungrouped tensors (K times fewer rows) and the layer is a gather: act(U[idx] + V + (s W_xyz)(xyz[idx] - centre)).
    -> [B, S*K, C1] channel-last (planes=True: the same as the fp16 activation image of conv_f16.hip, a uint8 tensor), or None
    when the route does not apply (the caller then groups and convolves)."""
    if (not FACTOR_FIRST_LAYER or not _fused.can_fuse(module, src_xyz_t, centre_xyz_t, src_feat) or not src_feat.is_cuda
            or idx.dtype != torch.int32 or (centre_feat is not None and not _fused.can_fuse(module, centre_feat))):
        return None
    from .._lib import check, lib, ptr, stream_ptr
    w, sc, sh = _fused.fold_conv_bn(conv, bn)
    C1, C = w.shape[0], src_feat.shape[1]
    Cc = centre_feat.shape[1] if centre_feat is not None else 0
    if C1 % 4 or C1 > 1024 or w.shape[1] != 3 + C + Cc:
        return None
    key = (w.data_ptr(), w._version, sc.data_ptr() if sc is not None else 0, order, C, Cc)
    hit = conv.__dict__.get("_l3d_factored")
    if hit is None or hit[0] != key:
        if order == 0:
            wx, wf, wc = w[:, :3], w[:, 3:3 + C], w[:, 3 + C:]
        else:
            wf, wx, wc = w[:, :C], w[:, C:C + 3], w[:, C + 3:]
        wx = (wx * sc[:, None] if sc is not None else wx).contiguous()
        # parameter-only terms of the plane route's bound, as Python floats (one sync per parameter version)
        wxr = float(wx.abs().sum(dim=1).max())
        shmax = float(sh.abs().max()) if sh is not None else 0.0
        hit = (key, (wf.contiguous(), wc.contiguous() if Cc else None, wx, wxr, shmax))
        conv.__dict__["_l3d_factored"] = hit
    wf, wc, wx, wxr, shmax = hit[1]
    B, N, _ = src_xyz_t.shape
    S, K = idx.shape[1], idx.shape[2]
    # the per-point products are ordinary 1x1 convs (K times fewer rows than the grouped layer); channel-last for the gather
    U = _fused.pointwise_conv(src_feat.float().contiguous(), wf, sc, None).transpose(1, 2).contiguous()           # [B,N,C1]
    V = None
    if Cc:
        V = _fused.pointwise_conv(centre_feat.float().contiguous(), wc, sc, sh).transpose(1, 2).contiguous()     # [B,S,C1]
    shp = sh if (V is None and sh is not None) else None
    sx, cx = src_xyz_t.contiguous(), centre_xyz_t.contiguous()
    if planes:
        # |output| <= max|U| + max|V| (or max|shift|) + max_r sum_d |wx_rd| * (max|src coordinate| + max|centre coordinate|): the four
        # data maxima in ONE launch (block maxima; the layer's kernel finishes the reduction and forms the bound itself)
        part = torch.empty(256, dtype=torch.float32, device=U.device)
        check(lib().l3d_absmax4_partials(ptr(U), U.numel(), ptr(V), V.numel() if V is not None else 0, ptr(sx), sx.numel(),
                                         ptr(cx), cx.numel(), ptr(part), stream_ptr()), "l3d_absmax4_partials")
        img = torch.empty(lib().l3d_f16_image_bytes(1, B * S * K, C1), dtype=torch.uint8, device=U.device)
        check(lib().l3d_group_first_layer_planes_auto(ptr(U), ptr(V), ptr(shp), ptr(wx), ptr(sx), ptr(cx), ptr(idx.contiguous()),
                                                      B, N, S, K, C1, 1, ptr(part), wxr, shmax if V is None else 0.0, ptr(img),
                                                      ptr(_fused.range_flag(U.device)), stream_ptr()),
              "l3d_group_first_layer_planes_auto")
        return img
    out = torch.empty((B, S * K, C1), dtype=torch.float32, device=U.device)
    check(lib().l3d_group_first_layer(ptr(U), ptr(V), ptr(shp), ptr(wx), ptr(sx), ptr(cx), ptr(idx.contiguous()),
                                      B, N, S, K, C1, 1, ptr(out), stream_ptr()), "l3d_group_first_layer")
    return out


def _f16_stack_ok(C1, convs, S, K, pool):
    """The layers behind a factored first layer can run as an f16x2 chain (conv_f16.hip: plane image in, plane image or
    grouped maxima out): every layer on one of its two tiles, max over K in the last layer's epilogue."""
    if not (F16_GROUPED_STACK and _fused.gemm_arith() == "f16x2" and len(convs) and pool and C1 in (64, 128, 256)
            and K in (8, 16, 32, 64) and S * K >= F16_GROUPED_MIN_ROWS):
        return False
    cin = C1
    for conv in convs:
        if not _fused.f16_eligible(cin, conv.out_channels, S * K):
            return False
        cin = conv.out_channels
    return True


def _f16_stack(img, B, S, K, convs, bns):
    """[conv1x1 + BN + ReLU]* on a plane image, max over K from the last layer's epilogue -> [B,C',S]"""
    N, last = S * K, len(convs) - 1
    for i, (conv, bn) in enumerate(zip(convs, bns)):
        w, sc, sh = _fused.fold_conv_bn(conv, bn)
        hit = conv.__dict__.get("_l3d_w_f16")
        if hit is None or hit[0] != (w.data_ptr(), w._version):
            hit = ((w.data_ptr(), w._version), _fused.split_weights_f16(w))
            conv.__dict__["_l3d_w_f16"] = hit
        if i == last:
            return _fused.pointwise_conv_f16_pool(img, B, N, hit[1], w.shape[1], w.shape[0], sc, sh, relu=True, group=K)[1]
        img = _fused.pointwise_conv_f16(img, B, N, hit[1], w.shape[1], w.shape[0], sc, sh, relu=True, out_planes=True)


def _grouped_input(src_xyz_t, centre_xyz_t, src_feat, centre_feat, idx, order, module):
    """[xyz[idx] - centre | feat[idx] | centre_feat] (order 0) or [feat[idx] | xyz[idx] - centre] (order 1) as ONE
    kernel (l3d_group_concat2) at inference; None -> the caller composes it from grouping ops (autograd route)."""
    if (not _fused.can_fuse(module, src_xyz_t, centre_xyz_t, src_feat) or not src_feat.is_cuda or idx.dtype != torch.int32
            or (centre_feat is not None and not _fused.can_fuse(module, centre_feat))):
        return None
    from .._lib import check, lib, ptr, stream_ptr
    B, N, _ = src_xyz_t.shape
    S, K = idx.shape[1], idx.shape[2]
    feat = src_feat.float().contiguous()
    C = feat.shape[1]
    cen = centre_feat.float().contiguous() if centre_feat is not None else None
    C1 = cen.shape[1] if cen is not None else 0
    out = torch.empty((B, 3 + C + C1, S, K), dtype=torch.float32, device=feat.device)
    check(lib().l3d_group_concat2(ptr(src_xyz_t.contiguous()), ptr(centre_xyz_t.contiguous()), ptr(feat), ptr(cen),
                                  ptr(idx.contiguous()), B, N, S, K, C, C1, order, ptr(out), stream_ptr()),
          "l3d_group_concat2")
    return out


class PointNetSetAbstraction(nn.Module):
    """reference :73-123.  xyz [B,3,N], points [B,D,N] -> new_xyz [B,3,S], new_points [B,D',S]."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.group_all = npoint, radius, nsample, group_all
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel + 3
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1, bias=False))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last_channel = out_channel
        self.queryandgroup = pointutils.GroupAll() if group_all else pointutils.QueryAndGroup(radius, nsample)

    def sample(self, xyz):
        """The furthest-point-sampling indices forward() would draw for `xyz` [B,3,N] (int32 [B,S]).  Sampling is 1024 dependent
        rounds on ONE workgroup per cloud (32 of 256 CUs at config 5) and depends on the coordinates only: a caller that streams
        batches can issue it for batch i + 1 on a second stream while batch i's ball query / grouping / MLP fill the other CUs,
        and hand the result to forward(..., fps_idx=...) (bench.py --workload c5 does)."""
        return pointutils.furthest_point_sample(xyz.permute(0, 2, 1).contiguous(), self.npoint)

    def forward(self, xyz, points, fps_idx=None):
        xyz_t = xyz.permute(0, 2, 1).contiguous()
        if not self.group_all:
            if fps_idx is None:
                with _fused.stage("fps"):
                    fps_idx = pointutils.furthest_point_sample(xyz_t, self.npoint)
            new_xyz = pointutils.gather_operation(xyz.contiguous(), fps_idx)          # [B,3,S]
        else:
            new_xyz = xyz
        if (_fused.SA_FUSED and not self.group_all and self.queryandgroup.use_xyz and self.queryandgroup.nsample in (8, 16, 32, 64)
                and xyz.is_cuda and _fused.can_fuse(self, xyz, points)):
            # a narrow stack (sa1: 6 -> 32 -> 32 -> 64): gather, three layers and the max over K in one kernel (sa_fused.hip)
            params = _fused.sa_mlp3_params(list(self.mlp_convs), list(self.mlp_bns), xyz.device)
            if params is not None and params[1] == (0 if points is None else points.shape[1]):
                new_xyz_t = new_xyz.transpose(2, 1).contiguous()
                with _fused.stage("ball_query"):
                    idx = pointutils.ball_query(self.queryandgroup.radius, self.queryandgroup.nsample, xyz_t, new_xyz_t)
                with _fused.stage("mlp"):
                    return new_xyz, _fused.sa_mlp3_fused(xyz_t, new_xyz_t, points, idx, params)
        new_points = self.queryandgroup(xyz_t, new_xyz.transpose(2, 1).contiguous(), points)   # [B,3+D,S,K]
        with _fused.stage("mlp"):
            return new_xyz, _mlp_stack(new_points, self.mlp_convs, self.mlp_bns, self, pool=True)


class FlowEmbedding(nn.Module):
    """reference :125-180."""

    def __init__(self, radius, nsample, in_channel, mlp, pooling='max', corr_func='concat', knn=True):
        super().__init__()
        self.radius, self.nsample, self.knn = radius, nsample, knn
        self.pooling, self.corr_func = pooling, corr_func
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel * 2 + 3
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1, bias=False))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last_channel = out_channel

    def forward(self, pos1, pos2, feature1, feature2):
        pos1_t = pos1.permute(0, 2, 1).contiguous()
        pos2_t = pos2.permute(0, 2, 1).contiguous()
        B, N, C = pos1_t.shape
        if self.knn:
            _, idx = pointutils.knn(self.nsample, pos1_t, pos2_t)
        else:
            idx, cnt = query_ball_point(self.radius, self.nsample, pos2_t, pos1_t, get_cnt=True)
            _, idx_knn = pointutils.knn(self.nsample, pos1_t, pos2_t)
            cnt = cnt.view(B, -1, 1).repeat(1, 1, self.nsample)
            idx = idx_knn[cnt > (self.nsample - 1)]
        S_, K_ = idx.shape[1], idx.shape[2]
        f16 = _f16_stack_ok(self.mlp_convs[0].out_channels, self.mlp_convs[1:], S_, K_, True)
        first = _factored_first_layer(pos2_t, pos1_t, feature2, feature1, idx, 0, self.mlp_convs[0], self.mlp_bns[0], self, planes=f16)
        if first is not None:
            if f16:
                return pos1, _f16_stack(first, B, S_, K_, self.mlp_convs[1:], self.mlp_bns[1:])
            return pos1, _mlp_stack(first, self.mlp_convs[1:], self.mlp_bns[1:], self, pool=True, cl_shape=(S_, K_))
        fused = _grouped_input(pos2_t, pos1_t, feature2, feature1, idx, 0, self)
        if fused is not None:
            return pos1, _mlp_stack(fused, self.mlp_convs, self.mlp_bns, self, pool=True)
        pos2_grouped = pointutils.grouping_operation(pos2.contiguous(), idx)           # [B,3,N,S]
        pos_diff = pos2_grouped - pos1.view(B, -1, N, 1)
        feat2_grouped = pointutils.grouping_operation(feature2.contiguous(), idx)
        feat_diff = torch.cat([feat2_grouped, feature1.view(B, -1, N, 1).repeat(1, 1, 1, self.nsample)], dim=1)
        feat1_new = torch.cat([pos_diff, feat_diff], dim=1)
        return pos1, _mlp_stack(feat1_new, self.mlp_convs, self.mlp_bns, self, pool=True)


class PointNetSetUpConv(nn.Module):
    """reference :182-242."""

    def __init__(self, nsample, radius, f1_channel, f2_channel, mlp, mlp2, knn=True):
        super().__init__()
        self.nsample, self.radius, self.knn = nsample, radius, knn
        self.mlp1_convs = nn.ModuleList()
        self.mlp2_convs = nn.ModuleList()
        last_channel = f2_channel + 3
        for out_channel in mlp:
            self.mlp1_convs.append(nn.Sequential(nn.Conv2d(last_channel, out_channel, 1, bias=False),
                                                 nn.BatchNorm2d(out_channel), nn.ReLU(inplace=False)))
            last_channel = out_channel
        last_channel = (mlp[-1] if len(mlp) != 0 else last_channel) + f1_channel
        for out_channel in mlp2:
            self.mlp2_convs.append(nn.Sequential(nn.Conv1d(last_channel, out_channel, 1, bias=False),
                                                 nn.BatchNorm1d(out_channel), nn.ReLU(inplace=False)))
            last_channel = out_channel

    def forward(self, pos1, pos2, feature1, feature2):
        pos1_t = pos1.permute(0, 2, 1).contiguous()
        pos2_t = pos2.permute(0, 2, 1).contiguous()
        B, C, N = pos1.shape
        if self.knn:
            _, idx = pointutils.knn(self.nsample, pos1_t, pos2_t)
        else:
            idx = query_ball_point(self.radius, self.nsample, pos2_t, pos1_t)
        first, f16 = None, False
        if len(self.mlp1_convs):
            f16 = _f16_stack_ok(self.mlp1_convs[0][0].out_channels, [s[0] for s in self.mlp1_convs[1:]], idx.shape[1], idx.shape[2], True)
            first = _factored_first_layer(pos2_t, pos1_t, feature2, None, idx, 1, self.mlp1_convs[0][0], self.mlp1_convs[0][1], self,
                                          planes=f16)
        feat_new = _grouped_input(pos2_t, pos1_t, feature2, None, idx, 1, self) if first is None else None
        if first is None and feat_new is None:
            pos2_grouped = pointutils.grouping_operation(pos2.contiguous(), idx)
            pos_diff = pos2_grouped - pos1.view(B, -1, N, 1)
            feat2_grouped = pointutils.grouping_operation(feature2.contiguous(), idx)
            feat_new = torch.cat([feat2_grouped, pos_diff], dim=1)
        if first is not None and f16:
            feat_new = _f16_stack(first, B, idx.shape[1], idx.shape[2], [s[0] for s in self.mlp1_convs[1:]], [s[1] for s in self.mlp1_convs[1:]])
        elif first is not None:
            feat_new = _mlp_stack(first, [s[0] for s in self.mlp1_convs[1:]], [s[1] for s in self.mlp1_convs[1:]], self, pool=True,
                                  cl_shape=tuple(idx.shape[1:]))
        else:
            feat_new = _mlp_stack(feat_new, [s[0] for s in self.mlp1_convs], [s[1] for s in self.mlp1_convs], self, pool=True)
        if feature1 is not None:
            feat_new = torch.cat([feat_new, feature1], dim=1)
        return _mlp_stack(feat_new, [s[0] for s in self.mlp2_convs], [s[1] for s in self.mlp2_convs], self)


class PointNetFeaturePropogation(nn.Module):
    """reference :244-286 (3-NN inverse-distance interpolation + Conv1d stack)."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last_channel = out_channel

    def forward(self, pos1, pos2, feature1, feature2):
        pos1_t = pos1.permute(0, 2, 1).contiguous()
        pos2_t = pos2.permute(0, 2, 1).contiguous()
        B, C, N = pos1.shape
        dists, idx = pointutils.three_nn(pos1_t, pos2_t)
        dists = dists.clamp_min(1e-10)
        weight = 1.0 / dists
        weight = weight / torch.sum(weight, -1, keepdim=True)
        # reference :268: sum(grouping_operation(feature2, idx) * weight, -1) -- the same three-term weighted
        # sum as pointnet2's three_interpolate, which is one fused kernel (no [B,C,N,3] temporary)
        if _fused.can_fuse(self, feature2, weight) and feature2.is_cuda and (feature1 is None or _fused.can_fuse(self, feature1)):
            from .._lib import check, lib, ptr, stream_ptr
            f2 = feature2.float().contiguous()
            f1 = feature1.float().contiguous() if feature1 is not None else None
            cin = f2.shape[1] + (f1.shape[1] if f1 is not None else 0)
            if (f1 is not None and cin % 16 and len(self.mlp_convs) and _fused.split_eligible(cin + (-cin) % 16, self.mlp_convs[0].out_channels, N)):
                # conv1's 259 input channels keep it off the bf16x3 / f16x2 kernels (Cin % 16): carry zero channels along
                f1 = torch.cat([f1, f1.new_zeros((B, (-cin) % 16, N))], dim=1)
            c, c1, m = f2.shape[1], (f1.shape[1] if f1 is not None else 0), f2.shape[2]
            feat_new = torch.empty((B, c + c1, N), dtype=torch.float32, device=f2.device)
            check(lib().l3d_three_interpolate_concat(B, c, m, N, ptr(f2), ptr(idx.contiguous()), ptr(weight.contiguous()),
                                                     ptr(f1), c1, ptr(feat_new), stream_ptr()), "l3d_three_interpolate_concat")
        else:
            interpolated_feat = pointutils.three_interpolate(feature2.contiguous(), idx, weight.contiguous())
            feat_new = torch.cat([interpolated_feat, feature1], 1) if feature1 is not None else interpolated_feat
        return _mlp_stack(feat_new, self.mlp_convs, self.mlp_bns, self)


class FlowNet3D(nn.Module):
    """reference :289-328 (layer hyper-parameters :293-303)."""

    def __init__(self):
        super().__init__()
        self.sa1 = PointNetSetAbstraction(npoint=1024, radius=0.5, nsample=16, in_channel=3, mlp=[32, 32, 64], group_all=False)
        self.sa2 = PointNetSetAbstraction(npoint=256, radius=1.0, nsample=16, in_channel=64, mlp=[64, 64, 128], group_all=False)
        self.sa3 = PointNetSetAbstraction(npoint=64, radius=2.0, nsample=8, in_channel=128, mlp=[128, 128, 256], group_all=False)
        self.sa4 = PointNetSetAbstraction(npoint=16, radius=4.0, nsample=8, in_channel=256, mlp=[256, 256, 512], group_all=False)
        self.fe_layer = FlowEmbedding(radius=10.0, nsample=64, in_channel=128, mlp=[128, 128, 128], pooling='max', corr_func='concat')
        self.su1 = PointNetSetUpConv(nsample=8, radius=2.4, f1_channel=256, f2_channel=512, mlp=[], mlp2=[256, 256])
        self.su2 = PointNetSetUpConv(nsample=8, radius=1.2, f1_channel=128 + 128, f2_channel=256, mlp=[128, 128, 256], mlp2=[256])
        self.su3 = PointNetSetUpConv(nsample=8, radius=0.6, f1_channel=64, f2_channel=256, mlp=[128, 128, 256], mlp2=[256])
        self.fp = PointNetFeaturePropogation(in_channel=256 + 3, mlp=[256, 256])
        self.conv1 = nn.Conv1d(256, 128, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm1d(128)
        self.conv2 = nn.Conv1d(128, 3, kernel_size=1, bias=True)

    def forward(self, pc1, pc2, feature1, feature2):
        """reference :305-328.  Eval-mode BatchNorm: the fused route in every grad mode (_fused.checkpointed)."""
        return _fused.checkpointed(self, self._forward, pc1, pc2, feature1, feature2)

    def _forward(self, pc1, pc2, feature1, feature2):
        if (_fused.can_fuse(self, pc1, pc2, feature1, feature2) and pc1.is_cuda and pc1.shape == pc2.shape
                and feature1.shape == feature2.shape):
            # sa1 / sa2 are applied to BOTH clouds with the same weights (reference :307-310) and every op in them
            # is per cloud (eval-mode BN), so the two passes run as one over the concatenated batch: furthest
            # point sampling is latency-bound with one workgroup per cloud -- 2B workgroups take as long as B.
            B = pc1.shape[0]
            l1_pc, l1_feature = self.sa1(torch.cat([pc1, pc2], 0), torch.cat([feature1, feature2], 0))
            l2_pc, l2_feature = self.sa2(l1_pc, l1_feature)
            l1_pc1, l1_feature1 = l1_pc[:B], l1_feature[:B]
            l2_pc1, l2_feature1, l2_pc2, l2_feature2 = l2_pc[:B], l2_feature[:B], l2_pc[B:], l2_feature[B:]
        else:
            l1_pc1, l1_feature1 = self.sa1(pc1, feature1)
            l2_pc1, l2_feature1 = self.sa2(l1_pc1, l1_feature1)
            l1_pc2, l1_feature2 = self.sa1(pc2, feature2)
            l2_pc2, l2_feature2 = self.sa2(l1_pc2, l1_feature2)
        _, l2_feature1_new = self.fe_layer(l2_pc1, l2_pc2, l2_feature1, l2_feature2)
        l3_pc1, l3_feature1 = self.sa3(l2_pc1, l2_feature1_new)
        l4_pc1, l4_feature1 = self.sa4(l3_pc1, l3_feature1)
        l3_fnew1 = self.su1(l3_pc1, l4_pc1, l3_feature1, l4_feature1)
        l2_fnew1 = self.su2(l2_pc1, l3_pc1, torch.cat([l2_feature1, l2_feature1_new], dim=1), l3_fnew1)
        l1_fnew1 = self.su3(l1_pc1, l2_pc1, l1_feature1, l2_fnew1)
        l0_fnew1 = self.fp(pc1, l1_pc1, feature1, l1_fnew1)
        x = _mlp_stack(l0_fnew1, [self.conv1], [self.bn1], self)
        if _fused.can_fuse(self, x):                  # the 128 -> 3 head on the narrow-head kernel (mlp.hip), not a torch conv
            w, sc, sh = _fused.fold_conv_bn(self.conv2)
            return _fused.pointwise_conv(x, w, sc, sh, relu=False)
        from ._train import conv_bn_act, hip_layers_ok
        return conv_bn_act(x.contiguous(), self.conv2, None, relu=False) if hip_layers_ok(x) else self.conv2(x)
