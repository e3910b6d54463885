"""The hot-path callers inside learning3d/utils/curvenet_util.py on MI355X: LPFA's neighbourhood grouping
(reference: utils/curvenet_util.py:229-291).  `knn` there is utils/model_common_utils.knn on the xyz coordinates with
add_one_to_k (:264) -- the fused HIP kNN -- and group_feature's gathers / concatenation run as one kernel
(l3d_lpfa_group).  The curve walk / aggregation modules of CurveNet are out of scope."""
import torch
import torch.nn as nn
import torch.nn.functional as F

import struct

from .._lib import check, f32c, lib, ptr, require_gpu, stream_ptr
from .model_common_utils import farthest_point_sample, index_points, knn, query_ball_point, square_distance  # noqa: F401


_ACT_LRELU = struct.unpack("<i", struct.pack("<f", 0.2))[0]      # activation code of the conv kernels: the LeakyReLU slope's fp32 bits


def lpfa_group(xyz, x, idx):
    """xyz [B,3,N], x [B,C,N] or None, idx int64 [B,N,k] -> (geo [B,9,N,k], diff [B,C,N,k] or None)"""
    require_gpu(xyz, idx)
    B, _, N = xyz.shape
    k = idx.shape[2]
    p = f32c(xyz.transpose(2, 1))
    geo = torch.empty((B, 9, N, k), dtype=torch.float32, device=xyz.device)
    diff, xc, C = None, None, 0
    if x is not None:
        xc = f32c(x)
        C = xc.shape[1]
        diff = torch.empty((B, C, N, k), dtype=torch.float32, device=xyz.device)
    check(lib().l3d_lpfa_group(ptr(p), ptr(xc), ptr(idx.contiguous()), B, N, C, k, ptr(geo), ptr(diff), stream_ptr()), "l3d_lpfa_group")
    return geo, diff


class LPFA(nn.Module):
    """reference: utils/curvenet_util.py:229-291 (same constructor, attribute names and state_dict keys)."""

    def __init__(self, in_channel, out_channel, k, mlp_num=2, initial=False):
        super(LPFA, self).__init__()
        self.k = k
        self.initial = initial
        if not initial:
            self.xyz2feature = nn.Sequential(nn.Conv2d(9, in_channel, kernel_size=1, bias=False), nn.BatchNorm2d(in_channel))
        mlp = []
        for _ in range(mlp_num):
            mlp.append(nn.Sequential(nn.Conv2d(in_channel, out_channel, 1, bias=False), nn.BatchNorm2d(out_channel), nn.LeakyReLU(0.2)))
            in_channel = out_channel
        self.mlp = nn.Sequential(*mlp)

    def _conv_bn(self, seq, h, act):
        """Conv2d + BatchNorm2d (+ LeakyReLU 0.2) of one nn.Sequential: with autograd live on the GPU the HIP conv / dgrad /
        wgrad + BatchNorm layer (models/_train.py), else the torch modules."""
        from ..models._train import conv_bn_act, hip_layers_ok
        if torch.is_grad_enabled() and hip_layers_ok(h) and (h.requires_grad or seq[0].weight.requires_grad):
            return conv_bn_act(h.contiguous(), seq[0], seq[1], relu=(_ACT_LRELU if act else 0))
        return seq(h)

    def forward(self, x, xyz, idx=None):
        x = self.group_feature(x, xyz, idx)
        for seq in self.mlp:
            x = self._conv_bn(seq, x, True)
        return x.max(dim=-1, keepdim=False)[0] if self.initial else x.mean(dim=-1, keepdim=False)

    def group_feature(self, x, xyz, idx):
        if idx is None:
            idx = knn(xyz, k=self.k, add_one_to_k=True)[:, :, :self.k]            # (batch_size, num_points, k)
        grad = torch.is_grad_enabled() and ((x is not None and x.requires_grad) or xyz.requires_grad)
        if grad:                                                                  # autograd: the reference's op sequence on the HIP kNN
            B, C, N = x.shape
            pts = xyz.transpose(2, 1).contiguous()
            nb = index_points(pts, idx) if not pts.requires_grad else torch.gather(
                pts.unsqueeze(1).expand(B, N, N, 3), 2, idx.unsqueeze(-1).expand(B, N, self.k, 3))
            ctr = pts.view(B, N, 1, 3).expand(-1, -1, self.k, -1)
            geo = torch.cat((ctr, nb, nb - ctr), dim=3).permute(0, 3, 1, 2).contiguous()
            if self.initial:
                return geo
            xt = x.transpose(2, 1)
            feat = torch.gather(xt.unsqueeze(1).expand(B, N, N, C), 2, idx.unsqueeze(-1).expand(B, N, self.k, C)) - xt.unsqueeze(2)
            return F.leaky_relu(feat.permute(0, 3, 1, 2) + self._conv_bn(self.xyz2feature, geo, False), 0.2)
        geo, diff = lpfa_group(xyz, None if self.initial else x, idx)
        if self.initial:
            return geo
        return F.leaky_relu(diff + self.xyz2feature(geo), 0.2)
