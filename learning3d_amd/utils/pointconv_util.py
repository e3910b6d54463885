"""Drop-in for learning3d/utils/pointconv_util.py on MI355X (BASELINE.json north_star names this file).

The point-set helpers call the HIP kernels: `square_distance`, `index_points`, `query_ball_point` are the same
functions as utils/model_common_utils.py's (identical bodies in the reference, pointconv_util.py:18-58, :85-105);
`farthest_point_sample` always starts at index 0 (:60-83) and `knn_point` ranks the EXPANDED distance and returns
indices only (:107-118).  The small PointConv modules around them (DensityNet, WeightNet, the two set-abstraction
layers, :205-371) keep the reference's attribute names, so checkpoints load unchanged; their 1x1-conv stacks over
[B,C,K,S] are torch modules (they are 8-16 channels wide: launch-bound, not on the hot path).
`compute_density` (:194-203) is fused: no [B,N,N] tensor.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .._lib import check, f32c, lib, ptr, require_gpu, stream_ptr
from . import model_common_utils as _m

square_distance = _m.square_distance
index_points = _m.index_points
query_ball_point = _m.query_ball_point


def farthest_point_sample(xyz, npoint):
    """reference: utils/pointconv_util.py:60-83 (first centroid is index 0)."""
    return _m.farthest_point_sample(xyz, npoint, start_with_first_point=True)


def knn_point(nsample, xyz, new_xyz):
    """reference: utils/pointconv_util.py:107-118.  xyz [B,N,3] searched, new_xyz [B,S,3] queries ->
    group_idx int64 [B,S,nsample]: the nsample smallest entries of square_distance(new_xyz, xyz).  The reference's
    topk(sorted=False) leaves the order within a row unspecified; here nearest first."""
    require_gpu(xyz, new_xyz)
    B, N, Cc = xyz.shape
    S = new_xyz.shape[1]
    if Cc != 3:
        raise NotImplementedError("knn_point: C=3 only")
    if nsample > N:
        raise RuntimeError("selected index k out of range")      # what torch.topk raises
    x, q = f32c(xyz), f32c(new_xyz)
    idx = torch.empty((B, S, nsample), dtype=torch.int64, device=xyz.device)
    check(lib().l3d_knn_point_expanded(nsample, ptr(x), ptr(q), B, N, S, ptr(idx), stream_ptr()),
          "l3d_knn_point_expanded")
    return idx


def sample_and_group(npoint, nsample, xyz, points, density_scale=None):
    """reference: :120-149.  xyz [B,N,3], points [B,N,D] -> new_xyz [B,S,3], new_points [B,S,K,3+D],
    grouped_xyz_norm [B,S,K,3], idx [B,S,K] (+ grouped_density)."""
    B, N, C = xyz.shape
    S = npoint
    fps_idx = farthest_point_sample(xyz, npoint)
    new_xyz = index_points(xyz, fps_idx)
    idx = knn_point(nsample, xyz, new_xyz)
    grouped_xyz_norm = index_points(xyz, idx) - new_xyz.view(B, S, 1, C)
    if points is not None:
        new_points = torch.cat([grouped_xyz_norm, index_points(points, idx)], dim=-1)
    else:
        new_points = grouped_xyz_norm
    if density_scale is None:
        return new_xyz, new_points, grouped_xyz_norm, idx
    return new_xyz, new_points, grouped_xyz_norm, idx, index_points(density_scale, idx)


def sample_and_group_all(xyz, points, density_scale=None):
    """reference: :151-172 (one group: the whole cloud, centred on its mean)."""
    B, N, C = xyz.shape
    new_xyz = xyz.mean(dim=1, keepdim=True)
    grouped_xyz = xyz.view(B, 1, N, C) - new_xyz.view(B, 1, 1, C)
    new_points = torch.cat([grouped_xyz, points.view(B, 1, N, -1)], dim=-1) if points is not None else grouped_xyz
    if density_scale is None:
        return new_xyz, new_points, grouped_xyz
    return new_xyz, new_points, grouped_xyz, density_scale.view(B, 1, N, 1)


def group(nsample, xyz, points):
    """reference: :174-192 (every point is a centre)."""
    B, N, C = xyz.shape
    idx = knn_point(nsample, xyz, xyz)
    grouped_xyz_norm = index_points(xyz, idx) - xyz.view(B, N, 1, C)
    if points is not None:
        return torch.cat([grouped_xyz_norm, index_points(points, idx)], dim=-1), grouped_xyz_norm
    return grouped_xyz_norm, grouped_xyz_norm


def compute_density(xyz, bandwidth):
    """reference: :194-203.  xyz [B,N,3] -> mean_j exp(-d2_ij / (2 bw^2)) / (2.5 bw), [B,N]; one fused pass
    (l3d_gaussian_density) instead of square_distance + exp + mean over a [B,N,N] tensor."""
    require_gpu(xyz)
    B, N, Cc = xyz.shape
    if Cc != 3:
        raise NotImplementedError("compute_density: C=3 only")
    x = f32c(xyz)
    out = torch.empty((B, N), dtype=torch.float32, device=xyz.device)
    check(lib().l3d_gaussian_density(ptr(x), B, N, float(bandwidth), ptr(out), stream_ptr()), "l3d_gaussian_density")
    return out


class DensityNet(nn.Module):
    """reference: :205-229 (the sigmoid branch `i == len(...)` is unreachable there too: every layer is ReLU)."""

    def __init__(self, hidden_unit=[16, 8]):
        super(DensityNet, self).__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        chans = [1] + list(hidden_unit) + [1]
        for cin, cout in zip(chans[:-1], chans[1:]):
            self.mlp_convs.append(nn.Conv2d(cin, cout, 1))
            self.mlp_bns.append(nn.BatchNorm2d(cout))

    def forward(self, density_scale):
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            density_scale = F.relu(bn(conv(density_scale)))
        return density_scale


class WeightNet(nn.Module):
    """reference: :231-259."""

    def __init__(self, in_channel, out_channel, hidden_unit=[8, 8]):
        super(WeightNet, self).__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        chans = [in_channel] + (list(hidden_unit) if hidden_unit else []) + [out_channel]
        for cin, cout in zip(chans[:-1], chans[1:]):
            self.mlp_convs.append(nn.Conv2d(cin, cout, 1))
            self.mlp_bns.append(nn.BatchNorm2d(cout))

    def forward(self, localized_xyz):
        weights = localized_xyz
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            weights = F.relu(bn(conv(weights)))
        return weights


class _PointConvBase(nn.Module):
    def __init__(self, npoint, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint = npoint
        self.nsample = nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last_channel = out_channel
        self.weightnet = WeightNet(3, 16)
        self.linear = nn.Linear(16 * mlp[-1], mlp[-1])
        self.bn_linear = nn.BatchNorm1d(mlp[-1])
        self.group_all = group_all

    def _tail(self, B, new_points, grouped_xyz_norm, new_xyz):
        grouped_xyz = grouped_xyz_norm.permute(0, 3, 2, 1)
        weights = self.weightnet(grouped_xyz)
        new_points = torch.matmul(input=new_points.permute(0, 3, 1, 2),
                                  other=weights.permute(0, 3, 2, 1)).view(B, self.npoint, -1)
        new_points = self.linear(new_points)
        new_points = F.relu(self.bn_linear(new_points.permute(0, 2, 1)))
        return new_xyz.permute(0, 2, 1), new_points


class PointConvSetAbstraction(_PointConvBase):
    """reference: :261-312.  xyz [B,3,N], points [B,D,N] -> new_xyz [B,3,S], new_points [B,D',S]."""

    def forward(self, xyz, points):
        B = xyz.shape[0]
        xyz = xyz.permute(0, 2, 1)
        if points is not None:
            points = points.permute(0, 2, 1)
        if self.group_all:
            new_xyz, new_points, grouped_xyz_norm = sample_and_group_all(xyz, points)
        else:
            new_xyz, new_points, grouped_xyz_norm, _ = sample_and_group(self.npoint, self.nsample, xyz, points)
        new_points = new_points.permute(0, 3, 2, 1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            new_points = F.relu(bn(conv(new_points)))
        return self._tail(B, new_points, grouped_xyz_norm, new_xyz)


class PointConvDensitySetAbstraction(_PointConvBase):
    """reference: :314-371 (adds the inverse-density scale through DensityNet)."""

    def __init__(self, npoint, nsample, in_channel, mlp, bandwidth, group_all):
        super().__init__(npoint, nsample, in_channel, mlp, group_all)
        self.densitynet = DensityNet()
        self.bandwidth = bandwidth

    def forward(self, xyz, points):
        B = xyz.shape[0]
        N = xyz.shape[2]
        xyz = xyz.permute(0, 2, 1)
        if points is not None:
            points = points.permute(0, 2, 1)
        xyz_density = compute_density(xyz, self.bandwidth)
        inverse_density = 1.0 / xyz_density
        if self.group_all:
            new_xyz, new_points, grouped_xyz_norm, grouped_density = sample_and_group_all(
                xyz, points, inverse_density.view(B, N, 1))
        else:
            new_xyz, new_points, grouped_xyz_norm, _, grouped_density = sample_and_group(
                self.npoint, self.nsample, xyz, points, inverse_density.view(B, N, 1))
        new_points = new_points.permute(0, 3, 2, 1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            new_points = F.relu(bn(conv(new_points)))
        inverse_max_density = grouped_density.max(dim=2, keepdim=True)[0]
        density_scale = grouped_density / inverse_max_density
        density_scale = self.densitynet(density_scale.permute(0, 3, 2, 1))
        new_points = new_points * density_scale
        return self._tail(B, new_points, grouped_xyz_norm, new_xyz)
