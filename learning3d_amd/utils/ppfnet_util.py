"""learning3d/utils/ppfnet_util.py's grouping helpers on top of the HIP FPS / ball-query / gather
kernels (sample_and_group, sample_and_group_multi, angle_difference).
reference: utils/ppfnet_util.py:11-28 (angle_difference), :134-170 (sample_and_group), :173-190 (angle),
:193-243 (sample_and_group_multi)."""
import numpy as np
import torch

from .model_common_utils import farthest_point_sample, index_points, query_ball_point, square_distance  # noqa: F401


def pc_normalize(pc):
    """numpy [N,3]: centre on the centroid, scale the farthest point to radius 1 (model_common_utils.py:11-17)"""
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


def angle_difference(src, dst):
    """pairwise angles between unit vectors: src [B,N,C], dst [B,M,C] -> [B,N,M]"""
    return torch.acos(torch.matmul(src, dst.permute(0, 2, 1)))


def angle(v1, v2):
    """atan2(|v1 x v2|, v1 . v2): well defined when either vector is zero"""
    v1, v2 = torch.broadcast_tensors(v1, v2)
    return torch.atan2(torch.norm(torch.linalg.cross(v1, v2, dim=-1), dim=-1), torch.sum(v1 * v2, dim=-1))


def _centres(npoint, xyz):
    B, N, _ = xyz.shape
    if npoint > 0:
        fps_idx = farthest_point_sample(xyz, npoint)
        return npoint, fps_idx, index_points(xyz, fps_idx)
    fps_idx = torch.arange(N, device=xyz.device)[None].repeat(B, 1)
    return N, fps_idx, xyz


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False):
    """FPS centres (all points if npoint <= 0) -> ball query -> centred neighbour coordinates (+ features).
    xyz [B,N,C], points [B,N,D] or None -> new_xyz [B,S,C], new_points [B,S,nsample,C(+D)]"""
    B, _, C = xyz.shape
    S, fps_idx, new_xyz = _centres(npoint, xyz)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = index_points(xyz, idx)
    new_points = grouped_xyz - new_xyz.view(B, S, 1, C)
    if points is not None:
        new_points = torch.cat([new_points, index_points(points, idx)], dim=-1)
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points


def sample_and_group_multi(npoint, radius, nsample, xyz, normals, returnfps=False):
    """as sample_and_group, returning {'xyz', 'dxyz', 'ppf'} with the 4-d point-pair features
    (angle(n_r, d), angle(n_i, d), angle(n_r, n_i), |d|); the centre is kept out of its own neighbourhood
    and only used as padding (query_ball_point's itself_indices)."""
    B, _, C = xyz.shape
    S, fps_idx, new_xyz = _centres(npoint, xyz)
    nr = (index_points(normals, fps_idx) if npoint > 0 else normals)[:, :, None, :]
    idx = query_ball_point(radius, nsample, xyz, new_xyz, itself_indices=fps_idx)
    grouped_xyz = index_points(xyz, idx)
    d = grouped_xyz - new_xyz.view(B, S, 1, C)
    ni = index_points(normals, idx)
    ppf = torch.stack([angle(nr, d), angle(ni, d), angle(nr, ni), torch.norm(d, dim=-1)], dim=-1)
    out = {"xyz": new_xyz, "dxyz": d, "ppf": ppf}
    if returnfps:
        return out, grouped_xyz, fps_idx
    return out
