"""Drop-in for learning3d/utils/lib/pointnet2_utils.py on MI355X.

The reference binds a separately built CUDA extension (`import pointnet2_cuda`, :7) that no longer
compiles on torch 2.x (THC headers); here the same autograd Functions / Modules call libl3d_hip.so.
Shapes, dtypes (int32 indices), contiguity asserts and argument order follow the reference
(file:line per class).
"""
import ctypes as C
from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from .._lib import check, f32c, lib, ptr, require_gpu, stream_ptr


# Backward of the gather-type ops: True = every target owns its sum and adds its contributions in ascending
# entry order (l3d_scatter_add_det: stable sort of the indices, then segment sums) -- same bits every run;
# False = the reference's scheme, fp32 atomicAdd scatter (group_points_gpu.cu:8-28 etc.): faster, run-to-run
# different in the low bits.
DETERMINISTIC_BACKWARD = True


def _scatter_add_det(src, idx, weight, T, div):
    """dst[b,c,t] = sum_{e: idx[b,e]==t, ascending e} src[b,c,e//div] * weight[b,e];  src [B,C,E/div], idx int32 [B,E]"""
    B, Cc = src.shape[0], src.shape[1]
    E = idx.numel() // B
    dst = torch.empty((B, Cc, T), dtype=torch.float32, device=src.device)
    ws = torch.empty(lib().l3d_scatter_add_det_workspace_bytes(B, T, E), dtype=torch.uint8, device=src.device)
    check(lib().l3d_scatter_add_det(ptr(src), ptr(idx), ptr(weight), B, Cc, T, E, div, ptr(ws), ptr(dst), stream_ptr()),
          "l3d_scatter_add_det")
    return dst


class FurthestPointSampling(Function):
    """reference: pointnet2_utils.py:10-33 -> K12 furthest_point_sampling_kernel."""

    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        assert xyz.is_contiguous()
        require_gpu(xyz)
        B, N, _ = xyz.size()
        output = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        check(lib().l3d_furthest_point_sampling(B, N, npoint, ptr(xyz), None, ptr(output), stream_ptr()),
              "l3d_furthest_point_sampling")
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    """reference: pointnet2_utils.py:39-70 -> K10 / K11."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        require_gpu(features, idx)
        B, npoint = idx.size()
        _, Cc, N = features.size()
        idx = idx.int()
        output = torch.empty((B, Cc, npoint), dtype=torch.float32, device=features.device)
        check(lib().l3d_gather_points(B, Cc, N, npoint, ptr(features), ptr(idx), ptr(output), stream_ptr()),
              "l3d_gather_points")
        ctx.for_backwards = (idx, Cc, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, Cc, N = ctx.for_backwards
        B, npoint = idx.size()
        g = grad_out.contiguous()
        if DETERMINISTIC_BACKWARD:
            return _scatter_add_det(g, idx, None, N, 1), None
        grad_features = torch.empty((B, Cc, N), dtype=torch.float32, device=grad_out.device)
        check(lib().l3d_gather_points_grad(B, Cc, N, npoint, ptr(g), ptr(idx), ptr(grad_features), stream_ptr()),
              "l3d_gather_points_grad")
        return grad_features, None


gather_operation = GatherOperation.apply


class KNN(Function):
    """reference: pointnet2_utils.py:72-101 -> K13 knn_kernel_fast.  Returns (sqrt(dist2), idx int32)."""

    @staticmethod
    def forward(ctx, k: int, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        require_gpu(unknown, known)
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty((B, N, k), dtype=torch.float32, device=unknown.device)
        idx = torch.empty((B, N, k), dtype=torch.int32, device=unknown.device)
        check(lib().l3d_knn(B, N, m, k, ptr(unknown), ptr(known), ptr(dist2), ptr(idx), stream_ptr()), "l3d_knn")
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None


knn = KNN.apply


class ThreeNN(Function):
    """reference: pointnet2_utils.py:103-133 -> K14 three_nn_kernel_fast."""

    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        require_gpu(unknown, known)
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty((B, N, 3), dtype=torch.float32, device=unknown.device)
        idx = torch.empty((B, N, 3), dtype=torch.int32, device=unknown.device)
        check(lib().l3d_three_nn(B, N, m, ptr(unknown), ptr(known), ptr(dist2), ptr(idx), stream_ptr()),
              "l3d_three_nn")
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """reference: pointnet2_utils.py:136-181 -> K15 / K16."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        assert weight.is_contiguous()
        require_gpu(features, idx, weight)
        B, c, m = features.size()
        n = idx.size(1)
        idx = idx.int()
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = torch.empty((B, c, n), dtype=torch.float32, device=features.device)
        check(lib().l3d_three_interpolate(B, c, m, n, ptr(features), ptr(idx), ptr(weight), ptr(output),
                                          stream_ptr()), "l3d_three_interpolate")
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        g = grad_out.contiguous()
        if DETERMINISTIC_BACKWARD:
            return _scatter_add_det(g, idx, weight, m, 3), None, None
        grad_features = torch.empty((B, c, m), dtype=torch.float32, device=grad_out.device)
        check(lib().l3d_three_interpolate_grad(B, c, n, m, ptr(g), ptr(idx), ptr(weight), ptr(grad_features),
                                               stream_ptr()), "l3d_three_interpolate_grad")
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """reference: pointnet2_utils.py:184-222 -> K8 / K9."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        require_gpu(features, idx)
        idx = idx.int()
        B, nfeatures, nsample = idx.size()
        _, Cc, N = features.size()
        output = torch.empty((B, Cc, nfeatures, nsample), dtype=torch.float32, device=features.device)
        check(lib().l3d_group_points(B, Cc, N, nfeatures, nsample, ptr(features), ptr(idx), ptr(output),
                                     stream_ptr()), "l3d_group_points")
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, N = ctx.for_backwards
        B, Cc, npoint, nsample = grad_out.size()
        g = grad_out.contiguous()
        if DETERMINISTIC_BACKWARD:
            return _scatter_add_det(g.view(B, Cc, npoint * nsample), idx, None, N, 1), None
        grad_features = torch.empty((B, Cc, N), dtype=torch.float32, device=grad_out.device)
        check(lib().l3d_group_points_grad(B, Cc, N, npoint, nsample, ptr(g), ptr(idx), ptr(grad_features),
                                          stream_ptr()), "l3d_group_points_grad")
        return grad_features, None


grouping_operation = GroupingOperation.apply


BALL_QUERY_CELLS = True      # ball_query on clouds of >= 2048 points through a per-cloud cell list; False: the scanning kernels


class BallQuery(Function):
    """reference: pointnet2_utils.py:225-253 -> K7 ball_query_kernel_fast."""

    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        require_gpu(xyz, new_xyz)
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.empty((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        # large clouds: scratch for the cell-list kernels (grouping.hip; the same indices as the scanning kernels)
        ws = torch.empty(B * (16 * N + 16448), dtype=torch.uint8, device=xyz.device) if BALL_QUERY_CELLS and N >= 2048 and nsample <= 64 else None
        check(lib().l3d_ball_query(B, N, npoint, C.c_float(radius), nsample, ptr(new_xyz), ptr(xyz), ptr(idx), ptr(ws),
                                   stream_ptr()), "l3d_ball_query")
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """reference: pointnet2_utils.py:259-292."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        from ..models._fused import stage
        with stage("ball_query"):
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        needs_grad = torch.is_grad_enabled() and (xyz.requires_grad or (features is not None and features.requires_grad))
        if not needs_grad and (self.use_xyz or features is not None):
            # one pass: gather + centring + concat (no grouped temporaries, no torch.cat copy)
            assert self.use_xyz or features is not None
            B, N, _ = xyz.shape
            S = new_xyz.shape[1]
            Cf = features.shape[1] if features is not None else 0
            x_, q_ = f32c(xyz), f32c(new_xyz)
            f_ = f32c(features) if features is not None else None
            out = torch.empty((B, (3 if self.use_xyz else 0) + Cf, S, self.nsample), dtype=torch.float32, device=xyz.device)
            with stage("group_kernel"):                  # the launch alone (bench.py --workload c5: the live HBM-roofline timing)
                rc = lib().l3d_group_concat(ptr(x_), ptr(q_), ptr(f_), ptr(idx), B, N, S, self.nsample, Cf, int(self.use_xyz),
                                            ptr(out), stream_ptr())
            check(rc, "l3d_group_concat")
            return out
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)                  # (B, 3, npoint, nsample)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return new_features


class GroupAll(nn.Module):
    """reference: pointnet2_utils.py:295-318 (pure view ops)."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            new_features = grouped_xyz
        return new_features
