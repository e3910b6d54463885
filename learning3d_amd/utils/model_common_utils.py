"""Drop-in for learning3d/utils/model_common_utils.py on MI355X.

Same function names, argument order, shapes, dtypes and return layouts as the reference
(file:line cited per function); the bodies call the hand-written HIP kernels through the C ABI
(include/l3d_hip.h) instead of chaining ATen ops that materialise [B,N,N] temporaries.
"""
import ctypes as C

import torch

from .._lib import L3DError, check, f32c, lib, ptr, require_gpu, stream_ptr


def _as_bn3(x_bcn):
    """[B,3,N] (any strides) -> contiguous [B,N,3] without a copy when x is a permuted view of one."""
    xt = x_bcn.transpose(2, 1)
    return f32c(xt)


def knn(x, k, add_one_to_k=False):
    """reference: utils/model_common_utils.py:3-9.   x [B,3,N] -> idx int64 [B,N,k].

    Ranks exactly the fp32 values the reference ranks (pd = -xx_j - inner_ij - xx_i with the
    sgemm dot product as an fma chain); under exact ties, where torch.topk's order is
    unspecified, the lower index comes first."""
    if add_one_to_k:
        k = k + 1
    require_gpu(x)
    if x.dim() != 3:
        raise ValueError("knn expects x of shape [B, C, N]")
    B, Cc, N = x.shape
    if k > N:
        raise RuntimeError("selected index k out of range")      # what torch.topk raises
    if Cc != 3:
        # feature-space graphs (PRNet's DGCNN, models/prnet.py:76-97; C = 64..256): the distance is a real
        # GEMM -> matrix cores (bf16x3) with the top-k as its epilogue, no [B,N,N] tensor
        xf = f32c(x)
        if k <= 64:                                              # any C (zero-padded to a multiple of 32 in the split pass)
            ws = torch.empty(lib().l3d_knn_feature_workspace_bytes(B, Cc, N), dtype=torch.uint8, device=x.device)
            idx = torch.empty((B, N, k), dtype=torch.int64, device=x.device)
            check(lib().l3d_knn_feature(ptr(xf), B, Cc, N, k, ptr(ws), ptr(idx), stream_ptr()), "l3d_knn_feature")
            return idx
        # k > 64 in feature space: no caller in the reference asks for it; the op sequence itself (:4-8) on the device
        inner = -2 * torch.matmul(xf.transpose(2, 1), xf)
        xx = torch.sum(xf ** 2, dim=1, keepdim=True)
        pairwise_distance = -xx - inner - xx.transpose(2, 1)
        return pairwise_distance.topk(k=k, dim=-1)[1]
    xyz = _as_bn3(x)
    idx = torch.empty((B, N, k), dtype=torch.int64, device=x.device)
    check(lib().l3d_knn_graph(ptr(xyz), B, N, k, ptr(idx), stream_ptr()), "l3d_knn_graph")
    return idx


def square_distance(src, dst):
    """reference: utils/model_common_utils.py:19-38.  [B,N,3],[B,M,3] -> [B,N,M] fp32."""
    require_gpu(src, dst)
    if torch.is_grad_enabled() and (src.requires_grad or dst.requires_grad):
        # differentiable like the reference's matmul form (:34-37): the forward kernel below, the gradients on l3d_bmm_f32
        if src.dtype == torch.float32 and dst.dtype == torch.float32 and src.shape[2] == dst.shape[2]:
            from ..models import _rows
            return _rows.square_distance(src, dst)
        B, N, _ = src.shape
        M = dst.shape[1]
        dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
        dist = dist + torch.sum(src ** 2, -1).view(B, N, 1)
        return dist + torch.sum(dst ** 2, -1).view(B, 1, M)
    B, N, Cc = src.shape
    M = dst.shape[1]
    if dst.shape[2] != Cc:
        raise RuntimeError(f"square_distance: channel mismatch {Cc} vs {dst.shape[2]}")
    s, d = f32c(src), f32c(dst)
    out = torch.empty((B, N, M), dtype=torch.float32, device=src.device)
    check(lib().l3d_square_distance(ptr(s), ptr(d), B, N, M, Cc, ptr(out), stream_ptr()), "l3d_square_distance")   # any C
    return out


def index_points(points, idx):
    """reference: utils/model_common_utils.py:40-56.  points [B,N,C], idx [B,S] or [B,S,K] (int64)
    -> [B,S,C] / [B,S,K,C]."""
    require_gpu(points, idx)
    if torch.is_grad_enabled() and points.requires_grad:
        # the reference's advanced indexing is differentiable w.r.t. points (:50-55): gather kernel forward, deterministic
        # scatter-add backward
        if points.dtype == torch.float32 and points.dim() == 3:
            from ..models import _rows
            return _rows.index_points(points, idx)
        B = points.shape[0]
        view_shape = [B] + [1] * (idx.dim() - 1)
        batch_indices = torch.arange(B, dtype=torch.long, device=points.device).view(view_shape).expand_as(idx)
        return points[batch_indices, idx.long(), :]
    B, N, Cc = points.shape
    p = f32c(points)
    ix = idx.to(torch.int64).contiguous().view(B, -1)
    S = ix.shape[1]
    out = torch.empty((B, S, Cc), dtype=torch.float32, device=points.device)
    check(lib().l3d_index_points(ptr(p), ptr(ix), B, N, Cc, S, ptr(out), stream_ptr()), "l3d_index_points")
    return out.view(*idx.shape, Cc)


def farthest_point_sample(xyz, npoint, start_with_first_point=False):
    """reference: utils/model_common_utils.py:58-82.  xyz [B,N,3] -> centroids int64 [B,npoint].
    The first centroid is drawn with torch.randint exactly like the reference (:70-73)."""
    require_gpu(xyz)
    B, N, Cc = xyz.shape
    if Cc != 3:
        raise NotImplementedError("farthest_point_sample: C=3 only")
    x = f32c(xyz)
    if not start_with_first_point:
        start = torch.randint(0, N, (B,), dtype=torch.long).to(xyz.device)
    else:
        start = None
    cent = torch.empty((B, npoint), dtype=torch.int64, device=xyz.device)
    check(lib().l3d_farthest_point_sample(ptr(x), B, N, npoint, ptr(start), None, ptr(cent), stream_ptr()),
          "l3d_farthest_point_sample")
    return cent


def knn_point(k, pos1, pos2):
    """reference: utils/model_common_utils.py:84-100.  pos1 [B,N,3] searched, pos2 [B,M,3] queries
    -> (val [B,M,k] L2 distances ascending, idx int64 [B,M,k])."""
    require_gpu(pos1, pos2)
    B, N, Cc = pos1.shape
    M = pos2.shape[1]
    if Cc != 3:
        raise NotImplementedError("knn_point: C=3 only")
    if torch.is_grad_enabled() and (pos1.requires_grad or pos2.requires_grad):
        # distances differentiable like the reference's (:94-100): indices from the kernel, values re-formed by torch
        with torch.no_grad():
            _, idx = knn_point(k, pos1, pos2)
        nb = torch.gather(pos1.unsqueeze(1).expand(B, M, N, 3), 2, idx.unsqueeze(-1).expand(B, M, k, 3))
        return torch.sqrt(torch.sum((nb - pos2.unsqueeze(2)) ** 2, -1)), idx
    p1, p2 = f32c(pos1), f32c(pos2)
    val = torch.empty((B, M, k), dtype=torch.float32, device=pos1.device)
    idx = torch.empty((B, M, k), dtype=torch.int64, device=pos1.device)
    check(lib().l3d_knn_point(k, ptr(p1), ptr(p2), B, N, M, ptr(val), ptr(idx), stream_ptr()), "l3d_knn_point")
    return val, idx


def query_ball_point(radius, nsample, xyz, new_xyz, get_cnt=False, itself_indices=None):
    """reference: utils/model_common_utils.py:102-130 (get_cnt) and utils/ppfnet_util.py:96-131
    (itself_indices).  xyz [B,N,3], new_xyz [B,S,3] -> group_idx int64 [B,S,nsample] (+ cnt [B,S])."""
    require_gpu(xyz, new_xyz)
    B, N, Cc = xyz.shape
    S = new_xyz.shape[1]
    if Cc != 3:
        raise NotImplementedError("query_ball_point: C=3 only")
    x, q = f32c(xyz), f32c(new_xyz)
    idx = torch.empty((B, S, nsample), dtype=torch.int64, device=xyz.device)
    cnt = torch.empty((B, S), dtype=torch.int64, device=xyz.device) if get_cnt else None
    it = itself_indices.to(torch.int64).contiguous() if itself_indices is not None else None
    check(lib().l3d_query_ball_point(C.c_float(radius), nsample, ptr(x), ptr(q), B, N, S, ptr(it), ptr(idx),
                                     ptr(cnt), stream_ptr()), "l3d_query_ball_point")
    return (idx, cnt) if get_cnt else idx


class _GraphFeature(torch.autograd.Function):
    """[B,N,k,2C] = (x[idx] ; x) through l3d_graph_feature; backward = the adjoint of the two gathers (the
    reference's version is differentiable through its advanced indexing, model_common_utils.py:146-154, and
    PRNet's layers 2-4 need that gradient: their graph features are built from learned features)."""

    @staticmethod
    def forward(ctx, x, idx):
        B, Cc, N = x.shape
        k = idx.shape[2]
        xt = _as_bn3(x) if Cc == 3 else f32c(x.transpose(2, 1))
        out = torch.empty((B, N, k, 2 * Cc), dtype=torch.float32, device=x.device)
        check(lib().l3d_graph_feature(ptr(xt), ptr(idx), B, N, Cc, k, ptr(out), stream_ptr()), "l3d_graph_feature")
        ctx.save_for_backward(idx)
        ctx.shape = (B, Cc, N)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        B, Cc, N = ctx.shape
        g = grad_out.contiguous()
        k = idx.shape[2]
        centre = g[..., Cc:].sum(dim=2).transpose(2, 1)                          # [B,C,N]
        # neighbour half: every point owns its sum, contributions added in ascending (n, j) order
        # (l3d_scatter_add_det, scatter_det.hip) -- deterministic, unlike index_add's fp32 atomics
        src = g[..., :Cc].permute(0, 3, 1, 2).reshape(B, Cc, N * k).contiguous()
        idx32 = idx.to(torch.int32).reshape(B, N * k).contiguous()
        dst = torch.empty((B, Cc, N), dtype=torch.float32, device=g.device)
        ws = torch.empty(lib().l3d_scatter_add_det_workspace_bytes(B, N, N * k), dtype=torch.uint8, device=g.device)
        check(lib().l3d_scatter_add_det(ptr(src), ptr(idx32), None, B, Cc, N, N * k, 1, ptr(ws), ptr(dst), stream_ptr()),
              "l3d_scatter_add_det")
        return dst + centre, None


def get_graph_feature(x, k=20, device=None):
    """reference: utils/model_common_utils.py:132-156.  x [B,C,N] -> [B,2C,N,k] as a permuted view of
    [B,N,k,2C] memory (the reference returns the same non-contiguous view): channels 0..C-1 are the
    neighbour's, C..2C-1 the centre's."""
    x = x.view(*x.size()[:3])
    require_gpu(x)
    idx = knn(x, k=k)
    return _GraphFeature.apply(x, idx).permute(0, 3, 1, 2)


__all__ = ["knn", "square_distance", "index_points", "farthest_point_sample", "knn_point",
           "query_ball_point", "get_graph_feature", "L3DError"]
