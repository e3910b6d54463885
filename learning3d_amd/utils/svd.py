"""Drop-in for learning3d/utils/svd.py on MI355X.

reference: utils/svd.py:5-59.  Two HIP launches: the score GEMM + softmax + weighted target sum as one
flash-style pass that never materialises the [B,N,N] scores (l3d_soft_correspondence, SURVEY.md 8(f)
rank 1), and everything after `src_corr` (centring, H, the per-item torch.svd / det / reflect loop with
its B host syncs, and t) as one kernel (l3d_kabsch).  With autograd enabled the score part runs through
torch ops (the fused kernel is forward-only).
"""
import math

import torch
import torch.nn as nn

from .._lib import check, f32c, lib, ptr, require_gpu, stream_ptr


def kabsch(src, src_corr):
    """src, src_corr [B,3,N] -> R [B,3,3], t [B,3]   (utils/svd.py:29-58)."""
    require_gpu(src, src_corr)
    s, c = f32c(src), f32c(src_corr)
    B, _, N = s.shape
    R = torch.empty((B, 3, 3), dtype=torch.float32, device=s.device)
    t = torch.empty((B, 3), dtype=torch.float32, device=s.device)
    check(lib().l3d_kabsch(ptr(s), ptr(c), B, N, ptr(R), ptr(t), None, stream_ptr()), "l3d_kabsch")
    return R, t


def soft_correspondence(src_embedding, tgt_embedding, tgt, scale=None):
    """src_emb [B,C,N], tgt_emb [B,C,M], tgt [B,3,M] -> src_corr [B,3,N]   (utils/svd.py:22-27);
    scale defaults to 1/sqrt(C)."""
    require_gpu(src_embedding, tgt_embedding, tgt)
    q, k, v = f32c(src_embedding), f32c(tgt_embedding), f32c(tgt)
    B, C, N = q.shape
    M = k.shape[2]
    assert k.shape[1] == C and v.shape == (B, 3, M)
    if scale is None:
        scale = 1.0 / math.sqrt(C)
    ws = torch.empty(lib().l3d_soft_correspondence_workspace_floats(B, N, M), dtype=torch.float32, device=q.device)
    out = torch.empty((B, 3, N), dtype=torch.float32, device=q.device)
    check(lib().l3d_soft_correspondence(ptr(q), ptr(k), ptr(v), B, C, N, M, float(scale), ptr(ws), ptr(out),
                                        stream_ptr()), "l3d_soft_correspondence")
    return out


def svd3x3_rotation(H):
    """H [B,3,3] -> R = V U^T with the det < 0 reflection fix   (utils/svd.py:38-49)."""
    require_gpu(H)
    h = f32c(H)
    R = torch.empty_like(h)
    check(lib().l3d_svd3x3_rotation(ptr(h), h.shape[0], ptr(R), stream_ptr()), "l3d_svd3x3_rotation")
    return R


class SVDHead(nn.Module):
    def __init__(self, emb_dims, input_shape="bnc"):
        super(SVDHead, self).__init__()
        self.emb_dims = emb_dims
        # kept so reference checkpoints / state_dicts load unchanged (utils/svd.py:9-10)
        self.reflect = nn.Parameter(torch.eye(3), requires_grad=False)
        self.reflect[2, 2] = -1
        self.input_shape = input_shape

    def forward(self, *input):
        src_embedding, tgt_embedding, src, tgt = input[0], input[1], input[2], input[3]
        batch_size = src.size(0)
        if self.input_shape == "bnc":
            src = src.permute(0, 2, 1)
            tgt = tgt.permute(0, 2, 1)
        d_k = src_embedding.size(1)
        fused = d_k % 16 == 0 and not (torch.is_grad_enabled() and
                                       (src_embedding.requires_grad or tgt_embedding.requires_grad or tgt.requires_grad))
        if fused:
            src_corr = soft_correspondence(src_embedding, tgt_embedding, tgt)
        else:
            scores = torch.matmul(src_embedding.transpose(2, 1).contiguous(), tgt_embedding) / math.sqrt(d_k)
            scores = torch.softmax(scores, dim=2)
            src_corr = torch.matmul(tgt, scores.transpose(2, 1).contiguous())
        R, t = kabsch(src, src_corr)
        return R, t.view(batch_size, 3)
